"""HIP-graph replay of the model forward (modeling.ProbabilisticRetinaNet.enable_graphs): the launch sequence of one image behind
one host call (AN:88-98 is a Python loop; the single-run configurations are host-bound without it)."""
import pytest
import torch

from pod_compare_amd import modeling

pytestmark = pytest.mark.gpu


def build(**kw):
    torch.manual_seed(0)
    m = modeling.ProbabilisticRetinaNet(cls_var_loss="loss_attenuation", cls_var_num_samples=10, bbox_cov_loss="negative_log_likelihood", **kw).cuda().eval()
    modeling.fold_frozen_bn(m)
    for q in m.parameters():
        q.requires_grad_(False)
    return m


def tensors(ho):
    return list(ho.cls) + list(ho.delta) + list(ho.cls_var) + list(ho.reg_var)


# Since FPN's p6 / p7 left MIOpen (round 5, conv1x1.Conv3x3S2) nothing in the channels-last forward accumulates with atomics: two evaluations
# of an image agree BIT FOR BIT (fixed-order partial sums, masks keyed by element index, abs-max records made by order-free maxima).  With a
# MIOpen convolution in the path (POD_HIP_P6P7=0, POD_CL_BACKBONE=0, POD_HIP_STEM=0) they agree to rounding only.
from pod_compare_amd import modeling as _modeling
EXACT = _modeling.HIP_P6P7 and _modeling.CL_BACKBONE and _modeling.HIP_STEM


def close(a, b):
    for x, y in zip(a if isinstance(a, list) else tensors(a), b if isinstance(b, list) else tensors(b)):
        assert x.shape == y.shape
        if EXACT:
            assert torch.equal(x, y)
        else:
            assert float((x - y).abs().max()) <= 2e-4 * max(1.0, float(y.abs().max()))


def test_graph_replay_equals_the_eager_forward_for_every_image_and_stream():
    m = build()
    g = torch.Generator(device="cuda").manual_seed(3)
    frames = [torch.randint(0, 256, (3, 200, 300), dtype=torch.uint8, device="cuda", generator=g) for _ in range(3)]
    eager = [m(f) for f in frames]
    eager = [[t.clone() for t in tensors(e)] for e in eager]
    m.enable_graphs()
    streams = [torch.cuda.current_stream(), torch.cuda.Stream()]
    for rep in range(2):
        for i, f in enumerate(frames):
            s = streams[(i + rep) % 2]
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                out = m(f)
                close(out, eager[i])                       # (the comparison is enqueued on s before the next replay there)
    assert len(m._graphs) == 2                             # one graph per stream, replayed for every image of that shape
    other = torch.randint(0, 256, (3, 160, 224), dtype=torch.uint8, device="cuda", generator=g)
    out = m(other)
    assert out.shapes[0] == (20, 28) and len(m._graphs) == 3
    m.enable_graphs(False)
    close(m(other), out)


def test_mc_dropout_forwards_replay_with_fresh_masks():
    """A captured MC-dropout forward must not repeat its masks: seed / offset are constants of the captured launches, so the Philox key
    folds in a device word -- one PER GRAPH -- that the graph itself bumps at the start of every replay.  Same epoch -> same bits; next
    epoch -> other masks; and the masks are real dropout (a fifth of the first activation's copies zeroed)."""
    m = build(dropout_rate=0.2).enable_graphs()
    f = torch.randint(0, 256, (3, 128, 160), dtype=torch.uint8, device="cuda")
    a = [t.clone() for t in tensors(m(f, num_mc_dropout_runs=3))]
    assert len(m._graphs) == 1
    epoch = m.graph_epoch()
    e1 = int(epoch.item())
    assert e1 == (1 << 32) + 1                                         # graph serial 1, first replay
    assert int(m.head._epoch.item()) == 0                              # the model's shared word is not what graphs write
    b = [t.clone() for t in tensors(m(f, num_mc_dropout_runs=3))]
    assert int(epoch.item()) == e1 + 1 and len(m._graphs) == 1
    assert not torch.equal(a[0], b[0])                                 # fresh masks
    assert not torch.equal(b[0][0], b[0][1])                           # and independent ones per run
    epoch.fill_(e1 - 1)                                                # the replay bumps it to e1 again
    c = [t.clone() for t in tensors(m(f, num_mc_dropout_runs=3))]
    close(a, c)                                                        # same epoch -> the same masks -> the same bits
    assert float((a[0] - b[0]).abs().max()) > 1e-3                      # while another epoch moves the outputs visibly
    m(f)                                                               # the dropout-free forward of the same model: its own graph
    assert len(m._graphs) == 2


def test_two_streams_replaying_mc_dropout_graphs_are_deterministic_and_never_share_masks():
    """apply_net runs two streams, each replaying its own graph, concurrently.  Each graph owns its epoch word (no read-modify-write race
    on a shared one), so the masks of replay i of graph j are a function of (seed, j, i): two identical sessions give identical outputs,
    whatever the GPU's timing, and no two forwards of a session share masks."""
    f = torch.randint(0, 256, (3, 128, 160), dtype=torch.uint8, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))

    def session():
        m = build(dropout_rate=0.2).enable_graphs()
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        outs = []
        for rep in range(4):
            for s in streams:                       # both streams busy at once: replays of the two graphs overlap on the device
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    o = m(f, num_mc_dropout_runs=3)
                    outs.append(o.cls[0].clone())
        torch.cuda.synchronize()
        assert len(m._graphs) == 2
        ep = sorted(int(e[3].item()) for e in m._graphs.values())
        assert ep == [(1 << 32) + 4, (2 << 32) + 4], ep
        return outs

    a, b = session(), session()
    close(a, b)
    for i in range(len(a)):
        for j in range(i + 1, len(a)):
            assert float((a[i] - a[j]).abs().max()) > 1e-3, (i, j)                        # fresh masks everywhere


def test_graphs_are_dropped_when_the_parameters_change():
    """A graph holds pointers to filters transformed at capture time: after load_state_dict / an in-place write / a move, the next forward
    must run on the new weights (it silently replayed the stale ones)."""
    m = build().enable_graphs()
    f = torch.randint(0, 256, (3, 128, 160), dtype=torch.uint8, device="cuda")
    a = [t.clone() for t in tensors(m(f))]
    assert len(m._graphs) == 1
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    sd["head.cls_score.weight"] = sd["head.cls_score.weight"] * 3.0
    m.load_state_dict(sd)
    b = [t.clone() for t in tensors(m(f))]
    assert len(m._graphs) == 1                                          # dropped and re-captured
    m.enable_graphs(False)
    want = tensors(m(f))
    close(b, want)
    assert float((a[0] - b[0]).abs().max()) > 1e-4
    m.enable_graphs()
    m(f)
    with torch.no_grad():
        m.head.bbox_pred.bias.add_(0.5)                                 # in-place write: the tensors' version counters
    c = [t.clone() for t in tensors(m(f))]
    L = len(m(f).cls)                                                   # tensors(): cls levels, then delta levels, ...
    assert float((c[L] - b[L]).abs().min()) > 0.4
    close([c[0]], [b[0]])
    gen = modeling._PARAM_GENERATION[0]
    m.float()                                                           # _apply: generation counter
    assert modeling._PARAM_GENERATION[0] > gen
    close([t.clone() for t in tensors(m(f))], c)
    # ... and on a SUBMODULE (a counter on the root alone missed it): the head's storage moves, the graph must not replay
    # against the freed filters
    fp = m._param_fingerprint()
    m.head.cuda()                                                       # (what model.head.to(...) calls: nn.Module._apply on the SUBMODULE)
    m.head._apply(lambda t: t.clone())                                  # new storage for every parameter of the head, versions untouched
    assert m._param_fingerprint() != fp
    with torch.no_grad():
        m.head.cls_score.weight.data = m.head.cls_score.weight.data * 2.0          # (.data: no version bump either)
        modeling.bump_param_generation()
    d = [t.clone() for t in tensors(m(f))]
    m.enable_graphs(False)
    close(d, tensors(m(f)))
    assert float((d[0] - c[0]).abs().max()) > 1e-4


def test_float_frames_of_different_range_replay_with_their_own_abs_max():
    """the warm-up forwards of a capture attach an abs-max record to the static input; a capture that re-used it would record no
    pod_absmax, and every replay would scale the stem's f16 split by the FIRST frame's range -- a later, louder float frame overflows f16.
    Float32 frames whose ranges differ by 40 x, replayed through one graph, must equal their eager forwards bit for bit."""
    m = build()
    g = torch.Generator(device="cuda").manual_seed(5)
    base = torch.randint(0, 256, (3, 160, 224), dtype=torch.uint8, device="cuda", generator=g).float()
    frames = [base * 0.05, base, base * 2.0, base * 0.05]
    eager = [[t.clone() for t in tensors(m(f))] for f in frames]
    assert all(bool(torch.isfinite(t).all()) for e in eager for t in e)
    m.enable_graphs()
    for rep in range(2):
        for f, e in zip(frames, eager):
            out = m(f)
            assert all(bool(torch.isfinite(t).all()) for t in tensors(out))
            close(out, e)
    assert len(m._graphs) == 1


def test_a_shape_is_captured_only_after_it_came_back():
    m = build().enable_graphs()
    m.graph_after_seen = 2
    f = torch.randint(0, 256, (3, 128, 160), dtype=torch.uint8, device="cuda")
    a = [t.clone() for t in tensors(m(f))]
    m(f)
    assert not m._graphs
    close(m(f), a)
    assert len(m._graphs) == 1


def test_parity_mode_and_the_miopen_head_path_are_not_captured():
    m = build(dropout_rate=0.2).enable_graphs()
    f = torch.randint(0, 256, (3, 128, 160), dtype=torch.uint8, device="cuda")
    old = modeling.WINO_HEAD
    modeling.WINO_HEAD = False
    try:
        m(f, num_mc_dropout_runs=3)                                    # MIOpen head path: its masks' offsets are launch arguments only
        assert not m._graphs
    finally:
        modeling.WINO_HEAD = old
