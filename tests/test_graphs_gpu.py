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


def close(a, b):
    # MIOpen's backbone kernels accumulate with atomics: two evaluations of the same image agree to rounding, not bit for bit
    for x, y in zip(a if isinstance(a, list) else tensors(a), b if isinstance(b, list) else tensors(b)):
        assert x.shape == y.shape
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


def test_dropout_forwards_are_not_captured():
    """A replay would repeat the first image's dropout masks (the Philox counter offsets are launch arguments)."""
    m = build(dropout_rate=0.2).enable_graphs()
    f = torch.randint(0, 256, (3, 128, 160), dtype=torch.uint8, device="cuda")
    a = m(f, num_mc_dropout_runs=3)
    b = m(f, num_mc_dropout_runs=3)
    assert not m._graphs and not torch.equal(a.cls[0], b.cls[0])
    m(f)                                                   # dropout off: captured
    assert len(m._graphs) == 1
