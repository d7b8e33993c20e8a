"""pod_nms_cluster through the C ABI against the oracle's restatement of detectron2 batched_nms -> torchvision nms
(call sites PI:554-560, IU:31-36, IU:83-89, IU:269-274): keep lists must be identical, element by element.

The kernel sweeps every class on its own workgroup when the coordinate trick really separates the classes and falls
back to one workgroup otherwise; the cases below hit both routes, the sort paths (rank sort <= 1024 members, bitonic
above), capacity > n, and the corner where the trick lets a box of one class suppress a box of another."""
import pytest
import torch

from oracle import pod_oracle as po
from pod_compare_amd import hip

pytestmark = pytest.mark.gpu


def run_kernel(boxes, scores, classes, num_classes=7, thr=0.5, max_det=100, cap=None, scratch=None):
    lib = hip.load()
    cfg = hip.PodConfig()
    cfg.max_detections, cfg.nms_thresh, cfg.num_classes = max_det, thr, num_classes
    n = boxes.shape[0]
    cap = max(n, 1) if cap is None else cap
    pad = lambda t: torch.cat([t, torch.zeros((cap - n,) + tuple(t.shape[1:]), dtype=t.dtype)]).cuda().contiguous()
    b, s, c = pad(boxes.float()), pad(scores.float()), pad(classes.to(torch.int32))
    keep = torch.full((hip.POD_MAX_DETECTIONS,), -1, dtype=torch.int32, device="cuda")
    n_keep = torch.full((1,), -7, dtype=torch.int32, device="cuda")
    nt = torch.tensor([n], dtype=torch.int32, device="cuda")
    if scratch is None:
        scratch = torch.zeros(lib.pod_nms_scratch_bytes(cap), dtype=torch.uint8, device="cuda")      # zeroed once (include/pod_mi355x.h)
    P = hip.ptr
    hip.check(lib.pod_nms_cluster(cfg, P(nt), cap, P(b), P(s), P(c), P(keep), P(n_keep), P(scratch), hip.current_stream()),
              "pod_nms_cluster")
    torch.cuda.synchronize()
    k = int(n_keep.item())
    flag = int(scratch[:4].view(torch.int32).item()) if n > 0 else 0
    return keep[:k].cpu().long(), flag


def clustered(n, seed, frame, num_classes=3):
    """n boxes in ~n/12 tight clusters (heavy suppression), some hanging out of the frame."""
    g = torch.Generator().manual_seed(seed)
    nc = max(1, n // 12)
    cc = torch.rand(nc, 2, generator=g) * torch.tensor(frame, dtype=torch.float32)
    cw = torch.rand(nc, 2, generator=g) * 200 + 20
    a = torch.randint(0, nc, (n,), generator=g)
    c = cc[a] + torch.randn(n, 2, generator=g) * 6
    wh = cw[a] * (1 + 0.1 * torch.randn(n, 2, generator=g)).clamp(0.5, 1.5)
    boxes = torch.cat([c - wh / 2, c + wh / 2], 1).contiguous()
    scores = torch.rand(n, generator=g)
    classes = (a % num_classes).to(torch.int32)
    return boxes, scores, classes


@pytest.mark.parametrize("frame", [(1344.0, 768.0), (600.0, 600.0), (300.0, 900.0)], ids=["landscape", "square", "portrait"])
@pytest.mark.parametrize("n", [1, 5, 64, 65, 200, 317, 1000, 1024, 1025, 3000, 8192])
def test_keep_list_equals_batched_nms(n, frame):
    boxes, scores, classes = clustered(n, 1000 + n, frame)
    ref = po.class_aware_nms(boxes, scores, classes.long(), 0.5)[:100]
    got, _ = run_kernel(boxes, scores, classes, cap=min(hip.POD_MAX_CANDIDATES, n + 37))
    assert torch.equal(got, ref)


def test_many_members_of_one_class_use_the_bitonic_path_per_class():
    boxes, scores, classes = clustered(6000, 77, (1344.0, 768.0), num_classes=2)     # ~3000 members per class
    ref = po.class_aware_nms(boxes, scores, classes.long(), 0.5)[:100]
    got, flag = run_kernel(boxes, scores, classes, num_classes=2)
    assert flag == 0
    assert torch.equal(got, ref)


def test_max_detections_truncation_and_threshold():
    boxes, scores, classes = clustered(2000, 5, (1344.0, 768.0), num_classes=7)
    for thr, md in [(0.3, 10), (0.7, 128), (0.5, 1)]:
        ref = po.class_aware_nms(boxes, scores, classes.long(), thr)[:md]
        got, _ = run_kernel(boxes, scores, classes, thr=thr, max_det=md)
        assert torch.equal(got, ref), (thr, md)


def test_empty_list():
    got, _ = run_kernel(torch.zeros(0, 4), torch.zeros(0), torch.zeros(0, dtype=torch.int32), cap=16)
    assert got.numel() == 0


def test_coordinate_trick_crosstalk_between_classes_is_reproduced():
    """boxes + class * (max + 1) does not separate classes when boxes lie outside the frame: a class-1 box in the
    negative quadrant lands on top of a class-0 box at the far corner and is suppressed by it (torchvision does this
    too).  The kernel must detect the pair and take the single-workgroup route."""
    boxes = torch.tensor([[901.0, 901.0, 1000.0, 1000.0],      # class 0, at the max corner; unit = 1001
                          [-100.0, -100.0, 0.0, 0.0],          # class 1 -> shifted [901, 901, 1001, 1001], IoU 0.98 with box 0
                          [300.0, 300.0, 400.0, 400.0],        # class 1, unaffected
                          [-100.0, -100.0, 0.0, 0.0]])         # class 0 twin of box 1: no overlap with box 0
    scores = torch.tensor([0.9, 0.8, 0.7, 0.6])
    classes = torch.tensor([0, 1, 1, 0], dtype=torch.int32)
    ref = po.class_aware_nms(boxes, scores, classes.long(), 0.5)
    assert ref.tolist() == [0, 2, 3]                            # box 1 is gone although it shares no class with box 0
    got, flag = run_kernel(boxes, scores, classes, num_classes=2)
    assert flag == 1
    assert torch.equal(got, ref)
    # surrounded by a few thousand ordinary boxes the answer must still be the reference's
    many, ms, mc = clustered(3000, 9, (900.0, 900.0), num_classes=2)
    many = many.clamp(-50.0, 950.0)
    b2, s2, c2 = torch.cat([boxes, many]), torch.cat([scores + 1.0, ms]), torch.cat([classes, mc])
    ref2 = po.class_aware_nms(b2, s2, c2.long(), 0.5)[:100]
    got2, flag2 = run_kernel(b2, s2, c2, num_classes=2)
    assert flag2 == 1 and 1 not in got2.tolist()
    assert torch.equal(got2, ref2)


def test_out_of_frame_boxes_without_crosstalk_stay_on_the_class_parallel_route():
    boxes, scores, classes = clustered(1500, 21, (1344.0, 768.0), num_classes=7)     # boxes reach below 0 and past the frame
    assert boxes.min() < -20
    _, flag = run_kernel(boxes, scores, classes)
    assert flag == 0


def test_class_id_out_of_range_takes_the_single_workgroup_route():
    boxes, scores, classes = clustered(300, 3, (1344.0, 768.0), num_classes=3)
    classes[7] = 9                                                                   # >= num_classes
    ref = po.class_aware_nms(boxes, scores, classes.long(), 0.5)[:100]
    got, flag = run_kernel(boxes, scores, classes, num_classes=3)
    assert flag == 1
    assert torch.equal(got, ref)


def test_early_stop_of_the_class_sweeps_is_exact_and_survives_scratch_reuse():
    """Every class fills up (thousands of barely overlapping boxes, 7 classes): a class sweep stops as soon as the other
    classes have published enough higher-scoring survivors (k4_nms.hip).  keep[:100] must still be the reference's, on every
    one of several calls that reuse ONE scratch buffer (the published scores of earlier calls carry an older generation tag)."""
    lib = hip.load()
    scratch = torch.zeros(lib.pod_nms_scratch_bytes(8192), dtype=torch.uint8, device="cuda")
    for seed, n, K in ((1, 4594, 7), (2, 4594, 7), (3, 700, 7), (4, 8192, 15), (5, 4594, 2), (6, 50, 7)):
        g = torch.Generator().manual_seed(seed)
        xy = torch.rand(n, 2, generator=g) * torch.tensor([1300.0, 740.0])
        wh = torch.rand(n, 2, generator=g) * 30 + 4
        boxes = torch.cat([xy, xy + wh], 1)
        scores = torch.rand(n, generator=g)
        if seed == 2:
            scores = (scores * 16).floor() / 16                    # heavy score ties across classes
        classes = torch.randint(0, K, (n,), generator=g).to(torch.int32)
        ref = po.class_aware_nms(boxes, scores, classes.long(), 0.5)[:100]
        got, flag = run_kernel(boxes, scores, classes, num_classes=K, cap=8192, scratch=scratch)
        assert flag == 0 and torch.equal(got, ref), seed
