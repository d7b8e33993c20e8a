"""pod_run_image (one C call per image) against the same launch sequence issued call by call from Python: both run the
in-kernel Philox draws with the same seed, so every output must be bit-identical; plus the workspace invariants the
one-call path relies on (counters and bitmap left zeroed by the kernels that consume them)."""
import pytest
import torch

from pod_compare_amd import hip, hotpath, synthetic

pytestmark = pytest.mark.gpu

CASES = [("bayes_od", 6, True, 4), ("bayes_od", 1, True, 4), ("standard_nms", 1, False, 0), ("standard_nms", 1, True, 4),
         ("anchor_statistics", 1, False, 0), ("anchor_statistics", 1, True, 4), ("ensembles", 5, False, 0),
         ("mc_dropout_ensembles", 4, True, 4)]


@pytest.mark.parametrize("mode,runs,cls_var,D", CASES)
def test_one_call_equals_call_by_call(mode, runs, cls_var, D):
    ho = synthetic.planted_head_outputs((384, 512), runs, seed=3 + runs, num_boxes=10, with_cls_var=cls_var, with_reg_var=D > 0)
    hd = ho.to("cuda")
    params = hotpath.PathParams(num_classes=ho.num_classes, num_anchors=ho.num_anchors)
    outs = []
    for one_call in (True, False, True):
        hp = hotpath.HotPath(ho.shapes, ho.anchors, params, n_runs=runs, has_cls_var=cls_var, cov_dims=D, device="cuda")
        for _ in range(2):     # twice on the same workspace: the second image must find the counters / bitmap clean
            det = hp.run(mode, hd.cls, hd.delta, hd.cls_var, hd.reg_var, image_size=(380, 500), out_size=(720, 1280), one_call=one_call)
        torch.cuda.synchronize()
        assert int(hp.counters.abs().sum().item()) == 0
        if hp.maybe_bits is not None:
            assert int(hp.maybe_bits.abs().sum().item()) == 0
        outs.append(det)
    m = outs[0].count()
    assert m > 0
    for other in outs[1:]:
        assert other.count() == m
        for name in ("boxes", "cov", "scores", "classes", "probs", "records"):
            assert torch.equal(getattr(outs[0], name)[:m], getattr(other, name)[:m]), name


def test_invalid_requests_are_refused_not_launched():
    ho = synthetic.planted_head_outputs((160, 224), 1, seed=1, num_boxes=4, with_cls_var=False, with_reg_var=False)
    hd = ho.to("cuda")
    params = hotpath.PathParams(num_classes=ho.num_classes, num_anchors=ho.num_anchors)
    hp = hotpath.HotPath(ho.shapes, ho.anchors, params, n_runs=1, has_cls_var=False, cov_dims=0, device="cuda")
    with pytest.raises(ValueError):
        hp.run("no_such_mode", hd.cls, hd.delta, image_size=(150, 210), out_size=(150, 210))
    with pytest.raises(hip.PodError):                      # BayesOD needs covariances (PI:562-636)
        hp.run("bayes_od", hd.cls, hd.delta, image_size=(150, 210), out_size=(150, 210))
    lv = hp._levels(hd.cls, hd.delta, None, None, None)
    out = hp.new_detections((150, 210))
    d = hip.PodDetections(*[out.ptr(n) for n in ("boxes", "cov", "scores", "classes", "probs", "records", "n_det")])
    assert hp.lib.pod_run_image(hp.cfg, lv, hp.ws, 7, 0, 0, 150, 210, 150, 210, d, hip.current_stream()) == -1
    assert hp.lib.pod_run_image(hp.cfg, lv, hp.ws, hip.POD_MODE_BAYES_OD, 0, 0, 150, 210, 150, 210, d, hip.current_stream()) == -1
    assert hp.lib.pod_run_image(hp.cfg, lv, hp.ws, 0, 0, 0, 0, 210, 150, 210, d, hip.current_stream()) == -1
    assert hp.lib.pod_run_image(hp.cfg, lv, None, 0, 0, 0, 150, 210, 150, 210, d, hip.current_stream()) == -1
