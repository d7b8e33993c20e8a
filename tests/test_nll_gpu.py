"""The "NLL parity" half of BASELINE.json's metric, end to end: the path's detections matched to the planted
ground truth of their inputs (EU:191-367: IoU >= 0.7 true positives) and scored with SR:68-74
(-log N(gt; mean, cov + 1e-2 I)), for both backends:

  (a) HIP eps-replay detections, scored with the HIP matching + NLL kernels (pod_match_groundtruth, pod_reg_nll);
  (b) the REFERENCE's own detections of the same inputs and draws (stored in the golden fixtures by oracle/make_golden.py),
      scored with the oracle's restatement of EU/SR (pinned against the reference's EU/SR in tests/test_eval_matching.py);
  (c) HIP native-RNG detections (the product mode), HIP scoring.

Bar: |NLL(a) - NLL(b)| <= 1e-3 with identical match counts; (c) differs from (a) only by the sampling noise of its own
draws (the exact check of the native mode against the oracle on ITS draws is tests/test_native_exact_gpu.py)."""
import os

import pytest
import torch

from oracle import pod_oracle as po
from pod_compare_amd import evaluation_utils as ev
from tests.helpers import GOLDEN, Golden
from tests.test_hip_parity import make_path

pytestmark = pytest.mark.gpu
COUNTS = ("true_positives", "duplicates", "false_positives", "false_negatives")


@pytest.mark.parametrize("name", ["full_cfg3_bayes_od_mc10_s1001", "cfg3_bayes_od_mc10_s31", "cfg2_bayes_od_regclsvar_s22",
                                  "standard_nms_regclsvar_s91", "bayes_od_ci_clsbayes_s71"])
def test_end_to_end_nll_parity(name):
    g = Golden(os.path.join(GOLDEN, name + ".npz"))
    ho = g.head_outputs()
    hd = ho.to("cuda")
    s = g.spec
    hp = make_path(ho, g.meta["topk"])
    image, out = tuple(g.meta["image"]), tuple(g.meta["out"])
    kw = dict(image_size=image, out_size=out, box_merge_mode=s.get("box_merge", "bayesian_inference"),
              cls_merge_mode=s.get("cls_merge", "max_score"))
    rep = hp.run(s["mode"], hd.cls, hd.delta, hd.cls_var, hd.reg_var, eps_fn=g.eps_source(), **kw)
    a = ev.score_against_planted([rep], [hd], image, out)
    b = po.score_against_planted([(g.t("pred_boxes"), g.t("pred_cls_probs"), g.t("pred_boxes_covariance"))], [ho], image, out)
    assert a["true_positives"] > 0
    assert [a[k] for k in COUNTS] == [b[k] for k in COUNTS]
    assert abs(a["nll"] - b["nll"]) <= 1e-3, (a, b)
    assert abs(a["mse"] - b["mse"]) <= 1e-3 * max(1.0, abs(b["mse"]))
    nat = hp.run(s["mode"], hd.cls, hd.delta, hd.cls_var, hd.reg_var, draw_id=1, **kw)
    c = ev.score_against_planted([nat], [hd], image, out)
    # other draws: a fused box can cross the IoU 0.7 line of its planted box (true positive <-> neither), nothing else moves
    assert c["true_positives"] + c["false_negatives"] + c["duplicates"] >= a["true_positives"] + a["false_negatives"]
    assert abs(c["true_positives"] - a["true_positives"]) <= 2 and c["false_positives"] == a["false_positives"]
    if [c[k] for k in COUNTS] == [a[k] for k in COUNTS]:
        # the NLL of a box moves with its 1000-sample covariance estimate (relative s.d. ~ sqrt(2/1000) = 4.5 % per entry)
        assert abs(c["nll"] - a["nll"]) <= 0.05 * max(1.0, abs(a["nll"])), (c, a)
