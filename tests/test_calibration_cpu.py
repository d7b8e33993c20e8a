"""pod_compare_amd.compute_calibration_errors against the reference's own offline_evaluation/compute_calibration_errors.py::main, run on
seeded partitions by oracle/make_golden_calib.py (tests/golden/calib_errors.npz): the four errors the reference computes itself are
pinned at full precision; of the fifth -- `calibration.get_calibration_error`, a third-party package absent here -- the ARGUMENTS the
reference hands over are pinned, its arithmetic is a restatement (parity unpinned) checked on properties only."""
import os

import numpy as np
import torch

from pod_compare_amd.compute_calibration_errors import calibration_errors, format_table, marginal_calibration_error

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "calib_errors.npz")


def load():
    z = np.load(GOLDEN)
    matched = {}
    for key in z.files:
        if "." in key:
            part, name = key.split(".", 1)
            matched.setdefault(part, {})[name] = torch.from_numpy(z[key])
    return z, matched


def test_errors_equal_the_references_main_on_the_same_partitions():
    z, matched = load()
    k = int(z["num_classes"])
    torch.manual_seed(0)                                         # the generator seeded torch the same way before the reference's randperm calls
    res = calibration_errors(matched, {i: i for i in range(k)}, marginal_fn=lambda p, l: 0.12345)
    got = [res["reg_expected_calibration_error"], res["reg_maximum_calibration_error"], res["cls_minimum_uncertainty_error"], res["reg_minimum_uncertainty_error"]]
    assert np.allclose(got, z["values"], rtol=1e-6, atol=0.0), (got, z["values"])
    probs, labels = res["cls_marginal_inputs"]                   # what the reference hands to calibration.get_calibration_error
    assert np.array_equal(probs, z["cal_probs"]) and np.array_equal(labels, z["cal_labels"])
    assert ["%.4f" % v for v in [res["cls_marginal_calibration_error"]] + got] == [str(s) for s in z["row"]]      # the row the reference prints
    assert "0.4117" in format_table(res)


def test_fixture_regenerates_from_the_reference():
    from oracle.refimport import reference_available
    if not reference_available():
        import pytest
        pytest.skip("reference tree absent")
    from oracle import make_golden_calib as gen
    z, _ = load()
    cap = gen.run_reference(gen.seeded_matched_results(k=int(z["num_classes"])), int(z["num_classes"]))
    assert cap["row"] == [str(s) for s in z["row"]]
    assert np.array_equal(cap["cal_probs"], z["cal_probs"])


def test_marginal_calibration_error_properties():
    """The restated package function: a calibrated score has (debiased) error ~0, a constant shift by d has error ~d, duplicated scores
    take the discrete path, bad labels are rejected."""
    rng = np.random.default_rng(0)
    p = rng.uniform(0.05, 0.95, 200000)
    y = (rng.uniform(size=p.size) < p).astype(np.int64)
    assert marginal_calibration_error(p, y) < 0.01
    shifted = np.clip(p + 0.1, 0.0, 1.0)
    assert abs(marginal_calibration_error(shifted, y) - 0.1) < 0.015
    q = np.repeat([0.2, 0.7], 5000)
    yq = np.concatenate([(rng.uniform(size=5000) < 0.4), (rng.uniform(size=5000) < 0.7)]).astype(np.int64)
    assert abs(marginal_calibration_error(q, yq) - (0.5 * 0.2 ** 2) ** 0.5) < 0.02
    import pytest
    with pytest.raises(ValueError):
        marginal_calibration_error(p, y.astype(np.float64))
    with pytest.raises(ValueError):
        marginal_calibration_error(p, y + 1)


def test_driver_runs_on_a_result_file_through_the_oracle_matching(tmp_path):
    """Result file + ground truth -> matching (the CPU oracle's, EU:191-367) -> the five errors: finite, in range, printable; the same
    partitions through `calibration_errors` twice with the same seed give the same numbers."""
    import json
    from oracle import pod_oracle as po
    from pod_compare_amd import evaluation_utils as ev
    from pod_compare_amd.compute_probabilistic_metrics import BDD_DATASET_ID_TO_CONTIGUOUS
    z = np.load(os.path.join(os.path.dirname(GOLDEN), "eval_metrics.npz"))
    predicted, gt = json.loads(str(z["predicted_json"])), json.loads(str(z["gt_json"]))
    pred, g = ev.eval_predictions_preprocess(predicted, 0.0, device="cpu"), ev.eval_gt_preprocess(gt, device="cpu")
    matched = po.match_predictions_to_groundtruth(pred["predicted_boxes"], pred["predicted_cls_probs"], pred["predicted_covar_mats"], g["gt_boxes"], g["gt_cat_idxs"], 0.1, 0.7)
    out = []
    for _ in range(2):
        torch.manual_seed(3)
        out.append(calibration_errors(matched, BDD_DATASET_ID_TO_CONTIGUOUS))
    for key in ("cls_marginal_calibration_error", "reg_expected_calibration_error", "reg_maximum_calibration_error", "cls_minimum_uncertainty_error", "reg_minimum_uncertainty_error"):
        assert out[0][key] == out[1][key] and np.isfinite(out[0][key]) and 0.0 <= out[0][key] <= 1.0, key
    assert "Reg Maximum Calibration Error" in format_table(out[0])
