"""popnet_amd.metrics (PCK / mAP restatement) against known answers produced by the
reference's own util/eval_pck.py and util/eval_mAP.py (tests/golden/make_golden.py::golden_metrics)."""
import importlib.util
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import popnet_amd  # noqa: E402,F401
from popnet_amd import metrics as M  # noqa: E402
from popnet_amd.config import KEYPOINTS  # noqa: E402

GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "metrics.json")))
_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
make_golden = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(make_golden)


def close(a, b, tol=1e-12):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return a.shape == b.shape and bool(np.all((np.abs(a - b) <= tol) | (np.isnan(a) & np.isnan(b))))


@pytest.mark.parametrize("case", GOLD["cases"], ids=lambda c: "seed%d" % c["seed"])
def test_pck_and_map_equal_reference(case):
    p2, p3, pc, g2, g3 = make_golden.metric_case(case["seed"])
    d2, k2 = M.eval_human_dataset_2d_PCKh(p2, g2, head_id=0, neck_id=1, num_joints=15, iou_th=0.5)
    d3, k3 = M.eval_human_dataset_3d(p2, g2, p3, g3, num_joints=15, dist_th=0.1, iou_th=0.5)
    assert close(k2, case["pck2d"]) and close(d2, case["err2d"])
    assert close(k3, case["pck3d"]) and close(d3, case["err3d"])
    a2 = M.eval_ap_mpii_v2(p2, pc, g2, [], 0, 1, KEYPOINTS, 0.5, verbose=False)
    a3 = M.eval_ap_3D(p3, pc, g3, [], KEYPOINTS, 0.1, verbose=False)
    assert close(a2, case["ap2d"], 1e-9) and close(a3, case["ap3d"], 1e-9)
    md = M.match_humans_3d(p2[0], g2[0], p3[0], g3[0], 0.5)
    assert close(np.array(md), np.array(case["match3d_img0"]))


def test_perfect_predictions_score_100():
    _, _, _, g2, _ = make_golden.metric_case(GOLD["perfect"]["seed"])
    a2 = M.eval_ap_mpii_v2(g2, [], g2, [], 0, 1, KEYPOINTS, 0.5, verbose=False)
    _, k2 = M.eval_human_dataset_2d_PCKh(g2, g2, head_id=0, neck_id=1)
    assert close(a2, GOLD["perfect"]["ap2d"], 1e-9) and close(k2, GOLD["perfect"]["pck2d"])
    assert abs(a2[-1] - 100.0) < 1e-9 and all(abs(v - 1.0) < 1e-12 for v in k2)


def test_edge_cases():
    g2 = [[[[10.0 + j, 20.0 + 3 * j] for j in range(15)]]]
    # no predictions at all: every distance -1, PCK 0, AP 0
    d, k = M.eval_human_dataset_2d_PCKh([[]], g2, 0, 1)
    assert all(v == 0 for v in k) and all(np.isnan(v) for v in d)
    with np.errstate(all="ignore"):
        ap = M.eval_ap_mpii_v2([[]], [[]], g2, [], 0, 1, KEYPOINTS, verbose=False)
    assert np.all(ap == 0)
    # a predicted person with no valid joint empties the image's prediction boxes (reference early return)
    bad = [[[-1, -1]] * 15]
    good = g2[0]
    d = M.match_humans_2d(bad + good, good)
    assert np.all(np.asarray(d) == -1)
    # bbox_ious of nothing
    assert M.bbox_ious(np.zeros((2, 4)), np.zeros((0, 4))).tolist() == [[-1.0], [-1.0]]


def test_cli_round_trip(tmp_path):
    p2, p3, pc, g2, g3 = make_golden.metric_case(GOLD["cases"][0]["seed"])
    labels = {"intrinsics": {"fx": 1, "fy": 1, "cx": 0, "cy": 0}}
    for i in range(len(g2)):
        labels["f%03d.npy" % i] = [{"2d_joints": a, "3d_joints": b} for a, b in zip(g2[i], g3[i])]
    res = {"human_pred_set_2d": p2, "human_pred_set_3d": p3, "human_pred_set_part_conf": pc}
    json.dump(labels, open(tmp_path / "labels.json", "w"))
    json.dump(res, open(tmp_path / "results.json", "w"))
    out = M.evaluate_mp_human_3d(str(tmp_path / "labels.json"), str(tmp_path / "results.json"), verbose=False)
    assert close(out["ap2d"], GOLD["cases"][0]["ap2d"], 1e-9) and close(out["pck3d"], GOLD["cases"][0]["pck3d"])


# ---------------------------------------------------------------------------------------------
# dataset ingest / result schema (host logic, no GPU)
# ---------------------------------------------------------------------------------------------
def test_dataset_listing_dtype_and_batching(tmp_path):
    from popnet_amd import dataset, _lib
    labels = {"b.npy": [{"2d_joints": [[1, 2]] * 15, "3d_joints": [[1, 2, 3]] * 15}], "intrinsics": {"fx": 2, "fy": 3, "cx": 4, "cy": 5},
              "a.npy": [], "c.npy": []}
    json.dump(labels, open(tmp_path / "labels.json", "w"))
    np.save(tmp_path / "b.npy", np.full((6, 4), 1.5, np.float16))
    np.save(tmp_path / "a.npy", np.full((6, 4), 2.5, np.float16))
    np.save(tmp_path / "c.npy", np.full((6, 4), 3.5, np.float64))
    fr = dataset.MP3DHPFrames(str(tmp_path), str(tmp_path / "labels.json"))
    assert fr.ids == ["b.npy", "a.npy", "c.npy"] and len(fr) == 3            # dict order, 'intrinsics' skipped
    assert fr.intrinsics == {"fx": 2, "fy": 3, "cx": 4, "cy": 5}
    assert fr.load(0).dtype == np.float16 and fr.load(2).dtype == np.float32   # float64 narrowed like Cvt2ndarray
    got = list(fr.batches([0, 1], 2))
    assert got[0][0] == [0, 1] and got[0][1].shape == (2, 6, 4)
    assert len(list(fr.batches([0, 1], 2, drop_last=True))) == 1 and len(list(fr.batches([0], 2, drop_last=True))) == 0
    with pytest.raises(_lib.PopnetError):
        list(fr.batches([1, 2], 2))                                            # mixed dtypes in one batch
    g2, g3 = fr.ground_truth()
    assert len(g2) == 3 and g2[0][0][0] == [1, 2] and g3[1] == []


def test_pose_records_to_result_schema():
    from popnet_amd import dataset, _lib
    recs = np.zeros(2, dtype=_lib.POSE_FRAME_DTYPE)
    recs[0]["n_persons"] = 2
    recs[0]["person_joint"][:2] = -1
    recs[0]["person_joint"][0, :3] = [4, 5, 6]
    recs[0]["joints_2d"][0, 0] = [10.5, 20.25]
    recs[0]["joints_3d"][0, 0] = [0.1, 0.2, 3.0]
    recs[0]["part_conf"][0, 0] = 0.75
    d = dataset.pose_records_to_lists(recs)
    assert len(d["human_pred_set_2d"]) == 2 and len(d["human_pred_set_2d"][0]) == 2 and d["human_pred_set_2d"][1] == []
    assert d["human_pred_set_2d"][0][0][0] == [10.5, 20.25] and d["human_pred_set_3d"][0][0][0] == [0.1, 0.2, 3.0]
    assert d["human_pred_set_visibility"][0][0][:4] == [1, 1, 1, 0] and d["human_pred_set_part_conf"][0][0][0] == 0.75
    json.dumps(d)                                                              # plain lists / floats only
    recs[1]["status"] = 1
    with pytest.raises(_lib.PopnetError):
        dataset.pose_records_to_lists(recs)
