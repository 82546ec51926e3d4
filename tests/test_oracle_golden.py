"""CPU: the oracle (oracle/) must reproduce the golden vectors that tests/golden/make_golden.py
produced by running the REFERENCE's own code.  This is what pins the oracle (prompt section 3)."""
import numpy as np
import pytest
import torch

from helpers import (CPP_CASES, PAFPROCESS_CASES, YOLO_ANCHORS, all_parse_case_names, coco_case, humans_to_array,
                     parse_case_inputs, state_dict_from_keys, yolo_maps, yolo_maps_predvis)
from oracle import cv2_resize, nets, parse_paf, parse_yolo, preproc
from popnet_amd import synth


def test_preprocess_matches_reference_transform_chain(golden):
    g = golden.forward
    frames = [g["sample_frame"], synth.synth_depth(1, 640, 480, seed=3)[0]]
    x = np.stack([preproc.preprocess_frame(f) for f in frames])
    assert x.shape == g["x"].shape
    assert np.array_equal(x, g["x"])            # same numpy ops around the same resize restatement


def test_rtpose_forward_matches_reference_module(golden):
    g = golden.forward
    sd = state_dict_from_keys(golden.keys["rtpose_light3d"], seed=0)
    (paf, heat, z), inter = nets.rtpose_light3d_forward(torch.from_numpy(g["x"]), sd, return_intermediate=True)
    for got, key in ((paf, "rt_paf"), (heat, "rt_heat"), (z, "rt_z")):
        assert np.allclose(got.numpy(), g[key], atol=1e-5, rtol=0), key
    assert np.allclose(inter["feat"].numpy()[:, ::8, ::2, ::2], g["rt_feat"], atol=1e-4)
    assert np.allclose(inter["paf1"].numpy()[:, :, ::4, ::4], g["rt_paf1"], atol=1e-5)
    assert np.allclose(inter["heat1"].numpy()[:, :, ::4, ::4], g["rt_heat1"], atol=1e-5)
    assert np.allclose(inter["z1"].numpy()[:, :, ::4, ::4], g["rt_z1"], atol=1e-5)


def test_yolo_forward_matches_reference_module(golden):
    g = golden.forward
    sd = state_dict_from_keys(golden.keys["yolo_posenet"], seed=1)
    out, inter = nets.yolo_posenet_forward(torch.from_numpy(g["x"]), sd, return_intermediate=True)
    assert np.allclose(out.numpy(), g["yolo_out"], atol=2e-4, rtol=0)
    assert np.allclose(inter["feat"].numpy()[:, ::8, ::2, ::2], g["yolo_feat"], atol=1e-3, rtol=1e-5)


@pytest.mark.parametrize("name", all_parse_case_names())
def test_parse_matches_reference(golden, name):
    heat, paf, z = parse_case_inputs(golden, name)
    rec = parse_paf.frame_to_records(heat.copy(), paf.copy(), z.copy())
    g = golden.parse
    jl = np.asarray(rec["joint_list"], dtype=np.float64).reshape(-1, 5)
    assoc = np.asarray(rec["assoc"], dtype=np.float64).reshape(-1, 17)
    assert np.array_equal(jl, g["%s_joint_list" % name])                    # coordinates, scores, ids: exact
    assert assoc.shape == g["%s_assoc" % name].shape
    assert np.array_equal(assoc[:, :15], g["%s_assoc" % name][:, :15])      # person assignment: exact
    assert np.array_equal(assoc[:, 16], g["%s_assoc" % name][:, 16])
    # person scores: the reference's dot product goes through BLAS (possible FMA): 1-ulp level
    assert np.allclose(assoc[:, 15], g["%s_assoc" % name][:, 15], rtol=1e-12, atol=1e-12)
    depths = np.array([[j[2] for j in h] for h in rec["humans_3d"]], dtype=np.float64).reshape(-1, 15)
    assert np.array_equal(depths, g["%s_depths" % name])
    assert np.array_equal(np.array(rec["conf"], dtype=np.float64).reshape(-1, 15), g["%s_conf" % name])


def test_find_peaks_reflect_border_and_plateau():
    img = np.zeros((6, 7), np.float32)
    img[0, 0] = 0.5
    img[5, 6] = 0.4
    img[2, 3] = img[2, 4] = 0.9          # two-cell plateau: both are peaks in the reference
    img[4, 1] = 0.1                      # == threshold: not a peak (strict >)
    pk = parse_paf.find_peaks(0.1, img)
    assert pk.tolist() == [[0, 0], [3, 2], [4, 2], [6, 5]]


@pytest.mark.parametrize("seed", [31, 32, 33])
def test_yolo_decode_matches_reference(golden, seed):
    pm = yolo_maps(seed, clusters=seed != 33)
    b, h, v = parse_yolo.parse_prior_pose(pm, YOLO_ANCHORS, 15, 224, 224, 3, 2, 0.5, 0.5)
    for i in range(pm.shape[0]):
        assert np.array_equal(np.array(b[i], np.float32).reshape(-1, 5), golden.yolo["s%d_%d_bbox" % (seed, i)])
        assert np.array_equal(np.array(h[i], np.float32).reshape(-1, 15, 3), golden.yolo["s%d_%d_human" % (seed, i)])
        assert np.array_equal(np.array(v[i], bool).reshape(-1, 15), golden.yolo["s%d_%d_vis" % (seed, i)])


@pytest.mark.parametrize("seed,P", PAFPROCESS_CASES)
def test_pafprocess_restatement_matches_compiled_reference(golden, seed, P):
    from oracle import pafprocess as pp
    pk, heat, paf = coco_case(seed, P)
    want = golden.pafprocess["s%d_p%d" % (seed, P)]
    assert np.array_equal(humans_to_array(pp.restated().run(pk, heat, paf)), want)
    ref = pp.reference()          # the reference's own C++ (oracle/_ref), when it has been built
    if ref is not None:
        assert np.array_equal(humans_to_array(ref.run(pk, heat, paf)), want)


@pytest.mark.parametrize("seed,P", CPP_CASES)
def test_paf_to_pose_cpp_restatement_matches_reference_function(golden, seed, P):
    """oracle.parse_paf.paf_to_pose_cpp == the reference's paf_to_pose_cpp (paf_to_pose.py:381-415) run on the reference's own compiled
    pafprocess.cpp (tests/golden/make_golden.py::golden_paf_to_pose_cpp): NMS rows and every Human / BodyPart field."""
    from oracle import pafprocess as pp
    heat, paf = synth.coco_maps(seed, P)
    g = golden.cpp
    assert np.array_equal(g["s%d_p%d_insum" % (seed, P)], [float(heat.astype(np.float64).sum()), float(paf.astype(np.float64).sum())])
    rows, per_type = parse_paf.paf_to_pose_cpp(heat.copy(), paf.copy(), pp.restated())
    nms = np.array([tuple(pk) + (j,) for j, pks in enumerate(per_type) for pk in pks], dtype=np.float64).reshape(-1, 5)
    assert np.array_equal(nms, g["s%d_p%d_nms" % (seed, P)])
    assert np.array_equal(rows, g["s%d_p%d" % (seed, P)])


def test_script_level_pipeline_matches_reference_eval_script(golden):
    """preproc -> forward -> parse -> read-out glue of the oracle == eval_data.json written by the
    reference's evaluation_rtpose_light3d_kdh3d_mpreal_ablation.py on the same two frames."""
    s = golden.script
    sd = state_dict_from_keys(golden.keys["rtpose_light3d"], seed=s["weight_seed"])
    sd["model2_2.12.bias"][:15] += torch.tensor(s["heat_bias_shift"])
    frames = synth.synth_depth(2, 640, 480, seed=s["depth_seed"])
    x = torch.from_numpy(preproc.preprocess_batch(frames))
    paf, heat, z = nets.rtpose_light3d_forward(x, sd)
    paf, heat, z = (a.numpy().transpose(0, 2, 3, 1) for a in (paf, heat, z))
    for b in range(2):
        rec = parse_paf.frame_to_records(heat[b].copy(), paf[b].copy(), z[b].copy())
        assert rec["visibility"] == s["human_pred_set_visibility"][b]
        assert np.allclose(np.array(rec["humans_2d"]).reshape(-1, 15, 2), np.array(s["human_pred_set_2d"][b]).reshape(-1, 15, 2), atol=1e-9)
        assert np.allclose(np.array(rec["humans_3d"]).reshape(-1, 15, 3), np.array(s["human_pred_set_3d"][b]).reshape(-1, 15, 3), atol=1e-5)
        assert np.allclose(np.array(rec["conf"]).reshape(-1, 15), np.array(s["human_pred_set_part_conf"][b]).reshape(-1, 15), atol=1e-6)


def test_yolo_decode_pred_vis_matches_reference(golden):
    """pred_vis=True (prior_pose_align.py:62,120,153-157): 5 + 4 J channels per anchor, visibility = in-bounds test x channel."""
    pm = yolo_maps_predvis(34)
    b, h, v = parse_yolo.parse_prior_pose(pm, YOLO_ANCHORS, 15, 224, 224, 3, 2, 0.5, 0.5, pred_vis=True)
    total = 0
    for i in range(pm.shape[0]):
        assert np.array_equal(np.array(b[i], np.float32).reshape(-1, 5), golden.yolo["pv_%d_bbox" % i])
        assert np.array_equal(np.array(h[i], np.float32).reshape(-1, 15, 3), golden.yolo["pv_%d_human" % i])
        assert np.array_equal(np.array(v[i], np.float32).reshape(-1, 15), golden.yolo["pv_%d_vis" % i])
        total += len(b[i])
    assert total >= 4


def test_yolo_script_level_pipeline_matches_reference_eval_script(golden):
    """preproc -> YoloPoseNet forward -> decode/NMS -> glue of the oracle == eval_data.json written by
    the reference's evaluation_yolo_posenet_kdh3d_mpreal.py on the same two frames."""
    from oracle import parse_yolo
    from popnet_amd.config import INTRINSICS, YOLO_ANCHORS
    s = golden.script_yolo
    sd = state_dict_from_keys(golden.keys["yolo_posenet"], seed=s["weight_seed"])
    sd["model2_4.0.weight"][[4, 54]] -= np.float32(s["conf_weight_shift"])
    frames = synth.synth_depth(2, 640, 480, seed=s["depth_seed"])
    x = torch.from_numpy(preproc.preprocess_batch(frames))
    out = nets.yolo_posenet_forward(x, sd).numpy()
    bb, hh, _ = parse_yolo.parse_prior_pose(out, YOLO_ANCHORS, 15, 224, 224, 3, 2, 0.5, 0.5)
    for b in range(2):
        g = parse_yolo.frame_glue(bb[b], hh[b], 15, 224, 480, 640, INTRINSICS)
        want2 = np.array(s["human_pred_set_2d"][b]).reshape(-1, 15, 2)
        assert g["humans_2d"].shape == want2.shape
        assert np.allclose(g["humans_2d"], want2, atol=2e-3)
        assert np.allclose(g["humans_3d"], np.array(s["human_pred_set_3d"][b]).reshape(-1, 15, 3), atol=1e-4)
        assert np.allclose(g["part_conf"], np.array(s["human_pred_set_part_conf"][b]).reshape(-1, 15), atol=1e-5)


def test_bicubic_point_evaluation_equals_materialised_upsample():
    rng = np.random.default_rng(3)
    a = rng.standard_normal((28, 28)).astype(np.float32)
    up = cv2_resize.resize(a, None, fx=8, fy=8, interpolation=cv2_resize.INTER_CUBIC)
    for py, px in [(0, 0), (223, 223), (5, 219), (100, 3), (111, 112), (7, 8)]:
        assert cv2_resize.bicubic_x8_at(a, py, px) == up[py, px]


def test_cv2_restatement_cross_check_against_torch_interpolate():
    """The cv2 boundary is 'parity unpinned' (no OpenCV here); this only cross-checks the restated
    algorithm against torch's implementation of the same kernels (different evaluation order)."""
    import torch.nn.functional as F
    rng = np.random.default_rng(0)
    a = rng.standard_normal((28, 28, 3)).astype(np.float32)
    up = cv2_resize.resize(a, None, fx=8, fy=8, interpolation=cv2_resize.INTER_CUBIC)
    ref = F.interpolate(torch.from_numpy(a).permute(2, 0, 1)[None], scale_factor=8, mode="bicubic", align_corners=False)[0].permute(1, 2, 0).numpy()
    assert np.abs(up - ref).max() < 5e-6
    d = (rng.random((640, 480)) * 6).astype(np.float32)
    lin = cv2_resize.resize(d, (224, 224), interpolation=cv2_resize.INTER_LINEAR)
    ref = F.interpolate(torch.from_numpy(d)[None, None], size=(224, 224), mode="bilinear", align_corners=False)[0, 0].numpy()
    assert np.abs(lin - ref).max() < 2e-3
