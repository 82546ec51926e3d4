"""CPU: host logic and the C-ABI boundary (no GPU compute calls)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

import popnet_amd  # noqa: F401
from popnet_amd import _lib, synth
from popnet_amd.network.rtpose_light3d import rtpose_light3d
from popnet_amd.network.yolo_posenet import YoloPoseNet

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "popnet_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set(re.findall(r"\b([a-z_][a-z0-9_]*)\s*\(", txt))
    legacy = {"process_paf", "get_num_humans", "get_part_cid", "get_score", "get_part_x", "get_part_y", "get_part_score"}
    return sorted(n for n in names if n.startswith("pn_") or n in legacy)


def test_library_loads_and_exports_every_declared_symbol():
    handle = _lib.lib()
    syms = header_symbols()
    assert len(syms) >= 28
    for s in syms:
        assert hasattr(handle, s), "libpopnet_hip.so does not export %s" % s
    # every symbol the Python binding uses is declared in the header too
    assert set(_lib.declared_symbols()) <= set(syms)
    assert handle.pn_abi_version() == 1


def test_record_layouts_match_the_c_structs():
    handle = _lib.lib()
    assert handle.pn_sizeof_pose_frame() == _lib.POSE_FRAME_DTYPE.itemsize
    assert handle.pn_sizeof_yolo_frame() == _lib.YOLO_FRAME_DTYPE.itemsize
    cfg = _lib.ParseCfg()
    handle.pn_parse_cfg_default(C.byref(cfg))
    assert (round(cfg.thresh_heatmap, 6), round(cfg.thresh_paf, 6), cfg.num_intermed_pts, cfg.downsample) == (0.1, 0.05, 10, 8)
    assert (cfg.w_org, cfg.h_org, cfg.input_size) == (480, 640, 224)
    assert cfg.fx == 504.1189880371094 and cfg.cy == 320.62640380859375


def test_cubic_tap_table_equals_oracle():
    from oracle.cv2_resize import cubic_coeffs
    handle = _lib.lib()
    for p in range(8):
        out = (C.c_float * 4)()
        x = np.float32((2 * p + 1) / 16.0)
        handle.pn_debug_cubic_coeffs(C.c_float(float(x)), out)
        assert np.array_equal(np.array(list(out), dtype=np.float32), cubic_coeffs(x))


def test_state_dict_keys_identical_to_reference(golden):
    for name, model in (("rtpose_light3d", rtpose_light3d(15, 14, 2, input_dim=1)), ("yolo_posenet", YoloPoseNet(15, input_dim=1))):
        mine = [[k, list(v.shape)] for k, v in model.state_dict().items()]
        assert mine == golden.keys[name], "state_dict layout of %s differs from the reference" % name


def test_module_prefixed_checkpoint_loads():
    m = rtpose_light3d(15, 14, 2, input_dim=1)
    sd = {"module." + k: v.clone() for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    assert len(m.state_dict()) == 234 and sum(p.numel() for p in m.parameters()) == 5525814
    y = YoloPoseNet(15, input_dim=1)
    assert len(y.state_dict()) == 223 and sum(p.numel() for p in y.parameters()) == 12412608


def test_no_cpu_fallback_cpu_tensor_is_rejected():
    m = rtpose_light3d(15, 14, 2, input_dim=1).eval()
    with pytest.raises(_lib.PopnetError, match="CUDA/ROCm tensor"):
        m(torch.zeros(1, 1, 224, 224))
    from popnet_amd.utils.paf_to_pose import parse_paf_batch, make_parse_cfg
    with pytest.raises(_lib.PopnetError):
        parse_paf_batch(torch.zeros(1, 16, 28, 28), torch.zeros(1, 28, 28, 28), torch.zeros(1, 15, 28, 28), make_parse_cfg())


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU error path")
def test_context_without_gpu_reports_an_error_not_a_crash():
    with pytest.raises(_lib.PopnetError, match="not available"):
        _lib.Context(0)
    # the legacy entry point returns a negative status instead of aborting
    pk = np.zeros((1, 1, 5), np.float32)
    rc = _lib.lib().process_paf(1, 1, 5, pk.ctypes.data_as(C.c_void_p), 8, 8, 19, None, 8, 8, 38,
                                np.zeros((8, 8, 38), np.float32).ctypes.data_as(C.c_void_p))
    assert rc < 0


def test_missing_library_fails_loudly(tmp_path):
    """Importing the compute path without libpopnet_hip.so must raise, never fall back."""
    code = (
        "import sys, os; sys.path.insert(0, %r)\n"
        "import popnet_amd\n"
        "from popnet_amd import _lib\n"
        "_lib.LIB_PATH = os.path.join(%r, 'nope.so')\n"
        "try:\n"
        "    _lib.lib()\n"
        "except _lib.PopnetError as e:\n"
        "    print('RAISED', 'no CPU fallback' in str(e))\n" % (ROOT, str(tmp_path)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert "RAISED True" in out.stdout, out.stdout + out.stderr


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "popnet_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), "%s imports the oracle" % f


def test_synthetic_weights_are_keyed_by_name(golden):
    a = synth.fill_state_dict({"x.weight": torch.empty(4, 3, 3, 3), "y.bias": torch.empty(5)}, seed=3)
    b = synth.fill_state_dict({"y.bias": torch.empty(5), "x.weight": torch.empty(4, 3, 3, 3)}, seed=3)
    assert all(np.array_equal(a[k], b[k]) for k in a)


def test_shard_indices_cover_all_frames_once():
    from popnet_amd.pipeline import shard_indices
    for n, w in ((4484, 8), (10, 3), (7, 8), (32, 1)):
        seen = sorted(i for r in range(w) for i in shard_indices(n, r, w))
        assert seen == list(range(n))


def test_bench_line_carries_the_parity_mode_inside_config_and_roofline():
    """bench.surface_in_parsed: the driver keeps `config` / `roofline` verbatim and only NAMES other keys, so the
    tolerance-meeting mode's throughput and fidelity, the H2D-inclusive rate and the headline mode's own tolerance verdict must live inside
    those two objects -- nested once and as flat scalars."""
    sys.path.insert(0, ROOT)
    import bench
    fid = {"frames": 96, "same_person_count": 95, "same_assignment": 95, "d3_m_max": 5.5e-5}
    out = {"config": {"workload": "w"}, "roofline": {"frac": 0.38, "conv_stack": {"frac": 0.31}},
           "h2d_inclusive": {"value": 63000.0, "fraction_of_value": 0.95, "host_link": {"GBps": 55.6}},
           "fidelity": {"meets_north_star_tolerance": False, "same_assignment": "29/96", "d3_m_max": 0.03},
           "parity_mode": {"dtype": "bf16x3", "value": 26000.0, "ms_per_step": 1.23, "h2d_inclusive": {"value": 25500.0},
                           "roofline": {"frac": 0.45, "algorithmic_tflops": 373.0, "peak": 2500.0, "conv_stack_physical_frac": 0.37},
                           "fidelity": {"threshold_calibrated_weights": fid, "separated_weights": fid}},
           "train_step": {"ms_per_step": 20.7, "bf16x3": {"ms_per_step": 9.7}}, "yolo": {"value": 97000.0, "conv_stack": {"frac": 0.26}},
           "postproc": {"us_per_step": 50.0}}
    bench.surface_in_parsed(out)
    cfg, rf = out["config"], out["roofline"]
    for block in (cfg["parity_mode"], rf["parity_mode"]):
        assert block["dtype"] == "bf16x3" and block["value"] == 26000.0 and block["h2d_inclusive"] == 25500.0
        assert block["same_assignment"] == "95/96" and block["meets_north_star_tolerance"] is True
        assert abs(block["frac_algorithmic"] - 373.0 / 2500.0) < 1e-4 and block["frac_physical"] == 0.45
    assert cfg["parity_mode_value"] == 26000.0 and cfg["parity_mode_meets_north_star_tolerance"] is True
    assert cfg["h2d_inclusive_value"] == 63000.0 and cfg["value_meets_north_star_tolerance"] is False
    assert cfg["train_step_ms_fp32"] == 20.7 and cfg["train_step_ms_bf16x3"] == 9.7 and cfg["yolo_value"] == 97000.0
    assert rf["conv_stack_frac"] == 0.31 and rf["postproc_us_per_step"] == 50.0
    bare = {"config": {}, "roofline": {}}                      # a line without the secondary legs (N > 1, --no-extras) stays valid
    bench.surface_in_parsed(bare)
    assert "parity_mode" not in bare["config"]
