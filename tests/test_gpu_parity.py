"""GPU (-m gpu): the HIP path, called through the C ABI of libpopnet_hip.so, against
  (1) the golden vectors produced by the reference itself (tests/golden),
  (2) the oracle on seeded inputs the oracle finishes in seconds,
  (3) size-independent properties at BASELINE's full batch size.
Bit-exact for every index / integer / byte result and for all float results of the parse path
(same operation order as the oracle); fp32 forward within 2e-4 absolute of the fp32 CPU reference
(different summation order), bf16 forward within the tolerance stated in each test.
`/root/reference` is never touched here."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from helpers import (CPP_CASES, PAFPROCESS_CASES, YOLO_ANCHORS, all_parse_case_names, coco_case, humans_to_array,
                     parse_case_inputs, state_dict_from_keys, yolo_maps)
from popnet_amd import _lib, synth
from popnet_amd.config import default_cfg

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _loaded_native():
    assert _lib.lib() is not None        # raises when the extension is missing: no silent fallback


# ---------------------------------------------------------------------------------------------
# pre-processing
# ---------------------------------------------------------------------------------------------
def _preprocess(gpu, frames, dtype):
    t = torch.from_numpy(np.ascontiguousarray(frames)).to(gpu)
    B, H, W = t.shape
    out = torch.empty((B, 1, 224, 224), device=gpu)
    ctx = _lib.Context.for_device(0)
    ctx.check(_lib.lib().pn_preprocess(ctx.handle, C.c_void_p(t.data_ptr()), dtype, B, H, W, C.c_void_p(out.data_ptr()),
                                       224, 6.0, 3.0, 2.0, _lib.current_stream_ptr(gpu)), "pn_preprocess")
    torch.cuda.synchronize()
    return out.cpu().numpy()


def test_preprocess_bit_exact_vs_reference_golden(gpu, golden):
    _loaded_native()
    g = golden.forward
    x0 = _preprocess(gpu, g["sample_frame"][None], _lib.PN_DEPTH_F16)          # 240x320 ITOP frame of the reference
    x1 = _preprocess(gpu, synth.synth_depth(1, 640, 480, seed=3), _lib.PN_DEPTH_F16)
    assert np.array_equal(x0[0], g["x"][0]) and np.array_equal(x1[0], g["x"][1])


def test_preprocess_f32_edge_values(gpu):
    from oracle import preproc
    rng = np.random.default_rng(9)
    d = rng.uniform(-1, 8, (3, 97, 131)).astype(np.float32)       # ragged size, values outside [0, depth_max]
    got = _preprocess(gpu, d, _lib.PN_DEPTH_F32)
    assert np.array_equal(got, preproc.preprocess_batch(d))
    assert got.min() >= -1.5 and got.max() <= 1.5


# ---------------------------------------------------------------------------------------------
# network forward
# ---------------------------------------------------------------------------------------------
def _rtpose(golden, prec, seed=0):
    from popnet_amd.network.rtpose_light3d import rtpose_light3d
    m = rtpose_light3d(15, 14, 2, input_dim=1).eval()
    m.load_state_dict(state_dict_from_keys(golden.keys["rtpose_light3d"], seed=seed))
    m.precision = prec
    return m


def test_rtpose_forward_fp32_vs_reference_golden(gpu, golden):
    g = golden.forward
    m = _rtpose(golden, "fp32")
    (paf, heat, z), saved = m(torch.from_numpy(g["x"]).to(gpu))
    torch.cuda.synchronize()
    for got, key in ((paf, "rt_paf"), (heat, "rt_heat"), (z, "rt_z")):
        assert np.abs(got.cpu().numpy() - g[key]).max() < 2e-4, key
    assert np.abs(m.stem_features(2).cpu().numpy()[:, ::8, ::2, ::2] - g["rt_feat"]).max() < 2e-4
    for got, key in ((saved[0], "rt_paf1"), (saved[1], "rt_heat1"), (saved[2], "rt_z1")):
        assert np.abs(got.cpu().numpy()[:, :, ::4, ::4] - g[key]).max() < 2e-4, key
    assert saved[3] is paf and saved[4] is heat and saved[5] is z
    assert abs(m.flops_per_frame() / 1e9 - 13.343) < 0.01


def test_rtpose_forward_bf16_tolerance(gpu, golden):
    """bf16 storage + bf16 MFMA, fp32 accumulate: maps within 0.15 absolute (range: paf/z (-2,2),
    heat (0,1)) and 0.02 mean absolute of the fp32 reference on seeded O(1)-scale weights."""
    g = golden.forward
    m = _rtpose(golden, "bf16")
    (paf, heat, z), _ = m(torch.from_numpy(g["x"]).to(gpu))
    torch.cuda.synchronize()
    for got, key in ((paf, "rt_paf"), (heat, "rt_heat"), (z, "rt_z")):
        d = np.abs(got.cpu().numpy() - g[key])
        assert d.max() < 0.15 and d.mean() < 0.02, (key, d.max(), d.mean())


def test_yolo_forward_fp32_vs_reference_golden(gpu, golden):
    from popnet_amd.network.yolo_posenet import YoloPoseNet
    g = golden.forward
    m = YoloPoseNet(15, input_dim=1).eval()
    m.load_state_dict(state_dict_from_keys(golden.keys["yolo_posenet"], seed=1))
    m.precision = "fp32"
    out = m(torch.from_numpy(g["x"]).to(gpu))
    torch.cuda.synchronize()
    # activations reach |x| ~ 150 with the seeded weights; tolerance is relative to that scale
    feat = m.backbone_features(2).cpu().numpy()[:, ::8, ::2, ::2]
    assert np.abs(feat - g["yolo_feat"]).max() < 1e-5 * np.abs(g["yolo_feat"]).max() + 1e-4
    assert np.abs(out.cpu().numpy() - g["yolo_out"]).max() < 2e-3
    assert abs(m.flops_per_frame() / 1e9 - 8.691) < 0.01


def test_forward_batch_invariance_and_ragged_batches(gpu, golden):
    """Frames are independent (eval-mode BN): a frame's maps must not depend on what else is in the
    batch or on the batch size (exercises partial tiles / cached problem descriptors)."""
    m = _rtpose(golden, "fp32")
    x = torch.from_numpy(np.random.default_rng(4).normal(0, 1, (5, 1, 224, 224)).astype(np.float32)).to(gpu)
    (p5, h5, z5), _ = m(x)
    (p1, h1, z1), _ = m(x[3:4])
    (p2, h2, z2), _ = m(x[[4, 0]])
    torch.cuda.synchronize()
    assert torch.equal(p5[3:4], p1) and torch.equal(h5[3:4], h1) and torch.equal(z5[3:4], z1)
    assert torch.equal(p5[[4, 0]], p2) and torch.equal(z5[[4, 0]], z2)


def test_forward_other_input_sizes(gpu, golden):
    """Fully convolutional: the ITOP-native 240x320 frame (pitch classes 64 / 32) against the oracle."""
    from oracle import nets
    m = _rtpose(golden, "fp32")
    x = torch.from_numpy(np.random.default_rng(5).normal(0, 1, (2, 1, 240, 320)).astype(np.float32))
    sd = state_dict_from_keys(golden.keys["rtpose_light3d"], seed=0)
    rp, rh, rz = nets.rtpose_light3d_forward(x, sd)
    (p, h, z), _ = m(x.to(gpu))
    torch.cuda.synchronize()
    assert p.shape == (2, 28, 30, 40)
    for got, ref in ((p, rp), (h, rh), (z, rz)):
        assert (got.cpu() - ref).abs().max() < 2e-4


@pytest.mark.parametrize("prec,tol", [("fp32", 2e-4), ("bf16x3", 5e-4), ("bf16", 0.15)])
def test_multi_channel_input_vs_oracle(gpu, golden, prec, tol):
    """input_dim != 1 (round 4; the reference constructors DEFAULT to input_dim = 3: rtpose_light3d.py:250, yolo_posenet.py:88): the 7x7 stem
    of a 3-channel input runs on the generic fp32 convolution primitive and hands its map to the NHWC layers; both networks, every precision
    mode, against the oracle's torch fp32 forward of the same state_dict (same tolerances as the single-channel tests).  A batch whose channel
    count does not match input_dim is refused."""
    from oracle import nets
    from popnet_amd.network.rtpose_light3d import rtpose_light3d
    from popnet_amd.network.yolo_posenet import YoloPoseNet
    x = torch.from_numpy(np.random.default_rng(56).normal(0, 1, (3, 3, 224, 224)).astype(np.float32))

    def widen(keys, seed):            # the golden key list with a 3-channel stem
        keys = [[k, ([64, 3, 7, 7] if k == "model0.conv1.weight" else s)] for k, s in keys]
        return state_dict_from_keys(keys, seed=seed)

    sd = widen(golden.keys["rtpose_light3d"], 0)
    m = rtpose_light3d(15, 14, 2).eval()               # input_dim = 3: the reference's default
    assert m.input_dim == 3
    m.load_state_dict(sd)
    m.precision = prec
    (p, h, z), _ = m(x.to(gpu))
    torch.cuda.synchronize()
    for got, ref, name in zip((p, h, z), nets.rtpose_light3d_forward(x, sd), ("paf", "heat", "z")):
        d = (got.cpu() - ref).abs()
        assert torch.isfinite(got).all() and float(d.max()) < tol, (name, float(d.max()))
    with pytest.raises(_lib.PopnetError, match="input_dim"):
        m(x[:, :1].to(gpu))
    sdy = widen(golden.keys["yolo_posenet"], 1)
    my = YoloPoseNet(15).eval()
    my.load_state_dict(sdy)
    my.precision = prec
    out = my(x.to(gpu))
    torch.cuda.synchronize()
    ref = nets.yolo_posenet_forward(x, sdy)
    d = (out.cpu() - ref).abs()
    # the single-channel Yolo tests' tolerances (activations reach |x| ~ 150 with these weights); plain bf16 flips saturated sigmoid casts
    # outright (range [-2, 2]): bounded in the mean, as the bf16 rtpose maps are
    assert torch.isfinite(out).all()
    if prec == "bf16":
        assert float(d.mean()) < 0.05, float(d.mean())
    else:
        assert float(d.max()) < {"fp32": 2e-3, "bf16x3": 5e-3}[prec], float(d.max())


@pytest.mark.parametrize("prec,tol", [("fp32", 2e-4), ("bf16x3", 5e-4), ("bf16", 0.15)])
def test_default_constructor_topology_vs_oracle(gpu, prec, tol):
    """`rtpose_light3d()` exactly as the reference constructs it by default (rtpose_light3d.py:250: 18 parts, 19 limbs, 2 stages, 3 input
    channels): 38 + 19 + 20 head channels, a 205-channel stage-2 input (rtpose_light3d.py:283-304,339) whose slices are NOT multiples of 4
    channels wide -- the concat buffer pads them (net.hip::build_rtpose).  Stage-2 maps and the stage-1 maps inside the concat buffer
    against the oracle's torch fp32 forward of the same state_dict, every precision mode; `YoloPoseNet()` (15 parts, 3 channels) likewise."""
    from oracle import nets
    from popnet_amd.network.rtpose_light3d import rtpose_light3d
    from popnet_amd.network.yolo_posenet import YoloPoseNet
    x = torch.from_numpy(np.random.default_rng(57).normal(0, 1, (3, 3, 224, 224)).astype(np.float32))
    m = rtpose_light3d().eval()
    assert (m.num_parts, m.num_limbs, m.num_stages, m.input_dim) == (18, 19, 2, 3)
    assert tuple(m.state_dict()["model2_1.0.weight"].shape) == (256, 205, 3, 3)
    synth.load_synth_weights(m, seed=8)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m.precision = prec
    (p, h, z), saved = m(x.to(gpu))
    torch.cuda.synchronize()
    assert tuple(p.shape) == (3, 38, 28, 28) and tuple(h.shape) == (3, 19, 28, 28) and tuple(z.shape) == (3, 20, 28, 28)
    (rp, rh, rz), inter = nets.rtpose_light3d_forward(x, sd, return_intermediate=True)
    for got, ref, name in ((p, rp, "paf"), (h, rh, "heat"), (z, rz, "z"), (saved[0], inter["paf1"], "paf1"), (saved[1], inter["heat1"], "heat1"),
                           (saved[2], inter["z1"], "z1")):
        d = (got.cpu() - ref).abs()
        assert torch.isfinite(got).all() and float(d.max()) < tol, (name, float(d.max()))
    my = YoloPoseNet().eval()
    assert (my.num_parts, my.input_dim) == (15, 3)
    synth.load_synth_weights(my, seed=9)
    sdy = {k: v.detach().clone() for k, v in my.state_dict().items()}
    my.precision = prec
    out = my(x.to(gpu))
    torch.cuda.synchronize()
    d = (out.cpu() - nets.yolo_posenet_forward(x, sdy)).abs()
    assert torch.isfinite(out).all()
    if prec == "bf16":
        assert float(d.mean()) < 0.05, float(d.mean())
    else:
        assert float(d.max()) < {"fp32": 2e-3, "bf16x3": 5e-3}[prec], float(d.max())


@pytest.mark.parametrize("hw", [(224, 224), (240, 320), (200, 232), (256, 192), (96, 480)])
def test_bf16_strip_kernel_equals_generic_kernel_bit_for_bit(gpu, golden, hw, monkeypatch):
    """conv3_kernel (24..30-column strip tiles, partial last tiles, merged narrow convs) accumulates in the same k order
    with the same MFMA as the generic kernel, so the bf16 maps must be IDENTICAL, for both networks and at sizes whose
    maps are not multiples of 28 wide (200x232 -> 25x29 / 50x58 / 100x116; 256x192 -> 24-wide strips + 32-wide maps
    that stay on the generic kernel).  And the bf16 rtpose maps stay within the stated tolerance of fp32."""
    from popnet_amd.network.yolo_posenet import YoloPoseNet
    H, W = hw
    x = torch.from_numpy(np.random.default_rng(6).normal(0, 1, (3, 1, H, W)).astype(np.float32)).to(gpu)

    def yolo():
        m = YoloPoseNet(15, input_dim=1).eval()
        m.load_state_dict(state_dict_from_keys(golden.keys["yolo_posenet"], seed=1))
        m.precision = "bf16"
        return m

    do_yolo = H % 16 == 0 and W % 16 == 0
    got = [t.clone() for t in _rtpose(golden, "bf16")(x)[0]]
    got_y = yolo()(x).clone() if do_yolo else None
    monkeypatch.setenv("POPNET_CONV4", "1")              # conv4_kernel on every eligible level (the default picks it by block count)
    got4 = [t.clone() for t in _rtpose(golden, "bf16")(x)[0]]
    got4_y = yolo()(x).clone() if do_yolo else None
    monkeypatch.delenv("POPNET_CONV4")
    monkeypatch.setenv("POPNET_NO_CONV3", "1")          # read when the net is compiled
    ref = [t.clone() for t in _rtpose(golden, "bf16")(x)[0]]
    ref_y = yolo()(x).clone() if do_yolo else None
    monkeypatch.delenv("POPNET_NO_CONV3")
    f32 = [t.clone() for t in _rtpose(golden, "fp32")(x)[0]]
    torch.cuda.synchronize()
    for a, a4, b, c, name in zip(got, got4, ref, f32, ("paf", "heat", "z")):
        assert torch.isfinite(a).all() and torch.equal(a, b), name
        assert torch.equal(a4, b), ("conv4", name)
        d = (a - c).abs()
        assert float(d.max()) < 0.15 and float(d.mean()) < 0.02, (name, float(d.max()), float(d.mean()))
    if do_yolo:
        assert torch.isfinite(got_y).all() and torch.equal(got_y, ref_y)
        assert torch.equal(got4_y, ref_y)


@pytest.mark.parametrize("switch", ["POPNET_CONV3_PT14=1", "POPNET_CONV3_PT14=2", "POPNET_CONV3_NBUF2=1", "POPNET_CONV3_RPG8=1", "POPNET_CONV4=1", "POPNET_CONV4=0", "POPNET_NO_BBLOCK=1", "POPNET_NO_TAILFUSE=1", "POPNET_NO_MIX=1", "POPNET_NO_POOLFUSE=1", "POPNET_GENERIC_C64=0", "POPNET_NO_STEMPOOL=1", "POPNET_NO_EMBED3=1", "POPNET_BB64_STATIC=1", "POPNET_BB64_HALVES=2"])
def test_bf16_optional_kernel_variants_are_bit_identical(gpu, golden, switch, monkeypatch):
    """The experiment switches of profiles/README.md (224-pixel wave tiles, double-buffered halo images, 8-row tiles on
    14-column maps; conv4_kernel on every / no level; the two layer1 BasicBlocks as two launches each instead of the fused
    bb64_kernel) select other kernels of the same arithmetic: same maps, bit for bit, as the
    default build of the net (rtpose at 224x224 with a ragged batch of 5; YoloPoseNet for its 14x14 layers)."""
    from popnet_amd.network.yolo_posenet import YoloPoseNet
    x = torch.from_numpy(np.random.default_rng(16).normal(0, 1, (5, 1, 224, 224)).astype(np.float32)).to(gpu)

    def yolo():
        m = YoloPoseNet(15, input_dim=1).eval()
        m.load_state_dict(state_dict_from_keys(golden.keys["yolo_posenet"], seed=1))
        m.precision = "bf16"
        return m

    ref = [t.clone() for t in _rtpose(golden, "bf16")(x)[0]]
    ref_y = yolo()(x).clone()
    k, v = switch.split("=")
    monkeypatch.setenv(k, v)                               # read when the net is compiled
    got = [t.clone() for t in _rtpose(golden, "bf16")(x)[0]]
    got_y = yolo()(x).clone()
    monkeypatch.delenv(k)
    torch.cuda.synchronize()
    for a, b, name in zip(got, ref, ("paf", "heat", "z")):
        assert torch.isfinite(a).all() and torch.equal(a, b), (switch, name)
    assert torch.isfinite(got_y).all() and torch.equal(got_y, ref_y), switch


def test_bb64_tile_tickets_are_left_at_zero_for_the_next_launch(gpu, golden, monkeypatch):
    """bb64_kernel hands its tiles out by ticket (round 5; launches with at least four tiles per workgroup); the last workgroup to sign
    off resets the two counters, so the same net -- eager or as a replayed graph -- gives the same maps launch after launch (a stale
    counter would skip every tile past the second per workgroup: 32 frames of 224x224 are 1792 tiles for 256 workgroups), and the
    maps of the static schedule (POPNET_BB64_STATIC=1, read when the net is compiled) bit for bit."""
    x = torch.from_numpy(np.random.default_rng(61).normal(0, 1, (32, 1, 224, 224)).astype(np.float32)).to(gpu)
    monkeypatch.setenv("POPNET_BB64_STATIC", "1")
    static = [t.clone() for t in _rtpose(golden, "bf16")(x)[0]]
    monkeypatch.delenv("POPNET_BB64_STATIC")
    net = _rtpose(golden, "bf16")
    first = [t.clone() for t in net(x)[0]]
    for a, b, name in zip(first, static, ("paf", "heat", "z")):
        assert torch.equal(a, b), ("static schedule", name)
    for _ in range(4):
        again = [t.clone() for t in net(x)[0]]
        for a, b, name in zip(again, first, ("paf", "heat", "z")):
            assert torch.equal(a, b), name
    assert all(torch.isfinite(t).all() for t in first)


@pytest.mark.parametrize("hw,B", [((224, 224), 3), ((200, 232), 2), ((256, 192), 2)])
def test_bf16x3_strip_kernels_equal_the_generic_kernel_bit_for_bit(gpu, golden, hw, B, monkeypatch):
    """The tolerance-meeting mode on the strip kernels (conv3_kernel, conv4_kernel where a level has >= 448 blocks or everywhere with
    POPNET_CONV4=1) against the generic kernel on every layer (POPNET_NO_CONV3=1), under
    precision="bf16x3" with two-plane tensors [hi | lo] whose third plane pair re-reads hi (ConvProblem::in_wrap): same k order, same MFMA,
    same epilogue arithmetic -> IDENTICAL maps, both networks, ragged maps included."""
    from popnet_amd.network.yolo_posenet import YoloPoseNet
    H, W = hw
    x = torch.from_numpy(np.random.default_rng(46).normal(0, 1, (B, 1, H, W)).astype(np.float32)).to(gpu)

    def yolo():
        m = YoloPoseNet(15, input_dim=1).eval()
        m.load_state_dict(state_dict_from_keys(golden.keys["yolo_posenet"], seed=1))
        m.precision = "bf16x3"
        return m

    do_yolo = H % 16 == 0 and W % 16 == 0
    got = [t.clone() for t in _rtpose(golden, "bf16x3")(x)[0]]
    got_y = yolo()(x).clone() if do_yolo else None
    monkeypatch.setenv("POPNET_CONV4", "1")
    got4 = [t.clone() for t in _rtpose(golden, "bf16x3")(x)[0]]
    monkeypatch.delenv("POPNET_CONV4")
    monkeypatch.setenv("POPNET_NO_CONV3", "1")          # read when the net is compiled
    ref = [t.clone() for t in _rtpose(golden, "bf16x3")(x)[0]]
    ref_y = yolo()(x).clone() if do_yolo else None
    monkeypatch.delenv("POPNET_NO_CONV3")
    torch.cuda.synchronize()
    for a, a4, b, name in zip(got, got4, ref, ("paf", "heat", "z")):
        assert torch.isfinite(a).all() and torch.equal(a, b), name
        assert torch.equal(a4, b), ("conv4", name)
    if do_yolo:
        assert torch.isfinite(got_y).all() and torch.equal(got_y, ref_y)


@pytest.mark.parametrize("hw,B", [((224, 224), 5), ((200, 232), 3), ((256, 192), 2), ((96, 480), 3), ((72, 136), 4)])
def test_bf16x3_fused_basic_block_equals_the_two_launch_plan_bit_for_bit(gpu, golden, hw, B, monkeypatch):
    """bb64x3_kernel: the BasicBlock(64)s of the split-bf16 nets as ONE launch -- two-plane LDS images, the
    intermediate map never leaves the CU (POPNET_BBLOCK_X3=1) -- against the two conv3_kernel launches per block, the default plan under
    precision="bf16x3": same k order (plane pair, half, tap), same MFMA, same epilogue arithmetic, so the maps must be IDENTICAL.
    Sizes: the bench's 112 x 112 maps (6-row tiles: 18 full + one 4-row tile), ragged widths (100 x 116 -> 25 / 29-column strips,
    128 x 96 -> 24-column strips, 48 x 240), a 36 x 68 map (six full row tiles, 23 / 22-column strips); rtpose's layer1 (two blocks at
    H / 2) and YoloPoseNet's (three blocks at H / 4)."""
    from popnet_amd.network.yolo_posenet import YoloPoseNet
    H, W = hw
    x = torch.from_numpy(np.random.default_rng(36).normal(0, 1, (B, 1, H, W)).astype(np.float32)).to(gpu)

    def yolo():
        m = YoloPoseNet(15, input_dim=1).eval()
        m.load_state_dict(state_dict_from_keys(golden.keys["yolo_posenet"], seed=1))
        m.precision = "bf16x3"
        return m

    do_yolo = H % 16 == 0 and W % 16 == 0
    monkeypatch.setenv("POPNET_BBLOCK_X3", "1")          # read when the net is compiled (the bf16x3 plan keeps two launches by default: net.hip::fuse_basic_blocks)
    got = [t.clone() for t in _rtpose(golden, "bf16x3")(x)[0]]
    got_y = yolo()(x).clone() if do_yolo else None
    monkeypatch.delenv("POPNET_BBLOCK_X3")
    ref = [t.clone() for t in _rtpose(golden, "bf16x3")(x)[0]]
    ref_y = yolo()(x).clone() if do_yolo else None
    f32 = [t.clone() for t in _rtpose(golden, "fp32")(x)[0]]
    torch.cuda.synchronize()
    for a, b, c, name in zip(got, ref, f32, ("paf", "heat", "z")):
        assert torch.isfinite(a).all() and torch.equal(a, b), name
        assert float((a - c).abs().max()) < 5e-4, name            # and the bf16x3 maps stay fp32-class
    if do_yolo:
        assert torch.isfinite(got_y).all() and torch.equal(got_y, ref_y)


@pytest.mark.parametrize("hw", [(224, 224), (256, 192), (96, 480), (240, 336)])
def test_yolo_stem_with_fused_maxpool_is_bit_identical(gpu, golden, hw, monkeypatch):
    """YoloPoseNet's conv1 - bn1 - relu - maxpool as one launch (stem7x7_pool_kernel: 8 x 7 pooled pixels per block, the window maxima
    taken in LDS) against the stem and the pool as two launches (POPNET_NO_STEMPOOL=1), at sizes whose pooled maps are not multiples of
    the block tile: same output tensor, bit for bit."""
    from popnet_amd.network.yolo_posenet import YoloPoseNet
    H, W = hw
    x = torch.from_numpy(np.random.default_rng(26).normal(0, 1, (3, 1, H, W)).astype(np.float32)).to(gpu)

    def yolo():
        m = YoloPoseNet(15, input_dim=1).eval()
        m.load_state_dict(state_dict_from_keys(golden.keys["yolo_posenet"], seed=1))
        m.precision = "bf16"
        return m

    got = yolo()(x).clone()
    monkeypatch.setenv("POPNET_NO_STEMPOOL", "1")          # read when the net is compiled
    ref = yolo()(x).clone()
    monkeypatch.delenv("POPNET_NO_STEMPOOL")
    torch.cuda.synchronize()
    assert torch.isfinite(got).all() and torch.equal(got, ref)


def test_bf16_batch_invariance_and_ragged_batches(gpu, golden):
    """Same as the fp32 invariance test, for the bf16 path (grouped launches, merged narrow convs, strip tiles)."""
    m = _rtpose(golden, "bf16")
    x = torch.from_numpy(np.random.default_rng(4).normal(0, 1, (7, 1, 224, 224)).astype(np.float32)).to(gpu)
    (p7, h7, z7), _ = m(x)
    p7, h7, z7 = p7.clone(), h7.clone(), z7.clone()
    (p1, h1, z1), _ = m(x[5:6])
    p1, h1, z1 = p1.clone(), h1.clone(), z1.clone()
    (p2, h2, z2), _ = m(x[[6, 1]])
    torch.cuda.synchronize()
    assert torch.equal(p7[5:6], p1) and torch.equal(h7[5:6], h1) and torch.equal(z7[5:6], z1)
    assert torch.equal(p7[[6, 1]], p2) and torch.equal(h7[[6, 1]], h2) and torch.equal(z7[[6, 1]], z2)


# ---------------------------------------------------------------------------------------------
# Open-Pose+ parsing
# ---------------------------------------------------------------------------------------------
def _parse(gpu, heat_hwc, paf_hwc, z_hwc):
    from popnet_amd.utils.paf_to_pose import make_parse_cfg, parse_paf_batch
    t = [torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))[None].to(gpu) for a in (heat_hwc, paf_hwc, z_hwc)]
    return parse_paf_batch(t[0], t[1], t[2], make_parse_cfg(default_cfg()))[0]


@pytest.mark.parametrize("name", all_parse_case_names())
def test_parse_bit_exact_vs_reference_golden(gpu, golden, name):
    from popnet_amd.utils.paf_to_pose import frame_assoc, frame_joint_list
    heat, paf, z = parse_case_inputs(golden, name)
    fr = _parse(gpu, heat, paf, z)
    g = golden.parse
    assert int(fr["status"]) == 0
    jl = frame_joint_list(fr).reshape(-1, 5)
    assoc = frame_assoc(fr).reshape(-1, 17)
    assert np.array_equal(jl, g["%s_joint_list" % name])                     # x, y, refined score, id, type
    assert assoc.shape == g["%s_assoc" % name].shape
    assert np.array_equal(assoc[:, :15], g["%s_assoc" % name][:, :15])       # person assignment indices
    assert np.array_equal(assoc[:, 16], g["%s_assoc" % name][:, 16])
    assert np.allclose(assoc[:, 15], g["%s_assoc" % name][:, 15], rtol=1e-12, atol=1e-12)   # BLAS-vs-plain 1 ulp
    n = assoc.shape[0]
    assert np.array_equal(fr["joints_3d"][:n, :, 2], g["%s_depths" % name])  # heat-weighted depth read-out
    assert np.array_equal(fr["part_conf"][:n], g["%s_conf" % name])


def test_paf_to_pose_api_returns_reference_structures(gpu, golden):
    """The per-frame drop-in: same call as tpm/lib/utils/paf_to_pose.py:354, same return types."""
    from popnet_amd.utils.paf_to_pose import paf_to_pose
    from popnet_amd.utils.common import paf_to_human_list, retrieve_depth_heat_weighted
    heat, paf, z = parse_case_inputs(golden, "planted_s4_p3")
    joint_list, assoc = paf_to_pose(heat, paf, default_cfg())
    assert joint_list.dtype == np.float64 and assoc.dtype == np.float64
    assert np.array_equal(joint_list, golden.parse["planted_s4_p3_joint_list"])
    humans, vis, conf = paf_to_human_list(joint_list, assoc)
    assert len(humans) == len(golden.parse["planted_s4_p3_assoc"]) and len(humans[0]) == 15 and len(vis[0]) == 15
    # empty frame -> empty arrays like the reference
    jl0, as0 = paf_to_pose(np.zeros((28, 28, 16), np.float32), np.zeros((28, 28, 28), np.float32), default_cfg())
    assert jl0.size == 0 and as0.size == 0
    # stand-alone depth read-out, including the in-place clamp of negative heat values
    depthmap = (z[:, :, 0] * 2 + 3).astype(np.float32)
    hm = heat[:, :, 0].copy()
    hm[3, 3] = -0.5
    from oracle import parse_paf as O
    hm_ref = hm.copy()
    want = O.retrieve_depth_heat_weighted([3, 4], depthmap, hm_ref, 1)
    got = retrieve_depth_heat_weighted([3, 4], depthmap, hm, 1)
    assert got == want and hm[3, 3] == 0.0 and np.array_equal(hm, hm_ref)
    assert retrieve_depth_heat_weighted([0, 27], depthmap, hm, 1) == O.retrieve_depth_heat_weighted([0, 27], depthmap, hm_ref, 1)


def test_parse_full_batch_matches_oracle_and_is_batch_invariant(gpu):
    """BASELINE batch (32 frames, 0..8 persons): every frame equals the oracle, and equals the same
    frame parsed alone (no cross-frame state in the workspace)."""
    from oracle import parse_paf as O
    from popnet_amd.utils.paf_to_pose import frame_assoc, frame_joint_list, make_parse_cfg, parse_paf_batch
    persons = [(i * 5) % 9 for i in range(32)]
    heat, paf, z = synth.planted_batch(77, persons)
    cfg = make_parse_cfg(default_cfg())
    th, tp, tz = (torch.from_numpy(a).to(gpu) for a in (heat, paf, z))
    frames = parse_paf_batch(th, tp, tz, cfg)
    again = parse_paf_batch(th, tp, tz, cfg)

    def valid(fr):          # the defined part of a record (slots past n_peaks / n_persons are unspecified)
        n, p = int(fr["n_peaks"]), int(fr["n_persons"])
        return (n, p, int(fr["status"]), fr["peak_x"][:n].tobytes(), fr["peak_y"][:n].tobytes(), fr["peak_score"][:n].tobytes(),
                fr["peak_type"][:n].tobytes(), fr["person_joint"][:p].tobytes(), fr["person_score"][:p].tobytes(),
                fr["person_count"][:p].tobytes(), fr["joints_2d"][:p].tobytes(), fr["joints_3d"][:p].tobytes(),
                fr["part_conf"][:p].tobytes())

    for b in range(32):
        assert valid(frames[b]) == valid(again[b])                           # idempotent / deterministic
    for b in (0, 5, 13, 31):
        alone = parse_paf_batch(th[b:b + 1], tp[b:b + 1], tz[b:b + 1], cfg)[0]
        assert valid(alone) == valid(frames[b])
    for b in range(32):
        rec = O.frame_to_records(heat[b].transpose(1, 2, 0).copy(), paf[b].transpose(1, 2, 0).copy(), z[b].transpose(1, 2, 0).copy())
        fr = frames[b]
        assert int(fr["status"]) == 0
        jl, assoc = frame_joint_list(fr), frame_assoc(fr)
        assert jl.shape == np.asarray(rec["joint_list"]).shape and (jl.size == 0 or np.array_equal(jl, rec["joint_list"]))
        assert assoc.shape == np.asarray(rec["assoc"]).shape
        if assoc.size:
            n = assoc.shape[0]
            assert np.array_equal(assoc[:, :15], rec["assoc"][:, :15])
            assert np.array_equal(fr["joints_2d"][:n], np.array(rec["humans_2d"]))
            assert np.array_equal(fr["joints_3d"][:n], np.array(rec["humans_3d"]))
            assert np.array_equal(fr["part_conf"][:n], np.array(rec["conf"]))


def _unbounded(gpu, heat_hwc, paf_hwc, z_hwc):
    from popnet_amd.utils.paf_to_pose import make_parse_cfg, parse_paf_unbounded
    t = [torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).to(gpu) for a in (heat_hwc, paf_hwc, z_hwc)]
    return parse_paf_unbounded(t[0], t[1], t[2], make_parse_cfg(default_cfg()))


def test_parse_overflow_second_pass_returns_the_reference_result(gpu):
    """The reference has no capacity limit (paf_to_pose.py:33-153,267-351); the fixed-size record has (32 peaks per joint map,
    32 persons).  A constant heat map is one giant plateau -- every one of the 784 cells of every joint map is a peak: the
    record carries the overflow status (never a silent truncation) and the per-frame API answers through the unbounded second
    pass (pn_parse_paf_unbounded) with the oracle's 15 x 784-row joint list."""
    from popnet_amd.utils.paf_to_pose import paf_to_pose
    from oracle import parse_paf as oparse
    heat = np.full((28, 28, 16), 0.5, np.float32)
    fr = _parse(gpu, heat, np.zeros((28, 28, 28), np.float32), np.zeros((28, 28, 15), np.float32))
    assert int(fr["status"]) & _lib.PN_FRAME_OVERFLOW_PEAKS
    jl, assoc = paf_to_pose(heat, np.zeros((28, 28, 28), np.float32), default_cfg())
    ref = oparse.nms(heat)
    want = np.concatenate([np.concatenate([pk, np.full((len(pk), 1), j)], axis=1) for j, pk in enumerate(ref)])
    assert jl.shape == (15 * 784, 5) and np.array_equal(jl, want)
    assert assoc.size == 0                                                   # no limb evidence in an all-zero PAF: no persons


@pytest.mark.parametrize("name", all_parse_case_names())
def test_unbounded_pass_equals_the_fixed_size_records_on_the_reference_goldens(gpu, golden, name):
    """Same arithmetic, same order: on frames that fit the records the second pass must give what the fast path gives -- and
    therefore what the reference gave (goldens)."""
    heat, paf, z = parse_case_inputs(golden, name)
    r = _unbounded(gpu, heat, paf, z)
    g = golden.parse
    assert np.array_equal(r["joint_list"].reshape(-1, 5), g["%s_joint_list" % name])
    want = g["%s_assoc" % name]
    got = r["person_to_joint_assoc"].reshape(-1, 17)
    assert got.shape == want.shape
    if want.size:
        assert np.array_equal(got[:, :15], want[:, :15]) and np.array_equal(got[:, 16], want[:, 16])
        assert np.allclose(got[:, 15], want[:, 15], rtol=1e-12, atol=0)
    fr = _parse(gpu, heat, paf, z)
    n = int(fr["n_persons"])
    assert n == len(r["joints_2d"])
    assert np.array_equal(fr["joints_2d"][:n], r["joints_2d"]) and np.array_equal(fr["joints_3d"][:n], r["joints_3d"])
    assert np.array_equal(fr["part_conf"][:n], r["part_conf"]) and np.array_equal(fr["person_score"][:n], got[:, 15] if n else np.zeros(0))


@pytest.mark.parametrize("persons,seed", [(40, 1), (48, 2)])
def test_more_persons_than_the_record_holds_vs_oracle(gpu, persons, seed):
    """40 / 48 planted persons in one frame: more than PN_MAX_PERSONS (and more than 32 peaks in every joint map).  The record
    flags it; the second pass equals the oracle exactly -- joint list, person assignment, 2D / 3D joints, confidences -- and
    PoseEngine.predict_lists hands such a frame back as result lists instead of raising."""
    from popnet_amd import synth
    from oracle import parse_paf as oparse
    heat, paf, z = (a[0].transpose(1, 2, 0) for a in synth.planted_batch(700 + seed, [persons], noise=0.01))
    fr = _parse(gpu, heat, paf, z)
    assert int(fr["status"]) != 0
    r = _unbounded(gpu, heat, paf, z)
    ref = oparse.frame_to_records(heat.copy(), paf.copy(), z.copy())
    assert np.array_equal(r["joint_list"], ref["joint_list"])
    oa = np.asarray(ref["assoc"]).reshape(-1, 17)
    got = r["person_to_joint_assoc"].reshape(-1, 17)
    assert got.shape == oa.shape and got.shape[0] > 32
    assert np.array_equal(got[:, :15], oa[:, :15]) and np.array_equal(got[:, 16], oa[:, 16]) and np.allclose(got[:, 15], oa[:, 15], rtol=1e-12, atol=0)
    assert np.array_equal(r["joints_2d"], np.array(ref["humans_2d"])) and np.array_equal(r["joints_3d"], np.array(ref["humans_3d"]))
    assert np.array_equal(r["part_conf"], np.array(ref["conf"]))
    # engine level: a batch with one such frame among ordinary ones comes back as result lists, nothing raises or truncates
    from popnet_amd.dataset import pose_records_to_lists
    from popnet_amd.pipeline import PoseEngine, records_to_numpy
    eng = PoseEngine(precision="fp32", device=gpu, max_batch=3)
    hb, pb, zb = synth.planted_batch(700 + seed, [persons, 2, 3], noise=0.01)
    eng.heat.copy_(torch.from_numpy(hb).to(gpu)); eng.paf.copy_(torch.from_numpy(pb).to(gpu)); eng.z.copy_(torch.from_numpy(zb).to(gpu))
    eng.parse(3)
    recs = records_to_numpy(eng.frames[:3])
    assert int(recs[0]["status"]) != 0 and int(recs[1]["status"]) == 0
    with pytest.raises(_lib.PopnetError, match="overflow"):
        pose_records_to_lists(recs)
    lists = eng.lists_from_records(recs)
    assert np.array_equal(np.array(lists["human_pred_set_3d"][0]), np.array(ref["humans_3d"])) and len(lists["human_pred_set_2d"][0]) == got.shape[0]
    assert len(lists["human_pred_set_2d"][1]) == int(recs[1]["n_persons"]) > 0


# ---------------------------------------------------------------------------------------------
# Yolo-Pose+ decode
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", [31, 32, 33])
def test_yolo_decode_bit_exact_vs_reference_golden(gpu, golden, seed):
    from popnet_amd.utils.prior_pose_align import parse_prior_pose
    pm = yolo_maps(seed, clusters=seed != 33)
    t = torch.from_numpy(pm.copy()).to(gpu)
    b, h, v = parse_prior_pose(t, YOLO_ANCHORS, 15, 224, 224, 3, 2, 0.5, 0.5)
    assert torch.equal(t.cpu(), torch.from_numpy(pm))                        # input not modified
    for i in range(pm.shape[0]):
        assert np.array_equal(np.array(b[i], np.float32).reshape(-1, 5), golden.yolo["s%d_%d_bbox" % (seed, i)])
        assert np.array_equal(np.array(h[i], np.float32).reshape(-1, 15, 3), golden.yolo["s%d_%d_human" % (seed, i)])
        assert np.array_equal(np.array(v[i], bool).reshape(-1, 15), golden.yolo["s%d_%d_vis" % (seed, i)])


def test_yolo_decode_inplace_reproduces_the_reference_side_effect(gpu):
    """parse_prior_pose(..., inplace=True): the caller's tensor ends up holding what the reference leaves in it (prior_pose_align.py:22-52,
    oracle.parse_yolo.decode_maps) bit for bit, a 3-D input gains its batch dimension, and a SECOND call then decodes the decoded maps --
    the double application SURVEY Appendix B records for the reference."""
    from oracle import parse_yolo as O
    from popnet_amd.utils.prior_pose_align import parse_prior_pose
    pm = yolo_maps(31)
    t = torch.from_numpy(pm.copy()).to(gpu)
    b0, h0, v0 = parse_prior_pose(t.clone(), YOLO_ANCHORS, 15, 224, 224, 3, 2, 0.5, 0.5)
    b1, h1, v1 = parse_prior_pose(t, YOLO_ANCHORS, 15, 224, 224, 3, 2, 0.5, 0.5, inplace=True)
    assert sum(len(x) for x in b1) > 0
    for i in range(pm.shape[0]):                                             # same results as the non-mutating call ...
        assert np.array_equal(np.array(b0[i]), np.array(b1[i])) and np.array_equal(np.array(h0[i]), np.array(h1[i])) and np.array_equal(np.array(v0[i]), np.array(v1[i]))
    dec = O.decode_maps(pm, YOLO_ANCHORS, 15, 3, 2)                         # ... and the reference's decoded maps left behind
    assert np.array_equal(t.cpu().numpy().reshape(dec.shape), dec)
    # second call: decodes the decoded maps, like the reference run twice (= the oracle on the mutated array)
    rb, rh, rv = O.parse_prior_pose(dec.reshape(pm.shape).copy(), YOLO_ANCHORS, 15, 224, 224, 3, 2, 0.5, 0.5)
    b2, h2, v2 = parse_prior_pose(t, YOLO_ANCHORS, 15, 224, 224, 3, 2, 0.5, 0.5, inplace=True)
    differs = False
    for i in range(pm.shape[0]):
        assert len(b2[i]) == len(rb[i])
        if len(b2[i]):
            assert np.array_equal(np.array(b2[i]), np.array(rb[i])) and np.array_equal(np.array(h2[i]), np.array(rh[i]))
        differs = differs or len(b2[i]) != len(b1[i]) or (len(b2[i]) and not np.array_equal(np.array(b2[i]), np.array(b1[i])))
    assert differs
    one = torch.from_numpy(pm[0].copy()).to(gpu)                             # 3-D input: unsqueezed in place like posemaps.unsqueeze_(0)
    parse_prior_pose(one, YOLO_ANCHORS, 15, 224, 224, 3, 2, 0.5, 0.5, inplace=True)
    assert one.dim() == 4 and np.array_equal(one.cpu().numpy().reshape(dec[:1].shape), dec[:1])
    with pytest.raises(_lib.PopnetError, match="contiguous float32"):
        parse_prior_pose(t.double(), YOLO_ANCHORS, 15, 224, 224, 3, 2, 0.5, 0.5, inplace=True)


def test_yolo_decode_dense_batch_vs_oracle(gpu):
    from oracle import parse_yolo as O
    from popnet_amd.utils.prior_pose_align import parse_prior_pose
    rng = np.random.default_rng(8)
    pm = rng.uniform(-1, 1, (32, 100, 14, 14)).astype(np.float32)
    pm[:, 4] = rng.uniform(0, 0.58, (32, 14, 14))
    pm[:, 54] = rng.uniform(0, 0.54, (32, 14, 14))
    pm[:, 2:4] = rng.uniform(0.5, 2, (32, 2, 14, 14))
    pm[:, 52:54] = rng.uniform(0.5, 2, (32, 2, 14, 14))
    rb, rh, rv = O.parse_prior_pose(pm.copy(), YOLO_ANCHORS, 15, 224, 224, 3, 2, 0.5, 0.5)
    b, h, v = parse_prior_pose(torch.from_numpy(pm).to(gpu), YOLO_ANCHORS, 15, 224, 224, 3, 2, 0.5, 0.5)
    for i in range(32):
        assert len(b[i]) == len(rb[i])
        if len(b[i]):
            assert np.array_equal(np.array(b[i]), np.array(rb[i]))
            assert np.array_equal(np.array(h[i]), np.array(rh[i]))
            assert np.array_equal(np.array(v[i]), np.array(rv[i]))


def test_yolo_decode_pred_vis_bit_exact_vs_reference_golden(gpu, golden):
    """parse_prior_pose(pred_vis=True) (prior_pose_align.py:62,120,153-157; VERDICT r02 missing #6): maps with 5 + 4 J channels per
    anchor; boxes, skeletons and the float visibilities (in-bounds test x predicted visibility) equal the reference's own."""
    from helpers import yolo_maps_predvis
    from popnet_amd.utils.prior_pose_align import parse_prior_pose
    pm = yolo_maps_predvis(34)
    t = torch.from_numpy(pm.copy()).to(gpu)
    b, h, v = parse_prior_pose(t, [(6., 3.), (12., 6.)], 15, 224, 224, 3, 2, 0.5, 0.5, pred_vis=True)
    assert torch.equal(t.cpu(), torch.from_numpy(pm))                       # the input is not modified
    total = 0
    for i in range(pm.shape[0]):
        assert np.array_equal(np.array(b[i], np.float32).reshape(-1, 5), golden.yolo["pv_%d_bbox" % i])
        assert np.array_equal(np.array(h[i], np.float32).reshape(-1, 15, 3), golden.yolo["pv_%d_human" % i])
        got = np.array(v[i], np.float32).reshape(-1, 15)
        assert got.dtype == np.float32 and np.array_equal(got, golden.yolo["pv_%d_vis" % i])
        total += len(b[i])
    assert total >= 4
    with pytest.raises(_lib.PopnetError, match="channels"):               # a 3J-channel map in pred_vis mode is refused, not misread
        parse_prior_pose(torch.from_numpy(yolo_maps(31)).to(gpu), [(6., 3.), (12., 6.)], 15, 224, 224, 3, 2, 0.5, 0.5, pred_vis=True)


def test_yolo_glue_bit_exact_vs_oracle(gpu):
    """joints_2d / joints_3d / bbox_org of pn_yolo_frame == the oracle restatement of the evaluation
    script's per-frame glue (float32), bit for bit."""
    from oracle import parse_yolo as O
    from popnet_amd.config import INTRINSICS
    from popnet_amd.utils.paf_to_pose import make_parse_cfg
    from popnet_amd.utils.prior_pose_align import parse_yolo_batch
    rng = np.random.default_rng(9)
    pm = rng.uniform(-1, 1, (8, 100, 14, 14)).astype(np.float32)
    pm[:, 4] = rng.uniform(0, 0.56, (8, 14, 14))
    pm[:, 54] = rng.uniform(0, 0.53, (8, 14, 14))
    pm[:, 2:4] = rng.uniform(0.5, 2, (8, 2, 14, 14))
    pm[:, 52:54] = rng.uniform(0.5, 2, (8, 2, 14, 14))
    rb, rh, _ = O.parse_prior_pose(pm.copy(), YOLO_ANCHORS, 15, 224, 224, 3, 2, 0.5, 0.5)
    cfg = make_parse_cfg(None, input_size=224, w_org=480, h_org=640, intrinsics=INTRINSICS)
    recs = parse_yolo_batch(torch.from_numpy(pm).to(gpu), YOLO_ANCHORS, 15, 224, 224, 3, 2, 0.5, 0.5, glue_cfg=cfg)
    seen = 0
    for i in range(8):
        n = int(recs[i]["n_det"])
        assert n == len(rb[i]) and int(recs[i]["status"]) == 0
        if n:
            g = O.frame_glue(rb[i], rh[i], 15, 224, 480, 640, INTRINSICS)
            assert np.array_equal(recs[i]["joints_2d"][:n], g["humans_2d"])
            assert np.array_equal(recs[i]["joints_3d"][:n], g["humans_3d"])
            assert np.array_equal(recs[i]["bbox_org"][:n], g["bboxes"][:, :4])
            assert np.array_equal(recs[i]["bbox"][:n, 4].astype(np.float64), g["part_conf"][:, 0])
            seen += n
    assert seen > 8


def test_yolo_end_to_end_vs_reference_eval_script(gpu, golden):
    """Depth frames -> YoloEngine (fp32 parity mode) == eval_data.json of the reference's own
    evaluation_yolo_posenet_kdh3d_mpreal.py on the same frames and weights: same detections in the
    same order, 3D joints within 1e-3 m (north_star tolerance), confidences 1e-4, 2D within 0.05 px (one unit
    of the network's joint channel spans 6 cells x 16 px x 640/224 = 274 px, so the 2e-4 fp32 forward
    tolerance maps to 0.05 px)."""
    from popnet_amd.pipeline import YoloEngine
    s = golden.script_yolo
    sd = state_dict_from_keys(golden.keys["yolo_posenet"], seed=s["weight_seed"])
    sd["model2_4.0.weight"][[4, 54]] -= np.float32(s["conf_weight_shift"])
    eng = YoloEngine(precision="fp32", state_dict=sd, device=gpu, max_batch=2)
    frames = synth.synth_depth(2, 640, 480, seed=s["depth_seed"])
    recs = eng.predict_host(torch.from_numpy(frames).to(gpu))
    total = 0
    for b in range(2):
        fr = recs[b]
        n = int(fr["n_det"])
        assert n == len(s["human_pred_set_2d"][b]) and int(fr["status"]) == 0
        if n:
            assert np.abs(fr["joints_2d"][:n] - np.array(s["human_pred_set_2d"][b])).max() < 5e-2
            assert np.abs(fr["joints_3d"][:n] - np.array(s["human_pred_set_3d"][b])).max() < 1e-3
            assert np.abs(fr["bbox"][:n, 4:5] - np.array(s["human_pred_set_part_conf"][b])).max() < 1e-4
        total += n
    assert total >= 2


def test_yolo_engine_bf16_full_batch_decode_is_self_consistent(gpu):
    """BASELINE configs[1] shape for Yolo-Pose+ (32 frames, bf16): the records must equal the ORACLE
    decode of the map the HIP forward produced, bit for bit."""
    from oracle import parse_yolo as O
    from popnet_amd.config import INTRINSICS
    from popnet_amd.pipeline import YoloEngine
    eng = YoloEngine(precision="bf16", device=gpu, max_batch=32)
    depth = torch.from_numpy(synth.synth_depth(32, 640, 480, seed=5)).to(gpu)
    recs = eng.predict_host(depth)
    out = eng.out.cpu().numpy()
    assert np.isfinite(out).all()
    rb, rh, rv = O.parse_prior_pose(out.copy(), YOLO_ANCHORS, 15, 224, 224, 3, 2, 0.5, 0.5)
    ndet = 0
    for b in range(32):
        n = int(recs[b]["n_det"])
        assert int(recs[b]["status"]) == 0 and n == len(rb[b])
        if n:
            g = O.frame_glue(rb[b], rh[b], 15, 224, 480, 640, INTRINSICS)
            assert np.array_equal(recs[b]["bbox"][:n], np.array(rb[b]))
            assert np.array_equal(recs[b]["human"][:n], np.array(rh[b]))
            assert np.array_equal(recs[b]["joints_3d"][:n], g["humans_3d"])
        ndet += n
    assert 8 <= ndet <= 32 * 12, ndet


# ---------------------------------------------------------------------------------------------
# legacy pafprocess plug-in ABI
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed,P", PAFPROCESS_CASES)
def test_pafprocess_abi_vs_compiled_reference_golden(gpu, golden, seed, P):
    from popnet_amd import pafprocess
    pk, heat, paf = coco_case(seed, P)
    got = humans_to_array(pafprocess.run(pk, heat, paf))
    assert np.array_equal(got, golden.pafprocess["s%d_p%d" % (seed, P)])


@pytest.mark.parametrize("seed,P", CPP_CASES)
def test_paf_to_pose_cpp_vs_reference_function_golden(gpu, golden, seed, P):
    """popnet_amd.utils.paf_to_pose.paf_to_pose_cpp (NMS on the GPU -> INTER_NEAREST x8 -> process_paf + getters in libpopnet_hip.so) ==
    the reference's own function (tpm/lib/utils/paf_to_pose.py:381-415) with its compiled pafprocess.cpp behind it: the NMS rows and
    every field of every Human / BodyPart, bit for bit."""
    from types import SimpleNamespace
    from popnet_amd import synth
    from popnet_amd.utils import paf_to_pose as P2P
    cfg = SimpleNamespace(MODEL=SimpleNamespace(DOWNSAMPLE=8, NUM_KEYPOINTS=18), TEST=SimpleNamespace(THRESH_HEATMAP=0.1))
    heat, paf = synth.coco_maps(seed, P)
    g = golden.cpp
    nms = P2P.NMS(heat.copy(), upsampFactor=8, config=cfg)
    rows = np.array([tuple(pk) + (j,) for j, pks in enumerate(nms) for pk in pks], dtype=np.float64).reshape(-1, 5)
    assert np.array_equal(rows, g["s%d_p%d_nms" % (seed, P)])
    humans = P2P.paf_to_pose_cpp(heat.copy(), paf.copy(), cfg)
    got = -np.ones((len(humans), 1 + 18 * 3))
    for i, h in enumerate(humans):
        got[i, 0] = h.score
        for p, bp in h.body_parts.items():
            assert bp.part_idx == p and bp.uidx.endswith("-%d" % p)
            got[i, 1 + 3 * p:4 + 3 * p] = (bp.x, bp.y, bp.score)
    assert np.array_equal(got, g["s%d_p%d" % (seed, P)])
    assert (len(humans) > 0) == (P > 0)


def test_nms_peaks_has_no_capacity_and_keeps_the_reference_order(gpu):
    """pn_nms_peaks (the NMS behind paf_to_pose_cpp) against the oracle's restatement of paf_to_pose.py:75-153 on maps the fixed-size parse
    records could not hold: a plateau-riddled map with hundreds of peaks, an empty map, a single-cell plateau pair at the border, 21 maps."""
    from types import SimpleNamespace
    from oracle import parse_paf as O
    from popnet_amd.utils import paf_to_pose as P2P
    rng = np.random.default_rng(77)
    h, w, nk = 28, 28, 21
    heat = rng.uniform(0.0, 0.09, (h, w, nk)).astype(np.float32)
    heat[:, :, 0] = (rng.integers(0, 3, (h, w)) * 0.25 + 0.2).astype(np.float32)         # three-level plateaus: > 100 peaks
    heat[:, :, 1] = 0.0                                                                   # nothing above the threshold
    heat[0, 0, 2] = heat[0, 1, 2] = 0.7                                                   # two-cell plateau in the corner
    heat[h - 1, w - 1, 3] = 0.9
    for j in range(4, nk):
        for _ in range(int(rng.integers(0, 6))):
            heat[rng.integers(0, h), rng.integers(0, w), j] = rng.uniform(0.2, 1.0)
    cfg = SimpleNamespace(MODEL=SimpleNamespace(DOWNSAMPLE=8, NUM_KEYPOINTS=nk), TEST=SimpleNamespace(THRESH_HEATMAP=0.1))
    got = P2P.NMS(heat.copy(), upsampFactor=8, config=cfg)
    ref = O.nms(heat.copy(), num_keypoints=nk)
    assert len(got) == nk and len(got[0]) > 100 and len(got[1]) == 0
    for j in range(nk):
        assert got[j].shape == ref[j].shape and np.array_equal(got[j], ref[j]), j


def test_pafprocess_rejects_out_of_range_peaks(gpu):
    from popnet_amd import pafprocess
    pk = np.array([[[500, 3, 0.9, 0, 1]]], np.float32)        # x outside the 216-wide map: the reference reads out of bounds
    with pytest.raises(_lib.PopnetError):
        pafprocess.process_paf(pk, np.zeros((184, 216, 19), np.float32), np.zeros((184, 216, 38), np.float32))


# ---------------------------------------------------------------------------------------------
# whole path
# ---------------------------------------------------------------------------------------------
def test_end_to_end_vs_reference_eval_script(gpu, golden):
    """Depth frames -> PoseEngine (fp32 parity mode) == eval_data.json of the reference's own
    evaluation script on the same frames and weights: person assignment / visibility exact, 2D exact,
    3D joints within 1e-3 m (north_star tolerance)."""
    from popnet_amd.pipeline import PoseEngine
    s = golden.script
    sd = state_dict_from_keys(golden.keys["rtpose_light3d"], seed=s["weight_seed"])
    sd["model2_2.12.bias"][:15] += torch.tensor(s["heat_bias_shift"])
    eng = PoseEngine(precision="fp32", state_dict=sd, device=gpu, max_batch=2)
    frames = synth.synth_depth(2, 640, 480, seed=s["depth_seed"])
    recs = eng.predict_host(torch.from_numpy(frames).to(gpu))
    for b in range(2):
        fr = recs[b]
        n = int(fr["n_persons"])
        assert n == len(s["human_pred_set_2d"][b]) and int(fr["status"]) == 0
        vis = (fr["person_joint"][:n] >= 0).astype(int)
        assert vis.tolist() == s["human_pred_set_visibility"][b]
        assert np.allclose(fr["joints_2d"][:n], np.array(s["human_pred_set_2d"][b]), atol=1e-9)
        assert np.abs(fr["joints_3d"][:n] - np.array(s["human_pred_set_3d"][b])).max() < 1e-3
        assert np.allclose(fr["part_conf"][:n], np.array(s["human_pred_set_part_conf"][b]), atol=1e-4)


def test_engine_bf16_full_batch_runs_and_parse_is_self_consistent(gpu):
    """BASELINE configs[1] (32 frames, bf16): the records must equal the ORACLE parse of the maps the
    HIP forward produced -- i.e. whatever bf16 does to the maps, the parse stage stays bit-exact."""
    from oracle import parse_paf as O
    from popnet_amd.pipeline import PoseEngine, records_to_numpy
    from popnet_amd.utils.paf_to_pose import frame_assoc, frame_joint_list
    eng = PoseEngine(precision="bf16", device=gpu, max_batch=32)
    depth = torch.from_numpy(synth.synth_depth(32, 640, 480, seed=5)).to(gpu)
    recs = records_to_numpy(eng.predict(depth))
    torch.cuda.synchronize()
    hp, hh, hz = (t.cpu().numpy().transpose(0, 2, 3, 1) for t in (eng.paf, eng.heat, eng.z))
    assert np.isfinite(hp).all() and np.isfinite(hh).all() and np.isfinite(hz).all()
    for b in range(0, 32, 5):
        rec = O.frame_to_records(hh[b].copy(), hp[b].copy(), hz[b].copy())
        jl, assoc = frame_joint_list(recs[b]), frame_assoc(recs[b])
        assert int(recs[b]["status"]) == 0
        assert jl.shape == np.asarray(rec["joint_list"]).shape and (jl.size == 0 or np.array_equal(jl, rec["joint_list"]))
        assert assoc.shape == np.asarray(rec["assoc"]).shape
        if assoc.size:
            assert np.array_equal(assoc[:, :15], rec["assoc"][:, :15])
            assert np.array_equal(recs[b]["joints_3d"][:assoc.shape[0]], np.array(rec["humans_3d"]))


def test_concurrent_engines_on_separate_streams_match_sequential(gpu):
    """bench.py keeps several batches in flight (one PoseEngine + HIP stream each): the records must not
    depend on what else is running."""
    from popnet_amd.pipeline import PoseEngine
    engs = [PoseEngine(precision="bf16", device=gpu, max_batch=8, private_ctx=True) for _ in range(3)]
    depths = [torch.from_numpy(synth.synth_depth(8, 640, 480, seed=40 + i)).to(gpu) for i in range(3)]
    ref = [e.predict(d).clone() for e, d in zip(engs, depths)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=gpu) for _ in range(3)]
    outs = [None] * 3
    for rep in range(4):
        for i in range(3):
            with torch.cuda.stream(streams[i]):
                outs[i] = engs[i].predict(depths[i]).clone()
    torch.cuda.synchronize()
    for i in range(3):
        assert torch.equal(outs[i], ref[i])


def test_hipgraph_replay_equals_eager(gpu):
    """bench.py replays one captured hipGraph per step: the replayed step must reproduce the eager
    maps and records exactly, for inputs different from the ones seen at capture time."""
    from popnet_amd.pipeline import PoseEngine
    eng = PoseEngine(precision="bf16", device=gpu, max_batch=32)
    item = _lib.POSE_FRAME_DTYPE.itemsize
    d1 = torch.from_numpy(synth.synth_depth(32, seed=1)).to(gpu)
    d2 = torch.from_numpy(synth.synth_depth(32, seed=2)).to(gpu)
    static_in = d1.clone()
    out = torch.zeros((32, item), device=gpu, dtype=torch.uint8)

    def body():
        eng.predict(static_in, out)

    def snapshot():
        torch.cuda.synchronize()
        r = out.cpu().numpy().view(_lib.POSE_FRAME_DTYPE).reshape(-1)
        return eng.heat.clone(), eng.paf.clone(), eng.z.clone(), [(int(f["n_peaks"]), int(f["n_persons"]), f["person_joint"][:int(f["n_persons"])].tobytes()) for f in r]

    for _ in range(2):
        body()
    eager = {}
    for name, d in (("d1", d1), ("d2", d2)):
        static_in.copy_(d)
        body()
        eager[name] = snapshot()
    side = torch.cuda.Stream(device=gpu)
    side.wait_stream(torch.cuda.current_stream(gpu))
    with torch.cuda.stream(side):
        body()
    torch.cuda.current_stream(gpu).wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        body()
    for _ in range(2):
        for name, d in (("d2", d2), ("d1", d1)):
            static_in.copy_(d)
            out.zero_(); eng.heat.zero_(); eng.paf.zero_(); eng.z.zero_()
            g.replay()
            h, p, z, recs = snapshot()
            assert torch.equal(h, eager[name][0]) and torch.equal(p, eager[name][1]) and torch.equal(z, eager[name][2])
            assert recs == eager[name][3]


# ---------------------------------------------------------------------------------------------
# the evaluation-script drop-in: labels.json + .npy frames in, eval_data.json + metrics out
# ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("net", ["rtpose", "yolo"])
def test_evaluate_mpreal_script_matches_reference_scripts(gpu, golden, tmp_path, net):
    """scripts/evaluate_mpreal.py on the same fake two-frame dataset and checkpoint the golden generator fed to
    the reference's evaluation scripts: eval_data.json must match theirs (assignment exact, 3D within 1e-3 m)."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("evaluate_mpreal", os.path.join(ROOT, "scripts", "evaluate_mpreal.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    s = golden.script if net == "rtpose" else golden.script_yolo
    key = "rtpose_light3d" if net == "rtpose" else "yolo_posenet"
    sd = state_dict_from_keys(golden.keys[key], seed=s["weight_seed"])
    if net == "rtpose":
        sd["model2_2.12.bias"][:15] += torch.tensor(s["heat_bias_shift"])
    else:
        sd["model2_4.0.weight"][[4, 54]] -= np.float32(s["conf_weight_shift"])
    torch.save({"module." + k: v for k, v in sd.items()}, tmp_path / "ckpt.pth")
    img_dir = tmp_path / "depth_maps"
    img_dir.mkdir()
    frames = synth.synth_depth(2, 640, 480, seed=s["depth_seed"])
    labels = {"intrinsics": {"fx": 504.1189880371094, "fy": 504.042724609375, "cx": 231.7421875, "cy": 320.62640380859375}}
    rng = np.random.default_rng(5)
    for i in range(2):
        np.save(img_dir / ("f%d.npy" % i), frames[i])
        j2 = rng.uniform(50, 400, (15, 2))
        labels["f%d.npy" % i] = [{"2d_joints": j2.tolist(), "3d_joints": np.c_[j2 / 200, np.full(15, 3.0)].tolist()}]
    json.dump(labels, open(tmp_path / "labels.json", "w"))
    out = mod.main(["--annotations", str(tmp_path / "labels.json"), "--image-dir", str(img_dir), "--w-org", "480", "--h-org", "640",
                    "--batch-size", "2", "--weight", str(tmp_path / "ckpt.pth"), "--output-dir", str(tmp_path / "out"), "--net", net])
    data = json.load(open(tmp_path / "out" / "eval_data.json"))
    assert len(data["human_pred_set_2d"]) == 2 and data["human_gt_set_2d"][0] == [labels["f0.npy"][0]["2d_joints"]]
    for b in range(2):
        want2, want3 = np.array(s["human_pred_set_2d"][b]), np.array(s["human_pred_set_3d"][b])
        got2, got3 = np.array(data["human_pred_set_2d"][b]), np.array(data["human_pred_set_3d"][b])
        assert got2.shape == want2.shape and got3.shape == want3.shape
        if got2.size:
            assert np.abs(got2 - want2).max() < (1e-9 if net == "rtpose" else 5e-2)
            assert np.abs(got3 - want3).max() < 1e-3
            assert np.abs(np.array(data["human_pred_set_part_conf"][b]) - np.array(s["human_pred_set_part_conf"][b])).max() < 1e-4
        if net == "rtpose":
            assert data["human_pred_set_visibility"][b] == s["human_pred_set_visibility"][b]
    assert out is not None and len(out["ap2d"]) == 16 and len(out["pck3d"]) == 15


def test_wire_records_match_full_records(gpu):
    """pn_pack_pose_frames: the compact records gathered across GPUs carry the same persons, the same joint assignment
    (exact) and the float32 rounding of the float64 values."""
    from popnet_amd.pipeline import PoseEngine, records_to_numpy, wire_to_lists
    from popnet_amd.dataset import pose_records_to_lists
    eng = PoseEngine(precision="bf16", device=gpu, max_batch=8)
    depth = torch.from_numpy(synth.synth_depth(8, 640, 480, seed=11)).to(gpu)
    frames = eng.predict(depth)
    wire = eng.pack(frames).cpu().numpy().view(_lib.POSE_WIRE_DTYPE).reshape(-1)
    full = records_to_numpy(frames)
    assert int(full["n_persons"].sum()) > 0
    for w, f in zip(wire, full):
        n = int(f["n_persons"])
        assert int(w["n_persons"]) == n and int(w["status"]) == int(f["status"]) and n <= _lib.PN_WIRE_MAX_PERSONS
        assert np.array_equal(w["person_joint"][:n], f["person_joint"][:n].astype(np.int16)) and np.all(w["person_joint"][n:] == -1)
        assert np.array_equal(w["vals"][:n, :, 0:2], f["joints_2d"][:n].astype(np.float32))
        assert np.array_equal(w["vals"][:n, :, 2:5], f["joints_3d"][:n].astype(np.float32))
        assert np.array_equal(w["vals"][:n, :, 5], f["part_conf"][:n].astype(np.float32))
    # pn_parse_paf_wire writes the compact records in the read-out launch itself: the same bytes as parse + pack
    fused = torch.zeros((8, _lib.POSE_WIRE_DTYPE.itemsize), device=gpu, dtype=torch.uint8)
    frames2 = eng.predict(depth, torch.empty_like(frames), fused)
    torch.cuda.synchronize()
    assert torch.equal(frames2, frames) and torch.equal(fused, eng.pack(frames))
    a, b = wire_to_lists(wire), pose_records_to_lists(full)
    assert a["human_pred_set_visibility"] == b["human_pred_set_visibility"]
    assert np.allclose(np.array(a["human_pred_set_3d"][0]), np.array(b["human_pred_set_3d"][0]), atol=1e-5) or len(a["human_pred_set_3d"][0]) == 0


def test_full_batch_permutation_equivariance_and_single_frame_consistency(gpu):
    """BASELINE configs[1] size (32 frames, bf16), size-independent properties of the whole path: frames are
    independent, so permuting the batch permutes the records bit for bit, and a frame processed alone gives the record
    it got inside the full batch (different tile counts, same per-frame arithmetic)."""
    from popnet_amd.pipeline import PoseEngine
    eng = PoseEngine(precision="bf16", device=gpu, max_batch=32)
    depth = torch.from_numpy(synth.synth_depth(32, 640, 480, seed=21)).to(gpu)
    base = eng.predict(depth).clone()
    perm = torch.from_numpy(np.random.default_rng(3).permutation(32)).to(gpu)
    shuffled = eng.predict(depth[perm]).clone()
    singles = [eng.predict(depth[i:i + 1]).clone() for i in (0, 13, 31)]
    torch.cuda.synchronize()
    assert torch.equal(shuffled, base[perm])
    for i, s in zip((0, 13, 31), singles):
        assert torch.equal(s[0], base[i])
    recs = base.cpu().numpy().view(_lib.POSE_FRAME_DTYPE).reshape(-1)
    assert int((recs["status"] != 0).sum()) == 0 and int(recs["n_peaks"].sum()) > 0


def test_yolo_full_batch_permutation_equivariance(gpu):
    from popnet_amd.pipeline import YoloEngine
    eng = YoloEngine(precision="bf16", device=gpu, max_batch=32)
    depth = torch.from_numpy(synth.synth_depth(32, 640, 480, seed=22)).to(gpu)
    base = eng.predict(depth).clone()
    perm = torch.from_numpy(np.random.default_rng(4).permutation(32)).to(gpu)
    shuffled = eng.predict(depth[perm]).clone()
    single = eng.predict(depth[7:8]).clone()
    torch.cuda.synchronize()
    assert torch.equal(shuffled, base[perm]) and torch.equal(single[0], base[7])


@pytest.mark.parametrize("shape,dtype", [((240, 320), np.float16), ((240, 320), np.float32), ((512, 480), np.float16)])
def test_engine_other_frame_sizes_and_dtypes(gpu, shape, dtype):
    """ITOP-sized (240x320) and 512-row frames, f16 and f32 storage: pre-processing equals the oracle bit for bit and the
    records equal the oracle parse of the maps (the rescale uses the engine's w_org / h_org)."""
    from oracle import parse_paf as O, preproc as opre
    from popnet_amd.pipeline import PoseEngine, records_to_numpy
    from popnet_amd.utils.paf_to_pose import frame_assoc
    H, W = shape
    eng = PoseEngine(precision="fp32", device=gpu, max_batch=3, w_org=W, h_org=H)
    depth = synth.synth_depth(3, H, W, seed=31, dtype=dtype)
    recs = records_to_numpy(eng.predict(torch.from_numpy(depth).to(gpu)))
    torch.cuda.synchronize()
    assert np.array_equal(eng.x[:3].cpu().numpy(), opre.preprocess_batch(depth))
    hp, hh, hz = (t[:3].cpu().numpy().transpose(0, 2, 3, 1) for t in (eng.paf, eng.heat, eng.z))
    compared = 0
    for b in range(3):
        if int(recs[b]["status"]):          # the head calibration is for 480x640 statistics: a crowded map may hit the
            continue                        # compile-time person-row limit, which is flagged, never silent
        compared += 1
        ref = O.frame_to_records(hh[b].copy(), hp[b].copy(), hz[b].copy(), w_org=W, h_org=H)
        a = frame_assoc(recs[b])
        assert a.shape[0] == len(ref["humans_3d"])
        if a.shape[0]:
            assert np.array_equal(recs[b]["joints_2d"][:a.shape[0]], np.array(ref["humans_2d"]).reshape(-1, 15, 2))
            assert np.array_equal(recs[b]["joints_3d"][:a.shape[0]], np.array(ref["humans_3d"]).reshape(-1, 15, 3))
    assert compared >= 1


@pytest.mark.parametrize("prec", ["bf16", "bf16x3"])
@pytest.mark.parametrize("shape,dtype", [((640, 480), np.float16), ((240, 320), np.float32), ((512, 480), np.float16)])
def test_frames_in_forward_equals_preprocess_plus_forward(gpu, prec, shape, dtype):
    """pn_rtpose_forward_frames / pn_yolo_forward_frames (the stem computes its input tile from the raw frames with pn_preprocess's
    arithmetic) give the SAME maps, bit for bit, as pn_preprocess + pn_*_forward, for f16 and f32 frames of three sizes; the fp32
    parity net refuses the call (it keeps the two-call form)."""
    import ctypes as C
    from popnet_amd import _lib
    from popnet_amd.pipeline import PoseEngine, YoloEngine
    H, W = shape
    depth = torch.from_numpy(synth.synth_depth(3, H, W, seed=37, dtype=dtype)).to(gpu)
    eng = PoseEngine(precision=prec, device=gpu, max_batch=3, w_org=W, h_org=H)
    B = eng.preprocess(depth)
    eng.forward(B)
    two = [t[:3].clone() for t in (eng.paf, eng.heat, eng.z)]
    for t in (eng.paf, eng.heat, eng.z):
        t.zero_()
    assert eng.forward_frames(depth) == 3
    torch.cuda.synchronize()
    for a, b, name in zip(two, (eng.paf, eng.heat, eng.z), ("paf", "heat", "z")):
        assert torch.isfinite(a).all() and torch.equal(a, b[:3]), name
    if prec == "bf16":
        ye = YoloEngine(precision=prec, device=gpu, max_batch=3, w_org=W, h_org=H)
        B = ye.preprocess(depth)
        ye.forward(B)
        ref = ye.out[:3].clone()
        ye.out.zero_()
        ye.forward_frames(depth)
        torch.cuda.synchronize()
        assert torch.equal(ref, ye.out[:3])
        f32 = PoseEngine(precision="fp32", device=gpu, max_batch=3, w_org=W, h_org=H)
        rc = _lib.lib().pn_rtpose_forward_frames(f32.net, C.c_void_p(depth.data_ptr()), _lib.PN_DEPTH_F16 if dtype == np.float16 else _lib.PN_DEPTH_F32, 3, H, W,
                                                 6.0, 3.0, 2.0, C.c_void_p(f32.paf.data_ptr()), C.c_void_p(f32.heat.data_ptr()), C.c_void_p(f32.z.data_ptr()), None)
        assert rc == -4                                 # PN_ERR_UNSUPPORTED (include/popnet_hip.h)


@pytest.mark.parametrize("net", ["rtpose", "yolo"])
def test_end_to_end_metrics_match_reference_script_and_metric_code(gpu, golden, tmp_path, net):
    """The closest thing to 'mAP / PCK identical on a test split' that can be shown without the dataset: 12 synthetic
    frames + a checkpoint through scripts/evaluate_mpreal.py (HIP path + popnet_amd.metrics) against the numbers the
    REFERENCE produced for the same frames, checkpoint and labels with its evaluation script and its own util/eval_*.py
    (tests/golden/make_golden.py::golden_script_metrics).  Same persons per frame; PCK / AP equal up to the effect of the
    2e-4 fp32 forward tolerance on joints that sit on a threshold (none do here: exact to 1e-6)."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("evaluate_mpreal", os.path.join(ROOT, "scripts", "evaluate_mpreal.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    s = golden.script_metrics if net == "rtpose" else golden.script_metrics_yolo
    sd = state_dict_from_keys(golden.keys["rtpose_light3d" if net == "rtpose" else "yolo_posenet"], seed=s["weight_seed"])
    if net == "rtpose":
        sd["model2_2.12.bias"][:15] += torch.tensor(s["heat_bias_shift"])
    else:
        sd["model2_4.0.weight"][[4, 54]] -= np.float32(s["conf_weight_shift"])
    torch.save({"module." + k: v for k, v in sd.items()}, tmp_path / "ckpt.pth")
    img_dir = tmp_path / "depth_maps"
    img_dir.mkdir()
    N = s["n_frames"]
    frames = synth.synth_depth(N, 640, 480, seed=s["depth_seed"])
    labels = {"intrinsics": {"fx": 504.1189880371094, "fy": 504.042724609375, "cx": 231.7421875, "cy": 320.62640380859375}}
    for i in range(N):
        np.save(img_dir / ("f%02d.npy" % i), frames[i])
        labels["f%02d.npy" % i] = [{"2d_joints": a, "3d_joints": b} for a, b in zip(s["gt_2d"][i], s["gt_3d"][i])]
    json.dump(labels, open(tmp_path / "labels.json", "w"))
    out = mod.main(["--annotations", str(tmp_path / "labels.json"), "--image-dir", str(img_dir), "--batch-size", "4",
                    "--weight", str(tmp_path / "ckpt.pth"), "--output-dir", str(tmp_path / "out"), "--net", net])
    data = json.load(open(tmp_path / "out" / "eval_data.json"))
    assert [len(f) for f in data["human_pred_set_2d"]] == s["persons_per_frame"]
    for key, tol in (("pck2d", 1e-6), ("pck3d", 1e-6), ("ap2d", 1e-4), ("ap3d", 1e-4), ("err2d", 1e-6 if net == "rtpose" else 2e-2), ("err3d", 1e-4)):
        a, b = np.array(out[key], dtype=np.float64), np.array(s[key], dtype=np.float64)
        assert a.shape == b.shape and np.all((np.abs(a - b) <= tol) | (np.isnan(a) & np.isnan(b))), (key, np.nanmax(np.abs(a - b)))


@pytest.mark.parametrize("graph", [True, False])
def test_streaming_engine_tickets_and_records(gpu, graph):
    """StreamingEngine (what bench.py drives): three batches in flight, every ticket's records equal what a plain
    PoseEngine returns for that slot's input, graph replay or eager."""
    from popnet_amd.pipeline import PoseEngine, StreamingEngine
    se = StreamingEngine(PoseEngine, depth=3, graph=graph, precision="bf16", device=gpu, max_batch=4)
    plain = PoseEngine(precision="bf16", device=gpu, max_batch=4)
    batches = [torch.from_numpy(synth.synth_depth(4, 640, 480, seed=60 + i)).to(gpu) for i in range(3)]
    want = [plain.predict(b).clone() for b in batches]
    for i in range(3):
        se.input(i).copy_(batches[i])
    torch.cuda.synchronize()
    se.capture()
    tickets = [se.submit() for _ in range(9)]
    for t in tickets[-3:]:
        se.wait(t)
        assert torch.equal(se.records(t), want[t % 3])
        assert torch.equal(se.host_records(t), want[t % 3].cpu())
    se.join()
    torch.cuda.synchronize()


@pytest.mark.parametrize("graph", [True, False])
def test_streaming_engine_host_handover(gpu, graph):
    """StreamingEngine.submit_host (bench.py's h2d_inclusive mode): pinned host batches copied on the copy stream into the
    slot's next input buffer while earlier steps run; 11 DIFFERENT batches through 3 slots x 2 buffers -- every ticket's
    records equal a plain engine's for that batch (no copy lands in a buffer a running step still reads)."""
    from popnet_amd.pipeline import PoseEngine, StreamingEngine
    se = StreamingEngine(PoseEngine, depth=3, pool=2, graph=graph, precision="bf16", device=gpu, max_batch=4)
    plain = PoseEngine(precision="bf16", device=gpu, max_batch=4)
    hosts = [torch.from_numpy(synth.synth_depth(4, 640, 480, seed=160 + i)).pin_memory() for i in range(11)]
    want = [plain.predict(h.to(gpu)).clone() for h in hosts]
    se.capture()
    got = []
    for i, h in enumerate(hosts):
        t = se.submit_host(h)
        with torch.cuda.stream(se.stream(t)):
            got.append(se.records(t).clone())        # on the slot's stream, right behind the step
    se.join()
    torch.cuda.synchronize()
    for i in range(11):
        assert torch.equal(got[i], want[i]), i


@pytest.mark.parametrize("noise,drop,sigma", [(0.03, 0.0, 0.8), (0.06, 0.2, 1.2), (0.10, 0.35, 0.6)])
def test_parse_fuzz_noisy_planted_maps_vs_oracle(gpu, noise, drop, sigma):
    """Harsher planted maps than the golden cases (more noise -> spurious peaks and weak limbs, dropped joints, wider /
    narrower blobs): joint list, assignment, 2D / 3D joints and confidences still equal the oracle bit for bit."""
    from oracle import parse_paf as O
    from popnet_amd.utils.paf_to_pose import frame_assoc, frame_joint_list, make_parse_cfg, parse_paf_batch
    persons = [(3 * i + 1) % 8 for i in range(16)]
    heat, paf, z = synth.planted_batch(int(noise * 1000) + 5, persons, noise=noise, drop_prob=drop, sigma=sigma)
    frames = parse_paf_batch(*(torch.from_numpy(a).to(gpu) for a in (heat, paf, z)), make_parse_cfg(default_cfg()))
    checked = 0
    for b in range(16):
        fr = frames[b]
        if int(fr["status"]):
            continue                       # compile-time limits hit on a very noisy map: flagged, not compared
        checked += 1
        rec = O.frame_to_records(heat[b].transpose(1, 2, 0).copy(), paf[b].transpose(1, 2, 0).copy(), z[b].transpose(1, 2, 0).copy())
        jl, assoc = frame_joint_list(fr), frame_assoc(fr)
        assert jl.shape == np.asarray(rec["joint_list"]).shape and (jl.size == 0 or np.array_equal(jl, rec["joint_list"]))
        ref_assoc = np.asarray(rec["assoc"]).reshape(-1, 17)
        assert int(fr["n_persons"]) == ref_assoc.shape[0]
        if ref_assoc.shape[0]:
            n = ref_assoc.shape[0]
            assert np.array_equal(assoc[:, :15], ref_assoc[:, :15])
            assert np.array_equal(fr["joints_2d"][:n], np.array(rec["humans_2d"]))
            assert np.array_equal(fr["joints_3d"][:n], np.array(rec["humans_3d"]))
            assert np.array_equal(fr["part_conf"][:n], np.array(rec["conf"]))
    assert checked >= 12


def test_streaming_sweep_equals_sequential_sweep(gpu, tmp_path):
    """dataset.run_sweep_streaming (3 batches in flight, ragged tail batch) returns exactly the records of the plain
    sequential sweep, in frame order."""
    import json
    from popnet_amd import dataset
    from popnet_amd.pipeline import PoseEngine, StreamingEngine
    frames = synth.synth_depth(11, 640, 480, seed=91)
    labels = {"intrinsics": {"fx": 504.1189880371094, "fy": 504.042724609375, "cx": 231.7421875, "cy": 320.62640380859375}}
    for i in range(11):
        np.save(tmp_path / ("g%02d.npy" % i), frames[i])
        labels["g%02d.npy" % i] = []
    json.dump(labels, open(tmp_path / "labels.json", "w"))
    fr = dataset.MP3DHPFrames(str(tmp_path), str(tmp_path / "labels.json"))
    seq = dataset.run_sweep(PoseEngine(precision="bf16", device=gpu, max_batch=4), fr, 4)
    se = StreamingEngine(PoseEngine, depth=3, frame_hw=(640, 480), precision="bf16", device=gpu, max_batch=4)
    se.capture()
    got = dataset.run_sweep_streaming(se, fr, 4)
    assert len(got) == 11 and got.tobytes() == seq.tobytes()
