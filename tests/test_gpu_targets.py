"""HIP training-target kernels (pn_compose_depth, pn_rasterize_targets, popnet_amd.targets.mpaug_batch) against the oracle
and against the reference's own outputs in tests/golden/targets.npz.

Bars.  Compositor, z maps, foreground masks, PAF maps and the network input: bit-exact (min / max / products by 0 or 1 and
IEEE +,-,*,/,sqrt only, all evaluated in the reference's precision).  Confidence maps: exp() of a float64 argument rounded
to float32 -- the device's double exp may differ from glibc's in the last float64 bit, which survives the float32 rounding
only on a rounding boundary, so the bar is one float32 ulp at 1.0 (1.2e-7 absolute) and the test also counts how many
cells differ at all.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "targets.npz"))
HEAT_TOL = 1.2e-7


def _dev(a, gpu, dt=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t.to(dt) if dt is not None else t).to(gpu)


def _raster_case(gpu, kp2d, kp3d, depth):
    from popnet_amd import targets
    P = kp2d.shape[0]
    k2 = _dev(kp2d.astype(np.float32)[None], gpu)
    kz = _dev(kp3d[None, :, :, 2].astype(np.float64), gpu)
    n = torch.tensor([P], dtype=torch.int32, device=gpu)
    out = targets.rasterize_targets(k2, kz, n, _dev(depth.astype(np.float32)[None], gpu))
    return [o[0].permute(1, 2, 0).cpu().numpy() for o in out]


def test_rasterizer_equals_oracle_and_reference_goldens(gpu):
    from oracle import targets as ot
    differing = 0
    for ci in range(int(G["n_gt"])):
        kp2d, kp3d, depth = G["gt%d_kp2d" % ci], G["gt%d_kp3d" % ci], G["gt%d_depth" % ci]
        heat, paf, z, fg = _raster_case(gpu, kp2d, kp3d, depth)
        # the device takes float32 joints and a float32 resized frame (what __getitem__ hands get_ground_truth): oracle on the same
        oh, op, oz, of = ot.ground_truth(kp2d.astype(np.float32), kp3d, depth.astype(np.float32))
        assert np.array_equal(fg, of.astype(np.float32)), ci
        assert np.array_equal(z, oz.astype(np.float32)) and oz.dtype == np.float32, ci
        assert np.array_equal(paf, op.astype(np.float32)), (ci, np.abs(paf - op).max())
        assert np.abs(heat - oh.astype(np.float32)).max() <= HEAT_TOL, ci
        differing += int((heat != oh.astype(np.float32)).sum())
        # and against what the reference's get_ground_truth returned for the float64 inputs: same maps up to the float32
        # rounding of the inputs (fg is decided by integer windows -> identical)
        assert np.array_equal(fg, G["gt%d_fg" % ci].astype(np.float32))
        assert np.abs(heat - G["gt%d_heat" % ci]).max() < 2e-6 and np.abs(z - G["gt%d_z" % ci]).max() < 1e-6
    assert differing <= 8, differing      # of 5 * 16 * 784 cells


def test_rasterizer_batch_with_ragged_person_counts(gpu):
    """One launch over frames holding 0 / 1 / 3 / 5 / 2 persons padded to Pmax = 5 == the single-frame results."""
    from popnet_amd import targets
    n_gt = int(G["n_gt"])
    Pm = max(G["gt%d_kp2d" % ci].shape[0] for ci in range(n_gt))
    k2 = np.full((n_gt, Pm, 15, 2), 1.0e9, dtype=np.float32)        # garbage beyond n_persons must never be read as a person
    kz = np.full((n_gt, Pm, 15), -5.0)
    dr = np.zeros((n_gt, 28, 28), dtype=np.float32)
    for ci in range(n_gt):
        P = G["gt%d_kp2d" % ci].shape[0]
        k2[ci, :P], kz[ci, :P], dr[ci] = G["gt%d_kp2d" % ci], G["gt%d_kp3d" % ci][:, :, 2], G["gt%d_depth" % ci]
    n = torch.tensor([G["gt%d_kp2d" % ci].shape[0] for ci in range(n_gt)], dtype=torch.int32, device=gpu)
    outs = targets.rasterize_targets(_dev(k2, gpu), _dev(kz, gpu), n, _dev(dr, gpu))
    for ci in range(n_gt):
        single = _raster_case(gpu, G["gt%d_kp2d" % ci], G["gt%d_kp3d" % ci], G["gt%d_depth" % ci])
        for o, s in zip(outs, single):
            assert np.array_equal(o[ci].permute(1, 2, 0).cpu().numpy(), s), ci


def test_compositor_equals_oracle(gpu):
    from popnet_amd import targets
    from oracle import targets as ot
    rng = np.random.default_rng(11)
    B, S, H, W = 5, 3, 96, 72
    for dt in (np.float16, np.float32):
        d = rng.uniform(0.3, 5.9, (B, S, H, W)).astype(dt)
        m = (rng.uniform(0, 1, (B, S, H, W)) < 0.35).astype(np.uint8)
        m[0] = 0                                             # nobody in frame 0: pure background
        m[1, :, :, :] = 1                                    # everybody everywhere: pure z-buffer minimum
        bg = rng.uniform(0, 6, (B, H, W)).astype(dt)
        n_src = np.array([3, 3, 1, 2, 3], dtype=np.int32)    # ragged: unused sources must not leak in
        got = targets.compose_depth(_dev(d, gpu), _dev(m, gpu), _dev(n_src, gpu), _dev(bg, gpu)).cpu().numpy()
        for b in range(B):
            want, _ = ot.compose_depth(d[b, :n_src[b]], m[b, :n_src[b]], bg[b])
            assert np.array_equal(got[b], want.astype(np.float32)), (dt, b)
    with pytest.raises(Exception, match="CUDA/ROCm tensor"):
        targets.compose_depth(torch.zeros(1, 1, 4, 4), torch.zeros(1, 1, 4, 4, dtype=torch.uint8), torch.ones(1, dtype=torch.int32), torch.zeros(1, 4, 4))


def test_mpaug_batch_equals_the_reference_dataset_items(gpu):
    """End to end: composed frame -> resize/clamp/normalise -> stride-8 resize -> targets == what KDH3D_Keypoints.__getitem__
    returned (golden), both items in one batch."""
    from popnet_amd import targets
    n = int(G["n_items"])
    fd = np.stack([G["it%d_fg_depth" % i] for i in range(n)])
    fm = np.stack([G["it%d_fg_mask" % i] for i in range(n)])
    bg = np.stack([G["it%d_bg" % i] for i in range(n)])
    k2 = np.stack([G["it%d_kp2d_org" % i] for i in range(n)]).astype(np.float32)
    k3 = np.stack([G["it%d_kp3d" % i] for i in range(n)])
    ns = torch.full((n,), fd.shape[1], dtype=torch.int32, device=gpu)
    npers = torch.full((n,), k2.shape[1], dtype=torch.int32, device=gpu)
    x, heat, paf, z, fg = targets.mpaug_batch(_dev(fd, gpu), _dev(fm, gpu), ns, _dev(bg, gpu), _dev(k2, gpu), _dev(k3, gpu), npers)
    for i in range(n):
        assert np.array_equal(x[i].cpu().numpy(), G["it%d_image" % i]), i
        assert np.array_equal(fg[i].cpu().numpy(), G["it%d_fg" % i]), i
        assert np.array_equal(z[i].cpu().numpy(), G["it%d_z" % i]), i
        assert np.array_equal(paf[i].cpu().numpy(), G["it%d_paf" % i]), i
        assert np.abs(heat[i].cpu().numpy() - G["it%d_heat" % i]).max() <= HEAT_TOL, i


def test_rasterizer_at_training_batch_size_properties(gpu):
    """Batch 64 x 8 persons (the training configuration's batch): maps stay in range, background = 1 - max, PAF vectors have
    norm <= 1, the fg mask is binary, and a permutation of the frames permutes the outputs."""
    from popnet_amd import synth, targets
    rng = np.random.default_rng(3)
    B, P = 64, 8
    k2 = np.zeros((B, P, 15, 2), dtype=np.float32)
    kz = np.zeros((B, P, 15))
    for b in range(B):
        j, d = synth.planted_persons(rng, P)
        k2[b], kz[b] = j, d[:, None] + rng.normal(0, 0.05, (P, 15))
    dr = rng.uniform(0, 6, (B, 28, 28)).astype(np.float32)
    n = torch.tensor(rng.integers(0, P + 1, B), dtype=torch.int32, device=gpu)
    heat, paf, z, fg = targets.rasterize_targets(_dev(k2, gpu), _dev(kz, gpu), n, _dev(dr, gpu))
    assert float(heat.min()) >= 0 and float(heat.max()) <= 1
    assert float((heat[:, 15] - (1 - heat[:, :15].max(1).values).clamp(min=0)).abs().max()) <= HEAT_TOL      # the kernel takes 1 - max in float64 before rounding
    norm = (paf[:, 0::2] ** 2 + paf[:, 1::2] ** 2).sqrt()
    assert float(norm.max()) <= 1 + 1e-6
    assert set(torch.unique(fg).tolist()) <= {0.0, 1.0} and float(z.min()) >= -1.5 and float(z.max()) <= 1.5
    perm = torch.randperm(B, device=gpu)
    h2, p2, z2, f2 = targets.rasterize_targets(_dev(k2, gpu)[perm], _dev(kz, gpu)[perm], n[perm], _dev(dr, gpu)[perm])
    assert torch.equal(h2, heat[perm]) and torch.equal(p2, paf[perm]) and torch.equal(z2, z[perm]) and torch.equal(f2, fg[perm])
