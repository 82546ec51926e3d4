"""Deterministic synthetic inputs shared by tests, golden-vector generation and bench.py.

No dataset or trained weights ship with the reference, so everything is seeded:
  * ``fill_state_dict``   weights keyed by parameter NAME (independent of module construction
                          order, so the reference modules and ours get identical tensors);
  * ``planted_maps``      network-output-like maps with P planted 15-joint skeletons
                          (Gaussian heat blobs, unit-vector PAF ribbons, constant z per person);
  * ``synth_depth``       depth frames shaped like MP-3DHP (480 wide x 640 high, metres, f16).
"""
import zlib

import numpy as np

from .config import LIMBS


def _rng_for(name, seed):
    return np.random.default_rng([zlib.crc32(name.encode()) & 0xffffffff, seed])


def fill_state_dict(sd, seed=0, gain=1.0):
    """Fills (a copy of) a torch state_dict in place-compatible form; returns {name: ndarray}.
    Conv weights ~ N(0, gain*sqrt(2/fan_in)) so activations keep O(1) scale through the stack
    (the reference's N(0, 0.01) init collapses every map to sigmoid(0)); BatchNorm affine and
    running statistics are randomised so folding is actually exercised."""
    out = {}
    for name, t in sd.items():
        shape = tuple(t.shape)
        rng = _rng_for(name, seed)
        if name.endswith("num_batches_tracked"):
            out[name] = np.zeros(shape, dtype=np.int64)
        elif name.endswith("running_mean"):
            out[name] = rng.normal(0, 0.05, shape).astype(np.float32)
        elif name.endswith("running_var"):
            out[name] = rng.uniform(0.8, 1.2, shape).astype(np.float32)
        elif len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            out[name] = (rng.standard_normal(shape) * gain * np.sqrt(2.0 / fan_in)).astype(np.float32)
        elif name.endswith(".weight"):            # BatchNorm gamma
            out[name] = rng.uniform(0.8, 1.2, shape).astype(np.float32)
        else:                                      # conv / BatchNorm bias
            out[name] = rng.normal(0, 0.05, shape).astype(np.float32)
    return out


def load_synth_weights(model, seed=0, gain=1.0):
    import torch
    arrays = fill_state_dict(model.state_dict(), seed, gain)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in arrays.items()})
    return model


# a standing person in a unit box: (x, y) of the 15 joints, y down
_TEMPLATE = np.array([
    [0.50, 0.05],  # head
    [0.50, 0.18],  # neck
    [0.36, 0.22], [0.64, 0.22],   # shoulders r, l
    [0.30, 0.38], [0.70, 0.38],   # elbows
    [0.27, 0.53], [0.73, 0.53],   # wrists
    [0.50, 0.45],  # torso
    [0.42, 0.55], [0.58, 0.55],   # hips
    [0.40, 0.75], [0.60, 0.75],   # knees
    [0.39, 0.95], [0.61, 0.95],   # ankles
])


def planted_persons(rng, n_persons, size=224, min_h=110, max_h=200):
    """Joint positions [P,15,2] in input-pixel coordinates plus a depth per person."""
    persons = []
    for _ in range(n_persons):
        hgt = rng.uniform(min_h, max_h)
        wid = hgt * rng.uniform(0.45, 0.6)
        x0 = rng.uniform(2, size - wid - 2)
        y0 = rng.uniform(2, size - hgt - 2)
        jit = rng.normal(0, 0.01, _TEMPLATE.shape)
        pts = (_TEMPLATE + jit) * [wid, hgt] + [x0, y0]
        persons.append(np.clip(pts, 1, size - 2))
    return np.array(persons).reshape(n_persons, 15, 2), rng.uniform(1.5, 4.5, n_persons)


def planted_maps(seed, n_persons, h=28, w=28, stride=8, sigma=0.8, noise=0.01, drop_prob=0.0):
    """Returns (heat [h,w,16], paf [h,w,28], z [h,w,15]) float32 HWC, like the network emits them
    (heat in (0,1), paf in (-2,2), z normalised), with n_persons planted skeletons."""
    rng = np.random.default_rng([seed, n_persons, 7])
    joints, depths = planted_persons(rng, n_persons, size=h * stride)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    heat = np.zeros((h, w, 16))
    paf = np.zeros((h, w, 28))
    cnt = np.zeros((h, w, 14))
    z = np.zeros((h, w, 15))
    for p in range(n_persons):
        jc = joints[p] / stride - 0.5 + 0.5            # joint position in cell units (cell centre = +0.5)
        present = rng.random(15) >= drop_prob
        for j in range(15):
            if not present[j]:
                continue
            g = np.exp(-((xx + 0.5 - jc[j, 0]) ** 2 + (yy + 0.5 - jc[j, 1]) ** 2) / (2 * sigma ** 2))
            heat[:, :, j] = np.maximum(heat[:, :, j], 0.9 * g)
            near = g > 0.05
            z[:, :, j][near] = (depths[p] - 3.0) / 2.0
        for l, (a, b) in enumerate(LIMBS):
            if not (present[a] and present[b]):
                continue
            pa, pb = jc[a], jc[b]
            d = pb - pa
            n = np.hypot(*d)
            if n < 1e-6:
                continue
            u = d / n
            rx, ry = xx + 0.5 - pa[0], yy + 0.5 - pa[1]
            along = rx * u[0] + ry * u[1]
            across = np.abs(rx * u[1] - ry * u[0])
            m = (along >= -0.5) & (along <= n + 0.5) & (across <= 1.0)
            paf[:, :, 2 * l][m] += u[0]
            paf[:, :, 2 * l + 1][m] += u[1]
            cnt[:, :, l][m] += 1
    for l in range(14):
        nz = cnt[:, :, l] > 0
        paf[:, :, 2 * l][nz] /= cnt[:, :, l][nz]
        paf[:, :, 2 * l + 1][nz] /= cnt[:, :, l][nz]
    heat[:, :, 15] = 1.0 - heat[:, :, :15].max(axis=2)
    heat += rng.uniform(0, noise, heat.shape)
    paf += rng.normal(0, noise, paf.shape)
    z += rng.normal(0, noise, z.shape)
    return (np.clip(heat, 0, 1).astype(np.float32), np.clip(paf, -2, 2).astype(np.float32),
            np.clip(z, -2, 2).astype(np.float32))


# COCO-18 topology of the `pafprocess` plug-in (tpm/lib/pafprocess/pafprocess.h:8-13): limb l joins parts COCO_PAIRS[l], its PAF x / y
# channels are COCO_PAIRS_NET[l]
COCO_PAIRS = [(1, 2), (1, 5), (2, 3), (3, 4), (5, 6), (6, 7), (1, 8), (8, 9), (9, 10), (1, 11), (11, 12), (12, 13), (1, 0),
              (0, 14), (14, 16), (0, 15), (15, 17), (2, 16), (5, 17)]
COCO_PAIRS_NET = [(12, 13), (20, 21), (14, 15), (16, 17), (22, 23), (24, 25), (0, 1), (2, 3), (4, 5), (6, 7), (8, 9), (10, 11),
                  (28, 29), (30, 31), (34, 35), (32, 33), (36, 37), (18, 19), (26, 27)]
_COCO_TMPL = np.array([[.5, .08], [.5, .2], [.38, .22], [.33, .38], [.3, .52], [.62, .22], [.67, .38], [.7, .52], [.44, .55],
                       [.43, .75], [.42, .95], [.56, .55], [.57, .75], [.58, .95], [.47, .05], [.53, .05], [.43, .07], [.57, .07]])


def coco_maps(seed, n_persons, h=28, w=28, sigma=0.9, noise=0.01, drop_prob=0.08):
    """Network-resolution maps of the COCO-18 caller of the path (`paf_to_pose_cpp`, tpm/lib/utils/paf_to_pose.py:381-415): heat [h, w, 19]
    with a Gaussian bump per visible part (plus the background channel), paf [h, w, 38] with unit vectors along every limb whose two parts
    are present, small noise everywhere.  Seeded: golden generator and tests build the same arrays."""
    rng = np.random.default_rng(seed)
    heat = rng.normal(0, noise, (h, w, 19)).astype(np.float32)
    paf = rng.normal(0, noise, (h, w, 38)).astype(np.float32)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    for _ in range(n_persons):
        hgt = rng.uniform(0.55, 0.9) * h
        wid = hgt * 0.6
        x0, y0 = rng.uniform(0.5, max(0.6, w - wid - 0.5)), rng.uniform(0.5, max(0.6, h - hgt - 0.5))
        pts = _COCO_TMPL * [wid, hgt] + [x0, y0]
        keep = rng.random(18) >= drop_prob
        amp = rng.uniform(0.5, 0.95, 18)
        for j in range(18):
            if keep[j]:
                heat[:, :, j] = np.maximum(heat[:, :, j], (amp[j] * np.exp(-((xx - pts[j, 0]) ** 2 + (yy - pts[j, 1]) ** 2) / (2 * sigma ** 2))).astype(np.float32))
        for l, (a, b) in enumerate(COCO_PAIRS):
            if not (keep[a] and keep[b]):
                continue
            d = pts[b] - pts[a]
            n = float(np.hypot(*d))
            if n < 1e-6:
                continue
            u = d / n
            rx, ry = xx - pts[a, 0], yy - pts[a, 1]
            along, across = rx * u[0] + ry * u[1], np.abs(rx * u[1] - ry * u[0])
            m = (along >= -0.5) & (along <= n + 0.5) & (across <= 0.8)
            paf[:, :, COCO_PAIRS_NET[l][0]][m] = np.float32(u[0])
            paf[:, :, COCO_PAIRS_NET[l][1]][m] = np.float32(u[1])
    return heat, paf


def planted_batch(seed, persons_per_frame, **kw):
    """Stacks planted_maps into NCHW float32 arrays: heat [B,16,h,w], paf [B,28,h,w], z [B,15,h,w]."""
    hs, ps, zs = [], [], []
    for i, n in enumerate(persons_per_frame):
        hm, pf, zz = planted_maps(seed * 1000 + i, n, **kw)
        hs.append(hm.transpose(2, 0, 1)); ps.append(pf.transpose(2, 0, 1)); zs.append(zz.transpose(2, 0, 1))
    return np.stack(hs), np.stack(ps), np.stack(zs)


def synth_depth(B, H=640, W=480, seed=1234, dtype=np.float16):
    """depth ~ clip(N(3.0, 0.8), 0, 6) with 4 % zeros, plus a smooth component so
    the bilinear resize sees structure.  [B, H, W] metres."""
    rng = np.random.default_rng(seed)
    base = rng.normal(3.0, 0.8, (B, H // 16 + 1, W // 16 + 1))
    up = np.repeat(np.repeat(base, 16, axis=1), 16, axis=2)[:, :H, :W]
    d = np.clip(up + rng.normal(0, 0.05, (B, H, W)), 0, 6)
    d[rng.random((B, H, W)) < 0.04] = 0
    return d.astype(dtype)


class SynthSweep:
    """A synthetic MP-3DHP split of n DISTINCT frames without n files: frame i is pool frame i % len(pool) rolled by
    i // len(pool) rows, so every frame has its own content and a sweep can be checked for order and coverage.  Duck-types
    dataset.MP3DHPFrames for run_sweep / run_sweep_streaming (BASELINE configs[2]: the 4 484-frame test_mpreal sweep)."""

    def __init__(self, n_frames, pool=64, H=640, W=480, seed=4484):
        self.n = int(n_frames)
        self.pool = synth_depth(pool, H, W, seed=seed)
        self.ids = ["synth_%05d.npy" % i for i in range(self.n)]
        from .config import INTRINSICS
        self.intrinsics = dict(INTRINSICS)

    def __len__(self):
        return self.n

    def load(self, index):
        return np.roll(self.pool[index % len(self.pool)], index // len(self.pool), axis=0)

    def batches(self, indices, batch_size, drop_last=False):
        indices = list(indices)
        for s in range(0, len(indices), batch_size):
            chunk = indices[s:s + batch_size]
            if drop_last and len(chunk) < batch_size:
                return
            yield chunk, np.stack([self.load(i) for i in chunk])


def init_like_state_dict(seed=0):
    """The state a training run starts from: the module's own constructor (conv weights N(0, 0.01) as
    rtpose_light3d._initialize_weights_norm does, third_party_methods/lib/network/rtpose_light3d.py:358-362; BatchNorm 1 / 0)."""
    import torch
    from .network.rtpose_light3d import rtpose_light3d
    g = torch.random.get_rng_state()
    torch.manual_seed(seed)
    sd = {k: v.detach().clone() for k, v in rtpose_light3d(15, 14, 2, input_dim=1).state_dict().items()}
    torch.random.set_rng_state(g)
    return sd
