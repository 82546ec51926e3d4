"""Shared helpers: seeded inputs identical to the ones tests/golden/make_golden.py fed the reference."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import popnet_amd  # noqa: E402,F401
from popnet_amd import synth  # noqa: E402

# must mirror tests/golden/make_golden.py
PARSE_CASES = [(1, 0), (2, 1), (3, 2), (4, 3), (5, 4), (6, 6), (7, 8), (8, 3), (9, 5)]
SPECIAL_CASES = ["border_plateau", "missing_joints", "crowded_noisy"]
YOLO_ANCHORS = [(6., 3.), (12., 6.)]


def state_dict_from_keys(keys, seed):
    """Seeded weights for a reference-format state_dict described by [[name, shape], ...]."""
    template = {k: torch.empty(tuple(s)) for k, s in keys}
    return {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(template, seed=seed).items()}


def parse_case_inputs(golden, name):
    if name in SPECIAL_CASES:
        return (golden.parse["in_%s_heat" % name], golden.parse["in_%s_paf" % name], golden.parse["in_%s_z" % name])
    s, p = name.split("_")[1:]
    heat, paf, z = synth.planted_maps(int(s[1:]), int(p[1:]))
    chk = golden.parse["%s_insum" % name]
    assert abs(float(heat.astype(np.float64).sum()) - chk[0]) < 1e-6 and abs(float(paf.astype(np.float64).sum()) - chk[1]) < 1e-6, \
        "planted maps differ from the ones the golden vectors were generated with"
    return heat, paf, z


def all_parse_case_names():
    return ["planted_s%d_p%d" % c for c in PARSE_CASES] + SPECIAL_CASES


def yolo_maps(seed, B=2, clusters=True):
    rng = np.random.default_rng(seed)
    pm = rng.uniform(-0.9, 0.9, (B, 100, 14, 14)).astype(np.float32)
    for a in (0, 1):
        pm[:, 50 * a + 2:50 * a + 4] = rng.uniform(0.6, 1.9, (B, 2, 14, 14))
        pm[:, 50 * a + 4] = rng.uniform(0.0, 0.45, (B, 14, 14))
        pm[:, 50 * a + 5:50 * a + 50] = rng.uniform(-1.9, 1.9, (B, 45, 14, 14))
    if clusters:
        for b in range(B):
            for _ in range(3):
                cy, cx = rng.integers(2, 12, 2)
                for dy in (0, 1):
                    for dx in (0, 1, 2):
                        a = int(rng.integers(0, 2))
                        pm[b, 50 * a + 4, cy + dy, cx + dx] = rng.uniform(0.55, 0.99)
                        pm[b, 50 * a + 0:50 * a + 2, cy + dy, cx + dx] = rng.uniform(-0.2, 0.2, 2)
                        pm[b, 50 * a + 2:50 * a + 4, cy + dy, cx + dx] = rng.uniform(1.5, 1.95, 2)
    return pm


def yolo_maps_predvis(seed, B=2):
    """[B, 2 x (5 + 4 x 15), 14, 14]: yolo_maps() with 15 predicted-visibility channels appended to every anchor (same
    generator as tests/golden/make_golden.py)."""
    base = yolo_maps(seed, B)
    rng = np.random.default_rng(seed + 1000)
    pm = np.zeros((B, 130, 14, 14), np.float32)
    for a in (0, 1):
        pm[:, 65 * a:65 * a + 50] = base[:, 50 * a:50 * a + 50]
        pm[:, 65 * a + 50:65 * a + 65] = rng.uniform(0.0, 1.0, (B, 15, 14, 14))
    return pm


def coco_case(seed, P, H=184, W=216):
    rng = np.random.default_rng(seed)
    pairs = [(1, 2), (1, 5), (2, 3), (3, 4), (5, 6), (6, 7), (1, 8), (8, 9), (9, 10), (1, 11), (11, 12), (12, 13), (1, 0),
             (0, 14), (14, 16), (0, 15), (15, 17), (2, 16), (5, 17)]
    net = [(12, 13), (20, 21), (14, 15), (16, 17), (22, 23), (24, 25), (0, 1), (2, 3), (4, 5), (6, 7), (8, 9), (10, 11),
           (28, 29), (30, 31), (34, 35), (32, 33), (36, 37), (18, 19), (26, 27)]
    tmpl = np.array([[.5, .08], [.5, .2], [.38, .22], [.33, .38], [.3, .52], [.62, .22], [.67, .38], [.7, .52], [.44, .55],
                     [.43, .75], [.42, .95], [.56, .55], [.57, .75], [.58, .95], [.47, .05], [.53, .05], [.43, .07], [.57, .07]])
    paf = rng.normal(0, 0.01, (H, W, 38)).astype(np.float32)
    peaks = []
    yy, xx = np.mgrid[0:H, 0:W]
    for _ in range(P):
        hgt = rng.uniform(90, 170)
        wid = hgt * 0.55
        x0, y0 = rng.uniform(2, W - wid - 2), rng.uniform(2, H - hgt - 2)
        pts = np.rint(tmpl * [wid, hgt] + [x0, y0]).astype(int)
        drop = rng.random(18) < 0.1
        for j in range(18):
            if not drop[j]:
                peaks.append((pts[j, 0], pts[j, 1], rng.uniform(0.3, 1.0), 0, j))
        for l, (a, b) in enumerate(pairs):
            d = pts[b] - pts[a]
            n = np.hypot(*d)
            if n < 1:
                continue
            u = d / n
            rx, ry = xx - pts[a, 0], yy - pts[a, 1]
            m = ((rx * u[0] + ry * u[1]) >= -1) & ((rx * u[0] + ry * u[1]) <= n + 1) & (np.abs(rx * u[1] - ry * u[0]) <= 3)
            paf[:, :, net[l][0]][m] = u[0]
            paf[:, :, net[l][1]][m] = u[1]
    peaks.sort(key=lambda r: r[4])
    pk = np.array(peaks, dtype=np.float32).reshape(1, -1, 5) if peaks else np.zeros((1, 0, 5), np.float32)
    return pk, np.zeros((H, W, 19), np.float32), paf


def humans_to_array(humans):
    arr = -np.ones((len(humans), 1 + 18 * 4), dtype=np.float64)
    for i, h in enumerate(humans):
        arr[i, 0] = h['score']
        for p, (cid, x, y, s) in h['parts'].items():
            arr[i, 1 + 4 * p:5 + 4 * p] = (cid, x, y, s)
    return arr


PAFPROCESS_CASES = [(40, 1), (41, 2), (42, 4), (43, 0), (44, 6)]
CPP_CASES = [(60, 1), (61, 2), (62, 3), (63, 0), (64, 5)]      # tests/golden/make_golden.py::CPP_CASES (paf_to_pose_cpp on synth.coco_maps)


# ---- training goldens (must mirror tests/golden/make_golden.py::train_case_inputs / sample_indices) ----
def train_case_inputs(seed=21, B=3, H=96, W=128):
    rng = np.random.default_rng(seed)
    h, w = H // 8, W // 8
    img = rng.normal(0, 1, (B, 1, H, W)).astype(np.float32)
    heat = rng.uniform(0, 1, (B, 16, h, w)).astype(np.float32)
    paf = rng.uniform(-1, 1, (B, 28, h, w)).astype(np.float32)
    z = rng.uniform(-1.5, 1.5, (B, 15, h, w)).astype(np.float32)
    fg = (rng.uniform(0, 1, (B, 15, h, w)) < 0.3).astype(np.float32)
    return img, heat, paf, z, fg


def sample_indices(name, numel, n=48):
    import zlib
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    return np.sort(rng.choice(numel, size=min(n, numel), replace=False))


# ---- engines against the CPU oracle, end to end (tests/test_gpu_precision.py, docs/lab-archive/oracle_fidelity.py) ----
def _oracle_signature(ref):
    """(peak count, person -> peak-id table) of one oracle frame: what 'bit-exact person assignment' compares."""
    assoc = np.asarray(ref["assoc"]).reshape(-1, 17)
    return len(ref["joint_list"]), assoc[:, :15].astype(np.int32)


def oracle_records(depth, state_dict, perturb=4, eps=1e-5, seed=0):
    """The reference's CPU path (oracle: preproc + torch-CPU fp32 forward + NumPy parse) on `depth` [N, 640, 480] f16.
    Returns one dict per frame: the oracle's records plus `fragile` -- True when the ORACLE's own person assignment changes
    under one of `perturb` random relative perturbations of size `eps` of its fp32 maps (1e-5 = the difference between two
    fp32 summation orders of these convolutions): a frame that holds a decision (a cell against the 0.1 threshold, two
    neighbouring cells against each other, a limb score against 0.05) inside float32's own noise."""
    from oracle import nets as onets, parse_paf as oparse, preproc as opre
    sd = {k: v.detach().cpu() for k, v in state_dict.items()}
    out = []
    rng = np.random.default_rng(seed)
    for s in range(0, len(depth), 32):
        x = torch.from_numpy(opre.preprocess_batch(depth[s:s + 32]))
        paf, heat, z = (a.numpy().transpose(0, 2, 3, 1) for a in onets.rtpose_light3d_forward(x, sd))
        for b in range(len(x)):
            ref = oparse.frame_to_records(heat[b].copy(), paf[b].copy(), z[b].copy())
            n0, a0 = _oracle_signature(ref)
            fragile = False
            for _ in range(perturb):
                hp = (heat[b] * (1 + eps * rng.standard_normal(heat[b].shape))).astype(np.float32)
                pp = (paf[b] * (1 + eps * rng.standard_normal(paf[b].shape)) + eps * rng.standard_normal(paf[b].shape)).astype(np.float32)
                alt = oparse.frame_to_records(hp, pp, z[b].copy())
                n1, a1 = _oracle_signature(alt)
                if n1 != n0 or a1.shape != a0.shape or not np.array_equal(a1, a0):
                    fragile = True
                    break
            ref["fragile"] = fragile
            out.append(ref)
    return out


def vs_oracle(recs, refs):
    """pn_pose_frame records (numpy) of an engine against oracle_records(): frames with identical assignment, the largest 3D /
    confidence difference on those, the differing frames and whether each of them is a fragile one."""
    same, differing, d3, dc = 0, [], [0.0], [0.0]
    for i, (r, ref) in enumerate(zip(recs, refs)):
        n0, a0 = _oracle_signature(ref)
        n = int(r["n_persons"])
        if int(r["status"]) or n != a0.shape[0] or int(r["n_peaks"]) != n0 or not np.array_equal(r["person_joint"][:n], a0):
            differing.append(i)
            continue
        same += 1
        if n:
            vis = r["person_joint"][:n] >= 0
            d3.append(float(np.abs(r["joints_3d"][:n] - np.array(ref["humans_3d"]))[vis].max()) if vis.any() else 0.0)
            dc.append(float(np.abs(r["part_conf"][:n] - np.array(ref["conf"]))[vis].max()) if vis.any() else 0.0)
    return {"frames": len(refs), "same_assignment": same, "differing": differing, "fragile": sum(bool(f["fragile"]) for f in refs),
            "differing_all_fragile": all(refs[i]["fragile"] for i in differing), "d3_m_max": max(d3), "conf_max": max(dc)}
