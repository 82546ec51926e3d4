#!/usr/bin/env python
"""Generates the golden vectors under tests/golden/ by running the REFERENCE ITSELF
(/root/reference, imported read-only in the build container) on seeded inputs.

    python tests/golden/make_golden.py [steps]          # rewrites tests/golden/*.npz / *.json
    python tests/golden/make_golden.py --check [steps]  # regenerates into a scratch dir and compares (tests/test_golden_recipe.py)

Only data leaves this script: inputs (or the seeds that regenerate them through
popnet_amd.synth) and the reference's outputs.  No reference source is copied.

Import shims (written to a temp dir, never shipped): the reference imports packages that are
absent here -- thop, yacs, torchvision, flow_vis, cv2 -- and uses numpy aliases removed in
numpy >= 1.24.  The cv2 shim routes cv2.resize to oracle/cv2_resize.py, so everything the
reference computes AROUND OpenCV is pinned by these vectors while OpenCV's own arithmetic stays
"parity unpinned" (see oracle/__init__.py).
"""
import json
import os
import runpy
import sys
import tempfile
import textwrap

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = HERE          # --check redirects the outputs to a scratch directory and compares them with the committed files
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
TPM = os.path.join(REF, "third_party_methods")
sys.path.insert(0, ROOT)

SHIMS = {
    "thop/__init__.py": """
        def profile(*a, **k): raise NotImplementedError
        def clever_format(*a, **k): raise NotImplementedError
    """,
    "yacs/__init__.py": "",
    "yacs/config.py": """
        class CfgNode(dict):
            def __init__(self, init_dict=None, key_list=None, new_allowed=False):
                super().__init__()
                for k, v in (init_dict or {}).items(): self[k] = v
            def __getattr__(self, k):
                try: return self[k]
                except KeyError: raise AttributeError(k)
            def __setattr__(self, k, v): self[k] = v
            def defrost(self): pass
            def freeze(self): pass
    """,
    "flow_vis/__init__.py": "",
    "cv2/__init__.py": """
        from oracle.cv2_resize import resize, INTER_NEAREST, INTER_LINEAR, INTER_CUBIC
        COLOR_GRAY2BGR = 8
    """,
    "torchvision/__init__.py": """
        import sys, types
        import numpy as np, torch
        class _Dummy:
            def __init__(self, *a, **k): pass
            def __call__(self, *a, **k): return _Dummy()
            def __getattr__(self, k): return _Dummy()
        class Compose:
            def __init__(self, ts): self.ts = ts
            def __call__(self, x):
                for t in self.ts: x = t(x)
                return x
        class ToTensor:
            def __call__(self, x):
                x = np.asarray(x)
                x = x[None] if x.ndim == 2 else x.transpose(2, 0, 1)
                return torch.from_numpy(np.ascontiguousarray(x))
        class Normalize:
            def __init__(self, mean, std): self.mean, self.std = mean, std
            def __call__(self, t):
                m = torch.as_tensor(self.mean, dtype=t.dtype).view(-1, 1, 1)
                s = torch.as_tensor(self.std, dtype=t.dtype).view(-1, 1, 1)
                return t.sub(m).div(s)
        transforms = types.ModuleType('torchvision.transforms')
        transforms.Compose, transforms.ToTensor, transforms.Normalize = Compose, ToTensor, Normalize
        def _ga(k):
            if k.startswith('__'): raise AttributeError(k)      # inspect.getmodule() probes __file__ on every module
            return _Dummy
        transforms.__getattr__ = _ga
        sys.modules['torchvision.transforms'] = transforms
        for sub in ('datasets', 'models', 'utils'):
            mm = types.ModuleType('torchvision.' + sub); mm.__getattr__ = _ga
            sys.modules['torchvision.' + sub] = mm; globals()[sub] = mm
    """,
}


def install_shims():
    d = tempfile.mkdtemp(prefix="popnet_shims_")
    for rel, src in SHIMS.items():
        p = os.path.join(d, rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, "w") as f:
            f.write(textwrap.dedent(src))
    sys.path.insert(0, d)
    sys.path.insert(1, TPM)
    np.int = int          # removed aliases the reference still uses (common.py:16,27,29)
    np.float = float
    import torch
    torch.Tensor.cuda = lambda self, *a, **k: self          # the scripts call .cuda() unconditionally
    torch.nn.Module.cuda = lambda self, *a, **k: self
    return d


def np_sd(arrays):
    import torch
    return {k: torch.from_numpy(v.copy()) for k, v in arrays.items()}


# ---- F5: state_dict layouts ------------------------------------------------------------------
def golden_state_dicts():
    from lib.network.rtpose_light3d import rtpose_light3d
    from lib.network.yolo_posenet import YoloPoseNet
    out = {}
    for name, m in (("rtpose_light3d", rtpose_light3d(15, 14, 2, input_dim=1)), ("yolo_posenet", YoloPoseNet(15, input_dim=1))):
        out[name] = [[k, list(v.shape)] for k, v in m.state_dict().items()]
    json.dump(out, open(os.path.join(OUT, "state_dict_keys.json"), "w"))
    print("F5 state_dict keys:", {k: len(v) for k, v in out.items()})


# ---- F1: network forward ---------------------------------------------------------------------
def reference_preprocess(frame, depth_max):
    """The reference's own transform chain on one frame (datasets_kdh3d_rtpose_mpreal.py:225-246 CR)."""
    import torch
    from lib.datasets import data_augmentation_2d3d as aug
    import torchvision
    pre = aug.Compose([aug.Cvt2ndarray(), aug.Resize(224)])
    image = np.asarray(frame).astype(float)
    image, _ = pre((image, []))
    image[image < 0] = 0
    image[image > depth_max] = depth_max
    tf = torchvision.transforms.Compose([torchvision.transforms.ToTensor(), torchvision.transforms.Normalize(mean=[3], std=[2])])
    return tf(image).numpy()


def golden_forward():
    import torch
    from popnet_amd import synth
    from lib.network.rtpose_light3d import rtpose_light3d
    from lib.network.yolo_posenet import YoloPoseNet
    sample = np.load(os.path.join(TPM, "00_02254.npy"))            # 240x320 f16 ITOP frame shipped with the reference
    frames = [sample, synth.synth_depth(1, 640, 480, seed=3)[0]]
    x = np.stack([reference_preprocess(f, 6) for f in frames]).astype(np.float32)
    torch.manual_seed(0)
    out = {"sample_frame": sample, "x": x}
    for name, model, seed in (("rt", rtpose_light3d(15, 14, 2, input_dim=1), 0), ("yolo", YoloPoseNet(15, input_dim=1), 1)):
        model.eval()
        model.load_state_dict(np_sd(synth.fill_state_dict(model.state_dict(), seed=seed)))
        with torch.no_grad():
            if name == "rt":
                (paf, heat, z), saved = model(torch.from_numpy(x))
                feat = model.model0(torch.from_numpy(x))
                out.update(rt_paf=paf.numpy(), rt_heat=heat.numpy(), rt_z=z.numpy(),
                           rt_paf1=saved[0].numpy()[:, :, ::4, ::4], rt_heat1=saved[1].numpy()[:, :, ::4, ::4],
                           rt_z1=saved[2].numpy()[:, :, ::4, ::4], rt_feat=feat.numpy()[:, ::8, ::2, ::2])
            else:
                y = model(torch.from_numpy(x))
                feat = model.model0(torch.from_numpy(x))
                out.update(yolo_out=y.numpy(), yolo_feat=feat.numpy()[:, ::8, ::2, ::2])
    np.savez_compressed(os.path.join(OUT, "forward.npz"), **out)
    print("F1 forward:", {k: v.shape for k, v in out.items()})


# ---- F2: pose parsing ------------------------------------------------------------------------
PARSE_CASES = [(1, 0), (2, 1), (3, 2), (4, 3), (5, 4), (6, 6), (7, 8), (8, 3), (9, 5)]   # (seed, persons)


def special_parse_maps():
    """Hand-built edge cases: peaks on the borders, an equal-valued two-cell plateau (both cells are
    peaks in the reference), and two persons whose limbs cross."""
    from popnet_amd import synth
    cases = {}
    heat, paf, z = synth.planted_maps(21, 2)
    heat = heat.copy()
    heat[:, :, 3] = 0.0
    heat[0, 0, 3] = 0.8; heat[27, 27, 3] = 0.7; heat[0, 13, 3] = 0.6; heat[14, 27, 3] = 0.65     # border / corner peaks
    heat[10, 10, 5] = heat[10, 11, 5] = 0.9                                                       # plateau
    cases["border_plateau"] = (heat, paf, z)
    h2, p2, z2 = synth.planted_maps(22, 2, drop_prob=0.25)
    cases["missing_joints"] = (h2, p2, z2)
    h3, p3, z3 = synth.planted_maps(23, 7, noise=0.03)
    cases["crowded_noisy"] = (h3, p3, z3)
    return cases


def reference_parse(heat, paf, z, cfg):
    """paf_to_pose + paf_to_human_list + the read-out glue, calling the reference functions."""
    from lib.utils.paf_to_pose import paf_to_pose
    from lib.utils.common import paf_to_human_list, retrieve_depth_heat_weighted
    heat = heat.copy()
    posedepth = z.copy()
    posedepth *= 2
    posedepth += 3
    joint_list, assoc = paf_to_pose(heat, paf, cfg)
    humans_2d, vis, conf = paf_to_human_list(joint_list, assoc)
    depths = []
    for i, human in enumerate(humans_2d):
        hd = np.ones(15) * -1
        for j, joint in enumerate(human):
            if vis[i][j] > 0.5:
                hd[j] = retrieve_depth_heat_weighted([int(joint[0] / 8), int(joint[1] / 8)], posedepth[:, :, j], heat[:, :, j], radius=1)
        depths.append(hd)
    return joint_list, assoc, np.array(depths), np.array(conf, dtype=np.float64)


def golden_parse():
    from popnet_amd import synth
    from popnet_amd.config import default_cfg
    cfg = default_cfg()
    out = {}
    cases = {"planted_s%d_p%d" % (s, p): synth.planted_maps(s, p) for s, p in PARSE_CASES}
    special = special_parse_maps()
    for name, (h, p, z) in special.items():
        out["in_%s_heat" % name], out["in_%s_paf" % name], out["in_%s_z" % name] = h, p, z
    cases.update(special)
    for name, (heat, paf, z) in cases.items():
        jl, assoc, depths, conf = reference_parse(heat, paf, z, cfg)
        out["%s_joint_list" % name] = np.asarray(jl, dtype=np.float64).reshape(-1, 5)
        out["%s_assoc" % name] = np.asarray(assoc, dtype=np.float64).reshape(-1, 17)
        out["%s_depths" % name] = np.asarray(depths, dtype=np.float64).reshape(-1, 15)
        out["%s_conf" % name] = np.asarray(conf, dtype=np.float64).reshape(-1, 15)
        out["%s_insum" % name] = np.array([float(heat.astype(np.float64).sum()), float(paf.astype(np.float64).sum())])
        print("F2 %-20s peaks %3d persons %2d" % (name, len(out["%s_joint_list" % name]), len(out["%s_assoc" % name])))
    out["case_names"] = np.array(sorted(cases))
    np.savez_compressed(os.path.join(OUT, "parse_paf.npz"), **out)


# ---- F3: yolo decode -------------------------------------------------------------------------
def yolo_maps(seed, B=2, clusters=True):
    rng = np.random.default_rng(seed)
    pm = rng.uniform(-0.9, 0.9, (B, 100, 14, 14)).astype(np.float32)
    for a in (0, 1):
        pm[:, 50 * a + 2:50 * a + 4] = rng.uniform(0.6, 1.9, (B, 2, 14, 14))
        pm[:, 50 * a + 4] = rng.uniform(0.0, 0.45, (B, 14, 14))
        pm[:, 50 * a + 5:50 * a + 50] = rng.uniform(-1.9, 1.9, (B, 45, 14, 14))
    if clusters:       # groups of >= 4 mutually overlapping boxes: exercises the keep-loop quirk
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
    """[B, 2 x (5 + 4 x 15), 14, 14]: yolo_maps() with 15 predicted-visibility channels appended to every anchor."""
    base = yolo_maps(seed, B)
    rng = np.random.default_rng(seed + 1000)
    pm = np.zeros((B, 130, 14, 14), np.float32)
    for a in (0, 1):
        pm[:, 65 * a:65 * a + 50] = base[:, 50 * a:50 * a + 50]
        pm[:, 65 * a + 50:65 * a + 65] = rng.uniform(0.0, 1.0, (B, 15, 14, 14))
    return pm


def golden_yolo():
    import torch
    from lib.utils.prior_pose_align import parse_prior_pose
    out = {}
    for seed in (31, 32, 33):
        pm = yolo_maps(seed, clusters=seed != 33)
        b, h, v = parse_prior_pose(torch.from_numpy(pm.copy()), [(6., 3.), (12., 6.)], 15, 224, 224, 3, 2, 0.5, 0.5)
        for i in range(pm.shape[0]):
            out["s%d_%d_bbox" % (seed, i)] = np.array(b[i], dtype=np.float32).reshape(-1, 5)
            out["s%d_%d_human" % (seed, i)] = np.array(h[i], dtype=np.float32).reshape(-1, 15, 3)
            out["s%d_%d_vis" % (seed, i)] = np.array(v[i], dtype=bool).reshape(-1, 15)
            print("F3 seed %d img %d: %d boxes" % (seed, i, len(b[i])))
    # pred_vis=True (:62,120,153-157): 5 + 4 J channels per anchor, visibility = in-bounds test x predicted visibility
    pm = yolo_maps_predvis(34)
    b, h, v = parse_prior_pose(torch.from_numpy(pm.copy()), [(6., 3.), (12., 6.)], 15, 224, 224, 3, 2, 0.5, 0.5, pred_vis=True)
    for i in range(pm.shape[0]):
        out["pv_%d_bbox" % i] = np.array(b[i], dtype=np.float32).reshape(-1, 5)
        out["pv_%d_human" % i] = np.array(h[i], dtype=np.float32).reshape(-1, 15, 3)
        out["pv_%d_vis" % i] = np.array(v[i], dtype=np.float32).reshape(-1, 15)
        print("F3 pred_vis img %d: %d boxes" % (i, len(b[i])))
    np.savez_compressed(os.path.join(OUT, "parse_yolo.npz"), **out)


# ---- F4: process_paf (compiled reference C++) ------------------------------------------------
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
    """[n, 1 + 18*4]: score, then per part (cid, x, y, score) or -1s."""
    arr = -np.ones((len(humans), 1 + 18 * 4), dtype=np.float64)
    for i, h in enumerate(humans):
        arr[i, 0] = h['score']
        for p, (cid, x, y, s) in h['parts'].items():
            arr[i, 1 + 4 * p:5 + 4 * p] = (cid, x, y, s)
    return arr


def golden_pafprocess():
    from oracle import pafprocess as pp
    pp.build()
    ref = pp.reference()
    assert ref is not None, "oracle/_ref/libpafprocess_ref.so missing (make -C oracle)"
    out = {}
    for seed, P in ((40, 1), (41, 2), (42, 4), (43, 0), (44, 6)):
        pk, heat, paf = coco_case(seed, P)
        humans = ref.run(pk, heat, paf)
        out["s%d_p%d" % (seed, P)] = humans_to_array(humans)
        print("F4 process_paf seed %d P=%d -> %d humans" % (seed, P, len(humans)))
    np.savez_compressed(os.path.join(OUT, "pafprocess.npz"), **out)


# ---- F4b: paf_to_pose_cpp, the reference's Python caller of process_paf -------------------------
CPP_CASES = ((60, 1), (61, 2), (62, 3), (63, 0), (64, 5))


def humans_objects_to_array(humans):
    """[n, 1 + 18*3]: human.score, then per COCO part (x, y, score) of human.body_parts or -1s -- everything coco_eval.py:270-290 reads."""
    arr = -np.ones((len(humans), 1 + 18 * 3), dtype=np.float64)
    for i, h in enumerate(humans):
        arr[i, 0] = h.score
        for p, bp in h.body_parts.items():
            assert bp.part_idx == p and bp.uidx.endswith("-%d" % p)
            arr[i, 1 + 3 * p:4 + 3 * p] = (bp.x, bp.y, bp.score)
    return arr


def golden_paf_to_pose_cpp():
    """The reference function tpm/lib/utils/paf_to_pose.py:381-415 itself, on seeded COCO-18 maps, with the reference's own pafprocess.cpp
    (compiled as-is into oracle/_ref) behind the module name the function expects -- the reference ships that import commented out
    (paf_to_pose.py:7), so the name is bound here the way coco_eval.py's environment would bind it."""
    import types
    import ctypes as C
    from types import SimpleNamespace
    from oracle import pafprocess as pp
    from popnet_amd import synth
    import lib.utils.paf_to_pose as ref_mod
    pp.build()
    ref = pp.reference()
    assert ref is not None, "oracle/_ref/libpafprocess_ref.so missing (make -C oracle)"
    shim = types.ModuleType("pafprocess")

    def process_paf(peaks, heat, paf):
        peaks, heat, paf = (np.ascontiguousarray(a, dtype=np.float32) for a in (peaks, heat, paf))
        fp = C.POINTER(C.c_float)
        return ref._pp(*peaks.shape, peaks.ctypes.data_as(fp), *heat.shape, heat.ctypes.data_as(fp), *paf.shape, paf.ctypes.data_as(fp))
    shim.process_paf = process_paf
    for k in ("get_num_humans", "get_part_cid", "get_score", "get_part_x", "get_part_y", "get_part_score"):
        setattr(shim, k, ref._g[k])
    ref_mod.pafprocess = shim
    cfg = SimpleNamespace(MODEL=SimpleNamespace(DOWNSAMPLE=8, NUM_KEYPOINTS=18), TEST=SimpleNamespace(THRESH_HEATMAP=0.1))
    out = {}
    for seed, P in CPP_CASES:
        heat, paf = synth.coco_maps(seed, P)
        humans = ref_mod.paf_to_pose_cpp(heat.copy(), paf.copy(), cfg)
        out["s%d_p%d" % (seed, P)] = humans_objects_to_array(humans)
        nms = ref_mod.NMS(heat.copy(), upsampFactor=8, config=cfg)
        out["s%d_p%d_nms" % (seed, P)] = np.array([tuple(pk) + (j,) for j, pks in enumerate(nms) for pk in pks], dtype=np.float64).reshape(-1, 5)
        out["s%d_p%d_insum" % (seed, P)] = np.array([float(heat.astype(np.float64).sum()), float(paf.astype(np.float64).sum())])
        print("F4b paf_to_pose_cpp seed %d P=%d -> %d peaks, %d humans" % (seed, P, len(out["s%d_p%d_nms" % (seed, P)]), len(humans)))
    np.savez_compressed(os.path.join(OUT, "paf_to_pose_cpp.npz"), **out)


# ---- F6: the reference evaluation SCRIPT, end to end, on a fake two-frame dataset ---------------
def calibrated_heat_bias(model, x, frac=0.004):
    """Per-channel stage-2 heat bias shift so ~frac of the cells pass THRESH_HEATMAP (CPU twin of
    popnet_amd.pipeline.calibrate_heads, evaluated with the reference model)."""
    import torch
    with torch.no_grad():
        (_, heat, _), _ = model(torch.from_numpy(x))
    s = heat[:, :15].double().clamp(1e-12, 1 - 1e-12)
    logit = torch.log(s / (1 - s)).permute(1, 0, 2, 3).reshape(15, -1)
    q = torch.quantile(logit, 1.0 - frac, dim=1)
    return (float(np.log(0.1 / 0.9)) - q).float().numpy()


def golden_script():
    import torch
    from popnet_amd import synth
    from lib.network.rtpose_light3d import rtpose_light3d
    work = tempfile.mkdtemp(prefix="popnet_fake_ds_")
    img_dir = os.path.join(work, "depth_maps")
    os.makedirs(img_dir)
    frames = synth.synth_depth(2, 640, 480, seed=77)
    labels = {"intrinsics": {"fx": 504.1189880371094, "fy": 504.042724609375, "cx": 231.7421875, "cy": 320.62640380859375}}
    rng = np.random.default_rng(5)
    for i in range(2):
        np.save(os.path.join(img_dir, "f%d.npy" % i), frames[i])
        j2 = rng.uniform(50, 400, (15, 2))
        labels["f%d.npy" % i] = [{"2d_joints": j2.tolist(), "3d_joints": np.c_[j2 / 200, np.full(15, 3.0)].tolist()}]
    ann = os.path.join(work, "labels.json")
    json.dump(labels, open(ann, "w"))
    model = rtpose_light3d(15, 14, 2, input_dim=1).eval()
    arrays = synth.fill_state_dict(model.state_dict(), seed=0)
    model.load_state_dict(np_sd(arrays))
    x = np.stack([reference_preprocess(f, 6) for f in frames]).astype(np.float32)
    shift = calibrated_heat_bias(model, x)
    arrays["model2_2.12.bias"][:15] += shift
    ckpt = os.path.join(work, "ckpt.pth")
    torch.save({"module." + k: torch.from_numpy(v) for k, v in arrays.items()}, ckpt)
    outdir = os.path.join(work, "out")
    argv = sys.argv
    cwd = os.getcwd()
    try:
        os.chdir(os.path.join(TPM, "evaluate"))
        sys.argv = ["eval", "--annotations", ann, "--image-dir", img_dir, "--w-org", "480", "--h-org", "640",
                    "--batch-size", "2", "--weight", ckpt, "--output-dir", outdir]
        import matplotlib
        matplotlib.use("Agg")
        runpy.run_path(os.path.join(TPM, "evaluate", "evaluation_rtpose_light3d_kdh3d_mpreal_ablation.py"), run_name="__main__")
    finally:
        sys.argv = argv
        os.chdir(cwd)
    data = json.load(open(os.path.join(outdir, "eval_data.json")))
    keep = {k: data[k] for k in ("human_pred_set_2d", "human_pred_set_3d", "human_pred_set_visibility", "human_pred_set_part_conf")}
    keep["heat_bias_shift"] = shift.tolist()
    keep["depth_seed"] = 77
    keep["weight_seed"] = 0
    json.dump(keep, open(os.path.join(OUT, "script_eval_data.json"), "w"))
    print("F6 script: persons per frame", [len(f) for f in keep["human_pred_set_2d"]])


# ---- F7: the reference Yolo-Pose+ evaluation SCRIPT, end to end, on a fake two-frame dataset ------
def yolo_conf_shift(model, x, frac=0.012):
    """The last YoloPoseNet conv has no bias, so the confidence channels are calibrated by shifting
    every weight of the two conf filters (channels 4 and 54) by -delta: logit -> logit - delta *
    conv(X, 1).  Returns the delta (grid search) that leaves ~frac of the cells above conf 0.5."""
    import torch
    feats = []
    hk = model.model2_4.register_forward_hook(lambda m, i, o: feats.append(i[0].detach()))
    with torch.no_grad():
        model(torch.from_numpy(x))
    hk.remove()
    X = feats[0]
    w = model.model2_4[0].weight.detach()
    base = torch.nn.functional.conv2d(X, w[[4, 54]], padding=1)
    ones = torch.nn.functional.conv2d(X, torch.ones_like(w[[4, 54]]), padding=1)
    best = None
    for d in np.linspace(0.0, 0.2, 801):
        fr = float(((base - d * ones) > 0).float().mean())
        if best is None or abs(fr - frac) < abs(best[1] - frac):
            best = (float(d), fr)
    return best[0]


def golden_script_yolo():
    import torch
    from popnet_amd import synth
    from lib.network.yolo_posenet import YoloPoseNet
    work = tempfile.mkdtemp(prefix="popnet_fake_ds_yolo_")
    img_dir = os.path.join(work, "depth_maps")
    os.makedirs(img_dir)
    frames = synth.synth_depth(2, 640, 480, seed=78)
    labels = {"intrinsics": {"fx": 504.1189880371094, "fy": 504.042724609375, "cx": 231.7421875, "cy": 320.62640380859375}}
    rng = np.random.default_rng(6)
    for i in range(2):
        np.save(os.path.join(img_dir, "f%d.npy" % i), frames[i])
        j2 = rng.uniform(50, 400, (15, 2))
        labels["f%d.npy" % i] = [{"2d_joints": j2.tolist(), "3d_joints": np.c_[j2 / 200, np.full(15, 3.0)].tolist()}]
    ann = os.path.join(work, "labels.json")
    json.dump(labels, open(ann, "w"))
    model = YoloPoseNet(15, input_dim=1).eval()
    arrays = synth.fill_state_dict(model.state_dict(), seed=1)
    model.load_state_dict(np_sd(arrays))
    x = np.stack([reference_preprocess(f, 6) for f in frames]).astype(np.float32)
    delta = yolo_conf_shift(model, x)
    arrays["model2_4.0.weight"][[4, 54]] -= np.float32(delta)
    ckpt = os.path.join(work, "ckpt.pth")
    torch.save({"module." + k: torch.from_numpy(v) for k, v in arrays.items()}, ckpt)
    outdir = os.path.join(work, "out")
    argv = sys.argv
    cwd = os.getcwd()
    try:
        os.chdir(os.path.join(TPM, "evaluate"))
        sys.argv = ["eval", "--val-annotations", ann, "--val-image-dir", img_dir, "--w-org", "480", "--h-org", "640",
                    "--batch-size", "2", "--weight", ckpt, "--output-dir", outdir]
        import matplotlib
        matplotlib.use("Agg")
        try:
            runpy.run_path(os.path.join(TPM, "evaluate", "evaluation_yolo_posenet_kdh3d_mpreal.py"), run_name="__main__")
        except Exception as e:      # the metric code after the dump may trip on numpy >= 1.24; the dump is what we keep
            print("F7: script raised after/while evaluating:", repr(e))
    finally:
        sys.argv = argv
        os.chdir(cwd)
    data = json.load(open(os.path.join(outdir, "eval_data.json")))
    keep = {k: data[k] for k in ("human_pred_set_2d", "human_pred_set_3d", "human_pred_set_part_conf")}
    keep["conf_weight_shift"] = delta
    keep["depth_seed"] = 78
    keep["weight_seed"] = 1
    json.dump(keep, open(os.path.join(OUT, "script_eval_data_yolo.json"), "w"))
    print("F7 yolo script: persons per frame", [len(f) for f in keep["human_pred_set_2d"]])


# ---- F8: metric known-answers from the reference's util/eval_pck.py and util/eval_mAP.py -----------
def metric_case(seed, n_img=14):
    """Seeded prediction / ground-truth sets in the result-schema shapes: 0-3 GT persons per image, predictions =
    jittered GT (some joints missing = [-1,-1], conf 0), dropped persons, extra false positives, tied confidences,
    images without predictions.  Every image keeps >= 1 GT person (the reference's mAP code needs one)."""
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = 504.1189880371094, 504.042724609375, 231.7421875, 320.62640380859375
    p2, p3, pc, g2, g3 = [], [], [], [], []
    for i in range(n_img):
        ng = int(rng.integers(1, 4))
        G2, G3, P2, P3, PC = [], [], [], [], []
        for g in range(ng):
            base = rng.uniform([60, 80], [420, 560])
            j2 = base + rng.uniform(-60, 60, (15, 2)) * [0.6, 1.6]
            z = rng.uniform(1.5, 4.5) + rng.normal(0, 0.05, 15)
            j3 = np.c_[(j2[:, 0] - cx) / fx * z, (j2[:, 1] - cy) / fy * z, z]
            G2.append(j2.tolist()); G3.append(j3.tolist())
            if rng.random() < 0.8:                       # detected
                noise = rng.choice([1.0, 6.0, 25.0])
                q2 = j2 + rng.normal(0, noise, (15, 2))
                qz = z + rng.normal(0, rng.choice([0.01, 0.06, 0.2]), 15)
                q3 = np.c_[(q2[:, 0] - cx) / fx * qz, (q2[:, 1] - cy) / fy * qz, qz]
                conf = np.round(rng.uniform(0.2, 1.0, 15), 1)          # coarse: many exact ties
                miss = rng.random(15) < 0.12
                q2[miss] = -1; q3[miss] = -1; conf[miss] = 0
                P2.append(q2.tolist()); P3.append(q3.tolist()); PC.append(conf.tolist())
        for _ in range(int(rng.integers(0, 2))):        # false positive somewhere else
            q2 = rng.uniform([0, 0], [480, 640], (15, 2)); qz = rng.uniform(1, 5, 15)
            P2.append(q2.tolist()); P3.append(np.c_[(q2[:, 0] - cx) / fx * qz, (q2[:, 1] - cy) / fy * qz, qz].tolist())
            PC.append(np.round(rng.uniform(0.1, 0.6, 15), 1).tolist())
        if i % 6 == 5:
            P2, P3, PC = [], [], []                      # image without predictions
        order = rng.permutation(len(P2))
        p2.append([P2[k] for k in order]); p3.append([P3[k] for k in order]); pc.append([PC[k] for k in order])
        g2.append(G2); g3.append(G3)
    return p2, p3, pc, g2, g3


def golden_metrics():
    sys.path.insert(0, REF)
    import copy
    import io
    import contextlib
    from util import eval_pck as RP, eval_mAP as RA, util_functions as RU
    names = RU.get_keypoints()
    out = {"cases": []}
    for seed in (101, 102, 103):
        p2, p3, pc, g2, g3 = metric_case(seed)
        with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
            d2, k2 = RP.eval_human_dataset_2d_PCKh(copy.deepcopy(p2), copy.deepcopy(g2), num_joints=15, head_id=0, neck_id=1, iou_th=0.5)
            d3, k3 = RP.eval_human_dataset_3d(copy.deepcopy(p2), copy.deepcopy(g2), copy.deepcopy(p3), copy.deepcopy(g3), num_joints=15, dist_th=0.1, iou_th=0.5)
            a2 = RA.eval_ap_mpii_v2(copy.deepcopy(p2), copy.deepcopy(pc), copy.deepcopy(g2), gt_visibility_set=[], head_id=0, neck_id=1, joint_names=names, thresh=0.5)
            a3 = RA.eval_ap_3D(copy.deepcopy(p3), copy.deepcopy(pc), copy.deepcopy(g3), gt_visibility_set=[], joint_names=names, thresh=0.1)
            md = [np.asarray(x).tolist() for x in RP.match_humans_3d(p2[0], g2[0], p3[0], g3[0], 0.5)]
        out["cases"].append({"seed": seed, "pck2d": [float(v) for v in k2], "err2d": [float(v) for v in d2],
                             "pck3d": [float(v) for v in k3], "err3d": [float(v) for v in d3],
                             "ap2d": np.asarray(a2).tolist(), "ap3d": np.asarray(a3).tolist(), "match3d_img0": md})
        print("F8 seed %d: PCK2D %.3f PCK3D %.3f AP2D %.2f AP3D %.2f" % (seed, np.mean(k2), np.mean(k3), a2[-1], a3[-1]))
    # perfect predictions: AP 100 / PCK 1
    p2, p3, pc, g2, g3 = metric_case(104)
    with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
        a2 = RA.eval_ap_mpii_v2(copy.deepcopy(g2), [], copy.deepcopy(g2), gt_visibility_set=[], head_id=0, neck_id=1, joint_names=names, thresh=0.5)
        d2, k2 = RP.eval_human_dataset_2d_PCKh(copy.deepcopy(g2), copy.deepcopy(g2), num_joints=15, head_id=0, neck_id=1, iou_th=0.5)
    out["perfect"] = {"seed": 104, "ap2d": np.asarray(a2).tolist(), "pck2d": [float(v) for v in k2]}
    json.dump(out, open(os.path.join(OUT, "metrics.json"), "w"))


# ---- F9: end-to-end metrics: the reference's evaluation script + its own metric code on a 12-frame synthetic split ----
def golden_script_metrics(net="rtpose"):
    """12 synthetic frames through the reference evaluation script (same checkpoint recipe as F6), ground truth derived from
    its own predictions (jittered, some persons dropped) so that PCK / mAP are non-trivial, metrics by util/eval_pck.py and
    util/eval_mAP.py.  The GPU test feeds the same frames + labels to scripts/evaluate_mpreal.py and compares the metrics."""
    import contextlib
    import copy
    import io
    import torch
    from popnet_amd import synth
    from lib.network.rtpose_light3d import rtpose_light3d
    from lib.network.yolo_posenet import YoloPoseNet
    N = 12
    work = tempfile.mkdtemp(prefix="popnet_fake_ds_metrics_")
    img_dir = os.path.join(work, "depth_maps")
    os.makedirs(img_dir)
    frames = synth.synth_depth(N, 640, 480, seed=79)
    intr = {"fx": 504.1189880371094, "fy": 504.042724609375, "cx": 231.7421875, "cy": 320.62640380859375}
    labels = {"intrinsics": intr}
    for i in range(N):
        np.save(os.path.join(img_dir, "f%02d.npy" % i), frames[i])
        labels["f%02d.npy" % i] = [{"2d_joints": [[10.0 + j, 20.0] for j in range(15)], "3d_joints": [[0.0, 0.0, 3.0]] * 15}]
    ann = os.path.join(work, "labels.json")
    json.dump(labels, open(ann, "w"))
    x = np.stack([reference_preprocess(f, 6) for f in frames[:4]]).astype(np.float32)
    if net == "rtpose":
        model = rtpose_light3d(15, 14, 2, input_dim=1).eval()
        arrays = synth.fill_state_dict(model.state_dict(), seed=0)
        model.load_state_dict(np_sd(arrays))
        shift = calibrated_heat_bias(model, x)
        arrays["model2_2.12.bias"][:15] += shift
        script, a_ann, a_img = "evaluation_rtpose_light3d_kdh3d_mpreal_ablation.py", "--annotations", "--image-dir"
    else:
        model = YoloPoseNet(15, input_dim=1).eval()
        arrays = synth.fill_state_dict(model.state_dict(), seed=1)
        model.load_state_dict(np_sd(arrays))
        shift = np.float32(yolo_conf_shift(model, x))
        arrays["model2_4.0.weight"][[4, 54]] -= shift
        script, a_ann, a_img = "evaluation_yolo_posenet_kdh3d_mpreal.py", "--val-annotations", "--val-image-dir"
    ckpt = os.path.join(work, "ckpt.pth")
    torch.save({"module." + k: torch.from_numpy(v) for k, v in arrays.items()}, ckpt)
    outdir = os.path.join(work, "out")
    argv, cwd = sys.argv, os.getcwd()
    try:
        os.chdir(os.path.join(TPM, "evaluate"))
        sys.argv = ["eval", a_ann, ann, a_img, img_dir, "--w-org", "480", "--h-org", "640",
                    "--batch-size", "4", "--weight", ckpt, "--output-dir", outdir]
        import matplotlib
        matplotlib.use("Agg")
        with contextlib.redirect_stdout(io.StringIO()):
            try:
                runpy.run_path(os.path.join(TPM, "evaluate", script), run_name="__main__")
            except Exception as e:      # the scripts' own metric tail may trip after eval_data.json has been written
                print("F9 (%s): script raised after the dump: %r" % (net, e), file=sys.stderr)
    finally:
        sys.argv = argv
        os.chdir(cwd)
    data = json.load(open(os.path.join(outdir, "eval_data.json")))
    p2, p3, pc = data["human_pred_set_2d"], data["human_pred_set_3d"], data["human_pred_set_part_conf"]
    rng = np.random.default_rng(17)
    g2, g3 = [], []
    for f in range(N):
        G2, G3 = [], []
        for h2, h3 in zip(p2[f], p3[f]):
            a2, a3 = np.array(h2, dtype=np.float64), np.array(h3, dtype=np.float64)
            vis = ~((a2[:, 0] == -1) & (a2[:, 1] == -1))
            if vis.sum() < 6 or rng.random() < 0.2:
                continue
            a2[~vis] = a2[vis].mean(0); a3[~vis] = a3[vis].mean(0)            # ground truth has every joint
            G2.append((a2 + rng.normal(0, 4.0, a2.shape)).tolist())
            G3.append((a3 + rng.normal(0, 0.04, a3.shape)).tolist())
        if not G2:                                                            # the reference's mAP code needs >= 1 GT person per image
            G2.append(rng.uniform(50, 400, (15, 2)).tolist()); G3.append(rng.uniform(-1, 4, (15, 3)).tolist())
        g2.append(G2); g3.append(G3)
    sys.path.insert(0, REF)
    from util import eval_pck as RP, eval_mAP as RA, util_functions as RU
    names = RU.get_keypoints()
    with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
        d2, k2 = RP.eval_human_dataset_2d_PCKh(copy.deepcopy(p2), copy.deepcopy(g2), num_joints=15, head_id=0, neck_id=1, iou_th=0.5)
        d3, k3 = RP.eval_human_dataset_3d(copy.deepcopy(p2), copy.deepcopy(g2), copy.deepcopy(p3), copy.deepcopy(g3), num_joints=15, dist_th=0.1, iou_th=0.5)
        a2 = RA.eval_ap_mpii_v2(copy.deepcopy(p2), copy.deepcopy(pc), copy.deepcopy(g2), gt_visibility_set=[], head_id=0, neck_id=1, joint_names=names, thresh=0.5)
        a3 = RA.eval_ap_3D(copy.deepcopy(p3), copy.deepcopy(pc), copy.deepcopy(g3), gt_visibility_set=[], joint_names=names, thresh=0.1)
    out = {"depth_seed": 79, "weight_seed": 0 if net == "rtpose" else 1, "n_frames": N, "gt_2d": g2, "gt_3d": g3,
           ("heat_bias_shift" if net == "rtpose" else "conf_weight_shift"): np.asarray(shift).tolist(),
           "persons_per_frame": [len(f) for f in p2],
           "pck2d": [float(v) for v in k2], "err2d": [float(v) for v in d2], "pck3d": [float(v) for v in k3], "err3d": [float(v) for v in d3],
           "ap2d": np.asarray(a2).tolist(), "ap3d": np.asarray(a3).tolist()}
    json.dump(out, open(os.path.join(OUT, "script_metrics.json" if net == "rtpose" else "script_metrics_yolo.json"), "w"))
    print("F9 (%s): persons/frame" % net, out["persons_per_frame"], "GT/frame", [len(f) for f in g2])
    print("F9: PCK2D %.3f PCK3D %.3f AP2D %.2f AP3D %.2f" % (np.nanmean(k2), np.nanmean(k3), a2[-1], a3[-1]))


def golden_targets():
    """F10: training targets.  (a) the reference's get_ground_truth (datasets_kdh3d_rtpose_mpaug.py:318-401 (CR)) on seeded
    annotation sets -- called on a stand-in `self` that carries only the attributes the method reads, so no dataset files
    are needed; (b) the reference dataset's __getitem__ itself (z-buffer compositor + Resize + clamp + depth_resize +
    targets, :223-286 (CR)) on a tiny fake MP-3DHP training tree written to a temp dir, with `random` seeded."""
    import importlib
    import random
    import types
    from popnet_amd import synth
    mod = importlib.import_module("lib.datasets.datasets_kdh3d_rtpose_mpaug")
    K = mod.KDH3D_Keypoints
    stub = types.SimpleNamespace(input_x=224, input_y=224, stride=8, strideZ=8, num_joints=15, z_radius=2,
                                 limb_ids=mod.kp_connections(mod.get_keypoints()), joint_names=mod.get_keypoints())
    stub.remove_illegal_joint = types.MethodType(K.remove_illegal_joint, stub)
    out = {}
    cases = [(0, 11), (1, 12), (3, 13), (5, 14), (2, 15)]
    for ci, (P, seed) in enumerate(cases):
        rng = np.random.default_rng(seed)
        joints, depths = synth.planted_persons(rng, P) if P else (np.zeros((0, 15, 2)), np.zeros(0))
        kp3 = np.concatenate([joints, np.broadcast_to(depths[:, None, None], (P, 15, 1)) + rng.normal(0, 0.05, (P, 15, 1))], 2) if P else np.zeros((0, 15, 3))
        if P >= 2:
            joints[1, 4] = [230.0, 50.0]                      # outside the input: joint and its limbs dropped
            joints[0, 7] = [-3.0, 10.0]
            joints[1, 11] = joints[1, 9]                      # zero-length limb
            joints[0, 13] = [223.9, 223.9]                    # last cell, window clipped
        if P == 5:
            joints[3] = joints[2] + 4.0                       # two persons on top of each other: clamp at 1, paf averaging, nearest z
            kp3[3, :, 2] = kp3[2, :, 2] - 0.4
        if P == 2:                                            # integer / half-integer cell coordinates: round-half-even of the limb box
            joints[0, 8] = [100.0, 100.0]; joints[0, 9] = [116.0, 132.0]; joints[1, 8] = [60.0, 20.0]; joints[1, 1] = [60.0, 68.0]
        anns = [{"2d_joints": joints[p].tolist(), "3d_joints": kp3[p].tolist()} for p in range(P)]
        dr = rng.uniform(-0.5, 6.5, (28, 28))
        if P:
            h, pf, z, fg = K.get_ground_truth(stub, anns, dr)
        else:                                                 # the reference indexes keypoints_2d[:, :, 0]: no annotations = no call; targets of an empty frame
            h = np.zeros((28, 28, 16)); h[:, :, 15] = 1.0
            pf = np.zeros((28, 28, 28)); fg = np.zeros((28, 28, 15))
            z = (np.clip(np.repeat(dr[:, :, None], 15, 2), 0, 6) - 3.0) / 2.0
        out.update({"gt%d_kp2d" % ci: joints, "gt%d_kp3d" % ci: kp3, "gt%d_depth" % ci: dr, "gt%d_heat" % ci: h, "gt%d_paf" % ci: pf,
                    "gt%d_z" % ci: z, "gt%d_fg" % ci: fg})
    out["n_gt"] = np.array(len(cases))

    # (b) the dataset class end to end on a fake tree
    from lib.datasets import data_augmentation_2d3d as aug
    d = tempfile.mkdtemp(prefix="popnet_mpaug_")
    for sub in ("img", "seg", "bg"):
        os.makedirs(os.path.join(d, sub))
    rng = np.random.default_rng(77)
    H, W = 320, 240                                           # (small frames keep the fixture small; the path is size-agnostic)
    ann_files = []
    for ii in range(5):                                       # five annotation sets (aug_mods indexes 0..4), two frames each
        ann = {"intrinsics": dict(mod.intrinsics)}
        for f in range(2):
            name = "s%d_%d.npy" % (ii, f)
            joints, depths = synth.planted_persons(rng, 1, size=224)
            j2 = joints[0] * [W / 224.0, H / 224.0]
            ann[name] = [{"2d_joints": j2.tolist(), "3d_joints": np.concatenate([j2, np.full((15, 1), depths[0])], 1).tolist()}]
            depth = np.clip(rng.normal(depths[0], 0.1, (H, W)), 0.3, 5.9)
            mask = np.zeros((H, W))
            x0, x1 = int(j2[:, 0].min()) - 10, int(j2[:, 0].max()) + 10
            y0, y1 = int(j2[:, 1].min()) - 10, int(j2[:, 1].max()) + 10
            mask[max(y0, 0):y1, max(x0, 0):x1] = 1.0
            np.save(os.path.join(d, "img", name), depth.astype(np.float16))
            np.save(os.path.join(d, "seg", name), mask.astype(np.uint8))
        path = os.path.join(d, "ann%d.json" % ii)
        json.dump(ann, open(path, "w"))
        ann_files.append(path)
    bgs = {}
    for f in range(2):
        name = "bg%d.npy" % f
        np.save(os.path.join(d, "bg", name), np.clip(rng.normal(4.5, 0.3, (H, W)), 0, 6).astype(np.float16))
        bgs[str(f)] = {"file_name": name}
    json.dump(bgs, open(os.path.join(d, "bg.json"), "w"))
    random.seed(5)
    ds = K(os.path.join(d, "img"), ann_files, preprocess=aug.Compose([aug.Cvt2ndarray(), aug.Resize(224)]), w_org=W, h_org=H,
           input_x=224, input_y=224, stride=8, z_radius=2, bg_file=os.path.join(d, "bg.json"), bg_dir=os.path.join(d, "bg"), seg_dir=os.path.join(d, "seg"))
    for idx in range(2):
        # replay the item's random draws to record WHICH sources it composes (same generator state before and after)
        st = random.getstate()
        mod_id = random.randint(0, len(mod.aug_mods) - 1)
        picks = []
        for ii in mod.aug_mods[mod_id]:
            if mod.uniform(0, 1) > 0.8:
                continue
            picks.append(ii)
        if not picks:
            picks.append(random.randint(0, len(ds.ids_list) - 1))
        random.setstate(st)
        image, heat, paf, z, fg, last2d, _ = ds[idx]
        names = [ds.ids_list[ii][idx % len(ds.ids_list[ii])] for ii in picks]
        bg_name = ds.bg_list[idx % ds.num_bg_images]["file_name"]
        kp2d = np.array([np.array(ds.anno_dic_list[ii][nm][0]["2d_joints"]) for ii, nm in zip(picks, names)])
        kp3d = np.array([np.array(ds.anno_dic_list[ii][nm][0]["3d_joints"]) for ii, nm in zip(picks, names)])
        out.update({"it%d_fg_depth" % idx: np.stack([np.load(os.path.join(d, "img", nm)) for nm in names]),
                    "it%d_fg_mask" % idx: np.stack([np.load(os.path.join(d, "seg", nm)) for nm in names]),
                    "it%d_bg" % idx: np.load(os.path.join(d, "bg", bg_name)), "it%d_kp2d_org" % idx: kp2d, "it%d_kp3d" % idx: kp3d,
                    "it%d_image" % idx: image.numpy(), "it%d_heat" % idx: heat.numpy(), "it%d_paf" % idx: paf.numpy(),
                    "it%d_z" % idx: z.numpy(), "it%d_fg" % idx: fg.numpy()})
    out["n_items"] = np.array(2)
    np.savez_compressed(os.path.join(OUT, "targets.npz"), **out)
    print("F10: targets.npz:", len(out), "arrays; persons per composed item", [out["it%d_kp3d" % i].shape[0] for i in range(2)])


def train_case_inputs(seed=21, B=3, H=96, W=128):
    """Seeded batch + targets of the training goldens (also imported by the tests)."""
    rng = np.random.default_rng(seed)
    h, w = H // 8, W // 8
    img = rng.normal(0, 1, (B, 1, H, W)).astype(np.float32)
    heat = rng.uniform(0, 1, (B, 16, h, w)).astype(np.float32)
    paf = rng.uniform(-1, 1, (B, 28, h, w)).astype(np.float32)
    z = rng.uniform(-1.5, 1.5, (B, 15, h, w)).astype(np.float32)
    fg = (rng.uniform(0, 1, (B, 15, h, w)) < 0.3).astype(np.float32)
    return img, heat, paf, z, fg


def sample_indices(name, numel, n=48):
    """Which entries of a tensor the fixture keeps (seeded by the parameter name)."""
    import zlib
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    return np.sort(rng.choice(numel, size=min(n, numel), replace=False))


def golden_train():
    """F11: the reference's OWN training step -- rtpose_light3d(...).train() forward, rtpose_light3d_loss_fgweight
    (lib/network/losses.py:65-106), total_loss.backward(), torch.optim.SGD(lr=1, momentum=0.9, nesterov=True).step()
    (train_rtpose_light3d_kdh3d_mpaug.py:160-180,313-316 (CR)) -- two consecutive steps on a seeded batch.  Kept per
    parameter: gradient L2 norm, sum and 48 sampled entries; the same for the parameters after each step; BN running
    statistics in full; the loss terms and the logged extrema."""
    import torch
    from popnet_amd import synth
    from lib.network.rtpose_light3d import rtpose_light3d
    from lib.network.losses import rtpose_light3d_loss_fgweight
    torch.manual_seed(0)
    model = rtpose_light3d(15, 14, 2, input_dim=1)
    model.load_state_dict(np_sd(synth.fill_state_dict(model.state_dict(), seed=0)))
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=1.0, momentum=0.9, weight_decay=0.0, nesterov=True)
    names = ["l1_paf", "l1_heat", "l1_z", "l2_paf", "l2_heat", "l2_z"]
    img, heat, paf, z, fg = [torch.from_numpy(a) for a in train_case_inputs()]
    out = {}
    for step in range(2):
        _, saved = model(img)
        total, log = rtpose_light3d_loss_fgweight(saved, heat, paf, z, fg, 2, names)
        opt.zero_grad()
        total.backward()
        out["s%d_loss" % step] = np.float64(total.item())
        out["s%d_terms" % step] = np.array([log[n] for n in names])
        out["s%d_extrema" % step] = np.array([log[k] for k in ("max_ht", "min_ht", "max_paf", "min_paf", "max_z", "min_z")])
        for name, p in model.named_parameters():
            if p.grad is None:          # model0.layer3.* does not exist in rtpose; every parameter takes part
                raise RuntimeError(name)
            g = p.grad.detach().numpy().ravel()
            idx = sample_indices(name, g.size)
            out["s%d_g_norm/%s" % (step, name)] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
            out["s%d_g_sum/%s" % (step, name)] = np.float64(g.astype(np.float64).sum())
            out["s%d_g_samp/%s" % (step, name)] = g[idx].copy()
        opt.step()
        for name, p in model.named_parameters():
            v = p.detach().numpy().ravel()
            out["s%d_p_samp/%s" % (step, name)] = v[sample_indices(name, v.size)].copy()
            out["s%d_p_norm/%s" % (step, name)] = np.float64(np.sqrt((v.astype(np.float64) ** 2).sum()))
        for name, b in model.named_buffers():
            if name.endswith("running_mean") or name.endswith("running_var"):
                out["s%d_stat/%s" % (step, name)] = b.detach().numpy().copy()
    out["saved_sizes"] = np.array([list(s.shape) for s in saved])
    np.savez_compressed(os.path.join(OUT, "train_step.npz"), **out)
    print("train_step.npz: loss", out["s0_loss"], "->", out["s1_loss"])


def _same(a, b):
    """Exact comparison of two golden payloads (NaN == NaN); returns a list of differing paths."""
    bad = []

    def walk(x, y, path):
        if isinstance(x, dict) and isinstance(y, dict):
            for k in sorted(set(x) | set(y)):
                if k not in x or k not in y:
                    bad.append(path + "/" + str(k) + " (missing)")
                else:
                    walk(x[k], y[k], path + "/" + str(k))
        elif isinstance(x, (list, tuple)) and isinstance(y, (list, tuple)):
            if len(x) != len(y):
                bad.append(path + " (length)")
            else:
                for i, (u, v) in enumerate(zip(x, y)):
                    walk(u, v, "%s[%d]" % (path, i))
        elif isinstance(x, float) and isinstance(y, float):
            if not (x == y or (x != x and y != y)):
                bad.append(path)
        elif x != y:
            bad.append(path)
    walk(a, b, "")
    return bad


def check_outputs(scratch):
    """Compares every file the selected steps wrote into `scratch` with the committed one of the same name."""
    failures = []
    for name in sorted(os.listdir(scratch)):
        new, old = os.path.join(scratch, name), os.path.join(HERE, name)
        if not os.path.exists(old):
            failures.append("%s: not committed" % name)
        elif name.endswith(".npz"):
            a, b = np.load(new, allow_pickle=False), np.load(old, allow_pickle=False)
            if sorted(a.files) != sorted(b.files):
                failures.append("%s: array names differ" % name)
                continue
            for k in a.files:
                x, y = a[k], b[k]
                if x.shape != y.shape or x.dtype != y.dtype or not np.array_equal(x, y, equal_nan=x.dtype.kind in "fc"):
                    failures.append("%s[%s]" % (name, k))
        else:
            bad = _same(json.load(open(new)), json.load(open(old)))
            failures += ["%s%s" % (name, b) for b in bad[:5]]
        print("check %-28s %s" % (name, "differs" if failures and failures[-1].startswith(name) else "identical"))
    return failures


if __name__ == "__main__":
    assert os.path.isdir(REF), "the reference tree is needed to (re)generate golden vectors"
    import torch.optim                 # noqa: F401  -- before the torchvision shim is importable: these pull in
    import torch.distributed.tensor    # noqa: F401     inspect.getmodule(), which walks every module in sys.modules
    install_shims()
    import popnet_amd  # noqa: F401
    args = sys.argv[1:]
    check = "--check" in args
    args = [a for a in args if a != "--check"]
    if check:
        OUT = tempfile.mkdtemp(prefix="popnet_golden_check_")
    which = args or ["keys", "forward", "parse", "yolo", "pafprocess", "cpp", "script", "script_yolo", "metrics", "script_metrics", "script_metrics_yolo", "targets", "train"]
    fns = {"keys": golden_state_dicts, "forward": golden_forward, "parse": golden_parse, "yolo": golden_yolo,
           "pafprocess": golden_pafprocess, "cpp": golden_paf_to_pose_cpp, "script": golden_script,
           "script_yolo": golden_script_yolo, "metrics": golden_metrics, "script_metrics": golden_script_metrics,
           "script_metrics_yolo": lambda: golden_script_metrics("yolo"), "targets": golden_targets, "train": golden_train}
    for w in which:
        fns[w]()
    if check:
        fails = check_outputs(OUT)
        if fails:
            print("GOLDEN CHECK FAILED:\n  " + "\n  ".join(fails))
            sys.exit(1)
        print("golden check ok: %d files regenerate identically" % len(os.listdir(OUT)))
