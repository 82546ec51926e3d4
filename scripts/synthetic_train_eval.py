#!/usr/bin/env python
"""Train -> infer -> evaluate, every stage on the GPU path of this repository, on a synthetic MP-3DHP-like task.

No dataset or checkpoint ships with the reference, so the fidelity figures elsewhere use random calibrated weights (noisy
maps, the worst case for a reduced-precision forward).  This script produces a TRAINED model instead and measures what a
user would see:
  1. scenes   stick-figure persons (14 limb capsules + head disc at the person's depth) over a 4.5 m background, 480 x 640,
              one or two persons per frame; composed by pn_compose_depth, targets by pn_rasterize_targets (popnet_amd.targets)
  2. train    popnet_amd.train.TrainEngine from the reference's initial state (N(0, 0.01) convs), captured hipGraph step
  3. infer    popnet_amd.pipeline.PoseEngine with the trained state_dict on HELD-OUT scenes, fp32 / bf16x3 / bf16
  4. evaluate popnet_amd.metrics (the reference's PCKh / 3D protocol) against the planted ground truth, and
              popnet_amd.fidelity-style agreement of the reduced-precision engines with the fp32 engine
Prints one JSON line.   python scripts/synthetic_train_eval.py [--steps 6000] [--precision bf16x3]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import popnet_amd  # noqa: E402,F401
from popnet_amd import metrics, synth, targets  # noqa: E402
from popnet_amd.config import INTRINSICS  # noqa: E402
from popnet_amd.pipeline import PoseEngine, records_to_numpy  # noqa: E402
from popnet_amd.train import TrainEngine  # noqa: E402

H, W = 640, 480
LIMBS = [(8, 9), (9, 11), (11, 13), (8, 10), (10, 12), (12, 14), (8, 1), (1, 2), (2, 4), (4, 6), (1, 3), (3, 5), (5, 7), (1, 0)]


def scenes(dev, B, seed, persons=(1, 2)):
    """-> (fg_depth [B,S,H,W] f16, fg_mask u8, n_src, bg, kp2d_org [B,S,15,2] f32, kp3d [B,S,15,3] f64, n_persons) on the device."""
    rng = np.random.default_rng(seed)
    S = max(persons)
    k2 = np.zeros((B, S, 15, 2), dtype=np.float32)
    zs = np.full((B, S), 6.0)
    n = rng.integers(persons[0], persons[1] + 1, B).astype(np.int32)
    for b in range(B):
        j, d = synth.planted_persons(rng, int(n[b]), size=224)
        k2[b, :n[b]] = j * np.array([W / 224.0, H / 224.0])
        zs[b, :n[b]] = d
    kp = torch.from_numpy(k2).to(dev)
    ys = torch.arange(H, device=dev, dtype=torch.float32).view(1, 1, H, 1)
    xs = torch.arange(W, device=dev, dtype=torch.float32).view(1, 1, 1, W)
    mask = torch.zeros((B, S, H, W), dtype=torch.bool, device=dev)
    for a, c in LIMBS:                                   # capsule of radius 11 px around every limb segment
        pa, pc = kp[:, :, a], kp[:, :, c]
        ax, ay = pa[..., 0].view(B, S, 1, 1), pa[..., 1].view(B, S, 1, 1)
        dx, dy = (pc[..., 0] - pa[..., 0]).view(B, S, 1, 1), (pc[..., 1] - pa[..., 1]).view(B, S, 1, 1)
        t = (((xs - ax) * dx + (ys - ay) * dy) / (dx * dx + dy * dy + 1e-6)).clamp(0, 1)
        mask |= ((xs - ax - t * dx) ** 2 + (ys - ay - t * dy) ** 2) < 11.0 ** 2
    hx, hy = kp[:, :, 0, 0].view(B, S, 1, 1), kp[:, :, 0, 1].view(B, S, 1, 1)
    mask |= ((xs - hx) ** 2 + (ys - hy) ** 2) < 16.0 ** 2
    valid = (torch.arange(S, device=dev).view(1, S) < torch.from_numpy(n).to(dev).view(B, 1)).view(B, S, 1, 1)
    mask &= valid
    z = torch.from_numpy(zs).to(dev, torch.float32).view(B, S, 1, 1)
    g = torch.Generator(device=dev).manual_seed(seed)
    depth = (z + 0.02 * torch.randn((B, S, H, W), device=dev, generator=g)).clamp(0.3, 5.9).to(torch.float16)
    bg = (4.5 + 0.05 * torch.randn((B, H, W), device=dev, generator=g)).clamp(0, 6).to(torch.float16)
    k3 = np.zeros((B, S, 15, 3))
    fx, fy, cx, cy = (INTRINSICS[k] for k in ("fx", "fy", "cx", "cy"))
    k3[..., 2] = zs[:, :, None]
    k3[..., 0] = (k2[..., 0] - cx) * k3[..., 2] / fx      # pinhole back-projection of the planted joints (util_functions.py:4)
    k3[..., 1] = (k2[..., 1] - cy) * k3[..., 2] / fy
    nt = torch.from_numpy(n).to(dev)
    return depth, mask.to(torch.uint8), nt, bg, kp, torch.from_numpy(k3).to(dev), nt.clone()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=6000)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--lr", type=float, default=0.2)
    ap.add_argument("--precision", default="bf16x3", choices=["fp32", "bf16x3", "fp32-nchw", "bf16x3-nchw"])
    ap.add_argument("--eval-frames", type=int, default=96)
    ap.add_argument("--pool", type=int, default=40, help="distinct training batches (generated once, cycled)")
    ap.add_argument("--seed", type=int, default=0, help="seed of the initial weights")
    ap.add_argument("--save", default=None, help="write the trained state_dict here")
    ap.add_argument("--load", default=None, help="skip the training: evaluate this state_dict (a file written by --save)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    t0 = time.time()
    pool = [[t.contiguous() for t in targets.mpaug_batch(*scenes(dev, args.batch, 1000 + i))] for i in range(args.pool if not args.load else 1)]
    t_data = time.time() - t0
    eng = TrainEngine(synth.init_like_state_dict(seed=args.seed) if not args.load else torch.load(args.load), device=dev, lr=args.lr, precision=args.precision)
    hist = []
    t0 = time.time()
    for k in range(args.steps if not args.load else 0):
        if k == 1:
            eng.capture(*pool[1], warmup_steps=0)
        if k == args.steps * 2 // 3:                      # one step-down of the learning rate
            eng.lr *= 0.2
            eng.capture(*pool[k % args.pool], warmup_steps=0)
        terms = eng.step(*pool[k % args.pool])
        if k % 100 == 0 or k == args.steps - 1:
            hist.append((k, round(float(terms.sum()), 5)))
    torch.cuda.synchronize()
    t_train = time.time() - t0
    sd = {k: v.cpu() for k, v in eng.state_dict().items()}
    if args.save:
        torch.save(sd, args.save)
    del eng

    # held-out scenes through the inference engines
    engines = {p: PoseEngine(precision=p, state_dict=sd, device=dev, max_batch=args.batch) for p in ("fp32", "bf16x3", "bf16")}
    recs = {p: [] for p in engines}
    gt2, gt3 = [], []
    for s in range((args.eval_frames + args.batch - 1) // args.batch):
        fd, fm, n_src, bg, k2, k3, npers = scenes(dev, args.batch, 9000 + s)
        frames = targets.compose_depth(fd, fm, n_src, bg).to(torch.float16)
        for p, e in engines.items():
            recs[p].append(records_to_numpy(e.predict(frames)).copy())
        for b in range(args.batch):
            gt2.append(k2[b, :int(npers[b])].double().cpu().numpy().tolist())
            gt3.append(k3[b, :int(npers[b])].cpu().numpy().tolist())
    out = {"train": {"steps": args.steps, "precision": args.precision, "batch": args.batch, "lr": args.lr, "seconds": round(t_train, 1),
                     "frames_per_s": round(args.steps * args.batch / t_train, 1), "loss": hist, "data_seconds": round(t_data, 1)}, "eval": {}}
    nf = len(gt2)
    for p in engines:
        r = np.concatenate(recs[p])[:nf]
        p2 = [fr["joints_2d"][:int(fr["n_persons"])].tolist() for fr in r]
        p3 = [fr["joints_3d"][:int(fr["n_persons"])].tolist() for fr in r]
        d2avg, kcp2 = metrics.eval_human_dataset_2d_PCKh(p2, gt2, 0, 1)
        d3avg, kcp3 = metrics.eval_human_dataset_3d(p2, gt2, p3, gt3)
        out["eval"][p] = {"frames": nf, "persons_found": int(sum(int(fr["n_persons"]) for fr in r)), "persons_planted": int(sum(len(g) for g in gt2)),
                          "overflow_frames": int(sum(int(fr["status"]) != 0 for fr in r)),
                          "pckh_2d_mean": round(float(np.nanmean(kcp2)), 4), "mean_2d_error_px": round(float(np.nanmean(d2avg)), 3),
                          "pck_3d_10cm_mean": round(float(np.nanmean(kcp3)), 4), "mean_3d_error_m": round(float(np.nanmean(d3avg)), 4)}
    ref = np.concatenate(recs["fp32"])[:nf]
    for p in ("bf16x3", "bf16"):
        r = np.concatenate(recs[p])[:nf]
        same_n = same_a = 0
        d3 = [np.zeros(1)]
        for fa, fb in zip(ref, r):
            na, nb = int(fa["n_persons"]), int(fb["n_persons"])
            if na != nb:
                continue
            same_n += 1
            if int(fa["n_peaks"]) != int(fb["n_peaks"]) or not np.array_equal(fa["person_joint"][:na], fb["person_joint"][:nb]):
                continue
            same_a += 1
            if na:
                vis = fa["person_joint"][:na] >= 0
                d3.append(np.abs(fa["joints_3d"][:na] - fb["joints_3d"][:na])[vis].ravel())
        d3 = np.concatenate(d3)
        out["eval"][p]["vs_fp32"] = {"same_person_count": same_n, "same_assignment": same_a, "d3_m_median": float(np.median(d3)), "d3_m_max": float(d3.max())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
