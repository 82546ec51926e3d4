"""Trains the synthetic task (scripts/synthetic_train_eval.py's loop) for N steps with a TrainEngine precision and prints checkpoint statistics that decide how well a
16-bit inference mode can follow fp32: smallest running variances, largest folded BatchNorm scales, largest weights.  usage: ckpt_stats.py <precision> [steps]"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("ste", os.path.join(ROOT, "scripts", "synthetic_train_eval.py"))
ste = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ste)
from popnet_amd import synth, targets  # noqa: E402
from popnet_amd.train import TrainEngine  # noqa: E402

prec = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
dev = torch.device("cuda:0")
pool = [[t.contiguous() for t in targets.mpaug_batch(*ste.scenes(dev, 32, 1000 + i))] for i in range(40)]
eng = TrainEngine(synth.init_like_state_dict(seed=0), device=dev, lr=0.2, precision=prec)
for k in range(steps):
    if k == steps * 2 // 3:
        eng.lr *= 0.2
    t = eng.step(*pool[k % 40])
    if k % 250 == 0 or k == steps - 1:
        print(prec, "step", k, "loss", float(t.sum()), "terms", t.cpu().numpy().round(5))
sd = eng.state_dict()
rows = []
for k, v in sd.items():
    if k.endswith("running_var"):
        g = sd[k.replace("running_var", "weight")]
        sc = (g / torch.sqrt(v + 1e-5)).abs()
        rows.append((float(v.min()), float(sc.max()), k))
rows.sort()
print("smallest running_var / largest |gamma / sqrt(var)|:")
for r in rows[:6]:
    print("   var_min %.3e  scale_max %.3e  %s" % r)
print("largest folded scales:", sorted(((b, c) for a, b, c in rows), reverse=True)[:4])
ratio = []
for k, v in sd.items():
    if k.endswith("running_var"):
        mu = sd[k.replace("running_var", "running_mean")]
        r = (mu.abs() / torch.sqrt(v + 1e-5))
        ratio.append((float(r.max()), float(r.median()), k))
ratio.sort(reverse=True)
print("largest |running_mean| / sqrt(running_var) per BatchNorm (max, median over channels):")
for r in ratio[:8]:
    print("   %.2f  %.2f  %s" % r)
bias = sorted(((float(v.abs().max()), k) for k, v in sd.items() if k.endswith(".bias") and (k.replace(".bias", ".weight") in sd) and sd[k.replace(".bias", ".weight")].dim() == 4), reverse=True)
print("largest conv biases:", bias[:4])
wmax = max((float(v.abs().max()), k) for k, v in sd.items() if k.endswith(".weight") and v.dim() == 4)
print("largest conv weight:", wmax, " any non-finite:", any(not torch.isfinite(v).all() for v in sd.values() if v.is_floating_point()))
