"""Dev check of the NHWC-planes training engine (TrainEngine(precision="bf16x3")) against the CPU oracle: loss terms and per-tensor
gradient errors, plus the running statistics.  usage: trainx_check.py [B H W]   (POPNET_TRAINX_WGRAD=legacy: train.hip's weight gradient)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from helpers import train_case_inputs  # noqa: E402
from oracle import train as otrain  # noqa: E402
from popnet_amd import synth  # noqa: E402
from popnet_amd.train import TrainEngine  # noqa: E402

B, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (2, 48, 64)
prec = sys.argv[4] if len(sys.argv) > 4 else "bf16x3"
dev = torch.device("cuda:0")
sd = synth.init_like_state_dict(seed=2)
batch = [torch.from_numpy(a) for a in train_case_inputs(seed=300 + H, B=B, H=H, W=W)]
t0 = time.time()
r = otrain.train_step(sd, *batch, apply=False)
print("oracle %.1fs terms" % (time.time() - t0), np.array(r["terms"]))
eng = TrainEngine(sd, device=dev, precision=prec)
terms = eng.forward_backward(*[t.to(dev) for t in batch]).cpu().numpy()
torch.cuda.synchronize()
print("engine terms", terms)
print("terms rel err", np.abs(terms - np.array(r["terms"])) / np.abs(np.array(r["terms"])))
worst = []
num = den = 0.0
for name, gr in r["grads"].items():
    g = eng.g[name].double().cpu()
    err, ref = float((g - gr.double()).norm()), float(gr.double().norm())
    num += err * err
    den += ref * ref
    worst.append((err / (ref + 1e-30), name, ref, float(g.norm())))
worst.sort(reverse=True)
print("whole-vector rel err %.3e" % np.sqrt(num / den))
for w in worst[:25]:
    print("  %.3e  %-40s |ref| %.3e |got| %.3e" % w)
print("  ...")
for w in worst[-5:]:
    print("  %.3e  %-40s |ref| %.3e |got| %.3e" % w)
if "new_sd" in r:
    for k, v in [(k, v) for k, v in r["new_sd"].items() if k.endswith("running_mean") or k.endswith("running_var")][:8]:
        print(k, float((eng.stats[k].cpu() - v).abs().max()))
