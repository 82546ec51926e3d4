"""Times the HIP training step (popnet_amd.train.TrainEngine) on synthetic data: B frames of 224x224, seeded init-like weights.
usage: python scripts/train_bench.py [B] [steps] [fp32|bf16x3]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import popnet_amd  # noqa: E402,F401
from popnet_amd import synth  # noqa: E402
from popnet_amd.train import TrainEngine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
prec = sys.argv[3] if len(sys.argv) > 3 else "fp32"
dev = torch.device("cuda:0")
sd = synth.init_like_state_dict(seed=3)
eng = TrainEngine(sd, device=dev, lr=0.01, precision=prec)
rng = np.random.default_rng(1)
batch = [torch.from_numpy(a).to(dev) for a in (rng.normal(0, 1, (B, 1, 224, 224)).astype(np.float32), rng.uniform(0, 1, (B, 16, 28, 28)).astype(np.float32),
                                              rng.uniform(-1, 1, (B, 28, 28, 28)).astype(np.float32), rng.uniform(-1.5, 1.5, (B, 15, 28, 28)).astype(np.float32),
                                              (rng.uniform(0, 1, (B, 15, 28, 28)) < 0.2).astype(np.float32))]
if os.environ.get("TRAIN_GRAPH", "1") != "0" and "nograph" not in sys.argv:
    eng.capture(*batch)
for _ in range(2):
    t = eng.step(*batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    t = eng.step(*batch)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
flops = 3 * 13.343e9 * B          # forward + data gradient + weight gradient (13.343 GFLOP per frame forward)
print(prec, "B=%d  %.2f ms/step  %.0f frames/s  %.1f TFLOP/s (3 x forward FLOPs)  loss terms %s" % (B, dt * 1e3, B / dt, flops / dt / 1e12, t.cpu().numpy()))
# host-side enqueue time per step (how far the Python loop runs ahead of the GPU)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    eng.step(*batch)
host = (time.perf_counter() - t0) / steps
torch.cuda.synchronize()
print("host enqueue %.2f ms/step" % (host * 1e3))
