"""Sanity run: 60 steps on one synthetic 32-frame batch at the reference's default lr = 1.0 (and 0.05), both precisions:
the loss must fall and stay finite (the graph-captured step)."""
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import popnet_amd  # noqa: E402,F401
from popnet_amd import synth  # noqa: E402
from popnet_amd.train import TrainEngine  # noqa: E402
sys.argv = sys.argv[:1]
import bench  # noqa: E402
dev = torch.device("cuda:0")
batch, _ = bench.synth_training_batch(dev, 32)
for prec in ("fp32", "bf16x3"):
    for lr in (1.0, 0.05):
        eng = TrainEngine(synth.init_like_state_dict(seed=3), device=dev, lr=lr, precision=prec)
        hist = []
        for k in range(60):
            if k == 1:
                eng.capture(*batch, warmup_steps=0)
            hist.append(float(eng.step(*batch).sum()))
        print(prec, "lr", lr, "loss:", " ".join("%.4f" % hist[i] for i in (0, 1, 2, 5, 10, 20, 40, 59)), "finite", bool(np.isfinite(hist).all()))
