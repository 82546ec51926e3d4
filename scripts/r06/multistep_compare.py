"""Five optimiser steps on changing batches with two TrainEngine precisions from the same state: per-tensor relative deviation of parameters and BatchNorm statistics
after every step.  A stale weight pack, a missed statistic or a wrong-step quantity shows as one tensor far from the rest.  usage: multistep_compare.py precA precB [B H W]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import train_case_inputs  # noqa: E402
from popnet_amd import synth  # noqa: E402
from popnet_amd.train import TrainEngine  # noqa: E402

pa, pb = sys.argv[1], sys.argv[2]
B, H, W = (int(v) for v in sys.argv[3:6]) if len(sys.argv) >= 6 else (4, 96, 128)
dev = torch.device("cuda:0")
sd = synth.init_like_state_dict(seed=4)
ea, eb = TrainEngine(sd, device=dev, lr=0.05, precision=pa), TrainEngine(sd, device=dev, lr=0.05, precision=pb)
for step in range(5):
    batch = [torch.from_numpy(a).to(dev) for a in train_case_inputs(seed=40 + step, B=B, H=H, W=W)]
    ta, tb = ea.step(*batch).clone(), eb.step(*batch).clone()
    torch.cuda.synchronize()
    worst = []
    for k in ea.p:
        d = float((ea.p[k].double() - eb.p[k].double()).norm()); n = float(eb.p[k].double().norm()) + 1e-30
        worst.append((d / n, k))
    for k in ea.stats:
        d = float((ea.stats[k].double() - eb.stats[k].double()).norm()); n = float(eb.stats[k].double().norm()) + 1e-30
        worst.append((d / n, "stat:" + k))
    worst.sort(reverse=True)
    print("step", step, "terms rel", float(((ta - tb).abs() / tb.abs()).max()), "worst", ["%.2e %s" % w for w in worst[:4]], "median %.2e" % np.median([w[0] for w in worst]))
