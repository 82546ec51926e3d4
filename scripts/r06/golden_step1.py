"""How far the step-1 BatchNorm statistics of the fp32 engines sit from the reference golden (tests/golden/train_step.npz): step 1 starts from an lr = 1 update,
so whatever differs in step 0's gradients (a ReLU mask flip, summation order) is amplified.  usage: golden_step1.py"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import state_dict_from_keys, train_case_inputs  # noqa: E402
from popnet_amd.train import TrainEngine  # noqa: E402

G = np.load(os.path.join(ROOT, "tests", "golden", "train_step.npz"))
keys = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_keys.json")))["rtpose_light3d"]
batch = [torch.from_numpy(a).cuda() for a in train_case_inputs()]
for prec in ("fp32", "fp32-nchw", "bf16x3"):
    eng = TrainEngine(state_dict_from_keys(keys, seed=0), device="cuda:0", precision=prec)
    for step in range(2):
        terms = eng.forward_backward(*batch).cpu().numpy()
        eng.apply()
        new = eng.state_dict()
        worst = (0, "")
        for k in G.files:
            if k.startswith("s%d_stat/" % step):
                v, ref = new[k.split("/", 1)[1]].cpu().numpy(), G[k]
                ex = float((np.abs(v - ref) / (1e-3 + 1e-2 * np.abs(ref))).max())        # > 1 fails the test's step-1 tolerance
                worst = max(worst, (ex, k))
        print(prec, "step", step, "terms rel", float(np.abs(terms / G["s%d_terms" % step] - 1).max()), "worst stat excess (x of rtol 1e-2 / atol 1e-3)", worst)
