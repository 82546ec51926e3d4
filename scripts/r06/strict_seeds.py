"""Which seeds of tests/test_gpu_train.py::test_all_gradients_within_1e4_of_autograd_strict are flip-free for a TrainEngine precision (48x64, B = 2)."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import train_case_inputs
from oracle import train as otrain
from popnet_amd import synth
from popnet_amd.train import TrainEngine
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
for seed in range(1, 13):
    sd = synth.init_like_state_dict(seed=seed)
    batch = [torch.from_numpy(a) for a in train_case_inputs(seed=100 + seed, B=2, H=48, W=64)]
    r = otrain.train_step(sd, *batch, apply=False)
    eng = TrainEngine(sd, device="cuda:0", precision=prec)
    eng.forward_backward(*[t.cuda() for t in batch])
    floor = 1e-6 * max(float(g.double().norm()) / np.sqrt(g.numel()) for g in r["grads"].values())
    worst = 0.0
    for n, g in r["grads"].items():
        ref = float(g.double().norm())
        if ref > 100 * floor * np.sqrt(g.numel()):
            worst = max(worst, float((eng.g[n].double().cpu() - g.double()).norm()) / ref)
    print(prec, "seed", seed, "worst tensor rel err %.2e" % worst, "flip-free" if worst < 2e-5 else "FLIP")
