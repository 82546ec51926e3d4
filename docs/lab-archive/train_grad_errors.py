"""Per-parameter gradient error of the HIP training step against fp64 autograd, next to the error of torch's own fp32 CPU
autograd against the same fp64 result (how much of the difference is conditioning, not implementation)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import popnet_amd  # noqa: E402,F401
from helpers import state_dict_from_keys, train_case_inputs  # noqa: E402
from oracle import train as otrain  # noqa: E402
from popnet_amd.train import TrainEngine  # noqa: E402
import json  # noqa: E402

keys = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_keys.json")))["rtpose_light3d"]
MODE = sys.argv[1] if len(sys.argv) > 1 else "golden"
if MODE == "golden":
    sd = state_dict_from_keys(keys, seed=0)
    batch = [torch.from_numpy(a) for a in train_case_inputs()]
elif MODE == "tiny":        # tiny seed: init-like weights, 48x64 input, B = 2 -- few enough activations that no ReLU mask flips
    from test_gpu_train import init_like_state_dict
    seed = int(sys.argv[2])
    sd = init_like_state_dict(keys, seed=seed)
    batch = [torch.from_numpy(a) for a in train_case_inputs(seed=100 + seed, B=2, H=48, W=64)]
else:                       # init B: the state a run starts from, 224x224
    from test_gpu_train import init_like_state_dict
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    sd = init_like_state_dict(keys, seed=3)
    rng = np.random.default_rng(8 + B)
    batch = [torch.from_numpy(a) for a in (rng.normal(0, 1, (B, 1, 224, 224)).astype(np.float32), rng.uniform(0, 1, (B, 16, 28, 28)).astype(np.float32),
                                           rng.uniform(-1, 1, (B, 28, 28, 28)).astype(np.float32), rng.uniform(-1.5, 1.5, (B, 15, 28, 28)).astype(np.float32),
                                           (rng.uniform(0, 1, (B, 15, 28, 28)) < 0.2).astype(np.float32))]
r32 = otrain.train_step(sd, *batch, apply=False)
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
r64 = otrain.train_step(sd64, *[b.double() for b in batch], apply=False, dtype=torch.float64)
eng = TrainEngine(sd, device="cuda:0", precision=os.environ.get("TRAIN_PREC", "fp32"))
terms = eng.forward_backward(*[b.cuda() for b in batch]).cpu().numpy()
print("terms hip", terms, "\nterms f32", r32["terms"], "\nterms f64", r64["terms"])
rows = []
for n, g64 in r64["grads"].items():
    ref = float(g64.norm())
    e_hip = float((eng.g[n].double().cpu() - g64).norm())
    e_t32 = float((r32["grads"][n].double() - g64).norm())
    rows.append((e_hip / max(ref, 1e-30), e_t32 / max(ref, 1e-30), ref, n))
rows.sort(reverse=True)
for r in rows[:25]:
    print("hip %.2e  torch32 %.2e  |g| %.3e  %s" % r)
print("median hip %.2e torch32 %.2e" % (np.median([r[0] for r in rows]), np.median([r[1] for r in rows])))
# against torch fp32 (the oracle the tests use)
num = den = 0.0
bad = []
for n, g32 in r32["grads"].items():
    e = float((eng.g[n].double().cpu() - g32.double()).norm())
    ref = float(g32.double().norm())
    num += e * e
    den += ref * ref
    if e > 1e-4 * ref:
        e64 = float((g32.double() - r64["grads"][n]).norm())
        bad.append((e / ref, e64 / ref, ref, n))
print("global rel err vs torch32: %.3e   (|g| %.3e)" % (np.sqrt(num / den), np.sqrt(den)))
for b in sorted(bad, reverse=True):
    print("vs32 %.2e  torch32-vs-f64 %.2e  |g| %.3e  %s" % b)
