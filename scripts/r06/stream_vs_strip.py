"""Gradients of the row-streaming and the strip weight-gradient kernels from the SAME state on real training batches (synthetic_train_eval's scenes), step after step:
engine S (strip, POPNET_TRAINX_WG_STREAM=0) drives the trajectory, engine R (row-streaming) gets S's parameters copied in before every step."""
import importlib.util
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("ste", os.path.join(ROOT, "scripts", "synthetic_train_eval.py"))
ste = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ste)
from popnet_amd import synth, targets  # noqa: E402
from popnet_amd.train import TrainEngine  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
pool = [[t.contiguous() for t in targets.mpaug_batch(*ste.scenes(dev, B, 1000 + i))] for i in range(40)]
sd = synth.init_like_state_dict(seed=0)
os.environ["POPNET_TRAINX_WG_STREAM"] = "0"
S = TrainEngine(sd, device=dev, lr=0.2, precision="bf16x3")
S.forward_backward(*pool[0])                      # finalizes S's trainer with the strip kernel
os.environ.pop("POPNET_TRAINX_WG_STREAM")
R = TrainEngine(sd, device=dev, lr=0.2, precision="bf16x3")
for k in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    R.flat_p.copy_(S.flat_p)
    for n in S.stats:
        R.stats[n].copy_(S.stats[n])
    b = pool[k % 40]
    S.forward_backward(*b)
    R.forward_backward(*b)
    torch.cuda.synchronize()
    worst = sorted(((float((R.g[n].double() - S.g[n].double()).norm() / (S.g[n].double().norm() + 1e-30)), n) for n in S.g if S.g[n].dim() == 4), reverse=True)[:3]
    fin = all(bool(torch.isfinite(R.g[n]).all()) for n in R.g)
    if k < 5 or worst[0][0] > 1e-5 or k % 100 == 0:
        print("step", k, "finite", fin, "worst conv-weight gradient deviations:", ["%.2e %s" % w for w in worst], "| all-tensor worst %.2e" % max(float((R.g[n].double() - S.g[n].double()).norm() / (S.g[n].double().norm() + 1e-30)) for n in S.g if S.g[n].double().norm() > 1e-7))
    S.apply()
