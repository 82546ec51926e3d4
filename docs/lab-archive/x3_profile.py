"""Per-kernel time of one precision mode (run under rocprofv3 --kernel-trace --stats): N eager steps of one PoseEngine."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import popnet_amd  # noqa
from popnet_amd import synth
from popnet_amd.pipeline import PoseEngine
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
e = PoseEngine(precision=prec, device="cuda:0", max_batch=32)
d = torch.from_numpy(synth.synth_depth(32, 640, 480, seed=3)).cuda()
for _ in range(3):
    e.predict(d)
torch.cuda.synchronize()
for _ in range(n):
    e.predict(d)
torch.cuda.synchronize()
print("done", prec, n)
