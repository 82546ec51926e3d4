"""Runs N bf16 forwards of rtpose_light3d at B=32 (for rocprofv3 --pmc passes)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import popnet_amd
from popnet_amd.pipeline import PoseEngine
from popnet_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
eng = PoseEngine(precision="bf16", device="cuda:0", max_batch=32)
d = torch.from_numpy(synth.synth_depth(32)).cuda()
for _ in range(n):
    eng.predict(d)
torch.cuda.synchronize()
