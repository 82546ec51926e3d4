"""How far the bf16 throughput mode is from the fp32 parity mode on the bench workload (documentation aid):
per frame: same number of persons? same joint assignment? 2D / 3D joint differences of the matching persons."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import popnet_amd
from popnet_amd import synth
from popnet_amd.pipeline import PoseEngine, records_to_numpy
n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
e32 = PoseEngine(precision="fp32", device="cuda:0", max_batch=32)
e16 = PoseEngine(precision="bf16", device="cuda:0", max_batch=32)
same_n = same_assign = frames = 0
d2, d3 = [], []
for s in range(n // 32):
    depth = torch.from_numpy(synth.synth_depth(32, 640, 480, seed=500 + s)).cuda()
    a, b = records_to_numpy(e32.predict(depth)), records_to_numpy(e16.predict(depth))
    for fa, fb in zip(a, b):
        frames += 1
        na, nb = int(fa["n_persons"]), int(fb["n_persons"])
        if na != nb:
            continue
        same_n += 1
        va, vb = fa["person_joint"][:na] >= 0, fb["person_joint"][:nb] >= 0
        if not np.array_equal(va, vb):
            continue
        same_assign += 1
        if na:
            d2.append(np.abs(fa["joints_2d"][:na] - fb["joints_2d"][:na])[va].ravel())
            d3.append(np.abs(fa["joints_3d"][:na] - fb["joints_3d"][:na])[va].ravel())
d2, d3 = np.concatenate(d2), np.concatenate(d3)
print("frames %d: same person count %d, same visible-joint pattern %d" % (frames, same_n, same_assign))
print("2D |diff| px: median %.3g p95 %.3g max %.3g ; 3D |diff| m: median %.3g p95 %.3g max %.3g" % (
    np.median(d2), np.percentile(d2, 95), d2.max(), np.median(d3), np.percentile(d3, 95), d3.max()))
