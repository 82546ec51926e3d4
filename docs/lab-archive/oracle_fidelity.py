#!/usr/bin/env python
"""fp32 and bf16x3 engines against the CPU ORACLE (torch-CPU fp32 forward + NumPy parse = the reference's path), end to end,
per synthetic weight set (calibrate_heads gain 1 = heat values crowd the detection threshold; 6 / 20 = spread).  Prints, per
(gain, precision): frames with identical person assignment, max 3D difference on those, and whether every differing frame is
FRAGILE -- a frame whose ORACLE result itself changes when its fp32 maps are perturbed by one part in 1e5 (the size of the
difference between any two fp32 summation orders).  Feeds the assertions of tests/test_gpu_precision.py.
Usage (GPU box): python docs/lab-archive/oracle_fidelity.py [frames]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from popnet_amd import synth
    from helpers import oracle_records, vs_oracle
    from popnet_amd.pipeline import PoseEngine, records_to_numpy
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    dev = torch.device("cuda", 0)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    for gain in (1.0, 6.0, 20.0):
        engs = {p: PoseEngine(precision=p, device=dev, max_batch=32, calib_gain=gain) for p in ("fp32", "bf16x3", "bf16")}
        sd = {k: v.detach().cpu() for k, v in engs["fp32"].model.state_dict().items()}
        depth = np.concatenate([synth.synth_depth(32, 640, 480, seed=500 + s) for s in range((n + 31) // 32)])[:n]
        ref = oracle_records(depth, sd, perturb=4)
        for p, e in engs.items():
            recs = np.concatenate([records_to_numpy(e.predict(torch.from_numpy(depth[i:i + 32]).to(dev))) for i in range(0, n, 32)])
            r = vs_oracle(recs, ref)
            print("gain %4.1f  %-6s vs oracle: same assignment %d/%d  d3 max %.3g m  conf max %.3g | fragile frames %d, differing frames %s, all differing fragile: %s"
                  % (gain, p, r["same_assignment"], n, r["d3_m_max"], r["conf_max"], r["fragile"], r["differing"], r["differing_all_fragile"]))


if __name__ == "__main__":
    main()
