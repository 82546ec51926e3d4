#!/usr/bin/env python
"""Per-layer mixed precision inside the bf16x3 net ("look for a cheaper tolerance-meeting configuration"):
POPNET_X3_BF16_CONVS runs the named convolutions as plain bf16 (one MFMA pass on the hi plane instead of three); every
configuration is compared with the fp32 engine on 96 frames of both synthetic weight sets (popnet_amd.fidelity) and timed
(one engine, eager).  Usage (GPU box): python docs/lab-archive/mixed_precision.py"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CONFIGS = [("bf16x3 everywhere", ""),
           ("layer1 (112x112 BasicBlocks, 28 % of the FLOPs) in bf16", "model0.layer1"),
           ("whole stem (model0) in bf16", "model0."),
           ("stage 1 in bf16", "model1_"),
           ("first conv of every stage branch in bf16", "model1_1.0,model1_2.0,model1_3.0,model2_1.0,model2_2.0,model2_3.0"),
           ("stage-2 heat branch only in bf16", "model2_2."),
           ("everything but the last conv of each stage-2 branch in bf16", "model0.,model1_,model2_1.0,model2_1.3,model2_1.6,model2_1.9,model2_2.0,model2_2.3,model2_2.6,model2_2.9,model2_3.0,model2_3.3,model2_3.6,model2_3.9")]


def main():
    from popnet_amd import synth
    from popnet_amd.fidelity import compare_engines
    from popnet_amd.pipeline import PoseEngine
    dev = torch.device("cuda", 0)
    depth = torch.from_numpy(synth.synth_depth(32, 640, 480, seed=77)).to(dev)
    ref = {g: PoseEngine(precision="fp32", device=dev, max_batch=32, calib_gain=g) for g in (1.0, 6.0)}
    for name, pats in CONFIGS:
        if pats:
            os.environ["POPNET_X3_BF16_CONVS"] = pats
        else:
            os.environ.pop("POPNET_X3_BF16_CONVS", None)
        eng = {g: PoseEngine(precision="bf16x3", device=dev, max_batch=32, calib_gain=g) for g in (1.0, 6.0)}
        e = eng[1.0]
        for _ in range(3):
            e.predict(depth)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            e.predict(depth)
        torch.cuda.synchronize()
        fps = 30 * 32 / (time.perf_counter() - t0)
        r1, r6 = compare_engines(ref[1.0], eng[1.0], 96), compare_engines(ref[6.0], eng[6.0], 96)
        print(json.dumps({"config": name, "frames_per_s_one_engine_eager": round(fps, 1),
                          "threshold_calibrated": {k: r1[k] for k in ("same_assignment", "d3_m_max", "d3_m_p95")},
                          "separated": {k: r6[k] for k in ("same_assignment", "d3_m_max", "d3_m_p95")},
                          "meets_tolerance": bool(r1["d3_m_max"] < 1e-3 and r6["d3_m_max"] < 1e-3 and r1["same_assignment"] >= 94 and r6["same_assignment"] >= 94)}))
        sys.stdout.flush()
        del eng


if __name__ == "__main__":
    main()
