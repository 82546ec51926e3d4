"""One-off wide fuzz of the GPU YOLO decode (parse_prior_pose) against the oracle: many seeds x confidence ranges (sparse to
dense candidate sets, near-threshold confidences, heavy box overlap for the NMS keep-loop quirk).
Run from the repo root on the GPU box:  python3 docs/lab-archive/yolo_fuzz_sweep.py [n_seeds]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
import popnet_amd  # noqa: F401
from popnet_amd.utils.prior_pose_align import parse_prior_pose
from oracle import parse_yolo as O

ANCHORS = [(6, 3), (12, 6)]
gpu = torch.device("cuda:0")
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
bad = checked = 0
t0 = time.time()
for seed in range(n_seeds):
    for hi in (0.505, 0.52, 0.56, 0.62, 0.75):          # fraction of cells above the 0.5 threshold grows with `hi`
        for wh in ((0.5, 2.0), (1.5, 2.0), (0.2, 0.6)):  # box sizes: mixed, large (heavy overlap), small (no overlap)
            rng = np.random.default_rng(100 * seed + int(hi * 1000) + int(wh[0] * 10))
            pm = rng.uniform(-1, 1, (32, 100, 14, 14)).astype(np.float32)
            pm[:, 4] = rng.uniform(0, hi, (32, 14, 14))
            pm[:, 54] = rng.uniform(0, hi, (32, 14, 14))
            pm[:, 2:4] = rng.uniform(wh[0], wh[1], (32, 2, 14, 14))
            pm[:, 52:54] = rng.uniform(wh[0], wh[1], (32, 2, 14, 14))
            if seed % 3 == 0:                             # exact ties in confidence: the sort must break them like torch does
                pm[:, 4] = np.round(pm[:, 4] * 50) / 50
                pm[:, 54] = np.round(pm[:, 54] * 50) / 50
            try:
                b, h, v = parse_prior_pose(torch.from_numpy(pm).to(gpu), ANCHORS, 15, 224, 224, 3, 2, 0.5, 0.5)
            except Exception as e:                        # candidate overflow is reported, not silent
                print("seed %d hi %.3f wh %s: %s" % (seed, hi, wh, str(e)[:80]), flush=True)
                continue
            rb, rh, rv = O.parse_prior_pose(pm.copy(), ANCHORS, 15, 224, 224, 3, 2, 0.5, 0.5)
            for i in range(32):
                checked += 1
                ok = len(b[i]) == len(rb[i])
                if ok and len(b[i]):
                    ok = np.array_equal(np.array(b[i]), np.array(rb[i])) and np.array_equal(np.array(h[i]), np.array(rh[i])) and np.array_equal(np.array(v[i]), np.array(rv[i]))
                if not ok:
                    bad += 1
                    print("MISMATCH seed %d hi %.3f wh %s frame %d (%d vs %d detections)" % (seed, hi, wh, i, len(b[i]), len(rb[i])), flush=True)
print("checked %d frames, %d mismatches, %.0f s" % (checked, bad, time.time() - t0))
