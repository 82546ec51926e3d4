"""One-off wide fuzz of the GPU pose parse against the oracle (the committed test runs a handful of settings):
many seeds x noise levels x drop probabilities x blob widths, 16 frames each; prints every mismatch.
Run from the repo root on the GPU box:  python3 docs/lab-archive/parse_fuzz_sweep.py [n_seeds]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
import popnet_amd  # noqa: F401
from popnet_amd import synth
from popnet_amd.utils.paf_to_pose import frame_assoc, frame_joint_list, make_parse_cfg, parse_paf_batch
from oracle import parse_paf as O
from popnet_amd.config import default_cfg

gpu = torch.device("cuda:0")
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
bad = checked = flagged = 0
t0 = time.time()
for seed in range(n_seeds):
    for noise in (0.0, 0.02, 0.05, 0.08):
        for drop in (0.0, 0.15, 0.3):
            for sigma in (0.6, 0.8, 1.2):
                persons = [(seed + 3 * i + 1) % 9 for i in range(16)]
                heat, paf, z = synth.planted_batch(1000 * seed + int(noise * 1000) + int(drop * 100) + int(sigma * 10), persons,
                                                   noise=noise, drop_prob=drop, sigma=sigma)
                frames = parse_paf_batch(*(torch.from_numpy(a).to(gpu) for a in (heat, paf, z)), make_parse_cfg(default_cfg()))
                for b in range(16):
                    fr = frames[b]
                    if int(fr["status"]):
                        flagged += 1
                        continue
                    checked += 1
                    rec = O.frame_to_records(heat[b].transpose(1, 2, 0).copy(), paf[b].transpose(1, 2, 0).copy(), z[b].transpose(1, 2, 0).copy())
                    jl, assoc = frame_joint_list(fr), frame_assoc(fr)
                    ref_assoc = np.asarray(rec["assoc"]).reshape(-1, 17)
                    n = ref_assoc.shape[0]
                    ok = jl.shape == np.asarray(rec["joint_list"]).shape and (jl.size == 0 or np.array_equal(jl, rec["joint_list"]))
                    ok = ok and int(fr["n_persons"]) == n
                    if ok and n:
                        ok = (np.array_equal(assoc[:, :15], ref_assoc[:, :15]) and np.array_equal(fr["joints_2d"][:n], np.array(rec["humans_2d"]))
                              and np.array_equal(fr["joints_3d"][:n], np.array(rec["humans_3d"])) and np.array_equal(fr["part_conf"][:n], np.array(rec["conf"])))
                    if not ok:
                        bad += 1
                        print("MISMATCH seed %d noise %.2f drop %.2f sigma %.1f frame %d (persons %d vs %d)" % (seed, noise, drop, sigma, b, int(fr["n_persons"]), n), flush=True)
print("checked %d frames, %d flagged (limits), %d mismatches, %.0f s" % (checked, flagged, bad, time.time() - t0))
