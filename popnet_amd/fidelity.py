"""How far a reduced-precision engine is from the fp32 parity engine, END TO END (records, not maps).

north_star asks for joints within 1e-3 m and bit-exact person assignment against the fp32 CPU reference; the fp32
engine meets that against the reference's own scripts (tests/test_gpu_parity.py).  This module measures the other
precision modes against the fp32 engine on the bench workload, frame by frame:
  same_assignment   frames whose person count AND person -> peak-id table (person_to_joint_assoc[:, :15]) are identical
  d3_*              |delta| of the 3D joints (metres) of the frames with identical assignment
It is what tests/test_gpu_precision.py asserts and what bench.py reports under "fidelity"."""
import numpy as np
import torch

from . import synth
from .pipeline import records_to_numpy


def compare_engines(ref, eng, n_frames=96, seed0=500):
    bs = min(ref.max_batch, eng.max_batch)
    frames = same_count = same_assign = 0
    d2, d3, dconf = [np.zeros(1)], [np.zeros(1)], [np.zeros(1)]
    for s in range((n_frames + bs - 1) // bs):
        depth = torch.from_numpy(synth.synth_depth(bs, 640, 480, seed=seed0 + s)).to(ref.device)
        a, b = records_to_numpy(ref.predict(depth)), records_to_numpy(eng.predict(depth))
        for fa, fb in zip(a, b):
            frames += 1
            na, nb = int(fa["n_persons"]), int(fb["n_persons"])
            if na != nb or int(fa["status"]) or int(fb["status"]):
                continue
            same_count += 1
            if int(fa["n_peaks"]) != int(fb["n_peaks"]) or not np.array_equal(fa["person_joint"][:na], fb["person_joint"][:nb]):
                continue
            same_assign += 1
            if na:
                vis = fa["person_joint"][:na] >= 0
                d2.append(np.abs(fa["joints_2d"][:na] - fb["joints_2d"][:na])[vis].ravel())
                d3.append(np.abs(fa["joints_3d"][:na] - fb["joints_3d"][:na])[vis].ravel())
                dconf.append(np.abs(fa["part_conf"][:na] - fb["part_conf"][:na])[vis].ravel())
    d2, d3, dconf = np.concatenate(d2), np.concatenate(d3), np.concatenate(dconf)
    return {"frames": frames, "same_person_count": same_count, "same_assignment": same_assign,
            "d2_px_max": float(d2.max()), "d3_m_median": float(np.median(d3)), "d3_m_p95": float(np.percentile(d3, 95)),
            "d3_m_max": float(d3.max()), "conf_max": float(dconf.max()), "joints_compared": int(d3.size // 3)}
