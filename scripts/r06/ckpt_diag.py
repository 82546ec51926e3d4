"""Where a 16-bit inference engine leaves the fp32 one on a given checkpoint: final maps and the stage-1 / feature activations, bf16 and bf16x3 against fp32.
usage: ckpt_diag.py <state_dict.pth>"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("ste", os.path.join(ROOT, "scripts", "synthetic_train_eval.py"))
ste = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ste)
from popnet_amd import targets  # noqa: E402
from popnet_amd.pipeline import PoseEngine, records_to_numpy  # noqa: E402

sd = torch.load(sys.argv[1])
dev = torch.device("cuda:0")
fd, fm, n_src, bg, k2, k3, npers = ste.scenes(dev, 32, 9000)
frames = targets.compose_depth(fd, fm, n_src, bg).to(torch.float16)
eng = {p: PoseEngine(precision=p, state_dict=sd, device=dev, max_batch=32) for p in ("fp32", "bf16x3", "bf16")}
rec = {p: records_to_numpy(e.predict(frames)).copy() for p, e in eng.items()}
torch.cuda.synchronize()
print("persons per frame fp32 ", rec["fp32"]["n_persons"][:16])
print("persons per frame bf16 ", rec["bf16"]["n_persons"][:16])
print("peaks per frame fp32   ", rec["fp32"]["n_peaks"][:16])
print("peaks per frame bf16   ", rec["bf16"]["n_peaks"][:16])
for p in ("bf16x3", "bf16"):
    for name in ("heat", "paf", "z"):
        a, b = getattr(eng["fp32"], name)[:32].float(), getattr(eng[p], name)[:32].float()
        print("%-7s %-5s max|fp32| %.3f  max|diff| %.3e  mean|diff| %.3e" % (p, name, float(a.abs().max()), float((a - b).abs().max()), float((a - b).abs().mean())))
