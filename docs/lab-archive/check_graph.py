"""Graph replay must produce exactly the records of eager execution (guards the persistent kernels'
tile counters being re-zeroed inside the captured graph)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import popnet_amd
from popnet_amd import synth, _lib
from popnet_amd.pipeline import PoseEngine
dev = torch.device("cuda", 0)
eng = PoseEngine(precision="bf16", device=dev, max_batch=32)
d1 = torch.from_numpy(synth.synth_depth(32, seed=1)).to(dev)
d2 = torch.from_numpy(synth.synth_depth(32, seed=2)).to(dev)
item = _lib.POSE_FRAME_DTYPE.itemsize
out = torch.zeros((32, item), device=dev, dtype=torch.uint8)
static_in = d1.clone()
def body():
    eng.predict(static_in, out)
for _ in range(3): body()
torch.cuda.synchronize()
eager = {}
for name, d in (("d1", d1), ("d2", d2)):
    static_in.copy_(d); body(); torch.cuda.synchronize()
    eager[name] = (eng.heat.clone(), eng.paf.clone(), out.clone())
side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side): body()
torch.cuda.current_stream(dev).wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side): body()
ok = True
for rep in range(3):
    for name, d in (("d1", d1), ("d2", d2)):
        static_in.copy_(d); out.zero_(); eng.heat.zero_(); g.replay(); torch.cuda.synchronize()
        same = torch.equal(eng.heat, eager[name][0]) and torch.equal(eng.paf, eager[name][1])
        recs_g = out.cpu().numpy().view(_lib.POSE_FRAME_DTYPE).reshape(-1); recs_e = eager[name][2].cpu().numpy().view(_lib.POSE_FRAME_DTYPE).reshape(-1)
        same_r = all(int(a['n_peaks']) == int(b['n_peaks']) and int(a['n_persons']) == int(b['n_persons']) for a, b in zip(recs_g, recs_e))
        print(rep, name, "maps identical:", same, "records identical:", same_r, "mean peaks", float(recs_g['n_peaks'].mean()))
        ok &= same and same_r
print("GRAPH_OK" if ok else "GRAPH_MISMATCH")
