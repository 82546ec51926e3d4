"""What does the PCIe hand-over cost the pipelined region, and why?  (run on the GPU box)
Same StreamingEngine / step structure as bench.py; regions of K steps, median of 3, for several hand-over variants:
  resident      submit(j): inputs already in HBM (bench `value`)
  h2d           submit_host(pinned batch): the product path (bench `h2d_inclusive`)
  h2d_nowait    the copy is issued but the step does not wait for it (timing only: isolates the cross-stream dependency)
  h2d_tiny      same dependency structure, 4 KB copies (isolates the PCIe / SDMA traffic itself)
  h2d_ahead2    the copy of step k + 2 is ENQUEUED before step k's graph (host-side order only)
"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from popnet_amd import _lib, synth
from popnet_amd.pipeline import PoseEngine, StreamingEngine

dev = torch.device("cuda", 0)
B, PIPE, POOL, K = 32, 3, 6, int(sys.argv[1]) if len(sys.argv) > 1 else 200
se = StreamingEngine(PoseEngine, depth=PIPE, pool=POOL, wire=True, graph=True, precision="bf16", device=dev, max_batch=B)
pinned = [torch.from_numpy(synth.synth_depth(B, 640, 480, seed=1234 + i)).pin_memory() for i in range(PIPE * POOL)]
for i, h in enumerate(pinned):
    se.input(i // POOL, i % POOL).copy_(h)
torch.cuda.synchronize()
se.capture()

def submit_variant(se, host_batch, mode):
    s = se._tickets % se.depth
    j = se._next_buf[s]
    se._next_buf[s] = (j + 1) % se.pool
    cs = se.copy_streams[s]
    if se._released[s][j] is not None:
        cs.wait_event(se._released[s][j])
    with torch.cuda.stream(cs):
        if mode == "h2d_tiny":
            se.inputs[s][j].view(-1)[:2048].copy_(host_batch.view(-1)[:2048], non_blocking=True)
        else:
            se.inputs[s][j][:len(host_batch)].copy_(host_batch, non_blocking=True)
        se._copied[s][j].record(cs)
    if mode != "h2d_nowait":
        se.streams[s].wait_event(se._copied[s][j])
    t = se.submit(j)
    if se._released[s][j] is None:
        se._released[s][j] = torch.cuda.Event()
    se._released[s][j].record(se.streams[s])
    return t

zero8 = torch.zeros((), device=dev, dtype=torch.uint8)
keep = torch.empty((K, B, _lib.POSE_WIRE_DTYPE.itemsize), device=dev, dtype=torch.uint8)


def region(mode):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        if mode.startswith("resident"):
            t = se.submit((k // PIPE) % POOL)
        elif mode in ("h2d", "h2d_keep", "h2d_keepk"):
            t = se.submit_host(pinned[k % len(pinned)])
        else:
            t = submit_variant(se, pinned[k % len(pinned)], mode)
        if mode.endswith("_keep"):                      # bench.py keeps every step's records (consistency check, the gather)
            with torch.cuda.stream(se.stream(t)):
                keep[k].copy_(se.wires[t % PIPE], non_blocking=True)
        if mode.endswith("_keepk"):                     # the same copy as an elementwise KERNEL (no hipMemcpyAsync)
            with torch.cuda.stream(se.stream(t)):
                torch.bitwise_or(se.wires[t % PIPE], zero8, out=keep[k])
    se.join()
    torch.cuda.synchronize()
    return time.perf_counter() - t0

for mode in ("resident", "h2d", "resident_keep", "h2d_keep", "resident_keepk", "h2d_keepk", "h2d_keep", "h2d_keepk"):
    region(mode)
    runs = sorted(region(mode) for _ in range(3))
    print("%-12s %8.1f frames/s  (%.4f ms/step; runs %s)" % (mode, K * B / runs[1], runs[1] / K * 1e3, ["%.0f" % (K * B / r) for r in runs]), flush=True)
