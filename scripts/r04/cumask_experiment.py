"""Experiment (round 4): SPATIAL instead of temporal multiplexing of the in-flight batches.  Each slot's stream is created with a CU mask
(hipExtStreamCreateWithCUMask): a third of the chip per batch.  A latency-bound launch (heads, 1x1, parse: < 2 blocks per CU on the whole
chip) takes about as long on 85 CUs as on 256 and leaves the other two thirds to the other batches; a persistent workgroup per CU
(bb64_kernel) no longer shuts the other batches out.  Compared with the product (three unmasked streams) in the same process.
    python3 scripts/r04/cumask_experiment.py [steps] [layout: contig | interleave | xcd]"""
import ctypes as C, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from popnet_amd import _lib, synth
from popnet_amd.pipeline import PoseEngine, StreamingEngine

dev = torch.device("cuda", 0)
B, POOL, K = 32, 6, int(sys.argv[1]) if len(sys.argv) > 1 else 200
layout = sys.argv[2] if len(sys.argv) > 2 else "contig"
PIPE = int(sys.argv[3]) if len(sys.argv) > 3 else 3
prec = sys.argv[4] if len(sys.argv) > 4 else "bf16"
hip = C.CDLL("libamdhip64.so")
NCU = torch.cuda.get_device_properties(0).multi_processor_count

def masked_stream(cus):
    words = (C.c_uint32 * 8)()
    for cu in cus:
        words[cu // 32] |= (1 << (cu % 32))
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev), len(cus)

def partition(i, n):
    if layout == "contig":
        lo, hi = i * NCU // n, (i + 1) * NCU // n
        return list(range(lo, hi))
    if layout == "interleave":
        return [c for c in range(NCU) if c % n == i]
    if layout == "xcd":                       # bit b <-> (xcd b % 8, cu b // 8) if the mask enumerates CUs round-robin over the XCDs: give whole XCDs
        xs = [x for x in range(8) if x % n == i]
        return [c for c in range(NCU) if (c % 8) in xs]
    raise SystemExit("layout?")

def run(masked):
    se = StreamingEngine(PoseEngine, depth=PIPE, pool=POOL, wire=True, graph=True, precision=prec, device=dev, max_batch=B)
    if masked:
        sts = [masked_stream(partition(i, PIPE)) for i in range(PIPE)]
        se.streams = [s for s, _ in sts]
        cur = torch.cuda.current_stream(dev)
        for st in se.streams:
            st.wait_stream(cur)
    for i in range(PIPE * POOL):
        se.input(i // POOL, i % POOL).copy_(torch.from_numpy(synth.synth_depth(B, 640, 480, seed=1234 + i)).to(dev))
    torch.cuda.synchronize()
    se.capture()
    def region():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(K):
            se.submit((k // PIPE) % POOL)
        se.join(); torch.cuda.synchronize()
        return time.perf_counter() - t0
    region()
    runs = sorted(region() for _ in range(3))
    rec = se.wires[0].clone()
    return K * B / runs[1], ["%.0f" % (K * B / r) for r in runs], rec

v0, r0, rec0 = run(False)
print("unmasked           %8.1f frames/s %s" % (v0, r0), flush=True)
v1, r1, rec1 = run(True)
print("CU-masked (%s, %d slots of ~%d CUs) %8.1f frames/s %s   same records: %s" % (layout, PIPE, NCU // PIPE, v1, r1, bool(torch.equal(rec0, rec1))), flush=True)
