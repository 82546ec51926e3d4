"""Host-to-device bandwidth of this box for the bench's 19.7 MB batches: one stream, several streams, split copies."""
import time
import torch
dev = torch.device("cuda:0")
N = 32 * 640 * 480 * 2
hosts = [torch.empty(N, dtype=torch.uint8).pin_memory() for _ in range(6)]
devs = [torch.empty(N, dtype=torch.uint8, device=dev) for _ in range(6)]
streams = [torch.cuda.Stream() for _ in range(6)]
def run(nstreams, pieces, reps=30):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(reps):
        for i in range(nstreams):
            with torch.cuda.stream(streams[i]):
                step = N // pieces
                for p in range(pieces):
                    devs[i][p * step:(p + 1) * step].copy_(hosts[i][p * step:(p + 1) * step], non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return reps * nstreams * N / dt / 1e9
for ns in (1, 2, 3, 6):
    for pieces in (1, 4):
        print("streams %d pieces %d: %.1f GB/s" % (ns, pieces, run(ns, pieces)))
big_h = torch.empty(8 * N, dtype=torch.uint8).pin_memory(); big_d = torch.empty(8 * N, dtype=torch.uint8, device=dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): big_d.copy_(big_h, non_blocking=True)
torch.cuda.synchronize(); print("one 157 MB copy: %.1f GB/s" % (10 * 8 * N / (time.perf_counter() - t0) / 1e9))
