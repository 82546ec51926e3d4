"""Raw pinned host -> device bandwidth of this box: one 19.7 MB batch per copy, on 1 / 2 / 4 copy streams (halves / quarters of the batch)."""
import time, torch
dev = torch.device("cuda", 0)
h = torch.empty((32, 640, 480), dtype=torch.float16).pin_memory()
d = torch.empty((32, 640, 480), dtype=torch.float16, device=dev)
for nstream in (1, 2, 4):
    ss = [torch.cuda.Stream(device=dev) for _ in range(nstream)]
    per = 32 // nstream
    def once():
        for i, s in enumerate(ss):
            with torch.cuda.stream(s):
                d[i * per:(i + 1) * per].copy_(h[i * per:(i + 1) * per], non_blocking=True)
    for _ in range(3):
        once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        once()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50
    print("%d stream(s): %.3f ms per 19.7 MB batch = %.1f GB/s" % (nstream, dt * 1e3, h.numel() * 2 / dt / 1e9), flush=True)
