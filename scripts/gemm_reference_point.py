"""Reference point for profiles/README.md: what the vendor GEMM library (hipBLASLt / rocBLAS through torch.matmul)
reaches on the plain GEMMs that have the convolutions' M x N x K -- without any im2col gather, halo, bias, residual
or activation, i.e. an upper bound for an im2col + library-GEMM design of the same layers.  bf16 in, fp32 accumulate."""
import time
import torch

assert torch.cuda.is_available()
dev = torch.device("cuda:0")
shapes = [  # (label, pixels, K = Cin * taps, Cout)
    ("28x28 256->256 3x3", 32 * 28 * 28, 256 * 9, 256),
    ("28x28 128->128 3x3", 32 * 28 * 28, 128 * 9, 128),
    ("56x56 128->128 3x3", 32 * 56 * 56, 128 * 9, 128),
    ("112x112 64->64 3x3", 32 * 112 * 112, 64 * 9, 64),
    ("28x28 256->128 1x1", 32 * 28 * 28, 256, 128),
    ("large square 8192^3", 8192, 8192, 8192),
]
for label, n, k, m in shapes:
    a = torch.randn(n, k, device=dev, dtype=torch.bfloat16)
    b = torch.randn(k, m, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        c = a @ b
    torch.cuda.synchronize()
    iters = 200 if n * k * m < 1e12 else 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        c = a @ b
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print("%-22s [%7d x %5d] x [%5d x %4d]: %8.2f us  %7.1f TFLOP/s" % (label, n, k, k, m, us, 2.0 * n * k * m / us / 1e6))
