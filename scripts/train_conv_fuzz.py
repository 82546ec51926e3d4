"""Random-shape sweep of the training convolution primitives (forward, data gradient, weight / bias gradient) against torch CPU,
both precision modes: ragged channel counts, maps narrower / wider than a tile, 1-row maps, batch 1..5.
usage: python scripts/train_conv_fuzz.py [cases = 120] [seed = 0]"""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import popnet_amd  # noqa: E402,F401
from popnet_amd import _lib  # noqa: E402


def rel(a, b):
    a, b = a.double().cpu().ravel(), b.double().cpu().ravel()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def run(cases=120, seed=0, verbose=True):
    rng = np.random.default_rng(seed)
    L = _lib.lib()
    dev = torch.device("cuda:0")
    s = _lib.current_stream_ptr(dev)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None      # noqa: E731
    worst = {"fp32": 0.0, "bf16x3": 0.0}
    for case in range(cases):
        ks = int(rng.choice([1, 3, 3, 3]))
        N, Cin, Cout = int(rng.integers(1, 6)), int(rng.integers(1, 200)), int(rng.integers(1, 200))
        H, W = int(rng.integers(1, 40)), int(rng.integers(1, 150))
        pad = ks // 2 if rng.random() < 0.8 else int(rng.integers(0, ks))
        if H + 2 * pad < ks or W + 2 * pad < ks:
            continue
        g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
        x = torch.randn(N, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, ks, ks, generator=g) / np.sqrt(Cin * ks * ks)
        b = torch.randn(Cout, generator=g)
        xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
        yr = F.conv2d(xr, wr, br, 1, pad)
        dy = torch.randn(yr.shape, generator=g)
        yr.backward(dy)
        xd, wd, bd, dyd = x.to(dev), w.to(dev), b.to(dev), dy.to(dev)
        for prec in ("fp32", "bf16x3"):
            ctx = _lib.Context(0)
            ctx.check(L.pn_train_set_precision(ctx.handle, _lib.PN_PREC_BF16X3 if prec == "bf16x3" else 0), "precision")
            y, dx, dw, db = torch.full(yr.shape, 9.0, device=dev), torch.full(x.shape, 9.0, device=dev), torch.full(w.shape, 9.0, device=dev), torch.full(b.shape, 9.0, device=dev)
            ctx.check(L.pn_conv2d_forward(ctx.handle, p(xd), p(wd), p(bd), p(y), N, Cin, H, W, Cout, ks, 1, pad, 0, s), "fwd")
            ctx.check(L.pn_conv2d_dgrad(ctx.handle, p(dyd), p(wd), p(dx), N, Cin, H, W, Cout, ks, pad, 0, s), "dgrad")
            ctx.check(L.pn_conv2d_wgrad(ctx.handle, p(xd), p(dyd), p(dw), p(db), N, Cin, H, W, Cout, ks, 1, pad, s), "wgrad")
            errs = (rel(y, yr.detach()), rel(dx, xr.grad), rel(dw, wr.grad), rel(db, br.grad))
            tol = 2e-5 if prec == "fp32" else 2e-4
            worst[prec] = max(worst[prec], max(errs[:3]))
            if max(errs[:3]) > tol or errs[3] > 2e-5 or not all(np.isfinite(errs)):
                raise AssertionError("case %d %s N=%d Cin=%d Cout=%d H=%d W=%d ks=%d pad=%d: fwd %.2e dgrad %.2e wgrad %.2e dbias %.2e" % ((case, prec, N, Cin, Cout, H, W, ks, pad) + errs))
    if verbose:
        print("%d cases ok; worst relative error fp32 %.2e, bf16x3 %.2e" % (cases, worst["fp32"], worst["bf16x3"]))
    return worst


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 120, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
