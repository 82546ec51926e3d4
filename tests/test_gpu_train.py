"""HIP training step (popnet_amd.train.TrainEngine + the pn_conv2d_* / pn_bn_train_* / pn_head_* / pn_sgd_nesterov kernels)
against the oracle (oracle/train.py = torch fp32 CPU autograd, pinned by the reference's own training step in
tests/golden/train_step.npz) and against those goldens directly (BASELINE configs[4]).

Bar: one training step's loss and ALL 5 525 814 gradients within 1e-4 relative of the reference module's
autograd on seeded inputs, per parameter tensor (||g - g_ref|| / ||g_ref||, with an absolute floor of 1e-6 of the largest
per-entry gradient rms for the tensors whose gradient is analytically zero: biases of convs that feed a BatchNorm hold
rounding noise on both sides).
What that bar can and cannot mean.  ReLU / LeakyReLU backward multiplies by a 0/1 (0.1/1) mask decided by the SIGN of a
pre-activation.  Two correct fp32 evaluations (this one and torch's, or torch's fp32 and its own fp64) differ by ~1e-7
relative in those pre-activations, so among millions of them a few land on different sides of zero; each such flip changes
the gradient of everything upstream by ~1 / sqrt(elements of that layer) ~ 2e-3 relative -- a discrete event, not an
accuracy defect.  Measured (docs/lab-archive/train_grad_errors.py): at a size with 250 k activations 5 of 6 seeds have NO
flip and then every tensor agrees to 2e-6; at 224 x 224 torch's OWN fp32 differs from its fp64 by 1e-3..8e-3 on whole branches.
Hence two kinds of test: (1) strict 1e-4 per tensor, no exceptions, at a flip-free size (the arithmetic is right);
(2) at the training configuration's size and on the reference goldens: the error against fp64 autograd must be in the
same class as torch fp32's own error against fp64 (median, maximum and whole-vector), and the loss terms agree to 1e-5.
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import sample_indices, state_dict_from_keys, train_case_inputs

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "train_step.npz"))
REL = 1e-4


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _rel(a, b):
    a, b = a.double().cpu().ravel(), b.double().cpu().ravel()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("shape", [
    # N, Cin, H, W, Cout, ks, stride, pad
    (2, 1, 64, 48, 64, 7, 2, 3),        # the stem
    (3, 64, 28, 20, 64, 3, 1, 1),
    (2, 187, 12, 16, 256, 3, 1, 1),     # stage-2 entry (ragged Cin)
    (2, 256, 12, 16, 128, 1, 1, 0),
    (2, 128, 12, 16, 28, 1, 1, 0),      # ragged Cout
    (2, 64, 9, 7, 15, 3, 1, 1),         # odd map, ragged Cout
    (1, 128, 56, 56, 128, 3, 1, 1),
    (2, 64, 20, 112, 64, 3, 1, 1),      # 112 columns: two 56-column tiles in bf16x3 mode
    (1, 40, 11, 70, 72, 3, 1, 1),       # ragged everything
])
@pytest.mark.parametrize("prec", ["fp32", "bf16x3", "bf16x3-wide"])
def test_conv_forward_dgrad_wgrad_vs_torch(gpu, shape, prec, monkeypatch):
    """bf16x3 (pn_train_set_precision): the 3x3 forward / data gradient on split-bf16 MFMA -- 16 mantissa bits per operand, so
    1e-4 of the tensor norm instead of 1e-5 (measured ~1e-5); every other kernel is the fp32 one in both modes."""
    from popnet_amd import _lib
    N, Cin, H, W, Cout, ks, stride, pad = shape
    L, ctx = _lib.lib(), _lib.Context(0)
    if prec == "bf16x3-wide":                # the 256-slot kernel, which the dispatcher only takes for launches of >= 448 blocks
        monkeypatch.setenv("POPNET_TRAIN_X3_WIDE", "1")
    ctx.check(L.pn_train_set_precision(ctx.handle, _lib.PN_PREC_BF16X3 if prec != "fp32" else 0), "precision")
    TOL = 1e-5 if prec == "fp32" else 1e-4
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / np.sqrt(Cin * ks * ks)
    b = torch.randn(Cout, generator=g)
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    yr = F.conv2d(xr, wr, br, stride, pad)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    xd, wd, bd, dyd = x.to(gpu), w.to(gpu), b.to(gpu), dy.to(gpu)
    y = torch.full(yr.shape, 7.0, device=gpu)
    s = _lib.current_stream_ptr(torch.device(gpu))
    ctx.check(L.pn_conv2d_forward(ctx.handle, _p(xd), _p(wd), _p(bd), _p(y), N, Cin, H, W, Cout, ks, stride, pad, 0, s), "fwd")
    assert _rel(y, yr.detach()) < TOL, _rel(y, yr.detach())
    ctx.check(L.pn_conv2d_forward(ctx.handle, _p(xd), _p(wd), None, _p(y), N, Cin, H, W, Cout, ks, stride, pad, 1, s), "fwd+=")     # accumulate, no bias
    assert _rel(y, 2 * yr.detach() - b.view(1, -1, 1, 1)) < TOL
    dw, db = torch.zeros_like(wd), torch.zeros_like(bd)
    ctx.check(L.pn_conv2d_wgrad(ctx.handle, _p(xd), _p(dyd), _p(dw), _p(db), N, Cin, H, W, Cout, ks, stride, pad, s), "wgrad")
    assert _rel(dw, wr.grad) < TOL and _rel(db, br.grad) < 1e-5, _rel(dw, wr.grad)
    dw2 = torch.zeros_like(wd)
    ctx.check(L.pn_conv2d_wgrad(ctx.handle, _p(xd), _p(dyd), _p(dw2), None, N, Cin, H, W, Cout, ks, stride, pad, s), "wgrad")
    assert torch.equal(dw, dw2)            # split reduction in slice order: deterministic
    if stride == 1:
        dx = torch.full(x.shape, 3.0, device=gpu)
        ctx.check(L.pn_conv2d_dgrad(ctx.handle, _p(dyd), _p(wd), _p(dx), N, Cin, H, W, Cout, ks, pad, 0, s), "dgrad")
        assert _rel(dx, xr.grad) < TOL, _rel(dx, xr.grad)
        ctx.check(L.pn_conv2d_dgrad(ctx.handle, _p(dyd), _p(wd), _p(dx), N, Cin, H, W, Cout, ks, pad, 1, s), "dgrad+=")
        assert _rel(dx, 2 * xr.grad) < TOL


@pytest.mark.parametrize("shape", [(8, 64, 28, 28, 128), (4, 256, 28, 28, 256), (4, 64, 56, 56, 64), (2, 64, 112, 112, 64), (6, 187, 28, 28, 128),
                                   (3, 96, 30, 20, 72), (2, 64, 27, 40, 64), (1, 32, 9, 64, 64)])
def test_vectorised_weight_gradient_equals_the_ping_pong_kernel_bit_for_bit(gpu, shape, monkeypatch):
    """Round 5: tconv3_wgrad_x3v_kernel (dY fragments straight from global memory, the X halo fetched by rows with 16-byte loads, one
    memory round trip of staging per tile) walks the same tiles and slices and issues the same MFMAs in the same order as
    tconv3_wgrad_x3pp_kernel (POPNET_TRAIN_WGRAD_NOVEC=1): identical weight gradients, bit for bit, on the network's map sizes, ragged
    channel counts, partial last row tiles and the narrowest / widest tile widths -- and within the split-bf16 tolerance of autograd."""
    from popnet_amd import _lib
    N, Cin, H, W, Cout = shape
    L, ctx = _lib.lib(), _lib.Context(0)
    ctx.check(L.pn_train_set_precision(ctx.handle, _lib.PN_PREC_BF16X3), "precision")
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(N, Cin, H, W, generator=g)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / np.sqrt(Cin * 9)).requires_grad_()
    yr = F.conv2d(x, w, None, 1, 1)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    xd, dyd = x.to(gpu), dy.to(gpu)
    s = _lib.current_stream_ptr(torch.device(gpu))
    out = []
    for novec in (False, True):
        if novec:
            monkeypatch.setenv("POPNET_TRAIN_WGRAD_NOVEC", "1")          # read at every call
        dw = torch.zeros((Cout, Cin, 3, 3), device=gpu)
        ctx.check(L.pn_conv2d_wgrad(ctx.handle, _p(xd), _p(dyd), _p(dw), None, N, Cin, H, W, Cout, 3, 1, 1, s), "wgrad")
        out.append(dw.clone())
    monkeypatch.delenv("POPNET_TRAIN_WGRAD_NOVEC")
    ctx.check(L.pn_train_set_precision(ctx.handle, 0), "precision")
    torch.cuda.synchronize()
    assert torch.isfinite(out[0]).all() and torch.equal(out[0], out[1])
    assert _rel(out[0], w.grad) < 1e-4, _rel(out[0], w.grad)


@pytest.mark.parametrize("hw", [(13, 9), (12, 8)])       # odd map: scalar kernels; H * W a multiple of 4: the 16-byte variants
@pytest.mark.parametrize("act,with_res", [(0, False), (1, False), (1, True), (2, False)])
def test_bn_train_forward_backward_vs_torch(gpu, act, with_res, hw):
    from popnet_amd import _lib
    L, ctx = _lib.lib(), _lib.Context.for_device(0)
    N, Cc, (H, W) = 3, 70, hw
    g = torch.Generator().manual_seed(5 + act)
    x = torch.randn(N, Cc, H, W, generator=g) * 2 + 0.5
    gamma, beta = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g)
    res = torch.randn(N, Cc, H, W, generator=g) if with_res else None
    rm, rv = torch.randn(Cc, generator=g), torch.rand(Cc, generator=g) + 0.5
    dy = torch.randn(N, Cc, H, W, generator=g)
    xr, gr, br = x.clone().requires_grad_(), gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    rr = res.clone().requires_grad_() if with_res else None
    rm_r, rv_r = rm.clone(), rv.clone()
    o = F.batch_norm(xr, rm_r, rv_r, gr, br, True, 0.1, 1e-5)
    if with_res:
        o = o + rr
    o = F.relu(o) if act == 1 else F.leaky_relu(o, 0.1) if act == 2 else o
    o.backward(dy)
    d = lambda t: t.to(gpu) if t is not None else None          # noqa: E731
    xd, gd, bd, rd, rmd, rvd, dyd = d(x), d(gamma), d(beta), d(res), d(rm), d(rv), d(dy)
    y, mean, invstd = torch.empty_like(xd), torch.empty(Cc, device=gpu), torch.empty(Cc, device=gpu)
    s = _lib.current_stream_ptr(torch.device(gpu))
    ctx.check(L.pn_bn_train_forward(ctx.handle, _p(xd), _p(gd), _p(bd), _p(rd), _p(y), _p(mean), _p(invstd), _p(rmd), _p(rvd), 0.1, 1e-5, act, N, Cc, H * W, s), "bn fwd")
    assert _rel(y, o.detach()) < 1e-5 and _rel(rmd, rm_r) < 1e-6 and _rel(rvd, rv_r) < 1e-6
    dx, dg, db = torch.empty_like(xd), torch.empty(Cc, device=gpu), torch.empty(Cc, device=gpu)
    dres = torch.full_like(xd, 1.0) if with_res else None
    ctx.check(L.pn_bn_train_backward(ctx.handle, _p(xd), _p(dyd), _p(y), _p(gd), _p(bd), _p(mean), _p(invstd), act, N, Cc, H * W, _p(dx), _p(dg), _p(db), _p(dres), 1, s), "bn bwd")
    assert _rel(dx, xr.grad) < 2e-5 and _rel(dg, gr.grad) < 1e-5 and _rel(db, br.grad) < 1e-5
    if not with_res:                                            # out = NULL: the mask recomputed from x -- the same gradients, bit for bit
        dx2, dg2, db2 = torch.empty_like(xd), torch.empty(Cc, device=gpu), torch.empty(Cc, device=gpu)
        ctx.check(L.pn_bn_train_backward(ctx.handle, _p(xd), _p(dyd), None, _p(gd), _p(bd), _p(mean), _p(invstd), act, N, Cc, H * W, _p(dx2), _p(dg2), _p(db2), None, 0, s), "bn bwd")
        assert torch.equal(dx, dx2) and torch.equal(dg, dg2) and torch.equal(db, db2)
    if with_res:
        assert _rel(dres, rr.grad + 1.0) < 1e-6            # accumulated onto what was there


def test_avgpool_head_sgd_vs_torch(gpu):
    from popnet_amd import _lib
    L, ctx = _lib.lib(), _lib.Context.for_device(0)
    s = _lib.current_stream_ptr(torch.device(gpu))
    g = torch.Generator().manual_seed(9)
    for (H, W) in ((14, 10), (13, 9)):
        x = torch.randn(2, 5, H, W, generator=g, requires_grad=True)
        yr = F.avg_pool2d(x, 3, 2, 1)
        dy = torch.randn(yr.shape, generator=g)
        yr.backward(dy)
        y, dx = torch.empty(yr.shape, device=gpu), torch.empty(x.shape, device=gpu)
        xd, dyd = x.detach().to(gpu), dy.to(gpu)
        ctx.check(L.pn_avgpool3s2_forward(ctx.handle, _p(xd), _p(y), 10, H, W, s), "pool")
        ctx.check(L.pn_avgpool3s2_backward(ctx.handle, _p(dyd), _p(dx), 10, H, W, s), "pool bwd")
        assert _rel(y, yr.detach()) < 1e-6 and _rel(dx, x.grad) < 1e-6
    # heads: kind 1 with fg weights into a channel slice, extra upstream gradient
    N, Cc, h, w, LD = 2, 15, 6, 5, 40
    v = torch.randn(N, Cc, h, w, generator=g, requires_grad=True)
    t, fg = torch.randn(N, Cc, h, w, generator=g), (torch.rand(N, Cc, h, w, generator=g) < 0.3).float()
    extra = torch.randn(N, LD, h, w, generator=g)
    out_r = (v.sigmoid() - 0.5) * 4
    loss_r = (((out_r - t) ** 2) * (0.1 + 0.9 * fg)).mean()
    (loss_r + (out_r * extra[:, 7:7 + Cc]).sum()).backward()
    vd, td, fgd, exd = v.detach().to(gpu), t.to(gpu), fg.to(gpu), extra.to(gpu)
    sg, cat, loss, dv = torch.empty_like(vd), torch.zeros(N, LD, h, w, device=gpu), torch.zeros(1, device=gpu), torch.empty_like(vd)
    ctx.check(L.pn_head_forward(ctx.handle, _p(vd), _p(td), _p(fgd), 1, N, Cc, h * w, _p(sg), C.c_void_p(cat.data_ptr() + 7 * h * w * 4), LD, _p(loss), s), "head")
    assert _rel(cat[:, 7:7 + Cc], out_r.detach()) < 1e-6 and float(cat[:, :7].abs().max()) == 0 and float(cat[:, 7 + Cc:].abs().max()) == 0
    assert abs(float(loss) - float(loss_r.detach())) < 1e-6 * float(loss_r.detach())
    ctx.check(L.pn_head_backward(ctx.handle, _p(sg), _p(td), _p(fgd), C.c_void_p(exd.data_ptr() + 7 * h * w * 4), LD, 1, N, Cc, h * w, _p(dv), s), "head bwd")
    assert _rel(dv, v.grad) < 1e-5
    # Nesterov SGD, three steps, against torch.optim.SGD -- without and with weight decay (--weight-decay of the trainer)
    for wd in (0.0, 1e-2):
        p = torch.randn(1000, generator=g)
        pr = p.clone().requires_grad_()
        opt = torch.optim.SGD([pr], lr=0.7, momentum=0.9, nesterov=True, weight_decay=wd)
        pd, buf = p.to(gpu), torch.zeros(1000, device=gpu)
        for k in range(3):
            gr = torch.randn(1000, generator=g)
            pr.grad = gr.clone()
            opt.step()
            ctx.check(L.pn_sgd_nesterov(ctx.handle, _p(pd), _p((gr * 2).to(gpu)), _p(buf), 1000, 0.7, 0.9, wd, 1 if k == 0 else 0, 0.5, s), "sgd")
            assert _rel(pd, pr.detach()) < 1e-6


def _engine(golden, gpu, **kw):
    from popnet_amd.train import TrainEngine
    return TrainEngine(state_dict_from_keys(golden.keys["rtpose_light3d"], seed=0), device=gpu, **kw)


def _floor(ref_grads):
    return 1e-6 * max(float(g.double().norm()) / np.sqrt(g.numel()) for g in ref_grads.values())


def _compare_grads_strict(eng, ref_grads):
    """Every parameter tensor: ||g - g_ref|| <= REL ||g_ref|| + floor.  Returns the worst ratio among the non-degenerate tensors."""
    floor, worst = _floor(ref_grads), 0.0
    assert set(eng.g) == set(ref_grads) and sum(g.numel() for g in ref_grads.values()) == eng.n_params
    for name, gr in ref_grads.items():
        err, ref = float((eng.g[name].double().cpu() - gr.double()).norm()), float(gr.double().norm())
        assert err <= REL * ref + floor * np.sqrt(gr.numel()), (name, err, ref)
        if ref > 100 * floor * np.sqrt(gr.numel()):
            worst = max(worst, err / ref)
    return worst


def _accuracy_class(eng, ref32, ref64):
    """Errors against fp64 autograd, this implementation next to torch fp32: (median, max, whole-vector) for both."""
    floor = _floor(ref32)
    eh, et, nh, nt, den = [], [], 0.0, 0.0, 0.0
    assert set(eng.g) == set(ref32)
    for name, g64 in ref64.items():
        ref = float(g64.norm())
        a, b = float((eng.g[name].double().cpu() - g64).norm()), float((ref32[name].double() - g64).norm())
        nh, nt, den = nh + a * a, nt + b * b, den + ref * ref
        if ref > 100 * floor * np.sqrt(g64.numel()):
            eh.append(a / ref)
            et.append(b / ref)
    return (np.median(eh), max(eh), np.sqrt(nh / den)), (np.median(et), max(et), np.sqrt(nt / den))


def _assert_same_class(hip, t32):
    """Error against fp64 autograd (median over tensors, maximum, whole vector) within 4x of torch fp32's own -- or within the
    budget of a handful of mask flips (median 5e-3, max 3e-2, whole vector 1e-2): whether and where a pre-activation lands
    on the other side of zero is luck on BOTH sides (on the golden case torch fp32 has no flip at all and this kernel two; at
    224 x 224, B = 2 torch has an early one and this kernel a late one), so neither side's error bounds the other's."""
    for h, t, what, cap in zip(hip, t32, ("median", "max", "whole vector"), (5e-3, 3e-2, 1e-2)):
        assert h <= max(4 * t + REL, cap), (what, hip, t32)


def _f64(sd):
    return {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}


def init_like_state_dict(keys, seed):
    """The state a training run starts from: every conv weight N(0, 0.01) (rtpose_light3d._initialize_weights_norm, :358-362),
    conv biases U(+-1/sqrt(fan_in)) (nn.Conv2d default), BatchNorm weight 1 / bias 0 / mean 0 / var 1."""
    g = torch.Generator().manual_seed(seed)
    shapes = dict((k, tuple(s)) for k, s in keys)
    sd = {}
    for k, shp in shapes.items():
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.tensor(0, dtype=torch.long)
        elif k.endswith("running_mean"):
            sd[k] = torch.zeros(shp)
        elif k.endswith("running_var"):
            sd[k] = torch.ones(shp)
        elif len(shp) == 4:
            sd[k] = torch.randn(shp, generator=g) * 0.01
        elif k.endswith(".weight"):
            sd[k] = torch.ones(shp)                                   # BatchNorm weight
        elif k[:-len(".bias")] + ".running_mean" in shapes:
            sd[k] = torch.zeros(shp)                                  # BatchNorm bias
        else:
            w = shapes[k[:-len(".bias")] + ".weight"]
            sd[k] = (torch.rand(shp, generator=g) * 2 - 1) / np.sqrt(w[1] * w[2] * w[3])
    return sd


def test_training_step_equals_reference_goldens_and_oracle(gpu, golden):
    """The golden case (B = 3, 96x128, O(1)-scale seeded weights): what the reference's own module, loss, backward() and
    torch.optim.SGD produced for two consecutive steps.  Step 0 starts from identical parameters: loss terms 2e-5, gradients
    in torch fp32's accuracy class + the stored samples, parameters / BatchNorm statistics after the step.  Step 1 starts
    from parameters that already differ by lr x (gradient differences) with lr = 1: its loss terms are compared at 5e-4, its
    gradients against the oracle restarted from the ENGINE's state, and the momentum arithmetic exactly."""
    from oracle import train as otrain
    eng = _engine(golden, gpu)
    assert eng.n_params == 5525814 and eng.flat_p.numel() - eng.n_params < 4 * len(eng.p)     # every parameter; <= 3 padding floats per tensor
    sd = state_dict_from_keys(golden.keys["rtpose_light3d"], seed=0)
    batch = [torch.from_numpy(a) for a in train_case_inputs()]
    dbatch = [b.to(gpu) for b in batch]
    for step in range(2):
        r = otrain.train_step(sd, *batch, apply=False)
        r64 = otrain.train_step(_f64(sd), *[b.double() for b in batch], apply=False, dtype=torch.float64)
        terms = eng.forward_backward(*dbatch).cpu().numpy()
        tol = 2e-5 if step == 0 else 5e-4
        assert np.allclose(terms, G["s%d_terms" % step], rtol=tol, atol=0), (terms, G["s%d_terms" % step])
        assert np.allclose(terms, r["terms"], rtol=2e-5, atol=0)
        _assert_same_class(*_accuracy_class(eng, r["grads"], r64["grads"]))
        close = total = 0                                           # the reference's own stored samples: a mask flip (module
        for name in eng.g:                                        # docstring) moves single entries, so: nearly all close, none wild
            gs = eng.g[name].cpu().numpy().ravel()
            rms = float(G["s%d_g_norm/%s" % (step, name)]) / np.sqrt(gs.size)
            fl = 1e-6 * float(G["s%d_g_norm/model2_1.12.weight" % step])
            d = np.abs(gs[sample_indices(name, gs.size)] - G["s%d_g_samp/%s" % (step, name)])
            assert d.max() <= (1 + 2 * step) * rms + fl, name
            close += int((d <= (2e-2 if step == 0 else 0.2) * rms + fl).sum())
            total += d.size
        assert close >= (0.98 if step == 0 else 0.8) * total, (close, total)
        p_old, m_old, g_now = eng.flat_p.clone(), eng.flat_m.clone(), eng.flat_g.clone()
        eng.apply()
        b = g_now if step == 0 else 0.9 * m_old + g_now            # torch.optim.SGD: buf = g, then mu buf + g; p -= lr (g + mu buf)
        assert _rel(eng.flat_p, p_old - 1.0 * (g_now + 0.9 * b)) < 1e-6 and _rel(eng.flat_m, b) < 1e-6
        new = eng.state_dict()
        for name in eng.p:
            v = new[name].cpu().numpy().ravel()
            ref = G["s%d_p_samp/%s" % (step, name)]
            rms = float(G["s%d_g_norm/%s" % (step, name)]) / np.sqrt(v.size)
            assert np.abs(v[sample_indices(name, v.size)] - ref).max() <= (1 + 9 * step) * (4 * rms + 1e-5 * max(1.0, float(np.abs(ref).max()))), (step, name)
        for k in G.files:
            if k.startswith("s%d_stat/" % step):
                # step 1 follows an lr = 1 update: the engines sit at 0.7 (NCHW fp32) / 1.07 (planes fp32) / 1.4 (planes bf16x3) of (1e-2, 1e-3) from the reference's
                # own step-1 statistics on this case (scripts/r06/golden_step1.py) while all agree with it to 3e-5 of that in step 0 -- hence (2e-2, 2e-3)
                assert np.allclose(new[k.split("/", 1)[1]].cpu().numpy(), G[k], rtol=2e-5 if step == 0 else 2e-2, atol=2e-6 if step == 0 else 2e-3), k
        sd = {k: v.cpu() for k, v in new.items()}                  # the oracle restarts from the engine's state
    assert int(new["model0.bn1.num_batches_tracked"]) == 2
    # the checkpoint goes straight into the inference engine (same keys as the reference's)
    assert set(new) == {k for k, _ in golden.keys["rtpose_light3d"]}


def test_reference_trainer_body_runs_with_import_swaps_only(gpu, golden):
    """VERDICT r02 item 8, the training side of the drop-in boundary: the per-batch body of the reference trainer
    (tpm/train_rtpose_light3d_kdh3d_mpaug.py:160-180,313-316 (CR)) verbatim -- DataParallel(model).cuda(), model.train(),
    `_, saved_for_loss = model(img)`, rtpose_light3d_loss_fgweight, optimizer.zero_grad(), total_loss.backward(),
    torch.optim.SGD(lr 1, momentum 0.9, nesterov).step() -- with ONLY the two imports swapped to popnet_amd.  autograd orders
    the calls; every convolution / BatchNorm / pooling, forward and backward, is a HIP primitive (network/_autograd.py).
    Two consecutive steps on the golden batch against (a) the reference's own stored loss terms, (b) TrainEngine (pinned to the
    goldens and the oracle above: same kernels, so the gradients agree to rounding of the element-wise head glue), (c) the
    momentum arithmetic of torch.optim.SGD on the module's parameters, (d) the BatchNorm statistics the reference stored."""
    from popnet_amd.network.rtpose_light3d import rtpose_light3d                 # was: from lib.network.rtpose_light3d import ...
    from popnet_amd.network.losses import build_names, rtpose_light3d_loss_fgweight   # was: from lib.network.losses import ...
    sd = state_dict_from_keys(golden.keys["rtpose_light3d"], seed=0)
    model = rtpose_light3d(15, 14, 2, input_dim=1)
    model.load_state_dict(sd)
    model = torch.nn.DataParallel(model, device_ids=[gpu.index]).cuda(gpu)
    params = [p for p in model.parameters() if p.requires_grad]
    optimizer = torch.optim.SGD(params, lr=1.0, momentum=0.9, weight_decay=0.0, nesterov=True)
    names = build_names(model.module.num_stages)
    eng = _engine(golden, gpu, precision="fp32-nchw")                            # the engine built on the SAME kernels the autograd wrappers call
    img, heatmap_target, paf_target, posedepth_target, fg_masks = [torch.from_numpy(a).cuda(gpu) for a in train_case_inputs()]
    model.train()
    for step in range(2):
        _, saved_for_loss = model(img)
        total_loss, saved_for_log = rtpose_light3d_loss_fgweight(saved_for_loss, heatmap_target, paf_target, posedepth_target, fg_masks, 2, names)
        optimizer.zero_grad()
        total_loss.backward()
        terms = np.array([saved_for_log[n] for n in names])
        tol = 2e-5 if step == 0 else 5e-4                                        # step 1 starts from lr = 1 x (gradient rounding) apart
        assert np.allclose(terms, G["s%d_terms" % step], rtol=tol, atol=0), (terms, G["s%d_terms" % step])
        assert np.allclose([saved_for_log[k] for k in ("max_ht", "min_ht", "max_paf", "min_paf", "max_z", "min_z")], G["s%d_extrema" % step], rtol=1e-4 if step == 0 else 1e-2, atol=1e-5 if step == 0 else 1e-3)
        eterms = eng.forward_backward(img, heatmap_target, paf_target, posedepth_target, fg_masks).cpu().numpy()
        assert np.allclose(terms, eterms, rtol=1e-5 if step == 0 else 5e-4, atol=0)
        if step == 0:                                                            # identical parameters on both sides: same kernels, same gradients
            named = dict(model.module.named_parameters())
            assert set(eng.g) == {k for k in named if not k.startswith("model0.layer3")}
            num = den = 0.0
            gmax = max(float(ge.double().norm()) for ge in eng.g.values())
            for k, ge in eng.g.items():
                d = float((named[k].grad - ge).double().norm())
                n = float(ge.double().norm())
                # (a conv bias in front of a BatchNorm has a mathematically zero gradient: both sides hold rounding noise there)
                assert d <= 1e-4 * n + 1e-7 * gmax, (k, d, n)
                num, den = num + d * d, den + n * n
            assert np.sqrt(num / den) < 2e-5
        before = {k: p.detach().clone() for k, p in model.module.named_parameters() if p.grad is not None}
        grads = {k: p.grad.detach().clone() for k, p in model.module.named_parameters() if p.grad is not None}
        bufs = {k: optimizer.state[p]["momentum_buffer"].clone() for k, p in model.module.named_parameters() if p in optimizer.state and "momentum_buffer" in optimizer.state[p]}
        optimizer.step()
        for k, p in model.module.named_parameters():
            if k not in grads:
                continue
            b = grads[k] if step == 0 else 0.9 * bufs[k] + grads[k]
            assert _rel(p.detach(), before[k] - 1.0 * (grads[k] + 0.9 * b)) < 1e-6, k
        eng.apply()
        new = {k: v for k, v in model.module.state_dict().items()}
        for k in G.files:
            if k.startswith("s%d_stat/" % step):
                assert np.allclose(new[k.split("/", 1)[1]].cpu().numpy(), G[k], rtol=2e-5 if step == 0 else 1e-2, atol=2e-6 if step == 0 else 1e-3), k
    assert int(model.module.state_dict()["model0.bn1.num_batches_tracked"]) == 2
    # back to inference with the trained weights: eval() re-folds them into the MFMA-packed net
    model.eval()
    model.module.precision = "fp32"
    with torch.no_grad():
        (paf, heat, z), _ = model.module(torch.randn(2, 1, 224, 224, device=gpu))          # the path's network input size
    assert torch.isfinite(paf).all() and tuple(heat.shape) == (2, 16, 28, 28)


@pytest.mark.parametrize("seed", [2, 3, 4, 6, 7, 8])
def test_all_gradients_within_1e4_of_autograd_strict(gpu, golden, seed):
    """The whole network, the state a run STARTS from (init_like_state_dict), 48x64 input, B = 2 -- few enough activations
    that no ReLU mask flips for these seeds (which seeds flip depends on the engine's summation order: 1 and 12 of 1..14 on the round-6 planes
    engine, 5 on the NCHW engine of rounds 2-5; module docstring): loss terms and ALL 5 525 814 gradients within 1e-4 relative of CPU autograd per
    tensor, no exception; in fact within 2e-5, the whole vector within 1e-5."""
    from oracle import train as otrain
    from popnet_amd.train import TrainEngine
    sd = init_like_state_dict(golden.keys["rtpose_light3d"], seed=seed)
    batch = [torch.from_numpy(a) for a in train_case_inputs(seed=100 + seed, B=2, H=48, W=64)]
    r = otrain.train_step(sd, *batch, apply=False)
    eng = TrainEngine(sd, device=gpu)
    terms = eng.forward_backward(*[t.to(gpu) for t in batch]).cpu().numpy()
    assert np.allclose(terms, r["terms"], rtol=1e-5, atol=0)
    assert _compare_grads_strict(eng, r["grads"]) < 2e-5
    num = sum(float((eng.g[n].double().cpu() - g.double()).norm()) ** 2 for n, g in r["grads"].items())
    den = sum(float(g.double().norm()) ** 2 for g in r["grads"].values())
    assert np.sqrt(num / den) < 1e-5


@pytest.mark.parametrize("size", [(2, 48, 64), (2, 224, 224)])
def test_bf16x3_training_mode(gpu, golden, size):
    """TrainEngine(precision="bf16x3"): the 3x3 convolutions of forward, data gradient and weight gradient on split-bf16 MFMA.
    Operands carry 16 mantissa bits, so pre-activations differ from fp32 by ~1e-5 instead of ~1e-7 and ReLU masks flip about a
    hundred times as often (module docstring): the loss terms agree to 1e-4 and the gradients sit in the flip-noise class
    (torch fp32 itself sits at 1e-3 .. 1e-2 against fp64 at 224 x 224; measured here: whole vector 9e-3, median tensor 7e-3) --
    bounds: whole vector 3e-2, median tensor 2e-2, worst tensor 1e-1.  A fast mode for training runs, not the parity mode."""
    from oracle import train as otrain
    from popnet_amd.train import TrainEngine
    B, H, W = size
    sd = init_like_state_dict(golden.keys["rtpose_light3d"], seed=2)
    batch = [torch.from_numpy(a) for a in train_case_inputs(seed=300 + H, B=B, H=H, W=W)]
    r = otrain.train_step(sd, *batch, apply=False)
    r64 = otrain.train_step(_f64(sd), *[b.double() for b in batch], apply=False, dtype=torch.float64)
    eng = TrainEngine(sd, device=gpu, precision="bf16x3")
    terms = eng.forward_backward(*[t.to(gpu) for t in batch]).cpu().numpy()
    assert np.allclose(terms, r["terms"], rtol=1e-4, atol=0), (terms, r["terms"])
    hip, t32 = _accuracy_class(eng, r["grads"], r64["grads"])
    assert hip[0] <= 2e-2 and hip[2] <= 3e-2 and hip[1] <= 1e-1, (hip, t32)
    with pytest.raises(ValueError):
        TrainEngine(sd, device=gpu, precision="fp16")


@pytest.mark.parametrize("size", [(3, 72, 40), (1, 104, 136)])
def test_fp32_planes_engine_at_ragged_sizes_vs_oracle_and_nchw_engine(gpu, golden, size):
    """precision="fp32" on the planes engine (one fp32 plane per tensor, generic fp32 inference kernel, K = 4 weight gradient) at ragged map sizes: loss terms to
    1e-5, gradients in torch fp32's own accuracy class against fp64 autograd, and against the NCHW fp32 engine (two exact-fp32 evaluations in other summation orders)."""
    from oracle import train as otrain
    from popnet_amd.train import TrainEngine
    B, H, W = size
    sd = init_like_state_dict(golden.keys["rtpose_light3d"], seed=6)
    batch = [torch.from_numpy(a) for a in train_case_inputs(seed=700 + H, B=B, H=H, W=W)]
    r = otrain.train_step(sd, *batch, apply=False)
    r64 = otrain.train_step(_f64(sd), *[b.double() for b in batch], apply=False, dtype=torch.float64)
    eng = TrainEngine(sd, device=gpu, precision="fp32")
    assert eng.planes
    terms = eng.forward_backward(*[t.to(gpu) for t in batch]).cpu().numpy()
    assert np.allclose(terms, r["terms"], rtol=1e-5, atol=0), (terms, r["terms"])
    _assert_same_class(*_accuracy_class(eng, r["grads"], r64["grads"]))
    old = TrainEngine(sd, device=gpu, precision="fp32-nchw")
    terms_old = old.forward_backward(*[t.to(gpu) for t in batch]).cpu().numpy()
    assert np.allclose(terms, terms_old, rtol=2e-6, atol=0)
    num = float((eng.flat_g.double() - old.flat_g.double()).norm()), float(old.flat_g.double().norm())
    assert num[0] <= 3e-2 * num[1], num                  # room for a few mask flips at these small sizes (1.6e-2 measured at 3 x 72 x 40); flip-free: ~3e-6
    for k in eng.stats:
        assert _rel(eng.stats[k], old.stats[k]) < 2e-6, k


@pytest.mark.parametrize("size", [(3, 72, 40), (1, 104, 136)])
def test_planes_engine_at_ragged_sizes_vs_oracle_and_nchw_engine(gpu, golden, size):
    """The round-6 planes engine (csrc/trainx.hip: NHWC [hi | lo] bf16 planes, inference kernels for forward and data gradient, transposed-LDS-read
    weight gradient) at map sizes that leave ragged strips, half-empty tiles and odd pooled maps (36x20 -> 18x10 -> 9x5; 52x68 -> 26x34 -> 13x17): the loss
    terms against the CPU oracle to 1e-4, the gradients in the flip-noise class against fp64 autograd -- and the same class against the NCHW engine of
    rounds 2-5 on the same batch (two split-bf16 evaluations with different rounding points)."""
    from oracle import train as otrain
    from popnet_amd.train import TrainEngine
    B, H, W = size
    sd = init_like_state_dict(golden.keys["rtpose_light3d"], seed=6)
    batch = [torch.from_numpy(a) for a in train_case_inputs(seed=700 + H, B=B, H=H, W=W)]
    r = otrain.train_step(sd, *batch, apply=False)
    r64 = otrain.train_step(_f64(sd), *[b.double() for b in batch], apply=False, dtype=torch.float64)
    eng = TrainEngine(sd, device=gpu, precision="bf16x3")
    assert eng.planes
    terms = eng.forward_backward(*[t.to(gpu) for t in batch]).cpu().numpy()
    assert np.allclose(terms, r["terms"], rtol=1e-4, atol=0), (terms, r["terms"])
    hip, t32 = _accuracy_class(eng, r["grads"], r64["grads"])
    assert hip[0] <= 2e-2 and hip[2] <= 3e-2 and hip[1] <= 1e-1, (hip, t32)
    old = TrainEngine(sd, device=gpu, precision="bf16x3-nchw")
    terms_old = old.forward_backward(*[t.to(gpu) for t in batch]).cpu().numpy()
    assert np.allclose(terms, terms_old, rtol=2e-5, atol=0)
    num = float((eng.flat_g.double() - old.flat_g.double()).norm()), float(old.flat_g.double().norm())
    assert num[0] <= 3e-2 * num[1], num
    for k in eng.stats:                                  # BatchNorm running statistics: same batch statistics to fp32 rounding
        assert _rel(eng.stats[k], old.stats[k]) < 2e-5, k
    # a bias in front of a train-mode BatchNorm has an identically zero gradient: the planes engine leaves exact zeros, autograd leaves rounding noise
    assert float(eng.g["model1_1.0.bias"].abs().max()) == 0.0 and float(r["grads"]["model1_1.0.bias"].abs().max()) < 1e-6


@pytest.mark.parametrize("prec", ["bf16x3", "fp32"])
def test_planes_engine_weight_gradient_kernel_and_stream_split_are_exact_restatements(gpu, golden, monkeypatch, prec):
    """(1) trainx_wgrad.h (pixel-K MFMA GEMM through ds_read_b64_tr_b16) against train.hip's NCHW weight-gradient kernels fed the SAME planes (POPNET_TRAINX_WGRAD=legacy:
    the operands are handed over as fp32 = hi + lo, which re-splits to the same hi / lo): same products, another summation order -- every convolution weight gradient
    within 2e-5 of the other; (2) the two-stream schedule (weight gradients beside the BatchNorm / data-gradient chain) against the one-stream one: bit-identical;
    (3) the stem writing / reading planes directly against the NCHW f32 hand-over tensors of the round's first builds (POPNET_TRAINX_STEM_HANDOVER=1), its BatchNorm
    backward applied inside the weight-gradient kernel against bn_bwd_apply_kernel + a stored dC0 (POPNET_TRAINX_STEM_BN=separate), one / four chunks in flight
    against two (POPNET_TRAINX_STEM_DEPTH): bit-identical;
    (4) the weight packs built per (row, 8 channels) against the per-group gather kernel (POPNET_TRAINX_PACK=gather): the same packs, so bit-identical."""
    from popnet_amd.train import TrainEngine
    sd = init_like_state_dict(golden.keys["rtpose_light3d"], seed=7)
    batch = [torch.from_numpy(a).to(gpu) for a in train_case_inputs(seed=910, B=2, H=96, W=64)]

    def run(**env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        e = TrainEngine(sd, device=gpu, precision=prec)
        t = e.forward_backward(*batch).clone()
        torch.cuda.synchronize()
        for k in env:
            monkeypatch.delenv(k)
        return e, t
    two, t2 = run()
    one, t1 = run(POPNET_TRAINX_STREAMS="1")
    assert torch.equal(t1, t2) and torch.equal(one.flat_g, two.flat_g)
    hand, th = run(POPNET_TRAINX_STEM_HANDOVER="1")
    assert torch.equal(th, t2) and torch.equal(hand.flat_g, two.flat_g)
    assert float(two.g["model0.conv1.weight"].abs().max()) > 0
    for k in two.stats:
        assert torch.equal(hand.stats[k], two.stats[k]), k
    for env in ({"POPNET_TRAINX_STEM_FWD": "gather"}, {"POPNET_TRAINX_STEM_BN": "separate"}, {"POPNET_TRAINX_STEM_DEPTH": "1"}, {"POPNET_TRAINX_STEM_BN": "separate", "POPNET_TRAINX_STEM_DEPTH": "4"}):
        alt, ta = run(**env)
        assert torch.equal(ta, t2) and torch.equal(alt.flat_g, two.flat_g), env
    gat, tg = run(POPNET_TRAINX_PACK="gather")
    assert torch.equal(tg, t2) and torch.equal(gat.flat_g, two.flat_g)
    leg, tl = run(POPNET_TRAINX_WGRAD="legacy")
    assert torch.equal(tl, t2)
    worst = 0.0
    for k in two.g:
        if k.endswith(".weight") and two.g[k].dim() == 4 and k != "model0.conv1.weight":
            worst = max(worst, _rel(two.g[k], leg.g[k]))
    assert 0 < worst < 2e-5, worst
    assert torch.equal(two.g["model0.conv1.weight"], leg.g["model0.conv1.weight"])        # the stem keeps train.hip's kernel in both


@pytest.mark.parametrize("prec", ["bf16x3", "fp32"])
def test_planes_engine_restatements_where_the_stage_levels_run_conv4_and_the_stem_tiles_are_ragged(gpu, golden, monkeypatch, prec):
    """The exact-restatement switches once more at a batch shape the small cases do not reach: 16 frames of 328 x 488 -- 41 x 61 stage maps (>= 448 conv4 blocks per level in
    bf16x3: the conv4 fragment order of the weight packs), 164 x 244 stem maps (8 x 16 stem tiles cut on both edges; 640 332 stem pixels = 625.3 slices of 1 024)."""
    from popnet_amd.train import TrainEngine
    sd = init_like_state_dict(golden.keys["rtpose_light3d"], seed=11)
    batch = [torch.from_numpy(a).to(gpu) for a in train_case_inputs(seed=912, B=16, H=328, W=488)]

    def run(**env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        e = TrainEngine(sd, device=gpu, precision=prec)
        t = e.forward_backward(*batch).clone()
        g = e.flat_g.clone()
        torch.cuda.synchronize()
        for k in env:
            monkeypatch.delenv(k)
        del e
        torch.cuda.empty_cache()
        return t, g
    t0, g0 = run()
    assert bool(torch.isfinite(g0).all()) and float(g0.abs().max()) > 0
    for env in ({"POPNET_TRAINX_PACK": "gather"}, {"POPNET_TRAINX_STEM_FWD": "gather", "POPNET_TRAINX_STEM_BN": "separate", "POPNET_TRAINX_STEM_DEPTH": "1"},
                {"POPNET_TRAINX_STEM_HANDOVER": "1", "POPNET_TRAINX_STREAMS": "1"}):
        t1, g1 = run(**env)
        assert torch.equal(t1, t0) and torch.equal(g1, g0), env


@pytest.mark.parametrize("B", [2, 5])
def test_training_step_at_network_input_size(gpu, golden, B):
    """224x224 (the training configuration's input) from the initial state: loss terms to 1e-5; gradients in torch fp32's own
    accuracy class against fp64 autograd (module docstring)."""
    from oracle import train as otrain
    from popnet_amd.train import TrainEngine
    rng = np.random.default_rng(8 + B)
    h = 28
    batch = [torch.from_numpy(a) for a in (rng.normal(0, 1, (B, 1, 224, 224)).astype(np.float32), rng.uniform(0, 1, (B, 16, h, h)).astype(np.float32),
                                           rng.uniform(-1, 1, (B, 28, h, h)).astype(np.float32), rng.uniform(-1.5, 1.5, (B, 15, h, h)).astype(np.float32),
                                           (rng.uniform(0, 1, (B, 15, h, h)) < 0.2).astype(np.float32))]
    sd = init_like_state_dict(golden.keys["rtpose_light3d"], seed=3)
    r = otrain.train_step(sd, *batch, apply=False)
    r64 = otrain.train_step(_f64(sd), *[b.double() for b in batch], apply=False, dtype=torch.float64)
    eng = TrainEngine(sd, device=gpu)
    terms = eng.forward_backward(*[t.to(gpu) for t in batch]).cpu().numpy()
    assert np.allclose(terms, r["terms"], rtol=1e-5, atol=0)
    _assert_same_class(*_accuracy_class(eng, r["grads"], r64["grads"]))


def test_two_replicas_average_like_dataparallel(gpu, golden):
    """Data parallel semantics: every replica normalises with ITS OWN batch statistics and the gradients are
    averaged -- two engines on the two halves of a batch, gradients averaged by hand (what the all-reduce + grad_scale of
    TrainEngine.apply does), against the oracle run the same way."""
    from oracle import train as otrain
    batch = [torch.from_numpy(a) for a in train_case_inputs(seed=33, B=4)]
    sd = state_dict_from_keys(golden.keys["rtpose_light3d"], seed=0)
    halves = [[b[:2].contiguous() for b in batch], [b[2:].contiguous() for b in batch]]
    refs = [otrain.train_step(sd, *hb, apply=False)["grads"] for hb in halves]
    ref = {k: (refs[0][k] + refs[1][k]) / 2 for k in refs[0]}
    engs = [_engine(golden, gpu), _engine(golden, gpu)]
    for e, hb in zip(engs, halves):
        e.forward_backward(*[t.to(gpu) for t in hb])
    engs[0].flat_g.add_(engs[1].flat_g)          # the all-reduce (sum)
    engs[0].world = 2                            # -> grad_scale 1/2 inside pn_sgd_nesterov; no process group needed for the check below
    before = engs[0].flat_p.clone()
    engs[0]._check(engs[0].L.pn_sgd_nesterov(engs[0].ctx.handle, _p(engs[0].flat_p), _p(engs[0].flat_g), _p(engs[0].flat_m), engs[0].flat_p.numel(), 1.0, 0.9, 0.0, 1, 0.5,
                                             None), "sgd")
    engs[0].flat_g.mul_(0.5)
    ref64 = [otrain.train_step(_f64(sd), *[b.double() for b in hb], apply=False, dtype=torch.float64)["grads"] for hb in halves]
    _assert_same_class(*_accuracy_class(engs[0], ref, {k: (ref64[0][k] + ref64[1][k]) / 2 for k in ref64[0]}))
    # first Nesterov step: p - lr (g + mu g) = p - 1.9 g
    assert _rel(engs[0].flat_p, before - 1.9 * engs[0].flat_g) < 1e-6


def test_train_script_on_a_fake_mpaug_dataset(gpu, tmp_path, capsys):
    """scripts/train_mpaug.py (the reference trainer's drop-in) end to end on a fake MP-3DHP training tree: five annotation
    sets, depth / mask / background .npy files -> composed batches and targets on the GPU -> training steps -> validation
    loss -> `best_pose.pth` in the reference's checkpoint format, which the inference engine loads."""
    import importlib.util
    import json
    import random
    from popnet_amd import synth
    from popnet_amd.pipeline import PoseEngine
    d = str(tmp_path)
    for sub in ("img", "seg", "bg"):
        os.makedirs(os.path.join(d, sub))
    rng = np.random.default_rng(7)
    H, W = 320, 240
    ann_files = []
    for ii in range(5):
        ann = {"intrinsics": {"fx": 504.1, "fy": 504.0, "cx": 231.7, "cy": 320.6}}
        for f in range(6):
            name = "s%d_%d.npy" % (ii, f)
            joints, depths = synth.planted_persons(rng, 1, size=224)
            j2 = joints[0] * [W / 224.0, H / 224.0]
            ann[name] = [{"2d_joints": j2.tolist(), "3d_joints": np.concatenate([j2, np.full((15, 1), depths[0])], 1).tolist()}]
            mask = np.zeros((H, W), dtype=np.uint8)
            mask[max(int(j2[:, 1].min()) - 10, 0):int(j2[:, 1].max()) + 10, max(int(j2[:, 0].min()) - 10, 0):int(j2[:, 0].max()) + 10] = 1
            np.save(os.path.join(d, "img", name), np.clip(rng.normal(depths[0], 0.1, (H, W)), 0.3, 5.9).astype(np.float16))
            np.save(os.path.join(d, "seg", name), mask)
        path = os.path.join(d, "ann%d.json" % ii)
        json.dump(ann, open(path, "w"))
        ann_files.append(path)
    bgs = {}
    for f in range(3):
        np.save(os.path.join(d, "bg", "bg%d.npy" % f), np.clip(rng.normal(4.5, 0.3, (H, W)), 0, 6).astype(np.float16))
        bgs[str(f)] = {"file_name": "bg%d.npy" % f}
    json.dump(bgs, open(os.path.join(d, "bg.json"), "w"))
    spec = importlib.util.spec_from_file_location("train_mpaug", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "train_mpaug.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    random.seed(1)
    best = mod.main(["--train-annotations"] + ann_files + ["--val-annotations"] + ann_files + ["--image-dir", os.path.join(d, "img"), "--bg-file", os.path.join(d, "bg.json"),
                    "--bg-dir", os.path.join(d, "bg"), "--seg-dir", os.path.join(d, "seg"), "--output-dir", os.path.join(d, "out"), "--batch-size", "3", "--lr", "0.05",
                    "--epochs", "3", "--print-freq", "1", "--seed", "1"])
    out = capsys.readouterr().out
    vals = [float(l.split("val loss")[1].split()[0]) for l in out.splitlines() if "val loss" in l]
    assert len(vals) == 3 and vals[-1] < vals[0] and abs(best - min(vals)) < 1e-4, out
    sd = torch.load(os.path.join(d, "out", "best_pose.pth"), map_location="cpu")
    assert all(k.startswith("module.") for k in sd) and len(sd) == 234
    eng = PoseEngine(precision="fp32", state_dict=sd, device=gpu, max_batch=2, w_org=W, h_org=H)
    recs = eng.predict(torch.from_numpy(np.stack([np.load(os.path.join(d, "img", "s0_0.npy")), np.load(os.path.join(d, "img", "s1_0.npy"))])).to(gpu))
    assert recs.shape[0] == 2
    # the sampler obeys the reference's control flow on this tree: one or two sources per item, masks are 0 / 1
    from popnet_amd import targets
    ts = targets.MPAugTrainSet(os.path.join(d, "img"), ann_files, os.path.join(d, "bg.json"), os.path.join(d, "bg"), os.path.join(d, "seg"), device=gpu)
    fd, fm, n_src, bg, k2, k3, npers = ts.batch([0, 1, 2, 3])
    assert fd.shape == (4, 2, H, W) and fm.dtype == torch.uint8 and int(fm.max()) == 1 and set(n_src.tolist()) <= {1, 2} and torch.equal(npers, n_src)


def test_captured_training_step_equals_eager(gpu, golden):
    """TrainEngine.capture: the whole step as one hipGraph replay -- same parameters, statistics and loss terms as the eager
    engine after the same five steps on changing batches (bit for bit: same kernels, same order)."""
    from popnet_amd.train import TrainEngine
    sd = init_like_state_dict(golden.keys["rtpose_light3d"], seed=4)
    batches = [[torch.from_numpy(a).to(gpu) for a in train_case_inputs(seed=500 + i, B=2, H=64, W=96)] for i in range(5)]
    eager, graph = TrainEngine(sd, device=gpu, lr=0.05, precision="fp32-nchw"), TrainEngine(sd, device=gpu, lr=0.05, precision="fp32-nchw")
    graph.capture(*batches[0])                       # two warm-up steps on batch 0 inside
    for _ in range(2):
        eager.step(*batches[0])
    for b in batches:
        te = eager.step(*b).clone()
        tg = graph.step(*b).clone()
        assert torch.equal(te, tg)
    torch.cuda.synchronize()
    assert torch.equal(eager.flat_p, graph.flat_p) and torch.equal(eager.flat_m, graph.flat_m)
    a, b = eager.state_dict(), graph.state_dict()
    assert all(torch.equal(a[k].cpu(), b[k].cpu()) for k in a) and int(b["model0.bn1.num_batches_tracked"]) == 7
    # another batch shape falls back to the eager path
    other = [torch.from_numpy(x).to(gpu) for x in train_case_inputs(seed=9, B=1, H=64, W=96)]
    assert torch.equal(eager.step(*other), graph.step(*other))
    # ... and a LARGER one (new activation buffers, the C-side scratch grows) must not free what the graph points to
    # (ADVICE r02): the replay at the captured shape right after still equals the eager engine bit for bit
    big = [torch.from_numpy(x).to(gpu) for x in train_case_inputs(seed=10, B=3, H=96, W=128)]
    assert torch.equal(eager.step(*big), graph.step(*big))
    for b in batches[:2]:
        assert torch.equal(eager.step(*b).clone(), graph.step(*b).clone())
    torch.cuda.synchronize()
    assert torch.equal(eager.flat_p, graph.flat_p) and torch.equal(eager.flat_m, graph.flat_m)


@pytest.mark.parametrize("prec", ["fp32-nchw", "bf16x3-nchw", "fp32", "bf16x3"])
def test_captured_step_survives_eager_steps_of_other_shapes(gpu, golden, prec):
    """a captured step graph bakes the weight-pack descriptor table (pointer, entry count, grid) into its wpack_all_kernel
    node; eager steps at OTHER resolutions after capture() add pack keys (another tile geometry = another (flip, x3) form of the same
    weights) and used to free and reallocate that table.  Now the table is append-only in a fixed block: the replay right after such
    steps still equals an eager engine bit for bit, in both precisions."""
    from popnet_amd.train import TrainEngine
    sd = init_like_state_dict(golden.keys["rtpose_light3d"], seed=5)
    mk = lambda seed, B, H, W: [torch.from_numpy(a).to(gpu) for a in train_case_inputs(seed=seed, B=B, H=H, W=W)]
    b0, b1 = mk(800, 2, 64, 96), mk(801, 2, 64, 96)
    eager, graph = TrainEngine(sd, device=gpu, lr=0.05, precision=prec), TrainEngine(sd, device=gpu, lr=0.05, precision=prec)
    graph.capture(*b0, graph=True)                   # (the planes engines stay eager by default: here their two-stream step is captured as well)
    assert graph._graph is not None
    for _ in range(2):
        eager.step(*b0)
    for shape in ((1, 224, 224), (3, 96, 128), (1, 40, 56), (2, 128, 64)):
        other = mk(900 + shape[1], *shape)
        assert torch.equal(eager.step(*other).clone(), graph.step(*other).clone()), shape
        for b in (b0, b1):                           # replays of the captured shape in between
            assert torch.equal(eager.step(*b).clone(), graph.step(*b).clone()), shape
    torch.cuda.synchronize()
    assert torch.equal(eager.flat_p, graph.flat_p) and torch.equal(eager.flat_m, graph.flat_m)


@pytest.mark.parametrize("prec", ["fp32-nchw", "bf16x3-nchw"])
def test_weight_pack_cache_equals_per_call_packs(gpu, golden, prec):
    """Round 4: the packed weights of the 3x3 training convolutions are cached in the engine's context and refreshed by
    ONE launch per step (pn_train_pack_refresh) instead of one pack launch per convolution call.  Same pack arithmetic: an engine with the
    cache and one without (pn_train_pack_cache(ctx, 0)) hold bit-identical parameters, momentum and statistics after four steps -- also when
    a weight is changed BEHIND the optimiser's back between two steps (the refresh at the start of a step repacks everything)."""
    from popnet_amd.train import TrainEngine
    sd = init_like_state_dict(golden.keys["rtpose_light3d"], seed=8)
    batches = [[torch.from_numpy(a).to(gpu) for a in train_case_inputs(seed=700 + i, B=2, H=64, W=96)] for i in range(4)]
    cached, plain = TrainEngine(sd, device=gpu, lr=0.05, precision=prec), TrainEngine(sd, device=gpu, lr=0.05, precision=prec)
    plain.ctx.check(plain.L.pn_train_pack_cache(plain.ctx.handle, 0), "pn_train_pack_cache")
    for i, b in enumerate(batches):
        if i == 2:                                    # an out-of-band edit of a 3x3 weight: both engines must see it in the next step
            for e in (cached, plain):
                e.p["model1_1.3.weight"].mul_(1.25)
        assert torch.equal(cached.step(*b).clone(), plain.step(*b).clone()), i
    torch.cuda.synchronize()
    assert torch.equal(cached.flat_p, plain.flat_p) and torch.equal(cached.flat_m, plain.flat_m)
    a, b2 = cached.state_dict(), plain.state_dict()
    assert all(torch.equal(a[k].cpu(), b2[k].cpu()) for k in a)


def test_synthetic_train_eval_script_runs(gpu, capsys, monkeypatch):
    """scripts/synthetic_train_eval.py (train on synthetic scenes -> checkpoint -> inference engines -> metrics) at a toy length:
    the plumbing between compositor, target rasteriser, captured training step, state_dict, PoseEngine and metrics holds and the
    loss falls.  (The full run -- 6 000 steps, PCKh-2D 0.96 -- is profiles/r02_synthetic_train_eval_*.json.)"""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("synthetic_train_eval", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "synthetic_train_eval.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr("sys.argv", ["synthetic_train_eval.py", "--steps", "60", "--pool", "4", "--eval-frames", "32", "--batch", "8"])
    mod.main()
    out = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    loss = [v for _, v in out["train"]["loss"]]
    assert loss[-1] < 0.25 * loss[0] and np.isfinite(loss).all()
    for p in ("fp32", "bf16x3", "bf16"):
        assert out["eval"][p]["frames"] == 32 and out["eval"][p]["overflow_frames"] == 0
    assert out["eval"]["bf16x3"]["vs_fp32"]["same_person_count"] >= 30


def test_trained_checkpoint_bf16x3_equals_fp32_on_every_held_out_frame_and_bf16_is_what_the_docs_say(gpu, capsys, monkeypatch):
    """the flip rate of the tolerance-meeting fast mode is ZERO where it should be -- on the maps of a
    TRAINED checkpoint, whose peaks stand clear of the detection threshold.  scripts/synthetic_train_eval.py at 1 500 steps
    (17 s of training on the GPU, hipGraph replay of TrainEngine in bf16x3 mode): synthetic stick-figure scenes -> compositor ->
    target rasteriser -> training -> checkpoint -> PoseEngine in fp32 and bf16x3 on 96 HELD-OUT frames.  Every frame: same person
    count, same peak list, same person -> peak assignment; 3D joints within 1e-3 m (measured 2.1e-5); and the run learns the task
    (PCKh-2D > 0.85 against the planted ground truth with popnet_amd.metrics, the reference's protocol)."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("synthetic_train_eval", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "synthetic_train_eval.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    # the checkpoint is trained in the PARITY mode (exact fp32, the reference's arithmetic; 24 s on the planes engine): the subject here is what the INFERENCE modes do
    # on a trained net.  (The 1 500-step synthetic schedule at lr 0.2 x momentum 0.9 is sensitive to 1e-7 perturbations of the gradients: of fourteen round-6 runs
    # that differed only in summation order -- engines, split-K slice counts, seeds -- two ended in a net that also emits dozens of weak spurious peaks, on which no
    # 16-bit mode can follow fp32 frame for frame; profiles/r06_notes.txt section 5.)
    monkeypatch.setattr("sys.argv", ["synthetic_train_eval.py", "--steps", "1500", "--precision", "fp32"])
    mod.main()
    out = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    ev = out["eval"]
    assert ev["fp32"]["persons_found"] <= 1.25 * ev["fp32"]["persons_planted"], ev["fp32"]          # a clean net: no forest of spurious detections
    assert ev["fp32"]["frames"] == 96 and ev["fp32"]["overflow_frames"] == 0 and ev["fp32"]["pckh_2d_mean"] > 0.85, ev["fp32"]
    v = ev["bf16x3"]["vs_fp32"]
    assert v["same_person_count"] == 96 and v["same_assignment"] == 96 and v["d3_m_max"] < 1e-3, v
    assert ev["bf16x3"]["persons_found"] == ev["fp32"]["persons_found"] and ev["bf16x3"]["pckh_2d_mean"] == ev["fp32"]["pckh_2d_mean"]
    # the HEADLINE dtype on the same trained checkpoint, pinned to the measured figures (the run is deterministic:
    # two runs on two boxes gave identical numbers) -- same person count in 96 / 96 held-out frames, same assignment in 94 / 96, median 3D
    # difference 0.37 mm, task accuracy PCKh-2D 0.9315 against 0.9283 for fp32.  What bf16 does NOT give is the 1e-3 m bound on every
    # joint (max 6.7 cm on one joint of one frame): it is the throughput mode, and bench.py prints its fidelity next to `value`.
    b = ev["bf16"]["vs_fp32"]
    assert b["same_person_count"] >= 90 and 75 <= b["same_assignment"] <= 96, b          # 94 / 91 / 82-94 on checkpoints of the NCHW, planes bf16x3 and planes fp32 engines
    # the checkpoint is a function of the training arithmetic: 3.708e-4 when the NCHW engine of rounds 2-5 trained it; 3.2e-4, 3.3e-4 and 4.6e-4 with three
    # builds of the round-6 planes engine (same tolerance class, other rounding points / split-K slice counts) -- pinned to the band they all sit in
    assert 2e-4 <= b["d3_m_median"] <= 1e-3, b
    assert abs(ev["bf16"]["pckh_2d_mean"] - ev["fp32"]["pckh_2d_mean"]) < 0.01 and ev["bf16"]["pckh_2d_mean"] > 0.85, (ev["bf16"], ev["fp32"])


def test_conv_primitives_random_shapes(gpu):
    """scripts/train_conv_fuzz.py at test length: 40 random (N, Cin, Cout, H, W, kernel, padding) cases -- ragged channel
    counts, maps narrower and wider than a tile, one-row maps -- forward, data gradient, weight and bias gradient against torch
    in both precision modes (a 150-case run: worst relative error 1.5e-6 in fp32, 4.7e-6 in bf16x3)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("train_conv_fuzz", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "train_conv_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    worst = mod.run(40, seed=7, verbose=False)
    assert worst["fp32"] < 2e-5 and worst["bf16x3"] < 2e-4
