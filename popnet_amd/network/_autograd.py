"""torch.autograd.Function wrappers over the training primitives of the C ABI (csrc/train.hip).

They close the training side of the drop-in boundary: with these, a module in train mode,
``model(img)`` -> the reference's loss -> ``loss.backward()`` -> ``torch.optim.SGD.step()`` run exactly as in
tpm/train_rtpose_light3d_kdh3d_mpaug.py:160-180 (CR) with the imports swapped and nothing else changed -- autograd only
ORDERS the calls; every convolution, BatchNorm and pooling, forward and backward, is one of

    pn_conv2d_forward / pn_conv2d_dgrad / pn_conv2d_wgrad        nn.Conv2d
    pn_bn_train_forward / pn_bn_train_backward                   nn.BatchNorm2d in train mode (+ residual add + ReLU / LeakyReLU)
    pn_avgpool3s2_forward / pn_avgpool3s2_backward               nn.AvgPool2d(3, 2, 1)

(the same kernels popnet_amd.train.TrainEngine drives without autograd; the engine stays the fast path: flat buffers, fused
head + loss kernels, one hipGraph per step).  No PyTorch convolution / normalisation kernel is ever run; there is no CPU fallback.
"""
import ctypes as C

import torch

from .. import _lib

ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _ctx(t):
    _lib.require_cuda_tensor(t, "input")
    if t.dtype != torch.float32:
        raise _lib.PopnetError("popnet_amd: the training path computes in float32 (got %s)" % t.dtype)
    return _lib.Context.for_device(t.device.index), _lib.current_stream_ptr(t.device)


class Conv2dFn(torch.autograd.Function):
    """y = conv2d(x, w, b, stride, padding)   x [N, Cin, H, W], w [Cout, Cin, k, k] (k in 1, 3, 7), NCHW float32."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad):
        x, w = x.contiguous(), w.contiguous()
        c, s = _ctx(x)
        N, Cin, H, W = x.shape
        Cout, ks = w.shape[0], w.shape[-1]
        Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
        y = torch.empty((N, Cout, Ho, Wo), device=x.device, dtype=torch.float32)
        c.check(_lib.lib().pn_conv2d_forward(c.handle, _ptr(x), _ptr(w), _ptr(b), _ptr(y), N, Cin, H, W, Cout, ks, stride, pad, 0, s), "pn_conv2d_forward")
        ctx.save_for_backward(x, w)
        ctx.geom = (stride, pad, b is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, pad, has_bias = ctx.geom
        dy = dy.contiguous()
        c, s = _ctx(dy)
        N, Cin, H, W = x.shape
        Cout, ks = w.shape[0], w.shape[-1]
        dw = torch.empty_like(w)
        db = torch.empty((Cout,), device=x.device, dtype=torch.float32) if has_bias else None
        c.check(_lib.lib().pn_conv2d_wgrad(c.handle, _ptr(x), _ptr(dy), _ptr(dw), _ptr(db), N, Cin, H, W, Cout, ks, stride, pad, s), "pn_conv2d_wgrad")
        dx = None
        if ctx.needs_input_grad[0]:
            if stride != 1:
                raise _lib.PopnetError("popnet_amd: the data gradient of a strided convolution is not built (rtpose_light3d's only strided "
                                       "convolution is the first layer, whose input needs no gradient)")
            dx = torch.empty_like(x)
            c.check(_lib.lib().pn_conv2d_dgrad(c.handle, _ptr(dy), _ptr(w), _ptr(dx), N, Cin, H, W, Cout, ks, pad, 0, s), "pn_conv2d_dgrad")
        return dx, dw, db, None, None


class BatchNormActFn(torch.autograd.Function):
    """y = act(batch_norm(x; batch statistics) [+ res]); updates running_mean / running_var in place like nn.BatchNorm2d.train()
    (momentum 0.1, unbiased running variance).  One primitive for `bn -> (+ identity) -> relu` of a BasicBlock and for
    `bn -> LeakyReLU(0.1)` of a stage layer."""

    @staticmethod
    def forward(ctx, x, gamma, beta, res, running_mean, running_var, act, momentum, eps):
        x = x.contiguous()
        res = res.contiguous() if res is not None else None
        c, s = _ctx(x)
        N, Cc, H, W = x.shape
        y = torch.empty_like(x)
        mean = torch.empty((Cc,), device=x.device, dtype=torch.float32)
        invstd = torch.empty((Cc,), device=x.device, dtype=torch.float32)
        c.check(_lib.lib().pn_bn_train_forward(c.handle, _ptr(x), _ptr(gamma), _ptr(beta), _ptr(res), _ptr(y), _ptr(mean), _ptr(invstd), _ptr(running_mean),
                                               _ptr(running_var), float(momentum), float(eps), int(act), N, Cc, H * W, s), "pn_bn_train_forward")
        ctx.save_for_backward(x, y if res is not None else None, gamma, beta, mean, invstd)
        ctx.act, ctx.has_res = int(act), res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, beta, mean, invstd = ctx.saved_tensors
        dy = dy.contiguous()
        c, s = _ctx(dy)
        N, Cc, H, W = x.shape
        dx, dgamma, dbeta = torch.empty_like(x), torch.empty_like(gamma), torch.empty_like(beta)
        dres = torch.empty_like(x) if ctx.has_res else None
        c.check(_lib.lib().pn_bn_train_backward(c.handle, _ptr(x), _ptr(dy), _ptr(y), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(invstd), ctx.act, N, Cc, H * W,
                                                _ptr(dx), _ptr(dgamma), _ptr(dbeta), _ptr(dres), 0, s), "pn_bn_train_backward")
        return dx, dgamma, dbeta, dres, None, None, None, None, None


class AvgPool3s2Fn(torch.autograd.Function):
    """nn.AvgPool2d(kernel_size=3, stride=2, padding=1) (count_include_pad, as the reference leaves it)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        c, s = _ctx(x)
        N, Cc, H, W = x.shape
        y = torch.empty((N, Cc, (H - 1) // 2 + 1, (W - 1) // 2 + 1), device=x.device, dtype=torch.float32)
        c.check(_lib.lib().pn_avgpool3s2_forward(c.handle, _ptr(x), _ptr(y), N * Cc, H, W, s), "pn_avgpool3s2_forward")
        ctx.shape = tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        c, s = _ctx(dy)
        N, Cc, H, W = ctx.shape
        dx = torch.empty(ctx.shape, device=dy.device, dtype=torch.float32)
        c.check(_lib.lib().pn_avgpool3s2_backward(c.handle, _ptr(dy), _ptr(dx), N * Cc, H, W, s), "pn_avgpool3s2_backward")
        return dx


def conv(x, m):
    """nn.Conv2d parameter holder `m` applied through Conv2dFn.  The HIP primitive is the reference's convolution -- square stride and
    zero padding, no dilation, no groups (rtpose_light3d.py:24-34, 222-246) -- anything else is refused rather than silently ignored."""
    if (tuple(m.dilation) != (1, 1) or m.groups != 1 or m.stride[0] != m.stride[1] or isinstance(m.padding, str)
            or m.padding[0] != m.padding[1] or m.padding_mode != "zeros"):
        raise _lib.PopnetError("popnet_amd: Conv2d(dilation=%s, groups=%s, stride=%s, padding=%s, padding_mode=%r) is outside what pn_conv2d_forward "
                               "computes (square stride / zero padding, no dilation, no groups)" % (m.dilation, m.groups, m.stride, m.padding, m.padding_mode))
    return Conv2dFn.apply(x, m.weight, m.bias, m.stride[0], m.padding[0])


def bn_act(x, m, act, res=None):
    """nn.BatchNorm2d parameter holder `m` in train mode (+ residual, + activation); counts the batch like torch does."""
    if m.num_batches_tracked is not None:
        m.num_batches_tracked.add_(1)
    if m.momentum is None:          # torch: cumulative moving average, factor 1 / num_batches_tracked (nn.modules.batchnorm._BatchNorm.forward)
        # The counter lives on the device: reading it every call is a host sync per BatchNorm layer (and illegal under stream capture).
        # A host-side twin, seeded once from the buffer and re-seeded whenever the buffer object or its version moved behind our back
        # (load_state_dict, .to()), counts the calls instead.
        nb = m.num_batches_tracked
        if nb is None:
            momentum = 0.0
        else:
            twin = m.__dict__.get("_popnet_nbt")
            if twin is None or twin[0] != nb.data_ptr() or twin[1] != nb._version - 1:
                count = int(nb)                      # one sync: first call, or somebody else wrote the buffer
            else:
                count = twin[2] + 1
            m.__dict__["_popnet_nbt"] = (nb.data_ptr(), nb._version, count)
            momentum = 1.0 / float(count)
    else:
        momentum = m.momentum
    return BatchNormActFn.apply(x, m.weight, m.bias, res, m.running_mean, m.running_var, act, momentum, m.eps)


def avgpool(x):
    return AvgPool3s2Fn.apply(x)
