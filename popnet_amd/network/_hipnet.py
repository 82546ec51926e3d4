"""Shared machinery of the two network modules: turns an ``nn.Module`` that merely HOLDS the
reference-format parameters into a compiled ``pn_net`` (BN folded, weights packed for MFMA) and
runs forward through the C ABI.  No PyTorch convolution is ever called."""
import ctypes as C
import os

import numpy as np
import torch
import torch.nn as nn

from .. import _lib

_PREC = {"fp32": _lib.PN_PREC_F32, "f32": _lib.PN_PREC_F32, "bf16": _lib.PN_PREC_BF16, "bf16x3": _lib.PN_PREC_BF16X3}


def default_precision():
    return os.environ.get("POPNET_PRECISION", "fp32")


class HipNetModule(nn.Module):
    """Base class.  Subclasses define the parameter-holding sub-modules (same names as the
    reference, so ``state_dict()`` / ``load_state_dict()`` are interchangeable) and ``_kind``.

    precision: "fp32" (parity mode: fp32 storage + fp32-input MFMA) or "bf16" (bf16 storage,
    bf16 MFMA, fp32 accumulate).  Default: $POPNET_PRECISION or "fp32".
    """
    _kind = None

    def __init__(self):
        super().__init__()
        self.precision = default_precision()
        self._net = None          # (handle, key)
        self._ctx = None
        self._pins = {}           # pn_net handle -> number of engines that hold it locked (a captured hipGraph refers to it)
        self._retired = []        # pinned handles this module no longer uses: destroyed when their last pin goes

    # ---- compilation ----------------------------------------------------------------------
    def _net_args(self):
        raise NotImplementedError

    def _weights_version(self):
        """(data_ptr, _version) of every parameter and buffer: changes when torch code re-assigns or writes a tensor in place.  The list of
        (owner, name) slots is collected once (the module tree of these parameter holders is static), so the per-forward cost is one pass over
        ~200 dictionary lookups, not a walk of the module tree; a tensor that is re-assigned or re-created (.cuda() / .to()) is
        looked up afresh every time.
        Writers that go through raw device pointers (the HIP training primitives updating BatchNorm running statistics, a TrainEngine
        writing into module buffers) do NOT bump `_version`: they must call invalidate() -- `_forward_train` does."""
        cache = self.__dict__.get("_version_tensors")
        if cache is not None:
            # structural check: a replaced sub-module (model.model0.conv1 = nn.Conv2d(...)), a slot registered later
            # (register_buffer / register_parameter / add_module) or a slot count that changed makes the cached slot list stale
            for d, n in cache[1]:
                if len(d) != n:
                    cache = None
                    break
            else:
                for d, k, ident in cache[2]:
                    if id(d.get(k)) != ident:
                        cache = None
                        break
        if cache is None:                      # (owner dict, name) of every parameter / buffer slot: a RE-ASSIGNED tensor is still seen
            slots, sizes, kids = [], [], []
            for mod in self.modules():
                slots += [(mod._parameters, k) for k in mod._parameters]
                slots += [(mod._buffers, k) for k in mod._buffers]
                sizes += [(mod._parameters, len(mod._parameters)), (mod._buffers, len(mod._buffers)), (mod._modules, len(mod._modules))]
                kids += [(mod._modules, k, id(c)) for k, c in mod._modules.items()]
            cache = (slots, sizes, kids)
            self.__dict__["_version_tensors"] = cache
        out = []
        for d, k in cache[0]:
            t = d.get(k)
            out.append((0, 0) if t is None else (t.data_ptr(), t._version))      # a slot that is (or became) None is a value like any other
        return tuple(out)

    def _apply(self, fn, *a, **kw):
        self.__dict__.pop("_version_tensors", None)
        return super()._apply(fn, *a, **kw)

    def _release(self):
        """Drops the compiled net.  A handle an engine holds locked (PoseEngine.lock / StreamingEngine.capture: captured hipGraphs
        point at its device buffers and launch descriptors) is NOT destroyed here -- it is retired and freed by the unlock, so a
        recompile under a captured engine can never turn a graph replay into a use-after-free."""
        if self._net is not None:
            h = self._net[0]
            if self._pins.get(h, 0) > 0:
                self._retired.append(h)
            else:
                _lib.lib().pn_net_destroy(h)
            self._net = None

    def _pin(self, handle):
        self._pins[handle] = self._pins.get(handle, 0) + 1

    def _unpin(self, handle):
        n = self._pins.get(handle, 0) - 1
        if n > 0:
            self._pins[handle] = n
            return
        self._pins.pop(handle, None)
        if handle in self._retired:
            self._retired.remove(handle)
            _lib.lib().pn_net_destroy(handle)

    def __del__(self):
        try:
            self._pins = {}
            for h in self._retired:
                _lib.lib().pn_net_destroy(h)
            self._retired = []
            self._release()
        except Exception:
            pass

    def invalidate(self):
        """Forces re-folding / re-packing of the weights on the next forward."""
        self._release()

    def load_state_dict(self, state_dict, strict=True, **kw):
        # reference checkpoints come from a DataParallel wrapper ("module." prefix,
        # tpm/evaluate/evaluation_rtpose_light3d_kdh3d_mpreal_ablation.py:136-139)
        if state_dict and all(k.startswith("module.") for k in state_dict):
            state_dict = type(state_dict)((k[len("module."):], v) for k, v in state_dict.items())
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self.__dict__.pop("_version_tensors", None)
        self.invalidate()
        return out

    def _compile(self, device, batch, in_h, in_w):
        if self.training:
            raise _lib.PopnetError("popnet_amd: the compiled (BN-folded, MFMA-packed) net is the eval-mode inference path -- call .eval(); in train "
                                   "mode rtpose_light3d.forward runs the autograd-wrapped training primitives (network/_autograd.py), and "
                                   "popnet_amd.train.TrainEngine.from_module(module).step(...) is the fast path for whole training steps")
        prec = _PREC.get(str(self.precision).lower())
        if prec is None:
            raise ValueError("precision must be 'fp32', 'bf16' or 'bf16x3', got %r" % (self.precision,))
        key = (device.index, prec, in_h, in_w, self._weights_version())
        if self._net is not None and self._net[1] == key and self._net[2] >= batch:
            return self._net[0]
        self._release()
        L = _lib.lib()
        ctx = _lib.Context.for_device(device.index)
        self._ctx = ctx
        kind, num_parts, a, input_dim = self._net_args()
        h = L.pn_net_create(ctx.handle, kind, num_parts, a, input_dim)
        if not h:
            raise _lib.PopnetError("pn_net_create failed: " + ctx.last_error())
        try:
            for name, t in self.state_dict().items():
                if name.endswith("num_batches_tracked") or name.startswith("model0.layer3."):
                    continue      # never executed by the reference forward (yolo_posenet.py:42,54)
                arr = np.ascontiguousarray(t.detach().to("cpu", torch.float32).numpy())
                shape = (C.c_int64 * arr.ndim)(*arr.shape)
                ctx.check(L.pn_net_set_tensor(h, name.encode(), arr.ctypes.data_as(C.c_void_p), shape, arr.ndim),
                          "pn_net_set_tensor(%s)" % name)
            max_batch = max(batch, int(os.environ.get("POPNET_MAX_BATCH", "0")))
            ctx.check(L.pn_net_finalize(h, prec, max_batch, in_h, in_w), "pn_net_finalize")
        except Exception:
            L.pn_net_destroy(h)
            raise
        self._net = (h, key, max_batch)
        return h

    def flops_per_frame(self):
        if self._net is None:
            raise _lib.PopnetError("net not compiled yet (run a forward first)")
        return _lib.lib().pn_net_flops_per_frame(self._net[0])

    def _check_input(self, x):
        _lib.require_cuda_tensor(x, "input")
        cin = int(getattr(self, "input_dim", 1))
        if x.dim() != 4 or x.shape[1] != cin:
            raise _lib.PopnetError("expected a [B,%d,H,W] batch (input_dim = %d), got %s" % (cin, cin, tuple(x.shape)))
        return x.contiguous().float()

    def _activation(self, name, B, shape, device):
        out = torch.empty((B,) + shape, device=device, dtype=torch.float32)
        self._ctx.check(_lib.lib().pn_net_copy_activation(self._net[0], name.encode(), B, C.c_void_p(out.data_ptr()),
                                                          _lib.current_stream_ptr(device)), "pn_net_copy_activation")
        return out


def bn_(c):
    return nn.BatchNorm2d(c)


def conv_(cin, cout, k, stride=1, bias=False):
    return nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=k // 2, bias=bias)
