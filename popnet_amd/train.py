"""One training step of rtpose_light3d on the GPU (BASELINE configs[4]).

Host-side mirror of the per-batch body of the reference trainer
  third_party_methods/train_rtpose_light3d_kdh3d_mpaug.py:160-180 (CR)   model(img) -> rtpose_light3d_loss_fgweight -> backward -> SGD
  third_party_methods/lib/network/rtpose_light3d.py:326-356              the module in train mode (BatchNorm on batch statistics)
  third_party_methods/lib/network/losses.py:65-106                       the loss
  torch.optim.SGD(lr 1.0, momentum 0.9, nesterov=True)                   train_rtpose_light3d_kdh3d_mpaug.py:313-316 (CR)
Every arithmetic step is a HIP kernel of csrc/train.hip behind the C ABI (pn_conv2d_forward / _dgrad / _wgrad,
pn_bn_train_forward / _backward, pn_avgpool3s2_*, pn_head_forward / _backward, pn_sgd_nesterov); this module owns the
tensors (parameters, gradients and momentum in ONE flat buffer each, activations kept for the backward pass) and the call
order autograd would produce.  PyTorch is used for device memory and the data-parallel all-reduce only; there is no CPU
fallback and no autograd.  Data parallel = the reference's DataParallel semantics: per-replica BatchNorm statistics,
gradients averaged over replicas -- one all-reduce of the flat 22 MB gradient buffer (RCCL over xGMI).
"""
import ctypes as C

import torch

from . import _lib

ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2
BN_MOMENTUM, BN_EPS = 0.1, 1e-5
HEADS = (("paf", 28, 1, False), ("heat", 16, 0, False), ("z", 15, 1, True))     # name, channels, kind, fg-weighted
LOSS_NAMES = ["l1_paf", "l1_heat", "l1_z", "l2_paf", "l2_heat", "l2_z"]


def _is_stat(k):
    return k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked")


class TrainEngine:
    """state_dict: reference-format (optionally `module.`-prefixed) rtpose_light3d(15, 14, 2, input_dim=1) checkpoint."""

    def __init__(self, state_dict, device="cuda:0", lr=1.0, momentum=0.9, weight_decay=0.0, process_group=None, world_size=1, precision="fp32"):
        """precision: "fp32" (every product an exact fp32 FMA chain on v_mfma_f32_16x16x4_f32: the parity mode) or "bf16x3" (split-bf16 MFMA: 16
        significant bits per operand, fp32 accumulate; 2.5x faster) -- both on the round-6 planes engine (csrc/trainx.hip); "fp32-nchw" / "bf16x3-nchw":
        the NCHW engine of rounds 2-5 (csrc/train.hip)."""
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.PopnetError("popnet_amd.train: a ROCm device is required -- the HIP path has no CPU fallback")
        self.L = _lib.lib()
        self.ctx = _lib.Context(self.device.index or 0)            # private: its own scratch and precision switch
        if precision not in ("fp32", "bf16x3", "fp32-nchw", "bf16x3-nchw"):
            raise ValueError("precision must be 'fp32', 'bf16x3', 'fp32-nchw' or 'bf16x3-nchw', got %r" % (precision,))
        self.precision = precision
        # "bf16x3" (round 6): the whole step on NHWC [hi | lo] bf16 planes -- csrc/trainx.hip, one C++ object per batch shape, forward and
        # data-gradient convolutions on the inference kernels, pixel-K MFMA weight gradient.  "bf16x3-nchw": the round 2-5 form (fp32 NCHW
        # tensors, only the 3x3 convolutions split) -- kept for the autograd wrappers' kernels and as a cross-check.
        # "fp32" (round 6 as well): the same engine on ONE fp32 plane per tensor -- every product an exact fp32 FMA chain (the generic fp32 inference kernel
        # for forward / data gradient, a K = 4 pixel weight gradient).  "fp32-nchw": the NCHW engine in exact fp32 (what the autograd wrappers call).
        self.planes = precision in ("bf16x3", "fp32")
        self._trainers = {}
        self.ctx.check(self.L.pn_train_set_precision(self.ctx.handle, _lib.PN_PREC_BF16X3 if precision.startswith("bf16x3") else 0), "pn_train_set_precision")
        # packed conv weights cached in the (private) context and refreshed by ONE launch at the start of every step (forward_backward)
        self.ctx.check(self.L.pn_train_pack_cache(self.ctx.handle, 1), "pn_train_pack_cache")
        self.lr, self.momentum, self.weight_decay = float(lr), float(momentum), float(weight_decay)
        self.group, self.world = process_group, int(world_size)
        sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state_dict.items()}
        self.extra = {k: v.clone() for k, v in sd.items() if k.startswith("model0.layer3")}       # carried, never trained
        names = [k for k in sd if not _is_stat(k) and not k.startswith("model0.layer3")]
        sizes = [sd[k].numel() for k in names]
        self.n_params = sum(sizes)
        # every tensor starts on a 16-byte boundary of the flat buffers (the kernels take their 16-byte paths only on aligned
        # pointers); the few padding floats are zero in all three buffers and stay zero under SGD and the all-reduce
        offs, total = [], 0
        for n in sizes:
            offs.append(total)
            total += (n + 3) // 4 * 4
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=self.device)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=self.device)
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=self.device)
        self.p, self.g = {}, {}
        self._offsets = dict(zip(names, zip(offs, sizes)))
        for k, n, off in zip(names, sizes, offs):
            self.p[k] = self.flat_p[off:off + n].view(sd[k].shape)
            self.g[k] = self.flat_g[off:off + n].view(sd[k].shape)
            self.p[k].copy_(sd[k].to(torch.float32))
        self.stats = {k: v.to(self.device, torch.float32).clone() for k, v in sd.items() if k.endswith("running_mean") or k.endswith("running_var")}
        self.tracked = {k: int(v) for k, v in sd.items() if k.endswith("num_batches_tracked")}
        self.steps = 0
        self.A = {}            # activations of the current step
        self._bufs = {}        # name -> tensor, reused across steps while the shape stays
        self.loss_terms = torch.zeros(6, dtype=torch.float32, device=self.device)
        self._graph, self._static = None, None

    @classmethod
    def from_module(cls, module, **kw):
        """From a popnet_amd.network.rtpose_light3d.rtpose_light3d (or the reference's own module): its state_dict is copied;
        module.load_state_dict(engine.state_dict()) hands the trained weights back to the inference path."""
        return cls(module.state_dict(), **kw)

    def _trainer(self, N, H, W):
        """The pn_trainer of this batch shape (plans the step and allocates every activation / gradient tensor once; a step at another
        shape gets its own, so a hipGraph captured for the first keeps pointing at live buffers)."""
        tr = self._trainers.get((N, H, W))
        if tr is None:
            tr = self.L.pn_trainer_create(self.ctx.handle)
            if not tr:
                raise _lib.PopnetError("pn_trainer_create failed")
            self._check(self.L.pn_trainer_set_precision(tr, _lib.PN_PREC_F32 if self.precision == "fp32" else _lib.PN_PREC_BF16X3), "pn_trainer_set_precision")
            for k, (off, n) in self._offsets.items():
                self._check(self.L.pn_trainer_set_param(tr, k.encode(), off, n), "pn_trainer_set_param")
            for k, v in self.stats.items():
                self._check(self.L.pn_trainer_set_stat(tr, k.encode(), self._ptr(v)), "pn_trainer_set_stat")
            self._check(self.L.pn_trainer_finalize(tr, self._ptr(self.flat_p), self._ptr(self.flat_g), N, H, W, BN_MOMENTUM, BN_EPS), "pn_trainer_finalize")
            self._trainers[(N, H, W)] = tr
        return tr

    def __del__(self):
        try:
            for tr in getattr(self, "_trainers", {}).values():
                self.L.pn_trainer_destroy(tr)
            self._trainers = {}
        except Exception:
            pass

    # ---- plumbing ----
    def _s(self):
        return _lib.current_stream_ptr(self.device)

    def _buf(self, name, shape, zero=False):
        # keyed by (name, shape): a step at another batch shape gets its own buffers and never frees the ones a captured
        # graph of the first shape still points to
        key = (name, tuple(shape))
        t = self._bufs.get(key)
        if t is None:
            t = self._bufs[key] = torch.empty(shape, dtype=torch.float32, device=self.device)
        if zero:
            t.zero_()
        return t

    @staticmethod
    def _ptr(t):
        return C.c_void_p(t.data_ptr()) if t is not None else None

    def _check(self, rc, what):
        self.ctx.check(rc, what)

    # ---- primitives (forward records what backward needs in self.A) ----
    def _conv(self, name, x, ks, stride=1, pad=0):
        w, b = self.p[name + ".weight"], self.p.get(name + ".bias")
        N, Cin, H, W = x.shape
        Cout = w.shape[0]
        Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
        y = self._buf("c:" + name, (N, Cout, Ho, Wo))
        self._check(self.L.pn_conv2d_forward(self.ctx.handle, self._ptr(x), self._ptr(w), self._ptr(b), self._ptr(y), N, Cin, H, W, Cout, ks, stride, pad, 0, self._s()),
                    "pn_conv2d_forward")
        self.A["x:" + name] = x
        return y

    def _conv_bwd(self, name, dy, ks, stride=1, pad=0, dx=None, accumulate=False, need_dx=True):
        """weight / bias gradients of conv `name`; input gradient into dx (allocated when None) unless need_dx is False."""
        x = self.A["x:" + name]
        w = self.p[name + ".weight"]
        N, Cin, H, W = x.shape
        Cout = w.shape[0]
        db = self.g.get(name + ".bias")
        self._check(self.L.pn_conv2d_wgrad(self.ctx.handle, self._ptr(x), self._ptr(dy), self._ptr(self.g[name + ".weight"]), self._ptr(db), N, Cin, H, W, Cout, ks,
                                           stride, pad, self._s()), "pn_conv2d_wgrad")
        if not need_dx:
            return None
        if stride != 1:
            raise _lib.PopnetError("popnet_amd.train: data gradient of a strided convolution is not needed by rtpose_light3d and not built")
        if dx is None:
            dx = self._buf("dx:" + name, x.shape)
        self._check(self.L.pn_conv2d_dgrad(self.ctx.handle, self._ptr(dy), self._ptr(w), self._ptr(dx), N, Cin, H, W, Cout, ks, pad, 1 if accumulate else 0, self._s()),
                    "pn_conv2d_dgrad")
        return dx

    def _bn(self, name, x, act, res=None):
        N, Cc, H, W = x.shape
        y = self._buf("a:" + name, x.shape)
        mean, invstd = self._buf("m:" + name, (Cc,)), self._buf("i:" + name, (Cc,))
        self._check(self.L.pn_bn_train_forward(self.ctx.handle, self._ptr(x), self._ptr(self.p[name + ".weight"]), self._ptr(self.p[name + ".bias"]), self._ptr(res),
                                               self._ptr(y), self._ptr(mean), self._ptr(invstd), self._ptr(self.stats[name + ".running_mean"]),
                                               self._ptr(self.stats[name + ".running_var"]), BN_MOMENTUM, BN_EPS, act, N, Cc, H * W, self._s()), "pn_bn_train_forward")
        self.A["bn:" + name] = (x, y if res is not None else None, mean, invstd, act)      # without a residual the backward recomputes the mask from x
        return y

    def _bn_bwd(self, name, dy, dres=None, dres_accumulate=False):
        x, y, mean, invstd, act = self.A["bn:" + name]
        N, Cc, H, W = x.shape
        dx = self._buf("dc:" + name, x.shape)
        self._check(self.L.pn_bn_train_backward(self.ctx.handle, self._ptr(x), self._ptr(dy), self._ptr(y), self._ptr(self.p[name + ".weight"]), self._ptr(self.p[name + ".bias"]), self._ptr(mean),
                                                self._ptr(invstd), act, N, Cc, H * W, self._ptr(dx), self._ptr(self.g[name + ".weight"]),
                                                self._ptr(self.g[name + ".bias"]), self._ptr(dres), 1 if dres_accumulate else 0, self._s()), "pn_bn_train_backward")
        return dx

    def _pool(self, name, x):
        N, Cc, H, W = x.shape
        y = self._buf("p:" + name, (N, Cc, (H - 1) // 2 + 1, (W - 1) // 2 + 1))
        self._check(self.L.pn_avgpool3s2_forward(self.ctx.handle, self._ptr(x), self._ptr(y), N * Cc, H, W, self._s()), "pn_avgpool3s2_forward")
        self.A["pool:" + name] = x.shape
        return y

    def _pool_bwd(self, name, dy):
        N, Cc, H, W = self.A["pool:" + name]
        dx = self._buf("dp:" + name, (N, Cc, H, W))
        self._check(self.L.pn_avgpool3s2_backward(self.ctx.handle, self._ptr(dy), self._ptr(dx), N * Cc, H, W, self._s()), "pn_avgpool3s2_backward")
        return dx

    # ---- composite modules ----
    def _block(self, p, x):
        """BasicBlock (rtpose_light3d.py:36-72)"""
        a1 = self._bn(p + ".bn1", self._conv(p + ".conv1", x, 3, 1, 1), ACT_RELU)
        c2 = self._conv(p + ".conv2", a1, 3, 1, 1)
        idn = x
        if (p + ".downsample.0.weight") in self.p:
            idn = self._bn(p + ".downsample.1", self._conv(p + ".downsample.0", x, 1, 1, 0), ACT_NONE)
        return self._bn(p + ".bn2", c2, ACT_RELU, res=idn)

    def _block_bwd(self, p, dout):
        x = self.A["x:" + p + ".conv1"]
        dx = self._buf("dx:" + p, x.shape)
        if (p + ".downsample.0.weight") in self.p:
            didn = self._buf("di:" + p, dout.shape)
            dc2 = self._bn_bwd(p + ".bn2", dout, dres=didn)
            dcd = self._bn_bwd(p + ".downsample.1", didn)
            self._conv_bwd(p + ".downsample.0", dcd, 1, 1, 0, dx=dx, accumulate=False)
        else:
            dc2 = self._bn_bwd(p + ".bn2", dout, dres=dx)              # identity path: dx = g
        da1 = self._conv_bwd(p + ".conv2", dc2, 3, 1, 1)
        dc1 = self._bn_bwd(p + ".bn1", da1)
        self._conv_bwd(p + ".conv1", dc1, 3, 1, 1, dx=dx, accumulate=True)
        return dx

    def _stage(self, p, x):
        """make_stages Sequential (rtpose_light3d.py:222-246): conv BN LeakyReLU(0.1) at 0, 3, 6, 9; bare conv at 12"""
        for i in (0, 3, 6, 9):
            ks = self.p["%s.%d.weight" % (p, i)].shape[-1]
            x = self._bn("%s.%d" % (p, i + 1), self._conv("%s.%d" % (p, i), x, ks, 1, ks // 2), ACT_LEAKY)
        ks = self.p[p + ".12.weight"].shape[-1]
        return self._conv(p + ".12", x, ks, 1, ks // 2)

    def _stage_bwd(self, p, dv, dx, accumulate):
        ks = self.p[p + ".12.weight"].shape[-1]
        d = self._conv_bwd(p + ".12", dv, ks, 1, ks // 2)
        for i in (9, 6, 3, 0):
            dc = self._bn_bwd("%s.%d" % (p, i + 1), d)
            ks = self.p["%s.%d.weight" % (p, i)].shape[-1]
            if i:
                d = self._conv_bwd("%s.%d" % (p, i), dc, ks, 1, ks // 2)
            else:
                self._conv_bwd("%s.%d" % (p, i), dc, ks, 1, ks // 2, dx=dx, accumulate=accumulate)

    # ---- the step ----
    def forward_backward(self, img, heat_gt, paf_gt, z_gt, fg_mask):
        """Fills self.flat_g (this replica's gradient of the total loss) and self.loss_terms [6]; updates the BN running statistics."""
        for t, n in ((img, "img"), (heat_gt, "heat_gt"), (paf_gt, "paf_gt"), (z_gt, "z_gt"), (fg_mask, "fg_mask")):
            _lib.require_cuda_tensor(t, n)
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise _lib.PopnetError("popnet_amd.train: %s must be a contiguous float32 tensor" % n)
        N, _, H, W = img.shape
        if H % 8 or W % 8:
            raise _lib.PopnetError("popnet_amd.train: input size must be a multiple of 8")
        h, w = H // 8, W // 8
        targets = {"paf": paf_gt, "heat": heat_gt, "z": z_gt}
        for (name, ch, _, _) in HEADS:
            if tuple(targets[name].shape) != (N, ch, h, w):
                raise _lib.PopnetError("popnet_amd.train: %s target must be [%d, %d, %d, %d]" % (name, N, ch, h, w))
        if tuple(fg_mask.shape) != (N, 15, h, w):
            raise _lib.PopnetError("popnet_amd.train: fg_mask must be [%d, 15, %d, %d]" % (N, h, w))
        self.A = {}
        L, ctx, s = self.L, self.ctx.handle, self._s()
        if self.planes:
            self._check(L.pn_trainer_forward_backward(self._trainer(N, H, W), self._ptr(img), self._ptr(heat_gt), self._ptr(paf_gt), self._ptr(z_gt), self._ptr(fg_mask),
                                                      self._ptr(self.loss_terms), s), "pn_trainer_forward_backward")
            for k in self.tracked:
                self.tracked[k] += 1
            return self.loss_terms
        self._check(L.pn_train_pack_refresh(ctx, s), "pn_train_pack_refresh")       # every cached weight pack, one launch (whatever changed the weights)
        # forward (rtpose_light3d.py:206-219, 328-354)
        a = self._bn("model0.bn1", self._conv("model0.conv1", img, 7, 2, 3), ACT_RELU)
        a = self._block("model0.layer1.1", self._block("model0.layer1.0", a))
        a = self._pool("1", a)
        a = self._block("model0.layer2.0", a)
        a = self._bn("model0.bn2", self._conv("model0.conv2", a, 1, 1, 0), ACT_RELU)
        feat = self._pool("2", a)
        cat = self._buf("cat", (N, 187, h, w))
        hw = h * w
        self._check(L.pn_slice_copy(ctx, self._ptr(feat), 128, C.c_void_p(cat.data_ptr() + 59 * hw * 4), 187, N, 128, hw, 0, s), "pn_slice_copy")
        sig = {}
        for stage in (1, 2):
            src = feat if stage == 1 else cat
            c0 = 0
            for b, (name, ch, kind, weighted) in enumerate(HEADS):
                v = self._stage("model%d_%d" % (stage, b + 1), src)
                sg = sig[(stage, name)] = self._buf("s:%d%s" % (stage, name), (N, ch, h, w))
                if stage == 1:
                    out, ld = C.c_void_p(cat.data_ptr() + c0 * hw * 4), 187
                else:
                    o = self._buf("o:" + name, (N, ch, h, w))
                    out, ld = self._ptr(o), ch
                self._check(L.pn_head_forward(ctx, self._ptr(v), self._ptr(targets[name]), self._ptr(fg_mask) if weighted else None, kind, N, ch, hw, self._ptr(sg), out, ld,
                                              C.c_void_p(self.loss_terms.data_ptr() + 4 * (3 * (stage - 1) + b)), s), "pn_head_forward")
                c0 += ch
        # backward
        dcat = self._buf("dcat", (N, 187, h, w))
        for b, (name, ch, kind, weighted) in enumerate(HEADS):
            dv = self._buf("dv:" + name, (N, ch, h, w))
            self._check(L.pn_head_backward(ctx, self._ptr(sig[(2, name)]), self._ptr(targets[name]), self._ptr(fg_mask) if weighted else None, None, ch, kind, N, ch, hw,
                                           self._ptr(dv), s), "pn_head_backward")
            self._stage_bwd("model2_%d" % (b + 1), dv, dcat, accumulate=b > 0)
        dfeat = self._buf("dfeat", (N, 128, h, w))
        self._check(L.pn_slice_copy(ctx, C.c_void_p(dcat.data_ptr() + 59 * hw * 4), 187, self._ptr(dfeat), 128, N, 128, hw, 0, s), "pn_slice_copy")
        c0 = 0
        for b, (name, ch, kind, weighted) in enumerate(HEADS):
            dv = self._buf("dv:" + name, (N, ch, h, w))
            self._check(L.pn_head_backward(ctx, self._ptr(sig[(1, name)]), self._ptr(targets[name]), self._ptr(fg_mask) if weighted else None,
                                           C.c_void_p(dcat.data_ptr() + c0 * hw * 4), 187, kind, N, ch, hw, self._ptr(dv), s), "pn_head_backward")
            self._stage_bwd("model1_%d" % (b + 1), dv, dfeat, accumulate=True)
            c0 += ch
        d = self._pool_bwd("2", dfeat)
        d = self._conv_bwd("model0.conv2", self._bn_bwd("model0.bn2", d), 1, 1, 0)
        d = self._block_bwd("model0.layer2.0", d)
        d = self._pool_bwd("1", d)
        d = self._block_bwd("model0.layer1.0", self._block_bwd("model0.layer1.1", d))
        self._conv_bwd("model0.conv1", self._bn_bwd("model0.bn1", d), 7, 2, 3, need_dx=False)
        for k in self.tracked:
            self.tracked[k] += 1
        return self.loss_terms

    @staticmethod
    def reduce_flat_gradient(flat_g, world, group=None, force=False):
        """The data-parallel exchange of a step: ONE all-reduce (sum) of the flat gradient buffer over the replicas; the 1 / world
        of DataParallel's mean is applied inside pn_sgd_nesterov (grad_scale).  Returns that scale.  (RCCL on the GPUs; the
        world_size-2 gloo test drives exactly this function on CPU tensors.)  force: run the collective at world 1 as well
        (bench.py --force-dist: puts RCCL on the hardware of a one-GPU box)."""
        if world > 1 or force:
            import torch.distributed as dist
            dist.all_reduce(flat_g, group=group)
        return 1.0 / world

    def apply(self):
        """Average the gradient over the replicas (one all-reduce of the flat buffer) and take the Nesterov SGD step."""
        if self.world > 1:
            self.reduce_flat_gradient(self.flat_g, self.world, self.group)
        self._check(self.L.pn_sgd_nesterov(self.ctx.handle, self._ptr(self.flat_p), self._ptr(self.flat_g), self._ptr(self.flat_m), self.flat_p.numel(), self.lr, self.momentum,
                                           self.weight_decay, 1 if self.steps == 0 else 0, 1.0 / self.world, self._s()), "pn_sgd_nesterov")
        self.steps += 1

    def step(self, img, heat_gt, paf_gt, z_gt, fg_mask):
        """-> device tensor [6] of loss terms (LOSS_NAMES); the total loss of losses.py:65-90 is their sum.  Asynchronous.
        After capture() a batch of the captured shape replays the hipGraph instead of issuing ~430 launches from Python."""
        if self._graph is not None and tuple(img.shape) == tuple(self._static[0].shape):
            for dst, src in zip(self._static, (img, heat_gt, paf_gt, z_gt, fg_mask)):
                dst.copy_(src, non_blocking=True)
            self._graph.replay()
            for k in self.tracked:
                self.tracked[k] += 1
            if self.world > 1:                       # the exchange and the update stay outside the graph
                self.apply()
            else:
                self.steps += 1
            return self.loss_terms
        terms = self.forward_backward(img, heat_gt, paf_gt, z_gt, fg_mask)
        self.apply()
        return terms

    def capture(self, img, heat_gt, paf_gt, z_gt, fg_mask, warmup_steps=2, graph=None):
        """graph=None: a hipGraph for the NCHW engines ("fp32", "bf16x3-nchw": ~430 launches issued from Python per eager step); the planes engine
        ("bf16x3") stays EAGER -- its step is ONE C call that issues every launch from C++ and runs the weight gradients on a second HIP stream beside
        the BatchNorm / data-gradient chain; replayed as a hipGraph that fork / join structure loses its overlap (7.31 ms against 6.97 ms eager, same
        box, round 6) -- only the warm-up steps run.  graph=True / False forces either form.

        Captures one step for this batch shape in a hipGraph (forward, loss, backward and -- single GPU -- the SGD update;
        with world > 1 the gradient all-reduce and the update stay eager behind the graph).  Runs two eager steps first: every
        buffer and the C-side scratch reach their final size, the momentum buffers exist (the graph bakes in first_step = 0).
        The learning rate is baked in too: call capture() again after changing `lr`.  warmup_steps = 0 captures without
        executing anything (an engine that has already stepped on this batch shape: the training loop's case)."""
        batch = (img, heat_gt, paf_gt, z_gt, fg_mask)
        self._graph = None
        if warmup_steps == 0 and self.steps == 0:
            raise _lib.PopnetError("popnet_amd.train: capture(warmup_steps=0) needs an engine that has already taken a step")
        for _ in range(warmup_steps):
            self.step(*batch)
        torch.cuda.synchronize(self.device)
        if graph is None:
            graph = not self.planes
        if not graph:
            return self
        # the pack cache's descriptor table is (re)built by an eager refresh -- never under a capture: bring it up to date now (a step run
        # before this call may have added entries after its own refresh)
        self._check(self.L.pn_train_pack_refresh(self.ctx.handle, self._s()), "pn_train_pack_refresh")
        torch.cuda.synchronize(self.device)
        self._static = [t.clone() for t in batch]
        # the graph will point into the context's scratch: a later, larger eager step must retire that block, not free it
        self._check(self.L.pn_train_ws_keep(self.ctx.handle, 1), "pn_train_ws_keep")
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                self.forward_backward(*self._static)
                if self.world == 1:
                    self._check(self.L.pn_sgd_nesterov(self.ctx.handle, self._ptr(self.flat_p), self._ptr(self.flat_g), self._ptr(self.flat_m), self.flat_p.numel(), self.lr,
                                                       self.momentum, self.weight_decay, 0, 1.0, self._s()), "pn_sgd_nesterov")
        torch.cuda.current_stream(self.device).wait_stream(side)
        for k in self.tracked:                       # forward_backward counted the captured (not executed) pass
            self.tracked[k] -= 1
        self._graph = g
        return self

    def state_dict(self, prefix=""):
        """Reference-format checkpoint (train_rtpose_light3d_kdh3d_mpaug.py:337 saves the DataParallel one: prefix='module.')."""
        out = {}
        for k, v in self.p.items():
            out[prefix + k] = v.detach().clone()
        for k, v in self.stats.items():
            out[prefix + k] = v.detach().clone()
        for k, v in self.tracked.items():
            out[prefix + k] = torch.tensor(v, dtype=torch.long)
        for k, v in self.extra.items():
            out[prefix + k] = v.clone()
        return out
