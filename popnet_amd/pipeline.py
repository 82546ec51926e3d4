"""Batched end-to-end engine: raw depth frames in, fixed-size pose records out, all on one GPU
stream with no host round trip in between.

    depth [B,H,W] f16/f32 (HBM) --pn_preprocess--> x [B,1,S,S] f32
        --pn_rtpose_forward--> paf / heat / z (f32 NCHW, 185 KB per frame)
        --pn_parse_paf--> pn_pose_frame[B]

This is the per-batch body of the reference's evaluation loop
(tpm/evaluate/evaluation_rtpose_light3d_kdh3d_mpreal_ablation.py:161-316) minus Python.
Multi-GPU: frames are independent, so ranks take disjoint frame shards, there is no
collective on the data path, and ONE all-gather of the fixed-size records (RCCL over xGMI via
torch.distributed, backend "nccl") assembles the result in global frame order.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, synth
from .config import DEPTH_MAX, DEPTH_MEAN, DEPTH_STD, INTRINSICS, default_cfg
from .network._hipnet import _PREC
from .network.rtpose_light3d import rtpose_light3d
from .utils.paf_to_pose import make_parse_cfg


import os as _os
_NO_FRAMES_IN = bool(_os.environ.get("POPNET_NO_FRAMES_IN"))     # experiment switch: pn_preprocess + pn_*_forward as two calls in every precision


class PoseEngine:
    def __init__(self, precision="bf16", state_dict=None, device=None, max_batch=32, input_size=224,
                 w_org=480, h_org=640, intrinsics=INTRINSICS, weight_seed=0, private_ctx=False, calib_gain=1.0):
        """private_ctx: give this engine its own pn_ctx (parse workspace), so that several engines can
        run concurrently on different HIP streams (bench.py pipelines consecutive batches that way).
        calib_gain: synthetic weights only -- logit gain of the heat head before calibrate_heads (1 = heat values crowd
        the detection threshold, the worst case for a reduced-precision forward; > 1 = peaks comfortably separated)."""
        if not torch.cuda.is_available():
            raise _lib.PopnetError("PoseEngine needs a GPU: the HIP path has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.max_batch, self.S = int(max_batch), int(input_size)
        self.model = rtpose_light3d(15, 14, 2, input_dim=1).eval()
        if state_dict is None:
            synth.load_synth_weights(self.model, seed=weight_seed)
            calibrate_heads(self.model, self.device, gain=calib_gain)
        else:
            self.model.load_state_dict(state_dict)
        self.model.precision = precision
        self.ctx = _lib.Context(self.device.index) if private_ctx else _lib.Context.for_device(self.device.index)
        self.L = _lib.lib()
        self.cfg = make_parse_cfg(default_cfg(), input_size=self.S, w_org=w_org, h_org=h_org, intrinsics=intrinsics,
                                  depth_mean=DEPTH_MEAN, depth_std=DEPTH_STD)
        # a PRIVATE context's parse scratch is sized once for max_batch and never moves again (a captured graph keeps the
        # pointer).  The shared per-device context keeps growing on demand: fixing it here would make an unrelated later
        # call with a larger batch (utils.paf_to_pose.parse_paf_batch) fail depending on construction order
        self.private_ctx = bool(private_ctx)
        if self.private_ctx:
            self.ctx.check(self.L.pn_parse_reserve(self.ctx.handle, self.max_batch), "pn_parse_reserve")
        self._locked = False
        self._locked_net = None
        h = self.S // 8
        d, f32 = self.device, torch.float32
        self.x = torch.empty((self.max_batch, 1, self.S, self.S), device=d, dtype=f32)
        self.paf = torch.empty((self.max_batch, 28, h, h), device=d, dtype=f32)
        self.heat = torch.empty((self.max_batch, 16, h, h), device=d, dtype=f32)
        self.z = torch.empty((self.max_batch, 15, h, h), device=d, dtype=f32)
        self.frames = torch.empty((self.max_batch, _lib.POSE_FRAME_DTYPE.itemsize), device=d, dtype=torch.uint8)
        self.flops_per_frame = self.L.pn_net_flops_per_frame(self.net)

    @property
    def net(self):
        """The model's CURRENT pn_net handle.  The module destroys its handle on invalidate() / load_state_dict() / a
        forward at another size or precision, so the engine never keeps a raw copy: it re-reads (or re-compiles) here."""
        # always through _compile: it compares the FULL key (device, precision, input size, weights version), so a net the
        # module compiled for another size / precision by a direct model(x) call is never run on this engine's buffers
        if self._locked:
            # a captured hipGraph refers to the pn_net this engine was locked on.  invalidate() / load_state_dict() / a train-mode
            # forward release the module's handle (the locked one is only retired, never freed: HipNetModule._release): refuse to
            # run -- and above all to silently compile a fresh net next to graphs that still replay the old one
            cur = self.model._net
            prec = _PREC.get(str(self.model.precision).lower())
            key = (self.device.index, prec, self.S, self.S, self.model._weights_version())
            if cur is None or cur[0] != self._locked_net or cur[1] != key or cur[2] < self.max_batch:
                raise _lib.PopnetError("PoseEngine: the module was invalidated, recompiled or modified while the engine is locked / captured "
                                       "(a captured hipGraph still refers to the pn_net of the lock): unlock -- lock(False) -- and capture again")
            return cur[0]
        return self.model._compile(self.device, self.max_batch, self.S, self.S)

    def lock(self, locked=True):
        """Freeze the net's launch descriptors (call after a warm-up forward, before capturing a hipGraph).  The locked handle
        is pinned in the module: nothing destroys it until lock(False)."""
        if locked:
            if self._locked:
                return
            net = self.net
            (self.model._ctx or self.ctx).check(self.L.pn_net_lock(net, 1), "pn_net_lock")
            self.model._pin(net)
            self._locked_net = net
            self._locked = True
        elif self._locked:
            net = self._locked_net
            self._locked = False
            self._locked_net = None
            (self.model._ctx or self.ctx).check(self.L.pn_net_lock(net, 0), "pn_net_lock")
            self.model._unpin(net)                                   # frees the handle if the module had retired it meanwhile

    # ---- stages (all asynchronous on the current stream) --------------------------------------
    def preprocess(self, depth):
        _lib.require_cuda_tensor(depth, "depth")
        if depth.dtype == torch.float16:
            dt = _lib.PN_DEPTH_F16
        elif depth.dtype == torch.float32:
            dt = _lib.PN_DEPTH_F32
        else:
            raise _lib.PopnetError("depth frames must be float16 or float32")
        B, H, W = depth.shape
        if B > self.max_batch:
            raise _lib.PopnetError("batch %d exceeds max_batch %d" % (B, self.max_batch))
        depth = depth.contiguous()
        self.ctx.check(self.L.pn_preprocess(self.ctx.handle, C.c_void_p(depth.data_ptr()), dt, B, H, W,
                                            C.c_void_p(self.x.data_ptr()), self.S, float(DEPTH_MAX), float(DEPTH_MEAN),
                                            float(DEPTH_STD), _lib.current_stream_ptr(self.device)), "pn_preprocess")
        return B

    def forward(self, B):
        net = self.net                                               # (the net reports errors through the context it was created on)
        (self.model._ctx or self.ctx).check(self.L.pn_rtpose_forward(net, C.c_void_p(self.x.data_ptr()), B, C.c_void_p(self.paf.data_ptr()),
                                                                     C.c_void_p(self.heat.data_ptr()), C.c_void_p(self.z.data_ptr()),
                                                                     _lib.current_stream_ptr(self.device)), "pn_rtpose_forward")

    def _frames_args(self, depth):
        _lib.require_cuda_tensor(depth, "depth")
        if depth.dtype not in (torch.float16, torch.float32):
            raise _lib.PopnetError("depth frames must be float16 or float32")
        B, H, W = depth.shape
        if B > self.max_batch:
            raise _lib.PopnetError("batch %d exceeds max_batch %d" % (B, self.max_batch))
        depth = depth.contiguous()                                   # (a copy, if one is made, is stream-ordered like every torch allocation)
        dt = _lib.PN_DEPTH_F16 if depth.dtype == torch.float16 else _lib.PN_DEPTH_F32
        return B, (C.c_void_p(depth.data_ptr()), dt, B, H, W, float(DEPTH_MAX), float(DEPTH_MEAN), float(DEPTH_STD))

    def forward_frames(self, depth):
        """preprocess + forward as ONE call on the raw frames (pn_rtpose_forward_frames: the stem resizes / clamps / normalises its own
        input tile; bit-identical maps, the pre-processed tensor is never written).  The fp32 parity mode keeps the two calls."""
        if _PREC.get(str(self.model.precision).lower()) == _lib.PN_PREC_F32 or _NO_FRAMES_IN:
            B = self.preprocess(depth)
            self.forward(B)
            return B
        B, args = self._frames_args(depth)
        net = self.net
        (self.model._ctx or self.ctx).check(self.L.pn_rtpose_forward_frames(net, *args, C.c_void_p(self.paf.data_ptr()), C.c_void_p(self.heat.data_ptr()),
                                                                            C.c_void_p(self.z.data_ptr()), _lib.current_stream_ptr(self.device)),
                                            "pn_rtpose_forward_frames")
        return B

    def parse(self, B, frames=None, wire=None):
        """wire: optional device uint8 tensor [>= B, sizeof(pn_pose_wire)] that receives the compact records in the same launch."""
        frames = self.frames if frames is None else frames
        h = self.S // 8
        self.ctx.check(self.L.pn_parse_paf_wire(self.ctx.handle, C.c_void_p(self.heat.data_ptr()), C.c_void_p(self.paf.data_ptr()),
                                                C.c_void_p(self.z.data_ptr()), B, h, h, C.byref(self.cfg), C.c_void_p(frames.data_ptr()),
                                                C.c_void_p(wire.data_ptr() if wire is not None else None),
                                                _lib.current_stream_ptr(self.device)), "pn_parse_paf_wire")

    def predict(self, depth, frames=None, wire=None):
        """depth [B,H,W] CUDA f16/f32 -> device uint8 tensor [B, sizeof(pn_pose_frame)] (no sync)."""
        B = self.forward_frames(depth)
        self.parse(B, frames, wire)
        return (self.frames if frames is None else frames)[:B]

    def predict_host(self, depth):
        return records_to_numpy(self.predict(depth))

    def predict_lists(self, depth):
        """depth [B,H,W] -> the reference's per-frame result lists (dataset.pose_records_to_lists), with NO capacity limit:
        a frame whose fixed-size record overflows (more than 32 peaks in a joint map or 32 persons) is parsed again from the
        maps of this batch by the unbounded second pass (pn_parse_paf_unbounded), so it comes back as the reference would
        return it instead of raising."""
        return self.lists_from_records(records_to_numpy(self.predict(depth)))

    def lists_from_records(self, recs):
        """Result lists of the batch whose maps are in self.heat / self.paf / self.z and whose records are `recs` (numpy)."""
        from .dataset import pose_records_to_lists
        from .utils.paf_to_pose import parse_paf_unbounded
        over = {i: parse_paf_unbounded(self.heat[i], self.paf[i], self.z[i], self.cfg) for i in range(len(recs)) if int(recs[i]["status"])}
        return pose_records_to_lists(recs, over)

    def pack(self, frames, wire=None):
        """pn_pose_frame records (device, [B, sizeof]) -> compact pn_pose_wire records (device uint8 [B, sizeof], no sync):
        the form that is gathered across GPUs (6.2 KB instead of 33 KB per frame)."""
        B = frames.shape[0]
        if wire is None:
            wire = torch.empty((B, _lib.POSE_WIRE_DTYPE.itemsize), device=frames.device, dtype=torch.uint8)
        self.ctx.check(self.L.pn_pack_pose_frames(self.ctx.handle, C.c_void_p(frames.data_ptr()), B, C.c_void_p(wire.data_ptr()),
                                                  _lib.current_stream_ptr(self.device)), "pn_pack_pose_frames")
        return wire[:B]


class YoloEngine:
    """Batched Yolo-Pose+ twin of PoseEngine: depth frames -> pn_preprocess -> pn_yolo_forward ->
    pn_parse_yolo (decode + box NMS + skeletons + the evaluation script's rescale / back-projection)
    -> pn_yolo_frame records.  Per-batch body of
    tpm/evaluate/evaluation_yolo_posenet_kdh3d_mpreal.py:137-217 minus Python."""

    def __init__(self, precision="bf16", state_dict=None, device=None, max_batch=32, input_size=224,
                 w_org=480, h_org=640, intrinsics=INTRINSICS, weight_seed=1, anchors=None,
                 conf_threshold=0.5, nms_threshold=0.5, private_ctx=False):
        from .config import YOLO_ANCHORS
        from .network.yolo_posenet import YoloPoseNet
        if not torch.cuda.is_available():
            raise _lib.PopnetError("YoloEngine needs a GPU: the HIP path has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.max_batch, self.S = int(max_batch), int(input_size)
        self.anchors = list(YOLO_ANCHORS if anchors is None else anchors)
        self.model = YoloPoseNet(15, input_dim=1, anchors=self.anchors).eval()
        if state_dict is None:
            synth.load_synth_weights(self.model, seed=weight_seed)
            calibrate_yolo_conf(self.model, self.device)
        else:
            self.model.load_state_dict(state_dict)
        self.model.precision = precision
        self.ctx = _lib.Context(self.device.index) if private_ctx else _lib.Context.for_device(self.device.index)
        self.L = _lib.lib()
        self.cfg = make_parse_cfg(default_cfg(), input_size=self.S, w_org=w_org, h_org=h_org, intrinsics=intrinsics,
                                  depth_mean=DEPTH_MEAN, depth_std=DEPTH_STD)
        self.conf_threshold, self.nms_threshold = float(conf_threshold), float(nms_threshold)
        flat = [float(v) for a in self.anchors for v in a]
        self._anchors_c = (C.c_float * len(flat))(*flat)
        d = self.device
        self.x = torch.empty((self.max_batch, 1, self.S, self.S), device=d, dtype=torch.float32)
        self.out = torch.empty((self.max_batch, len(self.anchors) * 50, self.S // 16, self.S // 16), device=d,
                               dtype=torch.float32)
        self.frames = torch.empty((self.max_batch, _lib.YOLO_FRAME_DTYPE.itemsize), device=d, dtype=torch.uint8)
        self._locked = False
        self._locked_net = None
        self.flops_per_frame = self.L.pn_net_flops_per_frame(self.net)

    preprocess = PoseEngine.preprocess
    net = PoseEngine.net
    lock = PoseEngine.lock

    def forward(self, B):
        net = self.net
        (self.model._ctx or self.ctx).check(self.L.pn_yolo_forward(net, C.c_void_p(self.x.data_ptr()), B, C.c_void_p(self.out.data_ptr()),
                                                                   _lib.current_stream_ptr(self.device)), "pn_yolo_forward")

    _frames_args = PoseEngine._frames_args

    def forward_frames(self, depth):
        """preprocess + forward as one call on the raw frames (pn_yolo_forward_frames); fp32 keeps the two calls."""
        if _PREC.get(str(self.model.precision).lower()) == _lib.PN_PREC_F32 or _NO_FRAMES_IN:
            B = self.preprocess(depth)
            self.forward(B)
            return B
        B, args = self._frames_args(depth)
        net = self.net
        (self.model._ctx or self.ctx).check(self.L.pn_yolo_forward_frames(net, *args, C.c_void_p(self.out.data_ptr()), _lib.current_stream_ptr(self.device)),
                                            "pn_yolo_forward_frames")
        return B

    def parse(self, B, frames=None):
        frames = self.frames if frames is None else frames
        h = self.S // 16
        self.ctx.check(self.L.pn_parse_yolo(self.ctx.handle, C.c_void_p(self.out.data_ptr()), B, h, h, self._anchors_c,
                                            len(self.anchors), 15, self.S, self.S, float(DEPTH_MEAN), float(DEPTH_STD),
                                            self.conf_threshold, self.nms_threshold, 0, C.byref(self.cfg),
                                            C.c_void_p(frames.data_ptr()), _lib.current_stream_ptr(self.device)),
                       "pn_parse_yolo")

    def predict(self, depth, frames=None):
        """depth [B,H,W] CUDA f16/f32 -> device uint8 tensor [B, sizeof(pn_yolo_frame)] (no sync)."""
        B = self.forward_frames(depth)
        self.parse(B, frames)
        return (self.frames if frames is None else frames)[:B]

    def predict_host(self, depth):
        return self.predict(depth).cpu().numpy().view(_lib.YOLO_FRAME_DTYPE).reshape(-1)


def calibrate_yolo_conf(model, device=None, frac=0.012, calib_frames=8, seed=99):
    """Synthetic-checkpoint calibration for YoloPoseNet (bench / smoke only; never applied to user
    weights).  Seeded random weights put every cell's confidence near sigmoid(0) = 0.5, i.e. ~200
    candidate boxes per frame.  The last conv has no bias, so the two confidence filters (channels 4
    and 54) are shifted by a constant -delta on every tap: delta is grid-searched on a seeded
    calibration batch (fp32 forward of the HIP path itself) so ~frac of the cells pass conf > 0.5."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    depth = torch.from_numpy(synth.synth_depth(calib_frames, seed=seed)).to(device)
    ctx = _lib.Context.for_device(device.index)
    x = torch.empty((calib_frames, 1, 224, 224), device=device, dtype=torch.float32)
    ctx.check(_lib.lib().pn_preprocess(ctx.handle, C.c_void_p(depth.data_ptr()), _lib.PN_DEPTH_F16, calib_frames,
                                       depth.shape[1], depth.shape[2], C.c_void_p(x.data_ptr()), 224, float(DEPTH_MAX),
                                       float(DEPTH_MEAN), float(DEPTH_STD), _lib.current_stream_ptr(device)), "pn_preprocess")
    prec = model.precision
    model.precision = "fp32"
    w0 = model.model2_4[0].weight.detach().clone()

    def frac_at(delta):
        with torch.no_grad():
            model.model2_4[0].weight.copy_(w0)
            model.model2_4[0].weight[[4, 54]] -= delta
        model.invalidate()
        out = model(x)
        return float((out[:, [4, 54]] > 0.5).float().mean())

    lo, hi = 0.0, 0.2
    for _ in range(12):                      # fraction above 0.5 falls as delta grows (inputs are mostly positive)
        mid = 0.5 * (lo + hi)
        if frac_at(mid) > frac:
            lo = mid
        else:
            hi = mid
    frac_at(hi)
    model.precision = prec
    model.invalidate()
    return model


class StreamingEngine:
    """Throughput front end: `depth` batches in flight on one GPU.

    Slot i owns a PoseEngine / YoloEngine (activations, a PRIVATE pn_ctx = its own parse scratch), `pool` static input
    buffers, a device and a pinned host record buffer, a HIP stream and -- once `capture()` has run -- one hipGraph of
    the whole step (pre-processing, forward, parsing, record D2H copy) PER INPUT BUFFER.  `submit(j)` replays slot
    (ticket mod depth) on input buffer j of that slot and returns at once; the latency-bound tail of a batch (head
    convolutions, pose assembly, the record copy) then overlaps with the convolutions of the next batches.  Fill
    `input(slot, j)` (e.g. with a non_blocking copy from pinned host memory issued on `stream(slot)`) before
    submitting; read `records(ticket)` / `host_records(ticket)` after `wait(ticket)`.  A slot must be waited for
    before it is submitted again.

    wire=True: what leaves the GPU per step is the compact pn_pose_wire form (6.2 KB instead of 33 KB per frame; PAF
    path only): `host_records()` then holds pn_pose_wire records, `records()` still the full pn_pose_frame ones.

    Every engine gets its own context and its net is locked after the warm-up (pn_net_lock): nothing a captured graph
    points at (parse scratch, launch descriptors) can be reallocated or rewritten by a later eager call."""

    def __init__(self, engine_cls=None, depth=3, frame_hw=(640, 480), frame_dtype=torch.float16, graph=True, pool=1,
                 wire=False, **engine_kw):
        engine_cls = PoseEngine if engine_cls is None else engine_cls
        self.depth = max(1, int(depth))
        self.pool = max(1, int(pool))
        engine_kw["private_ctx"] = True
        self.engines = [engine_cls(**engine_kw) for _ in range(self.depth)]
        e0 = self.engines[0]
        self.device, self.max_batch = e0.device, e0.max_batch
        item = e0.frames.shape[1]
        self.wire = bool(wire) and isinstance(e0, PoseEngine)
        hitem = _lib.POSE_WIRE_DTYPE.itemsize if self.wire else item
        self.inputs = [[torch.zeros((self.max_batch,) + tuple(frame_hw), device=self.device, dtype=frame_dtype)
                        for _ in range(self.pool)] for _ in range(self.depth)]
        self.recs = [torch.empty((self.max_batch, item), device=self.device, dtype=torch.uint8) for _ in range(self.depth)]
        self.wires = [torch.empty((self.max_batch, hitem), device=self.device, dtype=torch.uint8) if self.wire else None
                      for _ in range(self.depth)]
        self.host = [torch.empty((self.max_batch, hitem), dtype=torch.uint8, pin_memory=True) for _ in range(self.depth)]
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(self.depth)]
        self.events = [torch.cuda.Event() for _ in range(self.depth)]
        # host hand-over (submit_host): a copy stream per slot, "copied" / "no longer read" events per input buffer
        cs = torch.cuda.Stream(device=self.device)             # ONE copy stream: transfers leave in batch order, one at a time at link rate
        self.copy_streams = [cs for _ in range(self.depth)]
        self._copied = [[torch.cuda.Event() for _ in range(self.pool)] for _ in range(self.depth)]
        self._released = [[None] * self.pool for _ in range(self.depth)]
        self._next_buf = [0] * self.depth
        self._posted = []                                      # (slot, buffer) of the batches posted ahead by post_host, oldest first
        self.graphs = [[None] * self.pool for _ in range(self.depth)]
        self._want_graph = bool(graph)
        self._tickets = 0
        cur = torch.cuda.current_stream(self.device)
        for st in self.streams:
            st.wait_stream(cur)

    def input(self, slot, j=0):
        return self.inputs[slot % self.depth][j % self.pool]

    def stream(self, slot):
        return self.streams[slot % self.depth]

    def _body(self, s, j=0):
        e = self.engines[s]
        if self.wire:
            e.predict(self.inputs[s][j], self.recs[s], self.wires[s])      # full and compact records from the same read-out launch
            self.host[s].copy_(self.wires[s], non_blocking=True)
        else:
            e.predict(self.inputs[s][j], self.recs[s])
            self.host[s].copy_(self.recs[s], non_blocking=True)

    def capture(self):
        """Warm every slot eagerly, lock its net, then record its step as one hipGraph per input buffer (same kernels,
        same arguments)."""
        for s in range(self.depth):
            with torch.cuda.stream(self.streams[s]):
                self._body(s)
                self._body(s)
            torch.cuda.synchronize(self.device)
            self.engines[s].lock()
            if self._want_graph:
                for j in range(self.pool):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=self.streams[s]):
                        self._body(s, j)
                    self.graphs[s][j] = g
        torch.cuda.synchronize(self.device)

    def submit(self, j=0, eager=False):
        """Runs the next slot on input buffer j of that slot, on the slot's stream (asynchronous).  Returns the ticket."""
        t = self._tickets
        s = t % self.depth
        j %= self.pool
        with torch.cuda.stream(self.streams[s]):
            if self.graphs[s][j] is not None and not eager:
                self.graphs[s][j].replay()
            else:
                self._body(s, j)
            self.events[s].record(self.streams[s])
        self._tickets += 1
        return t

    def post_host(self, host_batch):
        """First half of the host hand-over: enqueues the PCIe transfer of a pinned host batch into the NEXT input buffer of the
        slot that will run it -- the slot of ticket `tickets + posted` -- on the copy stream, ordered only after the step that last
        read that buffer.  A streaming caller posts batch k + depth when it submits batch k, so a transfer is always in flight under
        the kernels of earlier batches (also across the start of a timed region: the frames of a live stream do not wait for the
        previous ones to finish).  At most depth * (pool - 1) batches may be posted ahead.  Returns the number now posted."""
        if len(self._posted) >= self.depth * max(1, self.pool - 1):
            raise _lib.PopnetError("StreamingEngine.post_host: %d batches already posted (depth %d, pool %d)" % (len(self._posted), self.depth, self.pool))
        s = (self._tickets + len(self._posted)) % self.depth
        j = self._next_buf[s]
        self._next_buf[s] = (j + 1) % self.pool
        cs = self.copy_streams[s]
        if self._released[s][j] is not None:
            cs.wait_event(self._released[s][j])
        with torch.cuda.stream(cs):
            self.inputs[s][j][:len(host_batch)].copy_(host_batch, non_blocking=True)
            self._copied[s][j].record(cs)
        self._posted.append((s, j))
        return len(self._posted)

    def submit_posted(self, eager=False):
        """Second half: runs the oldest posted batch on its slot once its transfer has landed.  Returns the ticket."""
        if not self._posted:
            raise _lib.PopnetError("StreamingEngine.submit_posted: nothing posted")
        s, j = self._posted.pop(0)
        assert s == self._tickets % self.depth
        self.streams[s].wait_event(self._copied[s][j])
        t = self.submit(j, eager)
        if self._released[s][j] is None:
            self._released[s][j] = torch.cuda.Event()
        self._released[s][j].record(self.streams[s])
        return t

    def drop_posted(self):
        """Forgets the batches posted but not submitted (their transfers still complete; the buffers are simply overwritten later)."""
        self._posted = []

    def submit_host(self, host_batch, eager=False):
        """Host hand-over: copies a pinned host batch into the slot's NEXT input buffer on the slot's copy stream -- ordered only
        after the step that last read that buffer, so with pool >= 2 the PCIe transfer of this slot's next batch runs under
        the kernels of its current one -- and runs the step once the copy has landed.  Returns the ticket."""
        if self._posted:
            raise _lib.PopnetError("StreamingEngine.submit_host: batches are posted ahead (post_host); use submit_posted")
        self.post_host(host_batch)
        return self.submit_posted(eager)

    def wait(self, ticket):
        self.events[ticket % self.depth].synchronize()

    def records(self, ticket):
        return self.recs[ticket % self.depth]

    def host_records(self, ticket):
        return self.host[ticket % self.depth]

    def join(self, stream=None):
        """Makes `stream` (default: the current one) wait for everything submitted so far; no host synchronisation."""
        stream = torch.cuda.current_stream(self.device) if stream is None else stream
        for st in self.streams:
            stream.wait_stream(st)


def wire_to_lists(wire):
    """pn_pose_wire records (numpy) -> the per-frame result-schema entries (float32 values widened to Python floats)."""
    out = {"human_pred_set_2d": [], "human_pred_set_3d": [], "human_pred_set_visibility": [], "human_pred_set_part_conf": []}
    for fr in wire:
        if int(fr["status"]):        # peak / person overflow, or more persons than the wire form carries: never truncate silently
            raise _lib.PopnetError("pose wire record overflow (status=%d, %d persons; the wire form carries %d): use the full pn_pose_frame records"
                                   % (int(fr["status"]), int(fr["n_persons"]), _lib.PN_WIRE_MAX_PERSONS))
        n = int(fr["n_persons"])
        v = fr["vals"][:n].astype(np.float64)
        out["human_pred_set_2d"].append(v[:, :, 0:2].tolist())
        out["human_pred_set_3d"].append(v[:, :, 2:5].tolist())
        out["human_pred_set_part_conf"].append(v[:, :, 5].tolist())
        out["human_pred_set_visibility"].append((fr["person_joint"][:n] >= 0).astype(int).tolist())
    return out


def records_to_numpy(frames_dev):
    return frames_dev.cpu().numpy().view(_lib.POSE_FRAME_DTYPE).reshape(-1)


def calibrate_heads(model, device=None, frac=0.004, calib_frames=8, seed=99, gain=1.0):
    """Synthetic-checkpoint calibration (bench / smoke only; never applied to user weights).

    With seeded random weights every heat map hovers around sigmoid(0) = 0.5 > THRESH_HEATMAP, i.e.
    hundreds of plateau peaks per joint -- an invalid parse workload (52 s/frame
    in the reference).  This shifts each stage-2 heat channel's bias so that only `frac` of its
    cells exceed the threshold on a seeded calibration batch (about three peaks per joint map, the
    density a trained network emits for a 2-3 person frame), so the end-to-end bench runs NMS,
    refinement, limb scoring and assembly on a realistic number of candidates with nothing skipped.
    The statistics come from one fp32 forward of the HIP path itself."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    depth = torch.from_numpy(synth.synth_depth(calib_frames, seed=seed)).to(device)
    ctx = _lib.Context.for_device(device.index)
    x = torch.empty((calib_frames, 1, 224, 224), device=device, dtype=torch.float32)
    ctx.check(_lib.lib().pn_preprocess(ctx.handle, C.c_void_p(depth.data_ptr()), _lib.PN_DEPTH_F16, calib_frames,
                                       depth.shape[1], depth.shape[2], C.c_void_p(x.data_ptr()), 224, float(DEPTH_MAX),
                                       float(DEPTH_MEAN), float(DEPTH_STD), _lib.current_stream_ptr(device)), "pn_preprocess")
    prec = model.precision
    model.precision = "fp32"
    if gain != 1.0:                          # spread the heat logits: fewer cells within a rounding error of the threshold
        with torch.no_grad():
            model.model2_2[12].weight[:15] *= float(gain)
            model.model2_2[12].bias[:15] *= float(gain)
        model.invalidate()
    (_, heat, _), _ = model(x)
    s = heat[:, :15].double().clamp(1e-12, 1 - 1e-12)
    logit = torch.log(s / (1 - s)).permute(1, 0, 2, 3).reshape(15, -1)
    q = torch.quantile(logit, 1.0 - frac, dim=1)
    thr = float(np.log(0.1 / 0.9))
    with torch.no_grad():
        model.model2_2[12].bias[:15] += (thr - q).float().cpu()
    model.precision = prec
    model.invalidate()
    return model


def shard_indices(n_frames, rank, world):
    """Frame i goes to rank i % world."""
    return list(range(rank, n_frames, world))


def gather_records(local_frames, n_frames, rank, world, group=None):
    """ONE all-gather of fixed-size records; returns [n_frames, itemsize] uint8 in global frame
    order on every rank.  local_frames: [ceil(n_frames/world) or fewer, itemsize] uint8 tensor on the
    rank's device (or CPU tensor with the gloo backend)."""
    import torch.distributed as dist
    per = (n_frames + world - 1) // world
    item = local_frames.shape[1]
    pad = torch.zeros((per, item), dtype=torch.uint8, device=local_frames.device)
    pad[:local_frames.shape[0]] = local_frames
    out = torch.empty((world * per, item), dtype=torch.uint8, device=local_frames.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return deinterleave(out.view(world, per, item), n_frames)


def deinterleave(per_rank, n_frames):
    """[world, per, item] records as the all-gather delivers them (rank-major) -> [n_frames, item] in global frame order:
    frame i was processed by rank i % world as its (i // world)-th frame (shard_indices)."""
    world = per_rank.shape[0]
    idx = torch.arange(n_frames, device=per_rank.device)
    return per_rank[idx % world, idx // world]
