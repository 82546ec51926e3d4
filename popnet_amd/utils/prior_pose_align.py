"""``parse_prior_pose`` with the reference's signature, computed on the GPU.

Drop-in for tpm/lib/utils/prior_pose_align.py:10-168: same arguments, same return structure
``(bboxes[B][n] float32[5], humans[B][n] float32[J,3], visibility[B][n] bool[J])``; with pred_vis=True the maps carry
5 + 4 J channels per anchor and ``visibility`` is float32[J] = in-bounds test x predicted visibility (:153-157).
One HIP workgroup per image decodes only the cells above the objectness threshold, sorts, builds
the IoU conflict matrix and runs the reference's suppression loop (csrc/parse_yolo.hip).
By default ``posemaps`` is NOT modified (calling the reference twice on the same tensor double-applies
the decode); ``inplace=True`` opts into the reference's side effect, bit for bit.
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib


def parse_yolo_batch(posemaps, anchors, num_joints, w_out, h_out, depth_mean, depth_std, conf_threshold,
                     nms_threshold, vis_margin=0, glue_cfg=None, frames=None, vis_pred=None):
    """posemaps: float32 CUDA tensor [B, A*(5+3J), h, w].  Returns host records (structured array), or --
    when a device `frames` buffer is given -- leaves them on the device (no synchronisation).  glue_cfg
    (a _lib.ParseCfg) additionally fills joints_2d / joints_3d / bbox_org like the evaluation script."""
    _lib.require_cuda_tensor(posemaps, "posemaps")
    if posemaps.dim() == 3:
        posemaps = posemaps.unsqueeze(0)
    pm = posemaps.contiguous().float()
    B, _, h, w = pm.shape
    dev = pm.device
    keep_on_device = frames is not None
    if frames is None:
        frames = torch.empty((B, _lib.YOLO_FRAME_DTYPE.itemsize), device=dev, dtype=torch.uint8)
    flat = [float(v) for a in anchors for v in a]
    arr = (C.c_float * len(flat))(*flat)
    ctx = _lib.Context.for_device(dev.index)
    want_c = len(anchors) * (5 + (4 if vis_pred is not None else 3) * num_joints)
    if pm.shape[1] != want_c:
        raise _lib.PopnetError("posemaps has %d channels, %d anchors x (5 + %d x %d joints) = %d expected"
                               % (pm.shape[1], len(anchors), 4 if vis_pred is not None else 3, num_joints, want_c))
    args = (ctx.handle, C.c_void_p(pm.data_ptr()), B, h, w, arr, len(anchors), num_joints,
            int(w_out), int(h_out), float(depth_mean), float(depth_std), float(conf_threshold),
            float(nms_threshold), int(vis_margin), C.byref(glue_cfg) if glue_cfg is not None else None,
            C.c_void_p(frames.data_ptr()))
    if vis_pred is not None:         # [B, PN_YOLO_MAX_DET, J] float32 device tensor
        ctx.check(_lib.lib().pn_parse_yolo_predvis(*args, C.c_void_p(vis_pred.data_ptr()), _lib.current_stream_ptr(dev)), "pn_parse_yolo_predvis")
    else:
        ctx.check(_lib.lib().pn_parse_yolo(*args, _lib.current_stream_ptr(dev)), "pn_parse_yolo")
    if keep_on_device:
        return frames[:B]
    return frames.cpu().numpy().view(_lib.YOLO_FRAME_DTYPE).reshape(B)


def decode_in_place(posemaps, anchors, num_joints, depth_mean, depth_std):
    """What the reference leaves behind in the caller's tensor (prior_pose_align.py:22-52): a 3-D input gains its batch dimension and,
    per anchor, channels dx dy w h become centre / size as fractions of the map, the joint x / y offsets become map fractions and the
    joint depths are un-normalised -- float32, in the reference's operation order (add then divide; multiply, add, divide)."""
    if posemaps.dim() == 3:
        posemaps.unsqueeze_(0)
    B, _, h, w = posemaps.shape
    A, J = len(anchors), num_joints
    v = posemaps.view(B, A, -1, h * w)
    dev = posemaps.device
    col = torch.arange(w, dtype=torch.float32, device=dev).repeat(h)                  # cell column / row of every flattened position
    row = torch.arange(h, dtype=torch.float32, device=dev).repeat_interleave(w)
    aw = torch.tensor([float(a[0]) for a in anchors], dtype=torch.float32, device=dev)
    ah = torch.tensor([float(a[1]) for a in anchors], dtype=torch.float32, device=dev)
    # divisors as device TENSORS: a Python-scalar divisor takes the GPU fast path x * (1 / w), one ulp off the true quotient the CPU
    # reference (and pn_parse_yolo) computes
    wt, ht = torch.full((), float(w), device=dev), torch.full((), float(h), device=dev)
    v[:, :, 0, :].add_(col).div_(wt)
    v[:, :, 1, :].add_(row).div_(ht)
    v[:, :, 2, :].mul_(aw.view(1, A, 1)).div_(wt)
    v[:, :, 3, :].mul_(ah.view(1, A, 1)).div_(ht)
    v[:, :, 5:5 + J, :].mul_((aw / 2.0).view(1, A, 1, 1)).add_(col).div_(wt)
    v[:, :, 5 + J:5 + 2 * J, :].mul_((ah / 2.0).view(1, A, 1, 1)).add_(row).div_(ht)
    v[:, :, 5 + 2 * J:5 + 3 * J, :].mul_(depth_std).add_(depth_mean)
    return posemaps


def parse_prior_pose(posemaps, anchors, num_joints, w_out, h_out, depth_mean, depth_std, conf_threshold=0.35,
                     nms_threshold=0.5, pred_vis=False, vis_margin=0, inplace=False):
    """inplace=False (default): `posemaps` is left untouched.  inplace=True reproduces the reference's side effect as well: after the
    call the caller's tensor holds the decoded maps (decode_in_place), so a SECOND call on it decodes the already-decoded values --
    exactly what calling the reference twice does.  Needs a contiguous float32 tensor (it is modified through a view)."""
    if inplace and (posemaps.dtype != torch.float32 or not posemaps.is_contiguous()):
        raise _lib.PopnetError("parse_prior_pose(inplace=True) needs a contiguous float32 tensor")
    vis_pred = None
    if pred_vis:
        B = 1 if posemaps.dim() == 3 else posemaps.shape[0]
        vis_pred = torch.empty((B, _lib.PN_YOLO_MAX_DET, num_joints), device=posemaps.device, dtype=torch.float32)
    recs = parse_yolo_batch(posemaps, anchors, num_joints, w_out, h_out, depth_mean, depth_std, conf_threshold,
                            nms_threshold, vis_margin, vis_pred=vis_pred)
    vp = vis_pred.cpu().numpy() if pred_vis else None
    if inplace:                     # the records above were decoded from the raw maps; now leave the reference's side effect behind
        decode_in_place(posemaps, anchors, num_joints, depth_mean, depth_std)
    bboxes_out, humans_prior, visibility = [], [], []
    for fr in recs:
        if int(fr['status']):
            raise _lib.PopnetError("yolo decode overflow (status=%d)" % int(fr['status']))
        n = int(fr['n_det'])
        if n == 0:
            bboxes_out.append([]); humans_prior.append([]); visibility.append([])
            continue
        bboxes_out.append([fr['bbox'][i].copy() for i in range(n)])
        humans_prior.append([fr['human'][i].copy() for i in range(n)])
        if pred_vis:
            visibility.append([vp[len(visibility)][i].copy() for i in range(n)])
        else:
            visibility.append([fr['visibility'][i].astype(bool) for i in range(n)])
    return bboxes_out, humans_prior, visibility
