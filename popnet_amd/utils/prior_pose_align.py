"""``parse_prior_pose`` with the reference's signature, computed on the GPU.

Drop-in for tpm/lib/utils/prior_pose_align.py:10-168: same arguments, same return structure
``(bboxes[B][n] float32[5], humans[B][n] float32[J,3], visibility[B][n] bool[J])``; with pred_vis=True the maps carry
5 + 4 J channels per anchor and ``visibility`` is float32[J] = in-bounds test x predicted visibility (:153-157).
One HIP workgroup per image decodes only the cells above the objectness threshold, sorts, builds
the IoU conflict matrix and runs the reference's suppression loop (csrc/parse_yolo.hip).
Unlike the reference, ``posemaps`` is NOT modified in place (calling the reference twice on the
same tensor double-applies the decode).
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


def parse_prior_pose(posemaps, anchors, num_joints, w_out, h_out, depth_mean, depth_std, conf_threshold=0.35,
                     nms_threshold=0.5, pred_vis=False, vis_margin=0):
    vis_pred = None
    if pred_vis:
        B = 1 if posemaps.dim() == 3 else posemaps.shape[0]
        vis_pred = torch.empty((B, _lib.PN_YOLO_MAX_DET, num_joints), device=posemaps.device, dtype=torch.float32)
    recs = parse_yolo_batch(posemaps, anchors, num_joints, w_out, h_out, depth_mean, depth_std, conf_threshold,
                            nms_threshold, vis_margin, vis_pred=vis_pred)
    vp = vis_pred.cpu().numpy() if pred_vis else None
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
