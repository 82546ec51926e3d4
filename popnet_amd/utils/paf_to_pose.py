"""``paf_to_pose`` with the reference's signature, computed on the GPU.

Drop-in for tpm/lib/utils/paf_to_pose.py:354-377 (NMS :75-153, find_connected_joints :156-264,
group_limbs_of_same_person :267-351): takes the HWC float32 network maps of ONE frame and returns
``(joint_list [N,5] float64, person_to_joint_assoc [P,J+2] float64)`` exactly as the reference
does -- but the work happens in three HIP kernels (csrc/parse_paf.hip) instead of Python loops over
scipy/cv2 calls.  For throughput use ``popnet_amd.pipeline.PoseEngine`` (whole batches stay on the
device); this per-frame wrapper exists for API compatibility and parity tests.
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib
from .. import pafprocess
from ..config import LIMBS
from .common_coco import BodyPart, Human

# same module-level names as the reference (paf_to_pose.py:28-30)
joint_to_limb_heatmap_relationship = [list(l) for l in LIMBS]
paf_xy_coords_per_limb = np.arange(2 * len(LIMBS)).reshape(-1, 2)
NUM_LIMBS = len(LIMBS)


def make_parse_cfg(config=None, input_size=224, w_org=480, h_org=640, intrinsics=None, depth_mean=3.0, depth_std=2.0):
    cfg = _lib.ParseCfg()
    _lib.lib().pn_parse_cfg_default(C.byref(cfg))
    if config is not None:
        cfg.thresh_heatmap = float(config.TEST.THRESH_HEATMAP)
        cfg.thresh_paf = float(config.TEST.THRESH_PAF)
        cfg.num_intermed_pts = int(config.TEST.NUM_INTERMED_PTS_BETWEEN_KEYPOINTS)
        cfg.downsample = int(config.MODEL.DOWNSAMPLE)
        if int(config.MODEL.NUM_KEYPOINTS) != _lib.PN_NUM_JOINTS:
            raise _lib.PopnetError("parse kernels are built for %d keypoints" % _lib.PN_NUM_JOINTS)
    cfg.input_size, cfg.w_org, cfg.h_org = int(input_size), int(w_org), int(h_org)
    if intrinsics is not None:
        cfg.fx, cfg.fy, cfg.cx, cfg.cy = (float(intrinsics[k]) for k in ('fx', 'fy', 'cx', 'cy'))
    cfg.depth_mean, cfg.depth_std = float(depth_mean), float(depth_std)
    return cfg


def parse_paf_batch(heat, paf, z, cfg, device=None):
    """heat [B,J+1,h,w], paf [B,2L,h,w], z [B,L+1,h,w]: float32 CUDA tensors (NCHW).
    Returns a numpy structured array of B pn_pose_frame records (copied to the host)."""
    for t, n in ((heat, "heat"), (paf, "paf"), (z, "z")):
        _lib.require_cuda_tensor(t, n)
    dev = heat.device
    B, _, h, w = heat.shape
    heat, paf, z = heat.contiguous().float(), paf.contiguous().float(), z.contiguous().float()
    frames = torch.empty((B, _lib.POSE_FRAME_DTYPE.itemsize), device=dev, dtype=torch.uint8)
    ctx = _lib.Context.for_device(dev.index)
    ctx.check(_lib.lib().pn_parse_paf(ctx.handle, C.c_void_p(heat.data_ptr()), C.c_void_p(paf.data_ptr()),
                                      C.c_void_p(z.data_ptr()), B, h, w, C.byref(cfg),
                                      C.c_void_p(frames.data_ptr()), _lib.current_stream_ptr(dev)), "pn_parse_paf")
    return frames.cpu().numpy().view(_lib.POSE_FRAME_DTYPE).reshape(B)


def frame_joint_list(fr):
    """pn_pose_frame -> the reference's joint_list ndarray ([N,5]: x, y, score, id, type)."""
    n = int(fr['n_peaks'])
    if n == 0:
        return np.array([])
    out = np.empty((n, 5), dtype=np.float64)
    out[:, 0] = fr['peak_x'][:n]
    out[:, 1] = fr['peak_y'][:n]
    out[:, 2] = fr['peak_score'][:n]
    out[:, 3] = np.arange(n)
    out[:, 4] = fr['peak_type'][:n]
    return out


def frame_assoc(fr):
    """pn_pose_frame -> the reference's person_to_joint_assoc ndarray ([P, J+2])."""
    p = int(fr['n_persons'])
    if p == 0:
        return np.array([])
    out = np.empty((p, _lib.PN_NUM_JOINTS + 2), dtype=np.float64)
    out[:, :_lib.PN_NUM_JOINTS] = fr['person_joint'][:p]
    out[:, -2] = fr['person_score'][:p]
    out[:, -1] = fr['person_count'][:p]
    return out


def parse_paf_unbounded(heat, paf, z, cfg):
    """The parse WITHOUT the record capacities (pn_parse_paf_unbounded: the second pass for a frame the fixed-size record flags
    as overflowing -- the reference itself has no limit, paf_to_pose.py:33-153,267-351).  heat [J+1,h,w], paf [2L,h,w],
    z [L+1,h,w]: float32 CUDA tensors of ONE frame.  Returns a dict of variable-length arrays: joint_list [N,5] and
    person_to_joint_assoc [P,J+2] as paf_to_pose returns them, plus joints_2d [P,J,2], joints_3d [P,J,3], part_conf [P,J]."""
    for t, n in ((heat, "heat"), (paf, "paf"), (z, "z")):
        _lib.require_cuda_tensor(t, n)
    dev = heat.device
    heat, paf, z = heat.contiguous().float(), paf.contiguous().float(), z.contiguous().float()
    h, w = heat.shape[-2], heat.shape[-1]
    ctx = _lib.Context.for_device(dev.index)
    L = _lib.lib()
    npk, npers = C.c_int(0), C.c_int(0)
    ctx.check(L.pn_parse_paf_unbounded(ctx.handle, C.c_void_p(heat.data_ptr()), C.c_void_p(paf.data_ptr()), C.c_void_p(z.data_ptr()), h, w,
                                       C.byref(cfg), C.byref(npk), C.byref(npers), _lib.current_stream_ptr(dev)), "pn_parse_paf_unbounded")
    N, P, J = npk.value, npers.value, _lib.PN_NUM_JOINTS
    xys, typ = np.zeros((N, 3), np.float32), np.zeros((N,), np.int32)
    pj, psc, pcn = np.zeros((P, J), np.int32), np.zeros((P,), np.float64), np.zeros((P,), np.int32)
    j2, j3, cf = np.zeros((P, J, 2), np.float64), np.zeros((P, J, 3), np.float64), np.zeros((P, J), np.float64)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    ctx.check(L.pn_parse_paf_unbounded_fetch(ctx.handle, vp(xys), vp(typ), vp(pj), vp(psc), vp(pcn), vp(j2), vp(j3), vp(cf)), "pn_parse_paf_unbounded_fetch")
    joint_list = np.empty((N, 5), np.float64)
    joint_list[:, :3] = xys
    joint_list[:, 3] = np.arange(N)
    joint_list[:, 4] = typ
    assoc = np.empty((P, J + 2), np.float64)
    assoc[:, :J], assoc[:, J], assoc[:, J + 1] = pj, psc, pcn
    return {"joint_list": joint_list if N else np.array([]), "person_to_joint_assoc": assoc if P else np.array([]),
            "person_joint": pj, "joints_2d": j2, "joints_3d": j3, "part_conf": cf}


def check_status(fr):
    if int(fr['status']):
        raise _lib.PopnetError("pose parse overflow (status=%d): more than %d peaks per joint or %d persons in a frame"
                               % (int(fr['status']), _lib.PN_MAX_PEAKS_PER_JOINT, _lib.PN_MAX_PERSONS))


def _device_of(*inputs):
    """The device the kernels run on: that of the first CUDA tensor among the inputs (a map that already lives on another GPU is parsed
    there, on that device's context and current stream -- ADVICE r05), else the current device."""
    for t in inputs:
        if isinstance(t, torch.Tensor) and t.is_cuda:
            return t.device
    return torch.device("cuda", torch.cuda.current_device())


def paf_to_pose(heatmaps, pafs, config):
    """heatmaps [h,w,J+1], pafs [h,w,2L]: float32 HWC ndarrays (or CUDA tensors) of one frame."""
    dev = _device_of(heatmaps, pafs)
    hm = torch.as_tensor(np.ascontiguousarray(heatmaps) if isinstance(heatmaps, np.ndarray) else heatmaps)
    pf = torch.as_tensor(np.ascontiguousarray(pafs) if isinstance(pafs, np.ndarray) else pafs)
    hm = hm.to(dev, torch.float32).permute(2, 0, 1)[None].contiguous()
    pf = pf.to(dev, torch.float32).permute(2, 0, 1)[None].contiguous()
    z = torch.zeros((1, NUM_LIMBS + 1, hm.shape[2], hm.shape[3]), device=dev, dtype=torch.float32)
    cfg = make_parse_cfg(config)
    fr = parse_paf_batch(hm, pf, z, cfg)[0]
    if int(fr['status']):
        # more peaks / persons than the fixed-size record holds: second pass without capacities (the reference has none)
        r = parse_paf_unbounded(hm[0], pf[0], z[0], cfg)
        return r["joint_list"], r["person_to_joint_assoc"]
    return frame_joint_list(fr), frame_assoc(fr)


def NMS(heatmaps, upsampFactor=1., bool_refine_center=True, bool_gaussian_filt=False, config=None):
    """tpm/lib/utils/paf_to_pose.py:75-153 on the GPU (pn_nms_peaks): heatmaps [h, w, >= NUM_KEYPOINTS] float32 (ndarray or CUDA tensor)
    -> a list of NUM_KEYPOINTS float64 arrays [n_j, 4] (x, y, score, running id), any number of maps (the COCO-18 caller included).
    Only the reference's default path is built: refined centres, no Gaussian filter, upsampFactor = 8."""
    if not bool_refine_center or bool_gaussian_filt:
        raise _lib.PopnetError("NMS: only bool_refine_center=True, bool_gaussian_filt=False is built (the reference's defaults)")
    if int(upsampFactor) != upsampFactor or int(upsampFactor) != 8:
        raise _lib.PopnetError("NMS: built for upsampFactor = 8 (MODEL.DOWNSAMPLE)")
    dev = _device_of(heatmaps)
    hm = torch.as_tensor(np.ascontiguousarray(heatmaps) if isinstance(heatmaps, np.ndarray) else heatmaps)
    nk = int(config.MODEL.NUM_KEYPOINTS)
    if hm.dim() != 3 or hm.shape[2] < nk:
        raise _lib.PopnetError("NMS: heatmaps must be [h, w, >= %d] (HWC), got %s" % (nk, tuple(hm.shape)))
    hm = hm.to(dev, torch.float32).permute(2, 0, 1)[:nk].contiguous()
    _, h, w = hm.shape
    cnt = torch.zeros((nk,), device=dev, dtype=torch.int32)
    xs, ys, sc = (torch.empty((nk, h * w), device=dev, dtype=torch.float32) for _ in range(3))
    ctx = _lib.Context.for_device(dev.index)
    ctx.check(_lib.lib().pn_nms_peaks(ctx.handle, C.c_void_p(hm.data_ptr()), nk, h, w, float(config.TEST.THRESH_HEATMAP), 8,
                                      C.c_void_p(cnt.data_ptr()), C.c_void_p(xs.data_ptr()), C.c_void_p(ys.data_ptr()),
                                      C.c_void_p(sc.data_ptr()), _lib.current_stream_ptr(dev)), "pn_nms_peaks")
    cnt = cnt.cpu().numpy()
    xs, ys, sc = xs.cpu().numpy(), ys.cpu().numpy(), sc.cpu().numpy()
    out, total = [], 0
    for j in range(nk):
        n = int(cnt[j])
        peaks = np.zeros((n, 4))
        peaks[:, 0], peaks[:, 1], peaks[:, 2] = xs[j, :n], ys[j, :n], sc[j, :n]
        peaks[:, 3] = np.arange(total, total + n)
        total += n
        out.append(peaks)
    return out


def paf_to_pose_cpp(heatmaps, pafs, config):
    """tpm/lib/utils/paf_to_pose.py:381-415: NMS -> joint_list [1, N, 5] float32 -> INTER_NEAREST x DOWNSAMPLE up-sampling of both maps ->
    `process_paf` + the six getters -> a list of `Human`s.  heatmaps [h, w, 19], pafs [h, w, 38] float32 HWC (COCO-18 topology, hard-coded
    on the C++ side: pafprocess.h:8-13).  NMS and process_paf run in libpopnet_hip.so; the nearest-neighbour up-sampling is an index
    repeat (cv2.INTER_NEAREST at an integer factor reads source cell floor(d / f))."""
    humans = []
    joint_list_per_joint_type = NMS(heatmaps, upsampFactor=config.MODEL.DOWNSAMPLE, config=config)
    joint_list = np.array([tuple(peak) + (joint_type,) for joint_type, joint_peaks in enumerate(joint_list_per_joint_type)
                           for peak in joint_peaks]).astype(np.float32)
    if joint_list.shape[0] > 0:
        joint_list = np.expand_dims(joint_list, 0)
        f = int(config.MODEL.DOWNSAMPLE)
        to_np = lambda a: a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
        paf_upsamp = np.repeat(np.repeat(to_np(pafs), f, axis=0), f, axis=1)
        heatmap_upsamp = np.repeat(np.repeat(to_np(heatmaps), f, axis=0), f, axis=1)
        pafprocess.process_paf(joint_list, heatmap_upsamp, paf_upsamp)
        for human_id in range(pafprocess.get_num_humans()):
            human = Human([])
            is_added = False
            for part_idx in range(config.MODEL.NUM_KEYPOINTS):
                c_idx = int(pafprocess.get_part_cid(human_id, part_idx))
                if c_idx < 0:
                    continue
                is_added = True
                human.body_parts[part_idx] = BodyPart('%d-%d' % (human_id, part_idx), part_idx,
                                                      float(pafprocess.get_part_x(c_idx)) / heatmap_upsamp.shape[1],
                                                      float(pafprocess.get_part_y(c_idx)) / heatmap_upsamp.shape[0],
                                                      pafprocess.get_part_score(c_idx))
            if is_added:
                human.score = pafprocess.get_score(human_id)
                humans.append(human)
    return humans
