"""The three ``lib/utils/common.py`` functions on the hot path, reference signatures.

  paf_to_human_list             tpm/lib/utils/common.py:5-32     (list re-shaping of GPU results)
  retrieve_depth_heat_weighted  tpm/lib/utils/common.py:272-293  (HIP kernel, pn_retrieve_depth)
  pos_3d_from_2d_and_depth      tpm/lib/utils/common.py:107-115  (pinhole back-projection)
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib


def paf_to_human_list(joint_list, person_to_joint_assoc):
    """Unfolds (joint_list, assoc) into per-person joint lists, visibility and confidences.
    Pure re-indexing of already-computed results (no arithmetic)."""
    humans, visibility, conf_vec = [], [], []
    for human in person_to_joint_assoc:
        idx = np.asarray(human[:-2]).astype(int)
        joints = [[-1, -1] if i < 0 else joint_list[i, :2].tolist() for i in idx]
        conf = [0 if i < 0 else float(joint_list[i, 2]) for i in idx]
        humans.append(joints)
        visibility.append((idx >= 0).astype(int).tolist())
        conf_vec.append(conf)
    return humans, visibility, conf_vec


def retrieve_depth_heat_weighted_many(centers, depthmap, heatmap, radius=1):
    """Heat-weighted depth for n centres [(x, y), ...] on one map pair.  depthmap / heatmap: [h, w]
    float32 ndarrays or CUDA tensors.  A host ``heatmap`` ndarray is clamped in place (negatives
    -> 0) like the reference does."""
    dev = torch.device("cuda", torch.cuda.current_device())
    dm = torch.as_tensor(np.ascontiguousarray(depthmap) if isinstance(depthmap, np.ndarray) else depthmap)
    dm = dm.to(dev, torch.float32).contiguous()
    host_heat = isinstance(heatmap, np.ndarray)
    hm = torch.as_tensor(np.ascontiguousarray(heatmap) if host_heat else heatmap).to(dev, torch.float32).contiguous()
    h, w = dm.shape
    c = torch.as_tensor(np.asarray(centers, dtype=np.int32).reshape(-1, 2)).to(dev)
    out = torch.empty((c.shape[0],), device=dev, dtype=torch.float32)
    ctx = _lib.Context.for_device(dev.index)
    ctx.check(_lib.lib().pn_retrieve_depth(ctx.handle, C.c_void_p(dm.data_ptr()), C.c_void_p(hm.data_ptr()), h, w,
                                           C.c_void_p(c.data_ptr()), c.shape[0], int(radius), C.c_void_p(out.data_ptr()),
                                           _lib.current_stream_ptr(dev)), "pn_retrieve_depth")
    if host_heat:
        heatmap[...] = hm.cpu().numpy()
    return out.cpu().numpy()


def retrieve_depth_heat_weighted(center, depthmap, heatmap, radius=1):
    return retrieve_depth_heat_weighted_many([[int(center[0]), int(center[1])]], depthmap, heatmap, radius)[0]


def pos_3d_from_2d_and_depth(x_2d, y_2d, Z, cx, cy, fx, fy):
    """Pinhole back-projection, element-wise float64 like the reference (three NumPy expressions)."""
    X = (x_2d - cx) / fx * Z
    Y = (y_2d - cy) / fy * Z
    return np.vstack([X, Y, Z]).T
