"""ctypes binding of libpopnet_hip.so (the C ABI declared in include/popnet_hip.h).

There is NO fallback: if the HIP library is missing or fails to load, importing any compute entry
point of this package raises.  The library is built in-tree by ``popnet_amd/build.py`` (or
``__graft_entry__.build()``).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("POPNET_LIB_PATH") or os.path.join(_HERE, "libpopnet_hip.so")      # the override is for kernel experiments (variant builds)

PN_OK = 0
PN_PREC_F32, PN_PREC_BF16, PN_PREC_BF16X3 = 0, 1, 2
PN_NET_RTPOSE_LIGHT3D, PN_NET_YOLO_POSENET = 0, 1
PN_DEPTH_F16, PN_DEPTH_F32 = 0, 1

PN_NUM_JOINTS = 15
PN_NUM_LIMBS = 14
PN_MAX_PEAKS_PER_JOINT = 32
PN_MAX_PEAKS = PN_NUM_JOINTS * PN_MAX_PEAKS_PER_JOINT
PN_MAX_PERSONS = 32
PN_YOLO_MAX_DET = 64
PN_FRAME_OVERFLOW_PEAKS, PN_FRAME_OVERFLOW_PERSONS = 1, 2


class PopnetError(RuntimeError):
    pass


class TargetCfg(C.Structure):
    """pn_target_cfg"""
    _fields_ = [("input_x", C.c_int), ("input_y", C.c_int), ("stride", C.c_int), ("z_radius", C.c_int),
                ("sigma", C.c_double), ("depth_max", C.c_double), ("depth_mean", C.c_double), ("depth_std", C.c_double)]


class ParseCfg(C.Structure):
    """pn_parse_cfg"""
    _fields_ = [("thresh_heatmap", C.c_float), ("thresh_paf", C.c_float),
                ("num_intermed_pts", C.c_int), ("downsample", C.c_int), ("input_size", C.c_int),
                ("w_org", C.c_int), ("h_org", C.c_int),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("depth_mean", C.c_float), ("depth_std", C.c_float)]


# numpy mirror of pn_pose_frame / pn_yolo_frame (align=True reproduces the C struct layout; the
# sizes are cross-checked against pn_sizeof_* at load time)
POSE_FRAME_DTYPE = np.dtype([
    ("n_persons", np.int32), ("n_peaks", np.int32), ("status", np.uint32), ("reserved", np.int32),
    ("peak_x", np.float32, (PN_MAX_PEAKS,)), ("peak_y", np.float32, (PN_MAX_PEAKS,)),
    ("peak_score", np.float32, (PN_MAX_PEAKS,)), ("peak_type", np.int32, (PN_MAX_PEAKS,)),
    ("person_joint", np.int32, (PN_MAX_PERSONS, PN_NUM_JOINTS)),
    ("person_score", np.float64, (PN_MAX_PERSONS,)),
    ("person_count", np.int32, (PN_MAX_PERSONS,)),
    ("joints_2d", np.float64, (PN_MAX_PERSONS, PN_NUM_JOINTS, 2)),
    ("joints_3d", np.float64, (PN_MAX_PERSONS, PN_NUM_JOINTS, 3)),
    ("part_conf", np.float64, (PN_MAX_PERSONS, PN_NUM_JOINTS)),
], align=True)

YOLO_FRAME_DTYPE = np.dtype([
    ("n_det", np.int32), ("n_candidates", np.int32), ("status", np.uint32), ("reserved", np.int32),
    ("bbox", np.float32, (PN_YOLO_MAX_DET, 5)),
    ("human", np.float32, (PN_YOLO_MAX_DET, PN_NUM_JOINTS, 3)),
    ("visibility", np.int32, (PN_YOLO_MAX_DET, PN_NUM_JOINTS)),
    ("joints_2d", np.float32, (PN_YOLO_MAX_DET, PN_NUM_JOINTS, 2)),
    ("joints_3d", np.float32, (PN_YOLO_MAX_DET, PN_NUM_JOINTS, 3)),
    ("bbox_org", np.float32, (PN_YOLO_MAX_DET, 4)),
], align=True)

PN_WIRE_MAX_PERSONS = 16
POSE_WIRE_DTYPE = np.dtype([
    ("n_persons", np.int32), ("status", np.uint32),
    ("person_joint", np.int16, (PN_WIRE_MAX_PERSONS, PN_NUM_JOINTS)),
    ("vals", np.float32, (PN_WIRE_MAX_PERSONS, PN_NUM_JOINTS, 6)),      # x, y (original frame), X, Y, Z (m), part confidence
], align=True)

_lib = None

_vp, _i, _f, _d, _sz = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t
_SIGNATURES = {
    "pn_abi_version": (_i, []),
    "pn_build_experiments": (_i, []),
    "pn_mfma_sustained": (_i, [_vp, _d, _i, C.POINTER(_d), C.POINTER(_d), _vp]),
    "pn_create": (_vp, [_i]),
    "pn_destroy": (None, [_vp]),
    "pn_last_error": (_i, [_vp, C.c_char_p, _sz]),
    "pn_preprocess": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _f, _f, _f, _vp]),
    "pn_net_create": (_vp, [_vp, _i, _i, _i, _i]),
    "pn_net_destroy": (None, [_vp]),
    "pn_net_set_tensor": (_i, [_vp, C.c_char_p, _vp, C.POINTER(C.c_int64), _i]),
    "pn_net_finalize": (_i, [_vp, _i, _i, _i, _i]),
    "pn_rtpose_forward": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "pn_yolo_forward": (_i, [_vp, _vp, _i, _vp, _vp]),
    "pn_rtpose_forward_frames": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _f, _f, _vp, _vp, _vp, _vp]),
    "pn_yolo_forward_frames": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _f, _f, _vp, _vp]),
    "pn_net_read_activation": (_i, [_vp, C.c_char_p, _i, _vp, _sz, _vp]),
    "pn_net_copy_activation": (_i, [_vp, C.c_char_p, _i, _vp, _vp]),
    "pn_net_flops_per_frame": (_d, [_vp]),
    "pn_net_lock": (_i, [_vp, _i]),
    "pn_parse_reserve": (_i, [_vp, _i]),
    "pn_net_profile_begin": (_i, [_vp]),
    "pn_net_profile_end": (_i, [_vp, C.POINTER(_d), C.POINTER(C.c_int64), C.POINTER(_d), C.POINTER(_d), C.POINTER(C.c_int64)]),
    "pn_net_profile_kernel": (_i, [_vp, _i, C.c_char_p, _sz, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "pn_parse_cfg_default": (None, [C.POINTER(ParseCfg)]),
    "pn_parse_paf": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, C.POINTER(ParseCfg), _vp, _vp]),
    "pn_parse_paf_wire": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, C.POINTER(ParseCfg), _vp, _vp, _vp]),
    "pn_target_cfg_default": (None, [C.POINTER(TargetCfg)]),
    "pn_compose_depth": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp]),
    "pn_rasterize_targets": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, C.POINTER(TargetCfg), _vp, _vp, _vp, _vp, _vp]),
    "pn_train_set_precision": (_i, [_vp, _i]),
    "pn_train_ws_keep": (_i, [_vp, _i]),
    "pn_train_pack_cache": (_i, [_vp, _i]),
    "pn_train_pack_refresh": (_i, [_vp, _vp]),
    "pn_conv2d_forward": (_i, [_vp, _vp, _vp, _vp, _vp] + [_i] * 9 + [_vp]),
    "pn_conv2d_dgrad": (_i, [_vp, _vp, _vp, _vp] + [_i] * 8 + [_vp]),
    "pn_conv2d_wgrad": (_i, [_vp, _vp, _vp, _vp, _vp] + [_i] * 8 + [_vp]),
    "pn_bn_train_forward": (_i, [_vp] * 10 + [_f, _f] + [_i] * 4 + [_vp]),
    "pn_bn_train_backward": (_i, [_vp] * 8 + [_i] * 4 + [_vp] * 4 + [_i, _vp]),
    "pn_avgpool3s2_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "pn_avgpool3s2_backward": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "pn_head_forward": (_i, [_vp, _vp, _vp, _vp] + [_i] * 4 + [_vp, _vp, _i, _vp, _vp]),
    "pn_head_backward": (_i, [_vp, _vp, _vp, _vp, _vp] + [_i] * 5 + [_vp, _vp]),
    "pn_slice_copy": (_i, [_vp, _vp, _i, _vp] + [_i] * 5 + [_vp]),
    "pn_sgd_nesterov": (_i, [_vp, _vp, _vp, _vp, C.c_size_t, _f, _f, _f, _i, _f, _vp]),
    "pn_trainer_create": (_vp, [_vp]),
    "pn_trainer_destroy": (None, [_vp]),
    "pn_trainer_set_precision": (_i, [_vp, _i]),
    "pn_trainer_set_param": (_i, [_vp, C.c_char_p, _sz, _sz]),
    "pn_trainer_set_stat": (_i, [_vp, C.c_char_p, _vp]),
    "pn_trainer_finalize": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _f]),
    "pn_trainer_forward_backward": (_i, [_vp] * 8),
    "pn_trainer_conv_flops": (_d, [_vp]),
    "pn_retrieve_depth": (_i, [_vp, _vp, _vp, _i, _i, _vp, _i, _i, _vp, _vp]),
    "pn_nms_peaks": (_i, [_vp, _vp, _i, _i, _i, _f, _i, _vp, _vp, _vp, _vp, _vp]),
    "pn_parse_yolo": (_i, [_vp, _vp, _i, _i, _i, C.POINTER(C.c_float), _i, _i, _i, _i, _f, _f, _f, _f, _i, C.POINTER(ParseCfg), _vp, _vp]),
    "pn_parse_paf_unbounded": (_i, [_vp, _vp, _vp, _vp, _i, _i, C.POINTER(ParseCfg), C.POINTER(C.c_int), C.POINTER(C.c_int), _vp]),
    "pn_parse_paf_unbounded_fetch": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pn_parse_yolo_predvis": (_i, [_vp, _vp, _i, _i, _i, C.POINTER(C.c_float), _i, _i, _i, _i, _f, _f, _f, _f, _i, C.POINTER(ParseCfg), _vp, _vp, _vp]),
    "pn_sizeof_pose_frame": (_sz, []),
    "pn_pack_pose_frames": (_i, [_vp, _vp, _i, _vp, _vp]),
    "pn_sizeof_pose_wire": (_sz, []),
    "pn_sizeof_yolo_frame": (_sz, []),
    "pn_debug_cubic_coeffs": (None, [_f, C.POINTER(C.c_float)]),
    "process_paf": (_i, [_i, _i, _i, _vp, _i, _i, _i, _vp, _i, _i, _i, _vp]),
    "get_num_humans": (_i, []),
    "get_part_cid": (_i, [_i, _i]),
    "get_score": (_f, [_i]),
    "get_part_x": (_i, [_i]),
    "get_part_y": (_i, [_i]),
    "get_part_score": (_f, [_i]),
}


def lib():
    """Loads (once) and returns the ctypes handle.  Raises PopnetError when the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PopnetError(
            "popnet_amd: %s not found -- the HIP extension is required (no CPU fallback). "
            "Build it with `python popnet_amd/build.py`." % LIB_PATH)
    try:
        handle = C.CDLL(LIB_PATH)
    except OSError as e:
        raise PopnetError("popnet_amd: cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(handle, name)      # AttributeError => ABI mismatch, let it surface
        fn.restype = res
        fn.argtypes = args
    if handle.pn_abi_version() != 1:
        raise PopnetError("popnet_amd: ABI version mismatch")
    if handle.pn_sizeof_pose_frame() != POSE_FRAME_DTYPE.itemsize:
        raise PopnetError("pn_pose_frame layout mismatch: C %d vs numpy %d"
                          % (handle.pn_sizeof_pose_frame(), POSE_FRAME_DTYPE.itemsize))
    if handle.pn_sizeof_pose_wire() != POSE_WIRE_DTYPE.itemsize:
        raise PopnetError("pn_pose_wire layout mismatch: C %d vs numpy %d" % (handle.pn_sizeof_pose_wire(), POSE_WIRE_DTYPE.itemsize))
    if handle.pn_sizeof_yolo_frame() != YOLO_FRAME_DTYPE.itemsize:
        raise PopnetError("pn_yolo_frame layout mismatch")
    _lib = handle
    return _lib


def declared_symbols():
    return sorted(_SIGNATURES)


class Context:
    """pn_ctx wrapper (one per device)."""
    _cache = {}

    def __init__(self, device_index):
        self.device_index = device_index
        self.handle = lib().pn_create(device_index)
        if not self.handle:
            raise PopnetError("pn_create failed")
        msg = self.last_error()
        if msg:
            raise PopnetError(msg)

    @classmethod
    def for_device(cls, device_index):
        ctx = cls._cache.get(device_index)
        if ctx is None:
            ctx = cls._cache[device_index] = cls(device_index)
        return ctx

    def __del__(self):                     # private contexts (engines that own their scratch) release it; the per-device ones live on
        try:
            if self.handle and Context._cache.get(self.device_index) is not self:
                lib().pn_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def last_error(self):
        buf = C.create_string_buffer(1024)
        lib().pn_last_error(self.handle, buf, 1024)
        return buf.value.decode()

    def check(self, rc, what):
        if rc != PN_OK:
            raise PopnetError("%s failed (%d): %s" % (what, rc, self.last_error()))


def require_cuda_tensor(t, name):
    import torch
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise PopnetError("popnet_amd: %s must be a CUDA/ROCm tensor -- the HIP path has no CPU fallback" % name)


def current_stream_ptr(device):
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
