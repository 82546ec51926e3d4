"""Module-level stand-in for the reference's SWIG module ``pafprocess``
(tpm/lib/pafprocess/pafprocess.py, generated from pafprocess.i): same seven function names and
argument meaning, numpy float32 arrays in (the SWIG typemap ``IN_ARRAY3`` of pafprocess.i:14 expands
one [D1,D2,D3] array into (int, int, int, float*)), results fetched one scalar at a time -- so
``paf_to_pose_cpp`` (tpm/lib/utils/paf_to_pose.py:381-415) runs unchanged on top of it.
The computation happens in libpopnet_hip.so (csrc/pafprocess_compat.hip) on the GPU.
"""
import ctypes as C

import numpy as np

from . import _lib


def _arr3(a, name):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 3:
        raise TypeError("%s: expected a 3-D float32 array, got %d-D" % (name, a.ndim))
    return a


def process_paf(peaks, heatmap, pafmap):
    peaks, heatmap, pafmap = _arr3(peaks, "peaks"), _arr3(heatmap, "heatmap"), _arr3(pafmap, "pafmap")
    rc = _lib.lib().process_paf(*peaks.shape, peaks.ctypes.data_as(C.c_void_p), *heatmap.shape,
                                heatmap.ctypes.data_as(C.c_void_p), *pafmap.shape, pafmap.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise _lib.PopnetError("process_paf failed with status %d" % rc)
    return rc


def get_num_humans():
    return _lib.lib().get_num_humans()


def get_part_cid(human_id, part_id):
    return _lib.lib().get_part_cid(int(human_id), int(part_id))


def get_score(human_id):
    return _lib.lib().get_score(int(human_id))


def get_part_x(cid):
    return _lib.lib().get_part_x(int(cid))


def get_part_y(cid):
    return _lib.lib().get_part_y(int(cid))


def get_part_score(cid):
    return _lib.lib().get_part_score(int(cid))


def run(peaks, heatmap, pafmap):
    """Convenience: process_paf + all getters -> list of dict(score, parts={part: (cid, x, y, score)})."""
    process_paf(peaks, heatmap, pafmap)
    humans = []
    for h in range(get_num_humans()):
        parts = {}
        for p in range(18):
            cid = get_part_cid(h, p)
            if cid >= 0:
                parts[p] = (cid, get_part_x(cid), get_part_y(cid), float(get_part_score(cid)))
        humans.append({'score': float(get_score(h)), 'parts': parts})
    return humans
