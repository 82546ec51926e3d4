"""ctypes access to the two CPU builds of `process_paf` (ORACLE; test infrastructure):
``restated()`` = oracle/pafprocess_oracle.c, ``reference()`` = the reference's own C++ compiled by
oracle/Makefile into oracle/_ref (None when that file is absent)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class _PafLib:
    def __init__(self, lib, names):
        self._lib = lib
        fp = C.POINTER(C.c_float)
        self._pp = getattr(lib, names['process_paf'])
        self._pp.restype = C.c_int
        self._pp.argtypes = [C.c_int] * 3 + [fp] + [C.c_int] * 3 + [fp] + [C.c_int] * 3 + [fp]
        self._g = {}
        for k, (res, args) in {'get_num_humans': (C.c_int, []), 'get_part_cid': (C.c_int, [C.c_int, C.c_int]),
                               'get_score': (C.c_float, [C.c_int]), 'get_part_x': (C.c_int, [C.c_int]),
                               'get_part_y': (C.c_int, [C.c_int]), 'get_part_score': (C.c_float, [C.c_int])}.items():
            f = getattr(lib, names[k])
            f.restype, f.argtypes = res, args
            self._g[k] = f

    def run(self, peaks, heat, paf):
        """peaks [1,N,5], heat [H,W,19], paf [H,W,38] float32 C-contiguous.  Returns a list of
        humans: dict(score, parts={part_id: (cid, x, y, score)})."""
        peaks, heat, paf = (np.ascontiguousarray(a, dtype=np.float32) for a in (peaks, heat, paf))
        fp = C.POINTER(C.c_float)
        rc = self._pp(*peaks.shape, peaks.ctypes.data_as(fp), *heat.shape, heat.ctypes.data_as(fp),
                      *paf.shape, paf.ctypes.data_as(fp))
        if rc != 0:
            raise RuntimeError("process_paf returned %d" % rc)
        humans = []
        for h in range(self._g['get_num_humans']()):
            parts = {}
            for p in range(18):
                cid = self._g['get_part_cid'](h, p)
                if cid >= 0:
                    parts[p] = (cid, self._g['get_part_x'](cid), self._g['get_part_y'](cid), float(self._g['get_part_score'](cid)))
            humans.append({'score': float(self._g['get_score'](h)), 'parts': parts})
        return humans


def build():
    subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)


def restated():
    path = os.path.join(_HERE, "_build", "libpafprocess_oracle.so")
    if not os.path.exists(path):
        build()
    names = {k: 'oracle_' + k for k in ('process_paf', 'get_num_humans', 'get_part_cid', 'get_score', 'get_part_x', 'get_part_y', 'get_part_score')}
    return _PafLib(C.CDLL(path), names)


def reference():
    path = os.path.join(_HERE, "_ref", "libpafprocess_ref.so")
    if not os.path.exists(path):
        return None
    # C++ (Itanium-mangled) names of pafprocess.h:53-59
    names = {'process_paf': '_Z11process_pafiiiPfiiiS_iiiS_', 'get_num_humans': '_Z14get_num_humansv',
             'get_part_cid': '_Z12get_part_cidii', 'get_score': '_Z9get_scorei', 'get_part_x': '_Z10get_part_xi',
             'get_part_y': '_Z10get_part_yi', 'get_part_score': '_Z14get_part_scorei'}
    return _PafLib(C.CDLL(path), names)
