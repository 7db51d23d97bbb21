"""TEST-ONLY host statement of the temporal logic (mor_tracker.cpp), built on demand with g++ into tests/host_tracker/libmor_tracker_host.so."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        out, src, hdr = os.path.join(_HERE, "libmor_tracker_host.so"), os.path.join(_HERE, "mor_tracker.cpp"), os.path.join(_HERE, "mor_tracker.h")
        if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", out, src])
        L = C.CDLL(out)
        vp, i32 = C.c_void_p, C.c_int
        L.mor_tracker_create.restype = vp
        L.mor_tracker_create.argtypes = [vp, i32, i32]
        L.mor_tracker_destroy.argtypes = [vp]
        L.mor_tracker_push.argtypes = [vp, i32, vp, vp, i32, vp, vp]
        L.mor_tracker_filter.argtypes = [vp, vp, vp, vp]
        L.mor_tracker_get.argtypes = [vp, vp, vp, vp, i32]
        _LIB = L
    return _LIB


class HostTracker:
    """mor_tracker alone — the T1/F1 state machine, usable without a GPU."""

    def __init__(self, params, n_bad=4, n_good=3):
        self._h = lib().mor_tracker_create(C.addressof(params), n_bad, n_good)
        self._K = 0

    def push(self, centroids, det, pairs=None):
        c = np.ascontiguousarray(centroids, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(det, np.uint8)
        self._K = len(d)
        if pairs is None:
            rc = lib().mor_tracker_push(self._h, len(d), c.ctypes.data, d.ctypes.data, -1, None, None)
        else:
            q = np.ascontiguousarray([p[0] for p in pairs], np.int32)
            m = np.ascontiguousarray([p[1] for p in pairs], np.int32)
            rc = lib().mor_tracker_push(self._h, len(d), c.ctypes.data, d.ctypes.data, len(q), q.ctypes.data, m.ctypes.data)
        assert rc == 0

    def filter(self, sizes):
        sz = np.ascontiguousarray(sizes, np.int32)
        mv = np.zeros(max(self._K, 1), np.uint8)
        n = C.c_uint64(0)
        assert lib().mor_tracker_filter(self._h, sz.ctypes.data, mv.ctypes.data, C.addressof(n)) == 0
        return mv[: self._K], int(n.value)

    def tracks(self):
        xyz, conf, mx = np.empty((4096, 3), np.float32), np.empty(4096, np.int32), np.empty(4096, np.int32)
        k = lib().mor_tracker_get(self._h, xyz.ctypes.data, conf.ctypes.data, mx.ctypes.data, 4096)
        return xyz[:k], conf[:k], mx[:k]

    def __del__(self):
        try:
            if self._h:
                lib().mor_tracker_destroy(self._h)
                self._h = None
        except Exception:
            pass
