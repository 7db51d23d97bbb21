"""ctypes binding of csrc/mor_synth.c — deterministic synthetic LiDAR streams."""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
SENSORS = {"hdl64": 0, "os128": 1, "agg10": 2, "hdl64_urban": 3}


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "csrc", "libmor_synth.so")
        if not os.path.exists(path):
            raise RuntimeError("libmor_synth.so not built — run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(path)
        L.mor_synth_points.restype = C.c_uint64
        L.mor_synth_points.argtypes = [C.c_int]
        L.mor_synth_frame.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.mor_synth_batch.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


def n_points(sensor):
    return int(lib().mor_synth_points(SENSORS[sensor]))


def frame(seed, sensor="hdl64", frame_idx=0):
    """Returns (xyzi float32 [N,4], pose float64 [7])."""
    n = n_points(sensor)
    out = np.empty((n, 4), np.float32)
    pose = np.empty(7, np.float64)
    rc = lib().mor_synth_frame(int(seed), SENSORS[sensor], int(frame_idx), out.ctypes.data, pose.ctypes.data)
    assert rc == 0
    return out, pose


def batch(seeds, frame_idx, sensor="hdl64"):
    """seeds, frame_idx: equal-length sequences.  Returns (xyzi [F,N,4], poses [F,7])."""
    seeds = np.ascontiguousarray(seeds, np.uint64)
    fi = np.ascontiguousarray(frame_idx, np.int32)
    n = n_points(sensor)
    out = np.empty((len(seeds), n, 4), np.float32)
    poses = np.empty((len(seeds), 7), np.float64)
    rc = lib().mor_synth_batch(SENSORS[sensor], len(seeds), seeds.ctypes.data, fi.ctypes.data, out.ctypes.data, poses.ctypes.data)
    assert rc == 0
    return out, poses
