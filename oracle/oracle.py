"""ctypes binding of the CPU oracle (oracle/libmor_oracle.so).

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; the product package (dynamicslamtool_amd) never does.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libmor_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.oracle_sizeof_params.restype = C.c_size_t
        L.oracle_create.restype = C.c_void_p
        L.oracle_create.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.oracle_destroy.argtypes = [C.c_void_p]
        L.oracle_push.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.oracle_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        for name in ("oracle_get_counts", "oracle_get_labels", "oracle_get_ground_indices", "oracle_get_centroids", "oracle_get_detection"):
            getattr(L, name).argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_get_clusters.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_get_correspondences.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.oracle_get_tracks.argtypes = [C.c_void_p] + [C.c_void_p] * 3
        L.oracle_get_prev_cluster_count.restype = C.c_uint32
        L.oracle_get_prev_clustered.restype = C.c_uint32
        L.oracle_get_prev_clustered.argtypes = [C.c_void_p]
        L.oracle_get_prev_transformed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_get_markers.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_get_moving_clusters.restype = C.c_uint32
        L.oracle_get_moving_clusters.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_get_prev_cluster_count.argtypes = [C.c_void_p]
        L.oracle_get_busy_seconds.restype = C.c_double
        L.oracle_get_busy_seconds.argtypes = [C.c_void_p]
        L.oracle_census_read.argtypes = [C.c_void_p, C.c_int]
        L.oracle_set_literal_pruning.argtypes = [C.c_int]
        _LIB = L
    return _LIB


CENSUS_KEYS = ("c1_pairs_within_4ulp_of_r2", "g2_pairs_within_4ulp_of_leaf2", "radius_visits_owed_to_slack", "radius_neighbours_owed_to_slack", "nn_visits_owed_to_slack",
               "nn_results_owed_to_slack", "nn_exact_ties", "equal_size_clusters", "volume_gates_within_1e-6", "method1_distances_within_4ulp_of_a_bound",
               "method2_points_within_1ulp_of_a_voxel_face", "g2_terms_within_1e-6_of_threshold", "radius_queries", "nn_queries")


def census_reset():
    lib().oracle_census_reset()


def census_read():
    """Process-wide margin counters of the oracle since the last reset (mor_oracle.h), as a dict."""
    a = (C.c_ulonglong * len(CENSUS_KEYS))()
    n = lib().oracle_census_read(a, len(CENSUS_KEYS))
    assert n == len(CENSUS_KEYS)
    return {k: int(a[i]) for i, k in enumerate(CENSUS_KEYS)}


def set_literal_pruning(on):
    """True: the kd-tree prunes with FLANN's literal test (no slack); False (default): with the slack that misses nothing."""
    lib().oracle_set_literal_pruning(1 if on else 0)


class Counts(C.Structure):
    _fields_ = [("n_in", C.c_uint64), ("n_trim", C.c_uint64), ("n_cloud", C.c_uint64), ("n_ground", C.c_uint64),
                ("n_clusters", C.c_uint32), ("n_clustered", C.c_uint32), ("n_corr", C.c_uint32), ("n_tracks", C.c_uint32)]


class Oracle:
    """One sensor stream through the CPU restatement (push → filter per frame)."""

    def __init__(self, params, n_bad=4, n_good=3):
        L = lib()
        assert L.oracle_sizeof_params() == C.sizeof(params), "params layout mismatch"
        self._p = params
        self._h = L.oracle_create(C.addressof(params), n_bad, n_good)

    def close(self):
        if self._h:
            lib().oracle_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def push(self, xyzi, pose, point_step=None, offsets=None):
        """xyzi: float32 [N,4] (x,y,z,intensity) unless point_step/offsets describe another blob."""
        if point_step is None:
            a = np.ascontiguousarray(xyzi, np.float32).reshape(-1, 4)
            n, step, offs, buf = a.shape[0], 16, (0, 4, 8, 12), a
        else:
            buf = np.ascontiguousarray(xyzi).view(np.uint8).reshape(-1)
            n, step, offs = buf.size // point_step, point_step, offsets
        pose = np.ascontiguousarray(pose, np.float64)
        self._last_n = n
        rc = lib().oracle_push(self._h, buf.ctypes.data, n, step, offs[0], offs[1], offs[2], offs[3], pose.ctypes.data)
        assert rc == 0
        return rc

    def counts(self):
        c = Counts()
        lib().oracle_get_counts(self._h, C.addressof(c))
        return c

    def filter(self):
        c = self.counts()
        out = np.empty((max(int(c.n_trim), 1), 4), np.float32)
        n = C.c_uint64(0)
        rc = lib().oracle_filter(self._h, out.ctypes.data, C.addressof(n))
        if rc != 0:
            raise RuntimeError("oracle_filter rc=%d" % rc)
        return out[: n.value].copy()

    def labels(self):
        c = self.counts()
        a = np.empty(max(int(c.n_trim), 1), np.int32)
        lib().oracle_get_labels(self._h, a.ctypes.data)
        return a[: c.n_trim]

    def ground_indices(self):
        c = self.counts()
        a = np.empty(max(int(c.n_ground), 1), np.int32)
        lib().oracle_get_ground_indices(self._h, a.ctypes.data)
        return a[: c.n_ground]

    def clusters(self):
        c = self.counts()
        off = np.zeros(c.n_clusters + 1, np.int32)
        idx = np.empty(max(int(c.n_clustered), 1), np.int32)
        lib().oracle_get_clusters(self._h, off.ctypes.data, idx.ctypes.data)
        return off, idx[: c.n_clustered]

    def centroids(self):
        c = self.counts()
        a = np.empty((max(int(c.n_clusters), 1), 3), np.float32)
        lib().oracle_get_centroids(self._h, a.ctypes.data)
        return a[: c.n_clusters]

    def detection(self):
        c = self.counts()
        a = np.zeros(max(int(c.n_clusters), 1), np.uint8)
        lib().oracle_get_detection(self._h, a.ctypes.data)
        return a[: c.n_clusters]

    def correspondences(self):
        c = self.counts()
        n = max(int(c.n_corr), 1)
        q, m = np.empty(n, np.int32), np.empty(n, np.int32)
        d, s = np.empty(n, np.float32), np.empty(n, np.float64)
        lib().oracle_get_correspondences(self._h, q.ctypes.data, m.ctypes.data, d.ctypes.data, s.ctypes.data)
        k = c.n_corr
        return q[:k], m[:k], d[:k], s[:k]

    def tracks(self):
        c = self.counts()
        n = max(int(c.n_tracks), 1)
        xyz, conf, mx = np.empty((n, 3), np.float32), np.empty(n, np.int32), np.empty(n, np.int32)
        lib().oracle_get_tracks(self._h, xyz.ctypes.data, conf.ctypes.data, mx.ctypes.data)
        k = c.n_tracks
        return xyz[:k], conf[:k], mx[:k]

    def markers(self):
        """mark_cluster (:7-58) of every cluster of cb: (position K×3 float-accumulated centroid, scale K×3 with 0 → 0.1)."""
        K = int(self.counts().n_clusters)
        pos, scale = np.empty((max(K, 1), 3), np.float32), np.empty((max(K, 1), 3), np.float32)
        lib().oracle_get_markers(self._h, pos.ctypes.data, scale.ctypes.data)
        return pos[:K], scale[:K]

    def moving_clusters(self):
        """Cluster index per tracked centroid the latest filterCloud visited, in mo_vec order (the markers of :641)."""
        n = int(lib().oracle_get_moving_clusters(self._h, None))
        out = np.empty(max(n, 1), np.int32)
        lib().oracle_get_moving_clusters(self._h, out.ctypes.data)
        return out[:n]

    def prev_transformed(self):
        """ca after the in-place transform of :540-551: (centroids K_prev×3, cluster points C_prev×4 in cluster order)."""
        K, Cn = int(lib().oracle_get_prev_cluster_count(self._h)), int(lib().oracle_get_prev_clustered(self._h))
        cen, pts = np.empty((max(K, 1), 3), np.float32), np.empty((max(Cn, 1), 4), np.float32)
        lib().oracle_get_prev_transformed(self._h, cen.ctypes.data, pts.ctypes.data)
        return cen[:K], pts[:Cn]

    def busy_seconds(self):
        return float(lib().oracle_get_busy_seconds(self._h))
