"""ctypes binding of libmor_hip.so (include/mor_hip.h).  No CPU fallback: if the library is missing
or there is no HIP device, construction raises."""
import ctypes as C
import os
import numpy as np

from .params import MorParams

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MOR_HIP_LIB") or os.path.join(_HERE, "csrc", "libmor_hip.so")
_LIB = None
MOR_NO_FIELD = 0xFFFFFFFF

# every symbol include/mor_hip.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "mor_sizeof_params", "mor_last_error", "mor_batch_create", "mor_batch_destroy", "mor_batch_streams", "mor_push_batch",
    "mor_filter_batch", "mor_filter_batch_ex", "mor_batch_set_async", "mor_batch_wait", "mor_get_output_device", "mor_create", "mor_push", "mor_filter", "mor_destroy", "mor_get_counts",
    "mor_get_labels", "mor_get_ground_indices", "mor_get_clusters", "mor_get_centroids", "mor_get_detection", "mor_get_boxes",
    "mor_get_correspondences", "mor_get_tracks", "mor_get_cluster_collection", "mor_get_stage_counts", "mor_device_alloc", "mor_device_free", "mor_host_alloc", "mor_host_free", "mor_host_register", "mor_host_unregister",
    "mor_device_upload", "mor_device_download", "mor_device_synchronize", "mor_device_count", "mor_get_last_timing",
    "mor_kernel_timing_enable", "mor_kernel_timing_read", "mor_get_frame_log", "mor_debug_read", "mor_debug_config", "mor_get_markers", "mor_kernel_timeline_read",
    "mor_device_numa_node", "mor_bind_thread_to_device_node", "mor_build_hash", "mor_get_moving_clusters",
]


class CloudView(C.Structure):
    _fields_ = [("data", C.c_void_p), ("n_points", C.c_uint64), ("point_step", C.c_uint32), ("off_x", C.c_uint32),
                ("off_y", C.c_uint32), ("off_z", C.c_uint32), ("off_intensity", C.c_uint32), ("on_device", C.c_int32)]


class Counts(C.Structure):
    _fields_ = [("n_in", C.c_uint64), ("n_trim", C.c_uint64), ("n_cloud", C.c_uint64), ("n_ground", C.c_uint64),
                ("n_clusters", C.c_uint32), ("n_clustered", C.c_uint32), ("n_corr", C.c_uint32), ("n_tracks", C.c_uint32)]


_PROFILER_ENV = ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCP_TOOL_LIB", "HSA_TOOLS_LIB", "ROCPROFILER_REGISTER_FORCE_LOAD")


def _profiler_preload(env=None):
    """True when this process runs under a profiler / tool preload (rocprofv3 exports ROCP_TOOL_LIBRARIES, LD_PRELOAD and ROCPROF* / ROCPROFILER_* variables):
    the GPU may then be initialised already, and no child compiler may be started from here."""
    env = os.environ if env is None else env
    return any(env.get(k) for k in _PROFILER_ENV) or any(k.startswith(("ROCPROFILER_", "ROCPROF_", "ROCPROFV3_")) for k in env)


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libmor_hip.so is not built (python -m dynamicslamtool_amd.build); there is no CPU fallback")
        if not os.environ.get("MOR_HIP_LIB") and not os.environ.get("MOR_ALLOW_STALE_LIB"):
            from . import build as _build
            if _build.stale("hip") and os.path.exists(_build._hipcc()) and not os.environ.get("MOR_NO_AUTOBUILD") and not _profiler_preload():
                # bench.py and the exp/ scripts come through here without having called build(): rebuild (seconds when only the hash differs,
                # a no-op for whoever comes second — ranks of one job take the lock in turn) rather than turn every edit into a RuntimeError.
                # NEVER under a profiler: rocprofv3's preloaded tool library initialises the GPU in every process that inherits its environment, and hipcc is a
                # chain of exec hops (sh -c, clang, lld) — a GPU-initialised process replacing its program takes the whole box down on this pool.  There the
                # loud RuntimeError below stays; profiles/collect.sh and the exp/pmc_*.sh scripts build before their first rocprofv3 line.
                import fcntl
                with open(os.path.join(_HERE, "csrc", ".build.lock"), "w") as lk:
                    fcntl.flock(lk, fcntl.LOCK_EX)
                    _build.build_hip(verbose=True)
            if _build.stale("hip"):   # a library that travelled with the tree but was built from other sources must not be what gets tested or measured
                raise RuntimeError("libmor_hip.so was built from other sources or flags than the ones in this tree (carries %r, tree is %r): run `python -m dynamicslamtool_amd.build`"
                                   % (_build.built_hash(LIB_PATH), _build.source_hash(*_build._targets()["hip"][1:])))
        # one hardware queue per stage stream of the frame pipeline (csrc/mor_engine.cpp): the application's choice, made
        # here for the test / bench processes before HIP initialises; a value the caller has set is left alone
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
        L = C.CDLL(LIB_PATH)
        vp, i32, u64 = C.c_void_p, C.c_int, C.c_uint64
        L.mor_sizeof_params.restype = C.c_size_t
        L.mor_last_error.restype = C.c_char_p
        L.mor_batch_create.restype = vp
        L.mor_batch_create.argtypes = [vp, i32, i32, i32, u64, i32, vp]
        L.mor_batch_destroy.argtypes = [vp]
        L.mor_batch_streams.argtypes = [vp]
        L.mor_push_batch.argtypes = [vp, vp, vp]
        L.mor_filter_batch.argtypes = [vp, vp, i32, vp]
        if hasattr(L, "mor_filter_batch_ex"):   # (absent from libraries of earlier rounds loaded through MOR_HIP_LIB for an A/B)
            L.mor_filter_batch_ex.argtypes = [vp, vp, i32, vp, C.c_uint32]
        L.mor_batch_set_async.argtypes = [vp, i32]
        L.mor_batch_wait.argtypes = [vp]
        L.mor_get_output_device.restype = vp
        L.mor_get_output_device.argtypes = [vp, i32, vp]
        L.mor_create.restype = vp
        L.mor_create.argtypes = [vp, i32, i32, u64, i32, vp]
        L.mor_push.argtypes = [vp, vp, u64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp]
        L.mor_filter.argtypes = [vp, vp, vp]
        L.mor_destroy.argtypes = [vp]
        for n in ("mor_get_counts", "mor_get_labels", "mor_get_ground_indices", "mor_get_centroids", "mor_get_detection", "mor_get_boxes", "mor_get_cluster_collection"):
            getattr(L, n).argtypes = [vp, i32, vp]
        L.mor_get_clusters.argtypes = [vp, i32, vp, vp]
        L.mor_get_correspondences.argtypes = [vp, i32, vp, vp, vp, vp]
        L.mor_get_tracks.argtypes = [vp, i32, vp, vp, vp]
        L.mor_get_stage_counts.argtypes = [vp, i32, vp, i32]
        L.mor_get_boxes.argtypes = [vp, i32, vp, vp]
        L.mor_get_markers.argtypes = [vp, i32, vp, vp]
        L.mor_get_moving_clusters.argtypes = [vp, i32, vp, vp]
        L.mor_device_alloc.restype = vp
        L.mor_device_alloc.argtypes = [i32, C.c_size_t]
        L.mor_device_free.argtypes = [i32, vp]
        L.mor_host_alloc.restype = vp
        L.mor_host_alloc.argtypes = [C.c_size_t]
        L.mor_host_free.argtypes = [vp]
        if hasattr(L, "mor_host_register"):
            L.mor_host_register.argtypes = [vp, C.c_size_t, vp]
            L.mor_host_unregister.argtypes = [vp]
        L.mor_device_upload.argtypes = [i32, vp, vp, C.c_size_t]
        L.mor_device_download.argtypes = [i32, vp, vp, C.c_size_t]
        L.mor_device_synchronize.argtypes = [i32]
        L.mor_get_last_timing.argtypes = [vp, vp, vp]
        L.mor_kernel_timing_enable.argtypes = [vp, i32]
        L.mor_kernel_timing_read.argtypes = [vp, i32, vp, C.c_size_t, vp, vp, i32]
        L.mor_get_frame_log.argtypes = [vp, u64, i32, vp]
        _LIB = L
    return _LIB


class MorError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise MorError("rc=%d: %s" % (rc, lib().mor_last_error().decode()))


def build_hash():
    """The sources + flags hash the loaded library carries (mor_build_hash): bench.py puts it into its record, so a library loaded through MOR_HIP_LIB /
    MOR_ALLOW_STALE_LIB — which bypass the staleness check — is named in what it measured."""
    L = lib()
    L.mor_build_hash.restype = C.c_char_p
    return (L.mor_build_hash() or b"").decode()


def device_count():
    return int(lib().mor_device_count())


def device_numa_node(device=0):
    """NUMA node the HIP device is attached to (−1: unknown)."""
    return int(lib().mor_device_numa_node(int(device)))


def bind_thread_to_device_node(device=0, share_index=0, share_count=1):
    """Restricts the calling thread to the CPUs of the device's NUMA node (a slice of them when several processes share the node);
    returns the number of CPUs kept, 0 when nothing was changed.  Create the batch and enqueue its frames from that thread."""
    return int(lib().mor_bind_thread_to_device_node(int(device), int(share_index), int(share_count)))


class DeviceBuffer:
    """A plain hipMalloc'd buffer (clouds kept resident in HBM for the bench / replay driver)."""

    def __init__(self, nbytes, device=0):
        self.device, self.nbytes = device, int(nbytes)
        self.ptr = lib().mor_device_alloc(device, self.nbytes)
        if not self.ptr:
            raise MorError(lib().mor_last_error().decode())

    def upload(self, arr, offset=0):
        a = np.ascontiguousarray(arr)
        assert offset + a.nbytes <= self.nbytes
        _check(lib().mor_device_upload(self.device, self.ptr + offset, a.ctypes.data, a.nbytes))

    def download(self, nbytes=None, offset=0, dtype=np.uint8):
        nbytes = self.nbytes - offset if nbytes is None else int(nbytes)
        out = np.empty(nbytes, np.uint8)
        _check(lib().mor_device_download(self.device, out.ctypes.data, self.ptr + offset, nbytes))
        return out.view(dtype)

    def free(self):
        if self.ptr:
            lib().mor_device_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class HostBuffer:
    """Page-locked host memory viewed as a numpy array (PCIe transfers to and from it run at DMA rate)."""

    def __init__(self, shape, dtype=np.float32):
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self.ptr = lib().mor_host_alloc(max(self.nbytes, 1))
        if not self.ptr:
            raise MorError(lib().mor_last_error().decode())
        self.array = np.ctypeslib.as_array((C.c_uint8 * max(self.nbytes, 1)).from_address(self.ptr))[: self.nbytes].view(dtype).reshape(shape)

    def free(self):
        if self.ptr:
            self.array = None
            lib().mor_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class MorBatch:
    """B independent MovingObjectRemoval streams on one MI355X (mor_batch)."""

    def __init__(self, params, n_streams=1, max_points=131072, n_bad=4, n_good=3, device=0):
        L = lib()
        assert L.mor_sizeof_params() == C.sizeof(MorParams)
        self.params, self.B, self.max_points, self.device = params, int(n_streams), int(max_points), device
        err = C.c_int(0)
        self._h = L.mor_batch_create(C.addressof(params), n_bad, n_good, self.B, self.max_points, device, C.addressof(err))
        if not self._h:
            raise MorError("mor_batch_create failed (rc=%d): %s" % (err.value, L.mor_last_error().decode()))
        self._views = (CloudView * self.B)()
        self._keep = None

    def close(self):
        if getattr(self, "_h", None):
            lib().mor_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- pushRawCloudAndPose for all streams
    def push(self, clouds, poses, point_step=16, offsets=(0, 4, 8, 12)):
        """clouds: list of B float32 [N,4] arrays (host) — or of (DeviceBuffer|int ptr, n_points) tuples
        for device-resident blobs.  poses: [B,7] float64."""
        assert len(clouds) == self.B
        keep = []
        for s, c in enumerate(clouds):
            v = self._views[s]
            v.point_step, v.off_x, v.off_y, v.off_z, v.off_intensity = point_step, offsets[0], offsets[1], offsets[2], offsets[3]
            if isinstance(c, tuple):
                buf, n = c
                v.data, v.n_points, v.on_device = (buf.ptr if isinstance(buf, DeviceBuffer) else int(buf)), int(n), 1
            else:
                a = np.ascontiguousarray(c)
                keep.append(a)
                v.data, v.n_points, v.on_device = a.ctypes.data, a.nbytes // point_step, 0
        poses = np.ascontiguousarray(poses, np.float64).reshape(self.B, 7)
        self._keep = keep
        _check(lib().mor_push_batch(self._h, C.addressof(self._views), poses.ctypes.data))

    def make_views(self, clouds, point_step=16, offsets=(0, 4, 8, 12)):
        """Pre-builds the mor_cloud_view array for device-resident clouds [(ptr|DeviceBuffer, n_points)] × B, so a
        replay loop pays no per-step Python marshalling (push_views)."""
        v = (CloudView * self.B)()
        for s, (buf, n) in enumerate(clouds):
            v[s].data, v[s].n_points, v[s].on_device = (buf.ptr if isinstance(buf, DeviceBuffer) else int(buf)), int(n), 1
            v[s].point_step, v[s].off_x, v[s].off_y, v[s].off_z, v[s].off_intensity = point_step, offsets[0], offsets[1], offsets[2], offsets[3]
        return v

    def make_host_views(self, arrays, point_step=16, offsets=(0, 4, 8, 12)):
        """The same for host-resident clouds (page-locked arrays, e.g. HostBuffer.array rows): staged by the library, beside the kernels of
        the frames in flight in asynchronous mode.  The arrays must stay alive and unchanged until their push has been waited for."""
        v = (CloudView * self.B)()
        for s, a in enumerate(arrays):
            v[s].data, v[s].n_points, v[s].on_device = a.ctypes.data, a.nbytes // point_step, 0
            v[s].point_step, v[s].off_x, v[s].off_y, v[s].off_z, v[s].off_intensity = point_step, offsets[0], offsets[1], offsets[2], offsets[3]
        return v

    def make_out_pointers(self, arrays):
        """Pointer table for filter_async_to: one output array per stream (device memory, or page-locked host memory the GPU writes directly)."""
        return (C.c_void_p * self.B)(*[a if isinstance(a, int) else a.ctypes.data for a in arrays])

    def filter_async_to(self, out_pointers, on_device=True):
        """filterCloud for all streams, enqueue only: every stream's filtered cloud goes to its pointer — device memory (written by the
        output kernel), or with on_device=False page-locked host memory (assembled on the device, carried out by DMA behind the kernels;
        each buffer holds the stream's input point count)."""
        _check(lib().mor_filter_batch(self._h, C.addressof(out_pointers), 1 if on_device else 0, None))

    def push_views(self, views, poses):
        poses = np.ascontiguousarray(poses, np.float64)
        _check(lib().mor_push_batch(self._h, C.addressof(views), poses.ctypes.data))

    def set_async(self, on=True):
        _check(lib().mor_batch_set_async(self._h, 1 if on else 0))

    def wait(self):
        _check(lib().mor_batch_wait(self._h))

    def filter_async(self):
        """filterCloud for all streams, enqueue only (async mode): results stay in the batch's device buffers."""
        _check(lib().mor_filter_batch(self._h, None, 0, None))

    def filter_device(self):
        """filterCloud for all streams, results left in the batch's device buffers; returns the n_out array."""
        if not hasattr(self, "_nout"):
            self._nout = (C.c_uint64 * self.B)()
        _check(lib().mor_filter_batch(self._h, None, 0, C.addressof(self._nout)))
        return self._nout

    # ---- filterCloud for all streams
    def filter(self, to_host=True):
        n_out = (C.c_uint64 * self.B)()
        if not to_host:
            _check(lib().mor_filter_batch(self._h, None, 0, C.addressof(n_out)))
            return [int(x) for x in n_out]
        outs = [np.empty((max(int(self._views[s].n_points), 1), 4), np.float32) for s in range(self.B)]
        ptrs = (C.c_void_p * self.B)(*[o.ctypes.data for o in outs])
        _check(lib().mor_filter_batch(self._h, C.addressof(ptrs), 0, C.addressof(n_out)))
        return [outs[s][: int(n_out[s])] for s in range(self.B)]

    def filter_into(self, outs):
        """filterCloud into caller-provided host arrays (e.g. HostBuffer.array), one per stream, each with room for the
        stream's input point count; returns the point counts."""
        n_out = (C.c_uint64 * self.B)()
        ptrs = (C.c_void_p * self.B)(*[o.ctypes.data for o in outs])
        _check(lib().mor_filter_batch(self._h, C.addressof(ptrs), 0, C.addressof(n_out)))
        return [int(x) for x in n_out]

    def filter_records32(self, outs):
        """filterCloud with the DEVICE writing PCL's 32-byte PointXYZI records (mor_filter_batch_ex, out_point_step = 32) straight into device-accessible
        memory: one page-locked array per stream (HostBuffer.array, room for the stream's input point count x 32 bytes); returns the point counts."""
        n_out = (C.c_uint64 * self.B)()
        ptrs = (C.c_void_p * self.B)(*[o if isinstance(o, int) else o.ctypes.data for o in outs])
        _check(lib().mor_filter_batch_ex(self._h, C.addressof(ptrs), 1, C.addressof(n_out), 32))
        return [int(x) for x in n_out]

    # ---- read-backs
    def counts(self, s=0):
        c = Counts()
        _check(lib().mor_get_counts(self._h, s, C.addressof(c)))
        return c

    def _get(self, fn, s, n, dtype, width=1):
        a = np.empty((max(int(n), 1), width) if width > 1 else max(int(n), 1), dtype)
        _check(fn(self._h, s, a.ctypes.data))
        return a[: int(n)]

    def labels(self, s=0):
        return self._get(lib().mor_get_labels, s, self.counts(s).n_trim, np.int32)

    def ground_indices(self, s=0):
        return self._get(lib().mor_get_ground_indices, s, self.counts(s).n_ground, np.int32)

    def clusters(self, s=0):
        c = self.counts(s)
        off = np.zeros(c.n_clusters + 1, np.int32)
        idx = np.empty(max(int(c.n_clustered), 1), np.int32)
        _check(lib().mor_get_clusters(self._h, s, off.ctypes.data, idx.ctypes.data))
        return off, idx[: c.n_clustered]

    def centroids(self, s=0):
        return self._get(lib().mor_get_centroids, s, self.counts(s).n_clusters, np.float32, 3)

    def boxes(self, s=0):
        """(min[K,3], max[K,3]) of the clusters' points (getMinMax3D)."""
        K = self.counts(s).n_clusters
        lo, hi = np.empty((max(K, 1), 3), np.float32), np.empty((max(K, 1), 3), np.float32)
        _check(lib().mor_get_boxes(self._h, s, lo.ctypes.data, hi.ctypes.data))
        return lo[:K], hi[:K]

    def markers(self, s=0):
        """Data of the reference's debug markers (mark_cluster, :7-58) per cluster: (position xyz = float-accumulated
        centroid, scale xyz = box extent, zero extents 0.1)."""
        K = self.counts(s).n_clusters
        pos, scale = np.empty((max(K, 1), 3), np.float32), np.empty((max(K, 1), 3), np.float32)
        _check(lib().mor_get_markers(self._h, s, pos.ctypes.data, scale.ctypes.data))
        return pos[:K], scale[:K]

    def moving_clusters(self, s=0):
        """Cluster index per tracked centroid the latest filterCloud visited, in mo_vec order (the reference's bounding-box markers, :641)."""
        n = C.c_uint32(0)
        _check(lib().mor_get_moving_clusters(self._h, s, None, C.byref(n)))
        out = np.empty(max(int(n.value), 1), np.int32)
        _check(lib().mor_get_moving_clusters(self._h, s, out.ctypes.data, C.byref(n)))
        return out[:int(n.value)]

    def detection(self, s=0):
        return self._get(lib().mor_get_detection, s, self.counts(s).n_clusters, np.uint8)

    def cluster_collection(self, s=0):
        return self._get(lib().mor_get_cluster_collection, s, self.counts(s).n_clustered, np.float32, 4)

    def correspondences(self, s=0):
        k = int(self.counts(s).n_corr)
        n = max(k, 1)
        q, m, d, sc = np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.float32), np.empty(n, np.float64)
        _check(lib().mor_get_correspondences(self._h, s, q.ctypes.data, m.ctypes.data, d.ctypes.data, sc.ctypes.data))
        return q[:k], m[:k], d[:k], sc[:k]

    def tracks(self, s=0):
        k = int(self.counts(s).n_tracks)
        n = max(k, 1)
        xyz, conf, mx = np.empty((n, 3), np.float32), np.empty(n, np.int32), np.empty(n, np.int32)
        _check(lib().mor_get_tracks(self._h, s, xyz.ctypes.data, conf.ctypes.data, mx.ctypes.data))
        return xyz[:k], conf[:k], mx[:k]

    def frame_log(self, frame, s=0):
        """Summary of frame `frame` (0-based push index) of stream s: dict of the mor_get_frame_log fields."""
        a = (C.c_int64 * 10)()
        _check(lib().mor_get_frame_log(self._h, int(frame), s, a))
        return dict(zip(("frame", "K", "C", "n_pairs", "cnt_sum", "det_sum", "n_mo_push", "n_mo_filter", "n_out", "flags"), [int(x) for x in a]))

    def debug_read(self, name, s=0, dtype=np.int32, count=None):
        """Development read-back of an intermediate device array (mor_debug_read): `count` elements of dtype."""
        L = lib()
        L.mor_debug_read.restype = C.c_longlong
        L.mor_debug_read.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_size_t]
        n = int(count if count is not None else self.max_points)
        out = np.zeros(max(n, 1), dtype)
        got = L.mor_debug_read(self._h, name.encode(), s, out.ctypes.data, out.nbytes)
        if got < 0:
            raise MorError(L.mor_last_error().decode())
        return out[:n]

    def debug_config(self):
        a = (C.c_int * 12)()
        lib().mor_debug_config.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        lib().mor_debug_config(self._h, a, 12)
        return dict(zip(("nx", "ny", "nz", "nrows", "P", "grid_mode", "cg_mode", "Hcell", "Kcap", "tiles_m", "cur", "prev"), [int(x) for x in a]))

    def stage_counts(self, s=0):
        a = (C.c_uint32 * 6)()
        _check(lib().mor_get_stage_counts(self._h, s, a, 6))
        return {"n_occ": int(a[0]), "n_defer": int(a[1]), "n_tier1b": int(a[2]), "C_prev": int(a[3]), "g2_exact": int(a[4]), "max_loc": int(a[5])}

    def output_device(self, s=0):
        n = C.c_uint64(0)
        p = lib().mor_get_output_device(self._h, s, C.addressof(n))
        return p, int(n.value)

    # ---- timing
    def last_timing(self):
        a, b = C.c_float(0), C.c_float(0)
        lib().mor_get_last_timing(self._h, C.addressof(a), C.addressof(b))
        return a.value, b.value

    def kernel_timing_enable(self, on=True):
        lib().mor_kernel_timing_enable(self._h, 1 if on else 0)

    def kernel_timing(self, reset=True):
        names = C.create_string_buffer(2048)
        ms = (C.c_float * 64)()
        ln = (C.c_uint32 * 64)()
        n = lib().mor_kernel_timing_read(self._h, 1 if reset else 0, names, 2048, ms, ln, 64)
        nm = names.value.decode().split(",")
        return {nm[i]: (float(ms[i]), int(ln[i])) for i in range(n)}

    def kernel_timeline(self, max_n=200000):
        """[(kernel name, start ms, end ms)] of the last timed leg (call after kernel_timing())."""
        names = C.create_string_buffer(2048)
        lib().mor_kernel_timing_read(self._h, 0, names, 2048, None, None, 64)
        nm = names.value.decode().split(",")
        ids, t0, t1 = (C.c_int * max_n)(), (C.c_float * max_n)(), (C.c_float * max_n)()
        lib().mor_kernel_timeline_read.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        n = lib().mor_kernel_timeline_read(self._h, ids, t0, t1, max_n)
        return [(nm[ids[i]], float(t0[i]), float(t1[i])) for i in range(n)]

    def synchronize(self):
        _check(lib().mor_device_synchronize(self.device))
