"""Item statistics of the cell graph's enumeration (k_cg_slab phase A), first frame only (no pair stage: the scoring kernels
share the counter slots).  Library built with -DMOR_EXP_STAMPS as exp/libmor_stamps.so."""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MOR_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmor_stamps.so")
from dynamicslamtool_amd import engine, kitti_params, synth
B = 64
W = sys.argv[1] if len(sys.argv) > 1 else "hdl64"
N = {"hdl64": 120000, "os128": 262144}[W]
b = engine.MorBatch(kitti_params(1), B, N)
L = engine.lib(); L.mor_exp_read_stamps.argtypes = [C.c_void_p, C.c_void_p]
out = np.zeros((B, 16), np.uint64)
xs, ps = synth.batch([2000 + s for s in range(B)], [0] * B, sensor=W)
L.mor_exp_read_stamps(b._h, out.ctypes.data)   # reset
b.push(list(xs), ps); b.filter(to_host=False)
L.mor_exp_read_stamps(b._h, out.ctypes.data)
c = out.astype(np.float64).sum(0)
print("items (cell, row) %d, with a non-empty window start %d (%.1f %%), queued pairs %d (%.2f per item, %.1f per batch of 64), batches %d" % (c[0], c[1], 100 * c[1] / c[0], c[2], c[2] / c[0], c[2] / c[3], c[3]))
print("per wave, us: A1 lookup (decode, row, lower bound, find) %.1f | window loads + queue %.1f | A2 %.1f   (waves %d)" % (c[4] / c[7] / 100, c[5] / c[7] / 100, c[6] / c[7] / 100, c[7]))
