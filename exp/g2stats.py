"""Phase times of k_g2_cov (voxel ground variant), first frame.  Library built with -DMOR_EXP_STAMPS as exp/libmor_stamps.so."""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MOR_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmor_stamps.so")
from dynamicslamtool_amd import engine, kitti_params, synth
B = 64
p = kitti_params(1); p.ground_method = 1
b = engine.MorBatch(p, B, 120000)
L = engine.lib(); L.mor_exp_read_stamps.argtypes = [C.c_void_p, C.c_void_p]
out = np.zeros((B, 16), np.uint64)
xs, ps = synth.batch([2000 + s for s in range(B)], [0] * B)
L.mor_exp_read_stamps(b._h, out.ctypes.data)
b.push(list(xs), ps); b.filter(to_host=False)
L.mor_exp_read_stamps(b._h, out.ctypes.data)
c = out.astype(np.float64).sum(0)
print("wave rounds %d (four voxels each); per round, us: centroid %.2f | rows %.2f | walk %.2f | reduce + verdict %.2f; candidates of the round's largest voxel %.1f" % (c[4], c[0] / c[4] / 100, c[1] / c[4] / 100, c[2] / c[4] / 100, c[3] / c[4] / 100, c[5] / c[4]))
print("voxels %d (%.0f per stream), neighbours %.1f, candidates %.1f per voxel, wide %d" % (c[6], c[6] / B, c[7] / c[6], c[8] / c[6], c[9]))
