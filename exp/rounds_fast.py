"""Rounds of the sampled own-cell scan in k_score_fast, per lane and per wave (library built with -DMOR_EXP_ROUNDS as exp/libmor_rounds.so)."""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MOR_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmor_rounds.so")
from dynamicslamtool_amd import engine, kitti_params, synth
B = 64
b = engine.MorBatch(kitti_params(1), B, 120000)
L = engine.lib(); L.mor_exp_read_stamps.argtypes = [C.c_void_p, C.c_void_p]
out = np.zeros((B, 16), np.uint64)
for f in range(4):
    xs, ps = synth.batch([2000 + s for s in range(B)], [f] * B)
    if f == 3: L.mor_exp_read_stamps(b._h, out.ctypes.data)
    b.push(list(xs), ps); b.filter(to_host=False)
L.mor_exp_read_stamps(b._h, out.ctypes.data)
c = out.astype(np.float64).sum(0)
print("queries %d, rounds per query %.2f; waves %d, slowest lane's rounds per wave %.2f; found within lb by round 1: %.1f %% of the queries; queries in cells > 64 points: %.1f %%" % (c[0], c[1] / c[0], c[2], c[3] / c[2], 100 * c[13] / c[0], 100 * c[14] / c[0]))
print("waves by their slowest lane's rounds 0..8:", [int(x) for x in c[4:13]])
