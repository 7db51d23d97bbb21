"""Phase times inside k_score_fast per wave (library built with -DMOR_EXP_STAMPS as exp/libmor_stamps.so)."""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MOR_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmor_stamps.so")
from dynamicslamtool_amd import engine, kitti_params, synth
B = 64
p = kitti_params(1)
b = engine.MorBatch(p, B, 120000)
L = engine.lib(); L.mor_exp_read_stamps.argtypes = [C.c_void_p, C.c_void_p]
out = np.zeros((B, 16), np.uint64)
for f in range(4):
    xs, ps = synth.batch([2000 + s for s in range(B)], [f] * B)
    if f == 3: L.mor_exp_read_stamps(b._h, out.ctypes.data)
    b.push(list(xs), ps); b.filter(to_host=False)
L.mor_exp_read_stamps(b._h, out.ctypes.data)
c = out[:, 8:16].astype(np.float64)
nw = c[:, 0].sum()
print("waves with queries %d; per wave us: chain to cell record %.2f | scan + classify %.2f | pushes %.2f | slowest wave %.1f" % (
    nw, c[:, 1].sum() / nw / 100, c[:, 2].sum() / nw / 100, c[:, 3].sum() / nw / 100, c[:, 4].max() / 100))
print("tier 1 → near %d block %d big(wave tier) %d" % (c[:, 5].sum(), c[:, 6].sum(), c[:, 7].sum()))
z = out[:, 0:4].astype(np.float64)
print("near → wave tier (budget) %d | block → wave tier: budget %d, E2 open %d (of which no matched cell in the block %d)" % (z[:, 0].sum(), z[:, 1].sum(), z[:, 2].sum(), z[:, 3].sum()))
z = out[:, 4:8].astype(np.float64); nb = z[:, 0].sum()
print("block tier: waves %d; per wave us: entry + q + 27 probes + resolves %.2f | cluster ids + selection %.2f | boxes + ranges + point scans %.2f" % (nb, z[:, 1].sum() / nb / 100, z[:, 2].sum() / nb / 100, z[:, 3].sum() / nb / 100))
