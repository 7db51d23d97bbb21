"""Phase times inside k_gridhash and k_cg_slab (library built with -DMOR_EXP_STAMPS as exp/libmor_stamps.so)."""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MOR_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmor_stamps.so")
from dynamicslamtool_amd import engine, kitti_params, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sensor = sys.argv[2] if len(sys.argv) > 2 else "hdl64"
MAXP = 32
p = kitti_params(1)
if len(sys.argv) > 3: p.ground_method = int(sys.argv[3])   # 1: the voxel-covariance ground removal
b = engine.MorBatch(p, B, synth.n_points(sensor))
L = engine.lib(); L.mor_exp_read_stamps2.argtypes = [C.c_void_p, C.c_void_p]
out = np.zeros((B, MAXP + 2, 16), np.uint64)
seeds = [2000 + s for s in range(B)]
for f in range(4):
    xs, ps = synth.batch(seeds, [f] * B, sensor)
    if f == 3: L.mor_exp_read_stamps2(b._h, out.ctypes.data)
    b.push(list(xs), ps); b.filter(to_host=False)
L.mor_exp_read_stamps2(b._h, out.ctypes.data)
P = b.debug_config()["P"]
print("P =", P)
g = out[:, MAXP, :].astype(np.int64)
names = ["entries sweep", "row counts", "row scan + copy + slabs", "row lists", "compact ids", "counts + scan", "cstart + init", "entries pass 2"]
cols = [0, 1, 6, 7, 8, 10, 11, 2, 3]
ph = np.diff(g[:, cols], axis=1) / 100.0
tot2 = ph.sum(1)
i2 = int(np.argmax(tot2))
print("k_gridhash (merge) phases, us: mean over streams | slowest stream %d (M %d n_occ %d)" % (i2, g[i2, 4], g[i2, 5]))
for n, col in zip(names, range(ph.shape[1])):
    print("   %-26s %7.1f | %7.1f" % (n, ph[:, col].mean(), ph[i2, col]))
print("   %-26s %7.1f | %7.1f" % ("total", tot2.mean(), tot2[i2]))
t = np.diff(g[:, 0:4], axis=1) / 100.0
print("k_gridhash per stream (us): sweep1 mean %.1f max %.1f | cells/rank/scan mean %.1f max %.1f | sweep2 mean %.1f max %.1f | total mean %.1f max %.1f" % (
    t[:, 0].mean(), t[:, 0].max(), t[:, 1].mean(), t[:, 1].max(), t[:, 2].mean(), t[:, 2].max(), t.sum(1).mean(), t.sum(1).max()))
i = int(np.argmax(t.sum(1)))
print("   slowest stream %d: M %d n_occ %d phases %s; kernel span %.1f us" % (i, g[i, 4], g[i, 5], np.round(t[i], 1).tolist(), (g[:, 3].max() - g[:, 0].min()) / 100.0))
order = np.argsort(-t.sum(1))[:6]
print("   six slowest: " + ", ".join("s%d M=%d %.0fus" % (k, g[k, 4], t[k].sum()) for k in order))
w = out[:, :MAXP, :].astype(np.int64).reshape(-1, 16)   # (slabs per stream follow the cell counts: up to MAXP)
w = w[w[:, 0] > 0]
tot = (w[:, 9] - w[:, 0]) / 100.0
ph = {"load": w[:, 1] - w[:, 0], "A": w[:, 2] - w[:, 1], "B1": w[:, 3] - w[:, 2], "B2": w[:, 4] - w[:, 3], "out": w[:, 9] - w[:, 4]}
print("k_cg_slab workgroups: %d, total us mean %.1f p50 %.1f p90 %.1f max %.1f; kernel span %.1f" % (len(w), tot.mean(), np.median(tot), np.percentile(tot, 90), tot.max(), (w[:, 9].max() - w[:, 0].min()) / 100.0))
for k, v in ph.items():
    v = v / 100.0
    print("   %-5s mean %7.1f p90 %7.1f max %7.1f" % (k, v.mean(), np.percentile(v, 90), v.max()))
print("   pairs listed mean %.0f max %d | for waves mean %.1f max %d | n_own mean %.0f max %d n_loc max %d" % (w[:, 10].mean(), w[:, 10].max(), w[:, 11].mean(), w[:, 11].max(), w[:, 14].mean(), w[:, 14].max(), w[:, 15].max()))
for k in np.argsort(-tot)[:8]:
    print("   slow wg: total %.0f us  " % tot[k] + " ".join("%s=%.0f" % (n, v[k] / 100.0) for n, v in ph.items()) + "  n1=%d n2=%d own=%d loc=%d" % (w[k, 10], w[k, 11], w[k, 14], w[k, 15]))
# round 4: where the slow slab workgroups sit (HW_ID of the recording wave: bits 8-11 CU, 13-15 SE (XCC id is not in it), and their row counts
hw = w[:, 13]; cu = (hw >> 8) & 15; se = (hw >> 13) & 7
print("   total us histogram (10-us bins from 20):", np.histogram(tot, bins=[0, 20, 30, 40, 50, 60, 70, 80, 100, 120, 200])[0].tolist())
for k in np.argsort(-tot)[:12]:
    print("   slow wg: %.0f us A=%.0f rows=%d own=%d loc=%d n1=%d  se=%d cu=%d start=%.0f us" % (tot[k], ph["A"][k] / 100.0, w[k, 12], w[k, 14], w[k, 15], w[k, 10], se[k], cu[k], (w[k, 0] - w[:, 0].min()) / 100.0))
print("   start times (us after the first): p50 %.0f p90 %.0f max %.0f" % tuple(np.percentile((w[:, 0] - w[:, 0].min()) / 100.0, [50, 90, 100])))
