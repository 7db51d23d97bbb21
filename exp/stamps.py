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
d = np.diff(out[:, :7].astype(np.int64), axis=1) / 100.0   # 100 MHz → µs
names = ["load", "boxes", "near", "shell", "comp", "select+rank+off"]
print("per-phase µs: mean / max over streams")
for i, n in enumerate(names): print("  %-16s %8.1f %8.1f" % (n, d[:, i].mean(), d[:, i].max()))
print("total mean %.1f max %.1f; kernel span %.1f" % (d.sum(1).mean(), d.sum(1).max(), (out[:, 6].max() - out[:, 0].min()) / 100.0))

c = out[:, 8:16]
for nm, o in (("near", 0), ("shell", 4)):
    A, B1, B2 = c[:, o].astype(np.float64) / 100, c[:, o + 1].astype(np.float64) / 100, c[:, o + 2].astype(np.float64) / 100
    n1, n2 = (c[:, o + 3] >> np.uint64(32)).astype(np.int64), (c[:, o + 3] & np.uint64(0xffffffff)).astype(np.int64)
    print("%s: A enumeration us mean %.1f max %.1f | B1 thread tests us mean %.1f max %.1f | B2 wave tests us mean %.1f max %.1f | pairs listed mean %.0f max %d | big undecided mean %.1f max %d" % (
        nm, A.mean(), A.max(), B1.mean(), B1.max(), B2.mean(), B2.max(), n1.mean(), n1.max(), n2.mean(), n2.max()))
