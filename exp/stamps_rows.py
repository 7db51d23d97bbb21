import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MOR_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmor_stamps.so")
from dynamicslamtool_amd import engine, kitti_params, synth
B = 64
p = kitti_params(1)
b = engine.MorBatch(p, B, 120000)
L = engine.lib(); L.mor_exp_read_stamps.argtypes = [C.c_void_p, C.c_void_p]
out = np.zeros((B, 16), np.uint64)
for f in range(5):
    xs, ps = synth.batch([2000 + s for s in range(B)], [f] * B)
    if f == 4: L.mor_exp_read_stamps(b._h, out.ctypes.data)
    b.push(list(xs), ps); b.filter(to_host=False)
L.mor_exp_read_stamps(b._h, out.ctypes.data)
o = out.astype(np.float64)
n = o[:, 12].sum()
print("queries", n, "E2", o[:, 9].sum(), "E1", o[:, 11].sum())
print("avg header us %.2f" % (o[:, 7].sum() / n / 100))
print("avg E2 us %.2f" % (o[:, 8].sum() / max(o[:, 9].sum(), 1) / 100))
print("avg E1 us %.2f" % (o[:, 10].sum() / max(o[:, 11].sum(), 1) / 100))
print("avg E1 points scanned %.1f" % (o[:, 13].sum() / max(o[:, 11].sum(), 1)))
print("max thread total us per stream: mean %.1f max %.1f" % (o[:, 14].mean() / 100, o[:, 14].max() / 100))
print("per-stream nq: mean %.0f max %.0f" % (o[:, 12].mean(), o[:, 12].max()))
