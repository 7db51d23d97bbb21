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
g0 = (~out[:, 8]).min() ; g1 = out[:, 9].max()
print("kernel span (first wave start → last wave end) %.1f us" % ((int(g1) - int(g0)) / 100.0))
st = (~out[:, 8]).astype(np.int64) - int(g0); en = out[:, 9].astype(np.int64) - int(g0)
print("per stream: first start us mean %.1f max %.1f; last end us mean %.1f max %.1f" % (st.mean() / 100, st.max() / 100, en.mean() / 100, en.max() / 100))
n = out[:, 11].astype(np.float64)
print("waves/stream %.0f; avg wave dur us %.2f; max wave dur us: mean %.1f max %.1f" % (n.mean(), (out[:, 10] / np.maximum(n, 1)).mean() / 100, out[:, 12].mean() / 100, out[:, 12].max() / 100))
