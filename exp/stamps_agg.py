import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MOR_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmor_stamps.so")
from dynamicslamtool_amd import engine, kitti_params, synth
B = 32; sensor = sys.argv[1] if len(sys.argv) > 1 else "agg10"
p = kitti_params(1)
npts = synth.n_points(sensor)
b = engine.MorBatch(p, B, npts)
L = engine.lib(); L.mor_exp_read_stamps.argtypes = [C.c_void_p, C.c_void_p]
out = np.zeros((B, 16), np.uint64)
names = ["load", "boxes", "near", "shell", "comp", "select+rank+off"]
for f in range(10):
    xs, ps = synth.batch([2000 + s for s in range(B)], [f] * B, sensor)
    L.mor_exp_read_stamps(b._h, out.ctypes.data)
    b.push(list(xs), ps); b.filter(to_host=False)
    L.mor_exp_read_stamps(b._h, out.ctypes.data)
    d = np.diff(out[:, :7].astype(np.int64), axis=1) / 100.0
    nocc = np.array([b.stage_counts(s_)["n_occ"] for s_ in range(B)])
    M = np.array([b.counts(s_).n_cloud for s_ in range(B)])
    w = int(np.argmax(d.sum(1)))
    print("frame %d: span %.0f us; slowest stream %d: n_occ %d M %d phases %s | n_occ max %d" % (f, (out[:, 6].max() - out[:, 0].min()) / 100.0, w, nocc[w], M[w], np.round(d[w]).astype(int), nocc.max()))
