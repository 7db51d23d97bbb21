import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MOR_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmor_stamps.so")
from dynamicslamtool_amd import engine, kitti_params, synth
B = 64; sensor = sys.argv[1] if len(sys.argv) > 1 else "hdl64"
p = kitti_params(1)
b = engine.MorBatch(p, B, synth.n_points(sensor))
L = engine.lib(); L.mor_exp_read_stamps.argtypes = [C.c_void_p, C.c_void_p]
out = np.zeros((B, 16), np.uint64)
for f in range(4):
    xs, ps = synth.batch([2000 + s for s in range(B)], [f] * B, sensor)
    if f == 3: L.mor_exp_read_stamps(b._h, out.ctypes.data)
    b.push(list(xs), ps); b.filter(to_host=False)
L.mor_exp_read_stamps(b._h, out.ctypes.data)
z = out[:, 0:4].astype(np.float64).sum(0)
print("cell graph A phase: (cell,row) items %d, with a non-empty row window %d (%.0f%%), neighbour pairs queued %d (%.2f per item), wave batches %d" % (z[0], z[1], 100 * z[1] / z[0], z[2], z[2] / z[0], z[3]))
