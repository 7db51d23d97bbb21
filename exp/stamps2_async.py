"""Slab workgroups of k_cg_slab inside the PIPELINED regime (asynchronous, four lanes): phase times of the last frame."""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MOR_HIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmor_stamps.so")
from dynamicslamtool_amd import engine, kitti_params, synth
B, npts, nf = 64, 120000, 8
p = kitti_params(1)
seeds = [2000 + s for s in range(B)]
buf = engine.DeviceBuffer(nf * B * npts * 16); poses = np.empty((nf, B, 7))
for f in range(nf):
    xs, ps = synth.batch(seeds, [f] * B); buf.upload(xs, f * B * npts * 16); poses[f] = ps
b = engine.MorBatch(p, B, npts)
views = [b.make_views([(buf.ptr + (f * B + s) * npts * 16, npts) for s in range(B)]) for f in range(nf)]
L = engine.lib(); L.mor_exp_read_stamps2.argtypes = [C.c_void_p, C.c_void_p]
MAXP = 32; out = np.zeros((B, MAXP + 2, 16), np.uint64)
def fr(i):
    k = i % (2 * (nf - 1)); return k if k < nf else 2 * (nf - 1) - k
for mode in ("sync", "async"):
    for i in range(4): b.push_views(views[fr(i)], poses[fr(i)]); b.filter_device()
    L.mor_exp_read_stamps2(b._h, out.ctypes.data)
    if mode == "async": b.set_async(True)
    for i in range(4, 24):
        b.push_views(views[fr(i)], poses[fr(i)])
        if mode == "async": b.filter_async()
        else: b.filter_device()
    b.wait(); b.set_async(False)
    L.mor_exp_read_stamps2(b._h, out.ctypes.data)
    P = b.debug_config()["P"]
    w = out[:, :P, :].astype(np.int64).reshape(-1, 16); w = w[w[:, 0] > 0]
    tot = (w[:, 4] - w[:, 0]) / 100.0
    ph = {"load": w[:, 1] - w[:, 0], "A": w[:, 2] - w[:, 1], "B1": w[:, 3] - w[:, 2], "B2": w[:, 4] - w[:, 3]}
    print("%s: slab workgroups %d, total us mean %.1f p50 %.1f p90 %.1f max %.1f; first start to last end %.1f" % (mode, len(w), tot.mean(), np.median(tot), np.percentile(tot, 90), tot.max(), (w[:, 4].max() - w[:, 0].min()) / 100.0))
    print("   " + "  ".join("%s mean %.1f p90 %.1f" % (k, (v / 100.0).mean(), np.percentile(v / 100.0, 90)) for k, v in ph.items()))
    st = (w[:, 0] - w[:, 0].min()) / 100.0
    print("   start times of the workgroups after the first: p50 %.1f p90 %.1f max %.1f us" % (np.median(st), np.percentile(st, 90), st.max()))
