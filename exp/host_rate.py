"""Host-side cost of enqueueing one step (push + filter) in asynchronous mode: per-call times of the first steps after an
idle GPU (the pinned argument ring of 8 slots throttles the host from the 9th step on) next to the steady-state period."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicslamtool_amd import engine, kitti_params, synth
B, npts, nf = 64, 120000, 12
p = kitti_params(1)
seeds = [2000 + s for s in range(B)]
buf = engine.DeviceBuffer(nf * B * npts * 16); poses = np.empty((nf, B, 7))
for f in range(nf):
    xs, ps = synth.batch(seeds, [f] * B); buf.upload(xs, f * B * npts * 16); poses[f] = ps
b = engine.MorBatch(p, B, npts)
views = [b.make_views([(buf.ptr + (f * B + s) * npts * 16, npts) for s in range(B)]) for f in range(nf)]
def fr(i):
    k = i % (2 * (nf - 1)); return k if k < nf else 2 * (nf - 1) - k
for i in range(4): b.push_views(views[fr(i)], poses[fr(i)]); b.filter_device()
b.set_async(True)
for rep in range(3):
    b.wait(); ts = []
    t0 = time.perf_counter()
    for i in range(4, 4 + 40):
        ta = time.perf_counter(); b.push_views(views[fr(i)], poses[fr(i)]); tb = time.perf_counter(); b.filter_async(); tc = time.perf_counter()
        ts.append((tb - ta, tc - tb))
    t1 = time.perf_counter(); b.wait(); t2 = time.perf_counter()
    print("first 6 steps push us:", [round(1e6 * a) for a, _ in ts[:6]], "filter us:", [round(1e6 * f) for _, f in ts[:6]])
    print("   steps 20-40: host per step %.1f us | whole run per step %.1f us" % (1e6 * sum(a + f for a, f in ts[20:]) / 20, 1e6 * (t2 - t0) / 40))
