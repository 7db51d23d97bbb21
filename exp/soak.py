"""Long asynchronous run (device-resident clouds, no waits) against a synchronous run of the same length: every one of the last 64 frame
summaries, the tracks and the output sizes must agree; the sticky error word must stay clear.  Usage: soak.py [steps] [voxel]   (voxel: the voxel-covariance ground variant)"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicslamtool_amd import engine, kitti_params, synth
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
B, npts, nf = 64, 120000, 12
p = kitti_params(1)
if len(sys.argv) > 2 and sys.argv[2] == "voxel": p.ground_method = 1
seeds = [2000 + s for s in range(B)]
buf = engine.DeviceBuffer(nf * B * npts * 16); poses = np.empty((nf, B, 7))
for f in range(nf):
    xs, ps = synth.batch(seeds, [f] * B); buf.upload(xs, f * B * npts * 16); poses[f] = ps
def fr(i):
    k = i % (2 * (nf - 1)); return k if k < nf else 2 * (nf - 1) - k
res = []
for mode in ("async", "sync"):
    b = engine.MorBatch(p, B, npts)
    views = [b.make_views([(buf.ptr + (f * B + s) * npts * 16, npts) for s in range(B)]) for f in range(nf)]
    if mode == "async": b.set_async(True)
    t0 = time.perf_counter()
    for i in range(steps):
        b.push_views(views[fr(i)], poses[fr(i)])
        if mode == "async": b.filter_async()
        else: b.filter_device()
    b.wait(); dt = time.perf_counter() - t0
    logs = [[b.frame_log(f, s) for s in range(B)] for f in range(steps - 64, steps)]
    tr = [tuple(np.asarray(x).tobytes() for x in b.tracks(s)) for s in range(B)]
    res.append((logs, tr, [b.output_device(s)[1] for s in range(B)]))
    print("%s: %d steps in %.2f s = %.0f frame-pairs/s; tracks max %d" % (mode, steps, dt, B * steps / dt, max(b.counts(s).n_tracks for s in range(B))))
    b.close()
ok = res[0] == res[1]
print("asynchronous == synchronous over the last 64 frames, tracks and outputs:", ok)
sys.exit(0 if ok else 1)
