"""Experiment: E engines of 64/E streams each on one GPU, each driven by its own host thread."""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicslamtool_amd import engine, kitti_params, synth
E = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B, npts, steps, warm, nf = 64 // E, 120000, 30, 3, 12
p = kitti_params(1)
engs = []
for e in range(E):
    seeds = [2000 + e * B + s for s in range(B)]
    buf = engine.DeviceBuffer(nf * B * npts * 16)
    poses = np.empty((nf, B, 7))
    for f in range(nf):
        xs, ps = synth.batch(seeds, [f] * B)
        buf.upload(xs, f * B * npts * 16); poses[f] = ps
    b = engine.MorBatch(p, B, npts)
    views = [b.make_views([(buf.ptr + (f * B + s) * npts * 16, npts) for s in range(B)]) for f in range(nf)]
    engs.append((b, views, poses, buf))
def frame_of(step):
    period = 2 * (nf - 1); k = step % period
    return k if k < nf else period - k
def run(e, n0, n):
    b, views, poses, _ = engs[e]
    for i in range(n0, n0 + n):
        f = frame_of(i); b.push_views(views[f], poses[f]); b.filter_async()
    b.wait()
for e in range(E): run(e, 0, warm)
for e in range(E): engs[e][0].set_async(True)
bar = threading.Barrier(E + 1)
def worker(e):
    bar.wait(); run(e, warm, steps); bar.wait()
ths = [threading.Thread(target=worker, args=(e,)) for e in range(E)]
for t in ths: t.start()
bar.wait(); t0 = time.perf_counter(); bar.wait(); dt = time.perf_counter() - t0
print("engines", E, "streams each", B, "frame-pairs/s", 64 * steps / dt, "ms per 64-stream step", 1e3 * dt / steps)
