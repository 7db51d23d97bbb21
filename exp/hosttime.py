"""Host time per step of the asynchronous pipeline: how long the enqueueing thread needs for push + filter of one batch (13 launches, events), against the period."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicslamtool_amd import engine, kitti_params, synth
B, npts, nf = 64, 120000, 8
engine.bind_thread_to_device_node(0)
p = kitti_params(1)
seeds = [2000 + s for s in range(B)]
buf = engine.DeviceBuffer(nf * B * npts * 16); poses = []
for f in range(nf):
    xs, ps = synth.batch(seeds, [f] * B); buf.upload(xs, f * B * npts * 16); poses.append(ps)
b = engine.MorBatch(p, B, npts)
views = [b.make_views([(buf.ptr + (f * B + s) * npts * 16, npts) for s in range(B)]) for f in range(nf)]
b.set_async(True)
for i in range(16): b.push_views(views[i % nf], poses[i % nf]); b.filter_async()
b.wait()
N = 200
t0 = time.perf_counter(); tp = tf = 0.0
for i in range(N):
    a = time.perf_counter(); b.push_views(views[i % nf], poses[i % nf]); c = time.perf_counter(); b.filter_async(); e = time.perf_counter()
    tp += c - a; tf += e - c
t1 = time.perf_counter(); b.wait(); t2 = time.perf_counter()
print("per step: push call %.1f us, filter call %.1f us (host thread busy %.1f us); loop %.1f us per step, after the last enqueue the GPU needed %.1f us more; period %.1f us" % (1e6 * tp / N, 1e6 * tf / N, 1e6 * (tp + tf) / N, 1e6 * (t1 - t0) / N, 1e6 * (t2 - t1), 1e6 * (t2 - t0) / N))
# the same calls on clouds of 2 000 points (the GPU is done long before the host): the host's own time per step
npts2 = 2000
b2 = engine.MorBatch(p, B, npts2)
xs2 = xs[:, :npts2].copy(); buf2 = engine.DeviceBuffer(B * npts2 * 16); buf2.upload(xs2, 0)
v2 = b2.make_views([(buf2.ptr + s * npts2 * 16, npts2) for s in range(B)])
b2.set_async(True)
for i in range(16): b2.push_views(v2, poses[i % nf]); b2.filter_async()
b2.wait()
t0 = time.perf_counter()
for i in range(N): b2.push_views(v2, poses[i % nf]); b2.filter_async()
t1 = time.perf_counter(); b2.wait(); t2 = time.perf_counter()
print("2 000-point clouds: loop %.1f us per step, GPU behind by %.1f us at the end" % (1e6 * (t1 - t0) / N, 1e6 * (t2 - t1)))
