import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dynamicslamtool_amd import engine, kitti_params, synth
B, npts = 64, 120000
p = kitti_params(1)
seeds = [2000 + s for s in range(B)]
hin = [engine.HostBuffer((B, npts, 4)) for _ in range(2)]; hout = [engine.HostBuffer((B, npts, 4)) for _ in range(2)]; pp = []
for f in range(2):
    xs, ps = synth.batch(seeds, [f] * B); hin[f].array[...] = np.asarray(xs).reshape(B, npts, 4); pp.append(np.ascontiguousarray(ps))
if "--prelude" in sys.argv:   # what bench.py does before its end-to-end legs: a device-resident leg on another batch, closed again
    import bench
    from dynamicslamtool_amd import shard
    leg = bench.Leg(engine, synth, shard, p, "hdl64_b64", 0, 0, 24)
    leg.timed_async(25)
    if "--timing" in sys.argv: leg.kernel_leg(8, sync=False)
    leg.close()
b = engine.MorBatch(p, B, npts)
views = [b.make_host_views([hin[f].array[s] for s in range(B)]) for f in range(2)]
optr = [b.make_out_pointers([hout[f].array[s] for s in range(B)]) for f in range(2)]
b.push_views(views[0], pp[0]); b.filter_into([hout[0].array[s] for s in range(B)])
tp = tf = 0.0
for r in range(6):
    t = time.perf_counter(); b.push_views(views[(r + 1) % 2], pp[(r + 1) % 2]); tp += time.perf_counter() - t
    t = time.perf_counter(); b.filter_into([hout[0].array[s] for s in range(B)]); tf += time.perf_counter() - t
print("sync: push %.2f ms, filter_into %.2f ms per step -> %.0f frame-pairs/s" % (tp / 6 * 1e3, tf / 6 * 1e3, B * 6 / (tp + tf)))
b.set_async(True)
for reps in (8, 24):
    t = time.perf_counter()
    for r in range(reps):
        b.push_views(views[(r + 1) % 2], pp[(r + 1) % 2]); b.filter_async_to(optr[r % 2], on_device=False)
    te = time.perf_counter() - t
    b.wait(); tt = time.perf_counter() - t
    print("async host in/out: %d steps, enqueue %.2f ms/step, total %.2f ms/step -> %.0f frame-pairs/s" % (reps, te / reps * 1e3, tt / reps * 1e3, B * reps / tt))
t = time.perf_counter()
for r in range(16):
    b.push_views(views[(r + 1) % 2], pp[(r + 1) % 2]); b.filter_async()
b.wait(); tt = time.perf_counter() - t
print("async host in, device out: %.2f ms/step -> %.0f" % (tt / 16 * 1e3, B * 16 / tt))
b.close()
