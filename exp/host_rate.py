"""Is the asynchronous leg bound by the host?  (1) the headline leg with the host's time inside push / filter per step; (2) the same calls on clouds cut to 1 000 points — the device
has next to nothing to do, so the step rate is the rate at which the host can enqueue frames.
usage: host_rate.py [workload] [steps]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from dynamicslamtool_amd import engine, kitti_params, synth, shard
engine.bind_thread_to_device_node(0)
wl = sys.argv[1] if len(sys.argv) > 1 else "hdl64_b64"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
_, _, _, mo, go = bench.WORKLOADS[wl]
p = kitti_params(mo or 1); p.ground_method = go if go is not None else 0
leg = bench.Leg(engine, synth, shard, p, wl, 0, 0, 12)
def run(views, n):
    b = leg.batch
    for i in range(8): b.push_views(views[leg.frame_of(i)], leg.poses[leg.frame_of(i)]); b.filter_device()
    b.set_async(True); b.synchronize()
    tp = tf = 0.0; t0 = time.perf_counter()
    for i in range(n):
        f = leg.frame_of(i)
        a = time.perf_counter(); b.push_views(views[f], leg.poses[f]); c = time.perf_counter(); b.filter_async(); e = time.perf_counter()
        tp += c - a; tf += e - c
    t1 = time.perf_counter(); b.wait(); b.synchronize(); t2 = time.perf_counter(); b.set_async(False)
    return {"period_us": round(1e6 * (t2 - t0) / n, 1), "enqueue_loop_us_per_step": round(1e6 * (t1 - t0) / n, 1), "host_in_push_us": round(1e6 * tp / n, 1), "host_in_filter_us": round(1e6 * tf / n, 1), "frame_pairs_per_s": round(leg.B * n / (t2 - t0))}
cb = leg.npts * 16
tiny = [leg.batch.make_views([(leg.buf.ptr + (f * leg.B + s) * cb, 1000) for s in range(leg.B)]) for f in range(leg.n_frames)]
out = {"workload": wl, "full": run(leg.views, steps), "tiny_clouds": run(tiny, steps), "full_again": run(leg.views, steps)}
print(json.dumps(out))
