"""A secondary leg of bench.py as it runs there (6 frames, ping-pong, 3 warm-up steps), timed N times; preceded by the legs the bench runs before it when --prelude.
usage: leg_repeat.py [--workload W] [--steps S] [--reps N] [--prelude]     (looks for a bimodal leg: round 5 saw the street scene's leg at half its rate in two fresh-box runs of nine)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from dynamicslamtool_amd import engine, kitti_params, synth, shard
engine.bind_thread_to_device_node(0)
a = sys.argv[1:]
def opt(name, default):
    return type(default)(a[a.index(name) + 1]) if name in a else default
wl, steps, reps = opt("--workload", "hdl64_urban_b64"), opt("--steps", 40), opt("--reps", 12)
def leg_of(name, n_frames):
    _, _, _, mo, go = bench.WORKLOADS[name]
    p = kitti_params(mo or 1); p.ground_method = go if go is not None else 0
    return bench.Leg(engine, synth, shard, p, name, 0, 0, n_frames)
if "--prelude" in a:
    for name, st in (("hdl64_b64", 125), ("os128_b64", 33), ("agg10_b32", 33)):
        lg = leg_of(name, 6)
        for _ in range(3): lg.step()
        lg.batch.synchronize()
        print(name, round(lg.B * st / lg.timed_async(st)), flush=True)
        lg.close()
lg = leg_of(wl, 6)
for _ in range(3): lg.step()
lg.batch.synchronize()
vals = [round(lg.B * steps / lg.timed_async(steps)) for _ in range(reps)]
print(json.dumps({"workload": wl, "steps": steps, "runs": vals}))
lg.close()
