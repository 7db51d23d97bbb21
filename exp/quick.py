"""Low-noise throughput of the headline leg under the current environment: median of N timed asynchronous legs (+ kernel table on request).
usage: quick.py [tag] [--kernels] [--workload W] [--steps S] [--reps N]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("PIN_NODE"):   # experiment: run (and allocate page-locked memory) on the cores of one NUMA node
    def _cpus(txt):
        out = []
        for part in txt.strip().split(","):
            a, _, b = part.partition("-")
            out += list(range(int(a), int(b or a) + 1))
        return out
    os.sched_setaffinity(0, _cpus(open("/sys/devices/system/node/node%s/cpulist" % os.environ["PIN_NODE"]).read()))
import numpy as np
import bench
from dynamicslamtool_amd import engine, kitti_params, synth, shard
if not os.environ.get("PIN_NODE"):
    engine.bind_thread_to_device_node(0)   # the enqueueing thread next to its GPU, as bench.py does
a = sys.argv[1:]
tag = a[0] if a and not a[0].startswith("--") else "run"
def opt(name, default):
    return type(default)(a[a.index(name) + 1]) if name in a else default
wl, steps, reps = opt("--workload", "hdl64_b64"), opt("--steps", 150), opt("--reps", 7)
_, _, _, mo, go = bench.WORKLOADS[wl]
p = kitti_params(mo or 1)
p.ground_method = go if go is not None else 0
leg = bench.Leg(engine, synth, shard, p, wl, 0, 0, 24 if wl.startswith("hdl64") else 8)
for _ in range(5):
    leg.step()
vals = []
trace = (lambda *a: print("TRACE", *a, file=sys.stderr, flush=True)) if os.environ.get("QUICK_TRACE") else (lambda *a: None)
for r in range(reps):
    dt = leg.timed_async(steps)
    vals.append(leg.B * steps / dt)
    trace("leg", r, "done at step", leg.step_no)
if os.environ.get("QUICK_DBGREC"):   # lab libraries that leave a record of an anomaly in a scratch array (round 6's hunt for the rare memory fault)
    leg.batch.synchronize()
    for s_ in range(leg.B):
        rec = leg.batch.debug_read("dbgrec", s_, np.int32, 12)
        if rec[0] == 0x600DBAD:
            print("DBGREC stream", s_, "frame", rec[1], "row", rec[2], "value", hex(int(rec[3]) & 0xffffffff), "nocc", rec[4], "M", rec[5], "tier", rec[6], "hint", rec[7], "H", rec[8], "gh_run calls", rec[9], "bad rows", rec[10], file=sys.stderr, flush=True)
logs = leg.logs(0, leg.step_no)
trace("logs read; synchronous replay of", leg.step_no, "steps")
ref = leg.replay_sync(leg.step_no)
trace("replay done")
fr, bad = bench.compare_logs(logs, ref)
out = {"tag": tag, "median": round(float(np.median(vals)), 0), "min": round(min(vals), 0), "max": round(max(vals), 0), "period_us": round(1e6 * leg.B / float(np.median(vals)), 1), "sane": not bad and bool(fr)}
if "--kernels" in a:
    k = leg.kernel_leg(40, sync=False)
    al = leg.kernel_leg(4, sync=True)
    out["sum_pipelined"] = round(sum(v["ms_total"] for v in k.values()) * 1e3 / 40, 1)
    out["sum_alone"] = round(sum(v["avg_us"] * v["launches"] / 4 for v in al.values()), 1)
    out["kernels"] = {n: [v["avg_us"], al.get(n, {}).get("avg_us")] for n, v in sorted(k.items(), key=lambda kv: -kv[1]["ms_total"])}
print(json.dumps(out))
leg.close()
