"""Is the ±3 % spread between processes a property of the process or of the batch?  Several batches, one after the other, in ONE
process: median throughput of each (the engine reads its knobs from the environment at every batch creation).
usage: bimodal.py N [ENV=VAL …]   — N batches per setting, settings interleaved"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from dynamicslamtool_amd import engine, kitti_params, synth, shard
p = kitti_params(1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
settings = sys.argv[2:] or ["-"]
out = {c: [] for c in settings}
for rep in range(n):
    for c in settings:
        kv = [] if c == "-" else [x.split("=", 1) for x in c.split(",")]
        for k, v in kv:
            os.environ[k] = v
        leg = bench.Leg(engine, synth, shard, p, "hdl64_b64", 0, 0, 24)
        for _ in range(5):
            leg.step()
        vals = []
        for _ in range(4):
            dt = leg.timed_async(100)
            vals.append(leg.B * 100 / dt)
        out[c].append(int(np.median(vals)))
        leg.close()
        for k, v in kv:
            del os.environ[k]
for c in settings:
    print(c, sorted(out[c]), "median", int(np.median(out[c])))
