"""Is the ±3 % spread between processes a property of the process or of the batch (its allocations)?  Several batches, one after the
other, in ONE process: median throughput of each."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from dynamicslamtool_amd import engine, kitti_params, synth, shard
p = kitti_params(1)
out = []
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    leg = bench.Leg(engine, synth, shard, p, "hdl64_b64", 0, 0, 24)
    for _ in range(5):
        leg.step()
    vals = []
    for _ in range(5):
        dt = leg.timed_async(100)
        vals.append(leg.B * 100 / dt)
    out.append(int(np.median(vals)))
    leg.close()
print(json.dumps(out))
