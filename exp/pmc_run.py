"""A handful of synchronous steps of one workload and nothing else — the program the PMC passes of profiles/collect.sh profile (rocprofv3's counter
collection costs about a second per dispatch on this stack: bench.py's legs and checks would make one pass take minutes).
usage: pmc_run.py <workload> [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from dynamicslamtool_amd import engine, kitti_params, synth, shard
wl = sys.argv[1] if len(sys.argv) > 1 else "hdl64_b64"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
_, _, _, mo, go = bench.WORKLOADS[wl]
p = kitti_params(mo or 1)
p.ground_method = go if go is not None else 0
leg = bench.Leg(engine, synth, shard, p, wl, 0, 0, steps + 1)
for _ in range(steps):
    leg.step()
leg.batch.synchronize()
print("pmc_run", wl, steps, "steps done")
leg.close()
