"""Per workload: occupied cells, non-ground points, clustered points and the largest slab of the cell graph per stream (min / median / max) after three steps; slabs per stream."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from dynamicslamtool_amd import engine, kitti_params, synth, shard
for wl in sys.argv[1:] or ["hdl64_b64", "os128_b64", "hdl64_urban_b64", "agg10_b32"]:
    _, _, _, mo, go = bench.WORKLOADS[wl]
    p = kitti_params(mo or 1); p.ground_method = go if go is not None else 0
    leg = bench.Leg(engine, synth, shard, p, wl, 0, 0, 6)
    for _ in range(4):
        leg.step()
    leg.batch.synchronize()
    sc = [leg.batch.stage_counts(s) for s in range(leg.B)]
    cn = [leg.batch.counts(s) for s in range(leg.B)]
    f = lambda v: "%d / %d / %d" % (min(v), int(np.median(v)), max(v))
    print(wl, "P", leg.batch.debug_config()["P"], "| cells", f([x["n_occ"] for x in sc]), "| largest slab", f([x["max_loc"] for x in sc]), "| M", f([int(c.n_cloud) for c in cn]),
          "| C", f([int(c.n_clustered) for c in cn]), "| K", f([int(c.n_clusters) for c in cn]), "| deferred", f([x["n_defer"] for x in sc]), "| tier1b", f([x["n_tier1b"] for x in sc]))
    leg.close()
