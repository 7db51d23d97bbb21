import sys, os
sys.path.insert(0, "/root/repo")
import bench, numpy as np
from dynamicslamtool_amd import engine, kitti_params, synth, shard
for wl in ("hdl64_b64", "os128_b64", "hdl64_urban_b64", "agg10_b32"):
    p = kitti_params(1)
    leg = bench.Leg(engine, synth, shard, p, wl, 0, 0, 4)
    for _ in range(3): leg.step()
    occ = sorted(leg.batch.stage_counts(s)["n_occ"] for s in range(leg.B))
    M = sorted(int(leg.batch.counts(s).n_cloud) for s in range(leg.B))
    print(wl, "n_occ min/median/max", occ[0], occ[len(occ)//2], occ[-1], " M", M[0], M[len(M)//2], M[-1])
    leg.close()
