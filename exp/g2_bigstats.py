"""Voxel ground variant: what k_g2_cov_big finds in its queue — per stream the queue length and the entries k_g2_cov_mid left open (min / median / max over the batch, the fullest streams)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from dynamicslamtool_amd import engine, kitti_params, synth, shard
p = kitti_params(1); p.ground_method = 1
leg = bench.Leg(engine, synth, shard, p, "hdl64_b64_voxel_ground", 0, 0, 4)
for _ in range(3):
    leg.step()
nb, op = [], []
for s in range(leg.B):
    n = int(leg.batch.debug_read("g2_nbig", s, np.int32, 1)[0])
    q = leg.batch.debug_read("g2_big", s, np.int32, max(n, 1))[:n]
    nb.append(n); op.append(int((q >= 0).sum()))
print(json.dumps({"queue": [min(nb), int(np.median(nb)), max(nb)], "open_for_big": [min(op), int(np.median(op)), max(op)], "sum_open": int(sum(op)), "top": sorted(op)[-8:]}))
leg.close()
