"""VERDICT round 5, item 5 — count first: what share of the method-1 queries (the clustered points of ca, transformed into cb's frame; /root/reference/src/MovingObjectRemoval.cpp:336-366)
would a ONE-LINE occupancy test resolve?  Per occupied cell of cb's clustering grid (edge 0.57·r) a bit per sub-voxel of edge cell/n ≤ √(lb/3): a query whose own sub-voxel holds a point of
the matched cluster has that point within √lb ⇒ its nearest-neighbour distance is ≤ lb ⇒ it is not counted (:356) — decided by one 64-byte line, no point loads.  Counted here from the
oracle's streams (test infrastructure: this script imports the oracle), for n = 7 and 8, next to the share of queries whose true nearest matched point lies within √lb at all.
usage: python tests/golden/make_subvoxel_census.py > tests/golden/subvoxel_census.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from scipy.spatial import cKDTree
from dynamicslamtool_amd import kitti_params, synth
from oracle.oracle import Oracle

def cells(p, pts, n):
    cs = np.float32(np.float32(p.ec_distance_threshold) * np.float32(0.57)); inv = np.float32(1.0) / cs
    o = np.array([-p.trim_x, -p.trim_y, p.gp_limit], np.float32)
    u = (pts[:, :3].astype(np.float32) - o) * inv            # fp32, as grid_cell does
    c = np.floor(u).astype(np.int64)
    sub = np.minimum(np.floor((u - np.floor(u)) * np.float32(n)).astype(np.int64), n - 1)
    key = (c[:, 0] * 4096 + c[:, 1]) * 4096 + c[:, 2]
    return key * 512 + (sub[:, 2] * n + sub[:, 1]) * n + sub[:, 0]

def run(sensor, cfg, streams, frames=3):
    p = kitti_params(1)
    tot = {"queries": 0, "nn_within_lb": 0, "bit_n7": 0, "bit_n8": 0}
    for s in range(streams):
        o = Oracle(p, 4, 3)
        prev_off = None
        for f in range(frames):
            x, pose = synth.frame(1000 * cfg + s, sensor, f)
            o.push(x, pose)
            raw = x[np.isfinite(x[:, :3]).all(1) & (np.abs(x[:, 0]) <= p.trim_x) & (np.abs(x[:, 1]) <= p.trim_y)]
            cloud = raw[(raw[:, 2] >= p.gp_limit) & (raw[:, 2] <= p.trim_z)]
            off, idx = o.clusters()
            if prev_off is not None:
                q_, m_, _, _ = o.correspondences()
                _, qpts = o.prev_transformed()
                for j, m in zip(q_, m_):
                    qs = qpts[prev_off[j]:prev_off[j + 1]]
                    mp = cloud[idx[off[m]:off[m + 1]]]
                    tot["queries"] += len(qs)
                    d, _ = cKDTree(mp[:, :3].astype(np.float64)).query(qs[:, :3].astype(np.float64))
                    tot["nn_within_lb"] += int((d * d <= p.pde_lb).sum())
                    for n in (7, 8):
                        occ = np.unique(cells(p, mp, n))
                        tot["bit_n%d" % n] += int(np.isin(cells(p, qs, n), occ).sum())
            prev_off = off.copy()
            o.filter()
        o.close()
    tot["share_nn_within_lb"] = round(tot["nn_within_lb"] / max(tot["queries"], 1), 4)
    for n in (7, 8):
        tot["share_bit_n%d" % n] = round(tot["bit_n%d" % n] / max(tot["queries"], 1), 4)
    return tot

if __name__ == "__main__":
    p = kitti_params(1)
    cs = p.ec_distance_threshold * 0.57
    out = {"profile": "kitti", "cell_edge": cs, "sqrt_lb_over_3": float(np.sqrt(p.pde_lb / 3)), "sub_edge_n7": cs / 7, "sub_edge_n8": cs / 8,
           "hdl64 (config 2, streams 0-7, 2 frame-pairs each)": run("hdl64", 2, 8), "hdl64_urban (config 6, streams 0-3, 2 frame-pairs each)": run("hdl64_urban", 6, 4)}
    print(json.dumps(out, indent=1))
