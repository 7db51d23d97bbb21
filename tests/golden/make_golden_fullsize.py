"""Full-size golden digests: the synthetic generator is part of this repo and deterministic, so a full-size fixture
needs no point data — only the expected digests.  For every (sensor, profile, seed, frame) this script runs the CPU
oracle (after tests/test_oracle_bruteforce.py holds) and records the counts and CRC-32 digests of the integer results
and of the filtered cloud's bytes.  Run from the repo root:  python tests/golden/make_golden_fullsize.py"""
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CASES = [   # name, sensor, method, ground_method, seeds, frames
    ("hdl64_m1", "hdl64", 1, 0, [2000, 2001, 2005, 2017], 4),
    ("hdl64_m2", "hdl64", 2, 0, [2003, 2033], 4),
    ("hdl64_m1_voxel_ground", "hdl64", 1, 1, [2002], 3),
    ("os128_m1", "os128", 1, 0, [3001], 3),
    ("os128_m2", "os128", 2, 0, [3002], 3),
    ("hdl64_urban_m1", "hdl64_urban", 1, 0, [6000], 3),
    ("hdl64_urban_m2", "hdl64_urban", 2, 0, [6001], 3),
]


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def digest(eng, out, s=None):
    """eng: Oracle (s None) or MorBatch (s = stream index); out: the filtered cloud of this frame."""
    a = (lambda name: getattr(eng, name)()) if s is None else (lambda name: getattr(eng, name)(s))
    c = a("counts")
    off, idx = a("clusters")
    q, m, dist, score = a("correspondences")
    xyz, conf, mx = a("tracks")
    return {
        "T": int(c.n_trim), "M": int(c.n_cloud), "G": int(c.n_ground), "K": int(c.n_clusters), "C": int(c.n_clustered), "pairs": int(c.n_corr),
        "labels": crc(a("labels")), "ground": crc(a("ground_indices")), "cl_off": crc(off), "cl_idx": crc(idx),
        "corr": crc(np.concatenate([q, m]).astype(np.int32)), "score": crc(np.asarray(score, np.float64)), "detection": crc(a("detection")),
        "tracks": int(len(conf)), "conf": crc(np.asarray(conf, np.int32)), "n_out": int(len(out)), "out": crc(out),
    }


def main():
    from dynamicslamtool_amd import kitti_params, synth
    from oracle.oracle import Oracle
    res = {}
    for name, sensor, method, gm, seeds, nf in CASES:
        p = kitti_params(method)
        p.ground_method = gm
        for seed in seeds:
            o = Oracle(p, 4, 3)
            for f in range(nf):
                x, pose = synth.frame(seed, sensor, f)
                o.push(x, pose)
                out = o.filter()
                # digests are taken after filterCloud (tracks include the filter's confidence updates)
                d = digest(o, out)
                res["%s/%d/%d" % (name, seed, f)] = d
            o.close()
    path = os.path.join(ROOT, "tests", "golden", "fullsize_digests.json")
    provenance = ("digests of oracle/mor_oracle.c outputs (this script); the cases hdl64_m1/2000, hdl64_m1/2005, hdl64_m2/2003, hdl64_m1_voxel_ground/2002, "
                  "os128_m1/3001, os128_m2/3002, hdl64_urban_m1/6000 and hdl64_urban_m2/6001 are reproduced without the oracle by the independent "
                  "full-size implementation tests/independent_fullsize.py (tests/test_oracle_independent_fullsize.py); no output of the real "
                  "reference (PCL/ROS, unbuildable here) backs any of them")
    json.dump({"cases": [list(c[:4]) + [c[4], c[5]] for c in CASES], "digests": res, "provenance": provenance}, open(path, "w"), indent=0, sort_keys=True)
    print("wrote", path, len(res), "frames")


if __name__ == "__main__":
    main()
