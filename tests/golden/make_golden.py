"""Generates the committed golden fixtures (inputs + expected outputs) from the CPU oracle.

Run from the repo root:  python tests/golden/make_golden.py
The oracle is used only AFTER it agrees with the definition-level brute force
(tests/test_oracle_bruteforce.py); this script re-checks that agreement for every fixture frame
before writing.  Fixtures are data only: seeded synthetic inputs and the oracle's outputs.
The reference itself ships no golden vectors and cannot run here.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from bruteforce import BruteMOR  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402
from scenes import scene_params, small_stream  # noqa: E402

CASES = [dict(name="small_m2_seed1", seed=1, method=2, frames=8), dict(name="small_m1_seed3", seed=3, method=1, frames=7)]


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    for case in CASES:
        p = scene_params(method_choice=case["method"])
        o, b = Oracle(p, 4, 3), BruteMOR(p, 4, 3)
        rec = {"method": np.int32(case["method"]), "seed": np.int32(case["seed"]), "n_frames": np.int32(case["frames"]),
               "n_bad": np.int32(4), "n_good": np.int32(3), "min_cluster_size": np.int32(p.min_cluster_size)}
        for f, (pts, pose) in enumerate(small_stream(case["seed"], n_frames=case["frames"])):
            o.push(pts, pose)
            b.push(pts, pose)
            assert np.array_equal(o.labels(), b.labels())
            q, m, d, s = o.correspondences()
            off, idx = o.clusters()
            xyz, conf, mx = o.tracks()
            out = o.filter()
            assert np.array_equal(out.view(np.uint32), b.filter().view(np.uint32))
            xyz2, conf2, mx2 = o.tracks()
            pre = "f%d_" % f
            rec.update({pre + "pts": pts, pre + "pose": pose, pre + "labels": o.labels(), pre + "ground": o.ground_indices(),
                        pre + "cl_off": off, pre + "cl_idx": idx, pre + "centroids": o.centroids(), pre + "detection": o.detection(),
                        pre + "corr_q": q, pre + "corr_m": m, pre + "corr_d": d, pre + "score": s,
                        pre + "tracks_push": xyz, pre + "conf_push": conf, pre + "tracks_filter": xyz2, pre + "conf_filter": conf2, pre + "out": out})
        np.savez_compressed(os.path.join(here, case["name"] + ".npz"), **rec)
        print("wrote", case["name"], "tracks", len(conf2))


if __name__ == "__main__":
    main()
