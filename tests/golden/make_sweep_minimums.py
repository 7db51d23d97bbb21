"""Oracle-only counts of tests/test_gpu_parity.py::test_parameter_sweep_on_small_streams (clusters and correspondences summed over the frames
and both streams of each case), committed as tests/golden/sweep_minimums.json so that a case which produces nothing cannot pass vacuously.
Run from the repo root: python tests/golden/make_sweep_minimums.py"""
import json
import os
import sys


ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.oracle import Oracle  # noqa: E402
from scenes import sweep_case  # noqa: E402

cases = []
for case in range(14):
    p, n_bad, n_good, streams = sweep_case(case)
    clusters = corr = 0
    for st in streams:
        o = Oracle(p, n_bad, n_good)
        for pts, pose in st:
            o.push(pts, pose)
            c = o.counts()
            clusters += int(c.n_clusters)
            corr += int(c.n_corr)
            o.filter()
        o.close()
    cases.append([clusters, corr])
    print(case, clusters, corr)
json.dump({"note": "oracle-only sums over 8 frames x 2 streams per case of test_parameter_sweep_on_small_streams: [clusters, correspondences]", "cases": cases},
          open(os.path.join(ROOT, "tests", "golden", "sweep_minimums.json"), "w"), indent=1)
