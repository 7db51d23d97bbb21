"""Generates tests/golden/margin_census.json (minutes of CPU: the oracle over a sample of every bench workload, twice): python tests/golden/make_margin_census.py"""
import json
import multiprocessing as mp
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import margin_census as mc  # noqa: E402

if __name__ == "__main__":
    from oracle import oracle as O
    O.build()
    with mp.Pool(min(6, os.cpu_count() or 1)) as pool:
        res = pool.map(mc.census_of, [(n, None) for n in mc.SAMPLES])
    out = {"_what": "margin census of the CPU oracle over a sample of every bench workload (tests/margin_census.py): decisions within a few ulp of their threshold, and what the "
                    "kd-tree owes to the slack of its pruning test (FLANN prunes without one); literal_pruning_identical: the same streams with FLANN's literal test gave identical records",
           "keys": list(O.CENSUS_KEYS), "workloads": dict(zip(mc.SAMPLES, res))}
    json.dump(out, open(mc.PATH, "w"), indent=1)
    for k, v in out["workloads"].items():
        print(k, v)
