"""Generator of tests/golden/fullbatch_digests.json (see tests/fullbatch.py).  Run from the repo root:
    python tests/golden/make_golden_fullbatch.py [case ...]
Every stream of every case goes through oracle/mor_oracle.c in worker processes (minutes on 8 cores: the 1 M-point clouds dominate)."""
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fullbatch as fb  # noqa: E402


def main():
    os.environ["OMP_NUM_THREADS"] = "1"
    names = sys.argv[1:] or list(fb.CASES)
    res = fb.load()["digests"] if os.path.exists(fb.PATH) else {}
    with mp.get_context("spawn").Pool(max(1, (os.cpu_count() or 2) - 1)) as pool:
        for name in names:
            t = time.time()
            seeds = fb.seeds_of(name)
            recs = pool.map(fb.oracle_stream, [(name, s) for s in seeds], chunksize=1)
            res[name] = {str(seed): r for seed, r in zip(seeds, recs)}
            print(name, len(seeds), "streams", round(time.time() - t, 1), "s", file=sys.stderr)
    prov = ("records of oracle/mor_oracle.c (tests/fullbatch.py: oracle_stream) for every stream of the case, frames 0..%d; fields %s; the oracle itself is held to the independent "
            "full-size implementation on streams of the same generators (tests/test_oracle_independent_fullsize.py); no output of the real reference (PCL/ROS, unbuildable here) backs them"
            % (fb.N_FRAMES - 1, ",".join(fb.FIELDS)))
    json.dump({"cases": {k: list(v) for k, v in fb.CASES.items()}, "fields": list(fb.FIELDS), "digests": res, "provenance": prov}, open(fb.PATH, "w"), separators=(",", ":"), sort_keys=True)
    print("wrote", fb.PATH, file=sys.stderr)


if __name__ == "__main__":
    main()
