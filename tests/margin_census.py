"""Margin census (tests/golden/margin_census.json): how much of the oracle's output could a real PCL / FLANN binary change?

The reference's arithmetic lives in libraries this image cannot build, so the restatement is pinned by definition-level second implementations only.  What two
CORRECT evaluations of the same definitions can disagree on is bounded by the decisions that sit within a few ulp of their thresholds, and by what the kd-tree owes to the slack
of its pruning test (FLANN prunes without one).  `census_of` runs the oracle over a sample of one bench workload — every stream twice: with the slack and with FLANN's literal
pruning test — and returns the margin counters (oracle.CENSUS_KEYS) plus whether the two modes gave identical records (fullbatch.record_*: counts and CRC-32 of labels,
correspondences + scores, detection flags, the filtered cloud's bytes, tracks).  TEST INFRASTRUCTURE."""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, "golden", "margin_census.json")
N_FRAMES = 3
# workload of bench.py → (sensor, BASELINE config index of the seeds, method, ground method, streams sampled)
SAMPLES = {
    "hdl64_b64": ("hdl64", 2, 1, 0, 16),
    "os128_b64": ("os128", 3, 1, 0, 8),
    "agg10_b32": ("agg10", 5, 1, 0, 2),
    "hdl64_urban_b64": ("hdl64_urban", 6, 1, 0, 8),
    "hdl64_b64_method2": ("hdl64", 2, 2, 0, 16),
    "hdl64_b64_voxel_ground": ("hdl64", 2, 1, 1, 8),
}


def _records(sensor, seed, method, ground, literal):
    from dynamicslamtool_amd import kitti_params, synth
    from oracle import oracle as O
    from fullbatch import record_after_filter, record_before_filter
    p = kitti_params(method)
    p.ground_method = ground
    O.set_literal_pruning(literal)
    try:
        o, recs = O.Oracle(p, 4, 3), []
        for f in range(N_FRAMES):
            x, ps = synth.frame(seed, sensor, f)
            o.push(x, ps)
            r = record_before_filter(o)
            recs.append(r + record_after_filter(o, o.filter()))
        o.close()
    finally:
        O.set_literal_pruning(False)
    return recs


def census_of(job):
    """(workload, number of streams or None) → {"streams", "frames", "census": {...}, "literal_pruning_identical": bool, "points": total input points}"""
    name, n = job
    from dynamicslamtool_amd import synth
    from oracle import oracle as O
    sensor, cfg, method, ground, n_def = SAMPLES[name]
    n = n or n_def
    O.census_reset()
    same = True
    for s in range(n):
        seed = 1000 * cfg + s
        a = _records(sensor, seed, method, ground, False)
        cen = O.census_read()          # (the literal run below repeats the same decisions: read the counters of the slack run only)
        b = _records(sensor, seed, method, ground, True)
        O.census_reset()
        same = same and a == b
        tot = cen if s == 0 else {k: tot[k] + cen[k] for k in cen}
    return {"streams": n, "frames": N_FRAMES, "points": n * N_FRAMES * synth.n_points(sensor), "census": tot, "literal_pruning_identical": same}


def load():
    return json.load(open(PATH))
