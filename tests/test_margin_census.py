"""CPU: the margin census of the oracle (tests/golden/margin_census.json, generator tests/golden/make_margin_census.py) is reproducible, and FLANN's literal pruning
test (no slack) gives the oracle's results on every stream the fixtures are made of.  What this pins: NOT the reference binary — nothing here can — but the size of the
exposure: only decisions within a few ulp of a threshold, or results the kd-tree owes to its pruning slack, can differ between two correct evaluations of the definitions."""
import numpy as np
import pytest

import margin_census as mc
from oracle import oracle as O
from scenes import scene_params, small_stream


@pytest.mark.parametrize("name", ["hdl64_b64", "hdl64_b64_method2"])
def test_census_of_a_workload_sample_is_reproducible(name):
    fix = mc.load()["workloads"][name]
    got = mc.census_of((name, fix["streams"]))
    assert got == fix


def test_no_result_of_any_workload_sample_is_owed_to_the_pruning_slack():
    """The committed census: in every workload sample FLANN's literal pruning gave identical records, no neighbour and no nearest neighbour was found only thanks to the slack,
    and the decisions at a threshold are a vanishing share of the decisions taken."""
    w = mc.load()["workloads"]
    assert set(w) == set(mc.SAMPLES)
    for name, r in w.items():
        c = r["census"]
        assert r["literal_pruning_identical"], name
        assert c["radius_neighbours_owed_to_slack"] == 0 and c["nn_results_owed_to_slack"] == 0, (name, c)
        assert c["radius_queries"] > 100000 and c["c1_pairs_within_4ulp_of_r2"] < 1e-2 * c["radius_queries"], (name, c)
        assert c["nn_exact_ties"] == 0 and c["volume_gates_within_1e-6"] == 0 and c["method1_distances_within_4ulp_of_a_bound"] == 0, (name, c)


@pytest.mark.parametrize("method", [1, 2])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_literal_pruning_reproduces_the_small_golden_streams(seed, method):
    """The streams tests/golden/small_*.npz are made of (and their siblings), frame by frame, every result array: identical with and without the slack."""
    p = scene_params(method_choice=method)
    frames = small_stream(seed, n_frames=9)
    res = []
    for literal in (False, True):
        O.set_literal_pruning(literal)
        try:
            o, out = O.Oracle(p, 4, 3), []
            for x, ps in frames:
                o.push(x, ps)
                q, m, d, sc = o.correspondences()
                out.append((o.labels().copy(), np.asarray(q).copy(), np.asarray(m).copy(), np.asarray(sc).copy(), o.detection().copy(), o.centroids().copy(), o.filter().copy()))
            o.close()
        finally:
            O.set_literal_pruning(False)
        res.append(out)
    for f, (a, b) in enumerate(zip(*res)):
        for u, v in zip(a, b):
            assert np.array_equal(u, v), (seed, method, f)
