"""Full-batch parity for the BASELINE 1-GPU configurations beyond configs[1] (which tests/test_gpu_parity.py holds to 64 live oracles): every stream of
os128 B = 64 (config 3), agg10 B = 32 (config 5), the voxel-covariance ground variant and method 2 on the 64 hdl64 streams — counts and CRC-32 of labels,
correspondences + scores, detection flags and the filtered cloud's bytes for the first two frame pairs, against tests/golden/fullbatch_digests.json
(records of the CPU oracle, generator tests/golden/make_golden_fullbatch.py)."""
import pytest

import fullbatch as fb


@pytest.mark.parametrize("name,seed", [("os128_b64", 3017), ("hdl64_b64_voxel_ground", 2040), ("hdl64_b64_method2", 2063)])
def test_oracle_reproduces_fullbatch_records(name, seed):
    """The fixture is what the oracle in the tree produces (a sample of the streams: the whole fixture is minutes of CPU)."""
    fx = fb.load()
    assert fx["fields"] == list(fb.FIELDS) and set(fx["digests"]) == set(fb.CASES)
    assert fb.oracle_stream((name, seed)) == fx["digests"][name][str(seed)]


def test_fixture_covers_every_stream_and_is_not_vacuous():
    fx = fb.load()["digests"]
    for name, (sensor, cfg, B, method, gm) in fb.CASES.items():
        assert sorted(int(k) for k in fx[name]) == [1000 * cfg + s for s in range(B)], name
        recs = list(fx[name].values())
        assert all(len(r) == fb.N_FRAMES and all(len(x) == len(fb.FIELDS) for x in r) for r in recs)
        last = [r[fb.N_FRAMES - 1] for r in recs]
        assert sum(x[fb.FIELDS.index("pairs")] for x in last) > 5 * B, name          # correspondences everywhere
        assert sum(x[fb.FIELDS.index("K")] for x in last) > 10 * B, name
        assert all(x[fb.FIELDS.index("n_out")] > 0 for x in last), name


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(fb.CASES))
def test_every_stream_of_the_batch_against_the_oracle_records(name):
    from dynamicslamtool_amd import synth
    from dynamicslamtool_amd.engine import MorBatch
    fx = fb.load()["digests"][name]
    sensor, _, B = fb.CASES[name][:3]
    seeds = fb.seeds_of(name)
    b = MorBatch(fb.params_of(name), B, synth.n_points(sensor), 4, 3)
    for f in range(fb.N_FRAMES):
        xs, ps = synth.batch(seeds, [f] * B, sensor)
        b.push(list(xs), ps)
        got = [fb.record_before_filter(b, s) for s in range(B)]
        outs = b.filter()
        for s, seed in enumerate(seeds):
            rec = got[s] + fb.record_after_filter(b, outs[s], s)
            assert rec == fx[str(seed)][f], (name, seed, f, dict(zip(fb.FIELDS, rec)), dict(zip(fb.FIELDS, fx[str(seed)][f])))
    b.close()
