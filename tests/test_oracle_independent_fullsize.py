"""Pins oracle/mor_oracle.c at FULL size against tests/independent_fullsize.py (scipy cKDTree candidates re-evaluated
with the exact fp32 predicates, scipy connected components, python tracking) and ties the committed full-size digests
(tests/golden/fullsize_digests.json) to that independent implementation: the digests are not only "the oracle's own
output" — every case below is reproduced without the oracle.  CPU only."""
import json
import os

import numpy as np
import pytest

from dynamicslamtool_amd import kitti_params, synth
from independent_fullsize import IndependentMOR
from oracle.oracle import Oracle
from test_golden import _fullsize

CASES = [("hdl64_m1", 2000, 3), ("hdl64_m1", 2005, 4), ("hdl64_m2", 2003, 4), ("os128_m1", 3001, 3),
         ("hdl64_m1_voxel_ground", 2002, 3), ("os128_m2", 3002, 3), ("hdl64_urban_m1", 6000, 3), ("hdl64_urban_m2", 6001, 3)]


def _compare(o, b, tag):
    co, cb = o.counts(), b.counts()
    for k in ("n_trim", "n_cloud", "n_ground", "n_clusters", "n_clustered", "n_corr", "n_tracks"):
        assert getattr(co, k) == getattr(cb, k), (tag, k, getattr(co, k), getattr(cb, k))
    assert np.array_equal(o.ground_indices(), b.ground_indices()), tag
    assert np.array_equal(o.labels(), b.labels()), tag
    for x, y in zip(o.clusters(), b.clusters()):
        assert np.array_equal(x, y), tag
    assert np.array_equal(o.centroids().view(np.uint32), b.centroids().view(np.uint32)), tag   # same sequential fp64 sums
    qo, mo, do, so = o.correspondences()
    qb, mb, db, sb = b.correspondences()
    assert np.array_equal(qo, qb) and np.array_equal(mo, mb) and np.array_equal(do, db) and np.array_equal(so, sb), tag
    assert np.array_equal(o.detection(), b.detection()), tag
    for x, y in zip(o.tracks(), b.tracks()):
        assert np.array_equal(x, y), tag


@pytest.mark.parametrize("name,seed,frames", CASES)
def test_oracle_matches_independent_fullsize(name, seed, frames):
    mod, fx = _fullsize()
    case = [c for c in fx["cases"] if c[0] == name][0]
    sensor, method, gm = case[1], case[2], case[3]
    p = kitti_params(method)
    p.ground_method = gm
    o, b = Oracle(p, 4, 3), IndependentMOR(p, 4, 3)
    moving = 0
    for f in range(frames):
        x, pose = synth.frame(seed, sensor, f)
        o.push(x, pose)
        b.push(x, pose)
        tag = "%s seed %d frame %d" % (name, seed, f)
        _compare(o, b, tag)
        moving += int(b.detection().sum())
        out_o, out_b = o.filter(), b.filter()
        assert out_o.shape == out_b.shape and np.array_equal(out_o.view(np.uint32), out_b.view(np.uint32)), tag
        # the committed digest of this frame, reproduced WITHOUT the oracle
        assert mod.digest(b, out_b) == fx["digests"]["%s/%d/%d" % (name, seed, f)], tag
    assert o.counts().n_clusters > 10 and o.counts().n_corr > 5 and (moving > 0 or gm == 1)   # (the voxel-ground case's three frames hold no mover yet)
    o.close()


def test_fixture_records_its_provenance():
    fx = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize_digests.json")))
    assert "independent" in fx.get("provenance", "")
