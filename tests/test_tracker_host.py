"""Host tracker (csrc/mor_tracker.cpp: checkMovingClusterChain / recurseFindClusterChain /
pushCentroid and filterCloud's loop) against the oracle, driven with the oracle's own per-frame
cluster summaries.  CPU only — a test-only host statement of the rules the device kernels k_track_push / k_track_filter implement (tests/host_tracker/)."""
import numpy as np
import pytest

from host_tracker import HostTracker
from oracle.oracle import Oracle
from scenes import scene_params, small_stream


@pytest.mark.parametrize("method,seed", [(1, 1), (2, 1), (2, 2), (1, 3), (2, 4)])
def test_tracker_follows_oracle(method, seed):
    p = scene_params(method_choice=method)
    o, t = Oracle(p, 4, 3), HostTracker(p, 4, 3)
    saw_tracks = False
    for f, (pts, pose) in enumerate(small_stream(seed, n_frames=9)):
        o.push(pts, pose)
        q, m, _, _ = o.correspondences()
        t.push(o.centroids(), o.detection(), None if f == 0 else list(zip(q, m)))
        xo, co, mo = o.tracks()
        xt, ct, mt = t.tracks()
        assert np.array_equal(xo, xt) and np.array_equal(co, ct) and np.array_equal(mo, mt), "after push, frame %d" % f
        off, _ = o.clusters()
        n_cloud, n_ground = o.counts().n_cloud, o.counts().n_ground
        moving, n_idx = t.filter(np.diff(off))
        out = o.filter()
        removed = int(np.diff(off)[moving.astype(bool)].sum()) if n_idx <= n_cloud else n_cloud
        assert len(out) == n_cloud - removed + n_ground, "frame %d" % f
        xo, co, mo = o.tracks()
        xt, ct, mt = t.tracks()
        assert np.array_equal(xo, xt) and np.array_equal(co, ct) and np.array_equal(mo, mt), "after filter, frame %d" % f
        saw_tracks |= len(co) > 0
    if (method, seed) == (2, 1):
        assert saw_tracks


def test_scripted_chain():
    """Known-answer sequence for the chain logic with n_bad=3: a cluster flagged moving in three
    consecutive frames along a correspondence chain 0→1→0 becomes a track with confidence n_good+1;
    a broken chain does not."""
    p = scene_params()
    t = HostTracker(p, 3, 2)
    c = np.array([[0, 0, 0], [1, 1, 1]], np.float32)
    t.push(c, [0, 0], None)
    t.push(c, [1, 0], [(0, 0), (1, 1)])          # res_vec = [ca(0,0), (1,0)]
    assert len(t.tracks()[1]) == 0
    t.push(c, [0, 1], [(0, 1), (1, 0)])          # size 3 ≥ n_bad: oldest frame has nothing moving → no track
    assert len(t.tracks()[1]) == 0
    t.push(c, [1, 0], [(1, 0), (0, 1)])          # oldest = (1,0): 0 →(0,1)→ det[1]=1 →(1,0)→ det[0]=1 → track at centroid 0
    xyz, conf, mx = t.tracks()
    assert len(conf) == 1 and conf[0] == 3 and mx[0] == 3 and np.array_equal(xyz[0], c[0])
    # filter: nearest centroid is cluster 0, flagged moving, d²=0 ≤ leave_off → confidence stays capped
    moving, n_idx = t.filter([5, 7])
    assert list(moving) == [1, 0] and n_idx == 5
    assert t.tracks()[1][0] == 3
