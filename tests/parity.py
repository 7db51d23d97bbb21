"""Shared HIP-vs-oracle comparison used by the -m gpu tests and by __graft_entry__.smoke().
Bar (BASELINE.json north_star): integer work bit-exact — counts, ground indices, labels, cluster
membership, correspondences, voxel/NN counts (scores are count/int ⇒ exact doubles), detection,
track confidences, and the filtered cloud byte-for-byte; float centroids within 1e-5."""
import numpy as np

CENTROID_TOL = 1e-5


def compare_frame(o, b, s=0, tag=""):
    """o: oracle.Oracle after push; b: MorBatch after push; s: stream index."""
    co, cb = o.counts(), b.counts(s)
    for k in ("n_in", "n_trim", "n_cloud", "n_ground", "n_clusters", "n_clustered", "n_corr", "n_tracks"):
        assert getattr(co, k) == getattr(cb, k), "%s %s: oracle %d hip %d" % (tag, k, getattr(co, k), getattr(cb, k))
    assert np.array_equal(o.ground_indices(), b.ground_indices(s)), tag + " gp_indices"
    assert np.array_equal(o.labels(), b.labels(s)), tag + " labels"
    off_o, idx_o = o.clusters()
    off_b, idx_b = b.clusters(s)
    assert np.array_equal(off_o, off_b) and np.array_equal(idx_o, idx_b), tag + " cluster_indices"
    cen_o, cen_b = o.centroids(), b.centroids(s)
    if len(cen_o):
        assert np.max(np.abs(cen_o.astype(np.float64) - cen_b.astype(np.float64))) <= CENTROID_TOL, tag + " centroids"
    qo, mo, do, so = o.correspondences()
    qb, mb, db, sb = b.correspondences(s)
    assert np.array_equal(qo, qb) and np.array_equal(mo, mb), tag + " correspondences"
    if len(do):
        assert np.max(np.abs(do.astype(np.float64) - db.astype(np.float64))) <= 1e-4, tag + " correspondence distances"
    assert np.array_equal(so, sb), tag + " movement scores"
    assert np.array_equal(o.detection(), b.detection(s)), tag + " detection_results"
    compare_tracks(o, b, s, tag)
    return int(np.count_nonzero(cen_o.view(np.uint32) != cen_b.view(np.uint32))) if len(cen_o) else 0


def compare_tracks(o, b, s=0, tag=""):
    xo, co, mo = o.tracks()
    xb, cb, mb = b.tracks(s)
    assert np.array_equal(co, cb) and np.array_equal(mo, mb), tag + " track confidences"
    if len(xo):
        assert np.max(np.abs(xo.astype(np.float64) - xb.astype(np.float64))) <= CENTROID_TOL, tag + " track centroids"


def compare_output(out_o, out_b, tag=""):
    assert out_o.shape == out_b.shape, "%s filtered cloud size: oracle %s hip %s" % (tag, out_o.shape, out_b.shape)
    assert np.array_equal(out_o.view(np.uint32), out_b.view(np.uint32)), tag + " filtered cloud bytes"


def compare_after_filter(o, b, s=0, tag=""):
    """After filterCloud on both: the tracked centroids (confidences moved by the loop) and the clusters the loop matched them to, in mo_vec order
    (the bounding-box markers the reference publishes at :641)."""
    compare_tracks(o, b, s, tag)
    assert np.array_equal(o.moving_clusters(), b.moving_clusters(s)), tag + " clusters matched by the filterCloud loop"
    return len(o.moving_clusters())
