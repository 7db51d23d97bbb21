"""Hand-checkable scenes of SURVEY §8c(3): the expected results follow from the definitions alone (strict d² < r²,
size limits inclusive, method-1 count of lb < d²min < ub, first-frame output layout), so they pin the oracle — and
the HIP path — without any reference binary.  `make(p)` returns an engine-like object with push / filter / counts /
labels / correspondences (the CPU Oracle or a one-stream MorBatch wrapped by the caller)."""
import numpy as np

POSE0 = np.array([0, 0, 0, 0, 0, 0, 1.0])


def line(x0, n, dy=0.05, y0=0.0, z=0.0):
    """n points spaced dy along y at x = x0."""
    return np.column_stack([np.full(n, x0), y0 + np.arange(n) * dy, np.full(n, z), np.full(n, 0.25)]).astype(np.float32)


def check_all(make, params):
    p = params(min_cluster_size=5)
    r = np.float32(p.ec_distance_threshold)            # 0.11
    r2 = np.float32(np.float64(r) * np.float64(r))     # what EuclideanClusterExtraction compares against

    # (1) two lines 0.12 m apart at r = 0.11 → 2 clusters; 0.10 m apart → 1 cluster
    for gap, want in ((0.12, 2), (0.10, 1)):
        e = make(p)
        e.push(np.concatenate([line(0.0, 10), line(gap, 10)]), POSE0)
        assert e.counts().n_clusters == want, gap
        e.close()

    # (2) strict '<': find x with fp32 d² == r² exactly → the two groups stay apart; one ulp less → they merge
    #     groups: 6 collinear points each (spacing 0.05 < r), group B shifted by x along x
    lo, hi = np.float32(0.0), np.float32(1.0)
    x_eq = None
    for cand in np.nextafter(np.float32(r), np.float32(0)) + np.arange(-8, 9, dtype=np.float32) * np.spacing(np.float32(r)):
        cand = np.float32(cand)
        if np.float32(cand * cand) == r2:
            x_eq = cand
            break
    assert x_eq is not None, "no fp32 x with x·x == r² near r (pick another r for this check)"
    for x, want in ((x_eq, 2), (np.nextafter(x_eq, np.float32(0)), 1)):
        while want == 1 and np.float32(x * x) >= r2:    # walk down to the first x whose square is below r²
            x = np.nextafter(x, np.float32(0))
        e = make(p)
        e.push(np.concatenate([line(0.0, 6), line(float(x), 6)]), POSE0)
        assert e.counts().n_clusters == want, (float(x), want)
        e.close()

    # (3) min / max cluster size are inclusive bounds: 4 | 5 points with min = 5; 12 | 13 points with max = 12
    e = make(p)
    e.push(np.concatenate([line(0.0, 4), line(1.0, 5)]), POSE0)
    c = e.counts()
    assert c.n_clusters == 1 and c.n_clustered == 5
    # (5) first-frame output = [non-ground in input order ‖ ground in input order] of the trimmed input
    out = e.filter()
    assert len(out) == 9
    e.close()
    p2 = params(min_cluster_size=5)
    p2.max_cluster_size = 12
    e = make(p2)
    e.push(np.concatenate([line(0.0, 12), line(1.0, 13)]), POSE0)
    c = e.counts()
    assert c.n_clusters == 1 and c.n_clustered == 12
    e.close()

    # (4) method 1: a cluster translated by t between two frames with identical poses.  A 4 × 7 × 3 lattice with
    #     spacing 0.05 shifted by t = 0.02 along x: the nearest neighbour of every previous point is its own shifted
    #     copy (0.02 < 0.05 − 0.02), so d²min ≈ 0.0004 for all n = 84 points.  lb < 0.0004 < ub → all counted →
    #     score = n / ((n + n) / 2) = 1; lb above or ub below 0.0004 → none counted → 0.  (A flat line would have a
    #     zero-volume AABB and be rejected by the volume gate: NaN ratio, reference .cpp:272-277.)
    g = np.stack(np.meshgrid(np.arange(4) * 0.05, np.arange(7) * 0.05, np.arange(3) * 0.05, indexing="ij"), -1).reshape(-1, 3)
    a = np.column_stack([g, np.full(len(g), 0.5)]).astype(np.float32)
    n = len(a)
    for lb, ub, want in ((0.0001, 0.01, 1.0), (0.001, 0.01, 0.0), (0.00001, 0.0002, 0.0)):
        pm = params(min_cluster_size=5)
        pm.method_choice = 1
        pm.pde_lb, pm.pde_ub = lb, ub
        e = make(pm)
        e.push(a, POSE0)
        e.filter()
        b = a.copy()
        b[:, 0] += np.float32(0.02)
        e.push(b, POSE0)
        c = e.counts()
        assert c.n_clusters == 1 and c.n_clustered == n
        q, m, dist, score = e.correspondences()
        assert len(q) == 1 and q[0] == 0 and m[0] == 0
        assert abs(float(dist[0]) - 0.0004) < 1e-6             # squared fp32 distance of the two centroids
        assert score[0] == want, (lb, ub, score[0])
        e.close()

    # (5b) first-frame output layout on a cloud with ground, trimmed-away and NaN points
    pg = params(min_cluster_size=5)
    pts = np.concatenate([line(0.0, 8), line(0.5, 3, z=float(pg.gp_limit) - 0.05), line(50.0, 4), line(0.2, 2)]).astype(np.float32)
    pts[-1, 0] = np.nan
    rng = np.random.default_rng(3)
    pts = pts[rng.permutation(len(pts))]
    e = make(pg)
    e.push(pts, POSE0)
    out = e.filter()
    fin = np.isfinite(pts[:, :3]).all(1)
    inside = fin & (np.abs(pts[:, 0]) <= pg.trim_x) & (np.abs(pts[:, 1]) <= pg.trim_y)
    trimmed = pts[inside]
    ground = (trimmed[:, 2] < pg.gp_limit) | (trimmed[:, 2] > pg.trim_z)
    want = np.concatenate([trimmed[~ground], trimmed[ground]])
    assert out.shape == want.shape and np.array_equal(out.view(np.uint32), want.view(np.uint32))
    e.close()
