"""Independent definition of the hot path at FULL size (120 000 / 262 144-point clouds) — the pin for
oracle/mor_oracle.c where the dense brute force of tests/bruteforce.py (O(n²) memory) stops.

Same mathematical definitions, nothing shared with the C oracle: candidate neighbours come from
scipy.spatial.cKDTree in fp64 with a widened radius / several nearest neighbours, and every candidate is
re-evaluated with the exact fp32 predicate (np.float32 operations in the reference's order), so the tree only
prunes — it never decides.  Components by scipy.sparse.csgraph.connected_components; correspondences, volume gate,
octree-change voxels, tracking and filterCloud are the definition-level code of tests/bruteforce.py.
Reference lines: /root/reference/src/MovingObjectRemoval.cpp (:202-218 clustering, :336-366 method 1)."""
import types

import numpy as np
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components
from scipy.spatial import cKDTree

import bruteforce as bf

f32 = np.float32
MAX_PAIRS = 40_000_000   # candidate pairs the test is willing to hold (a wall next to the sensor puts thousands of points into one cell)


def _sqdist_rows(a, b):
    """fp32 ((dx·dx)+(dy·dy))+(dz·dz) per row, every operation individually rounded (FLANN L2_Simple)."""
    d0, d1, d2 = a[:, 0] - b[:, 0], a[:, 1] - b[:, 1], a[:, 2] - b[:, 2]
    r = d0 * d0
    r = r + d1 * d1
    r = r + d2 * d2
    return r


def clusters(cloud_xyz, p):
    """:202-262 — connected components of {i~j ⇔ d²(i,j) < r²} (fp32, strict), size-filtered, ordered by (size desc, first index asc)."""
    n = len(cloud_xyz)
    if n == 0:
        return [], np.zeros((0, 3), f32)
    pts = np.ascontiguousarray(cloud_xyz, f32)
    tol = np.float64(f32(p.ec_distance_threshold))
    r2 = f32(tol * tol)
    tree = cKDTree(pts.astype(np.float64))
    if tree.count_neighbors(tree, tol * 1.001) > 2 * MAX_PAIRS + n:
        raise MemoryError("too many candidate pairs for the independent check")
    pairs = tree.query_pairs(tol * 1.001, output_type="ndarray")
    keep = _sqdist_rows(pts[pairs[:, 0]], pts[pairs[:, 1]]) < r2
    e = pairs[keep]
    nc, lab = connected_components(coo_matrix((np.ones(len(e), np.int8), (e[:, 0], e[:, 1])), shape=(n, n)), directed=False)
    order = np.argsort(lab, kind="stable")
    bounds = np.flatnonzero(np.r_[True, lab[order][1:] != lab[order][:-1], True])
    comps = [order[bounds[i]:bounds[i + 1]] for i in range(nc)]   # ascending indices inside a component (stable sort)
    comps = [c for c in comps if p.min_cluster_size <= len(c) <= p.max_cluster_size]
    comps.sort(key=lambda c: (-len(c), c[0]))
    cents = np.zeros((len(comps), 3), f32)
    for k, c in enumerate(comps):
        s = np.zeros(3, np.float64)
        for row in pts[c].astype(np.float64):   # sequential fp64 sum in index order, as compute3DCentroid<double>
            s += row
        cents[k] = (s / np.float64(len(c))).astype(f32)
    return comps, cents


def score_method1(c1, c2, p):
    """:336-366 — per point of c1 the squared fp32 distance to its nearest point of c2; count lb < d² < ub; / ((n1+n2)/2)."""
    c1, c2 = np.ascontiguousarray(c1, f32), np.ascontiguousarray(c2, f32)
    k = min(8, len(c2))
    _, nb = cKDTree(c2.astype(np.float64)).query(c1.astype(np.float64), k=k)
    nb = nb.reshape(len(c1), k)
    d = np.full(len(c1), np.inf, f32)
    for j in range(k):   # the fp32 minimum is among the few fp64-nearest candidates
        d = np.minimum(d, _sqdist_rows(c1, c2[nb[:, j]]))
    cnt = int(np.count_nonzero((d > f32(p.pde_lb)) & (d < f32(p.pde_ub))))
    return cnt / float((len(c1) + len(c2)) // 2)


def _seq_sum32(v):
    """fp32 sum of v in storage order, every add rounded (np.cumsum is a sequential recurrence in the array's dtype)."""
    return np.cumsum(v, dtype=f32)[-1]


def ground_voxel(xyzi, p):
    """:90-200 with the deterministic definitions of DESIGN.md §4.6 at full size: the neighbours of a voxel centroid come from a
    cKDTree ball query with a widened radius and are re-tested with the exact fp32 predicate; every ordered fp32 sum is a cumulative
    sum in float32.  Nothing shared with tests/bruteforce.py::ground_voxel but the definitions."""
    x, y, z = xyzi[:, 0], xyzi[:, 1], xyzi[:, 2]
    X, Y = f32(p.trim_x), f32(p.trim_y)
    fin = np.isfinite(x) & np.isfinite(y) & np.isfinite(z)
    raw = xyzi[fin & (x >= -X) & (x <= X) & (y >= -Y) & (y <= Y)]
    T = len(raw)
    is_ground = np.zeros(T, bool)
    if T:
        leaf = f32(p.gp_leaf)
        inv = f32(1.0) / leaf
        pts = np.ascontiguousarray(raw[:, :3], f32)
        ijk = np.floor(pts * inv).astype(np.int64)
        order = np.lexsort((np.arange(T), ijk[:, 0], ijk[:, 1], ijk[:, 2]))   # voxels in (z, y, x) order, ascending point index inside a voxel
        keys = ijk[order]
        starts = np.flatnonzero(np.r_[True, np.any(keys[1:] != keys[:-1], axis=1)])
        ends = np.r_[starts[1:], T]
        cents = np.empty((len(starts), 3), f32)
        for v, (s0, e0) in enumerate(zip(starts, ends)):
            q = pts[order[s0:e0]]
            n = f32(e0 - s0)
            cents[v] = (_seq_sum32(q[:, 0]) / n, _seq_sum32(q[:, 1]) / n, _seq_sum32(q[:, 2]) / n)
        rr = np.float64(leaf)
        r2 = f32(rr * rr)
        tree = cKDTree(pts.astype(np.float64))
        cand = tree.query_ball_point(cents.astype(np.float64), rr * 1.001, return_sorted=True)
        accepted = []
        for v, nb in enumerate(cand):
            nb = np.asarray(nb, np.int64)
            if len(nb) <= 3:
                continue
            c = cents[v]
            d = _sqdist_rows(np.broadcast_to(c, (len(nb), 3)), pts[nb])
            keep = d < r2
            nb, d = nb[keep], d[keep]
            if len(nb) <= 3:
                continue
            o = np.lexsort((nb, d))   # (d², index), as radiusSearch returns them
            q = pts[nb[o]]
            fn = f32(len(nb))
            cx, cy, cz = _seq_sum32(q[:, 0]) / fn, _seq_sum32(q[:, 1]) / fn, _seq_sum32(q[:, 2]) / fn
            dx, dy, dz = q[:, 0] - cx, q[:, 1] - cy, q[:, 2] - cz
            c12, c22, c02 = _seq_sum32(dy * dz), _seq_sum32(dz * dz), _seq_sum32(dz * dx)
            if abs(np.float64(c02)) < 0.001 and abs(np.float64(c12)) < 0.001 and abs(np.float64(c22)) < 0.001:
                accepted.append((int(f32(c[2] * f32(10))), nb))
        if accepted:
            cnt = {}
            for b, _ in accepted:
                cnt[b] = cnt.get(b, 0) + 1
            best = min(cnt, key=lambda b: (-cnt[b], b))
            for b, nb in accepted:
                if b == best:
                    is_ground[nb] = True
    return raw, np.flatnonzero(~is_ground), np.flatnonzero(is_ground)


class Counts(types.SimpleNamespace):
    pass


class IndependentMOR(bf.BruteMOR):
    """BruteMOR with the O(n²) pieces (clustering, method-1 scores, the voxel-covariance ground removal) replaced by the tree-pruned exact versions above, and the read-backs of
    oracle.Oracle so that the same digest / comparison code applies."""

    def push(self, xyzi, pose):
        saved = bf.clusters, bf.score_method1, bf.ground_voxel
        bf.clusters, bf.score_method1, bf.ground_voxel = clusters, score_method1, ground_voxel
        try:
            super().push(xyzi, pose)
        finally:
            bf.clusters, bf.score_method1, bf.ground_voxel = saved
        self._n_in = len(np.asarray(xyzi).reshape(-1, 4))

    def counts(self):
        cb = self.cb
        return Counts(n_in=self._n_in, n_trim=len(cb["raw"]), n_cloud=len(cb["cloud"]), n_ground=len(cb["gp"]), n_clusters=len(cb["comps"]),
                      n_clustered=int(sum(len(c) for c in cb["comps"])), n_corr=len(self.last_corr), n_tracks=len(self.mo))

    def ground_indices(self):
        return self.cb["gp"].astype(np.int32)

    def clusters(self):
        comps = self.cb["comps"]
        off = np.zeros(len(comps) + 1, np.int32)
        off[1:] = np.cumsum([len(c) for c in comps])
        return off, (np.concatenate(comps).astype(np.int32) if comps else np.zeros(0, np.int32))

    def centroids(self):
        return self.cb["cents"]

    def detection(self):
        return self.cb["det"].astype(np.uint8)

    def correspondences(self):
        q = np.array([c[0] for c in self.last_corr], np.int32)
        m = np.array([c[1] for c in self.last_corr], np.int32)
        d = np.array([c[2] for c in self.last_corr], np.float32)
        return q, m, d, np.array(self.last_score, np.float64)

    def tracks(self):
        xyz = np.array([t["c"] for t in self.mo], np.float32).reshape(-1, 3)
        return xyz, np.array([t["conf"] for t in self.mo], np.int32), np.array([t["maxc"] for t in self.mo], np.int32)
