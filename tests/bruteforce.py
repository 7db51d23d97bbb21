"""Definition-level brute force of the hot path in numpy/scipy — the PIN for oracle/mor_oracle.c.

Written independently of the C oracle, straight from the mathematical definitions the reference's
PCL calls reduce to: no kd-tree, no BFS, no octree — dense pairwise
fp32 distances, scipy connected components, python sets.  O(n²): small clouds only.
Reference lines: /root/reference/src/MovingObjectRemoval.cpp (cited per function).
"""
import numpy as np
from scipy.sparse import csr_matrix
from scipy.sparse.csgraph import connected_components

f32 = np.float32


def sqdist_matrix(a, b):
    """fp32 ((dx·dx)+(dy·dy))+(dz·dz), every op individually rounded (FLANN L2_Simple)."""
    a = a.astype(f32)
    b = b.astype(f32)
    d0 = a[:, None, 0] - b[None, :, 0]
    d1 = a[:, None, 1] - b[None, :, 1]
    d2 = a[:, None, 2] - b[None, :, 2]
    r = d0 * d0
    r = r + d1 * d1
    r = r + d2 * d2
    return r


def nn_lowest_index(q, pts):
    """1-NN with ties → lowest index.  Returns (idx, d²)."""
    d = sqdist_matrix(q, pts)
    idx = np.argmin(d, axis=1)  # argmin returns the first (lowest index) minimum
    return idx, d[np.arange(len(q)), idx]


def ground_crop(xyzi, p):
    """:62-88 — returns raw (T,4), cloud_src (indices into raw), gp (indices into raw)."""
    x, y, z = xyzi[:, 0], xyzi[:, 1], xyzi[:, 2]
    X, Y, Z = f32(p.trim_x), f32(p.trim_y), f32(p.trim_z)
    fin = np.isfinite(x) & np.isfinite(y) & np.isfinite(z)
    keep = fin & (x >= -X) & (x <= X) & (y >= -Y) & (y <= Y)
    raw = xyzi[keep]
    zz = raw[:, 2]
    outside = (zz < f32(p.gp_limit)) | (zz > Z)
    return raw, np.nonzero(~outside)[0], np.nonzero(outside)[0]


def ground_voxel(xyzi, p):
    """:90-200, intended semantics with the deterministic definitions of DESIGN.md §G2 — returns raw (T,4),
    cloud_src, gp (sorted unique ground indices into raw).  Pure definition: voxels by integer floor
    coordinates, fp32 sums in the stated orders, neighbours by the dense distance matrix."""
    x, y, z = xyzi[:, 0], xyzi[:, 1], xyzi[:, 2]
    X, Y = f32(p.trim_x), f32(p.trim_y)
    fin = np.isfinite(x) & np.isfinite(y) & np.isfinite(z)
    raw = xyzi[fin & (x >= -X) & (x <= X) & (y >= -Y) & (y <= Y)]
    T = len(raw)
    is_ground = np.zeros(T, bool)
    if T:
        leaf = f32(p.gp_leaf)
        inv = f32(1.0) / leaf
        ijk = np.floor(raw[:, :3].astype(f32) * inv).astype(np.int64)           # absolute voxel coordinates
        order = np.lexsort((np.arange(T), ijk[:, 0], ijk[:, 1], ijk[:, 2]))     # (z, y, x, index)
        keys = ijk[order]
        starts = np.nonzero(np.r_[True, np.any(keys[1:] != keys[:-1], axis=1)])[0]
        ends = np.r_[starts[1:], T]
        rr = np.float64(leaf)
        r2 = f32(rr * rr)
        pts = raw[:, :3].astype(f32)
        accepted = []   # (bin, neighbour indices)
        for s0, e0 in zip(starts, ends):
            idx = order[s0:e0]
            sx = sy = sz = f32(0)
            for i in idx:   # fp32 sequential, ascending point index
                sx = f32(sx + pts[i, 0]); sy = f32(sy + pts[i, 1]); sz = f32(sz + pts[i, 2])
            n = f32(len(idx))
            c = np.array([sx / n, sy / n, sz / n], f32)
            d = sqdist_matrix(c[None, :], pts)[0]
            nb = np.nonzero(d < r2)[0]
            if len(nb) <= 3:
                continue
            nb = nb[np.lexsort((nb, d[nb]))]   # (d², index)
            cx = cy = cz = f32(0)
            for i in nb:
                cx = f32(cx + pts[i, 0]); cy = f32(cy + pts[i, 1]); cz = f32(cz + pts[i, 2])
            fn = f32(len(nb))
            cx, cy, cz = f32(cx / fn), f32(cy / fn), f32(cz / fn)
            c02 = c12 = c22 = f32(0)
            for i in nb:
                dx, dy, dz = f32(pts[i, 0] - cx), f32(pts[i, 1] - cy), f32(pts[i, 2] - cz)
                c12 = f32(c12 + f32(dy * dz)); c22 = f32(c22 + f32(dz * dz)); c02 = f32(c02 + f32(dz * dx))
            if not (abs(np.float64(c02)) < 0.001 and abs(np.float64(c12)) < 0.001 and abs(np.float64(c22)) < 0.001):
                continue
            accepted.append((int(f32(c[2] * f32(10))), nb))   # truncation toward zero
        if accepted:
            bins = {}
            for b, _ in accepted:
                bins[b] = bins.get(b, 0) + 1
            best = min(bins, key=lambda b: (-bins[b], b))   # mode, ties → smallest key
            for b, nb in accepted:
                if b == best:
                    is_ground[nb] = True
    return raw, np.nonzero(~is_ground)[0], np.nonzero(is_ground)[0]


def clusters(cloud_xyz, p):
    """:202-262 — list of index arrays (ascending), ordered by (size desc, first index asc); centroids."""
    n = len(cloud_xyz)
    if n == 0:
        return [], np.zeros((0, 3), f32)
    tol = np.float64(f32(p.ec_distance_threshold))
    r2 = f32(tol * tol)
    adj = sqdist_matrix(cloud_xyz, cloud_xyz) < r2
    nc, lab = connected_components(csr_matrix(adj), directed=False)
    comps = [np.nonzero(lab == c)[0] for c in range(nc)]
    comps = [c for c in comps if p.min_cluster_size <= len(c) <= p.max_cluster_size]
    comps.sort(key=lambda c: (-len(c), c[0]))
    cents = np.zeros((len(comps), 3), f32)
    for k, c in enumerate(comps):
        s = np.zeros(3, np.float64)
        for i in c:  # sequential fp64 sum, as compute3DCentroid<double>
            s += cloud_xyz[i].astype(np.float64)
        cents[k] = (s / np.float64(len(c))).astype(f32)
    return comps, cents


def pose_to_tf(pose):
    x, y, z, w = pose[3:7]
    l2 = x * x + y * y + z * z + w * w
    if abs(l2 - 1.0) > 0.1:
        l = np.sqrt(l2)
        x, y, z, w = x / l, y / l, z / l, w / l
        l2 = x * x + y * y + z * z + w * w
    s = 2.0 / l2
    xs, ys, zs = x * s, y * s, z * s
    wx, wy, wz, xx, xy, xz, yy, yz, zz = w * xs, w * ys, w * zs, x * xs, x * ys, x * zs, y * ys, y * zs, z * zs
    R = np.array([[1 - (yy + zz), xy - wz, xz + wy], [xy + wz, 1 - (xx + zz), yz - wx], [xz - wy, yz + wx, 1 - (xx + yy)]], np.float64)
    return R, np.array(pose[:3], np.float64)


def relative_transform(pose_b, pose_a):
    """:536 t = cb.ps.inverseTimes(ca.ps) → fp32 3x4."""
    Rb, ob = pose_to_tf(pose_b)
    Ra, oa = pose_to_tf(pose_a)
    m = np.zeros((3, 4), np.float64)
    m[:, :3] = Rb.T @ Ra
    m[:, 3] = Rb.T @ (oa - ob)
    return m.astype(f32)


def transform(m, xyz):
    """:540-551 x' = ((m00·x + m01·y) + m02·z) + m03 in fp32."""
    xyz = xyz.astype(f32)
    out = np.empty_like(xyz)
    for r in range(3):
        out[:, r] = ((m[r, 0] * xyz[:, 0] + m[r, 1] * xyz[:, 1]) + m[r, 2] * xyz[:, 2]) + m[r, 3]
    return out


def volume(pts):
    mn, mx = pts.min(0).astype(f32), pts.max(0).astype(f32)
    e = mx - mn
    return np.float64(f32(f32(e[0] * e[1]) * e[2]))


def correspondences(prev_cents, prev_clusters, cur_cents, cur_clusters, p):
    """:285-307 reciprocal 1-NN + volume constraint; returns list of (query, match, d²)."""
    out = []
    if len(prev_cents) == 0 or len(cur_cents) == 0:
        return out
    fwd, dfwd = nn_lowest_index(prev_cents, cur_cents)
    bwd, _ = nn_lowest_index(cur_cents, prev_cents)
    for i in range(len(prev_cents)):
        j = fwd[i]
        if bwd[j] != i:
            continue
        vp, vc = volume(prev_clusters[i]), volume(cur_clusters[j])
        with np.errstate(invalid="ignore", divide="ignore"):
            diff = vp - vc
            ad = np.float64(abs(int(diff))) if (getattr(p, "volume_abs_int", 0) and np.isfinite(diff)) else abs(diff)   # :277 int abs(int) reading vs fabs
            ok = (ad / (vp + vc)) < np.float64(f32(p.volume_constraint))
        if not ok:
            continue
        out.append((i, int(j), dfwd[i]))
    return out


def score_method1(c1, c2, p):
    """:336-366"""
    _, d = nn_lowest_index(c1, c2)
    cnt = int(np.count_nonzero((d > f32(p.pde_lb)) & (d < f32(p.pde_ub))))
    return cnt / float((len(c1) + len(c2)) // 2)


def score_method2(c1, c2, res, anchor_half=0):
    """:309-334 — voxel lattice anchored at c1[0] − res (OctreePointCloud: first point ± res/2, then
    getKeyBitSize re-centres the 2-voxel root), keys by floor in fp64; count c2 points in voxels
    without c1 points."""
    res = np.float64(f32(res))
    p0 = c1[0].astype(np.float64)
    mn = p0 - res / 2
    mx = p0 + res / 2
    over = (2 * res - (mx - mn)) / 2.0
    anchor = mn if anchor_half else np.where(over > np.float64(np.finfo(np.float32).eps), mn - over, mn)   # opc_anchor = 1: the first box is not re-centred
    k1 = np.floor((c1.astype(np.float64) - anchor) / res).astype(np.int64)
    k2 = np.floor((c2.astype(np.float64) - anchor) / res).astype(np.int64)
    s1 = set(map(tuple, k1))
    return float(sum(1 for k in map(tuple, k2) if k not in s1))


class BruteMOR:
    """:516-696 with python containers."""

    def __init__(self, p, n_bad=4, n_good=3):
        self.p, self.moving_confidence, self.static_confidence = p, n_bad, n_good
        self.ca = None
        self.cb = None
        self.corrs_vec, self.res_vec, self.mo = [], [], []

    def push(self, xyzi, pose):
        p = self.p
        xyzi = np.asarray(xyzi, f32).reshape(-1, 4)
        self.ca = self.cb
        raw, cloud_src, gp = (ground_voxel if p.ground_method == 1 else ground_crop)(xyzi, p)
        cloud = raw[cloud_src]
        comps, cents = clusters(cloud[:, :3], p)
        cb = dict(raw=raw, cloud=cloud, cloud_src=cloud_src, gp=gp, comps=comps, cents=cents,
                  cl_pts=[cloud[c, :3].copy() for c in comps], det=np.zeros(len(comps), bool), pose=np.asarray(pose, np.float64))
        self.cb = cb
        self.last_corr, self.last_score = [], []
        ca = self.ca
        if ca is not None:
            m = relative_transform(cb["pose"], ca["pose"])
            ca["cents"] = transform(m, ca["cents"]) if len(ca["cents"]) else ca["cents"]
            ca["cl_pts"] = [transform(m, c) for c in ca["cl_pts"]]
            mp = correspondences(ca["cents"], ca["cl_pts"], cb["cents"], cb["cl_pts"], p)
            for (q, mt, d) in mp:
                c1, c2 = ca["cl_pts"][q], cb["cl_pts"][mt]
                if p.method_choice == 1:
                    s = score_method1(c1, c2, p)
                    thr = np.float64(f32(p.pde_distance_threshold))
                else:
                    s = score_method2(c1, c2, p.opc_resolution, getattr(p, "opc_anchor", 0))
                    thr = float((len(c1) + len(c2)) // p.opc_normalization_factor)
                cb["det"][mt] = s > thr
                self.last_score.append(s)
            self.last_corr = mp
            self._check_chain(mp, ca, cb)

    def _recurse(self, col, track):
        if col == len(self.corrs_vec):
            return track
        for (q, mt, _) in self.corrs_vec[col]:
            if q == track:
                if self.res_vec[col + 1][mt]:
                    return self._recurse(col + 1, mt)
                return -1
        return -1

    def _check_chain(self, mp, ca, cb):
        self.corrs_vec.append(list(mp))
        if len(self.res_vec) == 0:
            self.res_vec.append(ca["det"].copy())
        self.res_vec.append(cb["det"].copy())
        if len(self.res_vec) >= self.moving_confidence:
            for i in range(len(self.res_vec[0])):
                if self.res_vec[0][i]:
                    found = self._recurse(0, i)
                    if found != -1:
                        self._push_centroid(cb["cents"][found])
            self.corrs_vec.pop(0)
            self.res_vec.pop(0)

    def _push_centroid(self, pt):
        for m in self.mo:
            d = (pt - m["c"]).astype(np.float64)  # float - float in fp32, then pow in fp64
            if np.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) < np.float64(f32(self.p.catch_up_distance)):
                return
        self.mo.append(dict(c=pt.copy(), conf=self.static_confidence + 1, maxc=self.static_confidence + 1))

    def filter(self):
        cb, p = self.cb, self.p
        moving = []
        i = 0
        while i < len(self.mo):
            m = self.mo[i]
            if len(cb["cents"]) == 0:
                i += 1
                continue
            nn, d = nn_lowest_index(m["c"][None, :], cb["cents"])
            nn, d = int(nn[0]), d[0]
            moving.extend(cb["comps"][nn].tolist())
            if (not cb["det"][nn]) or d > f32(p.leave_off_distance):
                m["conf"] -= 1
                if m["conf"] == 0:
                    self.mo.pop(i)
                    continue
            else:
                m["c"] = cb["cents"][nn].copy()
                if m["conf"] < m["maxc"]:
                    m["conf"] += 1
            i += 1
        cloud = cb["cloud"]
        if len(moving) > len(cloud):
            kept = cloud[:0]
        else:
            mask = np.ones(len(cloud), bool)
            mask[np.array(moving, np.int64)] = False
            kept = cloud[mask]
        return np.concatenate([kept, cb["raw"][cb["gp"]]], axis=0)

    def labels(self):
        cb = self.cb
        lab = np.full(len(cb["raw"]), -2, np.int32)
        lab[cb["cloud_src"]] = -1
        for k, c in enumerate(cb["comps"]):
            lab[cb["cloud_src"][c]] = k
        return lab
