// kernels_clusters.h — part of mor_kernels.hip (one translation unit: #included there, in dependency order; not a stand-alone header).
// C2, P1, P2 (:221-307, :536-551): cluster extraction, centroids and boxes; transform of the previous frame; centroid correspondences + volume gate.
// Reference citations are file:line of /root/reference/src/MovingObjectRemoval.cpp.
// ------------------------------------------------------------------------------------ C2: per-cluster extraction + centroid + AABB
__device__ __forceinline__ void red6_block(Red6 &r, Red6 *sh) {
  // (min / max by the DPP all-reduce; the fp64 sums keep the shuffle tree: their ORDER of additions is part of the result — lane 0 of this tree is what the partials hold)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { r.sx += __shfl_down(r.sx, o, 64); r.sy += __shfl_down(r.sy, o, 64); r.sz += __shfl_down(r.sz, o, 64); }
  r.mnx = wave_fmin(r.mnx); r.mny = wave_fmin(r.mny); r.mnz = wave_fmin(r.mnz); r.mxx = wave_fmax(r.mxx); r.mxy = wave_fmax(r.mxy); r.mxz = wave_fmax(r.mxz);
  if (lane_id() == 0) sh[wave_id()] = r;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < MOR_BT / 64; ++w) {
      r.sx += sh[w].sx; r.sy += sh[w].sy; r.sz += sh[w].sz;
      r.mnx = fminf(r.mnx, sh[w].mnx); r.mny = fminf(r.mny, sh[w].mny); r.mnz = fminf(r.mnz, sh[w].mnz);
      r.mxx = fmaxf(r.mxx, sh[w].mxx); r.mxy = fmaxf(r.mxy, sh[w].mxy); r.mxz = fmaxf(r.mxz, sh[w].mxz);
    }
  }
  __syncthreads();
}
// Per-cluster reductions are split into chunks of MOR_CHUNK points so a 35 000-point wall does not serialise on one
// workgroup: work item w = (cluster k, chunk c) with chunk_off[k] ≤ w < chunk_off[k+1].  Partials are combined per
// cluster in chunk order by one thread — a fixed reduction tree, so centroids are identical from run to run.
__device__ __forceinline__ int chunk_cluster(const int *coff, int K, int w) {
  int a = 0, b = K;   // last k with coff[k] ≤ w
  while (b - a > 1) { int m = (a + b) >> 1; if (coff[m] <= w) a = m; else b = m; }
  return a;
}
// C2 in one launch (labels, cluster points, centroids, boxes) — no partition of the points by cluster id: the points are
// already grouped by cell, a cluster is a set of whole cells, and k_cg_final gave every cell its piece of the cluster's
// range of cl_pts.  Workgroups [0, tiles_m) of a stream move points: position j of `sorted` → slot j + shift(cell) —
// pieces of consecutive positions, coalesced on both sides — and write the label of cloud point j; workgroups
// [tiles_m, tiles_m + MOR_CLS_G) reduce the cell records (boxes, exact coordinate sums) of every cluster with one wave
// per cluster: centroid = Σ(double)p / n cast to fp32 (:239-243) from the exact sum, AABB for the volume gate.
#ifndef MOR_CLS_G
#define MOR_CLS_G 8
#endif
__device__ __forceinline__ void xform_prev_body(const MorDev &d, int s, int bx, int nbx, Red6 *sh, float *m);
__device__ __forceinline__ void cluster_pairs_body(const MorDev &d, int s, float4 *tile, int *sh);
#ifndef MOR_XF_G
#define MOR_XF_G 16   // workgroups per stream that transform the previous frame's clusters inside this launch
#endif
__global__ __launch_bounds__(MOR_BT) void k_clusters(MorDev d) {
  // the launch: B·g_out movers (shared out by the streams' point counts, map_block_work), then per stream MOR_CLS_G reducers and (with a previous frame) MOR_XF_G transformers
  const int xf_g = d.has_prev ? MOR_XF_G : 0, n_mv = d.B * d.g_out, n_rest = MOR_CLS_G + xf_g;
  int s, t, gmv = 0;
  const bool mover = (int)blockIdx.x < n_mv;
  if (mover) { if (!map_block_work(d, [&](int s_) { return ((int)d.info[s_].M + 4 * MOR_BT - 1) / (4 * MOR_BT); }, s, t, gmv, n_mv, (int)blockIdx.x)) return; }
  else { const int L = (int)blockIdx.x - n_mv; if ((d.B & 7) == 0 && d.xcd_map) { const int x = L & 7, r = L >> 3; s = (r / n_rest) * 8 + x; t = r % n_rest; } else { s = L / n_rest; t = L % n_rest; } s += d.s0; }
  const size_t so = (size_t)s * d.Nmax, ko = (size_t)s * d.Kcap;
  __shared__ Red6 l_red[MOR_BT / 64]; __shared__ float l_m[12]; __shared__ float4 l_tile[MOR_BT]; __shared__ int l_sh8[8], l_last;
  if (!mover && t >= MOR_CLS_G) xform_prev_body(d, s, t - MOR_CLS_G, xf_g, l_red, l_m);   // P1: ca → cb's frame (:536-551), beside the extraction of cb's clusters
  else if (mover) {
    const int M = d.info[s].M;
    float4 *dst = d.cl_pts[d.cur] + so; int *dcid = d.cl_cid[d.cur] + so;
    const int stride = gmv * MOR_BT;
    for (int j0 = t * MOR_BT + threadIdx.x; j0 < M; j0 += 4 * stride) {   // four positions per thread and round trip: point + cell id, then the cell's record, then the stores
      float4 p[4]; int sc[4]; int4 g[4];
      if (d.label_prefill) {   // little of the cloud is clustered (the voxel ground variant's pass B: a seventh): the cell's record first, the point only when it belongs to a cluster
#pragma unroll
        for (int u = 0; u < 4; ++u) sc[u] = d.scell[so + min(j0 + u * stride, M - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) g[u] = d.cgat[so + sc[u]];
#pragma unroll
        for (int u = 0; u < 4; ++u) p[u] = g[u].y >= 0 ? d.sorted[so + min(j0 + u * stride, M - 1)] : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int j = min(j0 + u * stride, M - 1); p[u] = d.sorted[so + j]; sc[u] = d.scell[so + j]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) g[u] = d.cgat[so + sc[u]];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = j0 + u * stride;
        if (j >= M) continue;
        if (g[u].y >= 0 || !d.label_prefill) st_stream(&d.pcid[so + __float_as_int(p[u].w)], g[u].y);   // label of the cloud point (its index travels in .w); read once more, by the output.  (label_prefill: k_gridplace has left −1 in every label — coalesced — and only the clustered points' labels are scattered from here)
        if (g[u].y < 0) continue;
        st_stream(&dst[j + g[u].x], p[u]); st_stream(&dcid[j + g[u].x], g[u].y);   // (cb's cluster points are read by the NEXT frame: streaming stores, they would only push this frame's cell-ordered points out of the L2 before its scoring tiers run)
        if (__float_as_int(p[u].w) == g[u].z) d.cl_first[d.cur][ko + g[u].y] = p[u];
      }
    }
  } else {
  const int K = d.info[s].K, lane = lane_id();
  const int *off = d.cl_off[d.cur] + (size_t)s * (d.Kcap + 1), *lcoff = d.cl_coff + (size_t)s * (d.Kcap + 1);
  for (int k = t * (MOR_BT / 64) + wave_id(); k < K; k += MOR_CLS_G * (MOR_BT / 64)) {
    CellAcc r; acc_clear(r);
    const int e1 = lcoff[k + 1];
    for (int e0 = lcoff[k] + lane; e0 < e1; e0 += 256) {   // four cells per lane and round trip (a wall of 3000 cells is 12 dependent rounds, not 47)
      int c[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) c[u] = d.clist[so + min(e0 + 64 * u, e1 - 1)];
      float4 lo[4], hi[4]; MorCellSum cs[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { lo[u] = d.cmeta[2 * (so + c[u])]; hi[u] = d.cmeta[2 * (so + c[u]) + 1]; cs[u] = d.csum[so + c[u]]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) if (e0 + 64 * u < e1) {
        r.lx = fminf(r.lx, lo[u].x); r.ly = fminf(r.ly, lo[u].y); r.lz = fminf(r.lz, lo[u].z); r.hx = fmaxf(r.hx, hi[u].x); r.hy = fmaxf(r.hy, hi[u].y); r.hz = fmaxf(r.hz, hi[u].z);
#pragma unroll
        for (int a = 0; a < 3; ++a) { r.a[a] += cs[u].a[a]; r.b[a] += cs[u].b[a]; }
      }
    }
    r.lx = wave_fmin(r.lx); r.ly = wave_fmin(r.ly); r.lz = wave_fmin(r.lz); r.hx = wave_fmax(r.hx); r.hy = wave_fmax(r.hy); r.hz = wave_fmax(r.hz);
#pragma unroll
    for (int a = 0; a < 3; ++a) { r.a[a] = wave_sum(r.a[a]); r.b[a] = wave_sum(r.b[a]); }   // (exact integer sums: any order)
    if (lane == 0) {
      const double n = (double)(off[k + 1] - off[k]);
      // (agent-scope stores: the correspondences are worked out by the stream's last workgroup of this launch, stream_last_block)
      st_agent_f4(&d.centroid[d.cur][ko + k], make_float4((float)(fx_value(r.a[0], r.b[0]) / n), (float)(fx_value(r.a[1], r.b[1]) / n), (float)(fx_value(r.a[2], r.b[2]) / n), 0.f));
      st_agent_f4(&d.amin[d.cur][ko + k], make_float4(r.lx, r.ly, r.lz, 0.f));
      st_agent_f4(&d.amax[d.cur][ko + k], make_float4(r.hx, r.hy, r.hz, 0.f));
      st_agent(&d.pair_of_cur[ko + k], -1);
      if (d.method == 2) st_agent(&d.vrec[2 * ko + d.Kcap + k].pr, -1);
    }
  }
  }
  // P2 (:264-307) in the stream's last workgroup to finish: boxes / centroids / first points of the transformed ca, both nearest-centroid
  // directions, the correspondences — everything per CLUSTER between the point kernels (round 2: two more launches, k_xform_prev and k_cluster_pairs)
  // (two tickets: the stream's movers — as many as its share of the launch — among themselves, then their last one with the reducers and transformers)
  if (mover && !stream_last_block(d.tickets + (size_t)s * TK_COUNT + TK_MOVERS, gmv, &l_last)) return;
  if (!stream_last_block(d.tickets + (size_t)s * TK_COUNT + TK_PAIRS, n_rest + 1, &l_last)) return;
  cluster_pairs_body(d, s, l_tile, l_sh8);
}

// ------------------------------------------------------------------------------------ P1: previous frame → current pose (:536-551)
__device__ __forceinline__ void xform(const float *m, float &x, float &y, float &z) {
  float a = x, b = y, c = z;
  x = ((m[0] * a + m[1] * b) + m[2] * c) + m[3];
  y = ((m[4] * a + m[5] * b) + m[6] * c) + m[7];
  z = ((m[8] * a + m[9] * b) + m[10] * c) + m[11];
}
__device__ __forceinline__ void xform_prev_body(const MorDev &d, int s, int bx, int nbx, Red6 *sh, float *m) {
  const int pv = d.prev, K = d.slot_kc[d.prev][s].x;
  if (K == 0) return;
  const size_t so = (size_t)s * d.Nmax;
  const int *off = d.cl_off[pv] + (size_t)s * (d.Kcap + 1), *coff = d.chunk_off[pv] + (size_t)s * (d.Kcap + 1);
  const int W = coff[K];
  if (threadIdx.x < 12) m[threadIdx.x] = d.args[s].xf[threadIdx.x];
  __syncthreads();
  for (int w = bx; w < W; w += nbx) {
    const int k = chunk_cluster(coff, K, w), b = off[k] + (w - coff[k]) * MOR_CHUNK, e = min(off[k + 1], b + MOR_CHUNK);
    Red6 r = {0, 0, 0, FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int j = b + threadIdx.x; j < e; j += MOR_BT) {
      float4 p = d.cl_pts[pv][so + j];
      xform(m, p.x, p.y, p.z);
      d.cl_pts[pv][so + j] = p;
      r.mnx = fminf(r.mnx, p.x); r.mny = fminf(r.mny, p.y); r.mnz = fminf(r.mnz, p.z);
      r.mxx = fmaxf(r.mxx, p.x); r.mxy = fmaxf(r.mxy, p.y); r.mxz = fmaxf(r.mxz, p.z);
    }
    red6_block(r, sh);
    if (threadIdx.x == 0) {   // (agent-scope stores: read by the stream's last workgroup of this launch)
      Red6 *o = &d.part_back[(size_t)s * d.Wcap + w];
      st_agent_f(&o->mnx, r.mnx); st_agent_f(&o->mny, r.mny); st_agent_f(&o->mnz, r.mnz); st_agent_f(&o->mxx, r.mxx); st_agent_f(&o->mxy, r.mxy); st_agent_f(&o->mxz, r.mxz);
    }
  }
}
// AABBs of the transformed clusters (volume gate), transformed centroids (:540-541)
__device__ __forceinline__ void xform_fin_body(const MorDev &d, int s) {
  const int pv = d.prev, K = d.slot_kc[d.prev][s].x;
  const int *coff = d.chunk_off[pv] + (size_t)s * (d.Kcap + 1);
  const Red6 *pt = d.part_back + (size_t)s * d.Wcap;
  const float *m = d.args[s].xf;
  for (int k = threadIdx.x; k < K; k += MOR_BT) {
    Red6 r = {0, 0, 0, FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int w = coff[k]; w < coff[k + 1]; ++w) {
      Red6 q; q.mnx = ld_agent_f(&pt[w].mnx); q.mny = ld_agent_f(&pt[w].mny); q.mnz = ld_agent_f(&pt[w].mnz); q.mxx = ld_agent_f(&pt[w].mxx); q.mxy = ld_agent_f(&pt[w].mxy); q.mxz = ld_agent_f(&pt[w].mxz);
      r.mnx = fminf(r.mnx, q.mnx); r.mny = fminf(r.mny, q.mny); r.mnz = fminf(r.mnz, q.mnz);
      r.mxx = fmaxf(r.mxx, q.mxx); r.mxy = fmaxf(r.mxy, q.mxy); r.mxz = fmaxf(r.mxz, q.mxz);
    }
    float4 c = d.centroid[pv][(size_t)s * d.Kcap + k];
    xform(m, c.x, c.y, c.z);
    d.xcent[(size_t)s * d.Kcap + k] = c;     // ca's own centroids stay as they are: the tail stage of frame k−1 may still be reading them
    float4 p0 = d.cl_first[pv][(size_t)s * d.Kcap + k];
    xform(m, p0.x, p0.y, p0.z);
    d.xfirst[(size_t)s * d.Kcap + k] = p0;
    d.xamin[(size_t)s * d.Kcap + k] = make_float4(r.mnx, r.mny, r.mnz, 0.f);
    d.xamax[(size_t)s * d.Kcap + k] = make_float4(r.mxx, r.mxy, r.mxz, 0.f);
    d.pair_of_prev[(size_t)s * d.Kcap + k] = -1;
    if (d.method == 2) d.vrec[2 * (size_t)s * d.Kcap + k].pr = -1;
    d.qrec[2 * ((size_t)s * d.Kcap + k)] = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));   // no pair (pairs_body fills in the matched ones)
  }
}

// ------------------------------------------------------------------------------------ P2: centroid correspondence (:285-307)
// dir 0: nearest current centroid of every previous centroid; dir 1: the reverse.  Squared fp32
// distance, ties → lowest index (ascending scan with strict <).
__device__ __forceinline__ void nn_centroid_body(const MorDev &d, int s, int dir, float4 *tile) {
  const int Kp = d.slot_kc[d.prev][s].x, Kc = d.info[s].K;
  const int Ksrc = dir == 0 ? Kp : Kc, Kdst = dir == 0 ? Kc : Kp;
  const float4 *cp = d.xcent + (size_t)s * d.Kcap, *cc = d.centroid[d.cur] + (size_t)s * d.Kcap;
  const float4 *src = dir == 0 ? cp : cc, *dst = dir == 0 ? cc : cp;
  for (int i0 = 0; i0 < Ksrc; i0 += MOR_BT) {
    const int i = i0 + threadIdx.x;
    // (cb's centroids were written by other workgroups of this launch: agent-scope loads; ca's transformed ones by this workgroup: plain)
    const float4 q = i < Ksrc ? (dir == 1 ? ld_agent_f4(&src[i]) : src[i]) : make_float4(0, 0, 0, 0);
    float best = INFINITY; int bi = -1;
    for (int b = 0; b < Kdst; b += MOR_BT) {
      __syncthreads();
      if (b + threadIdx.x < Kdst) tile[threadIdx.x] = dir == 0 ? ld_agent_f4(&dst[b + threadIdx.x]) : dst[b + threadIdx.x];
      __syncthreads();
      const int lim = min(MOR_BT, Kdst - b);
      for (int u = 0; u < lim; ++u) { const float dd = sqdist(q.x, q.y, q.z, tile[u].x, tile[u].y, tile[u].z); if (dd < best) { best = dd; bi = b + u; } }
    }
    if (i < Ksrc) {
      if (dir == 0) { d.nn_fwd[(size_t)s * d.Kcap + i] = bi; d.nn_fwd_d[(size_t)s * d.Kcap + i] = best; }
      else d.nn_bwd[(size_t)s * d.Kcap + i] = bi;
    }
  }
}
// lattice origin of a pair's voxel set: the first point p0 of the previous cluster (after the transform) − res/2, moved down by what getKeyBitSize adds (fp64, per axis)
__device__ __forceinline__ void vox_anchor(const MorDev &d, const float4 &p0, double (&mn)[3]) {
  const double res = d.opc_res, eps = (double)FLT_EPSILON;
  const float p0c[3] = {p0.x, p0.y, p0.z};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    double lo = (double)p0c[a] - res / 2, hi = (double)p0c[a] + res / 2;
    const double over = (2.0 * res - (hi - lo)) / 2.0;
    if (over > eps && !d.opc_anchor_half) lo -= over;   // getKeyBitSize on the empty tree re-centres the first box (mor_params.opc_anchor = 1: it does not)
    mn[a] = lo;
  }
}
// reciprocal test + volumeConstraint (:264-283), correspondences emitted in source-index order
__device__ __forceinline__ void pairs_body(const MorDev &d, int s, int *sh) {
  const int Kp = d.slot_kc[d.prev][s].x, Kc = d.info[s].K;
  int carry = 0;
  const size_t ko = (size_t)s * d.Kcap;
  for (int b = 0; b < Kp; b += MOR_BT) {
    int i = b + threadIdx.x, ok = 0, j = -1;
    if (i < Kp && Kc > 0) {
      j = d.nn_fwd[ko + i];
      if (j >= 0 && d.nn_bwd[ko + j] == i) {
        float4 a0 = d.xamin[ko + i], a1 = d.xamax[ko + i], c0 = ld_agent_f4(&d.amin[d.cur][ko + j]), c1 = ld_agent_f4(&d.amax[d.cur][ko + j]);
        float vp = (a1.x - a0.x) * (a1.y - a0.y); vp = vp * (a1.z - a0.z);
        float vc = (c1.x - c0.x) * (c1.y - c0.y); vc = vc * (c1.z - c0.z);
        double dp = (double)vp, dc = (double)vc;
        const double diff = dp - dc, ad = d.vol_abs_int ? (double)abs((int)diff) : fabs(diff);   // :277 unqualified abs: fabs (libstdc++ ≥ 6, default) or int abs(int)
        ok = (ad / (dp + dc)) < d.vol_thr;   // NaN (0/0) compares false ⇒ rejected, as in the reference
      }
    }
    int tot, e = block_excl_scan(ok, sh, &tot);
    if (ok) {
      int pr = carry + e;
      d.pair_q[ko + pr] = i; d.pair_m[ko + pr] = j; d.pair_d[ko + pr] = d.nn_fwd_d[ko + i]; d.pair_cnt[ko + pr] = 0;
      d.pair_of_prev[ko + i] = pr; d.pair_of_cur[ko + j] = pr;
      if (d.method == 2) {   // (ca's record and cb's: the lattice hangs on ca's first point after the transform, xfirst — written by this workgroup, above)
        MorVoxRec vr; vox_anchor(d, d.xfirst[ko + i], vr.mn); vr.pr = pr; vr.pad = 0;
        d.vrec[2 * ko + i] = vr; d.vrec[2 * ko + d.Kcap + j] = vr;
      }
      const float4 c0 = ld_agent_f4(&d.amin[d.cur][ko + j]), c1 = ld_agent_f4(&d.amax[d.cur][ko + j]);
      d.qrec[2 * (ko + i)] = make_float4(c0.x, c0.y, c0.z, __int_as_float(pr)); d.qrec[2 * (ko + i) + 1] = make_float4(c1.x, c1.y, c1.z, __int_as_float(j));
    }
    carry += tot;
  }
  if (threadIdx.x == 0) { d.info[s].n_pairs = carry; d.wl_nb[s] = 0ull; d.wl2_n[s] = 0; }
}
// One workgroup per stream: everything per CLUSTER between the point kernels — boxes, centroids and first points of the
// transformed ca (from the partials of k_clusters' transform workgroups), both nearest-centroid directions, the correspondences.
__device__ __forceinline__ void cluster_pairs_body(const MorDev &d, int s, float4 *tile, int *sh) {
  if (!d.has_prev) return;
  xform_fin_body(d, s);
  __threadfence_block();
  __syncthreads();
  nn_centroid_body(d, s, 0, tile);
  nn_centroid_body(d, s, 1, tile);
  __threadfence_block();
  __syncthreads();
  pairs_body(d, s, sh);
}

#define MOR_SCORE_G 64    // workgroups per stream of the worklist tiers
#define MOR_PDE_G 256
