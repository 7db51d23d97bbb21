// kernels_cellgraph.h — part of mor_kernels.hip (one translation unit: #included there, in dependency order; not a stand-alone header).
// C1 (:202-218): Euclidean clustering = connected components over cells — cell records (k_cellboxes), slab forests in LDS (k_cg_slab), their merge (k_cg_final).
// Reference citations are file:line of /root/reference/src/MovingObjectRemoval.cpp.
// ------------------------------------------------------------------------------------ C1: Euclidean clustering = connected components over cells
// Per-cell kernels over global memory are bound by chains of dependent loads (key → row table → key → parent →
// parent …, ≈ 1–2 µs a hop), so the cell graph is worked on in LDS: union-find forests with LDS atomics (cg_find /
// cg_unite below, also usable on global arrays with agent-scope accesses), distinct keys and row tables staged per
// workgroup; only the point coordinates of the few pairs nothing cheaper decides come from L2 / HBM.
template <bool LDS> __device__ __forceinline__ int cg_ld(const int *p) {
  return LDS ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool LDS> __device__ __forceinline__ void cg_st(int *p, int v) {
  if (LDS) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool LDS> __device__ __forceinline__ int cg_find(int *P, int x) {
  int p = cg_ld<LDS>(P + x);
  while (p != x) { int gp = cg_ld<LDS>(P + p); if (gp == p) return p; cg_st<LDS>(P + x, gp); x = gp; p = cg_ld<LDS>(P + x); }
  return x;
}
template <bool LDS> __device__ __forceinline__ int cg_unite(int *P, int a, int b) {
  int ra = cg_find<LDS>(P, a), rb = cg_find<LDS>(P, b);
  while (ra != rb) {
    if (ra < rb) { int x = ra; ra = rb; rb = x; }
    int old = atomicCAS(P + ra, ra, rb);
    if (old == ra) return rb;
    ra = cg_find<LDS>(P, old);
  }
  return ra;
}
// any pair (a ∈ A, b ∈ B) with d² < r²?  one thread; points fetched in blocks of 8 (B) × 4 (A) independent loads so
// a 16 × 16 test costs ≈10 memory round trips instead of 64
__device__ __forceinline__ bool pair_hit_serial(const float4 *sp, int a0, int na, int b0, int nb, float r2) {
  for (int ib0 = 0; ib0 < nb; ib0 += 8) {
    float4 q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) q[j] = sp[b0 + min(ib0 + j, nb - 1)];
    for (int ia0 = 0; ia0 < na; ia0 += 4) {
      float4 pa[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) pa[i] = sp[a0 + min(ia0 + i, na - 1)];
      bool hit = false;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) hit |= sqdist(pa[i].x, pa[i].y, pa[i].z, q[j].x, q[j].y, q[j].z) < r2;   // clamped duplicates repeat real pairs
      if (hit) return true;
    }
  }
  return false;
}
// the same question answered by a whole wave: lanes tile A × B as at × (64/at)
// squared distance from p to the box [lo, hi], scaled down a hair: ≥ r² ⇒ no point of the box is within r of p
__device__ __forceinline__ float point_box_gap2(const float4 &p, const float4 &lo, const float4 &hi) {
  float gx = fmaxf(fmaxf(lo.x - p.x, p.x - hi.x), 0.f), gy = fmaxf(fmaxf(lo.y - p.y, p.y - hi.y), 0.f), gz = fmaxf(fmaxf(lo.z - p.z, p.z - hi.z), 0.f);
  return (gx * gx + gy * gy + gz * gz) * 0.999f;
}
// any pair (a ∈ A, b ∈ B) with d² < r²?  One wave, for the pairs of big cells the thread-level sample could not decide
// (mostly true non-edges: two dense surfaces a cell apart).  A and B are read in coalesced chunks of 64 points; a point
// takes part only if it lies within r of the OTHER cell's point box, which removes nearly everything when the cells
// are two apart; the surviving B points of a chunk are broadcast by shuffles — no memory access in the inner loop.
// min / max of a value over the wave (all lanes get the result)
// (wave_fmin / wave_fmax: DPP + permlane swaps, kernels_common.h)
__device__ __forceinline__ bool pair_hit_wave(const float4 *sp, int a0, int na, int b0, int nb, float r2, int lane,
                                              const float4 &alo, const float4 &ahi, const float4 &blo, const float4 &bhi) {
  if (nb <= 256) {   // (a few chunks of B: the plain form)
    for (int ia0 = 0; ia0 < na; ia0 += 64) {
      const int ia = ia0 + lane; const float4 pa = sp[a0 + min(ia, na - 1)];
      const bool a_act = ia < na && point_box_gap2(pa, blo, bhi) < r2;
      if (!__ballot(a_act)) continue;
      for (int ib0 = 0; ib0 < nb; ib0 += 64) {
        const int ib = ib0 + lane; const float4 pb = sp[b0 + min(ib, nb - 1)];
        unsigned long long mb = __ballot(ib < nb && point_box_gap2(pb, alo, ahi) < r2);
        bool hit = false;
        while (mb) {
          const int l = __ffsll((long long)mb) - 1; mb &= mb - 1;
          const float bx = wave_bcast(pb.x, l), by = wave_bcast(pb.y, l), bz = wave_bcast(pb.z, l);   // (l comes from a ballot: uniform — v_readlane, no LDS)
          hit |= a_act && sqdist(pa.x, pa.y, pa.z, bx, by, bz) < r2;
        }
        if (__ballot(hit)) return true;
      }
    }
    return false;
  }
  // Two BIG cells (the 128-beam clouds put thousands of returns into a 28-cm cell next to the sensor: 15 M point pairs per step in 110 such non-edges, each proven by ONE wave —
  // 320 of k_cg_slab's 440 µs on `os128_b64`, found by cutting the kernel's phases out one at a time).  Points arrive in a cell in scan order, so 64 consecutive ones are a short
  // arc with a tight box: the boxes of B's chunks go to the lanes (lane l ← chunk l, 64 chunks at a time), and a chunk of A is compared with the chunk BOXES first — only chunk
  // pairs whose boxes are closer than r are looked at point by point.  Two parallel surfaces more than r apart are proven apart by a few hundred box tests.
  for (int cb0 = 0; cb0 < nb; cb0 += 64 * 64) {
    const int nch = min(64, (nb - cb0 + 63) >> 6);
    float clx = FLT_MAX, cly = FLT_MAX, clz = FLT_MAX, chx = -FLT_MAX, chy = -FLT_MAX, chz = -FLT_MAX;   // lane l: box of chunk l of this block
    for (int k = 0; k < nch; ++k) {
      const float4 p = sp[b0 + min(cb0 + k * 64 + lane, nb - 1)];   // (clamped lanes repeat a real point of the cell: the box stays a box of B's points)
      const float lx = wave_fmin(p.x), ly = wave_fmin(p.y), lz = wave_fmin(p.z), hx = wave_fmax(p.x), hy = wave_fmax(p.y), hz = wave_fmax(p.z);
      if (lane == k) { clx = lx; cly = ly; clz = lz; chx = hx; chy = hy; chz = hz; }
    }
    for (int ia0 = 0; ia0 < na; ia0 += 64) {
      const int ia = ia0 + lane; const float4 pa = sp[a0 + min(ia, na - 1)];
      const bool a_act = ia < na && point_box_gap2(pa, blo, bhi) < r2;
      if (!__ballot(a_act)) continue;
      // box of this chunk's candidates
      const float qlx = wave_fmin(a_act ? pa.x : FLT_MAX), qly = wave_fmin(a_act ? pa.y : FLT_MAX), qlz = wave_fmin(a_act ? pa.z : FLT_MAX);
      const float qhx = wave_fmax(a_act ? pa.x : -FLT_MAX), qhy = wave_fmax(a_act ? pa.y : -FLT_MAX), qhz = wave_fmax(a_act ? pa.z : -FLT_MAX);
      const float4 qlo = make_float4(qlx, qly, qlz, 0.f), qhi = make_float4(qhx, qhy, qhz, 0.f);
      const float gx = fmaxf(fmaxf(clx - qhx, qlx - chx), 0.f), gy = fmaxf(fmaxf(cly - qhy, qly - chy), 0.f), gz = fmaxf(fmaxf(clz - qhz, qlz - chz), 0.f);
      unsigned long long mc = __ballot(lane < nch && (gx * gx + gy * gy + gz * gz) * 0.999f < r2);   // chunks of B whose box comes within r of this chunk's
      while (mc) {
        const int k = __ffsll((long long)mc) - 1; mc &= mc - 1;
        const int ib = cb0 + k * 64 + lane; const float4 pb = sp[b0 + min(ib, nb - 1)];
        unsigned long long mb = __ballot(ib < nb && point_box_gap2(pb, qlo, qhi) < r2);
        bool hit = false;
        while (mb) {
          const int l = __ffsll((long long)mb) - 1; mb &= mb - 1;
          const float bx = wave_bcast(pb.x, l), by = wave_bcast(pb.y, l), bz = wave_bcast(pb.z, l);
          hit |= a_act && sqdist(pa.x, pa.y, pa.z, bx, by, bz) < r2;
        }
        if (__ballot(hit)) return true;
      }
    }
  }
  return false;
}
__device__ __forceinline__ void wave_box(const float4 *sp, int b, int e, int lane, float4 &lo, float4 &hi, int &mi) {
  mi = 0x7fffffff;
  float lx = FLT_MAX, ly = FLT_MAX, lz = FLT_MAX, hx = -FLT_MAX, hy = -FLT_MAX, hz = -FLT_MAX;
  for (int k = b + lane; k < e; k += 256) {   // four independent loads per lane and step
    const float4 p0 = sp[k], p1 = sp[min(k + 64, e - 1)], p2 = sp[min(k + 128, e - 1)], p3 = sp[min(k + 192, e - 1)];
    lx = fminf(fminf(lx, p0.x), fminf(p1.x, fminf(p2.x, p3.x))); ly = fminf(fminf(ly, p0.y), fminf(p1.y, fminf(p2.y, p3.y))); lz = fminf(fminf(lz, p0.z), fminf(p1.z, fminf(p2.z, p3.z)));
    hx = fmaxf(fmaxf(hx, p0.x), fmaxf(p1.x, fmaxf(p2.x, p3.x))); hy = fmaxf(fmaxf(hy, p0.y), fmaxf(p1.y, fmaxf(p2.y, p3.y))); hz = fmaxf(fmaxf(hz, p0.z), fmaxf(p1.z, fmaxf(p2.z, p3.z)));
    mi = min(min(mi, __float_as_int(p0.w)), min(__float_as_int(p1.w), min(__float_as_int(p2.w), __float_as_int(p3.w))));
  }
  mi = wave_imin(mi);
  lx = wave_fmin(lx); ly = wave_fmin(ly); lz = wave_fmin(lz); hx = wave_fmax(hx); hy = wave_fmax(hy); hz = wave_fmax(hz);
  lo = make_float4(lx, ly, lz, 0.f); hi = make_float4(hx, hy, hz, 0.f);
}

// Per occupied cell: the box of its points, its first point (sample for the quick edge test of the cell graph), its
// smallest cloud index and the exact sums of its coordinates — ONE streaming pass over `sorted`, balanced whatever the
// cell sizes are (a thread group per cell — round 1/2 — ended with the cells of thousands of points): a wave takes 256
// consecutive positions, four per lane; a lane folds its four points serially, the open runs at lane boundaries go
// through a segmented scan over the lanes (19 words × 6 DPP moves per 256 points: acc_segmented_scan), and whoever holds the last point of a
// cell writes its record.  Cells that continue into another wave tile are merged with atomics (min / max / integer
// add: order-free), their records were initialised by k_gridhash.
__global__ __launch_bounds__(MOR_BT) void k_cellboxes(MorDev d) {
  int s, bx, gbx;
  if (!map_block_work(d, [&](int s_) { return ((int)d.info[s_].M + (MOR_BT / 64) * CB_WTILE - 1) / ((MOR_BT / 64) * CB_WTILE); }, s, bx, gbx)) return;   // work: steps of one workgroup over the stream's cell-ordered points
  const int M = d.info[s].M, lane = lane_id(), wv = wave_id();
  const size_t so = (size_t)s * d.Nmax;
  const float4 *sp = d.sorted + so; const int *sc = d.scell + so;
  // A wave takes a CONTIGUOUS range of the stream's wave tiles and carries the run that is open at the right edge of a tile over to its next tile (through 19 words of LDS): only
  // the cells that span the edge between two waves' ranges are merged with atomics — thirteen each; with the tiles dealt out round robin, two cells of every tile were:
  // 95 000 atomics per launch on the headline path, 240 000 – 455 000 on the street scenes, the million-point clouds and the voxel ground variant.
  __shared__ int l_carry[MOR_BT / 64][20];
  const int nwaves = gbx * (MOR_BT / 64), ntile = (M + CB_WTILE - 1) / CB_WTILE, per = (ntile + nwaves - 1) / nwaves;
  const int t_begin = (bx * (MOR_BT / 64) + wv) * per, t_end = min(ntile, t_begin + per);
  bool carry_shared = false;   // the carried run began in front of this wave's range: another wave holds its head
  for (int t = t_begin; t < t_end; ++t) {
    const int base = t * CB_WTILE; const bool first = t == t_begin, last = t == t_end - 1;
    const int j0 = base + 4 * lane;
    int c[4]; float4 p[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int j = min(j0 + u, M - 1); c[u] = sc[j]; p[u] = sp[j]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) if (j0 + u >= M) c[u] = -1;
    int left = -2, right = -3;   // cells of the positions just outside the tile
    if (lane == 0 && base > 0) left = sc[base - 1];
    if (lane == 63 && base + CB_WTILE < M) right = sc[base + CB_WTILE];
    const int cw0 = wave_bcast(c[0], 0), cwl = wave_bcast(c[3], 63);
    const bool open_l = wave_bcast(left, 0) == cw0, open_r = wave_bcast(right, 63) == cwl;
    const int prevc = wave_shift_up1(c[3], left), nextc = wave_shift_down1(c[0], right);   // (lane 0 / 63: the cells just outside the tile)
    // the lane's tail run (the run holding its last position) and whether it began in an earlier lane
    int ts = 3;
    if (c[2] == c[3]) { ts = 2; if (c[1] == c[3]) { ts = 1; if (c[0] == c[3]) ts = 0; } }
    CellAcc S; acc_clear(S);
#pragma unroll
    for (int u = 0; u < 4; ++u) if (u >= ts) acc_point(S, p[u]);
    const bool cont = open_l && !first;   // the tile's first run continues what this wave carried over from its previous tile
    auto carried = [&]() { CellAcc r; const int *lc = l_carry[wv];
      r.lx = __int_as_float(lc[0]); r.ly = __int_as_float(lc[1]); r.lz = __int_as_float(lc[2]); r.hx = __int_as_float(lc[3]); r.hy = __int_as_float(lc[4]); r.hz = __int_as_float(lc[5]); r.mi = lc[6];
#pragma unroll
      for (int k = 0; k < 3; ++k) { r.a[k] = ((long long)lc[8 + 4 * k] << 32) | (unsigned)lc[7 + 4 * k]; r.b[k] = ((long long)lc[10 + 4 * k] << 32) | (unsigned)lc[9 + 4 * k]; }
      return r; };
    if (cont) {
      wave_lds_fence();
      if (lane == 0 && ts == 0) acc_merge(S, carried());   // (lane 0 lies inside that run altogether: the lanes behind it see the carried part through the scan)
    }
    const bool head = !(ts == 0 && c[0] == prevc) || lane == 0;
    const unsigned long long heads = __ballot(head);
    const int hl = 63 - __clzll((long long)(heads & (lanemask_lt() | (1ull << lane))));
    acc_segmented_scan(S, hl, lane);
    CellAcc acc = acc_dpp<0x138>(S);   // wave_shr:1 — the run reaching this lane from the left, up to the previous lane
    if (lane == 0 || c[0] != prevc) acc_clear(acc);
    if (cont && lane == 0) acc = carried();   // (read again rather than kept: nineteen registers across the scan were a wave per SIMD)
    const bool keep = open_r && !last;   // the run at the right edge goes on in this wave's next tile
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (c[u] < 0) break;
      if (c[u] != (u ? c[u - 1] : prevc)) d.crep[so + c[u]] = p[u];   // first position of the cell
      acc_point(acc, p[u]);
      const int nxt = u < 3 ? c[u + 1] : nextc;
      if (c[u] != nxt || (u == 3 && lane == 63)) {
        if (u == 3 && lane == 63 && keep) {
          int *lc = l_carry[wv];
          lc[0] = __float_as_int(acc.lx); lc[1] = __float_as_int(acc.ly); lc[2] = __float_as_int(acc.lz); lc[3] = __float_as_int(acc.hx); lc[4] = __float_as_int(acc.hy); lc[5] = __float_as_int(acc.hz); lc[6] = acc.mi;
#pragma unroll
          for (int k = 0; k < 3; ++k) { lc[7 + 4 * k] = (int)(unsigned)acc.a[k]; lc[8 + 4 * k] = (int)(acc.a[k] >> 32); lc[9 + 4 * k] = (int)(unsigned)acc.b[k]; lc[10 + 4 * k] = (int)(acc.b[k] >> 32); }
        } else
          acc_emit(d, so, c[u], acc, (c[u] == cw0 && open_l && (first || carry_shared)) || (c[u] == cwl && open_r));   // (pieces of a cell that lies across two waves' ranges: merged with atomics)
        acc_clear(acc);
      }
    }
    carry_shared = keep && cw0 == cwl && open_l && (first || carry_shared);
    if (keep) wave_lds_fence();
  }
}
// ------------------------------------------------------------------------------------ the cell graph over y-SLABS
// (One 1024-thread workgroup per stream — round 1 — kept 64 of the 256 CUs busy for 250–330 µs and ended with its slowest stream.)
// Cell keys are y-major, so a contiguous range of compact ids is the slab of space between two y planes: every stream's
// cells are cut into P slabs of about equal cell count (k_gridhash / k_cellboxes: slab_bounds), and one small workgroup
// per (stream, slab) runs both hook passes for the cells it OWNS over the forward half of the neighbourhood
// (dy ≥ 0), i.e. against its own cells and the cells of the next two y-slices (its look-ahead, owned by the next slab).
// Its union-find forest lives in its own LDS and covers own + look-ahead cells only; what it publishes is, per cell of
// that range, the LOCAL root.  A forest is equivalent to the edge set {(c, root(c))}, so k_cg_final re-unites
// (c, root_own(c)) and (c, root_lookahead-of-the-previous-slab(c)) in one forest per stream and gets exactly the
// components of the full edge set — every edge was found by the slab owning its lower-y cell.
#ifndef CGS_T
#define CGS_T 512
#endif
#ifndef CGS_CAP
#define CGS_CAP 1360      // local cells (own + look-ahead) held in LDS with everything the pair decisions need: nine words each (CGS_CW)
#endif
#ifndef CGS_ROWCAP
#define CGS_ROWCAP 2048   // local (y,z) rows held in LDS
#endif
#ifndef CGS_LISTW
#define CGS_LISTW 2048    // LDS words of the candidate-pair lists
#endif
#define CGS_NW (CGS_T / 64)
#define CGS_WLIST (CGS_LISTW / CGS_NW)        // LDS list entries per wave (one packed pair each)
#define CGS_WOVF (MOR_CGS_OVF / CGS_NW)       // global overflow entries per wave
#define CGS_QW 320                            // a wave's queue of neighbour pairs found by one batch of 64 (cell, row) items (≤ 5 each)
static_assert(CGS_CAP <= 16384, "pair lists pack two local cell ids into 28 bits");
// Candidate pairs are appended by the wave that finds them to ITS OWN list (LDS part + global overflow part): the
// position comes from a wave-uniform counter in a register, so enumeration needs no atomic and no round trip per append.
// (LDS mode: local ids < 16384, a pair is one word a << 14 | b; global mode: two words per pair, half the capacity.)
template <bool LDS> __device__ __forceinline__ int cgs_wlist_cap() { return (CGS_WLIST + CGS_WOVF) / (LDS ? 1 : 2); }
template <bool LDS> __device__ __forceinline__ void cgs_wlist_put(int *ovf, int *l_list, int w, int slot, int a, int b) {
  if (LDS) {
    if (slot < CGS_WLIST) l_list[w * CGS_WLIST + slot] = (a << 14) | b;
    else cg_st<false>(ovf + (size_t)w * CGS_WOVF + (slot - CGS_WLIST), (a << 14) | b);
  } else {
    if (2 * slot + 1 < CGS_WLIST) { l_list[w * CGS_WLIST + 2 * slot] = a; l_list[w * CGS_WLIST + 2 * slot + 1] = b; }
    else { int *o = ovf + (size_t)w * CGS_WOVF + (2 * slot - CGS_WLIST / 2 * 2); cg_st<false>(o, a); cg_st<false>(o + 1, b); }
  }
}
template <bool LDS> __device__ __forceinline__ void cgs_wlist_get(const int *ovf, const int *l_list, int w, int slot, int &a, int &b) {
  if (LDS) {
    const int c = slot < CGS_WLIST ? l_list[w * CGS_WLIST + slot] : cg_ld<false>(ovf + (size_t)w * CGS_WOVF + (slot - CGS_WLIST));
    a = c >> 14; b = c & 16383;
  } else {
    if (2 * slot + 1 < CGS_WLIST) { a = l_list[w * CGS_WLIST + 2 * slot]; b = l_list[w * CGS_WLIST + 2 * slot + 1]; }
    else { const int *o = ovf + (size_t)w * CGS_WOVF + (2 * slot - CGS_WLIST / 2 * 2); a = cg_ld<false>(o); b = cg_ld<false>(o + 1); }
  }
}
// second list (pairs for a whole wave), global only
template <bool LDS> __device__ __forceinline__ int cgs_list2_cap() { return MOR_CGS_OVF / (LDS ? 1 : 2); }
template <bool LDS> __device__ __forceinline__ void cgs_list2_put(int *ovf, int slot, int a, int b) {
  if (LDS) cg_st<false>(ovf + MOR_CGS_OVF + slot, (a << 14) | b); else { cg_st<false>(ovf + MOR_CGS_OVF + 2 * slot, a); cg_st<false>(ovf + MOR_CGS_OVF + 2 * slot + 1, b); }
}
template <bool LDS> __device__ __forceinline__ void cgs_list2_get(const int *ovf, int slot, int &a, int &b) {
  if (LDS) { const int c = cg_ld<false>(ovf + MOR_CGS_OVF + slot); a = c >> 14; b = c & 16383; } else { a = cg_ld<false>(ovf + MOR_CGS_OVF + 2 * slot); b = cg_ld<false>(ovf + MOR_CGS_OVF + 2 * slot + 1); }
}
// Point test of one pair by one thread, as few levels of dependent loads as possible: up to 8 points of the smaller
// cell in registers, the other cell streamed eight independent loads at a time (at most 24 of its points).
// Returns 1 = edge, 0 = no edge (every pair was looked at), −1 = undecided (a sample only: the pair goes to a wave).
__device__ __forceinline__ int pair_points_thread(const float4 *sp, int a0, int na, int b0, int nb, float r2) {
  if (na > nb) { int t = a0; a0 = b0; b0 = t; t = na; na = nb; nb = t; }
  float ax[8], ay[8], az[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { const float4 p = sp[a0 + min(i, na - 1)]; ax[i] = p.x; ay[i] = p.y; az[i] = p.z; }   // clamped duplicates repeat real points
  const int lim = min(nb, 24);
  for (int j0 = 0; j0 < lim; j0 += 8) {
    float4 q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) q[j] = sp[b0 + min(j0 + j, nb - 1)];
    bool hit = false;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int i = 0; i < 8; ++i) hit |= sqdist(ax[i], ay[i], az[i], q[j].x, q[j].y, q[j].z) < r2;
    if (hit) return 1;
  }
  return (na <= 8 && nb <= 24) ? 0 : -1;
}
// The hook pass of a slab.  Local ids: own cells [0, n_own), look-ahead [n_own, n_loc).  key[] = keys of the local cells,
// rows[] = row table of the slab's rows (row r0 first) holding LOCAL ids plus `rsub` (0 for the LDS copy, the slab's first
// compact id when the global table is read in place); soc = first slot of the slab's cells in the per-cell global arrays.
// LDS mode also holds, per local cell: pc[] (its coordinates, packed), one sample point (rx, ry, rz) and its point box
// (bx[0..5]: low corner, high corner) — so most pairs are decided without a single global load.  Every wave works
// on its own, in steps of 64 (own cell, neighbour row) items over the forward half of the 5×5×5 neighbourhood (dy ≥ 0;
// 13 rows, the five rows of the 3×3×3 block first):
//  A1  one lane per item: the ≤ 5 cells of the row's window come as one batch of independent LDS loads; those whose
//      parent differs from the cell's root go to the wave's queue.
//  A2  one lane per queued pair, all lanes busy: roots; an edge when the two SAMPLE points lie within r (most
//      neighbouring cells of one surface) or when the farthest corners of the two point boxes do; no edge when the
//      boxes are ≥ r apart; only what is left goes to the wave's candidate list.
// then, for the whole workgroup:
//  B1  one thread per listed pair: roots re-checked, then the points (pair_points_thread).
//  B2  one wave per pair the thread test could not finish (big cells): pruned exhaustive test.
template <bool LDS> struct CgsCells { const int *key, *pc; const float *rx, *ry, *rz; const unsigned *bx; int cap; };   // bx: three planes of `cap` words, (low, high) of one axis as two halves each (null: samples and boxes stay in global memory)
// a box edge [lo, hi] as two halves that contain it: lo rounded down, hi rounded up
__device__ __forceinline__ unsigned box_halves(float lo, float hi) { return (unsigned)__half_as_ushort(__float2half_rd(lo)) | ((unsigned)__half_as_ushort(__float2half_ru(hi)) << 16); }
__device__ __forceinline__ float box_lo(unsigned w) { return __half2float(__ushort_as_half((unsigned short)(w & 0xffffu))); }
__device__ __forceinline__ float box_hi(unsigned w) { return __half2float(__ushort_as_half((unsigned short)(w >> 16))); }
template <bool LDS, bool BOXL, typename RT> __device__ __forceinline__ void cgs_hooks(const MorDev &d, const MorGrid &G, size_t soc, int n_own, int n_loc, const CgsCells<LDS> &L, const int *start, const RT *rows, int rsub, int r0, int nlrows,
                                                              int *par, const float4 *sp, int *ovf, int *l_list, int *l_queue, int *l_wcnt, int *l_n2) {
  const float r2 = d.r2;
  const int *key = L.key;
  constexpr int NR = 13;   // rows of the forward half: (dy,dz) = (0,0) (0,1) (1,−1) (1,0) (1,1) — the 3×3×3 block — then (0,2), (1,±2), (2,−2…2)
  const int w = wave_id(), lane = lane_id();
  int *queue = l_queue + w * CGS_QW;
  int wcount = 0;
#define CGS_TICK(v)
  for (int it0 = w * 64; it0 < n_own * NR; it0 += CGS_T) {
    // ---- A1
    CGS_TICK(k0);
    const int it = it0 + lane;
    int a = 0, rowbase = 0, b = 0, hi = 0, ra = -1; bool same_row = false;
    if (it < n_own * NR) {
      const int ri = it / n_own; a = it - ri * n_own;   // row-major over the rows: all cells' near rows come first
      int dy, dz;
      if (ri < 5) { dy = ri >= 2; dz = ri < 2 ? ri : ri - 3; } else if (ri == 5) { dy = 0; dz = 2; } else if (ri < 8) { dy = 1; dz = ri == 6 ? -2 : 2; } else { dy = 2; dz = ri - 10; }
      same_row = dy == 0 && dz == 0;
      int x, y, z;
      if (LDS) { const unsigned q = (unsigned)L.pc[a]; x = (int)(q & 2047u); z = (int)((q >> 11) & 1023u); y = (int)(q >> 21); }
      else { const int ka = key[a], rowa = ka / G.nx; x = ka - rowa * G.nx; z = rowa % G.nz; y = rowa / G.nz; }
      if (y + dy < G.ny && (unsigned)(z + dz) < (unsigned)G.nz) {
        const int rr = grid_row(G, y + dy, z + dz), rl = rr - r0;
        if (rl >= 0 && rl < nlrows) {
          const int rlo = (int)rows[rl] - rsub, rn = (int)rows[rl + 1] - rsub - rlo;
          rowbase = rr * G.nx + x; b = rlo; hi = rlo + rn;
          if (rn > 5) b = cg_lower_bound8(key, rlo, rn, rowbase - 2);
          if (b < hi) ra = cg_find<LDS>(par, a);
        }
      }
    }
    CGS_TICK(k1);
    int kb[5], pb[5];   // the window holds at most five cells (x−2 … x+2): keys and parents as one batch of independent loads
#pragma unroll
    for (int u = 0; u < 5; ++u) { const int bi = min(b + u, max(hi - 1, 0)); kb[u] = key[bi]; pb[u] = cg_ld<LDS>(par + bi); }
    int qn = 0;
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int bb = b + u, dx = kb[u] - rowbase;
      const bool want = bb < hi && dx >= -2 && dx <= 2 && !(same_row && dx <= 0) && pb[u] != ra;
      const unsigned long long m = __ballot(want);
      if (want) queue[qn + __popcll(m & lanemask_lt())] = (lane << 26) | bb;   // (enumerating lane, neighbour): the cell is that lane's `a` (local ids < 2²⁶: mor_batch_create bounds max_points)
      qn += __popcll(m);
    }
    CGS_TICK(k2);
    // ---- A2
    for (int q0 = 0; q0 < qn; q0 += 64) {
      const bool act = q0 + lane < qn;
      const int qc = act ? queue[q0 + lane] : 0;
      const int qa = __shfl(a, (qc >> 26) & 63, 64), qb = qc & ((1 << 26) - 1);
      bool want = act && cg_find<LDS>(par, qa) != cg_find<LDS>(par, qb);
      if (want) {
        float pax, pay, paz, alx, aly, alz, ahx, ahy, ahz, qx, qy, qz, blx, bly, blz, bhx, bhy, bhz;
        if (BOXL) {
          const unsigned ax_ = L.bx[qa], ay_ = L.bx[L.cap + qa], az_ = L.bx[2 * L.cap + qa], bx_ = L.bx[qb], by_ = L.bx[L.cap + qb], bz_ = L.bx[2 * L.cap + qb];
          pax = L.rx[qa]; pay = L.ry[qa]; paz = L.rz[qa]; alx = box_lo(ax_); aly = box_lo(ay_); alz = box_lo(az_); ahx = box_hi(ax_); ahy = box_hi(ay_); ahz = box_hi(az_);
          qx = L.rx[qb]; qy = L.ry[qb]; qz = L.rz[qb]; blx = box_lo(bx_); bly = box_lo(by_); blz = box_lo(bz_); bhx = box_hi(bx_); bhy = box_hi(by_); bhz = box_hi(bz_);
        } else {
          const float4 q = d.crep[soc + qa], lo = d.cmeta[2 * (soc + qa)], h4 = d.cmeta[2 * (soc + qa) + 1]; pax = q.x; pay = q.y; paz = q.z; alx = lo.x; aly = lo.y; alz = lo.z; ahx = h4.x; ahy = h4.y; ahz = h4.z;
          const float4 q2 = d.crep[soc + qb], lo2 = d.cmeta[2 * (soc + qb)], h42 = d.cmeta[2 * (soc + qb) + 1]; qx = q2.x; qy = q2.y; qz = q2.z; blx = lo2.x; bly = lo2.y; blz = lo2.z; bhx = h42.x; bhy = h42.y; bhz = h42.z;
        }
        bool edge = sqdist(pax, pay, paz, qx, qy, qz) < r2;   // the two sample points are within r
        if (!edge) {
          const float gx = fmaxf(fmaxf(blx - ahx, alx - bhx), 0.f), gy = fmaxf(fmaxf(bly - ahy, aly - bhy), 0.f), gz = fmaxf(fmaxf(blz - ahz, alz - bhz), 0.f);
          if ((gx * gx + gy * gy + gz * gz) * 0.999f >= r2) want = false;   // boxes ≥ r apart: no edge
          else { const float sx = fmaxf(bhx - alx, ahx - blx), sy = fmaxf(bhy - aly, ahy - bly), sz = fmaxf(bhz - alz, ahz - blz); edge = (sx * sx + sy * sy + sz * sz) * 1.001f < r2; }   // farthest corners within r: every pair is an edge
        }
        if (edge) { cg_unite<LDS>(par, qa, qb); want = false; }
      }
      const unsigned long long m = __ballot(want);
      if (m) {
        if (want) {
          const int slot = wcount + __popcll(m & lanemask_lt());
          if (slot < cgs_wlist_cap<LDS>()) cgs_wlist_put<LDS>(ovf, l_list, w, slot, qa, qb);
          else { const int a0 = start[qa], b0 = start[qb]; if (pair_hit_serial(sp, a0, start[qa + 1] - a0, b0, start[qb + 1] - b0, r2)) cg_unite<LDS>(par, qa, qb); }   // lists full (never seen): settle it here
        }
        wcount += __popcll(m);
      }
    }
  }
  if (lane == 0) l_wcnt[w] = min(wcount, cgs_wlist_cap<LDS>());
  __threadfence_block();
  __syncthreads();
  // ---- B1: one thread per candidate pair (the waves' lists, back to back)
  int pre[CGS_NW + 1]; pre[0] = 0;
#pragma unroll
  for (int u = 0; u < CGS_NW; ++u) pre[u + 1] = pre[u] + l_wcnt[u];
  const int n1 = pre[CGS_NW];
  for (int h = threadIdx.x; h < n1; h += CGS_T) {
    int lw = 0;
#pragma unroll
    for (int u = 1; u < CGS_NW; ++u) lw += h >= pre[u];
    int a, b; cgs_wlist_get<LDS>(ovf, l_list, lw, h - pre[lw], a, b);
    if (cg_find<LDS>(par, a) == cg_find<LDS>(par, b)) continue;
    const int a0 = start[a], a1 = start[a + 1], b0 = start[b], b1 = start[b + 1];
    const int verdict = pair_points_thread(sp, a0, a1 - a0, b0, b1 - b0, r2);
    if (verdict > 0) cg_unite<LDS>(par, a, b);
    else if (verdict < 0) {
      const int slot = atomicAdd(l_n2, 1);
      if (slot < cgs_list2_cap<LDS>()) cgs_list2_put<LDS>(ovf, slot, a, b);
      else if (pair_hit_serial(sp, a0, a1 - a0, b0, b1 - b0, r2)) cg_unite<LDS>(par, a, b);
    }
  }
  __threadfence_block();
  __syncthreads();
  // ---- B2: one wave per pair left over
  const int n2 = min(*l_n2, cgs_list2_cap<LDS>());
  for (int h = w; h < n2; h += CGS_NW) {
    int a, b; cgs_list2_get<LDS>(ovf, h, a, b);
    if (cg_find<LDS>(par, a) == cg_find<LDS>(par, b)) continue;
    const float4 alo = d.cmeta[2 * (soc + a)], ahi = d.cmeta[2 * (soc + a) + 1], blo = d.cmeta[2 * (soc + b)], bhi = d.cmeta[2 * (soc + b) + 1];
    if (pair_hit_wave(sp, start[a], start[a + 1] - start[a], start[b], start[b + 1] - start[b], r2, lane, alo, ahi, blo, bhi) && lane == 0) cg_unite<LDS>(par, a, b);
  }
  __syncthreads();
}
template <bool LDS, bool BOXL, typename RT> __device__ __forceinline__ void cgs_body(const MorDev &d, const MorGrid &G, int s, size_t so, int c0, int n_own, int n_loc, const CgsCells<LDS> &L, const RT *rows, int rsub, int r0, int nlrows, int *par, int *ovf, int *l_list, int *l_queue, int *l_wcnt, int *l_n2) {
  const int *start = d.cstart + (size_t)s * (d.Nmax + 1) + c0;   // start[local id]: first position of the cell in `sorted`
  const float4 *sp = d.sorted + so;
  cgs_hooks<LDS, BOXL, RT>(d, G, so + c0, n_own, n_loc, L, start, rows, rsub, r0, nlrows, par, sp, ovf, l_list, l_queue, l_wcnt, l_n2);
  // local roots as global compact ids: own cells → lroot_a, look-ahead cells → lroot_b
  for (int c = threadIdx.x; c < n_loc; c += CGS_T) {
    const int r = c0 + cg_find<LDS>(par, c);
    if (c < n_own) st_agent(&d.lroot_a[so + c0 + c], r); else st_agent(&d.lroot_b[so + c0 + c], r);   // (agent scope: the merge may run in another slab's workgroup of this launch, stream_last_block)
  }
   // (HW_ID: wave, SIMD, CU, SE … of the recording wave)
}
template <bool LDS, int NT> __device__ __forceinline__ void cgf_body(const MorDev &d, int s, int nocc, int *par, int *scr, int *l_misc, const int *l_sc, const int *l_se);
// CAP: local cells (own + look-ahead) the workgroup holds in LDS (76 KB: two workgroups per CU).
// With d.cg_fused the stream's LAST slab workgroup to finish (stream_last_block) goes on with the merge of the slab forests and everything
// k_cg_final does, in the same LDS (three arrays of CGS_FCAP cells; streams with more cells: global-memory arrays) — one launch and one
// queueing delay less per frame; the host falls back to the separate k_cg_final launch (147 KB of LDS: 12 288 cells) when the previous
// frame's cell counts say a stream would not fit.
#define CGS_CW 9   // LDS words per local cell: key, parent, packed coordinates, sample point (3), point box as six halves (3)
#define CGS_SLAB_WORDS (CGS_CW * CGS_CAP + CGS_ROWCAP + 1 + CGS_LISTW + CGS_NW * CGS_QW)
#define CGS_ARENA (CGS_SLAB_WORDS > 2 * MOR_CGS_FCAP ? CGS_SLAB_WORDS : 2 * MOR_CGS_FCAP)   // (the slab layout of the default build is 18 945 words)
#define CGS_FCAP (CGS_ARENA / 2)
#define CGF_KL 512   // kept clusters per stream whose records the fast tail holds in LDS
#ifndef CGF_RC
#define CGF_RC 14   // cells per thread the fast tail keeps in registers (cgf_fast): streams of up to 14 · 512 = 7 168 cells in k_cg_slab, 14 336 in k_cg_final
#endif
template <int NT, int RC> __device__ __forceinline__ bool cgf_fast(const MorDev &d, int s, int nocc, int *par, int *scr, int *kl, int *l_misc, const int *l_sc, const int *l_se);
template <int CAP> __device__ __forceinline__ void cg_slab_body(const MorDev &d, int s, int j, int *l_arena, int *l_wcnt, int *l_n2p);
template <int CAP> __global__ __launch_bounds__(CGS_T) void k_cg_slab(MorDev d) {   // (keep it at ≤ 128 VGPRs — two workgroups per CU; a loop over several slabs per workgroup took 157: 228 → 350 µs in the pipeline)
  int s, j, Ps;
  if (!map_block_work<true>(d, [&](int s_) { return d.slab_p[s_]; }, s, j, Ps)) return;   // one workgroup per slab; the stream's number of slabs was fixed by slab_bounds within the launch's budget
  static_assert(CAP == CGS_CAP, "the LDS arena is laid out for CGS_CAP");
  __shared__ int l_arena[CGS_ARENA], l_wcnt[CGS_NW], l_n2, l_last;
  cg_slab_body<CAP>(d, s, j, l_arena, l_wcnt, &l_n2);
  if (!d.cg_fused) return;
  if (!stream_last_block(d.tickets + (size_t)s * TK_COUNT + TK_CGFINAL, Ps, &l_last)) return;
  const int nocc = d.info[s].n_occ;
  const size_t so = (size_t)s * d.Nmax;
  __shared__ int l_misc[1 + 3 * (CGS_T / 64)], l_sc[MOR_MAXP + 1], l_se[MOR_MAXP + 1];
  if (threadIdx.x <= Ps) { l_sc[threadIdx.x] = d.slab_c[(size_t)s * (MOR_MAXP + 1) + threadIdx.x]; l_se[threadIdx.x] = d.slab_e[(size_t)s * (MOR_MAXP + 1) + threadIdx.x]; }
  if (nocc <= CGS_FCAP && !d.cg_force_global) {
    __syncthreads();
    if (!d.cg_slow_tail && 2 * nocc + 7 * CGF_KL <= CGS_ARENA && cgf_fast<CGS_T, CGF_RC>(d, s, nocc, l_arena, l_arena + nocc, l_arena + 2 * nocc, l_misc, l_sc, l_se)) return;
    __syncthreads();
    int *l_par = l_arena, *l_a = l_arena + CGS_FCAP;
    for (int i = threadIdx.x; i < nocc; i += CGS_T) l_par[i] = i;
    __syncthreads();
    cgf_body<true, CGS_T>(d, s, nocc, l_par, l_a, l_misc, l_sc, l_se);
  } else {
    int *par = d.parent + so;
    for (int i = threadIdx.x; i < nocc; i += CGS_T) cg_st<false>(par + i, i);
    __threadfence_block();   // (the forest is this workgroup's alone and accessed with agent-scope operations; a device-wide fence writes the XCD's whole L2 back: 32 µs)
    __syncthreads();
    cgf_body<false, CGS_T>(d, s, nocc, par, d.csize + so, l_misc, l_sc, l_se);
  }
}
template <int CAP> __device__ __forceinline__ void cg_slab_body(const MorDev &d, int s, int j, int *l_arena, int *l_wcnt, int *l_n2p) {
  const MorGrid G = stream_grid(d, s);   // the stream's clustering grid (voxel ground variant: its own number of z layers)
  const size_t so = (size_t)s * d.Nmax;
  const int *sy = d.slab_y + (size_t)s * (MOR_MAXP + 1), *sc = d.slab_c + (size_t)s * (MOR_MAXP + 1), *se = d.slab_e + (size_t)s * (MOR_MAXP + 1);
  const int c0 = sc[j], c1 = sc[j + 1], c2 = se[j], n_own = c1 - c0, n_loc = c2 - c0;
  if (n_own <= 0) return;
  const int y0 = sy[j], y2 = min(sy[j + 1] + 2, G.ny), r0 = y0 * G.nz, nlrows = (y2 - y0) * G.nz;
  // CGS_CW·CAP words of cell data: CAP cells with everything in LDS (key, parent, packed coordinates, sample point, box), or —
  // slabs of up to 3·CAP cells, e.g. a façade across a y-slice — key, parent and packed coordinates only: the enumeration
  // (A1) and the forest stay in LDS, the decisions about queued pairs (A2) fetch samples and boxes from global memory
  // The point box of a cell is held as six HALVES, low corner rounded down and high corner rounded up (round 5: 9 instead of 12 words per cell — 1 360 instead of 1 024 cells with
  // everything in LDS, 4 080 with key / parent / coordinates only): the box only FILTERS pairs — "boxes ≥ r apart: no edge", "farthest corners within r: edge" — and a box that is
  // a few centimetres too big leaves both tests valid and a few more pairs to the point tests.  With the host's slab policy scaled to it (8 instead of 10 slabs per stream on the open
  // scenes — 512 workgroups, ONE round on the GPU's 512 slots —, 9 instead of 12 on the street scenes): hdl64_b64 +0.8 %, hdl64_urban_b64 +1.0 % (interleaved).
  int *l_cells = l_arena, *l_list = l_cells + CGS_CW * CAP + CGS_ROWCAP + 1, *l_queue = l_list + CGS_LISTW;
  unsigned short *l_rows = reinterpret_cast<unsigned short *>(l_cells + CGS_CW * CAP);   // local row table as 16-bit offsets (≤ 3·CAP local cells): 2·CGS_ROWCAP rows in CGS_ROWCAP + 1 words
  int &l_n2 = *l_n2p;
  int *ovf = d.cg_ovf + (size_t)(s * MOR_MAXP + j) * MOR_CGS_OVF * 2;   // [0, MOR_CGS_OVF): the waves' candidate lists, [MOR_CGS_OVF, 2·MOR_CGS_OVF): pairs for whole waves
  const int *g_rows = d.row_start + (size_t)s * (d.g.nrows + 1) + r0;
  if (threadIdx.x == 0) l_n2 = 0;
  const bool fits_rows = nlrows <= 2 * CGS_ROWCAP && !d.cg_force_global;   // (thick slabs of the sparse far ends of a cloud: 150 slices × 14 layers seen at 120 000 points; beyond the table they ran the global-memory path, 3× slower, and set the kernel's span)
  if (fits_rows && n_loc <= CAP) {
    int *l_key = l_cells, *l_par = l_cells + CAP, *l_pc = l_cells + 2 * CAP;
    float *l_rx = reinterpret_cast<float *>(l_cells + 3 * CAP), *l_ry = l_rx + CAP, *l_rz = l_rx + 2 * CAP; unsigned *l_bx = reinterpret_cast<unsigned *>(l_cells + 6 * CAP);
    const int *gk = d.ckey + so + c0; const float4 *grep = d.crep + so + c0, *gm = d.cmeta + 2 * (so + c0);
    for (int i = threadIdx.x; i < n_loc; i += CGS_T) {
      const int k = gk[i], row = k / G.nx;
      l_key[i] = k; l_par[i] = i; l_pc[i] = (int)((unsigned)(k - row * G.nx) | ((unsigned)(row % G.nz) << 11) | ((unsigned)(row / G.nz) << 21));
      const float4 q = grep[i], lo = gm[2 * i], hi4 = gm[2 * i + 1];
      l_rx[i] = q.x; l_ry[i] = q.y; l_rz[i] = q.z;
      l_bx[i] = box_halves(lo.x, hi4.x); l_bx[CAP + i] = box_halves(lo.y, hi4.y); l_bx[2 * CAP + i] = box_halves(lo.z, hi4.z);
    }
    for (int i = threadIdx.x; i <= nlrows; i += CGS_T) l_rows[i] = (unsigned short)(g_rows[i] - c0);
    __syncthreads();
    const CgsCells<true> L = {l_key, l_pc, l_rx, l_ry, l_rz, l_bx, CAP};
    cgs_body<true, true, unsigned short>(d, G, s, so, c0, n_own, n_loc, L, l_rows, 0, r0, nlrows, l_par, ovf, l_list, l_queue, l_wcnt, &l_n2);
  } else if (fits_rows && n_loc <= 3 * CAP) {
    int *l_key = l_cells, *l_par = l_cells + 3 * CAP, *l_pc = l_cells + 6 * CAP;
    const int *gk = d.ckey + so + c0;
    for (int i = threadIdx.x; i < n_loc; i += CGS_T) {
      const int k = gk[i], row = k / G.nx;
      l_key[i] = k; l_par[i] = i; l_pc[i] = (int)((unsigned)(k - row * G.nx) | ((unsigned)(row % G.nz) << 11) | ((unsigned)(row / G.nz) << 21));
    }
    for (int i = threadIdx.x; i <= nlrows; i += CGS_T) l_rows[i] = (unsigned short)(g_rows[i] - c0);
    __syncthreads();
    const CgsCells<true> L = {l_key, l_pc, nullptr, nullptr, nullptr, nullptr, 0};
    cgs_body<true, false, unsigned short>(d, G, s, so, c0, n_own, n_loc, L, l_rows, 0, r0, nlrows, l_par, ovf, l_list, l_queue, l_wcnt, &l_n2);
  } else {   // slab too big for LDS: the same code on global arrays (even and odd slabs use different forests: look-aheads overlap the next slab)
    int *par = ((j & 1) ? d.parent2 : d.parent) + so + c0;
    for (int i = threadIdx.x; i < n_loc; i += CGS_T) cg_st<false>(par + i, i);
    __threadfence_block();   // (the forest is this workgroup's alone and accessed with agent-scope operations; a device-wide fence writes the XCD's whole L2 back: 32 µs)
    __syncthreads();
    const CgsCells<false> L = {d.ckey + so + c0, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    cgs_body<false, false, int>(d, G, s, so, c0, n_own, n_loc, L, g_rows, c0, r0, nlrows, par, ovf, l_list, l_queue, l_wcnt, &l_n2);
  }
}
// One workgroup per stream: merges the slab forests, then components (size, smallest cloud index), the kept clusters
// (:215-216), their order, per-cell cluster ids, offsets — the tail of the former one-workgroup kernel.
#ifndef CGF_T
#define CGF_T 1024
#endif
#ifndef CGF_CAP
#define CGF_CAP 18432
#endif
// Two arrays of nocc ints besides the forest: `par` and ONE scratch array that is, in turn, the components' sizes, their smallest cloud indices, the
// roots' cluster ids and two per-cluster cursors (round 3 kept sizes and minima side by side: three arrays, 6 300 cells in the slab workgroup's LDS; two
// arrays hold 9 400, so the street scenes and the 262 144-point clouds merge inside k_cg_slab too).  kscr: 2·K ints of scratch when 2·K > nocc (never in practice).
template <bool LDS, int NT> __device__ __forceinline__ void cgf_body(const MorDev &d, int s, int nocc, int *par, int *scr, int *l_misc, const int *l_sc, const int *l_se) {
  int *size = scr, *mn = scr, *cidr = scr;
  const size_t so = (size_t)s * d.Nmax, ko = (size_t)s * d.Kcap;
  const int *start = d.cstart + (size_t)s * (d.Nmax + 1);
  const int lane = lane_id(), P = d.slab_p[s];
  // ---- merge: (c, local root in its own slab) and, for the look-ahead cells of the previous slab, (c, local root there)
  for (int c = threadIdx.x; c < nocc; c += NT) {
    cg_unite<LDS>(par, c, ld_agent(&d.lroot_a[so + c]));
    int j = 0;
    for (int k = 1; k < P; ++k) j += l_sc[k] <= c;            // slab owning c
    if (j > 0 && c < l_se[j - 1] && l_sc[j] > l_sc[j - 1]) cg_unite<LDS>(par, c, ld_agent(&d.lroot_b[so + c]));   // (an empty slab publishes nothing)
  }
  __threadfence_block();
  __syncthreads();
  for (int c = threadIdx.x; c < nocc; c += NT) { const int r = cg_find<LDS>(par, c); if (r != c) cg_st<LDS>(par + c, r); }
  __syncthreads();
  // ---- components: size (points) at the root
  for (int c = threadIdx.x; c < nocc; c += NT) cg_st<LDS>(size + c, 0);
  __threadfence_block();
  __syncthreads();
  for (int c = threadIdx.x; c < nocc; c += NT) { const int r = cg_ld<LDS>(par + c); atomicAdd(&size[r], start[c + 1] - start[c]); }
  __threadfence_block();
  __syncthreads();
  // ---- kept components (:215-216) → scratch list; K
  if (threadIdx.x == 0) l_misc[0] = 0;
  __syncthreads();
  for (int c = threadIdx.x; c < nocc; c += NT) {
    const bool root = cg_ld<LDS>(par + c) == c;
    const long long n = root ? (long long)cg_ld<LDS>(size + c) : 0;
    if (root && n >= d.min_cs && n <= d.max_cs) {
      const int k = atomicAdd(&l_misc[0], 1);
      if (k < d.Kcap) { d.kcell[ko + k] = c; d.ksize[ko + k] = (int)n; }
    }
  }
  __threadfence_block();
  __syncthreads();
  int K = l_misc[0];
  if (K > d.Kcap) { if (threadIdx.x == 0) mor_raise(d, s, 1u); K = d.Kcap; }
  __syncthreads();
  // ---- smallest cloud index of every component at its root (the scratch array again: the sizes of the kept ones are in ksize)
  for (int c = threadIdx.x; c < nocc; c += NT) cg_st<LDS>(mn + c, 0x7fffffff);
  __threadfence_block();
  __syncthreads();
  for (int c = threadIdx.x; c < nocc; c += NT) atomicMin(&mn[cg_ld<LDS>(par + c)], d.cmin[so + c]);
  __threadfence_block();
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += NT) d.kroot[ko + k] = cg_ld<LDS>(mn + d.kcell[ko + k]);
  __threadfence_block();
  __syncthreads();
  // ---- cluster order: size descending, ties by smaller first cloud index; rank by counting (the scratch array becomes the roots' cluster ids)
  for (int c = threadIdx.x; c < nocc; c += NT) cg_st<LDS>(cidr + c, -1);
  __threadfence_block();
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += NT) {
    const int my_sz = d.ksize[ko + k], my_rt = d.kroot[ko + k]; int rank = 0;
    for (int u = 0; u < K; ++u) { const int sz = d.ksize[ko + u], rt = d.kroot[ko + u]; rank += (sz > my_sz) || (sz == my_sz && rt < my_rt); }
    cg_st<LDS>(cidr + d.kcell[ko + k], rank);
    d.csz[ko + rank] = my_sz; d.krank_inv[ko + rank] = k;
  }
  __threadfence_block();
  __syncthreads();
  // ---- per-cell cluster id (a cell is a clique ⇒ one cluster); kept in place of the parent from here on
  for (int c = threadIdx.x; c < nocc; c += NT) { const int r = cg_ld<LDS>(par + c), id = cg_ld<LDS>(cidr + r); d.ccid[so + c] = id; reinterpret_cast<int *>(&d.cmeta[2 * (so + c)])[3] = id; cg_st<LDS>(par + c, id); }
  __threadfence_block();
  __syncthreads();
  // ---- cluster offsets (exclusive scan of sizes in cluster order), C, clear detection_results (:250-254)
  int *off = d.cl_off[d.cur] + (size_t)s * (d.Kcap + 1);
  int carry = 0;
  for (int b = 0; b < K; b += NT) {
    const int k = b + threadIdx.x, v = k < K ? d.csz[ko + k] : 0;
    const int inc = wave_incl_scan(v);
    if (lane == 63) l_misc[1 + wave_id()] = inc;
    __syncthreads();
    int basew = 0, tot = 0;
    for (int w = 0; w < NT / 64; ++w) { const int xw = l_misc[1 + w]; if (w < wave_id()) basew += xw; tot += xw; }
    __syncthreads();
    if (k < K) { off[k] = carry + basew + inc - v; d.det[ko + k] = 0; }
    carry += tot;
  }
  if (threadIdx.x == 0) { off[K] = carry; d.info[s].C = carry; d.info[s].K = K; d.slot_kc[d.cur][s] = make_int2(K, carry); }
  // ---- work items of the per-cluster reductions: cluster k owns ceil(size/MOR_CHUNK) chunks
  int *coff = d.chunk_off[d.cur] + (size_t)s * (d.Kcap + 1);
  carry = 0;
  __syncthreads();
  for (int b = 0; b < K; b += NT) {
    const int k = b + threadIdx.x, v = k < K ? (d.csz[ko + k] + MOR_CHUNK - 1) / MOR_CHUNK : 0;
    const int inc = wave_incl_scan(v);
    if (lane == 63) l_misc[1 + wave_id()] = inc;
    __syncthreads();
    int basew = 0, tot = 0;
    for (int w = 0; w < NT / 64; ++w) { const int xw = l_misc[1 + w]; if (w < wave_id()) basew += xw; tot += xw; }
    __syncthreads();
    if (k < K) coff[k] = carry + basew + inc - v;
    carry += tot;
  }
  if (threadIdx.x == 0) coff[K] = carry;
  // ---- the cells of every cluster as a list, and every cell's place in the cluster's range of cl_pts: cluster k owns
  //      cl_pts[off[k], off[k+1]); its cells take consecutive pieces of it in the order their atomics arrive, except the
  //      cell holding the cluster's first point (smallest cloud index), which takes the first piece.  (Any order will do:
  //      what is computed from cluster points — counts, existence tests, min / max, exact integer sums — does not depend
  //      on it; read-backs that promise the reference's order rebuild it from the labels.)
  int *ncell = scr, *cur = scr + K;   // the scratch array is free by now: [K] cells per cluster → first list entry; [K] next free slot of the cluster's range
  if (2 * K > nocc) { ncell = d.nn_bwd + ko; cur = d.nn_fwd + ko; }   // (more kept clusters than half the cells — min_cluster_size 1 on a sparse cloud: two [B][Kcap] arrays of the pair stage, which runs behind this kernel in the same frame and writes them before it reads them; LDS-scope accesses on global memory are fine inside one workgroup after the barriers)
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += NT) cg_st<LDS>(ncell + k, 0);
  __threadfence_block();
  __syncthreads();
  for (int c = threadIdx.x; c < nocc; c += NT) { const int k = cg_ld<LDS>(par + c); if (k >= 0) atomicAdd(&ncell[k], 1); }
  __threadfence_block();
  __syncthreads();
  int *lcoff = d.cl_coff + (size_t)s * (d.Kcap + 1);
  carry = 0;
  for (int b = 0; b < K; b += NT) {
    const int k = b + threadIdx.x, v = k < K ? cg_ld<LDS>(ncell + k) : 0;
    const int inc = wave_incl_scan(v);
    if (lane == 63) l_misc[1 + wave_id()] = inc;
    __syncthreads();
    int basew = 0, tot = 0;
    for (int w = 0; w < NT / 64; ++w) { const int xw = l_misc[1 + w]; if (w < wave_id()) basew += xw; tot += xw; }
    __syncthreads();
    if (k < K) { const int e = carry + basew + inc - v; lcoff[k] = e; cg_st<LDS>(ncell + k, e); }
    carry += tot;
  }
  if (threadIdx.x == 0) lcoff[K] = carry;
  __threadfence_block();
  __syncthreads();
  // the cell with the cluster's first point opens the range
  for (int c = threadIdx.x; c < nocc; c += NT) {
    const int k = cg_ld<LDS>(par + c);
    if (k >= 0 && d.cmin[so + c] == d.kroot[ko + d.krank_inv[ko + k]]) cg_st<LDS>(cur + k, off[k] + (start[c + 1] - start[c]));
  }
  __threadfence_block();
  __syncthreads();
  for (int c = threadIdx.x; c < nocc; c += NT) {
    const int k = cg_ld<LDS>(par + c);
    int4 g = make_int4(0, -1, -1, 0);
    if (k >= 0) {
      const int first = d.kroot[ko + d.krank_inv[ko + k]], n = start[c + 1] - start[c];
      const bool opens = d.cmin[so + c] == first;
      const int dst = opens ? off[k] : atomicAdd(&cur[k], n);
      g = make_int4(dst - start[c], k, opens ? first : -1, 0);
      d.clist[so + atomicAdd(&ncell[k], 1)] = c;
    }
    d.cgat[so + c] = g;
  }
}
// ---- the same tail for the common case, restructured around what it waits for (round 6).  Cutting the tail's phases out one at a time (exp/cut_time.py) put it at 46 of
// k_cg_slab's 108 µs alone on the open scenes and 61 of 184 on the street scenes, spread over every phase: each is a sweep over the stream's cells that begins with global loads
// (local roots, cell sizes, smallest indices, per-cluster records through two dependent hops) behind a workgroup barrier — a dozen exposed round trips per thread and phase.  Here a
// thread keeps the operands of ITS cells (cells tid, tid + NT, …: at most RC of them) in registers — one batch of coalesced loads for the whole tail —, the forest starts as the
// stars the slabs published (a plain store per cell: a slab's local root is the smallest id of its local component, so parent ≤ child holds) and only the look-ahead cells of every
// slab are united across slabs, and the per-cluster records (root cell, size, first cloud index, offset, cursors: CGF_KL of them at most) live in LDS.  Same results: every quantity is
// the one cgf_body computes.  Returns false — nothing published yet, the caller runs cgf_body from scratch — for streams with more than RC·NT cells or more than CGF_KL kept clusters.
template <int NT, int RC> __device__ __forceinline__ bool cgf_fast(const MorDev &d, int s, int nocc, int *par, int *scr, int *kl, int *l_misc, const int *l_sc, const int *l_se) {
  static_assert(CGF_KL <= NT && CGF_KL <= 1024, "one round of the per-cluster scans; ten bits of cluster id beside a cell's list entry");
  if (nocc > RC * NT) return false;
  const size_t so = (size_t)s * d.Nmax, ko = (size_t)s * d.Kcap;
  const int *start = d.cstart + (size_t)s * (d.Nmax + 1);
  const int tid = threadIdx.x, lane = lane_id(), wv = wave_id(), P = d.slab_p[s];
  int *k_cell = kl, *k_size = kl + CGF_KL, *k_root = kl + 2 * CGF_KL, *k_first = kl + 3 * CGF_KL, *k_off = kl + 4 * CGF_KL, *k_n = kl + 5 * CGF_KL, *k_cur = kl + 6 * CGF_KL;
  int n[RC], mnc[RC], r[RC];   // per own cell: points, smallest cloud index, root (later: cluster id)
  // ---- operands + the stars of the slab forests
#pragma unroll
  for (int i = 0; i < RC; ++i) {
    const int c = tid + i * NT;
    n[i] = 0; mnc[i] = 0x7fffffff; r[i] = -1;
    if (c < nocc) { const int la = ld_agent(&d.lroot_a[so + c]); n[i] = start[c + 1] - start[c]; mnc[i] = d.cmin[so + c]; par[c] = la; scr[c] = 0; }
  }
  if (tid == 0) l_misc[0] = 0;
  __syncthreads();
  // ---- cross-slab unions: the look-ahead cells of slab j − 1 are the first cells of slab j, [sc[j], se[j − 1]) — (c, its local root in slab j − 1's forest)
  {
    int tot = 0;
    for (int j = 1; j < P; ++j) tot += (l_sc[j] > l_sc[j - 1]) ? max(l_se[j - 1] - l_sc[j], 0) : 0;   // (an empty slab publishes nothing)
    for (int it = tid; it < tot; it += NT) {
      int j = 1, rem = it;
      for (; j < P; ++j) { const int len = (l_sc[j] > l_sc[j - 1]) ? max(l_se[j - 1] - l_sc[j], 0) : 0; if (rem < len) break; rem -= len; }
      const int c = l_sc[j] + rem;
      cg_unite<true>(par, c, ld_agent(&d.lroot_b[so + c]));
    }
  }
  __threadfence_block();
  __syncthreads();
#pragma unroll
  for (int i = 0; i < RC; ++i) { const int c = tid + i * NT; if (c < nocc) r[i] = cg_find<true>(par, c); }
  // ---- components: size (points) at the root
#pragma unroll
  for (int i = 0; i < RC; ++i) if (r[i] >= 0) atomicAdd(&scr[r[i]], n[i]);
  __threadfence_block();
  __syncthreads();
  // ---- kept components (:215-216): the root's scratch word (its size, > 0) becomes −(entry) − 2; the entry starts the minimum of its cloud indices
#pragma unroll
  for (int i = 0; i < RC; ++i) {
    const int c = tid + i * NT;
    if (r[i] == c) {
      const long long sz = (long long)cg_ld<true>(scr + c);
      if (sz >= d.min_cs && sz <= d.max_cs) { const int k = atomicAdd(&l_misc[0], 1); if (k < CGF_KL) { k_size[k] = (int)sz; k_root[k] = 0x7fffffff; cg_st<true>(scr + c, -k - 2); } }
    }
  }
  __threadfence_block();
  __syncthreads();
  const int K = l_misc[0];
  if (K > CGF_KL || K > d.Kcap) return false;
  // ---- smallest cloud index of every kept component (the others' are never asked for); r[i] becomes the cell's entry (−1: not in a kept component)
#pragma unroll
  for (int i = 0; i < RC; ++i) {
    int e = -1;
    if (r[i] >= 0) { const int v = cg_ld<true>(scr + r[i]); if (v < 0) { e = -v - 2; atomicMin(&k_root[e], mnc[i]); } }
    r[i] = e;
  }
  __threadfence_block();
  __syncthreads();
  // ---- cluster order: size descending, ties by smaller first cloud index; rank by counting
  if (tid < K) {
    const int my_sz = k_size[tid], my_rt = k_root[tid]; int rank = 0;
    for (int u = 0; u < K; ++u) { const int sz = k_size[u], rt = k_root[u]; rank += (sz > my_sz) || (sz == my_sz && rt < my_rt); }
    k_cell[tid] = rank;   // entry → cluster id
    k_first[rank] = my_rt; k_off[rank] = my_sz; k_n[rank] = 0; k_cur[rank] = 0;
  }
  __threadfence_block();
  __syncthreads();
  // ---- per-cell cluster id (a cell is a clique ⇒ one cluster); the cell's piece of its cluster's range of cl_pts and its entry in the cluster's list of cells, both RELATIVE for
  //      now (cgf_body: the cell holding the cluster's first point opens the range, the others take consecutive pieces behind it in the order their atomics arrive)
  int rel[RC];
#pragma unroll
  for (int i = 0; i < RC; ++i) {
    const int c = tid + i * NT;
    rel[i] = 0;
    if (c < nocc) {
      const int id = r[i] >= 0 ? k_cell[r[i]] : -1;
      d.ccid[so + c] = id; reinterpret_cast<int *>(&d.cmeta[2 * (so + c)])[3] = id;
      if (id >= 0) {
        const bool opens = mnc[i] == k_first[id];
        if (opens) k_root[id] = n[i]; else rel[i] = atomicAdd(&k_cur[id], n[i]);   // (k_root is free by now: per cluster, the points of the cell that opens its range)
        r[i] = id | (atomicAdd(&k_n[id], 1) << 10);                               // cluster id (< CGF_KL ≤ 1024) | entry in the cluster's list of cells
      } else r[i] = -1;
    } else r[i] = -1;
  }
  __threadfence_block();
  __syncthreads();
  // ---- three exclusive scans in cluster order, one round: points → cl_off, reduction chunks → chunk_off, cells → cl_coff; C, K, detection_results cleared (:250-254)
  {
    const bool v_ = tid < K;
    const int sz = v_ ? k_off[tid] : 0, ch = v_ ? (sz + MOR_CHUNK - 1) / MOR_CHUNK : 0, nc = v_ ? k_n[tid] : 0;
    const int i0 = wave_incl_scan(sz), i1 = wave_incl_scan(ch), i2 = wave_incl_scan(nc);
    int *part = l_misc + 1;   // (≥ 3 · NT / 64 words behind the counter: the callers' l_misc)
    if (lane == 63) { part[wv] = i0; part[NT / 64 + wv] = i1; part[2 * (NT / 64) + wv] = i2; }
    __syncthreads();
    int b0 = 0, b1 = 0, b2 = 0, t0 = 0, t1 = 0, t2 = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) { const int x0 = part[w], x1 = part[NT / 64 + w], x2 = part[2 * (NT / 64) + w]; if (w < wv) { b0 += x0; b1 += x1; b2 += x2; } t0 += x0; t1 += x1; t2 += x2; }
    int *off = d.cl_off[d.cur] + (size_t)s * (d.Kcap + 1), *coff = d.chunk_off[d.cur] + (size_t)s * (d.Kcap + 1), *lcoff = d.cl_coff + (size_t)s * (d.Kcap + 1);
    if (v_) {
      const int o0 = b0 + i0 - sz, o2 = b2 + i2 - nc;
      off[tid] = o0; coff[tid] = b1 + i1 - ch; lcoff[tid] = o2; d.det[ko + tid] = 0;
      k_off[tid] = o0; k_n[tid] = o2;
    }
    if (tid == 0) { off[K] = t0; coff[K] = t1; lcoff[K] = t2; d.info[s].C = t0; d.info[s].K = K; d.slot_kc[d.cur][s] = make_int2(K, t0); }
  }
  __threadfence_block();
  __syncthreads();
  // ---- every cell's place and the cells of every cluster as a list: absolute now
#pragma unroll
  for (int i = 0; i < RC; ++i) {
    const int c = tid + i * NT;
    if (c >= nocc) continue;
    int4 g = make_int4(0, -1, -1, 0);
    if (r[i] >= 0) {
      const int k = r[i] & 1023, first = k_first[k]; const bool opens = mnc[i] == first;
      const int dst = k_off[k] + (opens ? 0 : k_root[k] + rel[i]);
      g = make_int4(dst - start[c], k, opens ? first : -1, 0);
      d.clist[so + k_n[k] + (r[i] >> 10)] = c;
    }
    d.cgat[so + c] = g;
  }
  return true;
}
__global__ __launch_bounds__(CGF_T) void k_cg_final(MorDev d) {
  const int s = blockIdx.x + d.s0, nocc = d.info[s].n_occ;
  const size_t so = (size_t)s * d.Nmax;
  __shared__ int l_par[CGF_CAP], l_a[CGF_CAP], l_misc[1 + 3 * (CGF_T / 64)], l_sc[MOR_MAXP + 1], l_se[MOR_MAXP + 1];
  if (threadIdx.x <= d.slab_p[s]) { l_sc[threadIdx.x] = d.slab_c[(size_t)s * (MOR_MAXP + 1) + threadIdx.x]; l_se[threadIdx.x] = d.slab_e[(size_t)s * (MOR_MAXP + 1) + threadIdx.x]; }
  if (nocc <= CGF_CAP && !d.cg_force_global) {          // forest, sizes (then cluster ids), minima: three LDS arrays
    __syncthreads();
    if (!d.cg_slow_tail && nocc + 7 * CGF_KL <= CGF_CAP && cgf_fast<CGF_T, CGF_RC>(d, s, nocc, l_par, l_a, l_a + nocc, l_misc, l_sc, l_se)) return;
    __syncthreads();
    for (int i = threadIdx.x; i < nocc; i += CGF_T) l_par[i] = i;
    __syncthreads();
    cgf_body<true, CGF_T>(d, s, nocc, l_par, l_a, l_misc, l_sc, l_se);
  } else {
    int *par = d.parent + so;
    for (int i = threadIdx.x; i < nocc; i += CGF_T) cg_st<false>(par + i, i);
    __threadfence_block();   // (the forest is this workgroup's alone and accessed with agent-scope operations; a device-wide fence writes the XCD's whole L2 back: 32 µs)
    __syncthreads();
    cgf_body<false, CGF_T>(d, s, nocc, par, d.csize + so, l_misc, l_sc, l_se);
  }
}

