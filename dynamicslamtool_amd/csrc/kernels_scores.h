// kernels_scores.h — part of mor_kernels.hip (one translation unit: #included there, in dependency order; not a stand-alone header).
// P3, P4, P5 (:309-366, :580-606): movement scores by method 1 (three tiers) and method 2 (voxel hash set); thresholds + host summary.
// Reference citations are file:line of /root/reference/src/MovingObjectRemoval.cpp.
// ------------------------------------------------------------------------------------ P3: method 1 (:336-366)
// Per point q of a matched previous cluster: squared distance to the nearest point of the matched current cluster;
// count lb < d² < ub (:356).  Only the CLASS of that distance matters (≤ lb, inside (lb, ub), ≥ ub), so the search is
// two existence tests: E2 "some matched point closer than √ub" and E1 "some matched point within √lb"; q is counted
// iff E2 ∧ ¬E1.  All points of a grid cell belong to one component (the cell is a clique), so the cluster id is a
// per-CELL attribute (ccid, also in .w of the cell's box record): candidates are filtered cell by cell without
// touching their points, and a cell's point box gives a lower bound that prunes it.
//   tier 1  k_score_fast   thread per query, its own cell (LDS cell index → cell → points): static surfaces end here
//   tier 1a k_score_nb (front of the worklist)   thread per query with E2 known: the ≤ 7 neighbour cells across the walls q is close to
//   tier 1b k_score_nb (back of the worklist)    thread per query whose own cell has no matched point: E2 (then E1) in the 3×3×3 block
//   tier 2  k_score_pde    wave per query for what is left: big cells, matches farther than one cell
// Lesson of the profile: a thread's time is the NUMBER of dependent load levels (≈ 2 µs each under load), not bytes;
// every tier is written as a few levels of batched independent loads.
// Threshold T for the wave tier: a region with lower bound ≥ T can be skipped — ub while E2 is open; once best < ub
// only regions that could hold a point within lb matter (`best` then need not be the true minimum).
__device__ __forceinline__ float score_lim(float best, float lbn /* smallest float > lb */, float ub) { return best < ub ? fminf(best, lbn) : ub; }
__device__ __forceinline__ float box_dist2(const float4 &q, const float4 &lo, const float4 &hi) {
  float gx = fmaxf(fmaxf(lo.x - q.x, q.x - hi.x), 0.f), gy = fmaxf(fmaxf(lo.y - q.y, q.y - hi.y), 0.f), gz = fmaxf(fmaxf(lo.z - q.z, q.z - hi.z), 0.f);
  return (gx * gx + gy * gy + gz * gz) * 0.999f;   // conservative
}
// scan sorted positions [b,e) (one cell of the matched cluster), four independent loads at a time; returns as soon as best < stopv
template <int W> __device__ __forceinline__ void scan_ws(const float4 *sp, int b, int e, const float4 &q, float stopv, float &best, int &budget) {
  for (int k = b; k < e && budget > 0; k += W, budget -= W) {
    float4 p[W];
#pragma unroll
    for (int u = 0; u < W; ++u) p[u] = sp[min(k + u, e - 1)];
#pragma unroll
    for (int u = 0; u < W; ++u) best = fminf(best, sqdist(q.x, q.y, q.z, p[u].x, p[u].y, p[u].z));
    if (best < stopv) return;
  }
}
__device__ __forceinline__ void scan4s(const float4 *sp, int b, int e, const float4 &q, float stopv, float &best, int &budget) { scan_ws<4>(sp, b, e, q, stopv, best, budget); }
// (a wave pays for its slowest lane, and nearly every wave has a lane that goes through its whole budget: eight loads per
//  round trip halve the dependent levels of that lane)
__device__ __forceinline__ void scan8s(const float4 *sp, int b, int e, const float4 &q, float stopv, float &best, int &budget) { scan_ws<8>(sp, b, e, q, stopv, best, budget); }
// The same over a cell of more than `budget` points, sampled evenly: positions b, b+step, b+2·step, …  Points arrive in a
// cell in scan order, so the first 64 of a 1000-point cell all come from one corner of it; an even sample of the whole
// cell finds a point within √lb of q (if there is one: on a dense static surface ≈ 6 % of the cell's points qualify)
// nearly always, and only genuine misses go on to the wave tier, which scans the whole cell.
__device__ __forceinline__ void scan_sampled(const float4 *sp, int b, int e, const float4 &q, float stopv, float &best, int &budget) {
  const int n = e - b, step = max(n / budget, 1);
  for (int k = 0; k < n && budget > 0; k += 8 * step, budget -= 8) {
    float4 p[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) p[u] = sp[b + min(k + u * step, n - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u) best = fminf(best, sqdist(q.x, q.y, q.z, p[u].x, p[u].y, p[u].z));
    if (best < stopv) return;
  }
}
// Which neighbour cells can hold a point within √lb of q: per axis −1 / +1 when q lies within `slb` (√lb, padded for
// the rounding of the cell map) of the low / high wall of its cell, else 0.  Valid when 2·slb < cell edge.
__device__ __forceinline__ int near_side(float v, float o, float inv, float cs, int c, float slb) {
  const float f = ((v - o) * inv - (float)c) * cs;   // distance to the low wall
  return f <= slb ? -1 : (cs - f <= slb ? 1 : 0);
}
// wave-aggregated append of a query to a per-stream worklist; `back`: the list grows downwards from list[cap−1]
// An entry is (query, pair, matched cluster) so the next tier starts without the chain query → cluster → pair → match.
__device__ __forceinline__ void wl_push(bool want, int *n, int4 *list, int j, int pr, int target, int own_cell, bool back = false, int cap = 0) {
  unsigned long long m = __ballot(want);
  if (!m) return;
  int basew = 0, leader = __ffsll((long long)m) - 1;
  if (lane_id() == leader) basew = atomicAdd(n, __popcll(m));
  basew = wave_bcast(basew, leader);   // (the leader comes from a ballot: uniform — v_readlane instead of a trip through the LDS crossbar)
  if (want) { int pos = basew + __popcll(m & lanemask_lt()); list[back ? cap - 1 - pos : pos] = make_int4(j, pr, target, own_cell); }
}
// wave-aggregated count: all counted queries of a pair add to ONE address (a few dozen addresses per stream), and
// same-address atomics serialise in L2 — thousands of them per stream were the real cost of these kernels.  Lanes
// with the same pair are combined first (worklist order is cluster order, so usually one atomic per wave).
__device__ __forceinline__ void count_push(bool want, int *cnt, int pr) {
  unsigned long long m = __ballot(want);
  while (m) {
    const int l = __ffsll((long long)m) - 1, p = wave_bcast(pr, l);
    const unsigned long long same = __ballot(want && pr == p);
    if (lane_id() == l) atomicAdd(&cnt[p], __popcll(same));
    m &= ~same;
  }
}
// Tier 1 — one THREAD per query, its OWN cell only.  On a static surface a point of the matched cluster lies
// within √lb of q, almost always in q's own cell: ≈ 85 % of the queries end here (never counted).  The rest is
// compacted into worklists so the next tiers run full waves of like queries: `wl` front = E2 known (a matched point of
// the own cell closer than √ub), `wl` back = own cell without a matched point, `wl2` = big own cell (wave tier).
#ifndef SCF_BUDGET
#define SCF_BUDGET 64   // samples of its own cell a thread of tier 1 looks at
#endif
#define SCF_T 1024   // threads per workgroup of tier 1: sixteen waves share one LDS copy of the stream's cell index (loading it per 256 queries cost more than the lookups saved)
#ifndef SCF_MINW
#define SCF_MINW 8   // ≤ 64 VGPRs: two 1024-thread workgroups per CU (69 VGPRs were one)
#endif
__global__ __launch_bounds__(SCF_T, SCF_MINW) void k_score_fast(MorDev d) {
  int s, t0, g_fast;
  if (!map_block_work(d, [&](int s_) { return (d.slot_kc[d.prev][s_].y + SCF_T - 1) / SCF_T; }, s, t0, g_fast)) return;   // work: rounds of one workgroup over ca's cluster points
  const MorGrid G = stream_grid(d, s);   // the stream's clustering grid (voxel ground variant: its own number of z layers)
  const int pv = d.prev, Cp = d.slot_kc[d.prev][s].y;
  const size_t so = (size_t)s * d.Nmax, ko = (size_t)s * d.Kcap;
  const float4 *sp = d.sorted + so;
  __shared__ unsigned short l_idx[CIDX_CAP];
  if (t0 * SCF_T >= Cp) return;   // nothing for this workgroup: not worth a copy of the cell index
  const CellIdx I = cidx_load(d, G, s, l_idx);
  const int *st = d.cstart + (size_t)s * (d.Nmax + 1);
  const float lbn = nextafterf(d.pde_lb, INFINITY);
  const float slb = sqrtf(fmaxf(d.pde_lb, 0.f)) * 1.01f + G.cs * 1e-3f;
  const bool e1_local = 2.f * slb < G.cs;
  for (int base = t0 * SCF_T; base < Cp; base += g_fast * SCF_T) {
    const int j = base + threadIdx.x;
    bool nearq = false, blockq = false, big = false, counted = false; float best = INFINITY; int pr = -1, target = -1, own_c = -1;
    if (j < Cp) {
      // two short chains of dependent loads, issued side by side (no branch between them): cluster → its record (pair, matched cluster,
      // that cluster's box), and point → cell (LDS index) → range + cluster id of the cell → points.  (Round 1: seven levels, one after the other.)
      const int cidj = ld_stream(&d.cl_cid[pv][so + j]);
      const float4 q = ld_stream(&d.cl_pts[pv][so + j]);   // (read once here; the few queries the later tiers take up again fetch theirs from HBM)
      const int cx = cell_axis_unclamped(q.x, G.ox, G.inv_cs), cy = cell_axis_unclamped(q.y, G.oy, G.inv_cs), cz = cell_axis_unclamped(q.z, d.zorg[s], G.inv_cs);
      const int c = cidx_find(I, cx, cy, cz);   // (LDS: no global access)
      own_c = c;
      const float4 tlo = d.qrec[2 * (ko + cidj)], thi = d.qrec[2 * (ko + cidj) + 1];   // the matched cluster's box, the pair, the matched cluster: one record per previous cluster (pairs_body)
      const int cc = max(c, 0), cid = d.ccid[so + cc], b0 = c >= 0 ? st[cc] : 0, e0 = c >= 0 ? st[cc + 1] : 0;
      pr = __float_as_int(tlo.w); target = __float_as_int(thi.w);
      if (pr >= 0) {
        int budget = SCF_BUDGET;   // a big own cell that shows no close point among SCF_BUDGET evenly spread samples goes to the wave tier
        const bool reach = box_dist2(q, tlo, thi) < d.pde_ub;   // farther than √ub from the whole matched cluster: never counted
        if (reach && c >= 0 && cid == target) { scan_sampled(sp, b0, e0, q, lbn, best, budget); big = best > d.pde_lb && e0 - b0 > SCF_BUDGET; }
        if (reach && best > d.pde_lb && !big) {
          if (!e1_local) big = true;   // √lb reaches beyond the adjacent half-cells in this configuration: wave tier
          else if (best < d.pde_ub) {
            if (near_side(q.x, G.ox, G.inv_cs, G.cs, cx, slb) == 0 && near_side(q.y, G.oy, G.inv_cs, G.cs, cy, slb) == 0 && near_side(q.z, d.zorg[s], G.inv_cs, G.cs, cz, slb) == 0)
              counted = true;   // deep inside its cell: no other cell can hold a point within √lb ⇒ counted
            else nearq = true;
          } else blockq = true;
        }
      } else target = -1;
    }
    {  // the worklists: both ends of `wl` with ONE returning atomic per wave (the two counters share a 64-bit word), `wl2` with another,
       // both issued by lane 0 before either answer is used (one round trip instead of two)
      const unsigned long long mn = __ballot(nearq), mb = __ballot(blockq), mg = __ballot(big);
      if (mn | mb | mg) {
        unsigned long long base = 0ull; int base2 = 0;
        if (lane_id() == 0) {
          if (mn | mb) base = atomicAdd(&d.wl_nb[s], (unsigned long long)__popcll(mn) | ((unsigned long long)__popcll(mb) << 32));
          if (mg) base2 = atomicAdd(&d.wl2_n[s], __popcll(mg));
        }
        const int bn = wave_bcast((int)(unsigned)base, 0), bb = wave_bcast((int)(base >> 32), 0), b2 = wave_bcast(base2, 0);
        if (nearq) d.wl[so + bn + __popcll(mn & lanemask_lt())] = make_int4(j, pr, target, 0);
        if (blockq) d.wl[so + d.Nmax - 1 - (bb + __popcll(mb & lanemask_lt()))] = make_int4(j, pr, target, 0);
        if (big) d.wl2[so + b2 + __popcll(mg & lanemask_lt())] = make_int4(j, pr, target, own_c);   // (.w: the query's own cell — the wave tier starts there without looking it up again)
      }
    }
    count_push(counted, d.pair_cnt + ko, pr);
  }
}
// one batch of four cells: box records and point ranges with independent loads, then the scans.  A cell is scanned up
// to the first point within lb when its box allows one (E1); while E2 is open also when its box allows a point < ub.
__device__ __forceinline__ void scan_batch4(const MorDev &d, size_t so, const int *st, const float4 *sp, const int (&c4)[4], int target, bool check_target, const float4 &q,
                                            float lbn, float &best, int &budget) {
  float4 blo[4], bhi[4]; int b0[4], e0[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int c = max(c4[i], 0); blo[i] = d.cmeta[2 * (so + c)]; bhi[i] = d.cmeta[2 * (so + c) + 1]; b0[i] = st[c]; e0[i] = st[c + 1]; }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (c4[i] < 0 || (check_target && __float_as_int(blo[i].w) != target) || !(best > d.pde_lb) || budget <= 0) continue;
    const float bd = box_dist2(q, blo[i], bhi[i]);
    if (bd < lbn) scan8s(sp, b0[i], e0[i], q, lbn, best, budget);
    else if (!(best < d.pde_ub) && bd < d.pde_ub) scan8s(sp, b0[i], e0[i], q, d.pde_ub, best, budget);
  }
}
// Worklist tiers: SCN_T consecutive entries per workgroup and chunk.  (Measured and dropped: entry e → workgroup e % G; chunks of 64 dealt
// over the workgroups; lanes of a wave nrows apart — all slower.)
#define SCN_T 512   // threads per workgroup of tiers 1a / 1b (eight waves share one LDS copy of the cell index)

// Tier 1a — one THREAD per query with E2 known (worklist front).  E1: only the ≤ 7 neighbour cells across the walls q
// is close to can hold a point within √lb (the own cell was scanned by tier 1).  Three levels of loads: cell lookups (LDS index) →
// box records + ranges → points.  No such point ⇒ counted.
__device__ __forceinline__ void score_near_body(const MorDev &d, const CellIdx &I, int s, int chunk) {
  const MorGrid G = stream_grid(d, s);   // the stream's clustering grid (voxel ground variant: its own number of z layers)
  const int pv = d.prev, nq = (int)(unsigned)d.wl_nb[s];
  const size_t so = (size_t)s * d.Nmax, ko = (size_t)s * d.Kcap;
  const float4 *sp = d.sorted + so;
  const int *st = d.cstart + (size_t)s * (d.Nmax + 1);
  const float lbn = nextafterf(d.pde_lb, INFINITY);
  const float slb = sqrtf(fmaxf(d.pde_lb, 0.f)) * 1.01f + G.cs * 1e-3f;
  {
    const int w = chunk * SCN_T + threadIdx.x;
    bool defer = false, counted = false; int j = 0, pr = -1, target = -1, own_c = -1;
    if (w < nq) {
      const int4 we = d.wl[so + w]; j = we.x; pr = we.y; target = we.z;
      const float4 q = d.cl_pts[pv][so + j];
      const int cx = cell_axis_unclamped(q.x, G.ox, G.inv_cs), cy = cell_axis_unclamped(q.y, G.oy, G.inv_cs), cz = cell_axis_unclamped(q.z, d.zorg[s], G.inv_cs);
      const int sx = near_side(q.x, G.ox, G.inv_cs, G.cs, cx, slb), sy = near_side(q.y, G.oy, G.inv_cs, G.cs, cy, slb), sz = near_side(q.z, d.zorg[s], G.inv_cs, G.cs, cz, slb);
      int budget = d.t1_budget; float best = 0.5f * (d.pde_lb + d.pde_ub) ;   // any value inside (lb, ub): E2 holds
      if (!(best > d.pde_lb && best < d.pde_ub)) best = d.pde_ub * 0.999f;
      int id[8];
      id[0] = -1;
#pragma unroll
      for (int i = 1; i < 8; ++i) {
        const int ax = i & 1, ay = (i >> 1) & 1, az = i >> 2;
        const bool valid = !(ax && sx == 0) && !(ay && sy == 0) && !(az && sz == 0);
        id[i] = valid ? cidx_find(I, cx + ax * sx, cy + ay * sy, cz + az * sz) : -1;
      }
      const int ca[4] = {id[1], id[2], id[4], id[3]}, cb2[4] = {id[5], id[6], id[7], -1};   // face neighbours first
      scan_batch4(d, so, st, sp, ca, target, true, q, lbn, best, budget);
      if ((cb2[0] >= 0 || cb2[1] >= 0 || cb2[2] >= 0) && best > d.pde_lb) scan_batch4(d, so, st, sp, cb2, target, true, q, lbn, best, budget);
      if (best > d.pde_lb) { if (budget <= 0) { defer = true; own_c = cidx_find(I, cx, cy, cz); } else counted = true; }
    }
    count_push(counted, d.pair_cnt + ko, pr);
    wl_push(defer, &d.wl2_n[s], d.wl2 + so, j, pr, target, own_c);
  }
}
// Tier 1b — one THREAD per query whose own cell holds no matched point (worklist back).  The 26 other cells of the
// 3×3×3 block: cell lookups (LDS index, row by row) → cluster ids → up to 8 matched cells (those that can hold a point within √lb first) →
// box records + ranges → points.  E2 hit ⇒ E1 is decided by the same cells; no hit ⇒ the wider stencil is the wave tier's job.
__device__ __forceinline__ void score_block_body(const MorDev &d, const CellIdx &I, int s, int chunk) {
  const MorGrid G = stream_grid(d, s);   // the stream's clustering grid (voxel ground variant: its own number of z layers)
  const int pv = d.prev, nq = (int)(d.wl_nb[s] >> 32);
  const size_t so = (size_t)s * d.Nmax, ko = (size_t)s * d.Kcap;
  const float4 *sp = d.sorted + so;
  const int *st = d.cstart + (size_t)s * (d.Nmax + 1);
  const int *cid_c = d.ccid + so;
  const float lbn = nextafterf(d.pde_lb, INFINITY);
  const float slb = sqrtf(fmaxf(d.pde_lb, 0.f)) * 1.01f + G.cs * 1e-3f;
  const bool stencil27 = d.n_rows <= 9 && d.score_R <= 1;   // the whole search stencil is the 3×3×3 block
  {
    const int w = chunk * SCN_T + threadIdx.x;
    bool defer = false, counted = false; int j = 0, pr = -1, target = -1, own_c = -1;
    if (w < nq) {
      const int4 we = d.wl[so + d.Nmax - 1 - w]; j = we.x; pr = we.y; target = we.z;
      const float4 q = d.cl_pts[pv][so + j];
      const int cx = cell_axis_unclamped(q.x, G.ox, G.inv_cs), cy = cell_axis_unclamped(q.y, G.oy, G.inv_cs), cz = cell_axis_unclamped(q.z, d.zorg[s], G.inv_cs);
      const int sx = near_side(q.x, G.ox, G.inv_cs, G.cs, cx, slb), sy = near_side(q.y, G.oy, G.inv_cs, G.cs, cy, slb), sz = near_side(q.z, d.zorg[s], G.inv_cs, G.cs, cz, slb);
      int budget = d.t1_budget; float best = INFINITY;
      int id[27];   // the 3×3×3 block row by row: one row of the LDS index holds the (≤ 3) cells x − 1 … x + 1 as consecutive ids
#pragma unroll
      for (int rw = 0; rw < 9; ++rw) {
        const int y = cy + rw % 3 - 1, z = cz + rw / 3 - 1;
        int lo = 0, hi = 0;
        if ((unsigned)y < (unsigned)I.ny && (unsigned)z < (unsigned)I.nz) cidx_row(I, max(cx - 1, 0), min(cx + 1, I.nx - 1), y, z, lo, hi);
        int x0 = -9, x1 = -9, x2 = -9;   // x of the (≤ 3) cells found
        if (lo < hi) x0 = I.lds ? (int)I.cx16[lo] : I.ckey[lo] - (y * I.nz + z) * I.nx;
        if (lo + 1 < hi) x1 = I.lds ? (int)I.cx16[lo + 1] : I.ckey[lo + 1] - (y * I.nz + z) * I.nx;
        if (lo + 2 < hi) x2 = I.lds ? (int)I.cx16[lo + 2] : I.ckey[lo + 2] - (y * I.nz + z) * I.nx;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int xw = cx + dx - 1;
          id[rw * 3 + dx] = x0 == xw ? lo : x1 == xw ? lo + 1 : x2 == xw ? lo + 2 : -1;
        }
      }
      id[13] = -1;   // (the own cell was tier 1's)
      int mc[8]; int ncand = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) mc[i] = -1;
      {
        int cidv[27];
#pragma unroll
        for (int i = 0; i < 27; ++i) cidv[i] = cid_c[max(id[i], 0)];
#pragma unroll
        for (int pass = 0; pass < 2; ++pass)   // pass 0: cells that can hold a point within √lb (≤ 7), pass 1: the others
#pragma unroll
          for (int i = 0; i < 27; ++i) {
            const int dx = i % 3 - 1, dy = (i / 3) % 3 - 1, dz = i / 9 - 1;
            const bool nearc = (dx == 0 || dx == sx) && (dy == 0 || dy == sy) && (dz == 0 || dz == sz);
            if (id[i] >= 0 && cidv[i] == target && nearc == (pass == 0)) {
#pragma unroll
              for (int k = 0; k < 8; ++k) if (ncand == k) mc[k] = id[i];
              ++ncand;
            }
          }
      }
      if (ncand > 0) {
        const int ca[4] = {mc[0], mc[1], mc[2], mc[3]}, cb2[4] = {mc[4], mc[5], mc[6], mc[7]};
        scan_batch4(d, so, st, sp, ca, target, false, q, lbn, best, budget);
        if (ncand > 4 && best > d.pde_lb) scan_batch4(d, so, st, sp, cb2, target, false, q, lbn, best, budget);
      }
      if (best > d.pde_lb) {
        if (budget <= 0) defer = true;
        else if (best < d.pde_ub) counted = true;   // all cells that can hold a point within √lb were among the slots
        else if (!(stencil27 && ncand <= 8)) defer = true;              // E2 still open: wider search
      }
      if (defer) own_c = cidx_find(I, cx, cy, cz);   // (the wave tier starts at the query's own cell: its id travels with the entry)
    }
    count_push(counted, d.pair_cnt + ko, pr);
    wl_push(defer, &d.wl2_n[s], d.wl2 + so, j, pr, target, own_c);
  }
}
#ifndef SCN_MINW
#define SCN_MINW 1
#endif
// Tiers 1a and 1b in ONE launch (both only need tier 1's worklists; as two launches in two pieces of the frame pipeline they cost a
// launch boundary and a queueing delay each).  The two ends of the worklist are cut into chunks of SCN_T entries — the front's chunks
// first, then the back's — and the stream's g_score workgroups take the chunks round-robin, so the split between the two tiers follows
// the lists (≈ 1700 and ≈ 900 entries per stream on the headline workload: five chunks) and a workgroup without a chunk leaves before
// it copies the cell index.  A stream's workgroups share an XCD (its cell tables stay in that L2).
__global__ __launch_bounds__(SCN_T, SCN_MINW) void k_score_nb(MorDev d) {
  int s, bx, g_score;
  if (!map_block_work(d, [&](int s_) { const unsigned long long v = d.wl_nb[s_]; return ((int)(unsigned)v + SCN_T - 1) / SCN_T + ((int)(v >> 32) + SCN_T - 1) / SCN_T; }, s, bx, g_score)) return;   // work: chunks of the two worklists
  __shared__ unsigned short l_idx[CIDX_CAP];
  const unsigned long long nb = d.wl_nb[s];
  const int cn = ((int)(unsigned)nb + SCN_T - 1) / SCN_T, cb = ((int)(nb >> 32) + SCN_T - 1) / SCN_T;
  if (bx >= cn + cb) return;
  const CellIdx I = cidx_load(d, stream_grid(d, s), s, l_idx);
  for (int c = bx; c < cn + cb; c += g_score) {
    if (c < cn) score_near_body(d, I, s, c); else score_block_body(d, I, s, c - cn);
  }
}
__device__ __forceinline__ float wave_min(float v) { return wave_fmin(v); }
// Tier 2 — one WAVE per deferred query.  Every lane owns one ROW of the search stencil (nearest rows
// first, 64 rows per round) and walks that row's cells with a cursor: cell-level work (cluster id,
// box distance) is lane-parallel; every surviving cell is then scanned by the whole wave, 128 points
// per iteration, and `best` tightens the pruning of everything that follows.  Rows are ordered by
// their lower bound, so a round in which no row can beat min(best, ub) ends the search (a neighbour
// at d² ≥ ub is never counted), and so does best ≤ lb.
__device__ __forceinline__ float wave_scan_cell(const float4 *sp, int b0, int e0, const float4 &q, float lbv, int lane) {
  float local = INFINITY;
  for (int k0 = b0; k0 < e0; k0 += 256) {   // four loads per lane and round trip: a cell of 3000 points is 12 dependent levels, not 24
    const int k = k0 + lane;
    const float4 p = sp[min(k, e0 - 1)], p2 = sp[min(k + 64, e0 - 1)], p3 = sp[min(k + 128, e0 - 1)], p4 = sp[min(k + 192, e0 - 1)];
    local = fminf(fminf(local, fminf(sqdist(q.x, q.y, q.z, p.x, p.y, p.z), sqdist(q.x, q.y, q.z, p2.x, p2.y, p2.z))), fminf(sqdist(q.x, q.y, q.z, p3.x, p3.y, p3.z), sqdist(q.x, q.y, q.z, p4.x, p4.y, p4.z)));
    if (__ballot(local <= lbv)) break;
  }
  return wave_min(local);
}
#define SCP_T 256    // threads per workgroup of the wave tier.  (Tried: 1024-thread workgroups sharing an LDS copy of the cell index, 4 / 32 per stream: 346 / 90 µs against 56 — a stream's few hundred deferred queries want a thousand waves, and a workgroup with one query does not pay for a table.)
__device__ __forceinline__ void score_pde_body(const MorDev &d, int s, int bx, int g_pde, unsigned short *l_idx) {
  const MorGrid G = stream_grid(d, s);   // the stream's clustering grid (voxel ground variant: its own number of z layers)
  const int pv = d.prev, nq = d.wl2_n[s];
  const int wv = bx * (SCP_T / 64) + wave_id(), nw = g_pde * (SCP_T / 64), lane = lane_id();
  const size_t so = (size_t)s * d.Nmax, ko = (size_t)s * d.Kcap;
  const float4 *sp = d.sorted + so;
  const int *st = d.cstart + (size_t)s * (d.Nmax + 1);
  if (bx * (SCP_T / 64) >= nq) return;   // (uniform: none of the stream's deferred queries falls to this workgroup — most workgroups of most streams)
  const CellIdx I = cidx_load(d, G, s, l_idx);   // (l_idx null: the 32-bit tables in global memory)
  const int R = d.score_R;
  const int *cid_c = d.ccid + so;
  const float lbn = nextafterf(d.pde_lb, INFINITY);
  const float cs = G.cs * 0.999f;   // conservative cell edge for the row lower bounds
  int acc_pr = -1, acc = 0;   // counts of consecutive queries of one pair are flushed together
  for (int w = wv; w < nq; w += nw) {
    const int4 we = d.wl2[so + w];
    const int j = we.x, pr = we.y, target = we.z;
    const float4 q = d.cl_pts[pv][so + j];
    const int cx = cell_axis_unclamped(q.x, G.ox, G.inv_cs), cy = cell_axis_unclamped(q.y, G.oy, G.inv_cs), cz = cell_axis_unclamped(q.z, d.zorg[s], G.inv_cs);
    float best = INFINITY;
    {  // the query's own cell first (its id came with the entry: the tier that deferred the query had looked it up in its LDS index — here that was a chain of ten dependent loads)
      const int c = we.w;
      if (c >= 0 && d.ccid[so + c] == target) best = wave_scan_cell(sp, st[c], st[c + 1], q, d.pde_lb, lane);
    }
    for (int rb = 0; rb < d.n_rows && best > d.pde_lb; rb += 64) {
      // lanes: resolve one row each → cursor [cur, hi) over its cells
      int ro = rb + lane, cur = 0, hi = 0; float lbrow = INFINITY;
      if (ro < d.n_rows) {
        int dy = d.row_order[2 * ro], dz = d.row_order[2 * ro + 1];
        float ly = (float)max(abs(dy) - 1, 0) * cs, lz = (float)max(abs(dz) - 1, 0) * cs;
        lbrow = ly * ly + lz * lz;
        float room = score_lim(best, lbn, d.pde_ub) - lbrow;   // a useful neighbour in this row needs dx² < room
        int y = cy + dy, z = cz + dz;
        if (room > 0.f && (unsigned)y < (unsigned)G.ny && (unsigned)z < (unsigned)G.nz) {
          int rx = min(R, (int)(sqrtf(room) * G.inv_cs * 1.001f) + 1);
          int x0 = max(cx - rx, 0), x1 = min(cx + rx, G.nx - 1);
          if (x0 <= x1) cidx_row(I, x0, x1, y, z, cur, hi);
        }
      }
      if (wave_bcast(lbrow, 0) >= score_lim(best, lbn, d.pde_ub)) break;   // rows are ordered by their lower bound
      // cluster ids of the first 8 cells of the lane's row as one batch of independent loads → bit mask of matched cells
      unsigned rowmask = 0; const int base = cur;
      {
        int idv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) idv[u] = cid_c[min(base + u, d.Nmax - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) if (base + u < hi && idv[u] == target) rowmask |= 1u << u;
      }
      cur = min(base + 8, hi);   // cells beyond the batch are walked one by one
      for (;;) {
        // lane-parallel: advance to the next cell of the matched cluster whose box can still improve the class of `best`
        int cand = -1;
        const float lim = score_lim(best, lbn, d.pde_ub);
        if (lbrow < lim) {
          while (rowmask) {
            const int c = base + __ffs(rowmask) - 1; rowmask &= rowmask - 1;
            if (box_dist2(q, d.cmeta[2 * (so + c)], d.cmeta[2 * (so + c) + 1]) < lim) { cand = c; break; }
          }
          while (cand < 0 && cur < hi) {
            int c = cur++;
            const float4 blo = d.cmeta[2 * (so + c)], bhi = d.cmeta[2 * (so + c) + 1];
            if (__float_as_int(blo.w) == target && box_dist2(q, blo, bhi) < lim) { cand = c; break; }
          }
        }
        if (!__ballot(cand >= 0)) break;
        // small surviving cells are scanned by the lane that found them (all rows in parallel); big ones by the whole wave
        const int cb = cand >= 0 ? st[cand] : 0, ce = cand >= 0 ? st[cand + 1] : 0;
        const bool small = cand >= 0 && ce - cb <= 16;
        float local = INFINITY;
        if (small) { int budget = 0x7fffffff; scan4s(sp, cb, ce, q, lbn, local, budget); }
        best = fminf(best, wave_min(local));
        unsigned long long m = __ballot(cand >= 0 && !small);
        while (m && best > d.pde_lb) {
          int l = __ffsll((long long)m) - 1; m &= m - 1;
          int c = wave_bcast(cand, l);
          if (box_dist2(q, d.cmeta[2 * (so + c)], d.cmeta[2 * (so + c) + 1]) >= score_lim(best, lbn, d.pde_ub)) continue;   // best may have tightened since
          best = fminf(best, wave_scan_cell(sp, st[c], st[c + 1], q, d.pde_lb, lane));
        }
        if (best <= d.pde_lb) break;
      }
    }
    if (best > d.pde_lb && best < d.pde_ub) {
      if (pr != acc_pr) { if (lane == 0 && acc) atomicAdd(&d.pair_cnt[ko + acc_pr], acc); acc_pr = pr; acc = 0; }
      ++acc;
    }
  }
  if (lane == 0 && acc) atomicAdd(&d.pair_cnt[ko + acc_pr], acc);
}
__global__ __launch_bounds__(SCP_T) void k_score_pde(MorDev d) {
  // a wave per deferred query: the launch's workgroups go to the streams in proportion to their queues (a few hundred queries in one stream, none in the
  // next), spread over all XCDs
  int s, bx, g;
  if (!map_block_work<false, true>(d, [&](int s_) { return (d.wl2_n[s_] + SCP_T / 64 - 1) / (SCP_T / 64); }, s, bx, g)) return;
  score_pde_body(d, s, bx, g, nullptr);
}
// (Tried: thresholds + tracking step in the stream's last workgroup of this kernel.  The tracking step of frame k must follow frame
//  k − 1's filterCloud, so the whole wave tier then waited for it and the frames stopped overlapping: 150 k → 125 k frame-pairs/s.)

// ------------------------------------------------------------------------------------ P4: method 2 (:309-334)
// OctreePointCloudChangeDetector as a voxel hash set.  PCL grows its octree from the first inserted
// point p0 of the previous cluster: box = p0 ± res/2, which getKeyBitSize() widens to two voxels per
// axis and re-centres (min = p0 − res); every later growth shifts min by a multiple of res.  So the
// leaf lattice is {p0 − res + k·res}; keys by floor in fp64.
#define VOX_EMPTY 0xFFFFFFFFFFFFFFFFull
__device__ __forceinline__ int vox_table_size(const MorDev &d, int Cprev) { int h = 64; while (h < 2 * Cprev && h < d.Hcap) h <<= 1; return h; }
__device__ __forceinline__ unsigned long long vox_hash(unsigned long long k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33; return k; }
__device__ __forceinline__ bool vox_key(const MorDev &d, int pr, const double (&mn)[3], float4 p, unsigned long long &key) {
  const double res = d.opc_res;
  long long kk[3]; const float pc[3] = {p.x, p.y, p.z};
  bool ok = pr < 65535;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    // floor((p − min) / res) as the octree computes its keys (fp64 DIVISION) — by a multiplication with 1 / res where that cannot differ: the product lies within 3·2⁻⁵² · 65 536 < 1e-10 of the
    // quotient, so when it is farther than 1e-7 from an integer both have the same floor; the few points closer to a voxel face than that take the division.  (Three fp64 divisions per
    // point — thirty instructions each — were the bulk of the arithmetic of both voxel kernels.)
    const double t = (double)pc[a] - mn[a], q = t * d.opc_inv_res; double f = floor(q); const double r = q - f;
    if (r < 1e-7 || r > 1.0 - 1e-7) f = floor(t / res);
    kk[a] = (long long)f;
    ok = ok && kk[a] >= -32768 && kk[a] < 32768;
  }
  key = ((unsigned long long)pr << 48) | ((unsigned long long)(kk[0] + 32768) << 32) | ((unsigned long long)(kk[1] + 32768) << 16) | (unsigned long long)(kk[2] + 32768);
  return ok;
}
__global__ __launch_bounds__(MOR_BT) void k_vox_clear(MorDev d) {
  int s, bxc; map_block(d.B, 64, s, bxc);
  const int H = vox_table_size(d, d.slot_kc[d.prev][s].y);
  unsigned long long *tab = d.vox + (size_t)s * d.Hcap;
  for (int i = bxc * MOR_BT + threadIdx.x; i < H; i += 64 * MOR_BT) tab[i] = VOX_EMPTY;
}
__global__ __launch_bounds__(MOR_BT) void k_vox_insert(MorDev d) {
  int s, t, g;   // the launch's workgroups go to the streams in proportion to their cluster points (tiles of MOR_TILE)
  if (!map_block_work(d, [&](int s_) { return (d.slot_kc[d.prev][s_].y + MOR_TILE - 1) / MOR_TILE; }, s, t, g)) return;
  const int pv = d.prev, Cp = d.slot_kc[d.prev][s].y;
  const size_t so = (size_t)s * d.Nmax, ko = (size_t)s * d.Kcap;
  unsigned long long *tab = d.vox + (size_t)s * d.Hcap; int H = vox_table_size(d, Cp);
  for (int base = t * MOR_TILE; base < Cp; base += g * MOR_TILE)
  for (int j = base + threadIdx.x; j < min(base + MOR_TILE, Cp); j += MOR_BT) {
    const MorVoxRec vr = d.vrec[2 * ko + ld_stream(&d.cl_cid[pv][so + j])];
    const int pr = vr.pr;
    if (pr < 0) continue;
    unsigned long long key;
    if (!vox_key(d, pr, vr.mn, ld_stream(&d.cl_pts[pv][so + j]), key)) { mor_raise(d, s, 2u); continue; }
    unsigned h = (unsigned)vox_hash(key) & (H - 1);
    for (;;) {   // (a look first: most points find their voxel in the table already, and compare-and-swaps of many lanes on one slot queue up in L2)
      unsigned long long old = __hip_atomic_load(&tab[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == VOX_EMPTY) old = atomicCAS(&tab[h], VOX_EMPTY, key);
      if (old == VOX_EMPTY || old == key) break;
      h = (h + 1) & (H - 1);
    }
  }
}
__global__ __launch_bounds__(MOR_BT) void k_vox_probe(MorDev d) {
  int s, t, g;
  if (!map_block_work(d, [&](int s_) { return ((int)d.info[s_].C + MOR_TILE - 1) / MOR_TILE; }, s, t, g)) return;
  const int C = d.info[s].C;
  const size_t so = (size_t)s * d.Nmax, ko = (size_t)s * d.Kcap;
  const unsigned long long *tab = d.vox + (size_t)s * d.Hcap; int H = vox_table_size(d, d.slot_kc[d.prev][s].y);
  for (int base = t * MOR_TILE; base < C; base += g * MOR_TILE)
  for (int j0 = base; j0 < min(base + MOR_TILE, C); j0 += MOR_BT) {   // (all lanes stay in the loop: the counts of a wave are combined per pair before they go to memory)
    const int j = j0 + threadIdx.x;
    bool fresh = false; int pr = -1;
    if (j < min(base + MOR_TILE, C)) {
      const MorVoxRec vr = d.vrec[2 * ko + d.Kcap + ld_stream(&d.cl_cid[d.cur][so + j])];
      pr = vr.pr;
      if (pr >= 0) {
        unsigned long long key;
        if (!vox_key(d, pr, vr.mn, ld_stream(&d.cl_pts[d.cur][so + j]), key)) mor_raise(d, s, 2u);
        else {
          unsigned h = (unsigned)vox_hash(key) & (H - 1); bool found = false;
          for (;;) { unsigned long long v = tab[h]; if (v == key) { found = true; break; } if (v == VOX_EMPTY) break; h = (h + 1) & (H - 1); }
          fresh = !found;   // a point of cb in a voxel that holds no point of ca (:319-330)
        }
      }
    }
    count_push(fresh, d.pair_cnt + ko, pr);   // (one atomic per wave and pair: thousands of single adds to a pair's counter serialise in L2 — 180 µs of this kernel)
  }
}

// ------------------------------------------------------------------------------------ P5 + summary to the host
// scores → detection_results (:580-606); then everything the host tracker needs goes straight
// into pinned host memory (a few KB per stream), so the push needs exactly one stream sync.
// (runs at the head of k_track_push: one workgroup per stream, NT threads)
template <int NT> __device__ __forceinline__ void decide_body(const MorDev &d, int s) {
  const int pv = d.prev;
  const size_t ko = (size_t)s * d.Kcap;
  MorFrameInfo f = d.info[s];
  int np = d.has_prev ? (int)f.n_pairs : 0;
  const int *offc = d.cl_off[d.cur] + (size_t)s * (d.Kcap + 1), *offp = d.cl_off[pv] + (size_t)s * (d.Kcap + 1);
  for (int pr = threadIdx.x; pr < np; pr += NT) {
    int q = d.pair_q[ko + pr], m = d.pair_m[ko + pr];
    unsigned long long n1 = (unsigned long long)(offp[q + 1] - offp[q]), n2 = (unsigned long long)(offc[m + 1] - offc[m]);
    double cnt = (double)ld_agent(&d.pair_cnt[ko + pr]), score, thr;   // (agent scope: in the fused form the counts come from atomics of the other workgroups of this launch)
    if (d.method == 1) { score = cnt / (double)((n1 + n2) / 2ull); thr = d.pde_thr; }             // :361, :586
    else if (d.method == 2) { score = cnt; thr = (double)((n1 + n2) / (unsigned long long)d.opc_norm); } // :330, :590
    else { score = 0; thr = 0; }
    d.det[ko + m] = score > thr;                                                                     // :593-604
    d.h_pair_q[ko + pr] = q; d.h_pair_m[ko + pr] = m; d.h_pair_d[ko + pr] = d.pair_d[ko + pr]; d.h_score[ko + pr] = score;
  }
  __syncthreads();
  int K = (int)f.K;
  for (int k = threadIdx.x; k < K; k += NT) { d.h_centroid[ko + k] = d.centroid[d.cur][ko + k]; d.h_det[ko + k] = d.det[ko + k]; }
  for (int k = threadIdx.x; k <= K; k += NT) d.h_cl_off[(size_t)s * (d.Kcap + 1) + k] = offc[k];
  if (threadIdx.x == 0) {
    f.n_pairs = np; f.n_defer = (d.has_prev && d.method == 1) ? (uint32_t)d.wl2_n[s] : 0u; f.pad0 = (d.has_prev && d.method == 1) ? (uint32_t)((unsigned)d.wl_nb[s] + (unsigned)(d.wl_nb[s] >> 32)) : 0u;
    f.Kprev = d.has_prev ? d.slot_kc[d.prev][s].x : 0; f.Cprev = d.has_prev ? d.slot_kc[d.prev][s].y : 0;   // for the host mirror
    d.info[s].n_pairs = np;
    d.h_info[s] = f;
  }
  {  // per-frame summary for the host (tests compare every frame of an asynchronous run through it)
    __shared__ unsigned l_sum[2];
    if (threadIdx.x == 0) { l_sum[0] = 0; l_sum[1] = 0; }
    __syncthreads();
    unsigned cs = 0, ds = 0;
    for (int pr = threadIdx.x; pr < np; pr += NT) cs += (unsigned)ld_agent(&d.pair_cnt[ko + pr]) * (unsigned)(2 * pr + 1) + (unsigned)d.pair_m[ko + pr];
    for (int k = threadIdx.x; k < K; k += NT) ds += d.det[ko + k] ? (unsigned)(k + 1) : 0u;
    atomicAdd(&l_sum[0], cs); atomicAdd(&l_sum[1], ds);
    __syncthreads();
    if (threadIdx.x == 0) {
      MorFrameLog &L = d.h_log[(size_t)(d.frame_no % MOR_LOG_CAP) * d.Btot + s];
      L.frame = d.frame_no; L.K = K; L.C = (int)f.C; L.n_pairs = np; L.cnt_sum = l_sum[0]; L.det_sum = l_sum[1]; L.flags = (int)d.info[s].flags;
    }
  }
}

