// kernels_grid.h — part of mor_kernels.hip (one translation unit: #included there, in dependency order; not a stand-alone header).
// the grids: cells of key-sorted points (voxel ground variant), the 16-bit cell index of the scoring tiers, cells by counting (k_gridcount / k_gridhash / k_gridplace).
// Reference citations are file:line of /root/reference/src/MovingObjectRemoval.cpp.
// ------------------------------------------------------------------------------------ grid: distinct cells of the key-sorted points
// (the sort itself is the generic radix below: k_rhist / k_rscan / k_rscatter)
__device__ __forceinline__ bool is_head(const int *skey, int p) { return p == 0 || skey[p] != skey[p - 1]; }
__global__ __launch_bounds__(MOR_BT) void k_heads_count(MorDev d) {
  int s, t0; map_block(d.B, d.tiles_m, s, t0);
  const int M = d.info[s].M;
  const int *skey = d.rkeys[voxel_passes_of(d, s) & 1] + (size_t)s * d.Nmax;   // (the ping-pong buffer the stream's last radix pass wrote)
  __shared__ int sh[8];
  for (int t = t0; t * MOR_TILE < M; t += d.tiles_m) {   // grid-stride over the tiles this stream really has
    int base = t * MOR_TILE, c = 0, tot;
    for (int p = base + threadIdx.x; p < min(base + MOR_TILE, M); p += MOR_BT) c += is_head(skey, p);
    block_excl_scan(c, sh, &tot);
    if (threadIdx.x == 0) d.ktile_cnt[(size_t)s * d.tiles_max + t] = tot;
  }
}
// ------------------------------------------------------------------------------------ cell index of the scoring tiers (method 1)
// The scoring tiers look cells up by coordinates a few million times per batch.  Round 2 did that through a hash table in global memory
// (16-byte slots, eight per cell: cleared — 33 MB per step — and filled by the grid build, probed with random 16-byte loads that every
// other kernel in flight paid for).  The grid is small: a stream has a few thousand occupied cells in a few thousand (y,z) rows, so every
// scoring workgroup now keeps the stream's ROW TABLE (first compact id of every row) and the x of every cell in its own LDS as 16-bit
// values (k_gridhash writes them once per frame: rs16, cx16) and a lookup is two LDS reads plus a short scan of the row — no global access.
// Streams whose tables do not fit (more than CIDX_CAP entries, or ≥ 65 536 cells) use the 32-bit tables in global memory (L2).
#define CIDX_CAP 12288   // 16-bit entries per workgroup (24 KB: six 256-thread workgroups per CU)
struct CellIdx { const unsigned short *rs16, *cx16; const int *rs, *ckey; int nx, ny, nz; bool lds; };
__device__ __forceinline__ CellIdx cidx_load(const MorDev &d, const MorGrid &G, int s, unsigned short *l_idx) {   // l_idx null: the global tables (kernels with a handful of lookups per workgroup)
  CellIdx I; I.nx = G.nx; I.ny = G.ny; I.nz = G.nz;
  I.rs = d.row_start + (size_t)s * (d.g.nrows + 1); I.ckey = d.ckey + (size_t)s * d.Nmax;
  const int nocc = (int)d.info[s].n_occ, nr = G.nrows + 1;
  I.lds = l_idx != nullptr && nocc <= 65535 && nr + nocc + 2 <= CIDX_CAP;
  const int nr2 = (nr + 1) & ~1;   // (the x table starts at an even entry: both tables are copied two entries at a time)
  if (I.lds) {
    const unsigned *g_rs = reinterpret_cast<const unsigned *>(d.rs16 + (size_t)s * d.rs16_stride), *g_cx = reinterpret_cast<const unsigned *>(d.cx16 + (size_t)s * d.cx16_stride);   // (both strides are even: the copies below move two entries at a time from 4-byte aligned addresses)
    unsigned *l32 = reinterpret_cast<unsigned *>(l_idx);
    for (int i = threadIdx.x; i < nr2 / 2; i += blockDim.x) l32[i] = g_rs[i];
    for (int i = threadIdx.x; i < (nocc + 1) / 2; i += blockDim.x) l32[nr2 / 2 + i] = g_cx[i];
    I.rs16 = l_idx; I.cx16 = l_idx + nr2;
  } else { I.rs16 = nullptr; I.cx16 = nullptr; }
  if (l_idx) __syncthreads();
  return I;
}
// occupied cells with x in [x0, x1] of row (cy,cz): the consecutive compact ids [lo, hi)  (cy, cz inside the grid)
__device__ __forceinline__ void cidx_row(const CellIdx &I, int x0, int x1, int cy, int cz, int &lo, int &hi) {
  const int r = cy * I.nz + cz;
  if (!I.lds) {
    const int e = I.rs[r + 1], base = r * I.nx; int a = I.rs[r], b = e; const int k0 = base + x0, k1 = base + x1;   // (binary searches on purpose: the batched loads of row_range cost k_score_pde 25 registers — 8 → 5 waves per SIMD — and the run 2 %)
    while (a < b) { const int m = (a + b) >> 1; if (I.ckey[m] < k0) a = m + 1; else b = m; }
    lo = a; b = e;
    while (a < b) { const int m = (a + b) >> 1; if (I.ckey[m] <= k1) a = m + 1; else b = m; }
    hi = a; return;
  }
  int a = I.rs16[r]; const int e = I.rs16[r + 1];
  if (e - a > 8) { int b = e; while (a < b) { const int m = (a + b) >> 1; if ((int)I.cx16[m] < x0) a = m + 1; else b = m; } }   // long row (a wall along x)
  else while (a < e && (int)I.cx16[a] < x0) ++a;
  lo = a;
  while (a < e && (int)I.cx16[a] <= x1) ++a;   // (windows are a few cells wide)
  hi = a;
}
// compact id of cell (cx,cy,cz) or −1 when empty / outside
__device__ __forceinline__ int cidx_find(const CellIdx &I, int cx, int cy, int cz) {
  if ((unsigned)cx >= (unsigned)I.nx || (unsigned)cy >= (unsigned)I.ny || (unsigned)cz >= (unsigned)I.nz) return -1;
  int lo, hi; cidx_row(I, cx, cx, cy, cz, lo, hi);
  return lo < hi ? lo : -1;
}
__device__ __forceinline__ unsigned hash_slot(int key, unsigned hshift) { return ((unsigned)key * 0x9E3779B1u) >> hshift; }   // (the LDS / global cell tables of the grid build)
// linear key of cell (cx,cy,cz), −1 outside the grid
__device__ __forceinline__ int cell_key(const MorGrid &g, int cx, int cy, int cz) {
  if ((unsigned)cx >= (unsigned)g.nx || (unsigned)cy >= (unsigned)g.ny || (unsigned)cz >= (unsigned)g.nz) return -1;
  return grid_key(g, cx, cy, cz);
}
// per sorted position: compact cell id; heads publish the cell; every point lands in `sorted`
__global__ __launch_bounds__(MOR_BT) void k_heads_scatter(MorDev d) {
  int s, t0; map_block(d.B, d.tiles_m, s, t0);
  const MorGrid G = stream_grid(d, s);   // the stream's clustering grid (voxel ground variant: its own number of z layers)
  const int M = d.info[s].M;
  const size_t so = (size_t)s * d.Nmax;
  const int np_s = voxel_passes_of(d, s);
  const int *skey = d.rkeys[np_s & 1] + so, *sidx = d.rvals[np_s & 1] + so;   // (the ping-pong buffers the stream's last radix pass wrote: keys in order, inverse permutation)
  int *cstart = d.cstart + (size_t)s * (d.Nmax + 1);
  __shared__ int sh[12], l_gap[3 * 64], l_ng;
  int *rs = d.row_start + (size_t)s * (d.g.nrows + 1);
  const int nt = (M + MOR_TILE - 1) / MOR_TILE;
  if (threadIdx.x == 0) l_ng = 0;
  if (M == 0 && t0 == 0) for (int r = threadIdx.x; r <= G.nrows; r += MOR_BT) rs[r] = 0;   // no cells: every row starts (and ends) at 0
  {   // number of occupied cells: every workgroup sums the tile counts itself (no separate scan launch)
    int pre, nocc; wg_prefix_total(d.ktile_cnt + (size_t)s * d.tiles_max, 1, 0, nt, sh, pre, nocc);
    if (t0 == 0 && threadIdx.x == 0) { d.info[s].n_occ = nocc; cstart[nocc] = M; }
  }
  for (int t = t0; t * MOR_TILE < M; t += d.tiles_m) {
    const int base = t * MOR_TILE + wave_id() * 512;
    unsigned long long mh[8]; int cnt = 0;
#pragma unroll
    for (int it = 0; it < 8; ++it) { int p = base + it * 64 + lane_id(); mh[it] = __ballot(p < M && is_head(skey, p)); cnt += __popcll(mh[it]); }
    if (lane_id() == 0) sh[wave_id()] = cnt;
    __syncthreads();
    int r;
    { int tot; wg_prefix_total(d.ktile_cnt + (size_t)s * d.tiles_max, 1, t, nt, sh + 4, r, tot); }
    for (int w = 0; w < wave_id(); ++w) r += sh[w];
    __syncthreads();
    // The points MOVE to their places: `sidx` holds the inverse permutation (input index → sorted position, written by the last radix pass), so the
    // trimmed cloud is read in input order (coalesced) and every point is stored to its slot — writes nobody waits for, which the L2 combines (points that
    // follow each other in a sweep fall into the same or neighbouring voxels).  (Rounds 2–4 gathered: sorted position → input index → point, 7 M dependent
    // random 16-byte reads per step that each pulled a whole sector: 650 MB and 417 µs alone.)
    {
      int pi[8]; float4 pq[8];
#pragma unroll
      for (int it = 0; it < 8; ++it) { const int i = base + it * 64 + lane_id(); pi[it] = i < M ? ld_stream(sidx + i) : 0; pq[it] = ld_stream(&d.cloud[so + min(i, M - 1)]); }
#pragma unroll
      for (int it = 0; it < 8; ++it) { const int i = base + it * 64 + lane_id(); if (i < M) { float4 q = pq[it]; q.w = __int_as_float(i); d.sorted[so + pi[it]] = q; } }
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      int p = base + it * 64 + lane_id();
      if (p < M) {
        bool head = (mh[it] >> lane_id()) & 1ull;
        int c = r + __popcll(mh[it] & lanemask_lt()) + (head ? 1 : 0) - 1;
        if (head) {
          const int kc = skey[p];
          d.ckey[so + c] = kc; cstart[c] = p;
          // dense (y,z) row table: rs[r] = first cell with key ≥ r·nx.  The head of cell c owns the rows after its
          // predecessor's row up to its own (keys ascend), so the table is written without any search
          const int rc = kc / G.nx, rp = p > 0 ? skey[p - 1] / G.nx : -1;
          if (rc - rp > 16) { const int g = atomicAdd(&l_ng, 1); if (g < 64) { l_gap[3 * g] = rp; l_gap[3 * g + 1] = rc; l_gap[3 * g + 2] = c; } else for (int r = rp + 1; r <= rc; ++r) rs[r] = c; }
          else for (int r = rp + 1; r <= rc; ++r) rs[r] = c;
        }
        if (p == M - 1) {   // rows behind the last cell (and the end sentinel) start at n_occ
          const int rl = skey[p] / G.nx;
          const int g = atomicAdd(&l_ng, 1); if (g < 64) { l_gap[3 * g] = rl; l_gap[3 * g + 1] = G.nrows; l_gap[3 * g + 2] = c + 1; } else for (int r = rl + 1; r <= G.nrows; ++r) rs[r] = c + 1;
        }
        if (d.scell) d.scell[so + p] = c;
      }
      r += __popcll(mh[it]);
    }
    __syncthreads();
    for (int g = 0, ng = min(l_ng, 64); g < ng; ++g)   // long runs of empty rows (between z layers, before the first and after the last cell): the whole workgroup fills them
      for (int r = l_gap[3 * g] + 1 + threadIdx.x; r <= l_gap[3 * g + 1]; r += MOR_BT) rs[r] = l_gap[3 * g + 2];
    __syncthreads();
    if (threadIdx.x == 0) l_ng = 0;
    __syncthreads();
  }
}
// ------------------------------------------------------------------------------------ grid, hash path: cells by counting, not by sorting points
// A stream's non-ground cloud has FEW occupied cells (a few thousand) but a heavy-tailed number of points per cell
// (a wall two metres from the sensor puts thousands of returns into one 28-cm cell; M ranges 5 k … 60 k over the
// streams of one batch).  Sorting all points by key moves every point three times; what the later stages need is only:
// the distinct keys in ascending order (compact cell ids), the points grouped by cell (any order inside a cell — every
// consumer tests existence, takes a min / max or counts), and the (y,z) row table.  So the build
//   1. counts the points of every cell in LDS hash tables (open addressing; LDS atomics digest the hot cells) — chunk by chunk, many
//      workgroups per stream (k_gridcount),
//   2. merges the chunks' short lists of distinct cells in ONE workgroup per stream (k_gridhash) and orders the cells through the row
//      table: a cell's compact id is its row's first id plus the number of cells of the row with a smaller x — no sort,
//   3. turns the counts into ranges of `sorted`, and every point draws its position from an LDS cursor of its chunk's entry (k_gridplace).
// k_gridhash also writes the 16-bit row table + x of every cell that the scoring tiers copy into their LDS (CellIdx above).
// Streams with more cells than the LDS tables hold run the same code on tables in global memory (tiers 1 / 2).
#define CB_WTILE 256      // positions of `sorted` one wave of k_cellboxes handles per step (four consecutive ones per lane)
#ifndef GH_T
#define GH_T 1024
#endif
#ifndef GH_U
#define GH_U 4        // points per thread and round trip of the sweeps (8 / 12 measured: no gain — the sweeps are bound by LDS atomics on the hot cells, not by the loads)
#endif
#define GH_H 16384       // slots of the LDS table (cells ≤ 3/4 of it)
#define GH_ROWS 7039     // (y,z) rows the LDS copy of the row table holds
#ifndef GH_SHORT
#define GH_SHORT 32      // rows of at most this many cells rank their cells by counting; longer rows (a wall along x) through a bitmap of their x (≈ 1.5 µs of a wave per row: with 8 here the open scenes lost what the urban ones won)
#endif
template <int NT> __device__ __forceinline__ int block_excl_scan_n(int v, int *sh, int *total) {   // sh: ≥ NT/64 ints
  const int inc = wave_incl_scan(v);
  __syncthreads();
  if (lane_id() == 63) sh[wave_id()] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) { const int x = sh[w]; if (w < wave_id()) base += x; tot += x; }
  *total = tot;
  return base + inc - v;
}
template <bool L> __device__ __forceinline__ int gh_ld(const int *p) {
  return L ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool L> __device__ __forceinline__ void gh_st(int *p, int v) {
  if (L) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// in-place exclusive scan of a[0, n) by the whole workgroup; returns the total.  LDS arrays: a contiguous chunk per thread.
// Global arrays (long row tables, big cell lists): every wave owns a contiguous segment and walks it 64 elements at a time —
// coalesced accesses, one wave scan per step — instead of a chain of dependent single loads per thread.
template <bool L> __device__ __forceinline__ int gh_scan(int *a, int n, int *sh) {
  if (L) {
    const int chunk = (n + GH_T - 1) / GH_T, b = min((int)threadIdx.x * chunk, n), e = min(b + chunk, n);
    int sum = 0;
    for (int i = b; i < e; ++i) sum += gh_ld<L>(a + i);
    int total; int run = block_excl_scan_n<GH_T>(sum, sh, &total);
    for (int i = b; i < e; ++i) { const int v = gh_ld<L>(a + i); gh_st<L>(a + i, run); run += v; }
    __syncthreads();
    return total;
  }
  constexpr int NW = GH_T / 64;
  const int seg = ((n + NW - 1) / NW + 63) / 64 * 64, b = min(wave_id() * seg, n), e = min(b + seg, n), lane = lane_id();
  int sum = 0;
  for (int i = b + lane; i < e; i += 64) sum += gh_ld<L>(a + i);
  sum = wave_sum(sum);
  __syncthreads();
  if (lane == 0) sh[wave_id()] = sum;
  __syncthreads();
  int run = 0, total = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) { const int x = sh[w]; if (w < wave_id()) run += x; total += x; }
  for (int i0 = b; i0 < e; i0 += 64) {
    const int i = i0 + lane, v = i < e ? gh_ld<L>(a + i) : 0, inc = wave_incl_scan(v);
    if (i < e) gh_st<L>(a + i, run + inc - v);
    run += wave_bcast(inc, 63);
  }
  __syncthreads();
  return total;
}
// Slab boundaries of the cell graph (k_cg_slab): P slabs of whole y-slices with about equal cell counts, each at least
// two slices thick so that the two-slice look-ahead of a slab stays inside its successor.  rows = exclusive row table
// (rows[r] = first compact id of row r, rows[nrows] = n_occ); threads 0 … P of the calling workgroup take part; sh: ≥ 41 ints (sh[40] holds P).
template <bool L> __device__ __forceinline__ void slab_bounds(const MorDev &d, const MorGrid &G, int s, const int *rows, int nocc, int *sh) {
  const int ny = G.ny, nz = G.nz, j = threadIdx.x;
  // Slabs of this stream: the launch's width for every stream, or (map_block_work) as many as the stream's cells ask for at slab_T own cells a slab —
  // a stream of 5 400 cells then gets four times the workgroups of one with 1 300 instead of slabs four times as big.  k_cg_slab runs ONE slab per
  // workgroup and its launch holds P + 1 workgroups per stream, shared within an XCD group: every stream has one slab for sure and draws the others
  // from the group's budget (a counter in the group's first stream's ticket words, reset by the frame's first kernel); a stream that finds the
  // budget short gets fewer, larger slabs (never seen with slab_T from the previous frame's counts; any partition gives the same components).
  if (j == 0) {
    int P = d.P;
    if (d.prop_map && d.slab_T > 0) {
      const bool x8 = (d.B & 7) == 0 && d.xcd_map;
      const int ng = x8 ? d.B >> 3 : d.B, budget = ng * d.P, first = x8 ? d.s0 + ((s - d.s0) & 7) : d.s0;   // (P + 1 workgroups per stream in the launch, one of them the stream's own)
      const int want = max(1, min((nocc + d.slab_T - 1) / d.slab_T, min(MOR_MAXP, max(1, ny / 2)))) - 1;
      int extra = 0;
      if (want > 0) { const int base = __hip_atomic_fetch_add(d.tickets + (size_t)first * TK_COUNT + TK_SLABCNT, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); extra = max(0, min(want, budget - base)); }
      P = 1 + extra;
    }
    d.slab_p[s] = P; sh[40] = P;
  }
  __syncthreads();
  const int P = sh[40];
  __syncthreads();
  if (j <= P) {
    int y = j == 0 ? 0 : ny;
    if (j > 0 && j < P) {   // smallest y whose first cell id reaches the j-th share of the cells
      const int target = (int)((long long)nocc * j / P);
      int a = 0, b = ny;
      while (a < b) { const int m = (a + b) >> 1; if (gh_ld<L>(rows + m * nz) >= target) b = m; else a = m + 1; }
      y = a;
    }
    sh[j] = y;
  }
  __syncthreads();
  if (j == 0) for (int k = 1; k < P; ++k) sh[k] = min(max(sh[k], sh[k - 1] + 2), ny);
  __syncthreads();
  if (j <= P) {
    int *sy = d.slab_y + (size_t)s * (MOR_MAXP + 1), *sc = d.slab_c + (size_t)s * (MOR_MAXP + 1), *se = d.slab_e + (size_t)s * (MOR_MAXP + 1);
    sy[j] = sh[j]; sc[j] = gh_ld<L>(rows + sh[j] * nz);
    se[j] = j < P ? gh_ld<L>(rows + min(sh[j + 1] + 2, ny) * nz) : nocc;   // end of slab j's look-ahead (cells of the next two y-slices)
    if (j < P) atomicMax(&d.info[s].max_loc, (unsigned)(se[j] - sc[j]));   // the host picks the kernel variant of the next frames by it
  }
  __syncthreads();
}
// Runs of equal values in neighbouring lanes of a wave (valid lanes only): the lane that starts the run of this lane and, for
// a lane that starts a run, its length.  `worth`: the wave has at most half as many runs as points (else every lane is its own
// leader with length 1: sparse stretches of a cloud only pay for the test).
__device__ __forceinline__ void gh_runs(int v, bool valid, int &leader, int &len, bool &worth) {
  const int lane = (int)(threadIdx.x & 63), prev = wave_shift_up1(v, v);   // (lane 0's own value: its test below never looks at it)
  const unsigned long long mv = __ballot(valid), pv = mv << 1;
  const unsigned long long ml = __ballot(valid && (lane == 0 || !((pv >> lane) & 1ull) || prev != v));
  worth = 2 * __popcll(ml) <= __popcll(mv);
  leader = lane; len = 1;
  if (worth) {
    const unsigned long long below = ml & (lanemask_lt() | (1ull << lane));
    if (below) leader = 63 - __clzll((long long)below);
    const unsigned long long stop = (ml | ~mv) & (lane == 63 ? 0ull : ~((2ull << lane) - 1ull));
    len = (stop ? __ffsll((long long)stop) - 1 : 64) - lane;
  }
}
// The grid build over MANY workgroups per stream (round 2: one 1024-thread workgroup per stream swept all its points twice — 108 µs for
// the 57 000-point stream of the bench batch, 664 µs for the 420 000-point streams of agg10).  The points of a stream are cut into chunks
// of GC_CHUNK consecutive points; a chunk holds at most GC_CHUNK distinct cells, so its LDS table of GC_H slots can never overflow:
//   k_gridcount  (gc_P workgroups per stream, chunk after chunk): counts the points of every cell of the chunk in an LDS hash table and
//                writes the chunk's list of (cell key, count) and, per point, its entry in that list;
//   k_gridhash   (one workgroup per stream): merges the chunk lists — a few hundred entries per chunk instead of thousands of points — into
//                the stream's cell table, orders the cells, lays out the ranges and hands every chunk entry (cell id, first position);
//   k_gridplace  (as k_gridcount): every point draws its position from its chunk entry's LDS cursor and moves there.
// Points of one cell end up grouped by chunk and in arbitrary order inside a chunk's piece: every consumer tests existence, takes min / max,
// counts or adds exact integers.
#ifndef GC_HBITS
#define GC_CHUNK MOR_GC_CHUNK    // (6144-point chunks in 1024-thread workgroups with 64 KB of LDS took 20 µs alone and 110 µs in the pipeline: they waited for a CU with that much room)
#define GC_HBITS 12
#define GC_T 256
#endif
#define GC_H (1 << GC_HBITS)
#define GC_U (GC_CHUNK / GC_T)
__global__ __launch_bounds__(GC_T) void k_gridcount(MorDev d) {
  int s, j, gcp;
  if (!map_block_work(d, [&](int s_) { return ((int)d.info[s_].M + GC_CHUNK - 1) / GC_CHUNK; }, s, j, gcp)) return;   // work: the stream's chunks
  const int M = d.info[s].M, nch = (M + GC_CHUNK - 1) / GC_CHUNK, tid = threadIdx.x, lane = tid & 63;
  const size_t so = (size_t)s * d.Nmax;
  const int *pkey = d.pkey + so; int *pent = d.pslot + so;
  __shared__ int l_key[GC_H], l_cnt[GC_H], l_sh[GC_T / 64 + 1];
  constexpr unsigned hshift = 32 - GC_HBITS, mask = GC_H - 1; static_assert(GC_H >= GC_CHUNK + GC_CHUNK / 2, "a chunk's table cannot overflow");
  for (int c = j; c < nch; c += gcp) {
    for (int i = tid; i < GC_H; i += GC_T) { l_key[i] = 0; l_cnt[i] = 0; }
    __syncthreads();
    const int i0 = c * GC_CHUNK, i1 = min(i0 + GC_CHUNK, M);
    int key[GC_U], sl[GC_U];
#pragma unroll
    for (int u = 0; u < GC_U; ++u) { const int i = i0 + u * GC_T + tid; key[u] = i < i1 ? ld_stream(&pkey[i]) : -1; }
#pragma unroll
    for (int u = 0; u < GC_U; ++u) {
      // points arrive in scan order: neighbouring lanes often hold the same cell (a wall next to the sensor: all 64) — the first lane of a
      // run of equal keys counts the whole run with one LDS atomic (LDS atomics on one address serialise lane by lane)
      const bool valid = key[u] >= 0;
      int run_leader = lane, runlen = 1; bool worth = false;
      gh_runs(key[u], valid, run_leader, runlen, worth);
      unsigned h = hash_slot(max(key[u], 0), hshift);
      if (valid && run_leader == lane) {
        const int want = key[u] + 1;
        for (;;) {   // (cannot overflow: ≤ GC_CHUNK distinct keys in GC_H slots)
          int k = l_key[h];
          if (k == 0) { k = atomicCAS(&l_key[h], 0, want); if (k == 0) k = want; }
          if (k == want) break;
          h = (h + 1) & mask;
        }
        atomicAdd(&l_cnt[h], runlen);
      }
      if (worth) h = (unsigned)__shfl((int)h, run_leader, 64);
      sl[u] = (int)h;
    }
    __syncthreads();
    // the claimed slots as a list (any order): every thread looks at GC_H / GC_T consecutive slots
    int mine = 0;
#pragma unroll
    for (int q = 0; q < GC_H / GC_T; ++q) mine += l_key[tid * (GC_H / GC_T) + q] != 0;
    int total; int base = block_excl_scan_n<GC_T>(mine, l_sh, &total);
    int2 *list = d.gc_list + so + (size_t)c * GC_CHUNK;
#pragma unroll
    for (int q = 0; q < GC_H / GC_T; ++q) {
      const int h = tid * (GC_H / GC_T) + q, k = l_key[h];
      if (k != 0) { list[base] = make_int2(k - 1, l_cnt[h]); l_cnt[h] = base; ++base; }   // the slot now names its entry
    }
    if (tid == 0) d.gc_n[(size_t)s * d.gc_chunks + c] = total;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < GC_U; ++u) { const int i = i0 + u * GC_T + tid; if (i < i1) pent[i] = l_cnt[sl[u]]; }
    __syncthreads();
  }
}
__global__ __launch_bounds__(GC_T) void k_gridplace(MorDev d) {
  int s, j, gcp;
  if (!map_block_work(d, [&](int s_) { return ((int)d.info[s_].M + GC_CHUNK - 1) / GC_CHUNK; }, s, j, gcp)) return;   // work: the stream's chunks
  const int M = d.info[s].M, nch = (M + GC_CHUNK - 1) / GC_CHUNK, tid = threadIdx.x, lane = tid & 63;
  const size_t so = (size_t)s * d.Nmax;
  const int *pent = d.pslot + so; const float4 *cloud = d.cloud + so; float4 *sorted = d.sorted + so; int *scell = d.scell + so;
  const int tabsel = d.gc_tabsel[s]; const int2 *gtab = d.gc_tab + (size_t)s * 16384; const int *gkey = d.gh_key + (size_t)s * d.Hcell, *gval = d.gh_val + (size_t)s * d.Hcell;
  __shared__ int l_cell[GC_CHUNK], l_cur[GC_CHUNK];
  for (int c = j; c < nch; c += gcp) {
    const int ne = d.gc_n[(size_t)s * d.gc_chunks + c];
    const int2 *ent = d.gc_ent + so + (size_t)c * GC_CHUNK;
    for (int e = tid; e < ne; e += GC_T) {   // (slot, offset in the cell) → (compact cell id, first position of this chunk's piece of the cell)
      const int2 v = ent[e];
      int id1, first;
      if (tabsel) { const int2 tv = gtab[v.x]; id1 = tv.x; first = tv.y; } else { id1 = gkey[v.x]; first = gval[v.x]; }
      l_cell[e] = id1 - 1; l_cur[e] = first + v.y;
    }
    __syncthreads();
    const int i0 = c * GC_CHUNK, i1 = min(i0 + GC_CHUNK, M);
    int en[GC_U]; float4 q[GC_U];
#pragma unroll
    for (int u = 0; u < GC_U; ++u) { const int i = i0 + u * GC_T + tid; en[u] = i < i1 ? ld_stream(&pent[i]) : -1; q[u] = ld_stream(&cloud[min(i, max(M - 1, 0))]); }
#pragma unroll
    for (int u = 0; u < GC_U; ++u) {
      const bool valid = en[u] >= 0;
      int run_leader = lane, runlen = 1; bool worth = false;
      gh_runs(en[u], valid, run_leader, runlen, worth);
      int base = 0;
      if (valid && run_leader == lane) base = atomicAdd(&l_cur[en[u]], runlen);   // one cursor atomic per run of points of one cell
      if (worth) base = __shfl(base, run_leader, 64);
      if (valid) {
        const int i = i0 + u * GC_T + tid, pos = base + (lane - run_leader);
        q[u].w = __int_as_float(i);
        sorted[pos] = q[u]; scell[pos] = l_cell[en[u]];
        if (d.label_prefill) d.pcid[so + i] = -1;   // label of the cloud point until k_clusters says otherwise: written here in input order (coalesced) — k_clusters then stores the labels of CLUSTERED points only (its stores are scattered: the index travels with the point), a seventh of the points in the voxel ground variant's pass B
      }
    }
    __syncthreads();
  }
}
// TL / RL / CL: hash table / row table / per-cell lists in LDS (else global memory).  Returns false when the table
// overflowed (nothing published yet: the caller re-runs with a bigger table).  `cells` lists the claimed slots in
// discovery order — every per-cell phase walks it (a few entries per thread) instead of the whole table; `rowlist`
// first holds the x of the cells of every row, then (same memory) the point counts in compact-id order.
#ifndef GH_SW
#define GH_SW 8   // chunk entries per thread and round trip of the sweep (round 6: 4 → 8, + 0.5 … 1 % on the street scenes and the million-point clouds; 16: no further gain)
#endif
template <bool TL, bool RL, bool CL> __device__ __forceinline__ bool gh_run(const MorDev &d, const MorGrid &G, int s, int M, int *tkey, int *tval, int H, int cell_cap, int *rows, int *cells, int *rowlist, int *l_misc, int *l_sh, int *l_bits) {
  const size_t so = (size_t)s * d.Nmax;
  int *cstart = d.cstart + (size_t)s * (d.Nmax + 1), *ckey = d.ckey + so;
  const int nrows = G.nrows, nx = G.nx, tid = threadIdx.x;
  const int nch = (M + GC_CHUNK - 1) / GC_CHUNK;
  const int2 *clist = d.gc_list + so; int2 *cent = d.gc_ent + so; const int *cn = d.gc_n + (size_t)s * d.gc_chunks;
  int hbits = 0; while ((1 << hbits) < H) ++hbits;
  const unsigned hshift = 32 - hbits, mask = (unsigned)H - 1u;
  unsigned short *rs16 = d.rs16 + (size_t)s * d.rs16_stride, *cx16 = d.cx16 + (size_t)s * d.cx16_stride;   // 16-bit copies of the row table and the cells' x for the scoring tiers (cidx_load)
  for (int i = tid; i < H; i += GH_T) { gh_st<TL>(tkey + i, 0); gh_st<TL>(tval + i, 0); }
  if (tid == 0) { l_misc[0] = 0; l_misc[1] = 0; }
  __syncthreads();
  // ---- sweep over the chunks' lists (k_gridcount: the distinct cells of every chunk of GC_CHUNK points with their point counts): every
  //      entry finds (or claims) the slot of its cell and reserves its chunk's piece of the cell's range; (slot, offset in the cell) kept
  // (All entries of all chunks as ONE index space dealt over the 1024 threads — chunk after chunk with the whole workgroup idles most threads
  //  on the short lists of big clouds, a wave per chunk serialises the long lists of small ones: 57 → 9.5 µs and back to 35 in between.)
  int *cpre = l_bits;   // exclusive prefix of the chunks' entry counts (≤ 1024 chunks: 6 M points; beyond: chunk after chunk)
  if (nch <= (GH_T / 64) * 64) {
    const int mine = tid < nch ? cn[tid] : 0;
    int total; const int ex = block_excl_scan_n<GH_T>(mine, l_sh, &total);
    __syncthreads();
    cpre[tid] = ex;
    __syncthreads();
    // (GH_SW entries per thread and round trip: the workgroup has a CU to itself — sixteen waves are all that hides the latency of its loads —, and an entry per round trip left the
    //  sweep of the million-point clouds' 60 000 entries at 60 exposed trips)
    for (int g0 = tid; g0 < total; g0 += GH_SW * GH_T) {
      if (gh_ld<true>(&l_misc[1])) break;
      size_t at[GH_SW]; int2 kc[GH_SW];
#pragma unroll
      for (int u = 0; u < GH_SW; ++u) {
        const int g = min(g0 + u * GH_T, total - 1);
        int lo = 0, hi = nch - 1;   // last chunk whose prefix ≤ g
        while (lo < hi) { const int m = (lo + hi + 1) >> 1; if (cpre[m] <= g) lo = m; else hi = m - 1; }
        at[u] = (size_t)lo * GC_CHUNK + (g - cpre[lo]);
        kc[u] = clist[at[u]];
      }
#pragma unroll
      for (int u = 0; u < GH_SW; ++u) {
        if (g0 + u * GH_T >= total) break;
        const int want = kc[u].x + 1; unsigned h = hash_slot(kc[u].x, hshift); bool ok = false;
        for (int probes = 0; probes < H; ++probes) {
          int k = gh_ld<TL>(tkey + h);
          if (k == 0) {
            k = atomicCAS(tkey + h, 0, want);
            if (k == 0) { k = want; const int n = atomicAdd(&l_misc[0], 1); if (n < cell_cap) gh_st<CL>(cells + n, (int)h); else gh_st<true>(&l_misc[1], 1); }
          }
          if (k == want) { ok = true; break; }
          h = (h + 1) & mask;
        }
        if (ok) cent[at[u]] = make_int2((int)h, atomicAdd(tval + h, kc[u].y)); else gh_st<true>(&l_misc[1], 1);
      }
    }
  } else
  for (int c = 0; c < nch; ++c) {
    const int ne = cn[c];
    if (gh_ld<true>(&l_misc[1])) break;
    for (int e = tid; e < ne; e += GH_T) {
      const int2 kc = clist[(size_t)c * GC_CHUNK + e];
      const int want = kc.x + 1; unsigned h = hash_slot(kc.x, hshift); bool ok = false;
      for (int probes = 0; probes < H; ++probes) {
        int k = gh_ld<TL>(tkey + h);
        if (k == 0) {
          k = atomicCAS(tkey + h, 0, want);
          if (k == 0) { k = want; const int n = atomicAdd(&l_misc[0], 1); if (n < cell_cap) gh_st<CL>(cells + n, (int)h); else gh_st<true>(&l_misc[1], 1); }
        }
        if (k == want) { ok = true; break; }
        h = (h + 1) & mask;
      }
      if (ok) cent[(size_t)c * GC_CHUNK + e] = make_int2((int)h, atomicAdd(tval + h, kc.y)); else gh_st<true>(&l_misc[1], 1);
    }
  }
  __syncthreads();
  if (l_misc[1]) { __syncthreads(); return false; }
  const int nocc = l_misc[0];
  // ---- cells per row → row table
  for (int r = tid; r <= nrows; r += GH_T) gh_st<RL>(rows + r, 0);
  __syncthreads();
  for (int e = tid; e < nocc; e += GH_T) { const int key = gh_ld<TL>(tkey + gh_ld<CL>(cells + e)) - 1; atomicAdd(rows + key / nx, 1); }
  __syncthreads();
  gh_scan<RL>(rows, nrows, l_sh);
  if (tid == 0) gh_st<RL>(rows + nrows, nocc);
  __syncthreads();
  if (RL) { int *grs = d.row_start + (size_t)s * (d.g.nrows + 1); for (int r = tid; r <= nrows; r += GH_T) { const int v = rows[r]; grs[r] = v; rs16[r] = (unsigned short)v; } }
  else if (d.use_hash) for (int r = tid; r <= nrows; r += GH_T) rs16[r] = (unsigned short)gh_ld<RL>(rows + r);   // (meaningful while nocc ≤ 65 535: cidx_load checks)
  slab_bounds<RL>(d, G, s, rows, nocc, l_sh);
  // ---- the x of the cells of every row, listed (unordered) behind the row's first id.  The LDS copy of the row table
  //      serves as the fill cursor itself (rows[r] becomes the END of row r; the table proper is in global memory by now);
  //      a row table that lives in global memory stays intact and a scratch copy is the cursor
  int *fill = rows;
  if (!RL) {
    fill = d.gh_rowfill + (size_t)s * (d.g.nrows + 1);
    for (int r = tid; r < nrows; r += GH_T) gh_st<false>(fill + r, gh_ld<false>(rows + r));
    __syncthreads();
  }
  for (int e = tid; e < nocc; e += GH_T) {
    const int sl = gh_ld<CL>(cells + e), key = gh_ld<TL>(tkey + sl) - 1, r = key / nx;
    gh_st<CL>(rowlist + atomicAdd(fill + r, 1), TL ? ((key - r * nx) << 16) | sl : key - r * nx);   // (LDS-table tiers: the slot travels with the x — slots < 65 536)
  }
  __syncthreads();
  // ---- compact id = first id of the row + cells of the row with a smaller x; slot → id.  Short rows: every cell counts the smaller x of
  //      its row.  Long rows (a façade along x: 300 cells — counting is quadratic, 95 of 165 µs of this kernel on the urban scenes): one wave
  //      per row sets a bit per occupied x (x < 2048: 64 words), a wave scan of the popcounts gives every cell its rank in two LDS reads.
  for (int e = tid; e < nocc; e += GH_T) {
    const int sl = gh_ld<CL>(cells + e), k = gh_ld<TL>(tkey + sl), key = k - 1, r = key / nx, x = key - r * nx;
    const int b = RL ? (r ? gh_ld<RL>(rows + r - 1) : 0) : gh_ld<RL>(rows + r), e2 = RL ? gh_ld<RL>(rows + r) : gh_ld<RL>(rows + r + 1);
    if (TL && e2 - b > GH_SHORT) continue;
    int c = b;
    if (CL) { for (int q = b; q < e2; ++q) c += (TL ? gh_ld<CL>(rowlist + q) >> 16 : gh_ld<CL>(rowlist + q)) < x; }
    else {   // lists in global memory: eight independent loads per round trip
      for (int q = b; q < e2; q += 8) {
        int v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = gh_ld<CL>(rowlist + min(q + u, e2 - 1));
#pragma unroll
        for (int u = 0; u < 8; ++u) c += (q + u < e2) && (TL ? v[u] >> 16 : v[u]) < x;
      }
    }
    ckey[c] = key; cx16[c] = (unsigned short)x;
    gh_st<TL>(tkey + sl, c + 1);
  }
  if (TL) {
    __syncthreads();   // (the loop above has read every cell's key from its slot; the one below overwrites the slots of the long rows' cells)
    const int w = tid >> 6, lane = tid & 63;
    unsigned *bits = reinterpret_cast<unsigned *>(l_bits) + w * 64;
    for (int r0 = w * 64; r0 < nrows; r0 += (GH_T / 64) * 64) {
      const int r = r0 + lane;
      int b = 0, e2 = 0;
      if (r < nrows) { b = RL ? (r ? gh_ld<RL>(rows + r - 1) : 0) : gh_ld<RL>(rows + r); e2 = RL ? gh_ld<RL>(rows + r) : gh_ld<RL>(rows + r + 1); }
      unsigned long long m = __ballot(e2 - b > GH_SHORT);
      while (m) {
        const int l = __ffsll((long long)m) - 1; m &= m - 1;
        const int rb = __shfl(b, l, 64), re = __shfl(e2, l, 64), rr = r0 + l;
        // (the lanes of this wave hand bits to each other through LDS: a workgroup-scope fence between the steps makes the wave wait for its own LDS operations)
        bits[lane] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        for (int q = rb + lane; q < re; q += 64) { const int x = gh_ld<CL>(rowlist + q) >> 16; atomicOr(&bits[x >> 5], 1u << (x & 31)); }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        const int pc = __popc(bits[lane]), ex = wave_incl_scan(pc) - pc;   // occupied x below word `lane`
        for (int q0 = rb; q0 < re; q0 += 64) {   // (all lanes go through the shuffle: the word's prefix lives in lane x / 32)
          const int q = q0 + lane; const bool valid = q < re;
          const int v = valid ? gh_ld<CL>(rowlist + q) : 0, x = v >> 16, sl = v & 0xffff;
          const int c = rb + __shfl(ex, x >> 5, 64) + __popc(bits[x >> 5] & ((1u << (x & 31)) - 1u));
          if (valid) { ckey[c] = rr * nx + x; cx16[c] = (unsigned short)x; gh_st<TL>(tkey + sl, c + 1); }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      }
    }
  }
  __syncthreads();
  // ---- point counts in id order (same memory as the row lists) → first position of every cell
  int *cnt = rowlist;
  for (int e = tid; e < nocc; e += GH_T) { const int sl = gh_ld<CL>(cells + e); gh_st<CL>(cnt + gh_ld<TL>(tkey + sl) - 1, gh_ld<TL>(tval + sl)); }
  __syncthreads();
  gh_scan<CL>(cnt, nocc, l_sh);
  for (int c = tid; c < nocc; c += GH_T) {
    const int b0 = gh_ld<CL>(cnt + c), n = (c + 1 < nocc ? gh_ld<CL>(cnt + c + 1) : M) - b0;
    cstart[c] = b0;
    // records of the cells that span wave tiles of k_cellboxes start from the neutral element (their pieces are merged with atomics)
    if (b0 / CB_WTILE != (b0 + n - 1) / CB_WTILE) {
      d.cmeta[2 * (so + c)] = make_float4(FLT_MAX, FLT_MAX, FLT_MAX, 0.f); d.cmeta[2 * (so + c) + 1] = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, 0.f);
      d.cmin[so + c] = 0x7fffffff;
      MorCellSum z = {{0, 0, 0}, {0, 0, 0}}; d.csum[so + c] = z;
    }
  }
  if (tid == 0) { cstart[nocc] = M; d.info[s].n_occ = nocc; st_agent(&d.gh_hint[s], nocc); }   // (gh_hint is shared by all copies of the per-frame state: another lane's k_gridhash may read it meanwhile — agent-scope accesses; whichever tier it then starts at, the grid it builds is bit-identical)
  for (int e = tid; e < nocc; e += GH_T) { const int sl = gh_ld<CL>(cells + e); gh_st<TL>(tval + sl, gh_ld<CL>(cnt + gh_ld<TL>(tkey + sl) - 1)); }
  __syncthreads();
  // ---- the table itself (slot → compact cell id + 1, slot → first position of the cell) goes to global memory: k_gridplace turns its chunks'
  //      (slot, offset) entries into (cell, position) — no second sweep over the entries here, in the one workgroup the stream waits for
  if (TL) {
    int2 *gt = d.gc_tab + (size_t)s * 16384;
    for (int i = tid; i < H; i += GH_T) gt[i] = make_int2(gh_ld<TL>(tkey + i), gh_ld<TL>(tval + i));
  }
  if (tid == 0) d.gc_tabsel[s] = TL ? 1 : 0;   // 0: the table already lives in global memory (gh_key / gh_val)
  return true;
}
// LDS layouts of k_gridhash (ints): tier 0 — table of GH_H0 slots, row table, cell list and row lists all in LDS
// (≤ GH_C0 cells); tier 1 — table of GH_H slots and the row table in LDS, the per-cell lists in global scratch (≤ 3/4·GH_H
// cells); tier 2 — everything in global memory.  A stream starts at the tier the previous frame's cell counts suggest
// (d.gh_tier) and moves up when its table overflows.
#define GH_H0 8192
#define GH_C0 6144
#define GH_LDS_INTS (2 * GH_H + GH_ROWS + 1)
static_assert(2 * GH_H0 + GH_ROWS + 1 + 2 * GH_C0 <= GH_LDS_INTS, "tier-0 layout must fit the tier-1 arena");
#define GH_RUN(TL_, RL_, CL_, d_, ...) gh_run<TL_, RL_, CL_>(d_, G, __VA_ARGS__)
__global__ __launch_bounds__(GH_T) void k_gridhash(MorDev d) {
  const int s = blockIdx.x + d.s0, M = d.info[s].M;
  const MorGrid G = stream_grid(d, s);   // the stream's clustering grid (voxel ground variant: its own number of z layers)
  __shared__ int l_mem[GH_LDS_INTS], l_misc[4], l_sh[48], l_bits[(GH_T / 64) * 64];
  const bool rows_lds = G.nrows <= GH_ROWS;
  int *grows = d.row_start + (size_t)s * (d.g.nrows + 1);
  const size_t so = (size_t)s * d.Nmax;
  int *g_cells = d.gh_cells + so, *g_rowlist = d.gh_rowlist + so;
  bool done = false;
  // the tier the stream starts with: by its own cell count of the latest build (+ 1/16; the first frame starts small and moves up).  The host's estimate for the whole
  // batch — 5/4 of the largest stream — put every stream of the bench batch (≤ 5 400 cells) into tier 1 and the voxel ground variant's (10 300) into tier 2: −2.4 % / −4.5 %.
  // ONE read of the hint for the whole workgroup, handed round through LDS: the word is shared by all frames in flight, and another lane's k_gridhash of this very stream may
  // store a new count between the loads of two waves — with a load per thread the waves of a workgroup could start at DIFFERENT tiers (a street scene's 5 800 cells sit on the
  // boundary of the first two), run different instantiations against each other's barriers and leave cursors of a half-built row table to the stores of the fill loop: the
  // "Memory access fault by GPU" that one run in ten of 300 asynchronous steps of the street scenes died of since round 4 (found in round 6 with rocgdb: four waves in the tier-1
  // fill loop, twelve at a barrier of the tier-0 code; exp/fault_gdb.sh).
  if (threadIdx.x == 0) {
    int t_ = d.gh_tier;
    if (t_ < 0) { const int h = ld_agent(&d.gh_hint[s]); const long long need = (long long)h + h / 16; t_ = need > min(GH_C0, min(GH_H0, d.Hcell) / 4 * 3) ? (need > min(GH_H, d.Hcell) / 4 * 3 ? 2 : 1) : 0; }
    l_misc[3] = t_;
  }
  __syncthreads();
  const int tier = l_misc[3];
  if (tier <= 0) {
    const int H = min(GH_H0, d.Hcell);
    int *rows = l_mem + 2 * GH_H0, *cells = rows + GH_ROWS + 1, *rl = cells + GH_C0;
    if (rows_lds) done = GH_RUN(true, true, true, d, s, M, l_mem, l_mem + H, H, min(GH_C0, H / 4 * 3), rows, cells, rl, l_misc, l_sh, l_bits);
    else done = GH_RUN(true, false, true, d, s, M, l_mem, l_mem + H, H, min(GH_C0, H / 4 * 3), grows, cells, rl, l_misc, l_sh, l_bits);
  }
  if (!done && tier <= 1) {
    const int H = min(GH_H, d.Hcell);
    if (rows_lds) done = GH_RUN(true, true, false, d, s, M, l_mem, l_mem + H, H, H / 4 * 3, l_mem + 2 * GH_H, g_cells, g_rowlist, l_misc, l_sh, l_bits);
    else done = GH_RUN(true, false, false, d, s, M, l_mem, l_mem + H, H, H / 4 * 3, grows, g_cells, g_rowlist, l_misc, l_sh, l_bits);
  }
  if (!done) {   // table in global memory, sized for the cloud (cells ≤ M ≤ H/2)
    int H = 1024; while (H < 2 * M && H < d.Hcell) H <<= 1;
    GH_RUN(false, false, false, d, s, M, d.gh_key + (size_t)s * d.Hcell, d.gh_val + (size_t)s * d.Hcell, H, H, grows, g_cells, g_rowlist, l_misc, l_sh, l_bits);
  }
}
// ---- per-cell accumulators of the streaming cell pass (k_cellboxes): point box, smallest cloud index, exact coordinate sums
struct CellAcc { float lx, ly, lz, hx, hy, hz; int mi; long long a[3], b[3]; };
__device__ __forceinline__ void fx_split(float x, long long &a, long long &b) {   // x = a·2^-24 + b·2^-56 (MorCellSum); every step is exact for |x| ≥ 2^-32 (below: truncated at 2^-56)
  const double xd = (double)x, fa = floor(xd * 16777216.0);
  a = (long long)fa;
  b = (long long)((xd - fa * (1.0 / 16777216.0)) * 72057594037927936.0);
}
__device__ __forceinline__ double fx_value(long long a, long long b) { return (double)a * (1.0 / 16777216.0) + (double)b * (1.0 / 72057594037927936.0); }
__device__ __forceinline__ void acc_clear(CellAcc &r) { r.lx = r.ly = r.lz = FLT_MAX; r.hx = r.hy = r.hz = -FLT_MAX; r.mi = 0x7fffffff; r.a[0] = r.a[1] = r.a[2] = 0; r.b[0] = r.b[1] = r.b[2] = 0; }
__device__ __forceinline__ void acc_point(CellAcc &r, const float4 &p) {
  r.lx = fminf(r.lx, p.x); r.ly = fminf(r.ly, p.y); r.lz = fminf(r.lz, p.z); r.hx = fmaxf(r.hx, p.x); r.hy = fmaxf(r.hy, p.y); r.hz = fmaxf(r.hz, p.z);
  r.mi = min(r.mi, __float_as_int(p.w));
  long long a, b;
  fx_split(p.x, a, b); r.a[0] += a; r.b[0] += b; fx_split(p.y, a, b); r.a[1] += a; r.b[1] += b; fx_split(p.z, a, b); r.a[2] += a; r.b[2] += b;
}
__device__ __forceinline__ void acc_merge(CellAcc &r, const CellAcc &o) {
  r.lx = fminf(r.lx, o.lx); r.ly = fminf(r.ly, o.ly); r.lz = fminf(r.lz, o.lz); r.hx = fmaxf(r.hx, o.hx); r.hy = fmaxf(r.hy, o.hy); r.hz = fmaxf(r.hz, o.hz);
  r.mi = min(r.mi, o.mi);
#pragma unroll
  for (int k = 0; k < 3; ++k) { r.a[k] += o.a[k]; r.b[k] += o.b[k]; }
}
// an accumulator moved between lanes by DPP (nineteen v_mov_b32_dpp; lanes the control does not reach keep their own value — the caller's condition never merges those)
template <int CTRL, int ROWS, class T> __device__ __forceinline__ T dpp_val(const T &v) {
  const WaveWords<T> a = to_words(v); WaveWords<T> b;
#pragma unroll
  for (int i = 0; i < (int)(sizeof(T) / 4); ++i) b.w[i] = dpp_mov<CTRL, ROWS>(a.w[i], a.w[i]);
  return from_words<T>(b);
}
template <int CTRL, int ROWS = 0xF> __device__ __forceinline__ CellAcc acc_dpp(const CellAcc &r) {
  CellAcc t;
  t.lx = dpp_val<CTRL, ROWS>(r.lx); t.ly = dpp_val<CTRL, ROWS>(r.ly); t.lz = dpp_val<CTRL, ROWS>(r.lz); t.hx = dpp_val<CTRL, ROWS>(r.hx); t.hy = dpp_val<CTRL, ROWS>(r.hy); t.hz = dpp_val<CTRL, ROWS>(r.hz);
  t.mi = dpp_val<CTRL, ROWS>(r.mi);
#pragma unroll
  for (int k = 0; k < 3; ++k) { t.a[k] = dpp_val<CTRL, ROWS>(r.a[k]); t.b[k] = dpp_val<CTRL, ROWS>(r.b[k]); }
  return t;
}
// Segmented inclusive scan of the lanes' accumulators: lane l ends with the merge over lanes [hl, l], hl = the head of its segment (a lane ≤ l).  Four row_shr steps inside the
// 16-lane rows — lane l then covers [max(hl, start of its row), l] — then lane 15 of rows 0 / 2 into rows 1 / 3 (row_bcast15) and lane 31 into rows 2 and 3 (row_bcast31) for the
// segments that began in an earlier row.  Six steps as the Kogge–Stone form over ds_bpermute had, but VALU moves: no LDS instruction, no lgkmcnt wait.
__device__ __forceinline__ void acc_segmented_scan(CellAcc &S, int hl, int lane) {
#define MOR_SEG_STEP(CTRL, ROWS, COND) { const CellAcc t2 = acc_dpp<CTRL, ROWS>(S); if (COND) acc_merge(S, t2); }
  const int li = lane & 15;
  MOR_SEG_STEP(0x111, 0xF, li >= 1 && lane - 1 >= hl)
  MOR_SEG_STEP(0x112, 0xF, li >= 2 && lane - 2 >= hl)
  MOR_SEG_STEP(0x114, 0xF, li >= 4 && lane - 4 >= hl)
  MOR_SEG_STEP(0x118, 0xF, li >= 8 && lane - 8 >= hl)
  MOR_SEG_STEP(0x142, 0xA, (lane & 16) && hl < (lane & ~15))
  MOR_SEG_STEP(0x143, 0xC, lane >= 32 && hl < 32)
#undef MOR_SEG_STEP
}
// float min / max through integer atomics (no NaNs here; −0 is folded into +0 first)
__device__ __forceinline__ void atomic_fmin(float *p, float v) { v += 0.f; if (v >= 0.f) atomicMin((int *)p, __float_as_int(v)); else atomicMax((unsigned *)p, __float_as_uint(v)); }
__device__ __forceinline__ void atomic_fmax(float *p, float v) { v += 0.f; if (v >= 0.f) atomicMax((int *)p, __float_as_int(v)); else atomicMin((unsigned *)p, __float_as_uint(v)); }
// the record of cell c: alone (the cell lies inside one wave tile) or merged into what other waves deliver (k_gridhash
// initialised the records of the cells that span wave tiles)
__device__ __forceinline__ void acc_emit(const MorDev &d, size_t so, int c, const CellAcc &r, bool shared) {
  float *lo = reinterpret_cast<float *>(&d.cmeta[2 * (so + c)]), *hi = lo + 4;
  MorCellSum *cs = d.csum + so + c;
  if (!shared) {
    d.cmeta[2 * (so + c)] = make_float4(r.lx, r.ly, r.lz, 0.f); d.cmeta[2 * (so + c) + 1] = make_float4(r.hx, r.hy, r.hz, 0.f);
    d.cmin[so + c] = r.mi;
#pragma unroll
    for (int k = 0; k < 3; ++k) { cs->a[k] = r.a[k]; cs->b[k] = r.b[k]; }
  } else {
    atomic_fmin(lo, r.lx); atomic_fmin(lo + 1, r.ly); atomic_fmin(lo + 2, r.lz); atomic_fmax(hi, r.hx); atomic_fmax(hi + 1, r.hy); atomic_fmax(hi + 2, r.hz);
    atomicMin(&d.cmin[so + c], r.mi);
#pragma unroll
    for (int k = 0; k < 3; ++k) { atomicAdd((unsigned long long *)&cs->a[k], (unsigned long long)r.a[k]); atomicAdd((unsigned long long *)&cs->b[k], (unsigned long long)r.b[k]); }
  }
}
