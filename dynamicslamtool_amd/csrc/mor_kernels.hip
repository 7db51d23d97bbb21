// mor_kernels.hip — hand-written HIP kernels (gfx950 / CDNA4, wave64) for the hot path
//   pushRawCloudAndPose (/root/reference/src/MovingObjectRemoval.cpp:516-611) and
//   filterCloud (:613-696), batched over B independent streams.
//
// All of it is HBM / L2-bound integer, compare and scatter work — no MFMA.  Design rules applied:
//   * 16-byte points (float4) on device, never PCL's 32-byte struct; every streaming pass reads
//     1-KiB rows per wave instruction (64 lanes × 16 B).
//   * order-preserving compactions use wave ballots + one LDS exchange per 2048-point tile.
//   * the PCL kd-tree is replaced by a uniform grid keyed by linear cell id (a collision-free
//     spatial hash): occupancy bitmap + popcount rank, points counting-sorted by cell so an x-run of
//     cells is one contiguous range of the sorted array.
//   * cell edge 0.57·r ⇒ each cell is a clique of the cluster graph, so Euclidean clustering is
//     connected components over occupied CELLS: one wave per cell, lanes look up the neighbour
//     cells in the bitmap, then test point pairs across two cells 64 at a time and stop at the
//     first pair with d² < r².  Concurrent union-find with min-index hooking (atomicCAS on roots
//     only); cluster identity/order comes from the smallest cloud index of each component, so the
//     result does not depend on the schedule.
//   * one workgroup→(stream, tile) map that keeps all tiles of a stream on one XCD (blocks b and
//     b+8 share an XCD), so a stream's grid, sorted points and forest stay in one 4-MiB L2.
//     Correctness never depends on that placement: forest loads/stores are agent-scope relaxed
//     atomics, hooks are device-scope CAS, and every cross-kernel hand-off is a kernel boundary.
//   * fp32 predicates are evaluated exactly as the CPU reference does (individually rounded
//     mul/add, no FMA contraction): the file is compiled with -ffp-contract=off.
#include "mor_device.h"
#include <cfloat>
#include <cstddef>
#include <cstdlib>

const char *const mor_kernel_names[MK_COUNT] = {   // in MorKernelId order: "k_" + name = the __global__ function
    "classify", "scatter", "split", "heads_count", "heads_scatter", "cellboxes", "rhist", "rscan", "rscatter",
    "score_fast", "score_nb", "score_pde", "vox_clear", "vox_insert", "vox_probe",
    "out", "g2_cov_mid", "g2_cov", "g2_cov_big", "g2_mode", "g2_mark", "track_push", "track_filter", "gridhash", "gridcount", "gridplace", "cg_slab", "cg_final", "clusters"};

#ifdef MOR_EXP_STAMPS
#define RS_T(v) const unsigned long long v = wall_clock64()
#define RS_ADD(i, x) atomicAdd(&d.dbg[(size_t)s * 16 + (i)], (unsigned long long)(x))
#define RS_MAX(i, x) atomicMax(&d.dbg[(size_t)s * 16 + (i)], (unsigned long long)(x))
#else
#define RS_T(v)
#define RS_ADD(i, x)
#define RS_MAX(i, x)
#endif
#ifdef MOR_EXP_STAMPS
#define ST2(w, i) do { __syncthreads(); if (threadIdx.x == 0) d.dbg2[(size_t)(w) * 16 + (i)] = wall_clock64(); } while (0)
#define ST2V(w, i, v) do { if (threadIdx.x == 0) d.dbg2[(size_t)(w) * 16 + (i)] = (unsigned long long)(v); } while (0)
#else
#define ST2(w, i)
#define ST2V(w, i, v)
#endif
// ------------------------------------------------------------------------------------ helpers
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }
__device__ __forceinline__ unsigned long long lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// (stream, tile) of this workgroup.  With B a multiple of 8, all tiles of stream s run on the
// XCD group s % 8 (workgroups are dealt round-robin over the 8 XCDs): L2 locality only.
__device__ __forceinline__ void map_block_local(int B, int tiles, int &s, int &t, int xcd) {
  int L = blockIdx.x;
  if ((B & 7) == 0 && xcd) { int x = L & 7, r = L >> 3; s = (r / tiles) * 8 + x; t = r % tiles; }
  else { s = L / tiles; t = L % tiles; }
}
// d.B streams of this launch start at stream d.s0 of the batch (stream groups run on their own HIP streams)
#define map_block(B_, tiles_, s_, t_) do { map_block_local((B_), (tiles_), (s_), (t_), d.xcd_map); (s_) += d.s0; } while (0)
// Voxel ground variant: the clustering grid is laid out for 64 m of z (the variant does not crop in z), but a stream's cloud spans a few
// metres: pass A publishes the number of z layers the stream needs (gnz) and every kernel working on the clustering grid of that stream
// uses it — keys, the (y,z) row table and the slab tables then fit LDS as in the crop variant.  Strides of per-stream tables keep the
// configured row count (d.g.nrows).  gnz null: the grid as configured.  (A copy, not a patch of the kernel argument: patching `d`
// makes the compiler keep the whole 2 KB argument in scratch memory.)
__device__ __forceinline__ MorGrid stream_grid(const MorDev &d, int s) {
  MorGrid g = d.g;
  if (d.gnz) { g.nz = d.gnz[s]; g.nrows = g.ny * g.nz; }
  return g;
}

__device__ __forceinline__ int wave_incl_scan(int v) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { int n = __shfl_up(v, o, 64); if (lane_id() >= o) v += n; }
  return v;
}
// Work-proportional share-out of a launch's workgroups over the streams.  The streams of one batch differ a lot (the non-ground cloud of a stream of
// the bench batch has 5 000 … 57 000 points, 1 300 … 5 400 occupied cells): with the same number of workgroups for every stream a launch ends with the
// workgroups of its biggest stream walking chunk after chunk while the others have long left — the lane waits for that tail.  Here every workgroup
// works out, from the streams' work counts (wf(s): chunks of work of stream s, read from what earlier kernels of the frame left on the device — exact,
// no host estimate), how many workgroups each stream gets — one, plus its share of the spare ones in proportion to its work — and which stream and
// which of that stream's workgroups it is itself.  Any share is correct (kernels stride over their stream's chunks by `g`); a workgroup beyond the
// sum of the shares returns false and leaves.  With B a multiple of 8 the streams of XCD group x (s % 8 == x) share out the workgroups with
// blockIdx % 8 == x among themselves, so a stream's workgroups still meet in one L2 (map_block_local).  Called by all lanes of every wave before
// any divergence (every wave works it out for itself: a handful of loads and two wave scans, no LDS, no barrier); the results are wave-uniform.
// EXACT: wf(s) IS the number of workgroups of stream s (the launch holds at least their sum: the slabs of the cell graph, whose number per stream an
// earlier kernel fixed within the launch's budget).
// SPREAD: the streams share ALL workgroups of the launch (no XCD groups): for work that is small and uneven across streams.
// nblk / bid: the workgroups that take part and this one's number among them (default: the whole launch) — a launch may hold several kinds of workgroups.
template <bool EXACT = false, bool SPREAD = false, class WF> __device__ __forceinline__ bool map_block_work(const MorDev &d, WF wf, int &s, int &t, int &g, int nblk = -1, int bid = -1) {
  if (nblk < 0) { nblk = (int)gridDim.x; bid = (int)blockIdx.x; }
  const int lane = lane_id();
  const bool x8 = !SPREAD && (d.B & 7) == 0 && d.xcd_map && (nblk & 7) == 0;
  const int ng = x8 ? d.B >> 3 : d.B, G = x8 ? nblk >> 3 : nblk, x = x8 ? (bid & 7) : 0, r = x8 ? (bid >> 3) : bid, stp = x8 ? 8 : 1;
  if (!EXACT && (!d.prop_map || G < ng)) {   // same share for every stream (MOR_PROP_MAP=0; or fewer workgroups than streams: then the plain map with what there is)
    const int per = max(G / max(ng, 1), 1);
    const int i = r / per; if (i >= ng) return false;
    s = x + stp * i + d.s0; t = r - i * per; g = per; return true;
  }
  long long W = 0;
  if (!EXACT) for (int i0 = 0; i0 < ng; i0 += 64) {
    long long w = i0 + lane < ng ? (long long)wf(x + stp * (i0 + lane) + d.s0) : 0ll;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) w += ((long long)__shfl_xor((int)(w >> 32), o, 64) << 32) | (unsigned)__shfl_xor((int)(unsigned)w, o, 64);
    W += w;
  }
  const long long spare = G - ng;
  int carry = 0;
  for (int i0 = 0; i0 < ng; i0 += 64) {
    const int i = i0 + lane;
    const int gi = i < ng ? (EXACT ? (int)wf(x + stp * i + d.s0) : 1 + (int)((long long)wf(x + stp * i + d.s0) * spare / (W > 0 ? W : 1ll))) : 0;
    const int incl = wave_incl_scan(gi);
    const unsigned long long m = __ballot(i < ng && r < carry + incl);
    if (m) {
      const int l = __ffsll((long long)m) - 1;
      g = __builtin_amdgcn_readfirstlane(__shfl(gi, l, 64));
      t = __builtin_amdgcn_readfirstlane(r - (carry + __shfl(incl, l, 64) - g));
      s = __builtin_amdgcn_readfirstlane(x + stp * (i0 + l) + d.s0);
      return true;
    }
    carry += __shfl(incl, 63, 64);
  }
  return false;
}
// exclusive scan over the 256 threads of a workgroup; *total = sum.  sh: ≥ 5 ints of LDS.
__device__ __forceinline__ int block_excl_scan(int v, int *sh, int *total) {
  int inc = wave_incl_scan(v);
  if (lane_id() == 63) sh[wave_id()] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < MOR_BT / 64; ++w) { int x = sh[w]; if (w < wave_id()) base += x; tot += x; }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

// Σ c[i·stride] over i < t (prefix) and over i < nt (total), by the whole workgroup.  The per-tile count tables are a few
// hundred ints, so every workgroup re-derives its own offset instead of waiting for a separate one-workgroup scan
// kernel (a launch of ≈ 10 µs in the middle of each stage).  sh: ≥ 8 ints of LDS.
__device__ __forceinline__ void wg_prefix_total(const int *c, int stride, int t, int nt, int *sh, int &prefix, int &total) {
  int p = 0, a = 0;
  for (int i = threadIdx.x; i < nt; i += MOR_BT) { const int v = c[(size_t)i * stride]; a += v; p += i < t ? v : 0; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { p += __shfl_xor(p, o, 64); a += __shfl_xor(a, o, 64); }
  __syncthreads();
  if (lane_id() == 0) { sh[wave_id()] = p; sh[4 + wave_id()] = a; }
  __syncthreads();
  prefix = sh[0] + sh[1] + sh[2] + sh[3]; total = sh[4] + sh[5] + sh[6] + sh[7];
  __syncthreads();
}

__device__ __forceinline__ int ld_agent(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Streaming accesses: data that is written once and not read again soon (the ground points — 90 % of a sweep, read again only when the caller fetches the cloud;
// the filtered cloud) or read exactly once (the incoming cloud, ca's cluster points in tier 1) goes past the caches with the non-temporal hint, so that it does not
// push the frames' small hot tables out of the 4-MB L2s: the gather kernels live on their L2 hit rate.
typedef float mor_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_stream(float4 *p, const float4 &v) { const mor_v4f w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, reinterpret_cast<mor_v4f *>(p)); }
__device__ __forceinline__ float4 ld_stream(const float4 *p) { const mor_v4f w = __builtin_nontemporal_load(reinterpret_cast<const mor_v4f *>(p)); return make_float4(w.x, w.y, w.z, w.w); }
__device__ __forceinline__ int ld_stream(const int *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void st_stream(int *p, int v) { __builtin_nontemporal_store(v, p); }

__device__ __forceinline__ void st_agent_f(float *p, float v) { __hip_atomic_store(reinterpret_cast<int *>(p), __float_as_int(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent_f(const float *p) { return __int_as_float(__hip_atomic_load(reinterpret_cast<const int *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
__device__ __forceinline__ void st_agent_f4(float4 *p, const float4 &v) { float *f = reinterpret_cast<float *>(p); st_agent_f(f, v.x); st_agent_f(f + 1, v.y); st_agent_f(f + 2, v.z); st_agent_f(f + 3, v.w); }
__device__ __forceinline__ float4 ld_agent_f4(const float4 *p) { const float *f = reinterpret_cast<const float *>(p); return make_float4(ld_agent_f(f), ld_agent_f(f + 1), ld_agent_f(f + 2), ld_agent_f(f + 3)); }

// Error flags: into the frame's info record (reset at the start of every frame) and into the stream's sticky error word,
// which the host reports and clears at its next wait — so an error of ANY frame of an asynchronous run is reported, and
// flags raised after k_decide has copied the info record to the host are too.
__device__ __forceinline__ void mor_raise(const MorDev &d, int s, unsigned bit) { atomicOr(&d.info[s].flags, bit); atomicOr(&d.err[s], bit); }
// last kernel of a push / filter: refresh the pinned mirror of the sticky error word (one thread per stream)
__device__ __forceinline__ void mor_publish_err(const MorDev &d, int s) { d.h_err[s] = __hip_atomic_load(&d.err[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// "Last workgroup of the stream": every workgroup of a stream's share of a launch calls this once, after its last store; it returns true
// in exactly one of them — the one that arrives last — and that one may then read what all the others handed over.  No workgroup waits
// for another, so no assumption about residency or dispatch order is needed.  What is handed over must be written with agent-scope
// atomics or agent-scope (write-through, `sc1`) stores and read by the last workgroup with agent-scope loads (ld_agent): then no
// release fence is needed — an agent-scope release writes back the whole L2 of the XCD, and thousands of workgroups doing that per
// launch cost 200 µs (measured: k_score_pde 65 → 275 µs).  Every storing wave drains its stores, workgroup barrier, then one lane
// takes the ticket (MI355X_MICROARCH.md, hand-offs with `sc1` loads in place of the acquire).  The ticket word is reset by the last
// arriver for the next frame that uses this copy of the per-frame arrays (tickets exist once per frame in flight).  l_flag: one int of LDS.
__device__ __forceinline__ bool stream_last_block(int *ticket, int n_blocks, int *l_flag) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const int t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *l_flag = t == n_blocks - 1;
    if (t == n_blocks - 1) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  return *l_flag != 0;
}
enum { TK_TRACK = 0, TK_CGFINAL = 1, TK_PAIRS = 2, TK_SPLIT = 3, TK_OUT = 4, TK_SLABCNT = 5, TK_MOVERS = 6, TK_COUNT = 8 };   // ticket words per stream

// fromPCLPointCloud2 (:523): named float32 fields of a blob record → (x,y,z,intensity)
__device__ __forceinline__ float ld_f32_bytes(const char *p) {   // a float32 field at any byte address
  const unsigned char *u = reinterpret_cast<const unsigned char *>(p);
  return __uint_as_float((unsigned)u[0] | ((unsigned)u[1] << 8) | ((unsigned)u[2] << 16) | ((unsigned)u[3] << 24));
}
__device__ __forceinline__ float4 load_point(const MorStreamArgs &a, uint32_t i) {
  if (a.step == 16 && a.off_x == 0 && a.off_y == 4 && a.off_z == 8 && a.off_i == 12 && (reinterpret_cast<uintptr_t>(a.data) & 15) == 0) {
    // the incoming cloud is read exactly once: a streaming (non-temporal) load, so that 123 MB per step do not push the frames' small hot tables out of the 4-MB L2s
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f w = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(a.data) + i);
    return make_float4(w.x, w.y, w.z, w.w);
  }
  const char *r = reinterpret_cast<const char *>(a.data) + (size_t)i * a.step;
  float4 p;
  const uint32_t oi = a.off_i == 0xFFFFFFFFu ? 0u : a.off_i;
  if (((a.step | a.off_x | a.off_y | a.off_z | oi) & 3u) == 0 && (reinterpret_cast<uintptr_t>(a.data) & 3) == 0) {
    p.x = *reinterpret_cast<const float *>(r + a.off_x);
    p.y = *reinterpret_cast<const float *>(r + a.off_y);
    p.z = *reinterpret_cast<const float *>(r + a.off_z);
    p.w = (a.off_i == 0xFFFFFFFFu) ? 0.0f : *reinterpret_cast<const float *>(r + a.off_i);
  } else {   // packed records such as the Velodyne driver's 22-byte PointXYZIRT: fromPCLPointCloud2 memcpy's the fields, so do we
    p.x = ld_f32_bytes(r + a.off_x); p.y = ld_f32_bytes(r + a.off_y); p.z = ld_f32_bytes(r + a.off_z);
    p.w = (a.off_i == 0xFFFFFFFFu) ? 0.0f : ld_f32_bytes(r + a.off_i);
  }
  return p;
}

// groundPlaneRemoval(x,y,z) (:62-88): 0 = dropped by the x/y PassThrough pair (or non-finite),
// 1 = removed by the CropBox (→ gp_indices), 2 = kept in `cloud`.
__device__ __forceinline__ int classify(const MorDev &d, float4 p) {
  bool fin = __builtin_isfinite(p.x) && __builtin_isfinite(p.y) && __builtin_isfinite(p.z);
  if (!fin || p.x < -d.trim_x || p.x > d.trim_x || p.y < -d.trim_y || p.y > d.trim_y) return 0;
  if (d.gmode == 1) return 2;   // voxel variant, pass A: only the x/y PassThrough pair (:94-102)
  return (p.z < d.gp_limit || p.z > d.trim_z) ? 1 : 2;
}
// number of input records of stream s and record i for the current pass (pass B of the voxel variant re-reads the
// trimmed cloud and splits it by the ground flag, :194-198)
__device__ __forceinline__ uint32_t pass_count(const MorDev &d, const MorStreamArgs &a, int s) { return d.gmode == 2 ? d.info[s].T : a.n; }
__device__ __forceinline__ int pass_item(const MorDev &d, const MorStreamArgs &a, int s, uint32_t i, float4 &p) {
  if (d.gmode == 2) { p = d.rawbuf[(size_t)s * d.Nmax + i]; return d.is_ground[(size_t)s * d.Nmax + i] == d.frame_no + 1 ? 1 : 2; }   // (ground flags carry the frame's tag: no clearing pass)
  p = load_point(a, i);
  return classify(d, p);
}
__device__ __forceinline__ int float_ordered(float f) { int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float ordered_float(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }
// grid cell of a point: clustering grid (clamped, monotone map) or the VoxelGrid lattice (absolute multiples of the leaf)
__device__ __forceinline__ void grid_cell(const MorGrid &g, float4 p, float zorg, int zbase, int &cx, int &cy, int &cz, bool &clamped) {
  if (g.mode == 1) {
    cx = (int)floorf(p.x * g.inv_cs) - g.ibx; cy = (int)floorf(p.y * g.inv_cs) - g.iby; cz = (int)floorf(p.z * g.inv_cs) - zbase;
  } else {
    cx = (int)floorf((p.x - g.ox) * g.inv_cs); cy = (int)floorf((p.y - g.oy) * g.inv_cs); cz = (int)floorf((p.z - zorg) * g.inv_cs);
  }
  clamped = cx < 0 || cy < 0 || cz < 0 || cx >= g.nx || cy >= g.ny || cz >= g.nz;
  cx = min(max(cx, 0), g.nx - 1); cy = min(max(cy, 0), g.ny - 1); cz = min(max(cz, 0), g.nz - 1);
}

// Linear cell key, y-major: (cy·nz + cz)·nx + cx.  A (y,z) ROW is nx consecutive keys; a y-SLICE (all rows of one y) is
// nz·nx consecutive keys, so a contiguous range of the key-sorted cells is a slab of space between two y planes — the
// unit the cell graph is split over (k_cg_slab).
__device__ __forceinline__ int grid_row(const MorGrid &g, int cy, int cz) { return cy * g.nz + cz; }
__device__ __forceinline__ int grid_key(const MorGrid &g, int cx, int cy, int cz) { return grid_row(g, cy, cz) * g.nx + cx; }
__device__ __forceinline__ int cell_axis(float v, float o, float inv, int n) {
  int c = (int)floorf((v - o) * inv);
  return c < 0 ? 0 : (c >= n ? n - 1 : c);
}
__device__ __forceinline__ int cell_axis_unclamped(float v, float o, float inv) { return (int)floorf((v - o) * inv); }
#ifndef ROW_BATCH
#define ROW_BATCH 8
#endif
// first index in [lo, lo+n) whose key is ≥ k0, by 8-ary search: every step fetches its seven pivots with independent
// loads, so a row of 500 cells costs three load latencies instead of the nine dependent ones of a binary search (the
// wave pays the latency of its longest row in every iteration of the hook passes)
__device__ __forceinline__ int cg_lower_bound8(const int *key, int lo, int n, int k0) {
  // (loads first, unconditionally, then the comparisons with `&`: written as `in range && key[…] < k0` every load sat in its own branch with its own wait — seven
  //  round trips one after the other per step, by the ISA)
  while (n > 8) {
    const int step = (n + 7) >> 3;
    int kv[7], c = 0;
#pragma unroll
    for (int j = 1; j < 8; ++j) kv[j - 1] = key[min(lo + j * step, lo + n - 1)];
#pragma unroll
    for (int j = 1; j < 8; ++j) c += (int)(j * step < n) & (int)(kv[j - 1] < k0);
    lo += c * step; n = min(step, n - c * step);
  }
  int kv[8], below = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) kv[i] = key[lo + min(i, n - 1)];
#pragma unroll
  for (int i = 0; i < 8; ++i) below += (int)(i < n) & (int)(kv[i] < k0);
  return lo + below;
}
// occupied cells with x in [x0,x1] of row (cy,cz) have the consecutive compact ids [lo, hi)
// (the cells [lo0, e) of one row, keys ascending from base = row · nx)
__device__ __forceinline__ void row_range(const int *ckey, int lo0, int e, int base, int x0, int x1, int &lo, int &hi) {
  lo = lo0;
  const int k0 = base + x0, k1 = base + x1;
  if (e - lo > ROW_BATCH) {   // long row (a wall along x; every row of the voxel ground variant's lattice): 8-ary search — three round trips for 512 cells where two binary searches took eighteen
    lo = cg_lower_bound8(ckey, lo, e - lo, k0);
    const int w = min(e - lo, x1 - x0 + 1);   // cells that can lie in [k0, k1]
    if (w <= 8) {
      int kv[8], within = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) kv[i] = ckey[min(lo + i, e - 1)];   // (the row is not empty)
#pragma unroll
      for (int i = 0; i < 8; ++i) within += (int)(i < w) & (int)(kv[i] <= k1);
      hi = lo + within;
    } else hi = cg_lower_bound8(ckey, lo, e - lo, k1 + 1);
    return;
  }
  // short row: fetch up to 8 keys with independent loads (one memory latency, not a chain of them)
  const int n = e - lo;
  int below = 0, within = 0;
  if (n > 0) {
    int kv[ROW_BATCH];
#if ROW_BATCH == 8
    __builtin_memcpy(kv, ckey + lo, 32);   // two 16-byte loads (dword-aligned addresses are fine for global loads); entries beyond the row are keys of later rows — masked below — and the arrays end ROW_BATCH entries behind the last cell (mor_batch_create)
#else
#pragma unroll
    for (int i = 0; i < ROW_BATCH; ++i) kv[i] = ckey[lo + min(i, n - 1)];
#endif
#pragma unroll
    for (int i = 0; i < ROW_BATCH; ++i) { const bool v = i < n; below += (int)v & (int)(kv[i] < k0); within += (int)v & (int)(kv[i] >= k0) & (int)(kv[i] <= k1); }
  }
  lo += below; hi = lo + within;
}
__device__ __forceinline__ void row_cells(const MorGrid &g, const int *ckey, const int *rs, int x0, int x1, int cy, int cz, int &lo, int &hi) {
  const int r = grid_row(g, cy, cz);
  row_range(ckey, rs[r], rs[r + 1], r * g.nx, x0, x1, lo, hi);
}
// compact id of cell (cx,cy,cz) or −1 when empty / outside
__device__ __forceinline__ int cell_lookup(const MorGrid &g, const int *ckey, const int *rs, int cx, int cy, int cz) {
  if ((unsigned)cx >= (unsigned)g.nx || (unsigned)cy >= (unsigned)g.ny || (unsigned)cz >= (unsigned)g.nz) return -1;
  int lo, hi; row_cells(g, ckey, rs, cx, cx, cy, cz, lo, hi);
  return lo < hi ? lo : -1;
}

// L2_Simple: ((dx·dx)+(dy·dy))+(dz·dz), each operation rounded (no contraction)
__device__ __forceinline__ float sqdist(float ax, float ay, float az, float bx, float by, float bz) {
  float dx = ax - bx, dy = ay - by, dz = az - bz;
  float r = dx * dx; r = r + dy * dy; r = r + dz * dz;
  return r;
}

// start of a frame: input size, error flags and cluster counts of the stream's info record (not in pass B of the voxel
// variant, which continues the frame).  Runs in the first kernel of the frame, before any kernel that raises a flag.
__device__ __forceinline__ void reset_frame_info(const MorDev &d, int s, uint32_t n_points) {
  if (d.gmode == 2) return;
  MorFrameInfo &f = d.info[s];
  f.N = n_points; f.flags = 0; f.n_pairs = 0; f.K = 0; f.C = 0; f.max_loc = 0; f.g2_exact = 0;
  d.tickets[(size_t)s * TK_COUNT + TK_SLABCNT] = 0;   // slabs handed out so far to the streams of this stream's XCD group (slab_bounds; the word of the group's first stream counts)
}
// ------------------------------------------------------------------------------------ G1: trim + ground split
// pass 1: per-tile counts of (non-ground, ground)
__global__ __launch_bounds__(MOR_BT) void k_classify(MorDev d) {
  int s, t0; map_block(d.B, d.split_g, s, t0);
  const MorStreamArgs a = d.args[s];
  const uint32_t n_in = pass_count(d, a, s);
  __shared__ int sh[8];
  for (int t = t0; t < d.tiles; t += d.split_g) {   // split_g workgroups per stream walk its tiles: enough loads in flight for the HBM without holding every wave slot of the GPU
  uint32_t base = (uint32_t)t * MOR_TILE + wave_id() * 512;
  int c_ng = 0, c_g = 0;
  float zlo = INFINITY, zhi = -INFINITY;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    uint32_t i = base + it * 64 + lane_id();
    float4 p; int cls = (i < n_in) ? pass_item(d, a, s, i, p) : 0;
    if (d.gmode == 1 && cls == 2) { zlo = fminf(zlo, p.z); zhi = fmaxf(zhi, p.z); }
    c_ng += __popcll(__ballot(cls == 2));
    c_g += __popcll(__ballot(cls == 1));
  }
  if (d.gmode == 1) {   // z extent of the trimmed cloud: the voxel variant does not crop in z
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { zlo = fminf(zlo, __shfl_xor(zlo, o, 64)); zhi = fmaxf(zhi, __shfl_xor(zhi, o, 64)); }
    if (lane_id() == 0 && zlo <= zhi) { atomicMin(&d.zmin_i[s], float_ordered(zlo)); atomicMax(&d.zmax_i[s], float_ordered(zhi)); }
  }
  if (lane_id() == 0) { sh[wave_id()] = c_ng; sh[4 + wave_id()] = c_g; }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (t == 0) reset_frame_info(d, s, a.n);
    int *o = d.tile_cnt + ((size_t)s * d.tiles_max + t) * 2;
    o[0] = sh[0] + sh[1] + sh[2] + sh[3];
    o[1] = sh[4] + sh[5] + sh[6] + sh[7];
  }
  __syncthreads();
  }
}

// Voxel ground variant, pass A: z layers of the VoxelGrid lattice this stream's trimmed cloud needs (cz = floor(z·inv) − floor(zmin·inv)), at most the configured
// number.  The lattice is laid out for 64 m of z, a sweep spans a few metres: with the stream's own layer count the voxel keys are 23 instead of 27 bits (three
// radix passes instead of four) and the (y,z) row table 12 000 instead of 161 000 rows.  From the ordered-int z range k_classify leaves (final at the kernel boundary).
__device__ __forceinline__ int voxel_layers(const MorDev &d, int s) {
  const int zl = d.zmin_i[s], zh = d.zmax_i[s];
  if (zl > zh) return 1;   // no trimmed point
  const int l = (int)floorf(ordered_float(zh) * d.g.inv_cs) - (int)floorf(ordered_float(zl) * d.g.inv_cs) + 1;
  return max(1, min(d.g.nz, l));
}
// radix passes (8-bit digits) the stream's voxel keys need: keys < nx·ny·(its layers)
__device__ __forceinline__ int voxel_passes_of(const MorDev &d, int s) {
  const long long cells = (long long)d.g.nx * d.g.ny * (d.gnz ? d.gnz[s] : d.g.nz);
  const int bits = cells > 1 ? 64 - __clzll(cells - 1) : 1;
  return (bits + 7) >> 3;
}
// T, M, G of the frame (and, for pass A of the voxel variant, the z origin of its grids)
__device__ __forceinline__ void publish_split(const MorDev &d, int s, int n_ng, int n_g) {
  MorFrameInfo &f = d.info[s];
  f.M = n_ng; f.G = n_g; f.T = n_ng + n_g;
  if (d.gmode == 2) { d.zmin_i[s] = 0x7fffffff; d.zmax_i[s] = (int)0x80000000; }   // pass B ends the frame's use of the z range: ready for the next frame on this copy (no memset launches)
  if (d.gmode == 1) {   // grids of the voxel variant hang on the lowest trimmed point
    float zmin = f.T ? ordered_float(d.zmin_i[s]) : 0.f;
    d.zorg[s] = zmin; d.zbase[s] = (int)floorf(zmin * d.gv.inv_cs);
    if (d.gnz_out) {   // z layers of the clustering grid this stream needs (stream_grid)
      const float zmax = f.T ? ordered_float(d.zmax_i[s]) : 0.f;
      d.gnz_out[s] = max(1, min(d.cg_nz, (int)floorf((zmax - zmin) * d.cg_inv_cs) + 2));
    }
    if (d.vnz_out) d.vnz_out[s] = voxel_layers(d, s);   // layers of the VoxelGrid lattice (the later kernels of pass A read it through stream_grid)
  }
}
// pass 2: order-preserving split into `cloud` / ground, cell histogram, forest init
__global__ __launch_bounds__(MOR_BT) void k_scatter(MorDev d) {
  int s, t0; map_block(d.B, d.split_g, s, t0);
  MorGrid G = d.gmode == 1 ? d.g : stream_grid(d, s);   // the stream's clustering grid (voxel ground variant: its own number of z layers); pass A: the VoxelGrid lattice, whose layers for this stream follow from the z range k_classify left (below)
  if (d.gmode == 1) { G.nz = voxel_layers(d, s); G.nrows = G.ny * G.nz; }
  const MorStreamArgs a = d.args[s];
  const uint32_t n_in = pass_count(d, a, s);
  __shared__ int sh[8];
  for (int t = t0; t < d.tiles; t += d.split_g) {
  uint32_t base = (uint32_t)t * MOR_TILE + wave_id() * 512;
  int r_ng = 0, r_g = 0; float zorg = d.zorg[s]; int zbase = d.zbase[s];
  {   // own offsets from the per-tile counts of k_classify (no scan launch in between); tile 0 publishes the totals
    const int *tc = d.tile_cnt + (size_t)s * d.tiles_max * 2; int tot_ng, tot_g;
    wg_prefix_total(tc, 2, t, d.tiles, sh, r_ng, tot_ng);
    wg_prefix_total(tc + 1, 2, t, d.tiles, sh, r_g, tot_g);
    if (d.gmode == 1) { zorg = (tot_ng + tot_g) ? ordered_float(d.zmin_i[s]) : 0.f; zbase = (int)floorf(zorg * d.gv.inv_cs); }
    if (t == 0 && threadIdx.x == 0) publish_split(d, s, tot_ng, tot_g);
  }
  if ((uint32_t)t * MOR_TILE >= n_in) break;
  float4 p[8]; int cls[8]; unsigned long long m_ng[8], m_g[8];
  int c_ng = 0, c_g = 0;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    uint32_t i = base + it * 64 + lane_id();
    cls[it] = 0;
    if (i < n_in) cls[it] = pass_item(d, a, s, i, p[it]);
    m_ng[it] = __ballot(cls[it] == 2); m_g[it] = __ballot(cls[it] == 1);
    c_ng += __popcll(m_ng[it]); c_g += __popcll(m_g[it]);
  }
  if (lane_id() == 0) { sh[wave_id()] = c_ng; sh[4 + wave_id()] = c_g; }
  __syncthreads();
  for (int w = 0; w < wave_id(); ++w) { r_ng += sh[w]; r_g += sh[4 + w]; }
  const size_t so = (size_t)s * d.Nmax;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    int k_ng = r_ng + __popcll(m_ng[it] & lanemask_lt());
    int k_g = r_g + __popcll(m_g[it] & lanemask_lt());
    if (cls[it] == 2) {
      int cx, cy, cz; bool clamped; grid_cell(G, p[it], zorg, zbase, cx, cy, cz, clamped);
      if (clamped && d.gmode != 0) mor_raise(d, s, 8u);   // z extent beyond the grid: cells would no longer be cliques / voxels
      d.cloud[so + k_ng] = p[it];
      d.pkey[so + k_ng] = grid_key(G, cx, cy, cz);
    } else if (cls[it] == 1) {
      st_stream(&d.ground[2 * so + d.Nmax + k_g], p[it]);   // final place in filterCloud's output: [kept cloud, right-aligned to slot Nmax | ground from slot Nmax]
    }
    if (lane_id() == 0 && base + it * 64 < n_in) { unsigned long long *cm = d.cls_mask + ((size_t)s * d.cls_rows + (base + it * 64) / 64) * 2; cm[0] = m_ng[it]; cm[1] = m_g[it]; }   // the classes of these 64 records (split_store)
    r_ng += __popcll(m_ng[it]); r_g += __popcll(m_g[it]);
  }
  __syncthreads();
  }
}

// Single-READ variant of k_classify + k_scatter (crop-box variant and pass B of the voxel variant).  sp_g workgroups per stream take the
// stream's tiles from a ticket counter (TK_SPLIT), in order of arrival.  A tile's output offsets are the counts of all earlier tiles: a
// workgroup carries the prefix of its previous tile along and adds the aggregates of the tiles in between, which the other workgroups
// publish — right after their loads have landed — in 64-bit descriptors tagged with the frame (no reset pass; polled and published with
// agent-scope accesses).  Tiles handed out by ticket make the look-back safe whatever the dispatcher does: every tile below a
// workgroup's own was taken by a workgroup that is already running, and the owner of the lowest unpublished tile never waits for
// anything unpublished, so somebody always makes progress.  (The first form of this kernel gave workgroup g the tiles g, g + sp_g, …:
// a tile then waits for tiles of workgroups with HIGHER numbers, which may not be resident yet, and with four frames' splits in flight
// the wave slots of an XCD can fill up with such waiters — 4 × (sp_g − 1) ≥ 128: it stalled at sp_g = 48 on the 1 M-point clouds.)
// Every tile goes through three steps — loads issued, counted (its aggregate published), look-back + stores — and a workgroup holds
// two tiles: the NEXT tile is counted and published BEFORE the workgroup waits for the descriptors of the current one, so nobody ever
// waits for a tile whose owner is itself waiting, and the current tile's look-back and stores overlap the ticket for the tile after
// (taken by one lane, handed round through LDS).  Publishing a tile only when its turn to be stored comes — the natural order — made
// this 3× slower (234 against 79 µs alone): a workgroup's second tile lies right behind its first, the next workgroup's first tile
// waits for it, and the stream's workgroups end up running one after the other.  (Round 2's form — one workgroup per tile, look-back
// over all earlier tiles — read, waited and stored in lock step: 91 µs against 44 µs without the look-back.)  Per workgroup two
// tickets at the start and one per tile it stores: 2·sp_g + nt in all, and whoever takes the last one clears the counter for the next
// frame that uses this copy of the per-frame arrays.  A peer that never shows up raises the "look-back stalled" flag after
// SPLIT_SPIN_LIMIT polls instead of hanging.  Measured alone: B = 64 × 120 000 points 86 µs (16 workgroups per stream; static tiles
// 79 µs, count + scatter passes 36 + 58 µs and one more read of the cloud), B = 32 × 1 M points 306 µs (static tiles 360 µs); the
// pipelined throughput of both workloads is that of the static form or better.
// Tiles of the split are 1024 records (MOR_SP_ROWS = 4 rows of 64 per wave), half the 2048 of the other streaming kernels: a workgroup then goes through seven
// tiles instead of four per 120 000-point cloud and holds 32 instead of 64 data registers (105 → 73 VGPRs: six workgroups per CU); the stream's workgroups fall
// out of step sooner, so loads, look-back waits and stores of different workgroups overlap better (204.8–207.3 → 210.7–211.2 k frame-pairs/s; 512-record tiles 207.7 k).
// Tried on top: THREE tiles per workgroup — two counted and published, the third landing, so that no tile is counted right behind its own loads — 208 k: the
// exposed load latency is not what the split waits for.
#define SPLIT_SPIN_LIMIT (1u << 22)
#define SP_ROWS MOR_SP_ROWS
#define SP_TILE (4 * SP_ROWS * 64)   // records per tile of the single-read split
#define SP_DESC_STRIDE(d) ((size_t)(d).tiles_max * (8 / SP_ROWS))
__device__ __forceinline__ unsigned long long ld_agent64(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// loads only (no use of the data here: the wait for them belongs to split_tile, a step later); cls carries the ground flag of pass B
template <bool PASSB> __device__ __forceinline__ void split_load_tile(const MorDev &d, const MorStreamArgs &a, int s, uint32_t n_in, int t, float4 (&p)[SP_ROWS], int (&cls)[SP_ROWS]) {
  const uint32_t base = (uint32_t)t * SP_TILE + wave_id() * (SP_ROWS * 64);
#pragma unroll
  for (int it = 0; it < SP_ROWS; ++it) {
    const uint32_t i = min(base + it * 64 + lane_id(), n_in - 1);   // (clamped: out-of-range lanes repeat the last record and are masked in split_tile)
    if (PASSB) { p[it] = d.rawbuf[(size_t)s * d.Nmax + i]; cls[it] = d.is_ground[(size_t)s * d.Nmax + i] == d.frame_no + 1; }
    else { p[it] = load_point(a, i); cls[it] = 0; }
  }
}
struct SplitMeta { int tng, tg, wng, wg; };   // a counted tile: its totals and this wave's offsets inside it
template <bool PASSB> __device__ __forceinline__ int split_class(const MorDev &d, uint32_t n_in, uint32_t i, const float4 &p, int cls) {
  return i < n_in ? (PASSB ? (cls ? 1 : 2) : classify(d, p)) : 0;
}
// stage 1 of a tile (its loads were issued a step earlier): counts, and the tile's aggregate goes out to the other workgroups
template <bool PASSB> __device__ __forceinline__ void split_count(const MorDev &d, int s, int t, uint32_t n_in, unsigned epoch, const float4 (&p)[SP_ROWS], const int (&cls)[SP_ROWS], int *sh, SplitMeta &m) {
  int c_ng = 0, c_g = 0;
  const uint32_t base = (uint32_t)t * SP_TILE + wave_id() * (SP_ROWS * 64) + lane_id();
#pragma unroll
  for (int it = 0; it < SP_ROWS; ++it) {
    const int c = split_class<PASSB>(d, n_in, base + it * 64, p[it], cls[it]);
    const unsigned long long m_ng = __ballot(c == 2), m_g = __ballot(c == 1);
    c_ng += __popcll(m_ng); c_g += __popcll(m_g);
    // gp_indices (:86) and the trimmed-cloud index of every cloud point are read-backs only: instead of 4 bytes per trimmed point the split leaves the
    // classes of every 64 records as two bit masks (16 bytes: cloud, ground; rows that hold records only — a tile reaches beyond the stream's slice of the
    // array); the host rebuilds the index lists from them when asked (mor_get_ground_indices, mor_get_labels)
    if (lane_id() == 0 && base + it * 64 < n_in) { unsigned long long *cm = d.cls_mask + ((size_t)s * d.cls_rows + (base + it * 64) / 64) * 2; cm[0] = m_ng; cm[1] = m_g; }
  }
  if (lane_id() == 0) { sh[wave_id()] = c_ng; sh[4 + wave_id()] = c_g; }
  __syncthreads();
  m.tng = sh[0] + sh[1] + sh[2] + sh[3]; m.tg = sh[4] + sh[5] + sh[6] + sh[7];
  m.wng = 0; m.wg = 0;
  for (int w = 0; w < wave_id(); ++w) { m.wng += sh[w]; m.wg += sh[4 + w]; }
  if (threadIdx.x == 0)
    __hip_atomic_store(d.split_desc + (size_t)s * SP_DESC_STRIDE(d) + t, ((unsigned long long)epoch << 32) | ((unsigned long long)(unsigned)m.tng << 16) | (unsigned long long)(unsigned)m.tg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// stage 2: look-back over the tiles between this workgroup's previous tile and this one, then the stores.
// tk_next (thread 0 only): the ticket this workgroup has just taken for a later tile — passed on to all threads through s_ex[2]
template <bool PASSB> __device__ __forceinline__ void split_store(const MorDev &d, const MorGrid &G, int s, int t, int nt, int t_prev, uint32_t n_in, unsigned epoch, const float4 (&p)[SP_ROWS], const int (&cls)[SP_ROWS],
                                            const SplitMeta &m, int &ex_ng, int &ex_g, int *s_ex, int tk_next) {
  if (wave_id() == 0) {
    const unsigned long long *desc = d.split_desc + (size_t)s * SP_DESC_STRIDE(d);
    const int lane = lane_id();
    int an = 0, ag = 0;
    for (int hi = t - 1; hi > t_prev; hi -= 64) {   // 64 at a time (normally about sp_g of them in all)
      const int u = hi - lane;
      if (u > t_prev) {
        unsigned spins = 0;
        for (;;) {
          const unsigned long long v = ld_agent64(&desc[u]);
#ifdef MOR_EXP_SPLITVAR
          if ((d.t1_budget >> 16) & 1) break;   // experiment (exp/split_var.py): no look-back wait — prefixes are wrong, only the duration is read
#endif
          if ((unsigned)(v >> 32) == epoch) { an += (int)((v >> 16) & 0xffffu); ag += (int)(v & 0xffffu); break; }
          if (++spins > SPLIT_SPIN_LIMIT) { mor_raise(d, s, 64u); break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { an += __shfl_xor(an, o, 64); ag += __shfl_xor(ag, o, 64); }
    if (lane == 0) {
      s_ex[0] = ex_ng + an; s_ex[1] = ex_g + ag; s_ex[2] = tk_next;
      if (t == nt - 1) publish_split(d, s, ex_ng + an + m.tng, ex_g + ag + m.tg);   // the stream's last tile: T, M, G of the frame
    }
  }
  __syncthreads();
  int r_ng = s_ex[0], r_g = s_ex[1];
  ex_ng = r_ng + m.tng; ex_g = r_g + m.tg;   // prefix behind this tile: what the workgroup carries to its next one
  r_ng += m.wng; r_g += m.wg;
  const size_t so = (size_t)s * d.Nmax;
  const float zorg = d.zorg[s]; const int zbase = d.zbase[s];
  const uint32_t base = (uint32_t)t * SP_TILE + wave_id() * (SP_ROWS * 64) + lane_id();
#pragma unroll
  for (int it = 0; it < SP_ROWS; ++it) {
    const int c = split_class<PASSB>(d, n_in, base + it * 64, p[it], cls[it]);
    const unsigned long long m_ng = __ballot(c == 2), m_g = __ballot(c == 1);
    const int k_ng = r_ng + __popcll(m_ng & lanemask_lt()), k_g = r_g + __popcll(m_g & lanemask_lt());
    if (c == 2) {
      int cx, cy, cz; bool clamped; grid_cell(G, p[it], zorg, zbase, cx, cy, cz, clamped);
      if (clamped && d.gmode != 0) mor_raise(d, s, 8u);
      d.cloud[so + k_ng] = p[it];
      d.pkey[so + k_ng] = grid_key(G, cx, cy, cz);
    } else if (c == 1) {
#ifdef MOR_EXP_SPLITVAR
      if (!((d.t1_budget >> 17) & 1))   // experiment: no ground stores
#endif
      st_stream(&d.ground[2 * so + d.Nmax + k_g], p[it]);   // final place in filterCloud's output
    }
    r_ng += __popcll(m_ng); r_g += __popcll(m_g);
  }
}
template <bool PASSB> __global__ __launch_bounds__(MOR_BT, 4) void k_split(MorDev d) {   // (≤ 128 VGPRs at least — 73 with 1024-record tiles; with 2048-record tiles the compiler left to itself wandered between 126 and 150 registers with unrelated edits, and at 150 the split took 115 instead of 89 µs)
  int s, g; map_block(d.B, d.sp_g, s, g);
  const MorGrid G = stream_grid(d, s);   // the stream's clustering grid (voxel ground variant: its own number of z layers)
  // As the first kernel of a frame (crop variant) this one reads the stream's arguments straight from the page-locked slot the host filled,
  // and the owner of tile 0 leaves the device copy for the kernels behind it: no copy, no launch and no wait in front of the frame
  const MorStreamArgs a = d.args_src ? d.args_src[s] : d.args[s];
  const uint32_t n_in = pass_count(d, a, s);
  const int nt = (int)((n_in + SP_TILE - 1) / SP_TILE);
  const unsigned epoch = 2u * (unsigned)d.frame_no + (d.gmode == 2 ? 2u : 1u);   // never 0 (fresh descriptors), never the tag of an earlier pass over this table
  __shared__ int sh[16], s_ex[6];   // two copies of each, used in turn by the two halves of the loop: between two uses of a copy lies a workgroup barrier of the other half
  int *tk = d.tickets + (size_t)s * TK_COUNT + TK_SPLIT;
  const int tk_total = 2 * d.sp_g + nt;
  if (threadIdx.x == 0) {
    const int v = __hip_atomic_fetch_add(tk, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_ex[5] = v;
    if (v + 2 == tk_total) __hip_atomic_store(tk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v == 0) {   // the owner of tile 0 starts the frame: before any flag of this stream can be raised (every other tile waits for tile 0's descriptor)
      reset_frame_info(d, s, a.n);
      if (d.args_src) d.args_out[s] = a;
      if (nt == 0) publish_split(d, s, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  int t = __builtin_amdgcn_readfirstlane(s_ex[5]), t1 = t + 1, t_prev = -1;
  int ex_ng = 0, ex_g = 0;
  float4 pa[SP_ROWS], pb[SP_ROWS]; int ca[SP_ROWS], cb[SP_ROWS];
#ifdef MOR_EXP_SPLITVAR
  if ((d.t1_budget >> 19) & 1) {   // experiment: pure read of the stream's tiles (static tiles, two in flight), one dummy store
    float acc = 0.f;
    for (int tt = g; tt < nt; tt += 2 * d.sp_g) {
      split_load_tile<PASSB>(d, a, s, n_in, tt, pa, ca);
      if (tt + d.sp_g < nt) split_load_tile<PASSB>(d, a, s, n_in, tt + d.sp_g, pb, cb);
#pragma unroll
      for (int it = 0; it < SP_ROWS; ++it) acc += pa[it].x + pa[it].w;
      if (tt + d.sp_g < nt) {
#pragma unroll
        for (int it = 0; it < SP_ROWS; ++it) acc += pb[it].y;
      }
    }
    if (acc == 12345.678f) d.cloud[(size_t)s * d.Nmax] = make_float4(acc, 0, 0, 0);
    return;
  }
#endif
  SplitMeta ma, mb;
  if (t < nt) split_load_tile<PASSB>(d, a, s, n_in, t, pa, ca);
  if (t1 < nt) split_load_tile<PASSB>(d, a, s, n_in, t1, pb, cb);
  if (t < nt) split_count<PASSB>(d, s, t, n_in, epoch, pa, ca, sh, ma);
  while (t < nt) {   // pa: tile t, counted and published; pb: tile t1, loaded
    int nx = 0;
    if (t1 < nt) split_count<PASSB>(d, s, t1, n_in, epoch, pb, cb, sh + 8, mb);   // the next tile's aggregate is out before this workgroup waits for anybody
    if (threadIdx.x == 0) nx = __hip_atomic_fetch_add(tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    split_store<PASSB>(d, G, s, t, nt, t_prev, n_in, epoch, pa, ca, ma, ex_ng, ex_g, s_ex, nx);
    if (threadIdx.x == 0 && nx + 1 == tk_total) __hip_atomic_store(tk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t_prev = t;
    const int t2 = __builtin_amdgcn_readfirstlane(s_ex[2]);
    if (t2 < nt) split_load_tile<PASSB>(d, a, s, n_in, t2, pa, ca);
    if (t1 >= nt) break;
    if (t2 < nt) split_count<PASSB>(d, s, t2, n_in, epoch, pa, ca, sh, ma);
    if (threadIdx.x == 0) nx = __hip_atomic_fetch_add(tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    split_store<PASSB>(d, G, s, t1, nt, t_prev, n_in, epoch, pb, cb, mb, ex_ng, ex_g, s_ex + 3, nx);
    if (threadIdx.x == 0 && nx + 1 == tk_total) __hip_atomic_store(tk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t_prev = t1;
    const int t3 = __builtin_amdgcn_readfirstlane(s_ex[5]);
    if (t3 < nt) split_load_tile<PASSB>(d, a, s, n_in, t3, pb, cb);
    t = t2; t1 = t3;
  }
}

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
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  __syncthreads();
  if (lane == 0) sh[wave_id()] = sum;
  __syncthreads();
  int run = 0, total = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) { const int x = sh[w]; if (w < wave_id()) run += x; total += x; }
  for (int i0 = b; i0 < e; i0 += 64) {
    const int i = i0 + lane, v = i < e ? gh_ld<L>(a + i) : 0, inc = wave_incl_scan(v);
    if (i < e) gh_st<L>(a + i, run + inc - v);
    run += __shfl(inc, 63, 64);
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
  const int lane = (int)(threadIdx.x & 63), prev = __shfl_up(v, 1, 64);
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
      }
    }
    __syncthreads();
  }
}
// TL / RL / CL: hash table / row table / per-cell lists in LDS (else global memory).  Returns false when the table
// overflowed (nothing published yet: the caller re-runs with a bigger table).  `cells` lists the claimed slots in
// discovery order — every per-cell phase walks it (a few entries per thread) instead of the whole table; `rowlist`
// first holds the x of the cells of every row, then (same memory) the point counts in compact-id order.
template <bool TL, bool RL, bool CL> __device__ __forceinline__ bool gh_run(const MorDev &d, const MorGrid &G, int s, int M, int *tkey, int *tval, int H, int cell_cap, int *rows, int *cells, int *rowlist, int *l_misc, int *l_sh, int *l_bits) {
  const size_t so = (size_t)s * d.Nmax;
  int *cstart = d.cstart + (size_t)s * (d.Nmax + 1), *ckey = d.ckey + so;
  const int nrows = G.nrows, nx = G.nx, tid = threadIdx.x;
  const int nch = (M + GC_CHUNK - 1) / GC_CHUNK;
  const int2 *clist = d.gc_list + so; int2 *cent = d.gc_ent + so; const int *cn = d.gc_n + (size_t)s * d.gc_chunks;
  int hbits = 0; while ((1 << hbits) < H) ++hbits;
  const unsigned hshift = 32 - hbits, mask = (unsigned)H - 1u;
  unsigned short *rs16 = d.rs16 + (size_t)s * d.rs16_stride, *cx16 = d.cx16 + (size_t)s * d.cx16_stride;   // 16-bit copies of the row table and the cells' x for the scoring tiers (cidx_load)
  const size_t stw = (size_t)s * (MOR_MAXP + 2) + MOR_MAXP; (void)stw;
  ST2(stw, 0);
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
    for (int g = tid; g < total; g += GH_T) {
      if (gh_ld<true>(&l_misc[1])) break;
      int lo = 0, hi = nch - 1;   // last chunk whose prefix ≤ g
      while (lo < hi) { const int m = (lo + hi + 1) >> 1; if (cpre[m] <= g) lo = m; else hi = m - 1; }
      const size_t at = (size_t)lo * GC_CHUNK + (g - cpre[lo]);
      const int2 kc = clist[at];
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
      if (ok) cent[at] = make_int2((int)h, atomicAdd(tval + h, kc.y)); else gh_st<true>(&l_misc[1], 1);
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
  ST2(stw, 1);
  // ---- cells per row → row table
  for (int r = tid; r <= nrows; r += GH_T) gh_st<RL>(rows + r, 0);
  __syncthreads();
  for (int e = tid; e < nocc; e += GH_T) { const int key = gh_ld<TL>(tkey + gh_ld<CL>(cells + e)) - 1; atomicAdd(rows + key / nx, 1); }
  __syncthreads();
  ST2(stw, 6);
  gh_scan<RL>(rows, nrows, l_sh);
  if (tid == 0) gh_st<RL>(rows + nrows, nocc);
  __syncthreads();
  if (RL) { int *grs = d.row_start + (size_t)s * (d.g.nrows + 1); for (int r = tid; r <= nrows; r += GH_T) { const int v = rows[r]; grs[r] = v; rs16[r] = (unsigned short)v; } }
  else if (d.use_hash) for (int r = tid; r <= nrows; r += GH_T) rs16[r] = (unsigned short)gh_ld<RL>(rows + r);   // (meaningful while nocc ≤ 65 535: cidx_load checks)
  slab_bounds<RL>(d, G, s, rows, nocc, l_sh);
  ST2(stw, 7);
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
  ST2(stw, 8);
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
  ST2(stw, 10);
  // ---- point counts in id order (same memory as the row lists) → first position of every cell
  int *cnt = rowlist;
  for (int e = tid; e < nocc; e += GH_T) { const int sl = gh_ld<CL>(cells + e); gh_st<CL>(cnt + gh_ld<TL>(tkey + sl) - 1, gh_ld<TL>(tval + sl)); }
  __syncthreads();
  gh_scan<CL>(cnt, nocc, l_sh);
  ST2(stw, 11);
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
  ST2(stw, 2);
  // ---- the table itself (slot → compact cell id + 1, slot → first position of the cell) goes to global memory: k_gridplace turns its chunks'
  //      (slot, offset) entries into (cell, position) — no second sweep over the entries here, in the one workgroup the stream waits for
  if (TL) {
    int2 *gt = d.gc_tab + (size_t)s * 16384;
    for (int i = tid; i < H; i += GH_T) gt[i] = make_int2(gh_ld<TL>(tkey + i), gh_ld<TL>(tval + i));
  }
  if (tid == 0) d.gc_tabsel[s] = TL ? 1 : 0;   // 0: the table already lives in global memory (gh_key / gh_val)
  ST2(stw, 3); ST2V(stw, 4, M); ST2V(stw, 5, nocc);
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
  int tier = d.gh_tier;
  if (tier < 0) { const int h = ld_agent(&d.gh_hint[s]); const long long need = (long long)h + h / 16; tier = need > min(GH_C0, min(GH_H0, d.Hcell) / 4 * 3) ? (need > min(GH_H, d.Hcell) / 4 * 3 ? 2 : 1) : 0; }
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
__device__ __forceinline__ long long shfl_up_ll(long long v, int o) {
  int lo = __shfl_up((int)(unsigned)v, o, 64), hi = __shfl_up((int)(v >> 32), o, 64);
  return ((long long)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ CellAcc acc_shfl_up(const CellAcc &r, int o) {
  CellAcc t;
  t.lx = __shfl_up(r.lx, o, 64); t.ly = __shfl_up(r.ly, o, 64); t.lz = __shfl_up(r.lz, o, 64); t.hx = __shfl_up(r.hx, o, 64); t.hy = __shfl_up(r.hy, o, 64); t.hz = __shfl_up(r.hz, o, 64);
  t.mi = __shfl_up(r.mi, o, 64);
#pragma unroll
  for (int k = 0; k < 3; ++k) { t.a[k] = shfl_up_ll(r.a[k], o); t.b[k] = shfl_up_ll(r.b[k], o); }
  return t;
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
__device__ __forceinline__ bool pair_hit_wave(const float4 *sp, int a0, int na, int b0, int nb, float r2, int lane,
                                              const float4 &alo, const float4 &ahi, const float4 &blo, const float4 &bhi) {
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
        const float bx = __shfl(pb.x, l, 64), by = __shfl(pb.y, l, 64), bz = __shfl(pb.z, l, 64);
        hit |= a_act && sqdist(pa.x, pa.y, pa.z, bx, by, bz) < r2;
      }
      if (__ballot(hit)) return true;
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
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mi = min(mi, __shfl_xor(mi, o, 64));
    lx = fminf(lx, __shfl_xor(lx, o, 64)); ly = fminf(ly, __shfl_xor(ly, o, 64)); lz = fminf(lz, __shfl_xor(lz, o, 64));
    hx = fmaxf(hx, __shfl_xor(hx, o, 64)); hy = fmaxf(hy, __shfl_xor(hy, o, 64)); hz = fmaxf(hz, __shfl_xor(hz, o, 64));
  }
  lo = make_float4(lx, ly, lz, 0.f); hi = make_float4(hx, hy, hz, 0.f);
}

// Per occupied cell: the box of its points, its first point (sample for the quick edge test of the cell graph), its
// smallest cloud index and the exact sums of its coordinates — ONE streaming pass over `sorted`, balanced whatever the
// cell sizes are (a thread group per cell — round 1/2 — ended with the cells of thousands of points): a wave takes 256
// consecutive positions, four per lane; a lane folds its four points serially, the open runs at lane boundaries go
// through a segmented scan over the lanes (19 words × 6 shuffles per 256 points), and whoever holds the last point of a
// cell writes its record.  Cells that continue into another wave tile are merged with atomics (min / max / integer
// add: order-free), their records were initialised by k_gridhash.
__global__ __launch_bounds__(MOR_BT) void k_cellboxes(MorDev d) {
  int s, bx, gbx;
  if (!map_block_work(d, [&](int s_) { return ((int)d.info[s_].M + (MOR_BT / 64) * CB_WTILE - 1) / ((MOR_BT / 64) * CB_WTILE); }, s, bx, gbx)) return;   // work: steps of one workgroup over the stream's cell-ordered points
  const int M = d.info[s].M, lane = lane_id();
  const size_t so = (size_t)s * d.Nmax;
  const float4 *sp = d.sorted + so; const int *sc = d.scell + so;
  for (int base = (bx * (MOR_BT / 64) + wave_id()) * CB_WTILE; base < M; base += gbx * (MOR_BT / 64) * CB_WTILE) {
    const int j0 = base + 4 * lane;
    int c[4]; float4 p[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int j = min(j0 + u, M - 1); c[u] = sc[j]; p[u] = sp[j]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) if (j0 + u >= M) c[u] = -1;
    int left = -2, right = -3;   // cells of the positions just outside the tile
    if (lane == 0 && base > 0) left = sc[base - 1];
    if (lane == 63 && base + CB_WTILE < M) right = sc[base + CB_WTILE];
    const int cw0 = __shfl(c[0], 0, 64), cwl = __shfl(c[3], 63, 64);
    const bool open_l = __shfl(left, 0, 64) == cw0, open_r = __shfl(right, 63, 64) == cwl;
    int prevc = __shfl_up(c[3], 1, 64), nextc = __shfl_down(c[0], 1, 64);
    if (lane == 0) prevc = left;
    if (lane == 63) nextc = right;
    // the lane's tail run (the run holding its last position) and whether it began in an earlier lane
    int ts = 3;
    if (c[2] == c[3]) { ts = 2; if (c[1] == c[3]) { ts = 1; if (c[0] == c[3]) ts = 0; } }
    CellAcc S; acc_clear(S);
#pragma unroll
    for (int u = 0; u < 4; ++u) if (u >= ts) acc_point(S, p[u]);
    const bool head = !(ts == 0 && c[0] == prevc) || lane == 0;
    const unsigned long long heads = __ballot(head);
    const int hl = 63 - __clzll((long long)(heads & (lanemask_lt() | (1ull << lane))));
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const CellAcc t = acc_shfl_up(S, o); if (lane - o >= hl) acc_merge(S, t); }
    CellAcc acc = acc_shfl_up(S, 1);   // the run reaching this lane from the left, up to the previous lane
    if (lane == 0 || c[0] != prevc) acc_clear(acc);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (c[u] < 0) break;
      if (c[u] != (u ? c[u - 1] : prevc)) d.crep[so + c[u]] = p[u];   // first position of the cell
      acc_point(acc, p[u]);
      const int nxt = u < 3 ? c[u + 1] : nextc;
      if (c[u] != nxt || (u == 3 && lane == 63)) {
        acc_emit(d, so, c[u], acc, (c[u] == cw0 && open_l) || (c[u] == cwl && open_r));
        acc_clear(acc);
      }
    }
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
#define CGS_CAP 1024      // local cells (own + look-ahead) held in LDS
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
template <bool LDS> struct CgsCells { const int *key, *pc; const float *rx, *ry, *rz, *bx; int cap; };   // bx: six planes of `cap` floats (null: samples and boxes stay in global memory)
template <bool LDS, bool BOXL, typename RT> __device__ __forceinline__ void cgs_hooks(const MorDev &d, const MorGrid &G, size_t soc, int n_own, int n_loc, const CgsCells<LDS> &L, const int *start, const RT *rows, int rsub, int r0, int nlrows,
                                                              int *par, const float4 *sp, int *ovf, int *l_list, int *l_queue, int *l_wcnt, int *l_n2, size_t stw) {
  const float r2 = d.r2;
  const int *key = L.key;
  constexpr int NR = 13;   // rows of the forward half: (dy,dz) = (0,0) (0,1) (1,−1) (1,0) (1,1) — the 3×3×3 block — then (0,2), (1,±2), (2,−2…2)
  const int w = wave_id(), lane = lane_id();
  int *queue = l_queue + w * CGS_QW;
  int wcount = 0;
  ST2(stw, 1);
#ifdef MOR_EXP_STAMPS
  unsigned long long ta = 0, tb = 0, tc = 0, td = 0;
#define CGS_TICK(v) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long v = wall_clock64()
#else
#define CGS_TICK(v)
#endif
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
#ifdef MOR_EXP_STAMPS
    { const int s = (int)(soc / d.Nmax); const int n_it = __popcll(__ballot(it < n_own * NR)), n_ne = __popcll(__ballot(b < hi)); if (lane == 0) { RS_ADD(0, n_it); RS_ADD(1, n_ne); } }
#endif
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
#ifdef MOR_EXP_STAMPS
    { const int s = (int)(soc / d.Nmax); if (lane == 0) { RS_ADD(2, qn); RS_ADD(3, 1); } }
#endif
    // ---- A2
    for (int q0 = 0; q0 < qn; q0 += 64) {
      const bool act = q0 + lane < qn;
      const int qc = act ? queue[q0 + lane] : 0;
      const int qa = __shfl(a, (qc >> 26) & 63, 64), qb = qc & ((1 << 26) - 1);
      bool want = act && cg_find<LDS>(par, qa) != cg_find<LDS>(par, qb);
      if (want) {
        float pax, pay, paz, alx, aly, alz, ahx, ahy, ahz, qx, qy, qz, blx, bly, blz, bhx, bhy, bhz;
        if (BOXL) {
          pax = L.rx[qa]; pay = L.ry[qa]; paz = L.rz[qa]; alx = L.bx[qa]; aly = L.bx[L.cap + qa]; alz = L.bx[2 * L.cap + qa]; ahx = L.bx[3 * L.cap + qa]; ahy = L.bx[4 * L.cap + qa]; ahz = L.bx[5 * L.cap + qa];
          qx = L.rx[qb]; qy = L.ry[qb]; qz = L.rz[qb]; blx = L.bx[qb]; bly = L.bx[L.cap + qb]; blz = L.bx[2 * L.cap + qb]; bhx = L.bx[3 * L.cap + qb]; bhy = L.bx[4 * L.cap + qb]; bhz = L.bx[5 * L.cap + qb];
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
#ifdef MOR_EXP_STAMPS
    { CGS_TICK(k3); ta += k1 - k0; tb += k2 - k1; tc += k3 - k2; }
#endif
  }
#ifdef MOR_EXP_STAMPS
  { const int s = (int)(soc / d.Nmax); if (lane == 0) { RS_ADD(4, ta); RS_ADD(5, tb); RS_ADD(6, tc); RS_ADD(7, 1); } (void)td; }
#endif
  if (lane == 0) l_wcnt[w] = min(wcount, cgs_wlist_cap<LDS>());
  __threadfence_block();
  __syncthreads();
  ST2(stw, 2);
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
  ST2(stw, 3);
  // ---- B2: one wave per pair left over
  const int n2 = min(*l_n2, cgs_list2_cap<LDS>());
  for (int h = w; h < n2; h += CGS_NW) {
    int a, b; cgs_list2_get<LDS>(ovf, h, a, b);
    if (cg_find<LDS>(par, a) == cg_find<LDS>(par, b)) continue;
    const float4 alo = d.cmeta[2 * (soc + a)], ahi = d.cmeta[2 * (soc + a) + 1], blo = d.cmeta[2 * (soc + b)], bhi = d.cmeta[2 * (soc + b) + 1];
    if (pair_hit_wave(sp, start[a], start[a + 1] - start[a], start[b], start[b + 1] - start[b], r2, lane, alo, ahi, blo, bhi) && lane == 0) cg_unite<LDS>(par, a, b);
  }
  ST2(stw, 4);
  ST2V(stw, 10, n1); ST2V(stw, 11, n2);
  __syncthreads();
}
template <bool LDS, bool BOXL, typename RT> __device__ __forceinline__ void cgs_body(const MorDev &d, const MorGrid &G, int s, size_t so, int c0, int n_own, int n_loc, const CgsCells<LDS> &L, const RT *rows, int rsub, int r0, int nlrows, int *par, int *ovf, int *l_list, int *l_queue, int *l_wcnt, int *l_n2, size_t stwj) {
  const int *start = d.cstart + (size_t)s * (d.Nmax + 1) + c0;   // start[local id]: first position of the cell in `sorted`
  const float4 *sp = d.sorted + so;
  cgs_hooks<LDS, BOXL, RT>(d, G, so + c0, n_own, n_loc, L, start, rows, rsub, r0, nlrows, par, sp, ovf, l_list, l_queue, l_wcnt, l_n2, stwj);
  // local roots as global compact ids: own cells → lroot_a, look-ahead cells → lroot_b
  for (int c = threadIdx.x; c < n_loc; c += CGS_T) {
    const int r = c0 + cg_find<LDS>(par, c);
    if (c < n_own) st_agent(&d.lroot_a[so + c0 + c], r); else st_agent(&d.lroot_b[so + c0 + c], r);   // (agent scope: the merge may run in another slab's workgroup of this launch, stream_last_block)
  }
  ST2(stwj, 9); ST2V(stwj, 14, n_own); ST2V(stwj, 15, n_loc); ST2V(stwj, 12, nlrows); ST2V(stwj, 13, __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)));   // (HW_ID: wave, SIMD, CU, SE … of the recording wave)
}
template <bool LDS, int NT> __device__ __forceinline__ void cgf_body(const MorDev &d, int s, int nocc, int *par, int *scr, int *l_misc, const int *l_sc, const int *l_se);
// CAP: local cells (own + look-ahead) the workgroup holds in LDS (76 KB: two workgroups per CU).
// With d.cg_fused the stream's LAST slab workgroup to finish (stream_last_block) goes on with the merge of the slab forests and everything
// k_cg_final does, in the same LDS (three arrays of CGS_FCAP cells; streams with more cells: global-memory arrays) — one launch and one
// queueing delay less per frame; the host falls back to the separate k_cg_final launch (147 KB of LDS: 12 288 cells) when the previous
// frame's cell counts say a stream would not fit.
#define CGS_SLAB_WORDS (12 * CGS_CAP + CGS_ROWCAP + 1 + CGS_LISTW + CGS_NW * CGS_QW)
#define CGS_ARENA (CGS_SLAB_WORDS > 2 * MOR_CGS_FCAP ? CGS_SLAB_WORDS : 2 * MOR_CGS_FCAP)   // (the slab layout of the default build is 18 945 words)
#define CGS_FCAP (CGS_ARENA / 2)
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
  __shared__ int l_misc[1 + CGS_T / 64], l_sc[MOR_MAXP + 1], l_se[MOR_MAXP + 1];
  if (threadIdx.x <= Ps) { l_sc[threadIdx.x] = d.slab_c[(size_t)s * (MOR_MAXP + 1) + threadIdx.x]; l_se[threadIdx.x] = d.slab_e[(size_t)s * (MOR_MAXP + 1) + threadIdx.x]; }
  if (nocc <= CGS_FCAP && !d.cg_force_global) {
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
  const size_t stwj = (size_t)s * (MOR_MAXP + 2) + j; (void)stwj;
  ST2(stwj, 0);
  const int y0 = sy[j], y2 = min(sy[j + 1] + 2, G.ny), r0 = y0 * G.nz, nlrows = (y2 - y0) * G.nz;
  // 12·CAP words of cell data: CAP cells with everything in LDS (key, parent, packed coordinates, sample point, box), or —
  // slabs of up to 4·CAP cells, e.g. a façade across a y-slice — key, parent and packed coordinates only: the enumeration
  // (A1) and the forest stay in LDS, the decisions about queued pairs (A2) fetch samples and boxes from global memory
  int *l_cells = l_arena, *l_list = l_cells + 12 * CAP + CGS_ROWCAP + 1, *l_queue = l_list + CGS_LISTW;
  unsigned short *l_rows = reinterpret_cast<unsigned short *>(l_cells + 12 * CAP);   // local row table as 16-bit offsets (≤ 4·CAP local cells): 2·CGS_ROWCAP rows in CGS_ROWCAP + 1 words
  int &l_n2 = *l_n2p;
  int *ovf = d.cg_ovf + (size_t)(s * MOR_MAXP + j) * MOR_CGS_OVF * 2;   // [0, MOR_CGS_OVF): the waves' candidate lists, [MOR_CGS_OVF, 2·MOR_CGS_OVF): pairs for whole waves
  const int *g_rows = d.row_start + (size_t)s * (d.g.nrows + 1) + r0;
  if (threadIdx.x == 0) l_n2 = 0;
  const bool fits_rows = nlrows <= 2 * CGS_ROWCAP && !d.cg_force_global;   // (thick slabs of the sparse far ends of a cloud: 150 slices × 14 layers seen at 120 000 points; beyond the table they ran the global-memory path, 3× slower, and set the kernel's span)
  if (fits_rows && n_loc <= CAP) {
    int *l_key = l_cells, *l_par = l_cells + CAP, *l_pc = l_cells + 2 * CAP;
    float *l_rx = reinterpret_cast<float *>(l_cells + 3 * CAP), *l_ry = l_rx + CAP, *l_rz = l_rx + 2 * CAP, *l_bx = l_rx + 3 * CAP;
    const int *gk = d.ckey + so + c0; const float4 *grep = d.crep + so + c0, *gm = d.cmeta + 2 * (so + c0);
    for (int i = threadIdx.x; i < n_loc; i += CGS_T) {
      const int k = gk[i], row = k / G.nx;
      l_key[i] = k; l_par[i] = i; l_pc[i] = (int)((unsigned)(k - row * G.nx) | ((unsigned)(row % G.nz) << 11) | ((unsigned)(row / G.nz) << 21));
      const float4 q = grep[i], lo = gm[2 * i], hi4 = gm[2 * i + 1];
      l_rx[i] = q.x; l_ry[i] = q.y; l_rz[i] = q.z;
      l_bx[i] = lo.x; l_bx[CAP + i] = lo.y; l_bx[2 * CAP + i] = lo.z; l_bx[3 * CAP + i] = hi4.x; l_bx[4 * CAP + i] = hi4.y; l_bx[5 * CAP + i] = hi4.z;
    }
    for (int i = threadIdx.x; i <= nlrows; i += CGS_T) l_rows[i] = (unsigned short)(g_rows[i] - c0);
    __syncthreads();
    const CgsCells<true> L = {l_key, l_pc, l_rx, l_ry, l_rz, l_bx, CAP};
    cgs_body<true, true, unsigned short>(d, G, s, so, c0, n_own, n_loc, L, l_rows, 0, r0, nlrows, l_par, ovf, l_list, l_queue, l_wcnt, &l_n2, stwj);
  } else if (fits_rows && n_loc <= 4 * CAP) {
    int *l_key = l_cells, *l_par = l_cells + 4 * CAP, *l_pc = l_cells + 8 * CAP;
    const int *gk = d.ckey + so + c0;
    for (int i = threadIdx.x; i < n_loc; i += CGS_T) {
      const int k = gk[i], row = k / G.nx;
      l_key[i] = k; l_par[i] = i; l_pc[i] = (int)((unsigned)(k - row * G.nx) | ((unsigned)(row % G.nz) << 11) | ((unsigned)(row / G.nz) << 21));
    }
    for (int i = threadIdx.x; i <= nlrows; i += CGS_T) l_rows[i] = (unsigned short)(g_rows[i] - c0);
    __syncthreads();
    const CgsCells<true> L = {l_key, l_pc, nullptr, nullptr, nullptr, nullptr, 0};
    cgs_body<true, false, unsigned short>(d, G, s, so, c0, n_own, n_loc, L, l_rows, 0, r0, nlrows, l_par, ovf, l_list, l_queue, l_wcnt, &l_n2, stwj);
  } else {   // slab too big for LDS: the same code on global arrays (even and odd slabs use different forests: look-aheads overlap the next slab)
    int *par = ((j & 1) ? d.parent2 : d.parent) + so + c0;
    for (int i = threadIdx.x; i < n_loc; i += CGS_T) cg_st<false>(par + i, i);
    __threadfence_block();   // (the forest is this workgroup's alone and accessed with agent-scope operations; a device-wide fence writes the XCD's whole L2 back: 32 µs)
    __syncthreads();
    const CgsCells<false> L = {d.ckey + so + c0, nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    cgs_body<false, false, int>(d, G, s, so, c0, n_own, n_loc, L, g_rows, c0, r0, nlrows, par, ovf, l_list, l_queue, l_wcnt, &l_n2, stwj);
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
__global__ __launch_bounds__(CGF_T) void k_cg_final(MorDev d) {
  const int s = blockIdx.x + d.s0, nocc = d.info[s].n_occ;
  const size_t so = (size_t)s * d.Nmax;
  __shared__ int l_par[CGF_CAP], l_a[CGF_CAP], l_misc[1 + CGF_T / 64], l_sc[MOR_MAXP + 1], l_se[MOR_MAXP + 1];
  if (threadIdx.x <= d.slab_p[s]) { l_sc[threadIdx.x] = d.slab_c[(size_t)s * (MOR_MAXP + 1) + threadIdx.x]; l_se[threadIdx.x] = d.slab_e[(size_t)s * (MOR_MAXP + 1) + threadIdx.x]; }
  if (nocc <= CGF_CAP && !d.cg_force_global) {          // forest, sizes (then cluster ids), minima: three LDS arrays
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

// ------------------------------------------------------------------------------------ stable LSD radix sort, 8-bit digits, batched over streams
// used by the sort path of the grid (MOR_GRID=radix) and by the VoxelGrid pass of the voxel ground variant.
__device__ __forceinline__ void radix_item(const MorRadix &j, size_t so, int count, int i, int &key, int &val, bool &valid) {
  valid = i < count; key = 0; val = 0;
  if (!valid) return;
  key = j.kin[so + i]; val = j.vin ? j.vin[so + i] : i;
  if (j.drop_negative) valid = key >= 0;
}
__device__ __forceinline__ int radix_count(const MorDev &d, const MorRadix &j, int s) {
  if (j.skip_k_le > 0 && (int)d.info[s].K <= j.skip_k_le) return 0;   // all higher digits are zero: the previous pass already produced the final order
  if (j.vox && (j.shift >> 3) >= voxel_passes_of(d, s)) return 0;   // voxel keys of this stream end below this digit: its order is final (the consumers pick the buffer by the stream's pass count)
  return j.count_sel == 0 ? d.info[s].M : d.info[s].C;
}

__global__ __launch_bounds__(MOR_BT) void k_rhist(MorDev d, MorRadix j) {
  int s, t0; map_block(d.B, d.tiles_m, s, t0);
  const int count = radix_count(d, j, s);
  __shared__ int h[256];
  const size_t so = (size_t)s * d.Nmax;
  for (int t = t0; t * MOR_TILE < count; t += d.tiles_m) {
    const int base = t * MOR_TILE;
    h[threadIdx.x] = 0;
    __syncthreads();
    for (int i = base + threadIdx.x; i < min(base + MOR_TILE, count); i += MOR_BT) {
      int key, val; bool valid; radix_item(j, so, count, i, key, val, valid);
      if (valid) atomicAdd(&h[(key >> j.shift) & 255], 1);
    }
    __syncthreads();
    j.hist[((size_t)s * d.tiles_max + t) * 256 + threadIdx.x] = h[threadIdx.x];
  }
}
// one workgroup per stream, one thread per digit: offsets[tile][digit] = Σ smaller digits + Σ earlier tiles
__global__ __launch_bounds__(MOR_BT) void k_rscan(MorDev d, MorRadix j) {
  int s = blockIdx.x + d.s0; __shared__ int sh[8];
  int *h = j.hist + (size_t)s * d.tiles_max * 256 + threadIdx.x;
  const int nt = (radix_count(d, j, s) + MOR_TILE - 1) / MOR_TILE;
  int run = 0, t = 0;
  for (; t + 4 <= nt; t += 4) {   // four independent loads per step
    int v0 = h[t * 256], v1 = h[(t + 1) * 256], v2 = h[(t + 2) * 256], v3 = h[(t + 3) * 256];
    h[t * 256] = run; run += v0; h[(t + 1) * 256] = run; run += v1; h[(t + 2) * 256] = run; run += v2; h[(t + 3) * 256] = run; run += v3;
  }
  for (; t < nt; ++t) { int v = h[t * 256]; h[t * 256] = run; run += v; }
  int tot, base = block_excl_scan(run, sh, &tot);
  for (t = 0; t < nt; ++t) h[t * 256] += base;
}
__global__ __launch_bounds__(MOR_BT) void k_rscatter(MorDev d, MorRadix j) {
  int s, t0; map_block(d.B, d.tiles_m, s, t0);
  const int count = radix_count(d, j, s);
  const size_t so = (size_t)s * d.Nmax;
  const bool inverse = j.inverse && (!j.vox || (j.shift >> 3) == voxel_passes_of(d, s) - 1);   // the stream's LAST pass leaves the inverse permutation
  __shared__ int wcnt[4][256]; __shared__ int shs[8];
  for (int t = t0; t * MOR_TILE < count; t += d.tiles_m) {
    const int tb = t * MOR_TILE;
    for (int k = threadIdx.x; k < 4 * 256; k += MOR_BT) (&wcnt[0][0])[k] = 0;
    __syncthreads();
    int key[8], val[8], pre[8]; bool valid[8];
    int base = tb + wave_id() * 512;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      int i = base + it * 64 + lane_id();
      radix_item(j, so, count, i, key[it], val[it], valid[it]);
      int dg = (key[it] >> j.shift) & 255;
      unsigned long long peers = __ballot(valid[it]);
#pragma unroll
      for (int b = 0; b < 8; ++b) { unsigned long long m = __ballot((dg >> b) & 1); peers &= ((dg >> b) & 1) ? m : ~m; }
      pre[it] = 0;
      if (valid[it]) {
        int leader = __ffsll((long long)peers) - 1, rank = __popcll(peers & lanemask_lt()), basec = 0;
        if (lane_id() == leader) basec = atomicAdd(&wcnt[wave_id()][dg], __popcll(peers));
        basec = __shfl(basec, leader, 64);
        pre[it] = basec + rank;
      }
    }
    __syncthreads();
    {  // exclusive prefix over the 4 waves per digit + global offset of (tile, digit)
      int dg = threadIdx.x, run;
      if (j.fuse) {   // raw per-tile histograms: Σ smaller digits (all tiles) + Σ earlier tiles (this digit), re-derived per workgroup
        const int *hh = j.hist + (size_t)s * d.tiles_max * 256 + dg; const int nt = (count + MOR_TILE - 1) / MOR_TILE;
        int before = 0, all = 0, u = 0;
        for (; u + 4 <= nt; u += 4) {
          const int v0 = hh[u * 256], v1 = hh[(u + 1) * 256], v2 = hh[(u + 2) * 256], v3 = hh[(u + 3) * 256];
          all += v0 + v1 + v2 + v3; before += (u < t ? v0 : 0) + (u + 1 < t ? v1 : 0) + (u + 2 < t ? v2 : 0) + (u + 3 < t ? v3 : 0);
        }
        for (; u < nt; ++u) { const int v = hh[u * 256]; all += v; before += u < t ? v : 0; }
        int tot; run = block_excl_scan(all, shs, &tot) + before;
      } else run = j.hist[((size_t)s * d.tiles_max + t) * 256 + dg];
#pragma unroll
      for (int w = 0; w < 4; ++w) { int v = wcnt[w][dg]; wcnt[w][dg] = run; run += v; }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      if (!valid[it]) continue;
      int dg = (key[it] >> j.shift) & 255;
      int pos = wcnt[wave_id()][dg] + pre[it];
      if (j.kout) j.kout[so + pos] = key[it];
      if (inverse) j.vout[so + val[it]] = pos; else j.vout[so + pos] = val[it];
      if (j.vout2) j.vout2[so + pos] = val[it];
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------ C2: per-cluster extraction + centroid + AABB
__device__ __forceinline__ void red6_block(Red6 &r, Red6 *sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    r.sx += __shfl_down(r.sx, o, 64); r.sy += __shfl_down(r.sy, o, 64); r.sz += __shfl_down(r.sz, o, 64);
    r.mnx = fminf(r.mnx, __shfl_down(r.mnx, o, 64)); r.mny = fminf(r.mny, __shfl_down(r.mny, o, 64)); r.mnz = fminf(r.mnz, __shfl_down(r.mnz, o, 64));
    r.mxx = fmaxf(r.mxx, __shfl_down(r.mxx, o, 64)); r.mxy = fmaxf(r.mxy, __shfl_down(r.mxy, o, 64)); r.mxz = fmaxf(r.mxz, __shfl_down(r.mxz, o, 64));
  }
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
#define MOR_CLS_G 8
__device__ __forceinline__ void xform_prev_body(const MorDev &d, int s, int bx, int nbx, Red6 *sh, float *m);
__device__ __forceinline__ void cluster_pairs_body(const MorDev &d, int s, float4 *tile, int *sh);
#define MOR_XF_G 16   // workgroups per stream that transform the previous frame's clusters inside this launch
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
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int j = min(j0 + u * stride, M - 1); p[u] = d.sorted[so + j]; sc[u] = d.scell[so + j]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) g[u] = d.cgat[so + sc[u]];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = j0 + u * stride;
        if (j >= M) continue;
        st_stream(&d.pcid[so + __float_as_int(p[u].w)], g[u].y);   // label of the cloud point (its index travels in .w); read once more, by the output
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
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      r.lx = fminf(r.lx, __shfl_xor(r.lx, o, 64)); r.ly = fminf(r.ly, __shfl_xor(r.ly, o, 64)); r.lz = fminf(r.lz, __shfl_xor(r.lz, o, 64));
      r.hx = fmaxf(r.hx, __shfl_xor(r.hx, o, 64)); r.hy = fmaxf(r.hy, __shfl_xor(r.hy, o, 64)); r.hz = fmaxf(r.hz, __shfl_xor(r.hz, o, 64));
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        r.a[a] += ((long long)__shfl_xor((int)(r.a[a] >> 32), o, 64) << 32) | (unsigned)__shfl_xor((int)(unsigned)r.a[a], o, 64);
        r.b[a] += ((long long)__shfl_xor((int)(r.b[a] >> 32), o, 64) << 32) | (unsigned)__shfl_xor((int)(unsigned)r.b[a], o, 64);
      }
    }
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
__device__ __forceinline__ void wl_push(bool want, int *n, int4 *list, int j, int pr, int target, bool back = false, int cap = 0) {
  unsigned long long m = __ballot(want);
  if (!m) return;
  int basew = 0, leader = __ffsll((long long)m) - 1;
  if (lane_id() == leader) basew = atomicAdd(n, __popcll(m));
  basew = __shfl(basew, leader, 64);
  if (want) { int pos = basew + __popcll(m & lanemask_lt()); list[back ? cap - 1 - pos : pos] = make_int4(j, pr, target, 0); }
}
// wave-aggregated count: all counted queries of a pair add to ONE address (a few dozen addresses per stream), and
// same-address atomics serialise in L2 — thousands of them per stream were the real cost of these kernels.  Lanes
// with the same pair are combined first (worklist order is cluster order, so usually one atomic per wave).
__device__ __forceinline__ void count_push(bool want, int *cnt, int pr) {
  unsigned long long m = __ballot(want);
  while (m) {
    const int l = __ffsll((long long)m) - 1, p = __shfl(pr, l, 64);
    const unsigned long long same = __ballot(want && pr == p);
    if (lane_id() == l) atomicAdd(&cnt[p], __popcll(same));
    m &= ~same;
  }
}
// Tier 1 — one THREAD per query, its OWN cell only.  On a static surface a point of the matched cluster lies
// within √lb of q, almost always in q's own cell: ≈ 85 % of the queries end here (never counted).  The rest is
// compacted into worklists so the next tiers run full waves of like queries: `wl` front = E2 known (a matched point of
// the own cell closer than √ub), `wl` back = own cell without a matched point, `wl2` = big own cell (wave tier).
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
#ifdef MOR_EXP_T1CUT
  const int expv = d.t1_budget >> 16;   // experiment (exp/t1exp.py): cut tier 1 short — 1 before the cell index, 2 behind it, 3 behind the loads of the query, 4 before the scan, 5 before the pushes; results are wrong
  if (expv == 1) return;
#endif
  const CellIdx I = cidx_load(d, G, s, l_idx);
#ifdef MOR_EXP_T1CUT
  if (expv == 2) return;
#endif
  const int *st = d.cstart + (size_t)s * (d.Nmax + 1);
  const float lbn = nextafterf(d.pde_lb, INFINITY);
  const float slb = sqrtf(fmaxf(d.pde_lb, 0.f)) * 1.01f + G.cs * 1e-3f;
  const bool e1_local = 2.f * slb < G.cs;
  for (int base = t0 * SCF_T; base < Cp; base += g_fast * SCF_T) {
    const int j = base + threadIdx.x;
#ifdef MOR_EXP_T1CUT
    if (expv == 3) { if (j < Cp) { const int cidj = d.cl_cid[pv][so + j]; const float4 q = d.cl_pts[pv][so + j]; if (cidj == -12345 && q.x == 1.2345f) d.pair_cnt[ko] = 1; } continue; }
#endif
    bool nearq = false, blockq = false, big = false, counted = false; float best = INFINITY; int pr = -1, target = -1;
#ifdef MOR_EXP_ROUNDS
    int exp_rounds = 0, exp_cell = 0; bool exp_found = false;
#endif
    RS_T(f0);
#ifdef MOR_EXP_STAMPS
    unsigned long long f1 = f0, f2 = f0;
#endif
    if (j < Cp) {
      // two short chains of dependent loads, issued side by side (no branch between them): cluster → its record (pair, matched cluster,
      // that cluster's box), and point → cell (LDS index) → range + cluster id of the cell → points.  (Round 1: seven levels, one after the other.)
      const int cidj = ld_stream(&d.cl_cid[pv][so + j]);
      const float4 q = ld_stream(&d.cl_pts[pv][so + j]);   // (read once here; the few queries the later tiers take up again fetch theirs from HBM)
      const int cx = cell_axis_unclamped(q.x, G.ox, G.inv_cs), cy = cell_axis_unclamped(q.y, G.oy, G.inv_cs), cz = cell_axis_unclamped(q.z, d.zorg[s], G.inv_cs);
      const int c = cidx_find(I, cx, cy, cz);   // (LDS: no global access)
      const float4 tlo = d.qrec[2 * (ko + cidj)], thi = d.qrec[2 * (ko + cidj) + 1];   // the matched cluster's box, the pair, the matched cluster: one record per previous cluster (pairs_body)
      const int cc = max(c, 0), cid = d.ccid[so + cc], b0 = c >= 0 ? st[cc] : 0, e0 = c >= 0 ? st[cc + 1] : 0;
      pr = __float_as_int(tlo.w); target = __float_as_int(thi.w);
#ifdef MOR_EXP_T1CUT
      if (expv == 4) { if (cid == -12345 && b0 == -7 && e0 == -9 && pr == -12345) d.pair_cnt[ko] = 1; pr = -1; }
#endif
#ifdef MOR_EXP_STAMPS
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); f1 = wall_clock64();
#endif
      if (pr >= 0) {
        int budget = 64;   // a big own cell that shows no close point among 64 evenly spread samples goes to the wave tier
        const bool reach = box_dist2(q, tlo, thi) < d.pde_ub;   // farther than √ub from the whole matched cluster: never counted
        if (reach && c >= 0 && cid == target) { scan_sampled(sp, b0, e0, q, lbn, best, budget); big = best > d.pde_lb && e0 - b0 > 64; }
#ifdef MOR_EXP_ROUNDS
        exp_rounds = (64 - budget) / 8; exp_found = best <= d.pde_lb; exp_cell = e0 - b0;
#endif
        if (reach && best > d.pde_lb && !big) {
          if (!e1_local) big = true;   // √lb reaches beyond the adjacent half-cells in this configuration: wave tier
          else if (best < d.pde_ub) {
            if (near_side(q.x, G.ox, G.inv_cs, G.cs, cx, slb) == 0 && near_side(q.y, G.oy, G.inv_cs, G.cs, cy, slb) == 0 && near_side(q.z, d.zorg[s], G.inv_cs, G.cs, cz, slb) == 0)
              counted = true;   // deep inside its cell: no other cell can hold a point within √lb ⇒ counted
            else nearq = true;
          } else blockq = true;
        }
      } else target = -1;
#ifdef MOR_EXP_STAMPS
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); f2 = wall_clock64();
#endif
    }
#ifdef MOR_EXP_T1CUT
    if (expv == 5) { if (nearq && blockq && big && counted) d.pair_cnt[ko] = 2; nearq = blockq = big = counted = false; }
#endif
    {  // the worklists: both ends of `wl` with ONE returning atomic per wave (the two counters share a 64-bit word), `wl2` with another,
       // both issued by lane 0 before either answer is used (one round trip instead of two)
      const unsigned long long mn = __ballot(nearq), mb = __ballot(blockq), mg = __ballot(big);
      if (mn | mb | mg) {
        unsigned long long base = 0ull; int base2 = 0;
        if (lane_id() == 0) {
          if (mn | mb) base = atomicAdd(&d.wl_nb[s], (unsigned long long)__popcll(mn) | ((unsigned long long)__popcll(mb) << 32));
          if (mg) base2 = atomicAdd(&d.wl2_n[s], __popcll(mg));
        }
        const int bn = __shfl((int)(unsigned)base, 0, 64), bb = __shfl((int)(base >> 32), 0, 64), b2 = __shfl(base2, 0, 64);
        if (nearq) d.wl[so + bn + __popcll(mn & lanemask_lt())] = make_int4(j, pr, target, 0);
        if (blockq) d.wl[so + d.Nmax - 1 - (bb + __popcll(mb & lanemask_lt()))] = make_int4(j, pr, target, 0);
        if (big) d.wl2[so + b2 + __popcll(mg & lanemask_lt())] = make_int4(j, pr, target, 0);
      }
    }
    count_push(counted, d.pair_cnt + ko, pr);
#ifdef MOR_EXP_ROUNDS
    {  // experiment: rounds of the sampled scan per lane and per wave (the wave lives as long as its slowest lane)
      int mx = exp_rounds, sum = exp_rounds, nq_ = j < Cp ? 1 : 0, nf1 = (exp_found && exp_rounds <= 1) ? 1 : 0, nbig = exp_cell > 64 ? 1 : 0;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { mx = max(mx, __shfl_xor(mx, o, 64)); sum += __shfl_xor(sum, o, 64); nq_ += __shfl_xor(nq_, o, 64); nf1 += __shfl_xor(nf1, o, 64); nbig += __shfl_xor(nbig, o, 64); }
      if (lane_id() == 0 && nq_) { unsigned long long *g = d.dbg + (size_t)s * 16; atomicAdd(&g[0], (unsigned long long)nq_); atomicAdd(&g[1], (unsigned long long)sum); atomicAdd(&g[2], 1ull); atomicAdd(&g[3], (unsigned long long)mx); atomicAdd(&g[4 + min(mx, 8)], 1ull); atomicAdd(&g[13], (unsigned long long)nf1); atomicAdd(&g[14], (unsigned long long)nbig); }
    }
#endif
#ifdef MOR_EXP_STAMPS
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    { const int n_near = __popcll(__ballot(nearq)), n_block = __popcll(__ballot(blockq)), n_big = __popcll(__ballot(big));
      if (lane_id() == 0 && base < Cp) { const unsigned long long f3 = wall_clock64(); RS_ADD(8, 1); RS_ADD(9, f1 - f0); RS_ADD(10, f2 - f1); RS_ADD(11, f3 - f2); RS_MAX(12, f3 - f0); RS_ADD(13, n_near); RS_ADD(14, n_block); RS_ADD(15, n_big); } }
#endif
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
    bool defer = false, counted = false; int j = 0, pr = -1, target = -1;
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
      if (best > d.pde_lb) { if (budget <= 0) defer = true; else counted = true; }
    }
#ifdef MOR_EXP_STAMPS
    { const int nd = __popcll(__ballot(defer)); if (lane_id() == 0 && nd) RS_ADD(0, nd); }
#endif
    count_push(counted, d.pair_cnt + ko, pr);
    wl_push(defer, &d.wl2_n[s], d.wl2 + so, j, pr, target);
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
    bool defer = false, counted = false; int j = 0, pr = -1, target = -1;
#ifdef MOR_EXP_STAMPS
    bool dbg_budget = false, dbg_nocand = false;
    const unsigned long long g0 = wall_clock64(); unsigned long long g1 = g0, g2 = g0, g3 = g0, g4 = g0;
#endif
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
#ifdef MOR_EXP_STAMPS
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); g1 = wall_clock64();
#endif
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
#ifdef MOR_EXP_STAMPS
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); g2 = wall_clock64();
#endif
      if (ncand > 0) {
        const int ca[4] = {mc[0], mc[1], mc[2], mc[3]}, cb2[4] = {mc[4], mc[5], mc[6], mc[7]};
        scan_batch4(d, so, st, sp, ca, target, false, q, lbn, best, budget);
        if (ncand > 4 && best > d.pde_lb) scan_batch4(d, so, st, sp, cb2, target, false, q, lbn, best, budget);
      }
#ifdef MOR_EXP_STAMPS
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); g3 = wall_clock64();
#endif
      if (best > d.pde_lb) {
        if (budget <= 0) defer = true;
        else if (best < d.pde_ub) counted = true;   // all cells that can hold a point within √lb were among the slots
        else if (!(stencil27 && ncand <= 8)) defer = true;              // E2 still open: wider search
      }
#ifdef MOR_EXP_STAMPS
      dbg_budget = defer && budget <= 0; dbg_nocand = defer && ncand == 0;
#endif
    }
#ifdef MOR_EXP_STAMPS
    { const int n1 = __popcll(__ballot(dbg_budget)), n2 = __popcll(__ballot(defer && !dbg_budget)), n3 = __popcll(__ballot(dbg_nocand)); if (lane_id() == 0) { RS_ADD(1, n1); RS_ADD(2, n2); RS_ADD(3, n3); } }
#endif
    count_push(counted, d.pair_cnt + ko, pr);
    wl_push(defer, &d.wl2_n[s], d.wl2 + so, j, pr, target);
#ifdef MOR_EXP_STAMPS
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); g4 = wall_clock64();
    if (lane_id() == 0 && __ballot(w < nq)) { RS_ADD(4, 1); RS_ADD(5, g1 - g0); RS_ADD(6, g2 - g1); RS_ADD(7, g3 - g2); }
    (void)g4;
#endif
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
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
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
    {  // the query's own cell first
      const int c = cidx_find(I, cx, cy, cz);
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
      if (__shfl(lbrow, 0, 64) >= score_lim(best, lbn, d.pde_ub)) break;   // rows are ordered by their lower bound
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
          int c = __shfl(cand, l, 64);
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
    kk[a] = (long long)floor(((double)pc[a] - mn[a]) / res);
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

// ------------------------------------------------------------------------------------ G2: voxel-covariance ground removal (:90-200)
// Dead code in the reference (the call is commented out at :527 and would crash at :188); implemented with the
// intended semantics and the deterministic definitions of DESIGN.md §G2.  Pass A has trimmed the cloud in x/y
// and sorted it by VoxelGrid cell (stable ⇒ ascending point index inside a voxel).
#define G2_CAP 16384    // neighbours of one voxel centroid held in LDS as (d², index) keys (128 KiB of the CU's 160): big-voxel kernel
#define G2_SMALL 512    // … in the one-wave-per-voxel kernel (4 KiB: many workgroups per CU)
#define G2_CHUNK 1024   // coordinates staged per step of the ordered fp32 sums
// all trimmed points with d² < leaf² around q (radiusSearch, :125), appended to the LDS list in arbitrary order;
// the count keeps running beyond `cap` so the caller sees the overflow
__device__ __forceinline__ void g2_gather(const MorDev &d, int s, float4 q, unsigned long long *key, int *cnt, int cap) {
  const size_t so = (size_t)s * d.Nmax;
  const int *ckey = d.ckey + so, *rs = d.row_start + (size_t)s * (d.g.nrows + 1), *st = d.cstart + (size_t)s * (d.Nmax + 1);
  const MorGrid G = stream_grid(d, s);   // the lattice with the stream's own z layers
  int cx, cy, cz; bool cl; grid_cell(G, q, 0.f, d.zbase[s], cx, cy, cz, cl);
  for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) {
    const int y = cy + dy, z = cz + dz;
    if ((unsigned)y >= (unsigned)G.ny || (unsigned)z >= (unsigned)G.nz) continue;
    int lo, hi; row_cells(G, ckey, rs, max(cx - 1, 0), min(cx + 1, G.nx - 1), y, z, lo, hi);
    if (lo >= hi) continue;
    for (int k = st[lo] + threadIdx.x, e = st[hi]; k < e; k += blockDim.x) {
      const float4 p = d.sorted[so + k];
      const float dd = sqdist(q.x, q.y, q.z, p.x, p.y, p.z);
      if (dd < d.leaf_r2) {
        int slot = atomicAdd(cnt, 1);
        if (slot < cap) key[slot] = ((unsigned long long)__float_as_uint(dd) << 32) | (unsigned)__float_as_int(p.w);
      }
    }
  }
}
// Wave version (64-thread workgroups): lanes 0 … 8 resolve the nine (y,z) rows of the 3×3×3 voxel block in parallel —
// each row's three x-cells are one contiguous range of `sorted` — then the wave walks the concatenated candidates 64 at
// a time.  f(k, point) is called for every candidate within the radius.
template <class F> __device__ __forceinline__ void g2_for_neighbours(const MorDev &d, int s, const float4 &q, F f) {
  const size_t so = (size_t)s * d.Nmax;
  const int *ckey = d.ckey + so, *rs = d.row_start + (size_t)s * (d.g.nrows + 1), *st = d.cstart + (size_t)s * (d.Nmax + 1);
  const int lane = threadIdx.x & 63;
  const MorGrid G = stream_grid(d, s);
  int cx, cy, cz; bool cl; grid_cell(G, q, 0.f, d.zbase[s], cx, cy, cz, cl);
  int b0 = 0, len = 0;
  if (lane < 9) {
    const int y = cy + lane % 3 - 1, z = cz + lane / 3 - 1;
    if ((unsigned)y < (unsigned)G.ny && (unsigned)z < (unsigned)G.nz) {
      int lo, hi; row_cells(G, ckey, rs, max(cx - 1, 0), min(cx + 1, G.nx - 1), y, z, lo, hi);
      if (lo < hi) { b0 = st[lo]; len = st[hi] - b0; }
    }
  }
  int rb[9], rp[10]; rp[0] = 0;
#pragma unroll
  for (int r = 0; r < 9; ++r) { rb[r] = __shfl(b0, r, 64); rp[r + 1] = rp[r] + __shfl(len, r, 64); }
  for (int c = lane; c < rp[9]; c += 64) {
    int k = 0;
#pragma unroll
    for (int r = 0; r < 9; ++r) if (c >= rp[r] && c < rp[r + 1]) k = rb[r] + (c - rp[r]);
    const float4 p = d.sorted[so + k];
    const float dd = sqdist(q.x, q.y, q.z, p.x, p.y, p.z);
    if (dd < d.leaf_r2) f(dd, p);
  }
}
// Neighbours sorted by (d², index) as KdTreeFLANN::radiusSearch returns them; > 3 of them (:131); fp32 centroid (:142)
// and un-normalised scatter terms xz, yz, zz (:144) summed in that order (coordinates staged through LDS in chunks,
// one thread adds them up); an accepted voxel (:145) gets its z-bin (:166).  n = neighbours held in `key`.
template <int CHUNK> __device__ __forceinline__ int g2_voxel_bin(const MorDev &d, size_t so, const float4 &q, unsigned long long *key, int n, float *px, float *py, float *pz, float *acc) {
  int P = 4; while (P < n) P <<= 1;
  for (int i = n + threadIdx.x; i < P; i += blockDim.x) key[i] = ~0ull;
  __syncthreads();
  for (int k = 2; k <= P; k <<= 1) for (int j = k >> 1; j > 0; j >>= 1) {   // bitonic sort of the keys
    for (int i = threadIdx.x; i < P; i += blockDim.x) {
      int l = i ^ j;
      if (l > i) { bool up = (i & k) == 0; unsigned long long a = key[i], b = key[l]; if ((a > b) == up) { key[i] = b; key[l] = a; } }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { acc[0] = acc[1] = acc[2] = acc[3] = acc[4] = acc[5] = 0.f; }
  for (int pass = 0; pass < 2; ++pass) {   // pass 0: Σ p (centroid); pass 1: Σ (p−c) terms
    for (int c0 = 0; c0 < n; c0 += CHUNK) {
      const int m = min(CHUNK, n - c0);
      for (int i = threadIdx.x; i < m; i += blockDim.x) { float4 p = d.rawbuf[so + (int)(key[c0 + i] & 0xffffffffu)]; px[i] = p.x; py[i] = p.y; pz[i] = p.z; }
      __syncthreads();
      if (threadIdx.x < 3) {
        // Each sum is a serial chain by definition (fp32 adds in the neighbours' order), but the three sums of a pass are independent: lanes 0, 1, 2
        // of one wave run one chain each in lock step — x, y, z of the centroid, then the terms dz·dx, dy·dz, dz·dz.  Sixteen elements are loaded
        // ahead of the adds (the LDS latency is hidden, the add latency is what is left).
        const int t = threadIdx.x;
        const float *pa = t == 0 ? px : t == 1 ? py : pz;
        if (pass == 0) {
          float a = acc[t];
          int i = 0;
          for (; i + 16 <= m; i += 16) { float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = pa[i + u];
#pragma unroll
            for (int u = 0; u < 16; ++u) a += v[u]; }
          for (; i < m; ++i) a += pa[i];
          acc[t] = a;
        } else {
          const float ca = acc[t], cz = acc[2]; float a = acc[3 + t];   // lane 0: c02 = Σ dz·dx, lane 1: c12 = Σ dy·dz, lane 2: c22 = Σ dz·dz
          int i = 0;
          for (; i + 16 <= m; i += 16) { float va[16], vz[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) { va[u] = pa[i + u]; vz[u] = pz[i + u]; }
#pragma unroll
            for (int u = 0; u < 16; ++u) { const float da_ = va[u] - ca, dz = vz[u] - cz; a += (t == 1 ? da_ * dz : dz * da_); } }
          for (; i < m; ++i) { const float da_ = pa[i] - ca, dz = pz[i] - cz; a += (t == 1 ? da_ * dz : dz * da_); }
          acc[3 + t] = a;
        }
      }
      __syncthreads();
    }
    if (pass == 0 && threadIdx.x == 0) { const float fn = (float)n; acc[0] /= fn; acc[1] /= fn; acc[2] /= fn; }
    __syncthreads();
  }
  return ((double)fabsf(acc[3]) < 0.001 && (double)fabsf(acc[4]) < 0.001 && (double)fabsf(acc[5]) < 0.001) ? (int)(q.z * 10) : 0x7fffffff;
}
__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }   // keeps the compiler from moving LDS accesses across it; lanes of one wave then see each other's LDS writes
// Ordered fp32 sums (:142, :144) over coordinates laid out in rank order in LDS: Σp / n, then the scatter terms around it.  Each sum is a
// serial chain by definition, but the three sums of a pass are independent: lanes base, base + 1, base + 2 of the wave run one chain each
// in lock step (x, y, z of the centroid; then dz·dx, dy·dz, dz·dz), eight elements loaded ahead of the adds.  Called by ALL lanes of the
// wave (shuffles inside); `doit` and n are those of the lane's group; the verdict is valid in every lane of a group that did it.
__device__ __forceinline__ bool g2_ordered_sums3(const float *lx, const float *ly, const float *lz, int n, bool doit, int base) {
  const int t = lane_id() - base;
  const bool mine = doit && t >= 0 && t < 3;
  const float *pa = t == 0 ? lx : t == 1 ? ly : lz;
  float a = 0.f;
  if (mine) {
    int i = 0;
    for (; i + 8 <= n; i += 8) { float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = pa[i + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) a += v[u]; }
    for (; i < n; ++i) a += pa[i];
  }
  const float ca = a / (float)n;
  const float cz = __shfl(ca, (base + 2) & 63, 64);
  float c = 0.f;
  if (mine) {
    int i = 0;
    for (; i + 8 <= n; i += 8) { float va[8], vz[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { va[u] = pa[i + u]; vz[u] = lz[i + u]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { const float da = va[u] - ca, dz = vz[u] - cz; c += (t == 1 ? da * dz : dz * da); } }   // lane 0: dz·dx, lane 1: dy·dz, lane 2: dz·dz
    for (; i < n; ++i) { const float da = pa[i] - ca, dz = lz[i] - cz; c += (t == 1 ? da * dz : dz * da); }
  }
  const int ok = mine && (double)fabsf(c) < 0.001;
  return __shfl(ok, base & 63, 64) && __shfl(ok, (base + 1) & 63, 64) && __shfl(ok, (base + 2) & 63, 64);
}
// ---- The verdict of a voxel without its ordered sums.  What a voxel contributes is ONE BIT — all three scatter terms below 0.001 (:145) — and
// its z-bin; the fp32 sums in (d², index) order only matter when a term lies so close to 0.001 that the rounding of that very order decides.  So
// every tier first adds the terms up in fp64 in whatever order the lanes meet the neighbours (shifted by the voxel centroid q, one pass:
// Σa, Σc, Σa·c … with a = x − q.x; t = Σa·c − Σa·Σc / n), bounds how far the reference's fp32 evaluation can lie from that, and settles the voxel when
// the bound leaves no doubt; only the rest — none in the bench scenes — is sorted and summed in order.  The bound (u = 2⁻²⁴, γ_k = k·u / (1 − k·u)):
//   centroid, sequential fp32 sum and one division:  |c_ref − c| ≤ γ_n · X,  X ≥ max |x_i|                                   =: Δx
//   a term, two subtractions and a product:          |fl((z_i − cz_ref)·(x_i − cx_ref)) − (z_i − cz)(x_i − cx)| ≤ Δz·|a_i| + Δx·|c_i| + Δx·Δz + γ_3·(|a_i| + Δx)(|c_i| + Δz)
//   their sequential fp32 sum:                       ≤ γ_{n−1} · Σ (|a_i| + Δx)(|c_i| + Δz)
//   ⇒ |t_ref − t| ≤ (1 + γ)(Δz·Σ|a_i| + Δx·Σ|c_i| + n·Δx·Δz) + γ·Σ|a_i·c_i|,  γ = γ_{n+3}
// (the sums of absolute values around c are bounded through those around q).  The verdict is taken with TWICE that bound plus 1e-9 for the fp64 arithmetic here.
// (Sums of absolute values are not accumulated one by one: Σd² — the squared distances the radius test has just worked out — bounds them all: |a| ≤ (r + a²/r) / 2 gives
//  Σ|a_i| ≤ (n·r + Σd²/r) / 2 and |a·c| ≤ (a² + c²) / 2 gives Σ|a_i·c_i| ≤ Σd² / 2.  The sixteen-lane kernel's time IS this fp64 arithmetic — every candidate step pays for it as
//  soon as one lane has a hit, then the sums are reduced over the group —: seven additions per hit and eight reduced values instead of eleven and twelve.)
struct G2Acc { double Sa, Sb, Sc, Sac, Sbc, Scc, Sdd; int n; };
__device__ __forceinline__ void g2_acc_zero(G2Acc &A) { A.Sa = A.Sb = A.Sc = A.Sac = A.Sbc = A.Scc = A.Sdd = 0.0; A.n = 0; }
__device__ __forceinline__ void g2_acc_add(G2Acc &A, const float4 &q, const float4 &p, float dd /* sqdist(q, p) */) {
  const double a = (double)p.x - (double)q.x, b = (double)p.y - (double)q.y, c = (double)p.z - (double)q.z;   // exact: differences of two floats
  A.Sa += a; A.Sb += b; A.Sc += c; A.Sac += a * c; A.Sbc += b * c; A.Scc += c * c; A.Sdd += (double)dd;
  ++A.n;
}
template <int W> __device__ __forceinline__ void g2_acc_reduce(G2Acc &A) {   // over the W lanes of the caller's group (W = 16 or 64, aligned)
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) {
    A.Sa += __shfl_xor(A.Sa, o, 64); A.Sb += __shfl_xor(A.Sb, o, 64); A.Sc += __shfl_xor(A.Sc, o, 64);
    A.Sac += __shfl_xor(A.Sac, o, 64); A.Sbc += __shfl_xor(A.Sbc, o, 64); A.Scc += __shfl_xor(A.Scc, o, 64);
    A.Sdd += __shfl_xor(A.Sdd, o, 64); A.n += __shfl_xor(A.n, o, 64);
  }
}
// 1: accepted (:145 holds whatever the order), 0: rejected, −1: too close to call — the ordered sums decide.  n > 3.
__device__ __forceinline__ int g2_screen(const G2Acc &A, const float4 &q, double leaf_r /* √leaf² · 1.0001 + 1e-6, from the host */, double inv_r /* 1 / √leaf² */) {
  // (one division and no square root: sixteen lanes wait while one works this out for its group — r comes from the host, 1 / (1 − x) ≤ 1 + 2x for x ≤ ½)
  const double n = (double)A.n, u = 5.9604644775390625e-8, r = leaf_r;
  const double inv_n = 1.0 / n, ma = A.Sa * inv_n, mb = A.Sb * inv_n, mc = A.Sc * inv_n;   // centroid − q
  const double txz = A.Sac - A.Sa * mc, tyz = A.Sbc - A.Sb * mc, tzz = A.Scc - A.Sc * mc;
  const double fa = fabs(ma), fb = fabs(mb), fc = fabs(mc);
  const double sdd = A.Sdd * 1.000001 + 1e-12;   // ≥ Σ d_i² (the fp32 distances carry three roundings each)
  const double ab1 = 0.5 * (n * r + sdd * inv_r), ab2 = 0.5 * sdd;   // ≥ Σ|a_i|, Σ|b_i|, Σ|c_i|;  ≥ Σ|a_i·c_i|, Σ|b_i·c_i|
  const double sa = ab1 + n * fa, sb = ab1 + n * fb, sc = ab1 + n * fc;   // ≥ Σ|x_i − c| …
  const double axz = ab2 + fc * ab1 + fa * ab1 + n * fa * fc, ayz = ab2 + fc * ab1 + fb * ab1 + n * fb * fc, azz = A.Scc + 2.0 * fc * ab1 + n * fc * fc;   // ≥ Σ|(x_i − c)(z_i − c)| …
  if ((n + 4.0) * u > 0.25) return -1;   // (millions of neighbours: the bound says nothing any more)
  const double xg = (n + 4.0) * u, g = 1.01 * xg * (1.0 + 2.0 * xg);   // ≥ 1.01·γ_{n+4}
  const double Dx = g * (fabs((double)q.x) + r), Dy = g * (fabs((double)q.y) + r), Dz = g * (fabs((double)q.z) + r);
  const double Exz = (1.0 + g) * (Dz * sa + Dx * sc + n * Dx * Dz) + g * axz;
  const double Eyz = (1.0 + g) * (Dz * sb + Dy * sc + n * Dy * Dz) + g * ayz;
  const double Ezz = (1.0 + g) * (2.0 * Dz * sc + n * Dz * Dz) + g * azz;
  const double T = 0.001, tiny = 1e-9;
  const double lxz = fabs(txz) - 2.0 * Exz - tiny, lyz = fabs(tyz) - 2.0 * Eyz - tiny, lzz = fabs(tzz) - 2.0 * Ezz - tiny;   // lower bounds of |t_ref|
  if (lxz > T || lyz > T || lzz > T) return 0;
  const double hxz = fabs(txz) + 2.0 * Exz + tiny, hyz = fabs(tyz) + 2.0 * Eyz + tiny, hzz = fabs(tzz) + 2.0 * Ezz + tiny;   // upper bounds
  if (hxz < T && hyz < T && hzz < T) return 1;
  return -1;
}
// Sixteen lanes per voxel, sixteen voxels per 256-thread workgroup (a voxel centroid has a dozen neighbours on average, 96 %
// have ≤ 64): the group computes the voxel's centroid (dsc, :110-113 — fp32 sums in ascending point index, one lane), walks the points
// of the 3×3×3 voxel block (lanes 0–8 resolve the nine rows) and adds the neighbours within the radius into the screen's sums; the
// verdict is taken from those (above).  No LDS, no sort.  Queued for k_g2_cov_mid (a whole wave each): voxels with more than
// G2_NARROW_CAND candidates — dense surfaces next to the sensor, walked sixteen at a time they held their wave's other three groups up —
// and the voxels the screen could not settle (tagged: their ordered sums are due).
#define G2_NARROW_CAND 512
#define G2_Q_EXACT (1 << 30)   // queue entry: the screen has been through this voxel and left it to the ordered sums
#define G2_V_NONE 0x7fffffff   // bin word of a voxel without a bin (rejected, or ≤ 3 neighbours)
#define G2_COV_G 256   // workgroups per stream of k_g2_cov; 64 of the middle / big tiers, 128 of k_g2_mark
#ifdef MOR_EXP_STAMPS
#define G2_TICK(v) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long v = wall_clock64()
#else
#define G2_TICK(v)
#endif
__global__ __launch_bounds__(MOR_BT) void k_g2_cov(MorDev d) {
  int s, bxv; map_block(d.B, G2_COV_G, s, bxv);   // (a stream's workgroups on one XCD, as everywhere else: as a two-dimensional launch a stream's voxels went round all eight L2s)
  const int V = d.info[s].n_occ;
  const size_t so = (size_t)s * d.Nmax;
  const int grp = threadIdx.x >> 4, sub = threadIdx.x & 15, lane = lane_id();   // group in the workgroup, lane in the group
  const int *ckey = d.ckey + so, *rs = d.row_start + (size_t)s * (d.g.nrows + 1), *st = d.cstart + (size_t)s * (d.Nmax + 1);
  const float4 *sp = d.sorted + so;
  const int zbase = d.zbase[s]; const MorGrid G = stream_grid(d, s);   // (the lattice with the stream's own z layers)
  for (int v0 = bxv * (MOR_BT / 16); v0 < V; v0 += G2_COV_G * (MOR_BT / 16)) {
    const int v = v0 + grp; const bool act = v < V;
    G2_TICK(k0);
    // ---- voxel centroid: sequential fp32 sums over the voxel's points in ascending index (stable sort ⇒ storage order)
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (act && sub == 0) {
      float sx = 0.f, sy = 0.f, sz = 0.f; const int b0 = st[v], e0 = st[v + 1];
      for (int k = b0; k < e0; k += 8) {   // eight loads per round trip, the adds in index order
        float4 p[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) p[u] = sp[min(k + u, e0 - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) if (k + u < e0) { sx += p[u].x; sy += p[u].y; sz += p[u].z; }
      }
      const float n = (float)(e0 - b0);
      q = make_float4(sx / n, sy / n, sz / n, 0.f);
      d.vcent[so + v] = q;
    }
    q.x = __shfl(q.x, lane & 48, 64); q.y = __shfl(q.y, lane & 48, 64); q.z = __shfl(q.z, lane & 48, 64);
    G2_TICK(k1);
    // ---- the nine (y,z) rows of the 3×3×3 block: lanes 0 … 8 of the group, each row's three x-cells are one range of `sorted`
    int rb0 = 0, rlen = 0;
    if (act && sub < 9) {
      int cx, cy, cz; bool cl; grid_cell(G, q, 0.f, zbase, cx, cy, cz, cl);
      const int y = cy + sub % 3 - 1, z = cz + sub / 3 - 1;
      if ((unsigned)y < (unsigned)G.ny && (unsigned)z < (unsigned)G.nz) {
        int lo, hi; row_cells(G, ckey, rs, max(cx - 1, 0), min(cx + 1, G.nx - 1), y, z, lo, hi);
        if (lo < hi) { rb0 = st[lo]; rlen = st[hi] - rb0; }
      }
    }
    int rb[9], rp[10]; rp[0] = 0;
#pragma unroll
    for (int r = 0; r < 9; ++r) { rb[r] = __shfl(rb0, (lane & 48) + r, 64); rp[r + 1] = rp[r] + __shfl(rlen, (lane & 48) + r, 64); }
    G2_TICK(k2);
    // ---- walk: candidates sixteen at a time, four per lane and round trip; the hits go into the screen's sums
    const bool wide = rp[9] > G2_NARROW_CAND;   // (uniform in the group)
    const int ncand = wide ? 0 : rp[9];
    int wave_max = ncand;
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) wave_max = max(wave_max, __shfl_xor(wave_max, o, 64));
    G2Acc A; g2_acc_zero(A);
    for (int c0 = 0; c0 < wave_max; c0 += 64) {
      float4 pc[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + 16 * u + sub; int k = 0;
#pragma unroll
        for (int r = 0; r < 9; ++r) if (c >= rp[r] && c < rp[r + 1]) k = rb[r] + (c - rp[r]);
        pc[u] = sp[c < ncand ? k : 0];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + 16 * u + sub; const float4 p = pc[u];
        const float dd = sqdist(q.x, q.y, q.z, p.x, p.y, p.z);
        if (c < ncand && dd < d.leaf_r2) g2_acc_add(A, q, p, dd);
      }
    }
    G2_TICK(k3);
    g2_acc_reduce<16>(A);
    if (act && sub == 0) {
      const int verdict = wide ? -2 : A.n > 3 ? (d.g2_exact_only ? -1 : g2_screen(A, q, d.g2_r, d.g2_inv_r)) : 0;
      if (verdict < 0) d.g2_big[so + atomicAdd(&d.g2_nbig[s], 1)] = verdict == -1 ? (v | G2_Q_EXACT) : v;
      else d.vbin[so + v] = verdict ? (int)(q.z * 10) : 0x7fffffff;
    }
#ifdef MOR_EXP_STAMPS
    { G2_TICK(k4); if (lane == 0) { RS_ADD(0, k1 - k0); RS_ADD(1, k2 - k1); RS_ADD(2, k3 - k2); RS_ADD(3, k4 - k3); RS_ADD(4, 1); RS_ADD(5, wave_max); } if (act && sub == 0) { RS_ADD(6, 1); RS_ADD(7, A.n); RS_ADD(8, ncand); RS_ADD(9, wide); } }
#endif
  }
}
// The queued voxels, one WAVE per voxel: the same screen with sixty-four lanes for the voxels k_g2_cov did not walk; the ordered sums for those
// the screen leaves open, up to G2_MID_CAP neighbours in the wave's 20 KiB slice of LDS (gather with ballot compaction, rank by counting — the
// (d², index) keys are unique —, coordinates to their rank, ordered sums as three chains in three lanes).  Settled entries of the queue are
// complemented; what is left (open AND more than G2_MID_CAP neighbours) goes to k_g2_cov_big.  A wave works in its own slice of LDS: the
// order of ONE wave's LDS accesses — which the hardware keeps — is all its lanes need, not a workgroup barrier.
#define G2_MID_CAP 1024
__global__ __launch_bounds__(MOR_BT) void k_g2_cov_mid(MorDev d) {
  const int s = blockIdx.y + d.s0, nbig = d.g2_nbig[s], bxq = blockIdx.x, gq = gridDim.x;   // (spread over all XCDs: the queues are uneven across streams, and a workgroup holds 80 KB of LDS)
  const size_t so = (size_t)s * d.Nmax;
  const int wv = wave_id(), lane = lane_id();
  __shared__ unsigned long long l_key[MOR_BT / 64][G2_MID_CAP];
  __shared__ float l_x[MOR_BT / 64][G2_MID_CAP], l_y[MOR_BT / 64][G2_MID_CAP], l_z[MOR_BT / 64][G2_MID_CAP];
  const int *ckey = d.ckey + so, *rs = d.row_start + (size_t)s * (d.g.nrows + 1), *st = d.cstart + (size_t)s * (d.Nmax + 1);
  const float4 *sp = d.sorted + so;
  const int zbase = d.zbase[s]; const MorGrid G = stream_grid(d, s);   // (the lattice with the stream's own z layers)
  for (int w0 = bxq * (MOR_BT / 64); w0 < nbig; w0 += gq * (MOR_BT / 64)) {
   {
    const int w = w0 + wv;
    if (w >= nbig) continue;   // (wave-uniform; nothing below synchronises the workgroup)
    const int qe = d.g2_big[so + w], v = qe & ~G2_Q_EXACT;
    const float4 q = d.vcent[so + v];
    int rb0 = 0, rlen = 0;
    if (lane < 9) {
      int cx, cy, cz; bool cl; grid_cell(G, q, 0.f, zbase, cx, cy, cz, cl);
      const int y = cy + lane % 3 - 1, z = cz + lane / 3 - 1;
      if ((unsigned)y < (unsigned)G.ny && (unsigned)z < (unsigned)G.nz) {
        int lo, hi; row_cells(G, ckey, rs, max(cx - 1, 0), min(cx + 1, G.nx - 1), y, z, lo, hi);
        if (lo < hi) { rb0 = st[lo]; rlen = st[hi] - rb0; }
      }
    }
    int rb[9], rp[10]; rp[0] = 0;
#pragma unroll
    for (int r = 0; r < 9; ++r) { rb[r] = __shfl(rb0, r, 64); rp[r + 1] = rp[r] + __shfl(rlen, r, 64); }
    auto cand = [&](int c) { int k = 0;
#pragma unroll
      for (int r = 0; r < 9; ++r) if (c >= rp[r] && c < rp[r + 1]) k = rb[r] + (c - rp[r]);
      return k; };
    if (!(qe & G2_Q_EXACT) && !d.g2_exact_only) {   // not screened yet (too many candidates for sixteen lanes): 256 candidates per round trip
      G2Acc A; g2_acc_zero(A);
      for (int c0 = 0; c0 < rp[9]; c0 += 256) {
        float4 pc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int c = c0 + 64 * u + lane; pc[u] = sp[c < rp[9] ? cand(c) : 0]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int c = c0 + 64 * u + lane; const float4 p = pc[u]; const float dd = sqdist(q.x, q.y, q.z, p.x, p.y, p.z); if (c < rp[9] && dd < d.leaf_r2) g2_acc_add(A, q, p, dd); }
      }
      g2_acc_reduce<64>(A);
      const int verdict = A.n > 3 ? g2_screen(A, q, d.g2_r, d.g2_inv_r) : 0;
      if (verdict >= 0) {
        if (lane == 0) { d.vbin[so + v] = verdict ? (int)(q.z * 10) : G2_V_NONE; d.g2_big[so + w] = ~v; }
        continue;
      }
    }
    // ---- the ordered sums
    if (lane == 0) atomicAdd(&d.info[s].g2_exact, 1u);
    int n = 0;
    for (int c0 = 0; c0 < rp[9]; c0 += 64) {
      const int c = c0 + lane; bool hit = false; float dd = 0.f; float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < rp[9]) {
        p = sp[cand(c)];
        dd = sqdist(q.x, q.y, q.z, p.x, p.y, p.z);
        hit = dd < d.leaf_r2;
      }
      const unsigned long long m = __ballot(hit);
      if (hit) {
        const int slot = n + __popcll(m & lanemask_lt());
        if (slot < G2_MID_CAP) { l_key[wv][slot] = ((unsigned long long)__float_as_uint(dd) << 32) | (unsigned)__float_as_int(p.w); l_x[wv][slot] = p.x; l_y[wv][slot] = p.y; l_z[wv][slot] = p.z; }
      }
      n += __popcll(m);
    }
    wave_lds_fence();
    const bool mine = n <= G2_MID_CAP;
    if (!mine) { if (lane == 0) d.g2_big[so + w] = v; continue; }   // (the tags come off: k_g2_cov_big takes every entry ≥ 0)
    float ex[G2_MID_CAP / 64], ey[G2_MID_CAP / 64], ez[G2_MID_CAP / 64]; int er[G2_MID_CAP / 64];
#pragma unroll
    for (int u = 0; u < G2_MID_CAP / 64; ++u) {
      const int e = lane + 64 * u; er[u] = -1;
      if (e < n) {
        const unsigned long long ke = l_key[wv][e]; int r = 0;
        for (int j = 0; j < n; ++j) r += l_key[wv][j] < ke;
        er[u] = r; ex[u] = l_x[wv][e]; ey[u] = l_y[wv][e]; ez[u] = l_z[wv][e];
      }
    }
    wave_lds_fence();
#pragma unroll
    for (int u = 0; u < G2_MID_CAP / 64; ++u) if (er[u] >= 0) { l_x[wv][er[u]] = ex[u]; l_y[wv][er[u]] = ey[u]; l_z[wv][er[u]] = ez[u]; }
    wave_lds_fence();
    const bool acc3 = g2_ordered_sums3(l_x[wv], l_y[wv], l_z[wv], n, n > 3, 0);
    if (lane == 0) { d.vbin[so + v] = (n > 3 && acc3) ? (int)(q.z * 10) : G2_V_NONE; d.g2_big[so + w] = ~v; }
    wave_lds_fence();
   }
  }
}
// what the middle tier left: one 1024-thread workgroup each (the LDS lets only one live on a CU anyway: sixteen waves sort four times faster than
// four), up to G2_CAP neighbours in 128 KiB of LDS
#define G2_BIG_T 1024
__global__ __launch_bounds__(G2_BIG_T) void k_g2_cov_big(MorDev d) {
  const int s = blockIdx.y + d.s0, nbig = d.g2_nbig[s], bxq = blockIdx.x, gq = gridDim.x;   // (tried: two workgroups per stream shared out by the queues — the queue holds mostly entries the middle tier has settled, so a stream's few big voxels ended up behind each other in one workgroup: 4.3 ms)
  const size_t so = (size_t)s * d.Nmax;
  __shared__ unsigned long long key[G2_CAP];
  __shared__ float px[G2_CHUNK], py[G2_CHUNK], pz[G2_CHUNK];
  __shared__ int cnt;
  __shared__ float acc[6];
  for (int w = bxq; w < nbig; w += gq) {
    const int v = d.g2_big[so + w];
    if (v < 0) continue;   // settled by k_g2_cov_mid
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    const float4 q = d.vcent[so + v];
    g2_gather(d, s, q, key, &cnt, G2_CAP);
    __syncthreads();
    const int n = cnt;
    int bin = 0x7fffffff;
    if (n > G2_CAP) { if (threadIdx.x == 0) mor_raise(d, s, 16u); }
    else if (n > 3) bin = g2_voxel_bin<G2_CHUNK>(d, so, q, key, n, px, py, pz, acc);
    if (threadIdx.x == 0) d.vbin[so + v] = bin;
    __syncthreads();
  }
}
__global__ __launch_bounds__(MOR_BT) void k_g2_mode(MorDev d) {
  int s = blockIdx.x + d.s0, V = d.info[s].n_occ;
  const size_t so = (size_t)s * d.Nmax;
  __shared__ int hist[4096], best_cnt, best_bin;
  for (int i = threadIdx.x; i < 4096; i += MOR_BT) hist[i] = 0;
  if (threadIdx.x == 0) { best_cnt = 0; best_bin = 0x7fffffff; }
  __syncthreads();
  for (int v = threadIdx.x; v < V; v += MOR_BT) {
    int b = d.vbin[so + v];
    if (b == 0x7fffffff) continue;
    if (b < -2048 || b >= 2048) { mor_raise(d, s, 8u); continue; }
    atomicAdd(&hist[b + 2048], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 4096; i += MOR_BT) atomicMax(&best_cnt, hist[i]);
  __syncthreads();
  for (int i = threadIdx.x; i < 4096; i += MOR_BT) if (best_cnt > 0 && hist[i] == best_cnt) atomicMin(&best_bin, i - 2048);
  __syncthreads();
  if (threadIdx.x == 0) { d.mode_bin[s] = best_bin; d.g2_nbig[s] = 0; }   // (the queue of big voxels is empty again for the next frame on this copy)
}
// ground = union of the neighbour lists of the dominant bin's voxels (:184-191, de-duplicated): every trimmed point within the radius of such a
// voxel's centroid is marked (no list, no sort needed here).  Waves look at 64 voxels at a time and take the mode bin's voxels among them FOUR at a
// time, sixteen lanes each as in k_g2_cov (a centroid has 3.5 candidates: a whole wave per voxel — the first form — kept 55 lanes idle); voxels with
// more than G2_NARROW_CAND candidates are left to the whole wave afterwards.
__global__ __launch_bounds__(MOR_BT) void k_g2_mark(MorDev d) {
  int s, bxm; map_block(d.B, 128, s, bxm);
  const int V = d.info[s].n_occ, mode = d.mode_bin[s];
  if (mode == 0x7fffffff) return;
  const size_t so = (size_t)s * d.Nmax;
  const int lane = lane_id(), nw = 128 * (MOR_BT / 64), grp = lane >> 4, sub = lane & 15;
  const int *ckey = d.ckey + so, *rs = d.row_start + (size_t)s * (d.g.nrows + 1), *st = d.cstart + (size_t)s * (d.Nmax + 1);
  const float4 *sp = d.sorted + so;
  const MorGrid G = stream_grid(d, s);   // (the lattice with the stream's own z layers)
  const int zbase = d.zbase[s], tag = d.frame_no + 1;   // the frame's tag (never 0, never an earlier frame's on this copy of the array): nothing has to be cleared
  for (int v0 = (bxm * (MOR_BT / 64) + wave_id()) * 64; v0 < V; v0 += nw * 64) {
    unsigned long long m = __ballot(v0 + lane < V && d.vbin[so + min(v0 + lane, V - 1)] == mode);
    unsigned long long wide_m = 0;
    while (m) {
      // the group's voxel: the grp-th set bit of m; the four lowest bits leave m
      unsigned long long mm = m; int l = -1;
#pragma unroll
      for (int k = 0; k < 4; ++k) { if (mm) { if (k == grp) l = __ffsll((long long)mm) - 1; mm &= mm - 1; } }
      m = mm;
      const bool act = l >= 0;
      const float4 q = d.vcent[so + v0 + max(l, 0)];
      int rb0 = 0, rlen = 0;
      if (act && sub < 9) {
        int cx, cy, cz; bool cl; grid_cell(G, q, 0.f, zbase, cx, cy, cz, cl);
        const int y = cy + sub % 3 - 1, z = cz + sub / 3 - 1;
        if ((unsigned)y < (unsigned)G.ny && (unsigned)z < (unsigned)G.nz) {
          int lo, hi; row_cells(G, ckey, rs, max(cx - 1, 0), min(cx + 1, G.nx - 1), y, z, lo, hi);
          if (lo < hi) { rb0 = st[lo]; rlen = st[hi] - rb0; }
        }
      }
      int rb[9], rp[10]; rp[0] = 0;
#pragma unroll
      for (int r = 0; r < 9; ++r) { rb[r] = __shfl(rb0, (lane & 48) + r, 64); rp[r + 1] = rp[r] + __shfl(rlen, (lane & 48) + r, 64); }
      const bool wide = rp[9] > G2_NARROW_CAND;   // (uniform in the group)
      { const unsigned long long wb = __ballot(act && wide && sub == 0);   // one bit per group with a wide voxel: its voxel goes to the wave's list
        unsigned long long t = wb; while (t) { const int gl = __ffsll((long long)t) - 1; t &= t - 1; wide_m |= 1ull << __shfl(l, gl, 64); } }
      const int ncand = (act && !wide) ? rp[9] : 0;
      int wave_max = ncand;
#pragma unroll
      for (int o = 16; o < 64; o <<= 1) wave_max = max(wave_max, __shfl_xor(wave_max, o, 64));
      for (int c0 = 0; c0 < wave_max; c0 += 64) {   // four candidates per lane and round trip
        float4 pc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int c = c0 + 16 * u + sub; int k = 0;
#pragma unroll
          for (int r = 0; r < 9; ++r) if (c >= rp[r] && c < rp[r + 1]) k = rb[r] + (c - rp[r]);
          pc[u] = sp[c < ncand ? k : 0];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int c = c0 + 16 * u + sub; const float4 p = pc[u];
          if (c < ncand && sqdist(q.x, q.y, q.z, p.x, p.y, p.z) < d.leaf_r2) d.is_ground[so + __float_as_int(p.w)] = tag;
        }
      }
    }
    while (wide_m) {   // dense voxels next to the sensor: the whole wave walks their candidates
      const int l = __ffsll((long long)wide_m) - 1; wide_m &= wide_m - 1;
      const float4 q = d.vcent[so + v0 + l];
      g2_for_neighbours(d, s, q, [&](float, const float4 &p) { d.is_ground[so + __float_as_int(p.w)] = tag; });
    }
  }
}

// ------------------------------------------------------------------------------------ T1 + F1 tracking, on the device
// One workgroup (one wave) per stream.  The state is tiny (a few dozen clusters, pairs and tracked centroids) but the
// logic is sequential, so it is staged into LDS, run there (no chain of global-memory round trips) and written back.
// Streams whose vectors exceed the LDS slots run the same code on the global arrays.
#define TRK 384   // clusters / pairs per window slot held in LDS
// The head of a stream's tracking state (counts, window sizes) lives in LDS while a tracking kernel works on it; the tracked
// centroids themselves (up to MOR_TR_MAXT of them: 640 KB) stay in global memory: both kernels stream through them once per frame.
// (The whole struct in LDS made these 64-thread workgroups wait for a CU with 117 KB of LDS free.)
struct MorTrackHead { int n_mo, n_corr, n_res, has_cur, K_last, overflow, pad0, pad1; int corr_n[MOR_TR_NB], res_n[MOR_TR_NB + 1]; };
static_assert(sizeof(MorTrackHead) == offsetof(MorTrackDev, mo_c), "MorTrackHead is the prefix of MorTrackDev");
__device__ __forceinline__ void tr_load_head(const MorTrackDev &g, MorTrackHead &l, int lane) {
  const int *gs = reinterpret_cast<const int *>(&g); int *ls = reinterpret_cast<int *>(&l);
  for (int i = lane; i < (int)(sizeof(MorTrackHead) / sizeof(int)); i += 64) ls[i] = gs[i];
  __syncthreads();
}
__device__ __forceinline__ void tr_store_head(MorTrackDev &g, const MorTrackHead &l, int lane) {
  __syncthreads();
  int *gs = reinterpret_cast<int *>(&g); const int *ls = reinterpret_cast<const int *>(&l);
  for (int i = lane; i < (int)(sizeof(MorTrackHead) / sizeof(int)); i += 64) gs[i] = ls[i];
}
// checkMovingClusterChain (:478-514) with recurseFindClusterChain (:415-453) and pushCentroid (:455-476)
__global__ __launch_bounds__(64) void k_track_push(MorDev d) {
  const int s = blockIdx.x + d.s0, lane = threadIdx.x;
  constexpr int TRKN = TRK;
  decide_body<64>(d, s);   // P5: thresholds, detection_results, host summary
  __threadfence_block();
  __syncthreads();
  const int K = d.info[s].K, np = d.has_prev ? (int)d.info[s].n_pairs : -1;
  const size_t ko = (size_t)s * d.Kcap;
  __shared__ MorTrackHead t;
  __shared__ int2 l_corr[MOR_TR_NB][TRKN];
  __shared__ unsigned char l_res[MOR_TR_NB + 1][TRKN];
  __shared__ float4 l_cand[TRKN], l_acc[TRKN];   // centroids found at the end of a chain this frame; those of them already appended
  __shared__ unsigned char l_cnear[TRKN];
  MorTrackDev &gt = d.tr[s];
  tr_load_head(gt, t, lane);
  int2 *g_corr = d.tr_corr + (size_t)s * MOR_TR_NB * d.Kcap;
  unsigned char *g_res = d.tr_res + (size_t)s * (MOR_TR_NB + 1) * d.Kcap, *last = d.tr_lastdet + ko;
  bool fits = K <= TRKN && t.K_last <= TRKN && np <= TRKN;
  for (int c = 0; c < t.n_corr; ++c) fits = fits && t.corr_n[c] <= TRKN;
  for (int r = 0; r < t.n_res; ++r) fits = fits && t.res_n[r] <= TRKN;
  int2 *corr = g_corr; unsigned char *res = g_res; int stride = d.Kcap;
  if (fits) {   // stage the window
    for (int c = 0; c < t.n_corr; ++c) for (int j = lane; j < t.corr_n[c]; j += 64) l_corr[c][j] = g_corr[(size_t)c * d.Kcap + j];
    for (int r = 0; r < t.n_res; ++r) for (int k = lane; k < t.res_n[r]; k += 64) l_res[r][k] = g_res[(size_t)r * d.Kcap + k];
    corr = &l_corr[0][0]; res = &l_res[0][0]; stride = TRKN;
    __syncthreads();
  }
  const bool chain = np >= 0 && t.has_cur;
  if (chain) {
    const int cs_ = t.n_corr, rs0 = t.n_res;
    for (int j = lane; j < np; j += 64) corr[(size_t)cs_ * stride + j] = make_int2(d.pair_q[ko + j], d.pair_m[ko + j]);   // corrs_vec.push_back(mp) :483
    if (rs0 == 0) for (int k = lane; k < t.K_last; k += 64) res[k] = last[k];                                              // res_vec.push_back(res_ca) :484-488
    const int rs1 = rs0 == 0 ? 1 : rs0;
    for (int k = lane; k < K; k += 64) res[(size_t)rs1 * stride + k] = d.det[ko + k];                                     // res_vec.push_back(res_cb) :490
    __syncthreads();
    if (lane == 0) {
      t.corr_n[cs_] = np; t.n_corr = cs_ + 1;
      if (rs0 == 0) t.res_n[0] = t.K_last;
      t.res_n[rs1] = K; t.n_res = rs1 + 1;
    }
    __syncthreads();
    if (t.n_res >= d.moving_confidence) {                                                                                  // :492
      // the outer loop (clusters flagged in the oldest frame, in index order) is sequential — the order decides which
      // centroid wins inside catch_up_distance — but every inner search runs across the 64 lanes
      const int n0 = t.res_n[0], ncol = t.n_corr;
      // pushCentroid (:455-476) appends a centroid unless a tracked one lies within catch_up_distance — of those tracked before
      // this frame or appended earlier in it (the order of the flagged clusters decides which of two close ones wins).  So: collect
      // the chain ends in order; ONE pass over the tracked centroids marks the candidates that have an old neighbour (the tracks stream
      // from global memory once, four per lane and round trip — scanning them per candidate made a stream with 15 000 tracks take
      // milliseconds); then the candidates go through in order against the few appended before them.
      int i = 0;
      while (i < n0) {
        int nc_ = 0;
        for (; i < n0 && nc_ < TRKN; ++i) {
          if (!res[i]) continue;
          int track = i; bool ok = true;
          for (int col = 0; col < ncol && ok; ++col) {                                                                       // recurseFindClusterChain
            const int2 *c = corr + (size_t)col * stride; const int n = t.corr_n[col]; int match = -1;
            for (int j0 = 0; j0 < n && match < 0; j0 += 64) {
              const int j = j0 + lane; const int2 pr = j < n ? c[j] : make_int2(-1, -1);
              unsigned long long m = __ballot(pr.x == track);
              if (m) match = __shfl(pr.y, __ffsll((long long)m) - 1, 64);   // first pair whose query is `track`
            }
            if (match < 0 || !res[(size_t)(col + 1) * stride + match]) ok = false; else track = match;
          }
          if (!ok) continue;
          if (lane == 0) { l_cand[nc_] = d.centroid[d.cur][ko + track]; l_cnear[nc_] = 0; }                                  // pushCentroid(cb->centroid_collection[found])
          ++nc_;
        }
        __syncthreads();
        const int nm = t.n_mo;
        const float (*mc)[3] = gt.mo_c;
        for (int m0 = 0; m0 < nm; m0 += 256) {
          float tx[4], ty[4], tz[4]; bool tv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) { const int m = m0 + 64 * u + lane; tv[u] = m < nm; const int mm = min(m, nm - 1); tx[u] = mc[mm][0]; ty[u] = mc[mm][1]; tz[u] = mc[mm][2]; }
          for (int j = 0; j < nc_; ++j) {
            if (l_cnear[j]) continue;
            const float4 pt = l_cand[j]; bool hit = false;
#pragma unroll
            for (int u = 0; u < 4; ++u) { const double dx = (double)(pt.x - tx[u]), dy = (double)(pt.y - ty[u]), dz = (double)(pt.z - tz[u]); hit |= tv[u] && sqrt(dx * dx + dy * dy + dz * dz) < (double)d.catch_up; }
            if (__ballot(hit) && lane == 0) l_cnear[j] = 1;
          }
          __syncthreads();
        }
        int na = 0;
        for (int j = 0; j < nc_; ++j) {
          if (l_cnear[j]) continue;
          const float4 pt = l_cand[j]; bool near = false;
          for (int a0 = 0; a0 < na && !near; a0 += 64) {
            const int a = a0 + lane; bool hit = false;
            if (a < na) { const float4 q = l_acc[a]; const double dx = (double)(pt.x - q.x), dy = (double)(pt.y - q.y), dz = (double)(pt.z - q.z); hit = sqrt(dx * dx + dy * dy + dz * dz) < (double)d.catch_up; }
            near = __ballot(hit) != 0ull;
          }
          if (near) continue;
          const int nmc = t.n_mo;
          if (nmc >= MOR_TR_MAXT) { if (lane == 0) { t.overflow = 1; mor_raise(d, s, 32u); } }
          else {
            if (lane == 0) { gt.mo_c[nmc][0] = pt.x; gt.mo_c[nmc][1] = pt.y; gt.mo_c[nmc][2] = pt.z; gt.mo_conf[nmc] = gt.mo_max[nmc] = d.static_confidence + 1; t.n_mo = nmc + 1; l_acc[na] = pt; }   // header :91
            ++na;
          }
          __syncthreads();
        }
        __threadfence_block();   // appended centroids: visible to the pass of the next batch — the same wave (one-wave workgroup), so workgroup scope will do: an agent-scope fence writes back the XCD's whole L2, 64 times per step here (the later kernels see them by the kernel boundary)
        __syncthreads();
      }
    }
    __syncthreads();
    if (t.n_res >= d.moving_confidence) {   // pop_front of both deques (:511-512): shift the slots down
      const int nc = t.n_corr, nr = t.n_res;
      for (int col = 1; col < nc; ++col) { for (int j = lane; j < t.corr_n[col]; j += 64) corr[(size_t)(col - 1) * stride + j] = corr[(size_t)col * stride + j]; __syncthreads(); }
      for (int r = 1; r < nr; ++r) { for (int k = lane; k < t.res_n[r]; k += 64) res[(size_t)(r - 1) * stride + k] = res[(size_t)r * stride + k]; __syncthreads(); }
      if (lane == 0) {
        for (int col = 1; col < nc; ++col) t.corr_n[col - 1] = t.corr_n[col];
        for (int r = 1; r < nr; ++r) t.res_n[r - 1] = t.res_n[r];
        t.n_corr = nc - 1; t.n_res = nr - 1;
      }
    }
    __syncthreads();
    if (fits) {   // write the window back
      for (int c = 0; c < t.n_corr; ++c) for (int j = lane; j < t.corr_n[c]; j += 64) g_corr[(size_t)c * d.Kcap + j] = l_corr[c][j];
      for (int r = 0; r < t.n_res; ++r) for (int k = lane; k < t.res_n[r]; k += 64) g_res[(size_t)r * d.Kcap + k] = l_res[r][k];
    }
  }
  __syncthreads();
  for (int k = lane; k < K; k += 64) last[k] = d.det[ko + k];
  if (lane == 0) { t.K_last = K; t.has_cur = 1; d.h_log[(size_t)(d.frame_no % MOR_LOG_CAP) * d.Btot + s].n_mo_push = t.n_mo; }
  tr_store_head(gt, t, lane);
  if (lane == 0) mor_publish_err(d, s);
}
// filterCloud (:613-696) in two launches (round 3: k_track_filter | k_out_count | k_out_scatter).
// k_track_filter — the loop over mo_vec (:630-671): nearest current centroid of every tracked one (squared fp32 distance, ties → lowest index), its
//     whole cluster queued for removal before any test, confidence bookkeeping.  One workgroup per stream; it leaves the removal flags as a bit per
//     cluster, the ExtractIndices size-check flag and the number of kept cloud points, n_keep = M − Σ sizes of the flagged clusters (known without a
//     counting pass over the points: a cluster's size is the number of cloud points that carry its label).
// k_out — the output (:673-687) = [cloud minus moving clusters, original order] ++ [ground points].  The ground points were written to their final
//     place by the split kernel (from slot Nmax of the stream's 2·Nmax-slot `ground` buffer), so the result is assembled in place: the kept
//     cloud points go right-aligned in front of them, the result starts at slot Nmax − n_keep, and the bulk of the frame (the ground, ≈ 90 %
//     of a LiDAR sweep) is not copied again.  ONE pass over the labels: tiles of 2048 cloud points are handed out by ticket (a tile's
//     predecessors are then owned by workgroups that are already running), a tile publishes its kept count in a descriptor tagged with the
//     call's epoch and adds up the descriptors of the tiles below it (decoupled look-back; no count pass, no scan).  With caller-provided device
//     pointers both parts are copied out: the workgroups [tiles_m, tiles_m + tiles) of a stream copy the ground points behind the kept ones.
// (Both in ONE launch — the first workgroup of a stream to arrive runs the loop, the others poll a ready word — was correct and 5 % slower: a
//  stream's two thousand output workgroups sat in the GPU's wave slots spinning while one wave walked the tracks, and kept the other lanes' kernels out.)
// The keep test is ExtractIndices' negative set semantics; the size-check flag reproduces "more indices than points ⇒ empty output" (:676-678).
#define FLT_T MOR_BT
__device__ __forceinline__ void track_filter_body(const MorDev &d, int s, unsigned *l_mov /* Kcap / 32 words */) {
  const int K = d.info[s].K, tid = threadIdx.x, lane = tid & 63;
  const bool w0 = tid < 64;   // the loop itself is the work of one wave (as a kernel of its own it was a 64-thread workgroup); the other waves help with the tables and keep the barriers
  const size_t ko = (size_t)s * d.Kcap;
  const int *off = d.cl_off[d.cur] + (size_t)s * (d.Kcap + 1);
  __shared__ MorTrackHead t;
  __shared__ float4 l_cen[TRK];
  __shared__ int l_size[TRK];
  __shared__ unsigned char l_det[TRK];
  __shared__ unsigned long long l_tot;
  MorTrackDev &gt = d.tr[s];   // the tracked centroids are read once and written once (compacted in place): straight from / to global memory
  if (w0) { const int *gs = reinterpret_cast<const int *>(&gt); int *ls = reinterpret_cast<int *>(&t); for (int i = lane; i < (int)(sizeof(MorTrackHead) / sizeof(int)); i += 64) ls[i] = gs[i]; }
  const bool fits = K <= TRK;
  for (int k = tid; k < (d.Kcap + 31) / 32; k += FLT_T) l_mov[k] = 0u;
  if (fits) for (int k = tid; k < K; k += FLT_T) { l_cen[k] = d.centroid[d.cur][ko + k]; l_size[k] = off[k + 1] - off[k]; l_det[k] = d.det[ko + k]; }
  if (tid == 0) l_tot = 0ull;
  __syncthreads();
  // Every tracked centroid is handled independently of the others (its nearest cluster, its confidence, its own new
  // position); erasing only compacts the vector, order kept.  So: one lane per track, four tracks per lane and round trip (a stream
  // of the bench reaches 15 000 tracked centroids on long runs), survivors compacted in place with a ballot prefix (reads of a round
  // happen before its writes, and writes never pass reads).
  const int n_mo = t.n_mo;
  if (w0 && K > 0) {
    unsigned long long total = 0; int n_keep = 0;
    for (int i0 = 0; i0 < n_mo; i0 += 256) {
      float c0[4], c1[4], c2[4]; int conf[4], mx[4]; bool keep[4]; unsigned long long mine = 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int i = min(i0 + 64 * u + lane, n_mo - 1); c0[u] = gt.mo_c[i][0]; c1[u] = gt.mo_c[i][1]; c2[u] = gt.mo_c[i][2]; conf[u] = gt.mo_conf[i]; mx[u] = gt.mo_max[i]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        keep[u] = false;
        if (i0 + 64 * u + lane < n_mo) {
          float bd = INFINITY; int bi = 0;
          for (int k = 0; k < K; ++k) { const float4 c = fits ? l_cen[k] : d.centroid[d.cur][ko + k]; float dd = sqdist(c0[u], c1[u], c2[u], c.x, c.y, c.z); if (dd < bd) { bd = dd; bi = k; } }   // ties → lowest index
          atomicOr(&l_mov[bi >> 5], 1u << (bi & 31));                        // whole cluster queued for removal before any test (:644-648)
          d.tr_match[(size_t)s * (MOR_TR_MAXT + 1) + 1 + i0 + 64 * u + lane] = bi;   // (the marker the reference publishes for this tracked centroid, :641)
          mine += (unsigned long long)(fits ? l_size[bi] : off[bi + 1] - off[bi]);
          if (!(fits ? l_det[bi] : d.det[ko + bi]) || bd > d.leave_off) {    // squared vs un-squared: reference quirk kept (:650)
            keep[u] = --conf[u] != 0;                                        // erased at confidence 0 (:655-660)
          } else {
            const float4 c = fits ? l_cen[bi] : d.centroid[d.cur][ko + bi];
            c0[u] = c.x; c1[u] = c.y; c2[u] = c.z;                           // :664
            if (conf[u] < mx[u]) ++conf[u];                                  // :667
            keep[u] = true;
          }
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mine += ((unsigned long long)(unsigned)__shfl_xor((int)(mine >> 32), o, 64) << 32) | (unsigned)__shfl_xor((int)(unsigned)mine, o, 64);
      total += mine;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (all reads of this round are done — one wave: no barrier needed; survivors are compacted in place, order kept)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned long long km = __ballot(keep[u]);
        if (keep[u]) { const int o = n_keep + __popcll(km & lanemask_lt()); gt.mo_c[o][0] = c0[u]; gt.mo_c[o][1] = c1[u]; gt.mo_c[o][2] = c2[u]; gt.mo_conf[o] = conf[u]; gt.mo_max[o] = mx[u]; }
        n_keep += __popcll(km);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (lane == 0) { t.n_mo = n_keep; l_tot = total; }
  }
  __syncthreads();
  // ---- hand-over: removal bits, size-check flag (ExtractIndices: more indices than points ⇒ error, empty output), kept cloud points
  const unsigned M = d.info[s].M;
  const bool xerr = l_tot > (unsigned long long)M;
  unsigned removed = 0;
  for (int k = tid; k < K; k += FLT_T) if ((l_mov[k >> 5] >> (k & 31)) & 1u) removed += (unsigned)(fits ? l_size[k] : off[k + 1] - off[k]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) removed += (unsigned)__shfl_xor((int)removed, o, 64);
  __shared__ unsigned l_rem[FLT_T / 64];
  if (lane == 0) l_rem[tid >> 6] = removed;
  __syncthreads();
  unsigned *gm = d.moving + (size_t)s * (d.Kcap / 32 + 2);
  for (int k = tid; k < (K + 31) / 32; k += FLT_T) gm[k] = l_mov[k];
  if (tid == 0) {
    unsigned rem = 0; for (int w = 0; w < FLT_T / 64; ++w) rem += l_rem[w];
    const unsigned n_keep = xerr ? 0u : M - rem;
    gm[d.Kcap / 32] = xerr ? 1u : 0u; gm[d.Kcap / 32 + 1] = n_keep;
    d.info[s].n_keep = n_keep; d.h_nout[s] = (unsigned long long)n_keep + d.info[s].G; d.h_noff[s] = d.Nmax - (int)n_keep;
    MorFrameLog &L = d.h_log[(size_t)(d.frame_no % MOR_LOG_CAP) * d.Btot + s];
    L.n_mo_filter = t.n_mo; L.n_out = (unsigned long long)n_keep + d.info[s].G;
    d.tr_match[(size_t)s * (MOR_TR_MAXT + 1)] = K > 0 ? n_mo : 0;
    mor_publish_err(d, s);
  }
  if (w0) { int *gs = reinterpret_cast<int *>(&gt); const int *ls = reinterpret_cast<const int *>(&t); for (int i = lane; i < (int)(sizeof(MorTrackHead) / sizeof(int)); i += 64) gs[i] = ls[i]; }
}
__global__ __launch_bounds__(FLT_T) void k_track_filter(MorDev d) {
  __shared__ unsigned l_mov[MOR_KCAP_MAX / 32];
  track_filter_body(d, blockIdx.x + d.s0, l_mov);
}
__global__ __launch_bounds__(FLT_T) void k_out(MorDev d) {
  // the launch: B·g_out workgroups for the kept cloud points (shared out by the streams' tile counts; tiles go by ticket, so any share is correct), then,
  // with caller-provided pointers, B·tiles workgroups that copy the ground points
  const int n_cloud = d.B * d.g_out;
  const bool ground_wg = (int)blockIdx.x >= n_cloud;
  int s, t2, gs = 0;
  if (ground_wg) { int L = (int)blockIdx.x - n_cloud; s = L / d.tiles + d.s0; t2 = L % d.tiles; }
  else if (!map_block_work(d, [&](int s_) { return ((int)d.info[s_].M + MOR_TILE - 1) / MOR_TILE; }, s, t2, gs, n_cloud, (int)blockIdx.x)) return;
  const size_t so = (size_t)s * d.Nmax;
  float4 *og = d.ground + 2 * so;
  __shared__ unsigned l_mov[MOR_KCAP_MAX / 32];
  __shared__ int l_ex[4], sh[4];
  const unsigned epoch = d.filter_epoch;
  int *tk = d.tickets + (size_t)s * TK_COUNT + TK_OUT;
  const int M = d.info[s].M, nto = (M + MOR_TILE - 1) / MOR_TILE, tk_total = nto + gs;   // every cloud workgroup draws one ticket beyond its last tile
  int t = 0;
  if (!ground_wg) {
    if (threadIdx.x == 0) { const int v = __hip_atomic_fetch_add(tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); l_ex[0] = v; if (v + 1 == tk_total) __hip_atomic_store(tk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __syncthreads();
    t = __builtin_amdgcn_readfirstlane(l_ex[0]);
  }
  const unsigned *gm = d.moving + (size_t)s * (d.Kcap / 32 + 2);
  const int xerr = (int)gm[d.Kcap / 32], n_keep = (int)gm[d.Kcap / 32 + 1];   // (k_track_filter's results: a kernel boundary lies in between)
  if (ground_wg) {
    const int tg = t2, G = d.info[s].G, base = tg * MOR_TILE;
    float4 *out = d.out_ptrs[s];
    for (int i = base + threadIdx.x; i < min(base + MOR_TILE, G); i += FLT_T) st_stream(&out[n_keep + i], ld_stream(&og[d.Nmax + i]));
    return;
  }
  if (t >= nto) return;
  const int K = d.info[s].K;
  for (int k = threadIdx.x; k < (K + 31) / 32; k += FLT_T) l_mov[k] = gm[k];
  __syncthreads();
  float4 *dst = d.out_ptrs ? d.out_ptrs[s] : og + (d.Nmax - n_keep);
  unsigned long long *desc = d.out_desc + (size_t)s * d.tiles_max;
  int t_prev = -1, ex = 0;   // this workgroup's previous tile and the kept points up to and including it
  while (t < nto) {
    const int base = t * MOR_TILE + wave_id() * 512; int c = 0;
    unsigned long long mk[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int i = base + it * 64 + lane_id();
      bool keep = false;
      if (i < M && !xerr) { const int cid = ld_stream(&d.pcid[so + i]); keep = !(cid >= 0 && ((l_mov[cid >> 5] >> (cid & 31)) & 1u)); }   // cluster id per cloud point (written by k_clusters)
      mk[it] = __ballot(keep); c += __popcll(mk[it]);
    }
    if (lane_id() == 0) sh[wave_id()] = c;
    __syncthreads();
    const int tot = sh[0] + sh[1] + sh[2] + sh[3];
    if (threadIdx.x == 0) {
      __hip_atomic_store(desc + t, ((unsigned long long)epoch << 32) | (unsigned long long)(unsigned)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int v = __hip_atomic_fetch_add(tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the next tile of this workgroup (its loads come after the look-back: tiles are short)
      l_ex[1] = v; if (v + 1 == tk_total) __hip_atomic_store(tk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (wave_id() == 0) {   // look-back over the tiles between this workgroup's previous tile and this one
      int an = 0;
      for (int hi = t - 1; hi > t_prev; hi -= 64) {
        const int u = hi - lane_id();
        if (u > t_prev) {
          unsigned spins = 0;
          for (;;) {
            const unsigned long long v = ld_agent64(&desc[u]);
            if ((unsigned)(v >> 32) == epoch) { an += (int)(unsigned)v; break; }
            if (++spins > SPLIT_SPIN_LIMIT) { mor_raise(d, s, 64u); break; }
            __builtin_amdgcn_s_sleep(1);
          }
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) an += __shfl_xor(an, o, 64);
      if (lane_id() == 0) l_ex[2] = ex + an;
    }
    __syncthreads();
    int r = l_ex[2];
    ex = r + tot; t_prev = t;
    for (int w = 0; w < wave_id(); ++w) r += sh[w];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int i = base + it * 64 + lane_id();
      if ((mk[it] >> lane_id()) & 1ull) st_stream(&dst[r + __popcll(mk[it] & lanemask_lt())], ld_stream(&d.cloud[so + i]));   // (the filtered cloud is the caller's; the cloud is not read again on the device)
      r += __popcll(mk[it]);
    }
    t = __builtin_amdgcn_readfirstlane(l_ex[1]);
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------ launch sequences
static int mor_exp_dup() { static const int v = getenv("MOR_EXP_DUP") ? atoi(getenv("MOR_EXP_DUP")) : -1; return v; }   // experiment: launch kernel <id> twice (idempotent kernels only)
#define MOR_LAUNCH_T(id, kern, grid, threads, ...)                        \
  do {                                                                    \
    mor_timer_begin(tm, id, st);                                          \
    hipLaunchKernelGGL(kern, grid, dim3(threads), 0, st, __VA_ARGS__);    \
    if (mor_exp_dup() == (int)(id)) hipLaunchKernelGGL(kern, grid, dim3(threads), 0, st, __VA_ARGS__); \
    mor_timer_end(tm, id, st);                                            \
  } while (0)
#define MOR_LAUNCH(id, kern, grid, ...) MOR_LAUNCH_T(id, kern, grid, MOR_BT, __VA_ARGS__)

// part: 0 = both, 1 = the split only, 2 = the grid build only (the lane schedule runs them as two pieces)
static void mor_launch_split_and_grid(const MorDev &d, hipStream_t st, MorLaunchTimer *tm, int part = 0) {
  const dim3 gT(d.B * d.tiles), gM(d.B * d.tiles_m), gB(d.B);
  if (part == 2) goto grid;
  if (d.gmode != 1 && !d.two_pass_split) {
    if (d.gmode == 2) MOR_LAUNCH(MK_SPLIT, k_split<true>, dim3(d.B * d.sp_g), d);   // (pass B of the voxel ground variant: the records carry a ground flag; two instances so that the crop variant does not keep registers for it)
    else MOR_LAUNCH(MK_SPLIT, k_split<false>, dim3(d.B * d.sp_g), d);
  } else {
    const dim3 gS(d.B * d.split_g);
    MOR_LAUNCH(MK_CLASSIFY, k_classify, gS, d);
    MOR_LAUNCH(MK_SCATTER, k_scatter, gS, d);
  }
  if (part == 1) return;
grid:
  if (d.gmode != 1) {   // clustering grid by counting (k_gridhash); the VoxelGrid pass of the voxel ground variant needs the points of a voxel in index order: sort
    MOR_LAUNCH_T(MK_GRIDCOUNT, k_gridcount, dim3(d.B * d.gc_P), GC_T, d);
    MOR_LAUNCH_T(MK_GRIDHASH, k_gridhash, gB, GH_T, d);
    MOR_LAUNCH_T(MK_GRIDPLACE, k_gridplace, dim3(d.B * d.gc_P), GC_T, d);
  } else {
    for (int pass = 0; pass < d.cell_passes; ++pass) {   // points sorted by cell key; result in (skey, sidx) = buffers [cell_passes & 1]
      MorRadix j = {pass == 0 ? d.pkey : d.rkeys[pass & 1], pass == 0 ? nullptr : d.rvals[pass & 1], d.rkeys[(pass + 1) & 1], d.rvals[(pass + 1) & 1], 8 * pass, 0, 0, nullptr, d.rhist, 0, d.tiles_m <= 64, 1, 1};   // (a stream's last pass — by its own key width — leaves the inverse permutation: k_heads_scatter moves the points)
      MOR_LAUNCH(MK_RHIST, k_rhist, gM, d, j);
      if (!j.fuse) MOR_LAUNCH(MK_RSCAN, k_rscan, gB, d, j);
      MOR_LAUNCH(MK_RSCATTER, k_rscatter, gM, d, j);
    }
    MOR_LAUNCH(MK_HEADS_COUNT, k_heads_count, gM, d);
    MOR_LAUNCH(MK_HEADS_SCATTER, k_heads_scatter, gM, d);
  }
}
static void mor_launch_boxes(const MorDev &d, hipStream_t st, MorLaunchTimer *tm) {
  MOR_LAUNCH(MK_CELLBOXES, k_cellboxes, dim3(d.g_box * d.B), d);   // point boxes, smallest index, exact coordinate sums of every cell
}

// Voxel-covariance ground removal as six sub-pieces (the lane schedule runs sub-piece q of frame k beside other sub-pieces of
// frames k ± 1; as ONE piece its 7 ms were the period of the whole pipeline): pass A (trim, VoxelGrid sort), the 16-lane voxel
// kernel, the wave-per-voxel kernel, the big voxels, mode + marking, pass B (split by ground flag + clustering grid).
static void mor_launch_grid_sub(const MorDev &d, int sub, hipStream_t st, MorLaunchTimer *tm) {
  const dim3 gB(d.B);
  MorDev da = d; da.gmode = 1; da.g = d.gv; da.cloud = d.rawbuf; da.cell_passes = d.voxel_passes; da.tiles_m = d.tiles; da.use_hash = 0;
  da.gnz_out = d.gnz; da.gnz = d.vnz; da.vnz_out = d.vnz; da.cg_nz = d.g.nz; da.cg_inv_cs = d.g.inv_cs;   // (pass A's kernels behind the split see the lattice with the stream's own layers through stream_grid) da.scell = nullptr;   // (nobody reads the cell of a position of the voxel-ordered cloud)
  da.skey = d.rkeys[da.cell_passes & 1]; da.sidx = d.rvals[da.cell_passes & 1];
  if (sub == 0) {
    // (z range, ground flags and the queue of big voxels need no clearing launches: their last readers of the previous frame on this copy
    //  leave them ready — publish_split of pass B, the frame tag in is_ground, k_g2_mode)
    mor_launch_split_and_grid(da, st, tm);
  } else if (sub == 1) {
    MOR_LAUNCH_T(MK_G2_COV, k_g2_cov, dim3(G2_COV_G * d.B), MOR_BT, da);
  } else if (sub == 2) {
    MOR_LAUNCH(MK_G2_COV_MID, k_g2_cov_mid, dim3(64, d.B), da);
  } else if (sub == 3) {
    MOR_LAUNCH_T(MK_G2_COV_BIG, k_g2_cov_big, dim3(64, d.B), G2_BIG_T, da);
  } else if (sub == 4) {
    MOR_LAUNCH(MK_G2_MODE, k_g2_mode, gB, da);
    MOR_LAUNCH(MK_G2_MARK, k_g2_mark, dim3(128 * d.B), da);
  } else {
    MorDev db = d; db.gmode = 2;
    mor_launch_split_and_grid(db, st, tm);
  }
}

static void mor_launch_cellgraph(const MorDev &d, hipStream_t st, MorLaunchTimer *tm) {   // slabs (+ merge in each stream's last slab workgroup), or slabs | merge
  MOR_LAUNCH_T(MK_CG_SLAB, (k_cg_slab<CGS_CAP>), dim3(d.B * ((d.prop_map && d.slab_T > 0) ? d.P + 1 : d.P)), CGS_T, d);
  if (d.cg_fused) return;
  MOR_LAUNCH_T(MK_CG_FINAL, k_cg_final, dim3(d.B), CGF_T, d);
}

static void mor_launch_clusters(const MorDev &d, hipStream_t st, MorLaunchTimer *tm) {   // labels, cluster points, centroids, boxes; transform of ca; correspondences
  MOR_LAUNCH(MK_CLUSTERS, k_clusters, dim3(d.B * (d.g_out + MOR_CLS_G + (d.has_prev ? MOR_XF_G : 0))), d);
}
static void mor_launch_pairs(const MorDev &d, hipStream_t st, MorLaunchTimer *tm) {   // transform of ca, correspondences, first tiers of the scores
  const dim3 gT(d.B * d.tiles), gB(d.B), gKt((d.Kcap + MOR_BT - 1) / MOR_BT, d.B);
  if (d.has_prev) {
    if (d.method == 1) {
      if (d.pde_ub > 0.f && d.pde_ub > d.pde_lb) {
        MOR_LAUNCH_T(MK_SCORE_FAST, k_score_fast, dim3(d.B * d.g_fast), SCF_T, d);
        MOR_LAUNCH_T(MK_SCORE_NB, k_score_nb, dim3(d.g_score * d.B), SCN_T, d);
      }
    } else if (d.method == 2) {
      MOR_LAUNCH(MK_VOX_CLEAR, k_vox_clear, dim3(64 * d.B), d);
      MOR_LAUNCH(MK_VOX_INSERT, k_vox_insert, dim3(d.B * 2 * d.g_out), d);   // (cluster points of ca / cb: at most the cloud; 1024-point half tiles would do, so twice the cloud's width)
      MOR_LAUNCH(MK_VOX_PROBE, k_vox_probe, dim3(d.B * 2 * d.g_out), d);
    }
  }
}

static void mor_launch_scores2(const MorDev &d, hipStream_t st, MorLaunchTimer *tm) {
  if (d.has_prev && d.method == 1 && d.pde_ub > 0.f && d.pde_ub > d.pde_lb) {
    MOR_LAUNCH_T(MK_SCORE1, k_score_pde, dim3(d.g_pde * d.B), SCP_T, d);
  }
}
static void mor_launch_decide(const MorDev &d, hipStream_t st, MorLaunchTimer *tm) {
  MOR_LAUNCH_T(MK_TRACK_PUSH, k_track_push, dim3(d.B), 64, d);
}

// The launches of one push, in dependency order, as MOR_N_PIECES pieces; the engine assigns consecutive pieces to its
// stage streams.  Arrays written by one piece and read by a later one exist once per frame in flight; pieces 4's kernels
// (transform of ca … first score tiers) stay together because they mutate / read the previous frame's cluster points.
void mor_launch_piece(const MorDev &d, int piece, hipStream_t st, MorLaunchTimer *tm) {
  switch (piece) {
    case 1: mor_launch_boxes(d, st, tm); break;
    case 3: mor_launch_clusters(d, st, tm); break;
    case 4: mor_launch_pairs(d, st, tm); break;
    case 5: mor_launch_scores2(d, st, tm); break;
    case 6: mor_launch_decide(d, st, tm); break;
    case 2: mor_launch_cellgraph(d, st, tm); break;
    case 7: mor_launch_split_and_grid(d, st, tm, 1); break;   // the grid piece of the crop variant in two: split | grid build
    case 8: mor_launch_split_and_grid(d, st, tm, 2); break;
    case 10: case 11: case 12: case 13: case 14: case 15: mor_launch_grid_sub(d, piece - 10, st, tm); break;   // the grid piece of the voxel ground variant in six
    default: break;
  }
}

void mor_launch_filter(const MorDev &d, hipStream_t st, MorLaunchTimer *tm, int part) {   // part 1: the loop over mo_vec (:630-671; on EVERY filterCloud call, as in the reference — a second call on the same frame walks the tracks again; the next frame's tracking step waits for this only); part 2: the output
  if (part == 1) { MOR_LAUNCH_T(MK_TRACK_FILTER, k_track_filter, dim3(d.B), FLT_T, d); return; }
  MOR_LAUNCH_T(MK_OUT, k_out, dim3(d.B * (d.g_out + (d.out_ptrs ? d.tiles : 0))), FLT_T, d);
}

// A few KB from page-locked host memory into device memory, on the stream, by one workgroup (bytes: a multiple of 4)
__global__ __launch_bounds__(MOR_BT) void k_copy_small(uint4 *dst, const uint4 *src, int n16, unsigned *dst4, const unsigned *src4, int n4) {
  for (int i = threadIdx.x; i < n16; i += MOR_BT) dst[i] = src[i];
  for (int i = threadIdx.x; i < n4; i += MOR_BT) dst4[i] = src4[i];
}
void mor_launch_copy(void *dst, const void *src_pinned, size_t bytes, hipStream_t st) {
  const bool al = (((uintptr_t)dst | (uintptr_t)src_pinned) & 15u) == 0;
  const int n16 = al ? (int)(bytes / 16) : 0; const size_t done = (size_t)n16 * 16;
  hipLaunchKernelGGL(k_copy_small, dim3(1), dim3(MOR_BT), 0, st, (uint4 *)dst, (const uint4 *)src_pinned, n16, (unsigned *)((char *)dst + done), (const unsigned *)((const char *)src_pinned + done), (int)((bytes - done) / 4));
}
// workgroups of k_split one CU holds (registers decide): the host keeps sp_g × B within what the whole GPU holds at once
int mor_split_blocks_per_cu() {
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_split<false>, MOR_BT, 0) != hipSuccess) n = 2;
  return n < 1 ? 1 : n;
}
