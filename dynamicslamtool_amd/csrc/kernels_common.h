// kernels_common.h — part of mor_kernels.hip (one translation unit: #included there, in dependency order; not a stand-alone header).
// helpers shared by every stage: workgroup → (stream, tile) maps, scans, agent-scope accesses, hand-offs, blob field access, grid cells and rows.
// Reference citations are file:line of /root/reference/src/MovingObjectRemoval.cpp.
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

// ---- cross-lane primitives without the LDS crossbar (round 6).  `__shfl_xor` / `__shfl_up` compile to ds_bpermute_b32: an LDS instruction with its queue and an lgkmcnt
// wait per step — in kernels whose own counters call them LDS-bound.  gfx950 moves data between lanes in the VALU: DPP (quad_perm, row_shr, row_mirror, row_bcast, wave_shr)
// and v_permlane16_swap / v_permlane32_swap for the steps across 16-lane rows.
//   DPP controls: quad_perm 0x00–0xFF, row_shr:n 0x110+n, wave_shl:1 0x130, wave_shr:1 0x138, row_mirror 0x140, row_half_mirror 0x141, row_bcast15 0x142, row_bcast31 0x143
template <int CTRL, int ROWS = 0xF, bool ZERO_OOB = false> __device__ __forceinline__ int dpp_mov(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, ROWS, 0xF, ZERO_OOB); }
template <class T> struct WaveWords { static_assert(sizeof(T) % 4 == 0 && sizeof(T) <= 64, "whole dwords"); int w[sizeof(T) / 4]; };
template <class T> __device__ __forceinline__ WaveWords<T> to_words(const T &v) { WaveWords<T> r; __builtin_memcpy(&r, &v, sizeof(T)); return r; }
template <class T> __device__ __forceinline__ T from_words(const WaveWords<T> &w) { T r; __builtin_memcpy(&r, &w, sizeof(T)); return r; }
// op over all 64 lanes, result in every lane; op commutative and associative (min, max, integer / IEEE add of two operands, lexicographic best …).  Six steps: partners inside
// the quad (two quad_perms), the other quad of the 8 (row_half_mirror), the other 8 of the row (row_mirror), the neighbouring row (v_permlane16_swap), the other half
// (v_permlane32_swap) — after step k every lane of a 2^k group holds the group's result, so any partner of the other half-group will do.
// GROUP: the aligned group of lanes reduced over (2, 4, 8, 16, 32 or 64): the first log2(GROUP) steps.  For two-operand IEEE adds the partners of the mirror steps hold the same
// values as the xor partners of a butterfly would (groups are uniform by then), so the sums are those of the xor butterfly that starts with the nearest partner (offsets 1, 2, 4, …) bit for bit.
template <int GROUP, class T, class OP> __device__ __forceinline__ T wave_group_allreduce(T v, OP op) {
  static_assert(GROUP == 2 || GROUP == 4 || GROUP == 8 || GROUP == 16 || GROUP == 32 || GROUP == 64, "aligned power-of-two groups");
  constexpr int W = (int)(sizeof(T) / 4);
#define MOR_DPP_STEP(CTRL) { const WaveWords<T> a = to_words(v); WaveWords<T> b; _Pragma("unroll") for (int i = 0; i < W; ++i) b.w[i] = dpp_mov<CTRL>(a.w[i], a.w[i]); v = op(v, from_words<T>(b)); }
  MOR_DPP_STEP(0xB1)
  if (GROUP >= 4) MOR_DPP_STEP(0x4E)
  if (GROUP >= 8) MOR_DPP_STEP(0x141)
  if (GROUP >= 16) MOR_DPP_STEP(0x140)
#undef MOR_DPP_STEP
  if (GROUP >= 32) { const WaveWords<T> a = to_words(v); WaveWords<T> x, y;
#pragma unroll
    for (int i = 0; i < W; ++i) { const auto r = __builtin_amdgcn_permlane16_swap((unsigned)a.w[i], (unsigned)a.w[i], false, false); x.w[i] = (int)r[0]; y.w[i] = (int)r[1]; }
    v = op(from_words<T>(x), from_words<T>(y)); }
  if (GROUP >= 64) { const WaveWords<T> a = to_words(v); WaveWords<T> x, y;
#pragma unroll
    for (int i = 0; i < W; ++i) { const auto r = __builtin_amdgcn_permlane32_swap((unsigned)a.w[i], (unsigned)a.w[i], false, false); x.w[i] = (int)r[0]; y.w[i] = (int)r[1]; }
    v = op(from_words<T>(x), from_words<T>(y)); }
  return v;
}
template <class T, class OP> __device__ __forceinline__ T wave_allreduce(T v, OP op) { return wave_group_allreduce<64>(v, op); }
template <class T> __device__ __forceinline__ T wave_sum(T v) { return wave_allreduce(v, [](T a, T b) { return a + b; }); }
__device__ __forceinline__ float wave_fmin(float v) { return wave_allreduce(v, [](float a, float b) { return fminf(a, b); }); }   // min / max of a value over the wave (all lanes get the result)
__device__ __forceinline__ float wave_fmax(float v) { return wave_allreduce(v, [](float a, float b) { return fmaxf(a, b); }); }
__device__ __forceinline__ int wave_imin(int v) { return wave_allreduce(v, [](int a, int b) { return min(a, b); }); }
__device__ __forceinline__ int wave_imax(int v) { return wave_allreduce(v, [](int a, int b) { return max(a, b); }); }
// value of a lane every lane agrees on (a constant, or a lane picked from a ballot): v_readlane_b32 into an SGPR instead of a ds_bpermute
template <class T> __device__ __forceinline__ T wave_bcast(const T &v, int lane_uniform) {
  const WaveWords<T> a = to_words(v); WaveWords<T> b;
#pragma unroll
  for (int i = 0; i < (int)(sizeof(T) / 4); ++i) b.w[i] = __builtin_amdgcn_readlane(a.w[i], lane_uniform);
  return from_words<T>(b);
}
// lane l ← lane l − 1 (lane 0: `first`), lane l ← lane l + 1 (lane 63: `last`): one DPP move per dword
template <class T> __device__ __forceinline__ T wave_shift_up1(const T &v, const T &first) {
  const WaveWords<T> a = to_words(v), f = to_words(first); WaveWords<T> b;
#pragma unroll
  for (int i = 0; i < (int)(sizeof(T) / 4); ++i) b.w[i] = dpp_mov<0x138>(f.w[i], a.w[i]);   // wave_shr:1 (lane 0 keeps `old` = first)
  return from_words<T>(b);
}
template <class T> __device__ __forceinline__ T wave_shift_down1(const T &v, const T &last) {
  const WaveWords<T> a = to_words(v), f = to_words(last); WaveWords<T> b;
#pragma unroll
  for (int i = 0; i < (int)(sizeof(T) / 4); ++i) b.w[i] = dpp_mov<0x130>(f.w[i], a.w[i]);   // wave_shl:1 (lane 63 keeps `old` = last)
  return from_words<T>(b);
}
// inclusive prefix sum over the wave: four row_shr steps inside the 16-lane rows, then lane 15 of rows 0 / 2 into rows 1 / 3 (row_bcast15) and lane 31 into rows 2 and 3 (row_bcast31)
__device__ __forceinline__ int wave_incl_scan(int v) {
  v += dpp_mov<0x111, 0xF, true>(0, v); v += dpp_mov<0x112, 0xF, true>(0, v); v += dpp_mov<0x114, 0xF, true>(0, v); v += dpp_mov<0x118, 0xF, true>(0, v);
  v += dpp_mov<0x142, 0xA>(0, v); v += dpp_mov<0x143, 0xC>(0, v);
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
    const long long w = i0 + lane < ng ? (long long)wf(x + stp * (i0 + lane) + d.s0) : 0ll;
    W += wave_sum(w);
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
      g = wave_bcast(gi, l);
      t = __builtin_amdgcn_readfirstlane(r - (carry + wave_bcast(incl, l) - g));
      s = __builtin_amdgcn_readfirstlane(x + stp * (i0 + l) + d.s0);
      return true;
    }
    carry += wave_bcast(incl, 63);
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
  p = wave_sum(p); a = wave_sum(a);
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
__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }   // keeps the compiler from moving LDS accesses across it; lanes of one wave then see each other's LDS writes
enum { TK_TRACK = 0, TK_CGFINAL = 1, TK_PAIRS = 2, TK_SPLIT = 3, TK_OUT = 4, TK_SLABCNT = 5, TK_MOVERS = 6, TK_RADIX = 7, TK_COUNT = 8 };   // ticket words per stream

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
  if (d.gmode == 2) { p = d.rawbuf[(size_t)s * d.Nmax + i]; return d.is_ground[(size_t)s * d.Nmax + i] == d.g2_tag[s] ? 1 : 2; }   // (ground flags carry the frame's tag — the speculative one if the bet on the mode bin held, k_g2_mode —: no clearing pass)
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
// The same from the occupancy bits of the VoxelGrid lattice (voxel ground variant, round 5): bits[row][x], a row padded to chunks of 512 cells = eight words = one 64-byte line,
// and beside every word the compact id of the first cell at or behind its bit 0 (dir, written by k_g2_cent from the row table and the bits): the directory entry, the word of x0
// and its successor — THREE independent loads and two popcounts where the key search is a chain of four round trips and three dozen loads per row.  k_g2_cent writes both, whole rows
// at a time (first form: an atomic OR per voxel in k_heads_scatter — 35 µs of that kernel — and a clearing pass in k_g2_mark).  (First form: row table + the row's whole line, the cells in front of x0 counted from its eight words — a hundred ALU instructions per row.)
__device__ __forceinline__ void row_cells_bits(const unsigned long long *bits, const int *dir, int nw /* words per row */, int r, int x0, int x1, int &lo, int &hi) {
  const int w0 = x0 >> 6, sh = x0 & 63;
  const size_t o = (size_t)r * nw + w0;
  const unsigned long long a = bits[o], b1 = bits[o + (w0 + 1 < nw ? 1 : 0)];
  const int base = dir[o];
  const unsigned long long b = w0 + 1 < nw ? b1 : 0ull;
  const unsigned long long win = sh ? (a >> sh) | (b << (64 - sh)) : a;   // cells x0, x0 + 1, … in bits 0, 1, …
  lo = base + __popcll(a & ((1ull << sh) - 1ull));
  hi = lo + __popcll(win & ((2ull << (x1 - x0)) - 1ull));   // (x1 − x0 ≤ 62)
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
