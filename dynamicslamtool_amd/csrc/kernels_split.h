// kernels_split.h — part of mor_kernels.hip (one translation unit: #included there, in dependency order; not a stand-alone header).
// G1 (:62-88): trim + ground split — count / scatter passes and the single-read split with decoupled look-back.
// Reference citations are file:line of /root/reference/src/MovingObjectRemoval.cpp.
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
    zlo = wave_fmin(zlo); zhi = wave_fmax(zhi);
    if (lane_id() == 0 && zlo <= zhi) { atomicMin(&d.zmin_i[MOR_ZR * s], float_ordered(zlo)); atomicMax(&d.zmax_i[MOR_ZR * s], float_ordered(zhi)); }   // (a stream's word on a cache line of its own: 30 000 atomics per step on the FOUR lines of the dense arrays went through the L2 one after the other — 96 of the kernel's 129 µs)
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
  const int zl = d.zmin_i[MOR_ZR * s], zh = d.zmax_i[MOR_ZR * s];
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
// Voxel ground variant, pass A: what hangs on the z range of the trimmed cloud — the origin of both grids and their layer counts for this stream, the bet on the mode bin.
// The z range must be FINAL: k_scatter calls this behind k_classify's kernel boundary; the single-read pass A (k_split<1>), whose workgroups add to the range while
// others already store, leaves it to the first kernel behind it (k_rhist of radix pass 0).
__device__ __forceinline__ void publish_zgrid(const MorDev &d, int s) {
  const MorFrameInfo &f = d.info[s];
  const float zmin = f.T ? ordered_float(d.zmin_i[MOR_ZR * s]) : 0.f, zmax = f.T ? ordered_float(d.zmax_i[MOR_ZR * s]) : 0.f;
  d.zorg[s] = zmin; d.zbase[s] = (int)floorf(zmin * d.gv.inv_cs);   // grids of the voxel variant hang on the lowest trimmed point
  if (d.gnz_out) d.gnz_out[s] = max(1, min(d.cg_nz, (int)floorf((zmax - zmin) * d.cg_inv_cs) + 2));   // z layers of the clustering grid this stream needs (stream_grid)
  if (d.vnz_out) { d.vnz_out[s] = voxel_layers(d, s); d.g2_used[s] = d.g2_nobet ? 0x7ffffffe : ld_agent(&d.g2_pred[s]); }   // layers of the VoxelGrid lattice (the later kernels of pass A read it through stream_grid) … and the bet on the mode bin this frame's kernels mark by: ONE snapshot per frame (later frames' k_g2_mode update g2_pred while this frame's kernels run)
  if (f.T && (int)floorf(zmax * d.gv.inv_cs) - (int)floorf(zmin * d.gv.inv_cs) + 1 > d.gv.nz) mor_raise(d, s, 8u);   // z extent beyond the lattice: voxels of the top layer would be merged
}
// T, M, G of the frame (and, for pass A of the voxel variant, the z origin of its grids — unless the caller's z range is not final yet)
__device__ __forceinline__ void publish_split(const MorDev &d, int s, int n_ng, int n_g, bool z_final = true) {
  MorFrameInfo &f = d.info[s];
  f.M = n_ng; f.G = n_g; f.T = n_ng + n_g;
  if (d.gmode == 2) { d.zmin_i[MOR_ZR * s] = 0x7fffffff; d.zmax_i[MOR_ZR * s] = (int)0x80000000; }   // pass B ends the frame's use of the z range: ready for the next frame on this copy (no memset launches)
  if (d.gmode == 1 && z_final) publish_zgrid(d, s);
}
// The single-read pass A does not know the stream's lowest z cell when it stores a point's voxel: it leaves the three cell coordinates PACKED — x in the low bits, y above it,
// floor(z·inv) modulo what is left of the 32 bits (≥ 10 bits ≥ the lattice's layers: a stream whose cells span more raises the capacity flag, publish_zgrid) — and radix pass 0,
// which runs behind the kernel boundary, turns them into the stream's linear key on the fly (k_rhist, k_rscatter; the order of the packed and the linear keys is NOT the same,
// so every pass sorts linear keys).
struct VoxZ { int zb, nz, bx, by; };   // the stream's lowest z cell (absolute), its layers, the bit widths of x and y
__device__ __forceinline__ VoxZ voxel_z(const MorDev &d, int s) {
  VoxZ v; v.bx = 32 - __clz(d.g.nx - 1); v.by = 32 - __clz(d.g.ny - 1); v.zb = 0; v.nz = 1;
  const int zl = d.zmin_i[MOR_ZR * s], zh = d.zmax_i[MOR_ZR * s];
  if (zl <= zh) { v.zb = (int)floorf(ordered_float(zl) * d.g.inv_cs); v.nz = max(1, min(d.g.nz, (int)floorf(ordered_float(zh) * d.g.inv_cs) - v.zb + 1)); }
  return v;
}
__device__ __forceinline__ int voxel_pack(const MorGrid &g, float4 p, int bx, int by, bool &clamped) {
  int cx = (int)floorf(p.x * g.inv_cs) - g.ibx, cy = (int)floorf(p.y * g.inv_cs) - g.iby; const int cz = (int)floorf(p.z * g.inv_cs);
  clamped = cx < 0 || cy < 0 || cx >= g.nx || cy >= g.ny;
  cx = min(max(cx, 0), g.nx - 1); cy = min(max(cy, 0), g.ny - 1);
  return (int)((unsigned)cx | ((unsigned)cy << bx) | ((unsigned)cz << (bx + by)));   // (bx + by ≤ 22)
}
__device__ __forceinline__ int voxel_unpack(const MorDev &d, const VoxZ &v, int packed) {
  const unsigned k = (unsigned)packed;
  const int cx = (int)(k & ((1u << v.bx) - 1u)), cy = (int)((k >> v.bx) & ((1u << v.by) - 1u));
  const int cz = min((int)(((k >> (v.bx + v.by)) - (unsigned)v.zb) & ((1u << (32 - v.bx - v.by)) - 1u)), v.nz - 1);   // (clamped as grid_cell does; the flag is publish_zgrid's)
  return (cy * v.nz + cz) * d.g.nx + cx;
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
    if (d.gmode == 1) { zorg = (tot_ng + tot_g) ? ordered_float(d.zmin_i[MOR_ZR * s]) : 0.f; zbase = (int)floorf(zorg * d.gv.inv_cs); }
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
#define SP_NW MOR_SP_NW
#define SP_TILE (SP_NW * SP_ROWS * 64)   // records per tile of the single-read split
#define SP_NWS (SP_NW < 4 ? 4 : SP_NW)   // slots per count array in LDS
#define SP_DESC_STRIDE(d) ((size_t)(d).tiles_max * (8 / SP_ROWS) * 4 / SP_NW + 1)
__device__ __forceinline__ unsigned long long ld_agent64(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// loads only (no use of the data here: the wait for them belongs to split_tile, a step later); cls carries the ground flag of pass B
template <int PASS> __device__ __forceinline__ void split_load_tile(const MorDev &d, const MorStreamArgs &a, int s, uint32_t n_in, int t, float4 (&p)[SP_ROWS], int (&cls)[SP_ROWS]) {
  constexpr bool PASSB = PASS == 2;
  const int gtag = PASSB ? d.g2_tag[s] : 0;   // pass B: the frame's ground tag (k_g2_mode)
  const uint32_t base = (uint32_t)t * SP_TILE + wave_id() * (SP_ROWS * 64);
#pragma unroll
  for (int it = 0; it < SP_ROWS; ++it) {
    const uint32_t i = min(base + it * 64 + lane_id(), n_in - 1);   // (clamped: out-of-range lanes repeat the last record and are masked in split_tile)
    if (PASSB) { p[it] = d.rawbuf[(size_t)s * d.Nmax + i]; cls[it] = d.is_ground[(size_t)s * d.Nmax + i] == gtag; }
    else { p[it] = load_point(a, i); cls[it] = 0; }
  }
}
struct SplitMeta { int tng, tg, wng, wg; };   // a counted tile: its totals and this wave's offsets inside it
template <int PASS> __device__ __forceinline__ int split_class(const MorDev &d, uint32_t n_in, uint32_t i, const float4 &p, int cls) {
  return i < n_in ? (PASS == 2 ? (cls ? 1 : 2) : classify(d, p)) : 0;
}
// stage 1 of a tile (its loads were issued a step earlier): counts, and the tile's aggregate goes out to the other workgroups
template <int PASS> __device__ __forceinline__ void split_count(const MorDev &d, int s, int t, uint32_t n_in, unsigned epoch, const float4 (&p)[SP_ROWS], const int (&cls)[SP_ROWS], int *sh, SplitMeta &m) {
  int c_ng = 0, c_g = 0;
  float zlo = INFINITY, zhi = -INFINITY;   // (pass A)
  const uint32_t base = (uint32_t)t * SP_TILE + wave_id() * (SP_ROWS * 64) + lane_id();
#pragma unroll
  for (int it = 0; it < SP_ROWS; ++it) {
    const int c = split_class<PASS>(d, n_in, base + it * 64, p[it], cls[it]);
    const unsigned long long m_ng = __ballot(c == 2), m_g = __ballot(c == 1);
    c_ng += __popcll(m_ng); c_g += __popcll(m_g);
    if (PASS == 1 && c == 2) { zlo = fminf(zlo, p[it].z); zhi = fmaxf(zhi, p[it].z); }
    // gp_indices (:86) and the trimmed-cloud index of every cloud point are read-backs only: instead of 4 bytes per trimmed point the split leaves the
    // classes of every 64 records as two bit masks (16 bytes: cloud, ground; rows that hold records only — a tile reaches beyond the stream's slice of the
    // array); the host rebuilds the index lists from them when asked (mor_get_ground_indices, mor_get_labels)
    if (lane_id() == 0 && base + it * 64 < n_in) { unsigned long long *cm = d.cls_mask + ((size_t)s * d.cls_rows + (base + it * 64) / 64) * 2; cm[0] = m_ng; cm[1] = m_g; }
  }
  if (PASS == 1) {   // z extent of the trimmed cloud (the voxel variant does not crop in z): final at the kernel boundary, for radix pass 0 and everything behind it
    zlo = wave_fmin(zlo); zhi = wave_fmax(zhi);
    if (lane_id() == 0 && zlo <= zhi) { atomicMin(&d.zmin_i[MOR_ZR * s], float_ordered(zlo)); atomicMax(&d.zmax_i[MOR_ZR * s], float_ordered(zhi)); }
  }
  if (lane_id() == 0) { sh[wave_id()] = c_ng; sh[SP_NWS + wave_id()] = c_g; }
  __syncthreads();
  m.tng = 0; m.tg = 0; m.wng = 0; m.wg = 0;
#pragma unroll
  for (int w = 0; w < SP_NW; ++w) { m.tng += sh[w]; m.tg += sh[SP_NWS + w]; if (w < wave_id()) { m.wng += sh[w]; m.wg += sh[SP_NWS + w]; } }
  if (threadIdx.x == 0)
    __hip_atomic_store(d.split_desc + (size_t)s * SP_DESC_STRIDE(d) + t, ((unsigned long long)epoch << 32) | ((unsigned long long)(unsigned)m.tng << 16) | (unsigned long long)(unsigned)m.tg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// stage 2: look-back over the tiles between this workgroup's previous tile and this one, then the stores.
// tk_next (thread 0 only): the ticket this workgroup has just taken for a later tile — passed on to all threads through s_ex[2]
template <int PASS> __device__ __forceinline__ void split_store(const MorDev &d, const MorGrid &G, int s, int t, int nt, int t_prev, uint32_t n_in, unsigned epoch, const float4 (&p)[SP_ROWS], const int (&cls)[SP_ROWS],
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
          if ((unsigned)(v >> 32) == epoch) { an += (int)((v >> 16) & 0xffffu); ag += (int)(v & 0xffffu); break; }
          if (++spins > SPLIT_SPIN_LIMIT) { mor_raise(d, s, 64u); break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
    }
    an = wave_sum(an); ag = wave_sum(ag);
    if (lane == 0) {
      s_ex[0] = ex_ng + an; s_ex[1] = ex_g + ag; s_ex[2] = tk_next;
      if (t == nt - 1) publish_split(d, s, ex_ng + an + m.tng, ex_g + ag + m.tg, PASS != 1);   // the stream's last tile: T, M, G of the frame (pass A: the z range is not final yet — other workgroups may still be counting; k_rhist of radix pass 0 publishes what hangs on it)
    }
  }
  __syncthreads();
  int r_ng = s_ex[0], r_g = s_ex[1];
  ex_ng = r_ng + m.tng; ex_g = r_g + m.tg;   // prefix behind this tile: what the workgroup carries to its next one
  r_ng += m.wng; r_g += m.wg;
  const size_t so = (size_t)s * d.Nmax;
  const float zorg = PASS == 1 ? 0.f : d.zorg[s]; const int zbase = PASS == 1 ? 0 : d.zbase[s];
  const int vbx = 32 - __clz(d.g.nx - 1), vby = 32 - __clz(d.g.ny - 1);   // (pass A: bit widths of the packed voxel coordinates)
  const uint32_t base = (uint32_t)t * SP_TILE + wave_id() * (SP_ROWS * 64) + lane_id();
#pragma unroll
  for (int it = 0; it < SP_ROWS; ++it) {
    const int c = split_class<PASS>(d, n_in, base + it * 64, p[it], cls[it]);
    const unsigned long long m_ng = __ballot(c == 2), m_g = __ballot(c == 1);
    const int k_ng = r_ng + __popcll(m_ng & lanemask_lt()), k_g = r_g + __popcll(m_g & lanemask_lt());
    if (c == 2) {
      bool clamped; int key;
      if (PASS == 1) key = voxel_pack(d.g, p[it], vbx, vby, clamped);   // (the stream's lowest z cell is not known yet: radix pass 0 makes the linear key, voxel_unpack)
      else { int cx, cy, cz; grid_cell(G, p[it], zorg, zbase, cx, cy, cz, clamped); key = grid_key(G, cx, cy, cz); }
      if (clamped && d.gmode != 0) mor_raise(d, s, 8u);
      d.cloud[so + k_ng] = p[it];
      d.pkey[so + k_ng] = key;
    } else if (c == 1) {
      st_stream(&d.ground[2 * so + d.Nmax + k_g], p[it]);   // final place in filterCloud's output
    }
    r_ng += __popcll(m_ng); r_g += __popcll(m_g);
  }
}
template <int PASS> __global__ __launch_bounds__(64 * SP_NW, 4) void k_split(MorDev d) {   // (≤ 128 VGPRs at least — 73 with 1024-record tiles; with 2048-record tiles the compiler left to itself wandered between 126 and 150 registers with unrelated edits, and at 150 the split took 115 instead of 89 µs)
  int s, g; map_block(d.B, d.sp_g, s, g);
  const MorGrid G = PASS == 1 ? d.g : stream_grid(d, s);   // the stream's clustering grid (voxel ground variant: its own number of z layers; its pass A packs lattice coordinates instead)
  // As the first kernel of a frame (crop variant, pass A of the voxel ground variant) this one reads the stream's arguments straight from the page-locked slot the host filled,
  // and the owner of tile 0 leaves the device copy for the kernels behind it: no copy, no launch and no wait in front of the frame
  const MorStreamArgs a = d.args_src ? d.args_src[s] : d.args[s];
  const uint32_t n_in = pass_count(d, a, s);
  const int nt = (int)((n_in + SP_TILE - 1) / SP_TILE);
  const unsigned epoch = 2u * (unsigned)d.frame_no + (PASS == 2 ? 2u : 1u);   // never 0 (fresh descriptors), never the tag of an earlier pass over this table
  __shared__ int sh[4 * SP_NWS], s_ex[6];   // two copies of each, used in turn by the two halves of the loop: between two uses of a copy lies a workgroup barrier of the other half
  int *tk = d.tickets + (size_t)s * TK_COUNT + TK_SPLIT;
  const int tk_total = 2 * d.sp_g + nt;
  if (threadIdx.x == 0) {
    const int v = __hip_atomic_fetch_add(tk, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_ex[5] = v;
    if (v + 2 == tk_total) __hip_atomic_store(tk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v == 0) {   // the owner of tile 0 starts the frame: before any flag of this stream can be raised (every other tile waits for tile 0's descriptor)
      reset_frame_info(d, s, a.n);
      if (d.args_src) d.args_out[s] = a;
      if (nt == 0) publish_split(d, s, 0, 0, PASS != 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  int t = __builtin_amdgcn_readfirstlane(s_ex[5]), t1 = t + 1, t_prev = -1;
  int ex_ng = 0, ex_g = 0;
  float4 pa[SP_ROWS], pb[SP_ROWS]; int ca[SP_ROWS], cb[SP_ROWS];
  SplitMeta ma, mb;
  if (t < nt) split_load_tile<PASS>(d, a, s, n_in, t, pa, ca);
  if (t1 < nt) split_load_tile<PASS>(d, a, s, n_in, t1, pb, cb);
  if (t < nt) split_count<PASS>(d, s, t, n_in, epoch, pa, ca, sh, ma);
  while (t < nt) {   // pa: tile t, counted and published; pb: tile t1, loaded
    int nx = 0;
    if (t1 < nt) split_count<PASS>(d, s, t1, n_in, epoch, pb, cb, sh + 2 * SP_NWS, mb);   // the next tile's aggregate is out before this workgroup waits for anybody
    if (threadIdx.x == 0) nx = __hip_atomic_fetch_add(tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    split_store<PASS>(d, G, s, t, nt, t_prev, n_in, epoch, pa, ca, ma, ex_ng, ex_g, s_ex, nx);
    if (threadIdx.x == 0 && nx + 1 == tk_total) __hip_atomic_store(tk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t_prev = t;
    const int t2 = __builtin_amdgcn_readfirstlane(s_ex[2]);
    if (t2 < nt) split_load_tile<PASS>(d, a, s, n_in, t2, pa, ca);
    if (t1 >= nt) break;
    if (t2 < nt) split_count<PASS>(d, s, t2, n_in, epoch, pa, ca, sh, ma);
    if (threadIdx.x == 0) nx = __hip_atomic_fetch_add(tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    split_store<PASS>(d, G, s, t1, nt, t_prev, n_in, epoch, pb, cb, mb, ex_ng, ex_g, s_ex + 3, nx);
    if (threadIdx.x == 0 && nx + 1 == tk_total) __hip_atomic_store(tk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t_prev = t1;
    const int t3 = __builtin_amdgcn_readfirstlane(s_ex[5]);
    if (t3 < nt) split_load_tile<PASS>(d, a, s, n_in, t3, pb, cb);
    t = t2; t1 = t3;
  }
}

