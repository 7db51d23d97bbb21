// kernels_radix.h — part of mor_kernels.hip (one translation unit: #included there, in dependency order; not a stand-alone header).
// stable LSD radix sort, 8-bit digits, batched over streams (VoxelGrid pass of the voxel ground variant).
// Reference citations are file:line of /root/reference/src/MovingObjectRemoval.cpp.
// ------------------------------------------------------------------------------------ stable LSD radix sort, 8-bit digits, batched over streams
// used by the sort path of the grid (MOR_GRID=radix) and by the VoxelGrid pass of the voxel ground variant.
__device__ __forceinline__ void radix_item(const MorDev &d, const MorRadix &j, const VoxZ &vz, size_t so, int count, int i, int &key, int &val, bool &valid) {
  valid = i < count; key = 0; val = 0;
  if (!valid) return;
  key = j.kin[so + i]; val = j.vin ? j.vin[so + i] : i;
  if (j.unpack) key = voxel_unpack(d, vz, key);   // (pass 0 behind the single-read pass A: packed lattice coordinates → the stream's linear voxel key)
  if (j.drop_negative) valid = key >= 0;
}
__device__ __forceinline__ int radix_count(const MorDev &d, const MorRadix &j, int s) {
  if (j.skip_k_le > 0 && (int)d.info[s].K <= j.skip_k_le) return 0;   // all higher digits are zero: the previous pass already produced the final order
  if (j.vox && (j.shift >> 3) >= voxel_passes_of(d, s)) return 0;   // voxel keys of this stream end below this digit: its order is final (the consumers pick the buffer by the stream's pass count)
  return j.count_sel == 0 ? d.info[s].M : d.info[s].C;
}

// With j.fuse the stream's LAST workgroup to arrive (stream_last_block) turns the stream's raw per-tile histograms into offsets in place — offsets[tile][digit] = Σ smaller digits +
// Σ earlier tiles, one thread per digit — so that k_rscatter reads ONE word per thread and tile.  (Rounds 2 – 5 had every workgroup of k_rscatter re-derive its offsets from the raw
// histograms of all the stream's tiles: 59 loads per thread and tile, 226 MB of L2 reads per pass for 112 MB of keys and values — the scatter's price in the pipeline, 50 µs per pass.)
#ifndef RH_COL
#define RH_COL 32   // tiles per round trip of the offsets scan
#endif
__global__ __launch_bounds__(MOR_BT) void k_rhist(MorDev d, MorRadix j) {
  int s, t0; map_block(d.B, d.tiles_m, s, t0);
  if (j.unpack && t0 == 0 && threadIdx.x == 0) publish_zgrid(d, s);   // (the single-read pass A left the z range to its kernel boundary: origin and layers of the stream's grids for the kernels BEHIND this one; this launch works them out itself, voxel_z)
  const int count = j.unpack ? (int)d.info[s].M : radix_count(d, j, s);   // (pass 0: every stream takes part; radix_count's test reads the layer count this very kernel publishes)
  const VoxZ vz = j.unpack ? voxel_z(d, s) : VoxZ{0, 1, 0, 0};
  __shared__ int h[256], sh[8], l_last;
  const size_t so = (size_t)s * d.Nmax;
  for (int t = t0; t * MOR_TILE < count; t += d.tiles_m) {
    const int base = t * MOR_TILE;
    h[threadIdx.x] = 0;
    __syncthreads();
    for (int i = base + threadIdx.x; i < min(base + MOR_TILE, count); i += MOR_BT) {
      int key, val; bool valid; radix_item(d, j, vz, so, count, i, key, val, valid);
      if (valid) atomicAdd(&h[(key >> j.shift) & 255], 1);
    }
    __syncthreads();
    st_agent(&j.hist[((size_t)s * d.tiles_max + t) * 256 + threadIdx.x], h[threadIdx.x]);   // (agent scope: the scan below may run in a workgroup on another XCD)
    __syncthreads();
  }
  if (!j.fuse || count == 0) return;
  if (!stream_last_block(d.tickets + (size_t)s * TK_COUNT + TK_RADIX, d.tiles_m, &l_last)) return;
  int *hh = j.hist + (size_t)s * d.tiles_max * 256 + threadIdx.x;
  const int nt = (count + MOR_TILE - 1) / MOR_TILE;
  if (nt <= 4 * RH_COL) {   // the digit's column RH_COL tiles per round trip: totals first (the base needs the sum over ALL tiles), then the running sums — loads of a batch are independent, so a
    // stream of 59 tiles takes four round trips where the loop below takes eight and a second pass; RH_COL registers per thread (64 of them cost the histogram loop above its waves: − 1 %)
    int total = 0;
    for (int t0 = 0; t0 < nt; t0 += RH_COL) {
      int v[RH_COL];
#pragma unroll
      for (int u = 0; u < RH_COL; ++u) v[u] = t0 + u < nt ? ld_agent(&hh[(t0 + u) * 256]) : 0;
#pragma unroll
      for (int u = 0; u < RH_COL; ++u) total += v[u];
    }
    int tot; int run = block_excl_scan(total, sh, &tot);
    for (int t0 = 0; t0 < nt; t0 += RH_COL) {
      int v[RH_COL];
#pragma unroll
      for (int u = 0; u < RH_COL; ++u) v[u] = t0 + u < nt ? ld_agent(&hh[(t0 + u) * 256]) : 0;
#pragma unroll
      for (int u = 0; u < RH_COL; ++u) { const int x = v[u]; v[u] = run; run += x; }
#pragma unroll
      for (int u = 0; u < RH_COL; ++u) if (t0 + u < nt) hh[(t0 + u) * 256] = v[u];
    }
    return;
  }
  int run = 0, t = 0;
  for (; t + 8 <= nt; t += 8) {   // eight independent loads per step
    int v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = ld_agent(&hh[(t + u) * 256]);
#pragma unroll
    for (int u = 0; u < 8; ++u) { hh[(t + u) * 256] = run; run += v[u]; }
  }
  for (; t < nt; ++t) { const int v = ld_agent(&hh[t * 256]); hh[t * 256] = run; run += v; }
  int tot; const int base = block_excl_scan(run, sh, &tot);
  for (t = 0; t < nt; ++t) hh[t * 256] += base;
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
  const VoxZ vz = j.unpack ? voxel_z(d, s) : VoxZ{0, 1, 0, 0};
  const size_t so = (size_t)s * d.Nmax;
  const bool inverse = j.inverse && (!j.vox || (j.shift >> 3) == voxel_passes_of(d, s) - 1);   // the stream's LAST pass leaves the inverse permutation
  __shared__ int wcnt[4][256];
  for (int t = t0; t * MOR_TILE < count; t += d.tiles_m) {
    const int tb = t * MOR_TILE;
    for (int k = threadIdx.x; k < 4 * 256; k += MOR_BT) (&wcnt[0][0])[k] = 0;
    __syncthreads();
    int key[8], val[8], pre[8]; bool valid[8];
    int base = tb + wave_id() * 512;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      int i = base + it * 64 + lane_id();
      radix_item(d, j, vz, so, count, i, key[it], val[it], valid[it]);
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
      run = j.hist[((size_t)s * d.tiles_max + t) * 256 + dg];   // (k_rscan, or the tail of k_rhist)
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

