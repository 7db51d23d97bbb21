// kernels_ground_voxel.h — part of mor_kernels.hip (one translation unit: #included there, in dependency order; not a stand-alone header).
// G2 (:90-200): voxel-covariance ground removal — verdict per voxel (sixteen lanes / a wave / a workgroup), mode bin, marks.
// Reference citations are file:line of /root/reference/src/MovingObjectRemoval.cpp.
// ------------------------------------------------------------------------------------ G2: voxel-covariance ground removal (:90-200)
// Dead code in the reference (the call is commented out at :527 and would crash at :188); implemented with the
// intended semantics and the deterministic definitions of DESIGN.md §G2.  Pass A has trimmed the cloud in x/y
// and sorted it by VoxelGrid cell (stable ⇒ ascending point index inside a voxel).
#define G2_CAP 16384    // neighbours of one voxel centroid held in LDS as (d², index) keys (128 KiB of the CU's 160): big-voxel kernel
#define G2_SMALL 512    // … in the one-wave-per-voxel kernel (4 KiB: many workgroups per CU)
#define G2_CHUNK 1024   // coordinates staged per step of the ordered fp32 sums
// the nine (y,z) rows of the 3×3×3 voxel block around q, row `row` of them: the points of its three x-cells are positions [b0, b0 + len) of `sorted` (occupancy bits + directory
// of the lattice, or — lattices too large for them, MOR_G2_NOBITS — a search of the sorted keys)
__device__ __forceinline__ void g2_row_range(const MorDev &d, int s, const MorGrid &G, int cx, int cy, int cz, int row, int &b0, int &len) {
  b0 = 0; len = 0;
  const int y = cy + row % 3 - 1, z = cz + row / 3 - 1;
  if ((unsigned)y >= (unsigned)G.ny || (unsigned)z >= (unsigned)G.nz) return;
  const size_t so = (size_t)s * d.Nmax;
  const int *st = d.cstart + (size_t)s * (d.Nmax + 1);
  int lo, hi;
  if (d.g2_bits) { const size_t bo = (size_t)s * d.g.nrows * (size_t)(d.g2_nch * 8); row_cells_bits(d.g2_bits + bo, d.g2_dir + bo, d.g2_nch * 8, grid_row(G, y, z), max(cx - 1, 0), min(cx + 1, G.nx - 1), lo, hi); }
  else row_cells(G, d.ckey + so, d.row_start + (size_t)s * (d.g.nrows + 1), max(cx - 1, 0), min(cx + 1, G.nx - 1), y, z, lo, hi);
  if (lo < hi) { b0 = st[lo]; len = st[hi] - b0; }
}
// all trimmed points with d² < leaf² around q (radiusSearch, :125), appended to the LDS list in arbitrary order; the count keeps running beyond `cap` so the caller sees the
// overflow.  Threads 0 … 8 resolve the nine rows side by side (one after the other they were a chain of forty round trips in front of every voxel of the big tier), then the
// whole workgroup walks them.  l_rng: 18 ints of LDS.
__device__ __forceinline__ void g2_gather(const MorDev &d, int s, float4 q, unsigned long long *key, int *cnt, int cap, int *l_rng) {
  const size_t so = (size_t)s * d.Nmax;
  const MorGrid G = stream_grid(d, s);   // the lattice with the stream's own z layers
  int cx, cy, cz; bool cl; grid_cell(G, q, 0.f, d.zbase[s], cx, cy, cz, cl);
  if (threadIdx.x < 9) { int b0, len; g2_row_range(d, s, G, cx, cy, cz, threadIdx.x, b0, len); l_rng[2 * threadIdx.x] = b0; l_rng[2 * threadIdx.x + 1] = len; }
  __syncthreads();
  for (int r = 0; r < 9; ++r) {
    const int b0 = l_rng[2 * r], len = l_rng[2 * r + 1];
    for (int k = b0 + threadIdx.x, e = b0 + len; k < e; k += blockDim.x) {
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
  const int lane = threadIdx.x & 63;
  const MorGrid G = stream_grid(d, s);
  int cx, cy, cz; bool cl; grid_cell(G, q, 0.f, d.zbase[s], cx, cy, cz, cl);
  int b0 = 0, len = 0;
  if (lane < 9) g2_row_range(d, s, G, cx, cy, cz, lane, b0, len);
  int rb[9], rp[10]; rp[0] = 0;
#pragma unroll
  for (int r = 0; r < 9; ++r) { rb[r] = wave_bcast(b0, r); rp[r + 1] = rp[r] + wave_bcast(len, r); }
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
// (Round 5, second step: the sums themselves are fp32 — differences to q, their products, a lane's hits one after the other, then the tree over the lanes: at most n + 8 roundings
//  on any path, each relative to |a|, |a·c| … — and the screen charges them to its bound: every sum is off by at most g'·Σ|terms|, g' = 1.01·γ_{n+10}, which moves a term t by at
//  most g'·(Σ|a·c| + |mc|·Σ|a| + |ma|·Σ|c|) ≤ g'·axz — the last summand of the bound once more.  The coordinates here are decimetres around q where the reference's are tens of
//  metres around the origin: its roundings, not these, are what the bound consists of.  Measured: nothing by itself — see G2_GW.)
struct G2Acc { float Sa, Sb, Sc, Sac, Sbc, Scc, Sdd; int n; };
__device__ __forceinline__ void g2_acc_zero(G2Acc &A) { A.Sa = A.Sb = A.Sc = A.Sac = A.Sbc = A.Scc = A.Sdd = 0.f; A.n = 0; }
__device__ __forceinline__ void g2_acc_add(G2Acc &A, const float4 &q, const float4 &p, float dd /* sqdist(q, p) */) {
  const float a = p.x - q.x, b = p.y - q.y, c = p.z - q.z;
  A.Sa += a; A.Sb += b; A.Sc += c; A.Sac += a * c; A.Sbc += b * c; A.Scc += c * c; A.Sdd += dd;
  ++A.n;
}
template <int W> __device__ __forceinline__ void g2_acc_reduce(G2Acc &A) {   // over the W lanes of the caller's group (W = 16 or 64, aligned)
  // (DPP / permlane partners — kernels_common.h.  The additions pair up nearest lanes first, where the shuffle form of round 5 started with the farthest: other roundings of the same
  //  fp32 sums, which the screen's bound covers for ANY order — g2_screen charges n + 10 roundings)
  auto add = [](auto a, auto b) { return a + b; };
  A.Sa = wave_group_allreduce<W>(A.Sa, add); A.Sb = wave_group_allreduce<W>(A.Sb, add); A.Sc = wave_group_allreduce<W>(A.Sc, add);
  A.Sac = wave_group_allreduce<W>(A.Sac, add); A.Sbc = wave_group_allreduce<W>(A.Sbc, add); A.Scc = wave_group_allreduce<W>(A.Scc, add);
  A.Sdd = wave_group_allreduce<W>(A.Sdd, add); A.n = wave_group_allreduce<W>(A.n, add);
}
// 1: accepted (:145 holds whatever the order), 0: rejected, −1: too close to call — the ordered sums decide.  n > 3.
__device__ __forceinline__ int g2_screen(const G2Acc &A, const float4 &q, double leaf_r /* √leaf² · 1.0001 + 1e-6, from the host */, double inv_r /* 1 / √leaf² */) {
  // (one division and no square root: sixteen lanes wait while one works this out for its group — r comes from the host, 1 / (1 − x) ≤ 1 + 2x for x ≤ ½)
  const double n = (double)A.n, u = 5.9604644775390625e-8, r = leaf_r;
  if ((n + 10.0) * u > 0.25) return -1;   // (millions of neighbours: the bounds say nothing any more)
  const double xo = (n + 10.0) * u, go = 1.01 * xo * (1.0 + 2.0 * xo);   // ≥ 1.01·γ_{n+10}: what the fp32 sums of THIS kernel can be off by, relative to the sums of the absolute values of their terms
  const double Sa = A.Sa, Sb = A.Sb, Sc = A.Sc, Sac = A.Sac, Sbc = A.Sbc, Scc = A.Scc;
  const double inv_n = 1.0 / n, ma = Sa * inv_n, mb = Sb * inv_n, mc = Sc * inv_n;   // centroid − q
  const double txz = Sac - Sa * mc, tyz = Sbc - Sb * mc, tzz = Scc - Sc * mc;
  const double fa = fabs(ma), fb = fabs(mb), fc = fabs(mc);
  const double sdd = (double)A.Sdd * (1.000001 + 2.0 * go) + 1e-12;   // ≥ Σ d_i² (the fp32 distances carry three roundings each, their fp32 sum its own)
  const double ab1 = 0.5 * (n * r + sdd * inv_r), ab2 = 0.5 * sdd;   // ≥ Σ|a_i|, Σ|b_i|, Σ|c_i|;  ≥ Σ|a_i·c_i|, Σ|b_i·c_i|
  const double sa = ab1 + n * fa, sb = ab1 + n * fb, sc = ab1 + n * fc;   // ≥ Σ|x_i − c| …
  const double axz = ab2 + fc * ab1 + fa * ab1 + n * fa * fc, ayz = ab2 + fc * ab1 + fb * ab1 + n * fb * fc, azz = Scc * (1.0 + 2.0 * go) + 2.0 * fc * ab1 + n * fc * fc;   // ≥ Σ|(x_i − c)(z_i − c)| …
  const double xg = (n + 4.0) * u, g = 1.01 * xg * (1.0 + 2.0 * xg);   // ≥ 1.01·γ_{n+4}
  const double Dx = g * (fabs((double)q.x) + r), Dy = g * (fabs((double)q.y) + r), Dz = g * (fabs((double)q.z) + r);
  const double Exz = (1.0 + g) * (Dz * sa + Dx * sc + n * Dx * Dz) + (g + go) * axz;   // (the reference's roundings + this kernel's)
  const double Eyz = (1.0 + g) * (Dz * sb + Dy * sc + n * Dy * Dz) + (g + go) * ayz;
  const double Ezz = (1.0 + g) * (2.0 * Dz * sc + n * Dz * Dz) + (g + go) * azz;
  const double T = 0.001, tiny = 1e-9;
  const double lxz = fabs(txz) - 2.0 * Exz - tiny, lyz = fabs(tyz) - 2.0 * Eyz - tiny, lzz = fabs(tzz) - 2.0 * Ezz - tiny;   // lower bounds of |t_ref|
  if (lxz > T || lyz > T || lzz > T) return 0;
  const double hxz = fabs(txz) + 2.0 * Exz + tiny, hyz = fabs(tyz) + 2.0 * Eyz + tiny, hzz = fabs(tzz) + 2.0 * Ezz + tiny;   // upper bounds
  if (hxz < T && hyz < T && hzz < T) return 1;
  return -1;
}
// k_g2_cov: a few lanes per voxel (G2_GW, below), 64 voxels per 256-thread workgroup (a voxel centroid has a dozen neighbours on average, 96 % have ≤ 64): the group reads
// the voxel's centroid (k_g2_cent), resolves the nine rows of the 3×3×3 voxel block, walks their points and adds the neighbours within the radius into the screen's sums; the
// verdict is taken from those (above).  No LDS, no sort.  Queued for k_g2_cov_mid (a whole wave each): voxels with more than G2_NARROW_CAND candidates — dense surfaces next
// to the sensor, walked a few at a time they held their wave's other groups up — and the voxels the screen could not settle (tagged: their ordered sums are due).
#define G2_NARROW_CAND 512
#define G2_Q_EXACT (1 << 30)   // queue entry: the screen has been through this voxel and left it to the ordered sums
#define G2_V_NONE 0x7fffffff   // bin word of a voxel without a bin (rejected, or ≤ 3 neighbours)
#define G2_COV_G 128   // workgroups per stream of k_g2_cov; 64 of the middle / big tiers, 128 of k_g2_mark
#define G2_CENT_G 32   // workgroups per stream of k_g2_cent (a thread per voxel)
// Voxel centroids (dsc, :110-113): sequential fp32 sums over a voxel's points in ascending index (stable sort ⇒ storage order), one THREAD per voxel, sixteen loads per round
// trip.  Its own launch since round 5: k_g2_cov — whose time, cut into pieces, was 58 % the chains of dependent loads in front of its walks (voxel range → points → centroid →
// its cell → row table → key search → ranges → candidates, at five waves per SIMD) — now starts its row lookups from the voxel's own key at once, beside ONE load of the centroid.
// (196 µs alone, 38 with every voxel cut to 64 points: a few streams hold a voxel of 1 400 – 3 000 points — something right at the sensor —, one stream 215 voxels of 47 000
//  points.  Tried: the wave of the voxel's thread taking its big voxels one after the other, 64 points per coalesced load — they share a wave, 560 µs; a queue of the voxels over
//  64 points and a second launch with a wave each, sums by broadcast reads from LDS — 37 + 61 µs alone and NOTHING in the pipeline (35.10 against 35.06 k frame-pairs/s,
//  interleaved): a tail on a few CUs is not what the other lanes' kernels wait for; a wave per 64 voxels, their points — one range of `sorted` — staged 512 at a time through
//  LDS, every lane adding up its own voxel's part of the chunk: 214 µs alone and nothing either, 43.55 against 43.55 k.)
__global__ __launch_bounds__(MOR_BT) void k_g2_cent(MorDev d) {
  int s, bx; map_block(d.B, G2_CENT_G, s, bx);
  const int V = d.info[s].n_occ;
  const size_t so = (size_t)s * d.Nmax;
  const int *st = d.cstart + (size_t)s * (d.Nmax + 1);
  const float4 *sp = d.sorted + so;
  if (d.g2_bits) {   // the lattice's occupancy bits and their directory (row_cells_bits): a thread per (y,z) row of the stream's own layers — the row's cells from the row table, their
    // x from the sorted keys, eight at a time → the row's words, written whole (empty rows and words too: nothing to clear, no atomics) → first cell of every word
    const MorGrid G = stream_grid(d, s);
    const int nrows = V > 0 ? G.nrows : 0, nch = d.g2_nch;
    const int *rs = d.row_start + (size_t)s * (d.g.nrows + 1), *ckey = d.ckey + so;
    const size_t bo = (size_t)s * d.g.nrows * (size_t)(nch * 8);
    for (int r = bx * MOR_BT + threadIdx.x; r < nrows; r += G2_CENT_G * MOR_BT) {
      const int c0 = rs[r], c1 = rs[r + 1], kb = r * G.nx;
      int c = c0;
      for (int k = 0; k < nch; ++k) {   // (one chunk up to 512 cells in x)
        unsigned long long w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = 0ull;
        for (int cc = c0; cc < c1; cc += 8) {   // (every chunk looks at all the row's cells: a second chunk is rare)
          int kv[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) kv[i] = ckey[min(cc + i, c1 - 1)];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int x = kv[i] - kb - 512 * k;
            const unsigned long long bit = (cc + i < c1 && (unsigned)x < 512u) ? 1ull << (x & 63) : 0ull;
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] |= (x >> 6) == j ? bit : 0ull;
          }
        }
        ulonglong2 *wo = reinterpret_cast<ulonglong2 *>(d.g2_bits + bo + ((size_t)r * nch + k) * 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) wo[i] = make_ulonglong2(w[2 * i], w[2 * i + 1]);
        int o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { o[i] = c; c += __popcll(w[i]); }
        int4 *dd = reinterpret_cast<int4 *>(d.g2_dir + bo + ((size_t)r * nch + k) * 8);
        dd[0] = make_int4(o[0], o[1], o[2], o[3]); dd[1] = make_int4(o[4], o[5], o[6], o[7]);
      }
    }
  }
  for (int v = bx * MOR_BT + threadIdx.x; v < V; v += G2_CENT_G * MOR_BT) {
    float sx = 0.f, sy = 0.f, sz = 0.f; const int b0 = st[v], n = st[v + 1] - b0;
    for (int k = 0; k < n; k += 16) {
      float4 p[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) p[u] = sp[b0 + min(k + u, n - 1)];
#pragma unroll
      for (int u = 0; u < 16; ++u) if (k + u < n) { sx += p[u].x; sy += p[u].y; sz += p[u].z; }
    }
    const float fn = (float)n;
    d.vcent[so + v] = make_float4(sx / fn, sy / fn, sz / fn, 0.f);
  }
}
// G2_GW lanes per voxel.  Sixteen until round 5; the kernel's time is rounds of waves × the chain of dependent loads of an iteration (centroid + key → directory + bits → point
// ranges → candidates → marks), not arithmetic — the same sums in fp32 instead of fp64 changed nothing before the compiler's register count moved — so what counts is how many
// voxels a resident wave carries: 16 / 8 / 4 / 2 / 1 lanes per voxel → 38.2 / 40.9 / 42.1 / 42.5 / 41.6 k frame-pairs/s (interleaved; alone 491 / 340 / 303 / 329 µs).  A lane
// takes rows sub, sub + G2_GW, … of the nine and four candidates per round trip.  (Five waves per SIMD asked of the compiler: at 105 registers — four waves — the pipeline lost 2 %.)
#define G2_GW 4
__global__ __launch_bounds__(MOR_BT, 5) void k_g2_cov(MorDev d) {
  int s, bxv; map_block(d.B, G2_COV_G, s, bxv);   // (a stream's workgroups on one XCD, as everywhere else: as a two-dimensional launch a stream's voxels went round all eight L2s)
  const int V = d.info[s].n_occ;
  const size_t so = (size_t)s * d.Nmax;
  const int grp = threadIdx.x / G2_GW, sub = threadIdx.x % G2_GW, lane = lane_id(), lane0 = lane & ~(G2_GW - 1);   // group in the workgroup, lane in the group
  const int *ckey = d.ckey + so;
  const float4 *sp = d.sorted + so;
  const int zbase = d.zbase[s]; const MorGrid G = stream_grid(d, s);   // (the lattice with the stream's own z layers)
  const int pred = d.g2_used[s], tag_spec = 2 * d.frame_no + 1;   // the mode bin this frame's kernels bet on (pass A's snapshot of the latest known one), and the tag of the marks made on that bet
  for (int v0 = bxv * (MOR_BT / G2_GW); v0 < V; v0 += G2_COV_G * (MOR_BT / G2_GW)) {
    const int v = v0 + grp; const bool act = v < V;
    // ---- the voxel's centroid (k_g2_cent) and its key, one load each; the nine (y,z) rows of the 3×3×3 block start from the KEY's cell at once — lanes 0 … 8 of the group,
    //      each row's three x-cells are one range of `sorted` — and are looked up again from the centroid's cell in the rare case that the fp32 centroid rounds into a neighbour
    const float4 q = d.vcent[so + min(v, V - 1)];
    const int key = ckey[min(v, V - 1)], krow = key / G.nx;
    int cx = key - krow * G.nx, cy = krow / G.nz, cz = krow - cy * G.nz;
    {
      int qx, qy, qz; bool cl; grid_cell(G, q, 0.f, zbase, qx, qy, qz, cl);
      if (qx != cx || qy != cy || qz != cz) { cx = qx; cy = qy; cz = qz; }   // (the block is the one around the CENTROID's cell, :125)
    }
    constexpr int RPL = (9 + G2_GW - 1) / G2_GW;   // rows per lane (lane `sub` takes rows sub, sub + G2_GW)
    int rb0[RPL], rlen[RPL];
#pragma unroll
    for (int j = 0; j < RPL; ++j) {
      rb0[j] = 0; rlen[j] = 0;
      const int row = sub + j * G2_GW;
      if (act && row < 9) {
        g2_row_range(d, s, G, cx, cy, cz, row, rb0[j], rlen[j]);
      }
    }
    int rb[9], rp[10]; rp[0] = 0;
#pragma unroll
    for (int r = 0; r < 9; ++r) { rb[r] = __shfl(rb0[r / G2_GW], lane0 + r % G2_GW, 64); rp[r + 1] = rp[r] + __shfl(rlen[r / G2_GW], lane0 + r % G2_GW, 64); }
    // ---- walk: candidates sixteen at a time, four per lane and round trip; the hits go into the screen's sums
    const bool wide = rp[9] > G2_NARROW_CAND;   // (uniform in the group)
    const int ncand = wide ? 0 : rp[9];
    int wave_max = ncand;
    wave_max = wave_imax(wave_max);
    G2Acc A; g2_acc_zero(A);
    for (int c0 = 0; c0 < wave_max; c0 += 4 * G2_GW) {
      float4 pc[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + G2_GW * u + sub; int k = 0;
#pragma unroll
        for (int r = 0; r < 9; ++r) if (c >= rp[r] && c < rp[r + 1]) k = rb[r] + (c - rp[r]);
        pc[u] = sp[c < ncand ? k : 0];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + G2_GW * u + sub; const float4 p = pc[u];
        const float dd = sqdist(q.x, q.y, q.z, p.x, p.y, p.z);
        if (c < ncand && dd < d.leaf_r2) g2_acc_add(A, q, p, dd);
      }
    }
    g2_acc_reduce<G2_GW>(A);
    int spec = 0;
    if (act && sub == 0) {
      const int verdict = wide ? -2 : A.n > 3 ? (d.g2_exact_only ? -1 : g2_screen(A, q, d.g2_r, d.g2_inv_r)) : 0;
      if (verdict < 0) d.g2_big[so + atomicAdd(&d.g2_nbig[s], 1)] = verdict == -1 ? (v | G2_Q_EXACT) : v;
      else d.vbin[so + v] = verdict ? (int)(q.z * 10) : 0x7fffffff;
      spec = verdict == 1 && (int)(q.z * 10) == pred;
    }
    // ---- speculative marks (see k_g2_mode): an accepted voxel of the PREDICTED mode bin marks its neighbours at once, while its candidate ranges are at hand
    spec = __shfl(spec, lane0, 64);
    const int nmark = spec ? ncand : 0;
    int mark_max = nmark;
    mark_max = wave_imax(mark_max);
    for (int c0 = 0; c0 < mark_max; c0 += 4 * G2_GW) {
      float4 pc[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + G2_GW * u + sub; int k = 0;
#pragma unroll
        for (int r = 0; r < 9; ++r) if (c >= rp[r] && c < rp[r + 1]) k = rb[r] + (c - rp[r]);
        pc[u] = sp[c < nmark ? k : 0];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + G2_GW * u + sub; const float4 p = pc[u];
        if (c < nmark && sqdist(q.x, q.y, q.z, p.x, p.y, p.z) < d.leaf_r2) d.is_ground[so + __float_as_int(p.w)] = tag_spec;
      }
    }
  }
}
// The queued voxels, one WAVE per voxel: the same screen with sixty-four lanes for the voxels k_g2_cov did not walk; the ordered sums for those
// the screen leaves open, up to G2_MID_CAP neighbours in the wave's 5 KiB slice of LDS (gather with ballot compaction, rank by counting — the
// (d², index) keys are unique —, coordinates to their rank, ordered sums as three chains in three lanes).  Settled entries of the queue are
// complemented; what is left (open AND more than G2_MID_CAP neighbours) goes to k_g2_cov_big.  A wave works in its own slice of LDS: the
// order of ONE wave's LDS accesses — which the hardware keeps — is all its lanes need, not a workgroup barrier.
#ifndef G2_MID_CAP
#define G2_MID_CAP 256   // (round 5: 1024 — 80 KB of LDS per workgroup, two per CU — made this kernel wait for CUs the other lanes' kernels had not filled: 149 µs alone, 1 030 in the pipeline; the voxels the screen leaves open have a few dozen neighbours)
#endif
__global__ __launch_bounds__(MOR_BT) void k_g2_cov_mid(MorDev d) {
  const int s = blockIdx.y + d.s0, nbig = d.g2_nbig[s], bxq = blockIdx.x, gq = gridDim.x;   // (spread over all XCDs: the queues are uneven across streams, and a workgroup holds 80 KB of LDS)
  const size_t so = (size_t)s * d.Nmax;
  const int wv = wave_id(), lane = lane_id();
  __shared__ unsigned long long l_key[MOR_BT / 64][G2_MID_CAP];
  __shared__ float l_x[MOR_BT / 64][G2_MID_CAP], l_y[MOR_BT / 64][G2_MID_CAP], l_z[MOR_BT / 64][G2_MID_CAP];
  const float4 *sp = d.sorted + so;
  const int zbase = d.zbase[s]; const MorGrid G = stream_grid(d, s);   // (the lattice with the stream's own z layers)
  for (int w0 = bxq * (MOR_BT / 64); w0 < nbig; w0 += gq * (MOR_BT / 64)) {
   {
    const int w = w0 + wv;
    if (w >= nbig) continue;   // (wave-uniform; nothing below synchronises the workgroup)
    const int qe = d.g2_big[so + w], v = qe & ~G2_Q_EXACT;
    const float4 q = d.vcent[so + v];
    int rb0 = 0, rlen = 0;
    if (lane < 9) { int cx, cy, cz; bool cl; grid_cell(G, q, 0.f, zbase, cx, cy, cz, cl); g2_row_range(d, s, G, cx, cy, cz, lane, rb0, rlen); }
    int rb[9], rp[10]; rp[0] = 0;
#pragma unroll
    for (int r = 0; r < 9; ++r) { rb[r] = wave_bcast(rb0, r); rp[r + 1] = rp[r] + wave_bcast(rlen, r); }
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
        if (verdict == 1 && (int)(q.z * 10) == d.g2_used[s])   // speculative marks (k_g2_mode)
          for (int c0 = 0; c0 < rp[9]; c0 += 64) { const int c = c0 + lane; if (c < rp[9]) { const float4 p = sp[cand(c)]; if (sqdist(q.x, q.y, q.z, p.x, p.y, p.z) < d.leaf_r2) d.is_ground[so + __float_as_int(p.w)] = 2 * d.frame_no + 1; } }
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
    if (!mine) {   // (the tags come off: k_g2_cov_big takes every entry ≥ 0 — from the batch-wide list of open voxels; a list that is full sends its workgroups through the streams' queues)
      if (lane == 0) { d.g2_big[so + w] = v; const int e = atomicAdd(&d.g2_nopen[0], 1); if (e < d.g2_opencap) d.g2_open[e] = make_int2(s, v); }
      continue;
    }
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
    if (n > 3 && acc3 && (int)(q.z * 10) == d.g2_used[s]) {   // speculative marks (k_g2_mode): the keys stay where the gather put them, a neighbour's index in the low half
      for (int e = lane; e < n; e += 64) d.is_ground[so + (int)(unsigned)l_key[wv][e]] = 2 * d.frame_no + 1;
    }
    if (lane == 0) { d.vbin[so + v] = (n > 3 && acc3) ? (int)(q.z * 10) : G2_V_NONE; d.g2_big[so + w] = ~v; }
    wave_lds_fence();
   }
  }
}
// what the middle tier left: one 1024-thread workgroup each (the LDS lets only one live on a CU anyway: sixteen waves sort four times faster than
// four), up to G2_CAP neighbours in 128 KiB of LDS
#define G2_BIG_T 1024
#define G2_BIG_WG 256   // workgroups of k_g2_cov_big (one per CU: its LDS lets only one live there)
__device__ __forceinline__ void g2_big_voxel(const MorDev &d, int s, int v, unsigned long long *key, float *px, float *py, float *pz, int *cnt, int *l_rng, float *acc) {   // all threads of the workgroup
  const size_t so = (size_t)s * d.Nmax;
  if (threadIdx.x == 0) *cnt = 0;
  __syncthreads();
  const float4 q = d.vcent[so + v];
  g2_gather(d, s, q, key, cnt, G2_CAP, l_rng);
  __syncthreads();
  const int n = *cnt;
  int bin = 0x7fffffff;
  if (n > G2_CAP) { if (threadIdx.x == 0) mor_raise(d, s, 16u); }
  else if (n > 3) bin = g2_voxel_bin<G2_CHUNK>(d, so, q, key, n, px, py, pz, acc);
  if (bin != 0x7fffffff && bin == d.g2_used[s]) for (int i = threadIdx.x; i < n; i += G2_BIG_T) d.is_ground[so + (int)(unsigned)key[i]] = 2 * d.frame_no + 1;   // speculative marks (k_g2_mode)
  if (threadIdx.x == 0) d.vbin[so + v] = bin;
  __syncthreads();
}
// Round 5 gave every stream eight workgroups that walked the stream's queue (thousands of entries, nearly all settled) one dependent global load at a time and worked through whatever
// open voxels fell to them — 427 µs alone, set by the stream with the most.  Now k_g2_cov_mid lists what it leaves open for the whole batch and the workgroups take entries by ticket.
__global__ __launch_bounds__(G2_BIG_T) void k_g2_cov_big(MorDev d) {
  __shared__ unsigned long long key[G2_CAP];
  __shared__ float px[G2_CHUNK], py[G2_CHUNK], pz[G2_CHUNK];
  __shared__ int cnt, l_rng[18], l_take;
  __shared__ float acc[6];
  const int nopen = d.g2_nopen[0];
  if (nopen <= d.g2_opencap) {
    for (;;) {
      if (threadIdx.x == 0) l_take = atomicAdd(&d.g2_nopen[1], 1);
      __syncthreads();
      const int e = l_take;
      __syncthreads();
      if (e >= nopen) return;
      const int2 sv = d.g2_open[e];
      g2_big_voxel(d, sv.x, sv.y, key, px, py, pz, &cnt, l_rng, acc);
    }
  }
  // the list overflowed (never seen: 256 open voxels per stream on average): stream by stream through the queues, 1024 entries at a time
  __shared__ int l_open[G2_BIG_T], l_nopen;
  for (int s = (int)blockIdx.x + d.s0; s < d.s0 + d.B; s += (int)gridDim.x) {
    const int nbig = d.g2_nbig[s]; const size_t so = (size_t)s * d.Nmax;
    for (int base = 0; base < nbig; base += G2_BIG_T) {
      if (threadIdx.x == 0) l_nopen = 0;
      __syncthreads();
      { const int w = base + (int)threadIdx.x; const int v_ = w < nbig ? d.g2_big[so + w] : -1; if (v_ >= 0) l_open[atomicAdd(&l_nopen, 1)] = v_; }
      __syncthreads();
      const int no = l_nopen;
      for (int oi = 0; oi < no; ++oi) g2_big_voxel(d, s, l_open[oi], key, px, py, pz, &cnt, l_rng, acc);
      __syncthreads();
    }
  }
}
__global__ __launch_bounds__(MOR_BT) void k_g2_mode(MorDev d) {
  int s = blockIdx.x + d.s0, V = d.info[s].n_occ;
  const size_t so = (size_t)s * d.Nmax;
  __shared__ int hist[4096], best_cnt, best_bin;
  for (int i = threadIdx.x; i < 4096; i += MOR_BT) hist[i] = 0;
  if (threadIdx.x == 0) { best_cnt = 0; best_bin = 0x7fffffff; }
  __syncthreads();
  for (int v0 = threadIdx.x; v0 < V; v0 += 8 * MOR_BT) {   // eight bin words per thread and round trip (one at a time this one-workgroup kernel was 90 dependent round trips: 72 µs)
    int bb[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) bb[u] = d.vbin[so + min(v0 + u * MOR_BT, V - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int b = bb[u];
      if (v0 + u * MOR_BT >= V || b == 0x7fffffff) continue;
      if (b < -2048 || b >= 2048) { mor_raise(d, s, 8u); continue; }
      atomicAdd(&hist[b + 2048], 1);
    }
  }
  __syncthreads();
  {  // the fullest bin, ties → the smallest bin: every thread over its sixteen bins, then the wave by shuffles, then one atomic pair per wave (4 096 atomics on one LDS word each way before)
    int cnt = 0, bin = 0x7fffffff;
    for (int i = threadIdx.x; i < 4096; i += MOR_BT) { const int h = hist[i]; if (h > cnt) { cnt = h; bin = i - 2048; } }   // (ascending i: a later bin with the same count does not replace an earlier one)
    { const int2 best = wave_allreduce(make_int2(cnt, bin), [](int2 a, int2 b) { return (b.x > a.x || (b.x == a.x && b.y < a.y)) ? b : a; }); cnt = best.x; bin = best.y; }
    if (lane_id() == 0) atomicMax(&best_cnt, cnt);
    __syncthreads();
    if (lane_id() == 0 && cnt > 0 && cnt == best_cnt) atomicMin(&best_bin, bin);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    // The mode bin is known only now, but it hardly ever moves from frame to frame (it is the height of the ground): the kernels that take the verdicts have marked the neighbours of
    // the accepted voxels of the bin they BET on (g2_used: pass A's snapshot of the latest mode any frame of the stream has reported) with the tag 2·frame + 1 — while their candidate
    // ranges were at hand, without k_g2_mark's chains of lookups.  If the bet holds, those marks are the frame's ground and k_g2_mark has nothing to do for this stream; if not
    // (the first frame, a ramp), the frame's tag is 2·frame + 2 and k_g2_mark writes it for the true mode bin — the marks of the lost bet are simply never looked at.
    const int pred = d.g2_used[s];
    d.g2_tag[s] = (best_bin != 0x7fffffff && best_bin == pred) ? 2 * d.frame_no + 1 : 2 * d.frame_no + 2;
    st_agent(&d.g2_pred[s], best_bin);
    d.mode_bin[s] = best_bin; d.g2_nbig[s] = 0;
    if (s == d.s0) { d.g2_nopen[0] = 0; d.g2_nopen[1] = 0; }   // (the batch-wide list of open voxels: its readers are done — a kernel boundary lies in between)
  }   // (the queue of big voxels is empty again for the next frame on this copy)
}
// ground = union of the neighbour lists of the dominant bin's voxels (:184-191, de-duplicated): every trimmed point within the radius of such a
// voxel's centroid is marked (no list, no sort needed here).  Waves look at 64 voxels at a time and take the mode bin's voxels among them FOUR at a
// time, sixteen lanes each as in k_g2_cov (a centroid has 3.5 candidates: a whole wave per voxel — the first form — kept 55 lanes idle); voxels with
// more than G2_NARROW_CAND candidates are left to the whole wave afterwards.
__global__ __launch_bounds__(MOR_BT) void k_g2_mark(MorDev d) {
  int s, bxm; map_block(d.B, 128, s, bxm);
  const int V = d.info[s].n_occ, mode = d.mode_bin[s];
  const size_t so = (size_t)s * d.Nmax;
  if (mode == 0x7fffffff || d.g2_tag[s] == 2 * d.frame_no + 1) return;   // (no accepted voxel at all; or the bet on the mode bin held: the marks are there already)
  const int lane = lane_id(), nw = 128 * (MOR_BT / 64), grp = lane >> 4, sub = lane & 15;
  const float4 *sp = d.sorted + so;
  const MorGrid G = stream_grid(d, s);   // (the lattice with the stream's own z layers)
  const int zbase = d.zbase[s], tag = 2 * d.frame_no + 2;   // the frame's tag when the bet was lost (never 0, never an earlier frame's on this copy of the array): nothing has to be cleared
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
      if (act && sub < 9) { int cx, cy, cz; bool cl; grid_cell(G, q, 0.f, zbase, cx, cy, cz, cl); g2_row_range(d, s, G, cx, cy, cz, sub, rb0, rlen); }
      int rb[9], rp[10]; rp[0] = 0;
#pragma unroll
      for (int r = 0; r < 9; ++r) { rb[r] = __shfl(rb0, (lane & 48) + r, 64); rp[r + 1] = rp[r] + __shfl(rlen, (lane & 48) + r, 64); }
      const bool wide = rp[9] > G2_NARROW_CAND;   // (uniform in the group)
      { const unsigned long long wb = __ballot(act && wide && sub == 0);   // one bit per group with a wide voxel: its voxel goes to the wave's list
        unsigned long long t = wb; while (t) { const int gl = __ffsll((long long)t) - 1; t &= t - 1; wide_m |= 1ull << __shfl(l, gl, 64); } }
      const int ncand = (act && !wide) ? rp[9] : 0;
      int wave_max = ncand;
      wave_max = wave_imax(wave_max);
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

