// kernels_track_out.h — part of mor_kernels.hip (one translation unit: #included there, in dependency order; not a stand-alone header).
// T1 + F1 (:415-514, :613-696): tracking on the device, filterCloud: removal bits and the one-pass output.
// Reference citations are file:line of /root/reference/src/MovingObjectRemoval.cpp.
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
__device__ __forceinline__ void track_push_body(const MorDev &d, int s) {   // a 64-thread workgroup
  const int lane = threadIdx.x;
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
              if (m) match = wave_bcast(pr.y, __ffsll((long long)m) - 1);   // first pair whose query is `track`
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
__global__ __launch_bounds__(64) void k_track_push(MorDev d) { track_push_body(d, blockIdx.x + d.s0); }
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
template <int NT> __device__ __forceinline__ void track_filter_body(const MorDev &d, int s, unsigned *l_mov /* Kcap / 32 words */) {   // NT threads (FLT_T as a kernel of its own, 64 behind the tracking step)
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
  for (int k = tid; k < (d.Kcap + 31) / 32; k += NT) l_mov[k] = 0u;
  if (fits) for (int k = tid; k < K; k += NT) { l_cen[k] = d.centroid[d.cur][ko + k]; l_size[k] = off[k + 1] - off[k]; l_det[k] = d.det[ko + k]; }
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
      total += wave_sum(mine);
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
  for (int k = tid; k < K; k += NT) if ((l_mov[k >> 5] >> (k & 31)) & 1u) removed += (unsigned)(fits ? l_size[k] : off[k + 1] - off[k]);
  removed = wave_sum(removed);
  __shared__ unsigned l_rem[NT / 64];
  if (lane == 0) l_rem[tid >> 6] = removed;
  __syncthreads();
  unsigned *gm = d.moving + (size_t)s * (d.Kcap / 32 + 2);
  for (int k = tid; k < (K + 31) / 32; k += NT) gm[k] = l_mov[k];
  if (tid == 0) {
    unsigned rem = 0; for (int w = 0; w < NT / 64; ++w) rem += l_rem[w];
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
  track_filter_body<FLT_T>(d, blockIdx.x + d.s0, l_mov);
}
// The tracking step of a push and the loop of the filterCloud that follows it as ONE launch (round 6): in asynchronous mode the engine holds the tracking step of a push back until
// it knows what comes next — a filterCloud (the usual case: this kernel; one launch and one queueing delay less per frame, the tracks' head read once) or something else (then
// k_track_push alone).  Both are the work of one wave per stream; the second part reads what the first left in global memory from the same CU, behind a workgroup barrier.
__global__ __launch_bounds__(64) void k_track_push_filter(MorDev d) {
  const int s = blockIdx.x + d.s0;
  track_push_body(d, s);
  __threadfence_block();
  __syncthreads();
  __shared__ unsigned l_mov[MOR_KCAP_MAX / 32];
  track_filter_body<64>(d, s, l_mov);
}
// one point of the filtered cloud into slot idx of a caller-provided buffer: packed (x,y,z,intensity), or — out_step32 — the 32-byte PointXYZI record of toPCLPointCloud2 (:690):
// x@0 y@4 z@8 (1.0f @12) intensity@16, zeros behind it; a lane writes 32 consecutive bytes, a wave 2 KB
__device__ __forceinline__ void out_store(float4 *dst, int idx, const float4 &p, int step32) {
  if (step32) { st_stream(&dst[2 * (size_t)idx], make_float4(p.x, p.y, p.z, 1.0f)); st_stream(&dst[2 * (size_t)idx + 1], make_float4(p.w, 0.f, 0.f, 0.f)); }
  else st_stream(&dst[idx], p);
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
    for (int i = base + threadIdx.x; i < min(base + MOR_TILE, G); i += FLT_T) out_store(out, n_keep + i, ld_stream(&og[d.Nmax + i]), d.out_step32);
    return;
  }
  if (t >= nto) return;
  const int K = d.info[s].K;
  for (int k = threadIdx.x; k < (K + 31) / 32; k += FLT_T) l_mov[k] = gm[k];
  __syncthreads();
  float4 *dst = d.out_ptrs ? d.out_ptrs[s] : og + (d.Nmax - n_keep);
  const int step32 = d.out_ptrs ? d.out_step32 : 0;
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
      an = wave_sum(an);
      if (lane_id() == 0) l_ex[2] = ex + an;
    }
    __syncthreads();
    int r = l_ex[2];
    ex = r + tot; t_prev = t;
    for (int w = 0; w < wave_id(); ++w) r += sh[w];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int i = base + it * 64 + lane_id();
      if ((mk[it] >> lane_id()) & 1ull) out_store(dst, r + __popcll(mk[it] & lanemask_lt()), ld_stream(&d.cloud[so + i]), step32);   // (the filtered cloud is the caller's; the cloud is not read again on the device)
      r += __popcll(mk[it]);
    }
    t = __builtin_amdgcn_readfirstlane(l_ex[1]);
    __syncthreads();
  }
}

