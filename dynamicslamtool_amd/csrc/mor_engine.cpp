// mor_engine.cpp — host side of libmor_hip.so: the C ABI of include/mor_hip.h, buffer management
// in HBM, the per-batch launch sequence, and the glue to the host tracker.
// Mirrors MovingObjectRemoval::pushRawCloudAndPose / filterCloud
// (/root/reference/src/MovingObjectRemoval.cpp:516-611, :613-696) for B independent streams.
#include "mor_device.h"
#include "../../include/mor_hip.h"
#include <sched.h>
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#define MOR_ARGS_RING 8

// The frame pipeline wants its four lane streams on four different hardware queues.  The ROCm runtime multiplexes
// streams onto GPU_MAX_HW_QUEUES (default 4) queues in creation order; the three copy streams of the device (below) are created first,
// then the lanes sf, sc, sm, sb — seven streams: with the default of four queues the lanes share queues with each other and with the copy
// streams (107 k instead of 180 k frame-pairs/s), so the PROCESS sets GPU_MAX_HW_QUEUES=8 (or more) before HIP initialises — the integrator's
// decision, this library never touches the environment (the Python binding and the replay driver set it).  Measured on
// MI355X (round 4, GPU_MAX_HW_QUEUES=16): 5 / 6 / 8 lanes give 169 / 176 / 184 k against 186 k with four — more frames in flight do not help,
// the GPU's memory pipeline is busy with four.

// The three copy streams of a device (read-backs, host → device, device → host) are created ONCE per process and device and shared by all
// batches on it (copies of different batches queue behind each other; they share the link anyway).  The runtime binds a stream to its DMA
// engine when the stream is created, and streams created after others have come and gone get worse ones: the second batch of a process
// moved device → host at 10 GB/s instead of 55 (exp/e2e_probe.py --prelude).
struct MorCopyStreams { hipStream_t st = nullptr, h2d = nullptr, d2h = nullptr; };
static MorCopyStreams *copy_streams(int device) {
  static std::map<int, MorCopyStreams> all; static std::mutex mu;   // (two adapters may be constructed from two threads: the map and the creation are guarded; the streams are never released)
  std::lock_guard<std::mutex> lk(mu);
  MorCopyStreams &c = all[device];
  if (!c.st) {
    if (hipStreamCreateWithFlags(&c.st, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&c.h2d, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&c.d2h, hipStreamNonBlocking) != hipSuccess) return nullptr;
  }
  return &c;
}

static thread_local std::string g_last_error;
static int set_error(int code, const char *fmt, ...) {
  char buf[512]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  g_last_error = buf; return code;
}
#define HIP_TRY(expr)                                                                                         \
  do { hipError_t e_ = (expr); if (e_ != hipSuccess) return set_error(MOR_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); } while (0)

// ------------------------------------------------------------------ per-kernel event timing
struct MorLaunchTimer {
  bool enabled = false;
  struct Rec { int id; hipEvent_t a, b; };
  std::vector<Rec> recs; size_t used = 0;
  double ms[MK_COUNT] = {0}; uint32_t launches[MK_COUNT] = {0};
  struct Span { int id; float t0, t1; };
  std::vector<Span> timeline;   // kernels of the last collected leg, milliseconds since its first recorded event (all streams on one clock)
  void collect() {
    if (used) timeline.clear();
    for (size_t i = 0; i < used; ++i) {
      float t = 0;
      if (hipEventElapsedTime(&t, recs[i].a, recs[i].b) == hipSuccess) {
        ms[recs[i].id] += t; launches[recs[i].id]++;
        float s0 = 0;
        if (hipEventElapsedTime(&s0, recs[0].a, recs[i].a) == hipSuccess) timeline.push_back({recs[i].id, s0, s0 + t});
      }
    }
    used = 0;
  }
  ~MorLaunchTimer() { for (auto &r : recs) { hipEventDestroy(r.a); hipEventDestroy(r.b); } }
};
void mor_timer_begin(MorLaunchTimer *tm, int id, hipStream_t st) {
  if (!tm || !tm->enabled) return;
  if (tm->used == tm->recs.size()) { MorLaunchTimer::Rec r; r.id = id; hipEventCreate(&r.a); hipEventCreate(&r.b); tm->recs.push_back(r); }
  tm->recs[tm->used].id = id;
  hipEventRecord(tm->recs[tm->used].a, st);
}
void mor_timer_end(MorLaunchTimer *tm, int id, hipStream_t st) {
  if (!tm || !tm->enabled) return;
  (void)id; hipEventRecord(tm->recs[tm->used].b, st); tm->used++;
}

// ------------------------------------------------------------------ batch
struct PoseTf { double R[3][3], o[3]; };

struct mor_batch {
  mor_params p; int n_bad, n_good, B, device; uint64_t Nmax;
  hipStream_t st = nullptr; hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // st: copies of read-backs; ev: push/filter timing
  // Four in-order HIP streams ("lanes") form a software pipeline over frames (see mor_push_batch): frame k runs ALL its
  // pieces, then its filterCloud, on stream k % n_lanes, so every stream carries the same work whatever the pieces cost.
  // Every array a frame writes exists once per frame in flight, so the only orderings between frames are the true ones:
  // the pair stage of frame k follows the clusters of frame k − 1 (ca), the tracking steps (last piece, filterCloud's loop)
  // follow the previous tracking step, the output kernels of filterCloud run in frame order, and frame k waits for frame
  // k − depth to have left its copy of the arrays.  (Round 2 also ordered piece p of frame k behind piece p of frame k − 1 to
  // protect shared scratch arrays: sixteen more event records and waits per frame — each a packet the command processor
  // handles between two kernels of a lane.  Rounds 1–2 also had a stage schedule, pieces on fixed streams; removed.)
  hipStream_t sf = nullptr, sc = nullptr, sm = nullptr, sb = nullptr;   // lanes 1 … 4
  hipEvent_t ev_clusters[MOR_MAX_SLOTS] = {}, ev_pairs[MOR_MAX_SLOTS] = {}, ev_tpush[MOR_MAX_SLOTS] = {}, ev_back[MOR_MAX_SLOTS] = {};
  int n_pieces = 0, piece_id[MOR_MAX_PIECES] = {};   // the pieces of a push in order (ids: mor_device.h)
  int n_lanes = 4;
  hipStream_t extra[4] = {nullptr, nullptr, nullptr, nullptr};   // lanes 5 … 8 (MOR_LANES; measured: more than four busy streams are served worse)
  hipStream_t lane_stream(uint64_t k) const { const int i = (int)(k % (uint64_t)n_lanes); return i == 0 ? sf : i == 1 ? sc : i == 2 ? sm : i == 3 ? sb : extra[i - 4]; }
  hipEvent_t ev_track[MOR_MAX_SLOTS] = {};      // recorded after the tracking step of a push / the tracking loop of a filterCloud
  hipEvent_t *last_track = nullptr;             // the latest of them
  // Copies between host and device memory run on two streams of their own, one per direction: the runtime serves a stream's copies by
  // one DMA engine, and with the copies on the lanes the two directions shared engines — host → device and device → host took turns
  // (4.85 ms per step of 64 × 120 000 points each way instead of 2.8 ms with both directions at once, exp/tools/duplex.cpp).
  hipStream_t s_h2d_[1] = {nullptr}, s_d2h_[1] = {nullptr};   // (one per direction: with two per direction the four streams shared engines again — 7 k instead of 16 k frame-pairs/s end to end; blobs that lie back to back in host memory travel as ONE copy, which is what makes a stream reach the link rate)
  hipEvent_t ev_h2d[MOR_MAX_SLOTS] = {}, ev_d2h[MOR_MAX_SLOTS] = {};   // staged input of frame k is on the device; its filtered clouds have left the output staging area
  bool d2h_used[MOR_MAX_SLOTS] = {};
  hipEvent_t ev_out[MOR_MAX_SLOTS] = {};        // recorded after the output kernels of a filterCloud whose clouds leave by DMA (the copies follow it)
  // The stream every such event was last recorded on: a wait for an event of the SAME stream is implied by stream order, and every
  // packet the command processor does not have to fetch, resolve and signal is a few microseconds of a lane (exp/gaps.py)
  hipStream_t ev_back_st[MOR_MAX_SLOTS] = {}; hipStream_t last_track_st = nullptr;
  MorDev dtemp[MOR_MAX_DEPTH];               // descriptor templates, frame k uses dtemp[k % depth] (static part + pointers)
  MorDev d;                                  // descriptor of the latest pushed frame
  MorStreamArgs *d_args_s[MOR_MAX_DEPTH] = {};
  std::vector<void *> dev_allocs, host_allocs;
  MorStreamArgs *h_args_ring = nullptr, *h_args = nullptr;   // pinned ring of MOR_ARGS_RING slots (async pushes), current slot
  uint64_t pipe_depth = 4, n_slots = 5;       // frames in flight (= copies of the per-frame arrays); cluster-array slots = depth + 1
  int env_cg_p = 0;         // test knob from the environment (MOR_CG_P: slabs per stream of the cell graph), read once at creation
  bool async = false, pending = false;   // async: push/filter only enqueue; pending: work enqueued since the last wait
  bool track_held = false, fuse_track = true;   // asynchronous mode: the tracking step of the latest push has not been launched yet — the filterCloud that follows launches it together with its own loop (k_track_push_filter); anything else launches it alone first (flush_track).  MOR_FUSE_TRACK=0: never held back
  float4 **h_outptrs = nullptr, **d_outptrs = nullptr;   // caller-provided output pointers: pinned ring of MOR_ARGS_RING tables (one per filterCloud in flight), device copy per frame in flight
  hipEvent_t outptr_ev[MOR_ARGS_RING] = {}; uint64_t n_filters = 0, n_filter_calls = 0;
  float4 *d_outstage = nullptr;   // [depth][B][Nmax]  asynchronous filterCloud into host memory: the filtered clouds are assembled here and leave by DMA (allocated at the first such call)
  std::vector<uint64_t> last_n;   // points per stream of the latest push
  unsigned char *d_stage = nullptr; size_t stage_stride = 0;   // staging for host-resident input blobs: one area per frame in flight (frame k: area k mod depth), so the copy of frame k + 1 runs beside the kernels of frame k
  std::vector<PoseTf> prev_pose;
  uint64_t frame = 0;
  bool filtered = false;
  float push_ms = 0, filter_ms = 0;
  MorLaunchTimer timer;
};

template <class T> static bool dalloc(mor_batch *b, T *&ptr, size_t n) {
  void *v = nullptr; if (hipMalloc(&v, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return false;
  b->dev_allocs.push_back(v); ptr = (T *)v; return true;
}
template <class T> static bool halloc(mor_batch *b, T *&ptr, size_t n) {
  void *v = nullptr; if (hipHostMalloc(&v, std::max<size_t>(n, 1) * sizeof(T), hipHostMallocDefault) != hipSuccess) return false;
  b->host_allocs.push_back(v); memset(v, 0, std::max<size_t>(n, 1) * sizeof(T)); ptr = (T *)v; return true;
}

// tf::poseMsgToTF (:524) — quaternion renormalised only when |len² − 1| > 0.1, setRotation with s = 2/len²
static void pose_to_tf(const double *p, PoseTf &t) {
  double x = p[3], y = p[4], z = p[5], w = p[6];
  double l2 = x * x + y * y + z * z + w * w;
  if (std::fabs(l2 - 1.0) > 0.1) { double l = std::sqrt(l2); x /= l; y /= l; z /= l; w /= l; l2 = x * x + y * y + z * z + w * w; }
  double s = 2.0 / l2, xs = x * s, ys = y * s, zs = z * s;
  double wx = w * xs, wy = w * ys, wz = w * zs, xx = x * xs, xy = x * ys, xz = x * zs, yy = y * ys, yz = y * zs, zz = z * zs;
  t.R[0][0] = 1.0 - (yy + zz); t.R[0][1] = xy - wz; t.R[0][2] = xz + wy;
  t.R[1][0] = xy + wz; t.R[1][1] = 1.0 - (xx + zz); t.R[1][2] = yz - wx;
  t.R[2][0] = xz - wy; t.R[2][1] = yz + wx; t.R[2][2] = 1.0 - (xx + yy);
  t.o[0] = p[0]; t.o[1] = p[1]; t.o[2] = p[2];
}
// cb.ps.inverseTimes(ca.ps) (:536) in fp64, cast to the fp32 matrix pcl_ros::transformPointCloud applies
static void relative_transform(const PoseTf &cb, const PoseTf &ca, float m[12]) {
  double v[3] = {ca.o[0] - cb.o[0], ca.o[1] - cb.o[1], ca.o[2] - cb.o[2]};
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) m[4 * i + j] = (float)(cb.R[0][i] * ca.R[0][j] + cb.R[1][i] * ca.R[1][j] + cb.R[2][i] * ca.R[2][j]);
    m[4 * i + 3] = (float)(cb.R[0][i] * v[0] + cb.R[1][i] * v[1] + cb.R[2][i] * v[2]);
  }
}

static int configure(mor_batch *b) {
  const mor_params &p = b->p; MorDev &d = b->d;
  if (!(p.ec_distance_threshold > 0.f) || !(p.trim_x > 0.f) || !(p.trim_y > 0.f)) return set_error(MOR_ERR_INVALID, "ec_distance_threshold, trim_x, trim_y must be > 0");
  if (p.method_choice == 2 && p.opc_normalization_factor <= 0) return set_error(MOR_ERR_INVALID, "opc_normalization_factor must be > 0 for method 2");
  if (p.method_choice == 2 && !(p.opc_resolution > 0.f)) return set_error(MOR_ERR_INVALID, "opc_resolution must be > 0");
  if (p.ground_method != 0 && p.ground_method != 1) return set_error(MOR_ERR_INVALID, "ground_method must be 0 (crop box) or 1 (voxel covariance)");
  if (p.ground_method == 1 && !(p.gp_leaf > 0.f)) return set_error(MOR_ERR_INVALID, "gp_leaf must be > 0 for the voxel-covariance ground removal");
  d.B = b->B; d.Btot = b->B; d.s0 = 0; d.Nmax = (int)b->Nmax;
  long long mn = std::max<long long>(p.min_cluster_size, 1);
  d.Kcap = (int)std::min<long long>((long long)b->Nmax / mn + 1, MOR_KCAP_MAX);
  d.Kcap = (d.Kcap + 31) & ~31;   // (a whole number of words of the removal bit mask)
  d.tiles_max = (int)((b->Nmax + MOR_TILE - 1) / MOR_TILE);
  d.trim_x = p.trim_x; d.trim_y = p.trim_y; d.trim_z = p.trim_z; d.gp_limit = p.gp_limit;
  double tol = (double)p.ec_distance_threshold; d.r2 = (float)(tol * tol);   // KdTreeFLANN::radiusSearch: (float)(radius·radius)
  d.min_cs = p.min_cluster_size; d.max_cs = p.max_cluster_size;
  d.pde_lb = p.pde_lb; d.pde_ub = p.pde_ub; d.pde_thr = (double)p.pde_distance_threshold; d.vol_thr = (double)p.volume_constraint;
  d.opc_res = (double)p.opc_resolution; d.opc_inv_res = p.opc_resolution > 0.f ? 1.0 / (double)p.opc_resolution : 0.0; d.method = p.method_choice; d.opc_norm = p.opc_normalization_factor; d.vol_abs_int = p.volume_abs_int ? 1 : 0; d.opc_anchor_half = p.opc_anchor ? 1 : 0;
  // grid: cell edge 0.57·r (cell diagonal 0.987·r < r ⇒ a cell is a clique; the 1.3 % margin dwarfs the
  // fp32 rounding of the cell map, ≤ 1e-3 cell at ≤ 2048 cells per axis)
  float cs = p.ec_distance_threshold * 0.57f;
  float zlo = p.gp_limit, zhi = std::max(p.trim_z, p.gp_limit);
  d.gmode = p.ground_method;
  const double zspan_voxel_mode = 64.0;   // voxel variant: no z crop; grids hang on the lowest trimmed point and cover this span
  double nx = std::floor(2.0 * p.trim_x / cs) + 1, ny = std::floor(2.0 * p.trim_y / cs) + 1, nz = std::floor((double)(zhi - zlo) / cs) + 1;
  if (d.gmode == 1) nz = std::min(1024.0, std::ceil(zspan_voxel_mode / cs) + 2);
  if (nx > 2048 || ny > 2048 || nz > 1024 || nx * ny * nz >= 2147483648.0 || ny * nz > 1048576.0)
    return set_error(MOR_ERR_INVALID, "grid %gx%gx%g cells too large: trim box too big for ec_distance_threshold %g", nx, ny, nz, (double)p.ec_distance_threshold);
  d.g.nx = (int)nx; d.g.ny = (int)ny; d.g.nz = (int)nz; d.g.nrows = d.g.ny * d.g.nz;
  d.g.keybits = 1; while ((1ll << d.g.keybits) < (long long)(nx * ny * nz)) ++d.g.keybits;
  d.cell_passes = (d.g.keybits + 7) / 8;
  d.g.ox = -p.trim_x; d.g.oy = -p.trim_y; d.g.oz = zlo; d.g.cs = cs; d.g.inv_cs = 1.0f / cs; d.g.mode = 0; d.g.ibx = d.g.iby = 0;
  d.gv = d.g; d.voxel_passes = 0; d.leaf_r2 = 0.f; d.g2_r = 0.0; d.g2_inv_r = 0.0; d.g2_bits = nullptr; d.g2_dir = nullptr; d.g2_nch = 0;
  if (d.gmode == 1) {   // VoxelGrid lattice: cells at absolute multiples of the leaf (pcl::VoxelGrid: floor(x·inv_leaf)), :110-113
    MorGrid &v = d.gv; v.mode = 1; v.cs = p.gp_leaf; v.inv_cs = 1.0f / p.gp_leaf;
    v.ibx = (int)std::floor(-p.trim_x * v.inv_cs); v.iby = (int)std::floor(-p.trim_y * v.inv_cs);
    double vx = (double)((int)std::floor(p.trim_x * v.inv_cs) - v.ibx + 1), vy = (double)((int)std::floor(p.trim_y * v.inv_cs) - v.iby + 1);
    double vz = std::min(1024.0, std::ceil(zspan_voxel_mode / p.gp_leaf) + 2);
    if (vx > 2048 || vy > 2048 || vx * vy * vz >= 2147483648.0 || vy * vz > 1048576.0)
      return set_error(MOR_ERR_INVALID, "voxel lattice %gx%gx%g too large: trim box too big for gp_leaf %g", vx, vy, vz, (double)p.gp_leaf);
    v.nx = (int)vx; v.ny = (int)vy; v.nz = (int)vz; v.nrows = v.ny * v.nz;
    v.keybits = 1; while ((1ll << v.keybits) < (long long)(vx * vy * vz)) ++v.keybits;
    d.voxel_passes = (v.keybits + 7) / 8;
    d.g2_nch = (v.nx + 511) / 512;
    double lf = (double)p.gp_leaf; d.leaf_r2 = (float)(lf * lf); d.g2_r = std::sqrt((double)d.leaf_r2) * 1.0001 + 1e-6; d.g2_inv_r = 1.0001 / std::sqrt((double)d.leaf_r2);   // radiusSearch(…, gp_leaf): (float)(radius·radius)
  }
  d.score_R = p.pde_ub > 0.f ? (int)std::floor(std::sqrt((double)p.pde_ub) * d.g.inv_cs * 1.001) + 1 : 1;
  d.g2_passa2 = getenv("MOR_G2_PASSA2") ? atoi(getenv("MOR_G2_PASSA2")) != 0 : 0;
  d.two_pass_split = (getenv("MOR_SINGLE_PASS_SPLIT") && atoi(getenv("MOR_SINGLE_PASS_SPLIT")) == 0) ? 1 : 0;   // default: the single-read split (k_split); MOR_SINGLE_PASS_SPLIT=0: count pass + scatter pass (every split of the frame; MOR_G2_PASSA2=1: pass A of the voxel ground variant only)
  {  // workgroups per stream of the single-read split: what the device holds at once, shared out over the streams (a matter of speed only:
     // tiles are handed out by ticket, so the look-back does not depend on which workgroups are resident)
    int ncu = 256; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, b->device);
    const int hold = std::max(1, mor_split_blocks_per_cu()) * ncu;
    d.sp_g = std::max(2, std::min(std::max(2, 32 * 6 / MOR_SP_NW), hold / b->B));   // (at most 24 per stream — round 6: 16 left a third of the GPU's slots empty at B = 32, + 1.3 % on the million-point clouds)
    if (getenv("MOR_SP_G")) d.sp_g = std::max(2, std::min(64, atoi(getenv("MOR_SP_G"))));   // (test knob: tests/test_gpu_parity.py runs 64 per stream, four times what the GPU holds)
    if (b->B * 2 > hold) d.two_pass_split = 1;
  }
  d.xcd_map = 1;   // the workgroups of a stream share an XCD (streams spread over all XCDs: −24 %)
  d.prop_map = getenv("MOR_PROP_MAP") ? atoi(getenv("MOR_PROP_MAP")) != 0 : 1; d.label_prefill = 0;
  // workgroups per stream (launch widths; the kernels share them out over the streams in proportion to the streams' work, map_block_work): tier 1 of the scores
  // (1024 threads each, two per CU), tiers 1a + 1b together (512 threads; they share out the chunks of the two worklists), wave tier (256 threads: a wave per
  // deferred query; 24 × B workgroups hold about one query per wave slot of the GPU — with 256 × B, one workgroup per four queries of the fullest stream, the
  // launch was mostly workgroups that found nothing to do: 175 k against 182 k frame-pairs/s), cell boxes
  d.g_fast = 6; d.g_score = 4; d.g_pde = 24; d.g_box = 32;
  d.use_hash = d.method == 1;
  { size_t hc = 1024; while (hc < 4 * (size_t)d.Nmax) hc <<= 1; d.Hcell = (int)hc; }
  d.gnz = nullptr; d.gnz_out = nullptr; d.vnz = nullptr; d.vnz_out = nullptr; d.cg_nz = d.g.nz; d.cg_inv_cs = d.g.inv_cs;
  d.t1_budget = 64;   // points a thread of the worklist tiers looks at before it hands its query to the wave tier
  // test switches (defaults: the fast paths): MOR_GH_TIER=1|2 starts k_gridhash with its big LDS table / its global-memory table; MOR_CG_GLOBAL forces the global-memory forests
  d.gh_tier = getenv("MOR_GH_TIER") ? atoi(getenv("MOR_GH_TIER")) : -1; d.cg_force_global = getenv("MOR_CG_GLOBAL") ? 1 : 0; d.cg_slow_tail = getenv("MOR_CG_SLOW_TAIL") ? 1 : 0; d.g2_nobet = getenv("MOR_G2_NOBET") ? 1 : 0; d.g2_exact_only = getenv("MOR_G2_EXACT") ? 1 : 0;
  d.P = 1;
  if (d.score_R > 60) return set_error(MOR_ERR_INVALID, "pde_ub %g needs a %d-cell search radius (> 60)", (double)p.pde_ub, d.score_R);
  int bits = 1; while ((1 << bits) < d.Kcap) ++bits;
  d.radix_passes = (bits + 7) / 8;
  d.Hcap = 64; while (d.Hcap < 2 * d.Nmax) d.Hcap <<= 1;
  return MOR_OK;
}

// Waits for everything enqueued on the batch (no error reporting: read-backs use this, so that a capacity or HIP error of an
// earlier frame of an asynchronous run stays in the sticky words until mor_batch_wait / the next synchronous push or filter reports it).
static int flush_track(mor_batch *b) {   // the held-back tracking step of the latest push, as a launch of its own on that frame's lane
  if (!b->track_held) return MOR_OK;
  hipStream_t lane = b->lane_stream(b->frame - 1);
  if (b->last_track && b->last_track_st != lane) HIP_TRY(hipStreamWaitEvent(lane, *b->last_track, 0));   // tracking state: after the previous frame's filterCloud loop
  mor_launch_piece(b->d, 6, lane, &b->timer);
  b->track_held = false;
  return MOR_OK;
}
static int sync_all(mor_batch *b) {
  if (!b->pending) return MOR_OK;
  { const int rcf = flush_track(b); if (rcf != MOR_OK) return rcf; }
  HIP_TRY(hipStreamSynchronize(b->sf)); HIP_TRY(hipStreamSynchronize(b->sc)); HIP_TRY(hipStreamSynchronize(b->sm)); HIP_TRY(hipStreamSynchronize(b->sb));
  for (auto &x : b->extra) if (x) HIP_TRY(hipStreamSynchronize(x));
  HIP_TRY(hipStreamSynchronize(b->s_h2d_[0])); HIP_TRY(hipStreamSynchronize(b->s_d2h_[0]));
  b->pending = false;
  b->timer.collect();
  return MOR_OK;
}
// Turns the sticky per-stream error words (every flag any kernel has raised since the last report, whichever frame it belonged
// to; mirrored to pinned memory by the last kernel of every push and filter) into an error code; reporting clears them.
static int report_errors(mor_batch *b) {
  int rc = MOR_OK;
  const MorDev &d = b->d;
  bool any = false;
  for (int s = 0; s < d.B; ++s) {
    const unsigned f = d.h_err[s];
    if (!f) continue;
    any = true;
    if (f & 1u) rc = set_error(MOR_ERR_CAPACITY, "stream %d: more than %d clusters", s, d.Kcap);
    if (f & 2u) rc = set_error(MOR_ERR_CAPACITY, "stream %d: voxel key out of range in method 2", s);
    if (f & 8u) rc = set_error(MOR_ERR_CAPACITY, "stream %d: z extent of the trimmed cloud exceeds the 64 m the voxel ground variant covers", s);
    if (f & 16u) rc = set_error(MOR_ERR_CAPACITY, "stream %d: more than 16384 points within gp_leaf of a voxel centroid", s);
    if (f & 64u) rc = set_error(MOR_ERR_HIP, "stream %d: look-back of the single-pass split stalled", s);
    if (f & 32u) rc = set_error(MOR_ERR_CAPACITY, "stream %d: more than %d tracked moving centroids", s, MOR_TR_MAXT);
    d.h_err[s] = 0;
  }
  if (any) HIP_TRY(hipMemset(d.err, 0, sizeof(unsigned) * d.B));
  return rc;
}
static int wait_all_checked(mor_batch *b) {
  const int rc = sync_all(b);
  return rc != MOR_OK ? rc : report_errors(b);
}

extern "C" {

#ifndef MOR_SRC_HASH_STR
#define MOR_SRC_HASH_STR "MOR_SRC_HASH=unknown"
#endif
// hash of the sources + compiler flags this library was built from (dynamicslamtool_amd/build.py embeds it; the loader compares it with the tree)
const char *mor_build_hash(void) { return MOR_SRC_HASH_STR; }
size_t mor_sizeof_params(void) { return sizeof(mor_params); }
const char *mor_last_error(void) { return g_last_error.c_str(); }
int mor_device_count(void) { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }

// NUMA node of a HIP device: its PCI address → /sys/bus/pci/devices/<address>/numa_node
int mor_device_numa_node(int device) {
  char id[64] = {0};
  if (hipDeviceGetPCIBusId(id, (int)sizeof id - 1, device) != hipSuccess) return -1;
  for (char *c = id; *c; ++c) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
  char path[160]; snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", id);
  FILE *f = fopen(path, "r"); if (!f) return -1;
  int node = -1; if (fscanf(f, "%d", &node) != 1) node = -1;
  fclose(f);
  return node;
}
int mor_bind_thread_to_device_node(int device, int share_index, int share_count) {
  const int node = mor_device_numa_node(device);
  if (node < 0) return 0;
  char path[96]; snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
  FILE *f = fopen(path, "r"); if (!f) return 0;
  char buf[4096] = {0}; const size_t got = fread(buf, 1, sizeof buf - 1, f); fclose(f);
  if (!got) return 0;
  cpu_set_t allowed, want; CPU_ZERO(&want);
  if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return 0;
  std::vector<int> cpus;
  for (char *p = buf; *p;) {   // "0-63,128-191"
    char *e; long a = strtol(p, &e, 10); if (e == p) break; long b2 = a;
    if (*e == '-') { p = e + 1; b2 = strtol(p, &e, 10); }
    for (long c = a; c <= b2 && c < CPU_SETSIZE; ++c) if (CPU_ISSET((int)c, &allowed)) cpus.push_back((int)c);
    p = (*e == ',') ? e + 1 : e; if (*p == '\n') break;
  }
  if (cpus.empty()) return 0;
  size_t lo = 0, hi = cpus.size();
  if (share_count > 1 && share_index >= 0 && share_index < share_count && cpus.size() >= (size_t)share_count) { const size_t per = cpus.size() / share_count; lo = per * share_index; hi = lo + per; }
  for (size_t i = lo; i < hi; ++i) CPU_SET(cpus[i], &want);
  if (sched_setaffinity(0, sizeof want, &want) != 0) return 0;
  return (int)(hi - lo);
}

void mor_batch_destroy(mor_batch *b) {
  if (!b) return;
  hipSetDevice(b->device);
  if (b->st) hipStreamSynchronize(b->st);
  for (auto &x : b->s_h2d_) if (x) hipStreamSynchronize(x);
  for (auto &x : b->s_d2h_) if (x) hipStreamSynchronize(x);
  if (b->sf) hipStreamSynchronize(b->sf);
  if (b->sc) hipStreamSynchronize(b->sc);
  if (b->sm) hipStreamSynchronize(b->sm);
  if (b->sb) hipStreamSynchronize(b->sb);
  for (auto &x : b->extra) if (x) { hipStreamSynchronize(x); hipStreamDestroy(x); }
  for (auto &ev : b->outptr_ev) if (ev) hipEventDestroy(ev);
  for (auto *arr : {b->ev_clusters, b->ev_pairs, b->ev_tpush}) for (int i = 0; i < MOR_MAX_SLOTS; ++i) if (arr[i]) hipEventDestroy(arr[i]);
  for (auto &ev : b->ev_back) if (ev) hipEventDestroy(ev);
  for (auto &ev : b->ev_track) if (ev) hipEventDestroy(ev);
  for (auto &ev : b->ev_out) if (ev) hipEventDestroy(ev);
  for (auto &ev : b->ev_h2d) if (ev) hipEventDestroy(ev);
  for (auto &ev : b->ev_d2h) if (ev) hipEventDestroy(ev);
  if (b->sf) hipStreamDestroy(b->sf);
  if (b->sc) hipStreamDestroy(b->sc);
  if (b->sm) hipStreamDestroy(b->sm);
  if (b->sb) hipStreamDestroy(b->sb);
  for (void *p : b->dev_allocs) hipFree(p);
  for (void *p : b->host_allocs) hipHostFree(p);
  if (b->d_stage) hipFree(b->d_stage);
  if (b->d_outstage) hipFree(b->d_outstage);
  for (auto &e : b->ev) if (e) hipEventDestroy(e);
  delete b;
}

mor_batch *mor_batch_create(const mor_params *p, int n_bad, int n_good, int n_streams, uint64_t max_points, int device, int *err) {
  int rc = MOR_OK; mor_batch *b = nullptr;
  auto fail = [&](int code) -> mor_batch * { if (err) *err = code; if (b) mor_batch_destroy(b); return nullptr; };
  if (!p || n_streams < 1 || max_points < 1 || max_points > (1ull << 26)) return fail(set_error(MOR_ERR_INVALID, "bad arguments to mor_batch_create"));
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(set_error(MOR_ERR_HIP, "no HIP device available (this library has no CPU fallback)"));
  if (device < 0 || device >= ndev) return fail(set_error(MOR_ERR_INVALID, "device %d out of range (%d devices)", device, ndev));
  if (hipSetDevice(device) != hipSuccess) return fail(set_error(MOR_ERR_HIP, "hipSetDevice(%d) failed", device));
  b = new mor_batch(); memset(&b->d, 0, sizeof b->d);
  b->p = *p; b->n_bad = n_bad; b->n_good = n_good; b->B = n_streams; b->device = device; b->Nmax = max_points;
  if (getenv("MOR_PIPE_DEPTH")) b->pipe_depth = std::min(MOR_MAX_DEPTH, std::max(1, atoi(getenv("MOR_PIPE_DEPTH"))));
  b->n_slots = b->pipe_depth + 1;
  if (getenv("MOR_CG_P")) b->env_cg_p = atoi(getenv("MOR_CG_P"));
  if (getenv("MOR_FUSE_TRACK")) b->fuse_track = atoi(getenv("MOR_FUSE_TRACK")) != 0;
  if ((rc = configure(b)) != MOR_OK) return fail(rc);
  if (n_bad > MOR_TR_NB) return fail(set_error(MOR_ERR_INVALID, "n_bad = %d: windows longer than %d frames are not supported", n_bad, MOR_TR_NB));
  for (auto &ev : b->outptr_ev) if (hipEventCreate(&ev) != hipSuccess) return fail(set_error(MOR_ERR_HIP, "hipEventCreate failed"));
  {
    MorCopyStreams *cs = copy_streams(device);
    if (!cs) return fail(set_error(MOR_ERR_HIP, "hipStreamCreate failed"));
    b->st = cs->st; b->s_h2d_[0] = cs->h2d; b->s_d2h_[0] = cs->d2h;
  }
  for (auto &e : b->ev) if (hipEventCreate(&e) != hipSuccess) return fail(set_error(MOR_ERR_HIP, "hipEventCreate failed"));
  // (Tried and measured without effect on the pipeline: highest stream priority for the cell-graph stream, and CU masks
  //  that give it 32-96 CUs of its own.)
  if (hipStreamCreateWithFlags(&b->sf, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&b->sc, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&b->sm, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&b->sb, hipStreamNonBlocking) != hipSuccess) return fail(set_error(MOR_ERR_HIP, "hipStreamCreate failed"));
  for (auto *arr : {b->ev_clusters, b->ev_pairs, b->ev_tpush}) for (int i = 0; i < MOR_MAX_SLOTS; ++i) if (hipEventCreateWithFlags(&arr[i], hipEventDisableTiming) != hipSuccess) return fail(set_error(MOR_ERR_HIP, "hipEventCreate failed"));
  for (auto &ev : b->ev_back) if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return fail(set_error(MOR_ERR_HIP, "hipEventCreate failed"));
  for (auto &ev : b->ev_track) if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return fail(set_error(MOR_ERR_HIP, "hipEventCreate failed"));
  for (auto &ev : b->ev_out) if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return fail(set_error(MOR_ERR_HIP, "hipEventCreate failed"));
  for (auto &ev : b->ev_h2d) if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return fail(set_error(MOR_ERR_HIP, "hipEventCreate failed"));
  for (auto &ev : b->ev_d2h) if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return fail(set_error(MOR_ERR_HIP, "hipEventCreate failed"));

  if (p->ground_method == 0) { const int ids[8] = {7, 8, 1, 2, 3, 4, 5, 6}; b->n_pieces = 8; for (int i = 0; i < 8; ++i) b->piece_id[i] = ids[i]; }
  else { const int ids[12] = {10, 11, 12, 13, 14, 15, 1, 2, 3, 4, 5, 6}; b->n_pieces = 12; for (int i = 0; i < 12; ++i) b->piece_id[i] = ids[i]; }
  b->n_lanes = (int)std::min<uint64_t>(4, b->pipe_depth);
  if (getenv("MOR_LANES")) b->n_lanes = std::max(1, std::min<int>((int)std::min<uint64_t>(8, b->pipe_depth), atoi(getenv("MOR_LANES"))));
  for (int i = 4; i < b->n_lanes; ++i) if (hipStreamCreateWithFlags(&b->extra[i - 4], hipStreamNonBlocking) != hipSuccess) return fail(set_error(MOR_ERR_HIP, "hipStreamCreate failed"));
  MorDev &d = b->d; const size_t B = d.B, N = d.Nmax, K = d.Kcap, T = d.tiles_max;
  bool ok = true;
  {  // method-1 search stencil: (dy,dz) rows ordered by their distance lower bound, then by centre distance
    const int R = d.score_R, side = 2 * R + 1;
    std::vector<std::pair<std::pair<int, int>, std::pair<int, int>>> rows;
    for (int dz = -R; dz <= R; ++dz) for (int dy = -R; dy <= R; ++dy) {
      int ly = std::max(std::abs(dy) - 1, 0), lz = std::max(std::abs(dz) - 1, 0);
      rows.push_back({{ly * ly + lz * lz, dy * dy + dz * dz}, {dy, dz}});
    }
    std::sort(rows.begin(), rows.end());
    std::vector<signed char> tab(2 * (size_t)side * side);
    for (size_t i = 0; i < rows.size(); ++i) { tab[2 * i] = (signed char)rows[i].second.first; tab[2 * i + 1] = (signed char)rows[i].second.second; }
    signed char *dtab = nullptr;
    ok = ok && dalloc(b, dtab, tab.size());
    if (ok) ok = hipMemcpy(dtab, tab.data(), tab.size(), hipMemcpyHostToDevice) == hipSuccess;
    d.row_order = dtab; d.n_rows = side * side;
  }
  d.Wcap = (int)(N / MOR_CHUNK + K + 2);
  d.gc_chunks = (int)(N / MOR_GC_CHUNK + 1); d.gc_P = 1;
  d.rs16_stride = (int)(((size_t)std::max(d.g.nrows, d.gv.nrows) + 1 + 7) & ~(size_t)7);
  d.cls_rows = (int)((N + 63) / 64);
  d.cx16_stride = (int)((N + 2) & ~(size_t)1);   // even, and one entry beyond the last cell (the scoring tiers copy the table two entries at a time)
  d.moving_confidence = n_bad; d.static_confidence = n_good; d.leave_off = p->leave_off_distance; d.catch_up = p->catch_up_distance;
  // ---- shared by all frames: the cluster arrays (frame-slotted: cb, ca and the frames in flight behind them), the tracking state (strictly
  //      serial across frames), the sticky error words, the pinned host mirrors (written by the serial tracking / output steps)
  for (int i = 0; i < (int)b->n_slots; ++i)
    ok = ok && dalloc(b, d.cl_pts[i], B * N) && dalloc(b, d.cl_cid[i], B * N) && dalloc(b, d.cl_off[i], B * (K + 1)) && dalloc(b, d.chunk_off[i], B * (K + 1)) && dalloc(b, d.centroid[i], B * K) && dalloc(b, d.amin[i], B * K) && dalloc(b, d.amax[i], B * K) && dalloc(b, d.cl_first[i], B * K) && dalloc(b, d.slot_kc[i], B) && hipMemset(d.slot_kc[i], 0, B * sizeof(int2)) == hipSuccess;
  ok = ok && dalloc(b, d.err, B) && hipMemset(d.err, 0, B * sizeof(unsigned)) == hipSuccess && halloc(b, d.h_err, B) && halloc(b, d.h_log, B * (size_t)MOR_LOG_CAP) && dalloc(b, b->d_outptrs, B * MOR_MAX_DEPTH);
  ok = ok && dalloc(b, d.tr, B) && hipMemset(d.tr, 0, B * sizeof(MorTrackDev)) == hipSuccess && dalloc(b, d.tr_corr, B * MOR_TR_NB * K) && dalloc(b, d.tr_res, B * (MOR_TR_NB + 1) * K) && dalloc(b, d.tr_lastdet, B * K) && dalloc(b, d.tr_match, B * (size_t)(MOR_TR_MAXT + 1)) && hipMemset(d.tr_match, 0, B * (size_t)(MOR_TR_MAXT + 1) * sizeof(int)) == hipSuccess;
  ok = ok && halloc(b, b->h_args_ring, B * MOR_ARGS_RING) && halloc(b, b->h_outptrs, B * MOR_ARGS_RING);
  b->h_args = b->h_args_ring;
  ok = ok && halloc(b, d.h_info, B) && halloc(b, d.h_centroid, B * K) && halloc(b, d.h_cl_off, B * (K + 1)) && halloc(b, d.h_det, B * K);
  ok = ok && halloc(b, d.h_pair_q, B * K) && halloc(b, d.h_pair_m, B * K) && halloc(b, d.h_pair_d, B * K) && halloc(b, d.h_score, B * K) && halloc(b, d.h_nout, B) && halloc(b, d.h_noff, B);
  if (!ok) return fail(set_error(MOR_ERR_HIP, "device/host allocation failed (B=%d, max_points=%llu)", d.B, (unsigned long long)max_points));
  // ---- once per frame in flight (frame k uses copy k mod depth): EVERY array a push or a filter writes outside the shared set above,
  //      scratch included — so two frames never share a buffer and the only orderings between their launches are the true dependencies
  //      (mor_push_batch).  288 GB of HBM3E make that cheap: ≈ 4 GB per copy at B = 64 × 120 000 points.
  const size_t R1 = (size_t)std::max(d.g.nrows, d.gv.nrows) + 1;
  std::vector<float> z0(B, p->gp_limit);   // crop-box variant: the clustering grid starts at gp_limit for every stream
  ok = dalloc(b, d.gh_hint, B) && hipMemset(d.gh_hint, 0, B * sizeof(int)) == hipSuccess;   // (one for all copies: a stream's cell count of the latest grid build, the next build's tier hint)
  ok = ok && dalloc(b, d.g2_pred, B) && hipMemsetD32((hipDeviceptr_t)d.g2_pred, 0x7fffffff, B) == hipSuccess;   // (one for all copies: the latest mode bin of the voxel ground variant — the next frames' bet)
  if (!ok) return fail(set_error(MOR_ERR_HIP, "device allocation failed (B=%d)", d.B));
  for (int c = 0; c < (int)b->pipe_depth; ++c) {
    MorDev o = d; MorStreamArgs *dargs = nullptr;
    ok = dalloc(b, dargs, B) && dalloc(b, o.info, B) && hipMemset(o.info, 0, B * sizeof(MorFrameInfo)) == hipSuccess && dalloc(b, o.tickets, B * 8) && hipMemset(o.tickets, 0, B * 8 * sizeof(int)) == hipSuccess;
    ok = ok && dalloc(b, o.tile_cnt, B * T * 2) && dalloc(b, o.split_desc, B * (T * (8 / MOR_SP_ROWS) * 4 / MOR_SP_NW + 1)) && hipMemset(o.split_desc, 0, B * (T * (8 / MOR_SP_ROWS) * 4 / MOR_SP_NW + 1) * sizeof(unsigned long long)) == hipSuccess;   // (frame tags of the single-read split start at 1)
    ok = ok && dalloc(b, o.cloud, B * N) && dalloc(b, o.ground, 2 * B * N) && dalloc(b, o.cls_mask, B * (size_t)d.cls_rows * 2) && dalloc(b, o.pkey, B * N) && dalloc(b, o.pslot, B * N) && dalloc(b, o.gc_list, B * N) && dalloc(b, o.gc_ent, B * N) && dalloc(b, o.gc_n, B * (size_t)d.gc_chunks) && dalloc(b, o.gc_tab, B * (size_t)16384) && dalloc(b, o.gc_tabsel, B);
    ok = ok && dalloc(b, o.gh_rowlist, B * N) && dalloc(b, o.gh_cells, B * N) && dalloc(b, o.gh_rowfill, B * R1) && dalloc(b, o.gh_key, B * (size_t)d.Hcell) && dalloc(b, o.gh_val, B * (size_t)d.Hcell);
    ok = ok && dalloc(b, o.ckey, B * N + 8) && dalloc(b, o.cstart, B * (N + 1)) && dalloc(b, o.row_start, B * R1) && dalloc(b, o.cmin, B * N) && dalloc(b, o.cmeta, 2 * B * N) && dalloc(b, o.crep, B * N) && dalloc(b, o.sorted, B * N) && dalloc(b, o.scell, B * N) && dalloc(b, o.csum, B * N);
    ok = ok && dalloc(b, o.slab_y, B * (MOR_MAXP + 1)) && dalloc(b, o.slab_c, B * (MOR_MAXP + 1)) && dalloc(b, o.slab_e, B * (MOR_MAXP + 1)) && dalloc(b, o.slab_p, B) && dalloc(b, o.lroot_a, B * N) && dalloc(b, o.lroot_b, B * N) && dalloc(b, o.parent, B * N) && dalloc(b, o.parent2, B * N) && dalloc(b, o.cg_ovf, B * (size_t)MOR_MAXP * MOR_CGS_OVF * 2);
    ok = ok && dalloc(b, o.croot, B * N) && dalloc(b, o.csize, B * N) && dalloc(b, o.compmin, B * N) && dalloc(b, o.cid_of_root, B * N) && dalloc(b, o.pcid, B * N) && dalloc(b, o.ccid, B * N) && dalloc(b, o.cgat, B * N) && dalloc(b, o.clist, B * N) && dalloc(b, o.cl_coff, B * (K + 1));
    ok = ok && dalloc(b, o.ktile_cnt, B * T) && dalloc(b, o.kcell, B * K) && dalloc(b, o.kroot, B * K) && dalloc(b, o.ksize, B * K) && dalloc(b, o.csz, B * K) && dalloc(b, o.krank_inv, B * K);
    ok = ok && dalloc(b, o.xcent, B * K) && dalloc(b, o.xamin, B * K) && dalloc(b, o.xamax, B * K) && dalloc(b, o.xfirst, B * K) && dalloc(b, o.part_back, B * (size_t)d.Wcap);
    ok = ok && dalloc(b, o.nn_fwd, B * K) && dalloc(b, o.nn_bwd, B * K) && dalloc(b, o.nn_fwd_d, B * K) && dalloc(b, o.pair_q, B * K) && dalloc(b, o.pair_m, B * K) && dalloc(b, o.pair_d, B * K);
    ok = ok && dalloc(b, o.pair_cnt, B * K) && dalloc(b, o.pair_of_prev, B * K) && dalloc(b, o.qrec, B * K * 2) && dalloc(b, o.pair_of_cur, B * K) && dalloc(b, o.det, B * K);
    ok = ok && dalloc(b, o.wl, B * N) && dalloc(b, o.wl_nb, B) && dalloc(b, o.wl2, B * N) && dalloc(b, o.wl2_n, B);
    ok = ok && dalloc(b, o.rs16, B * (size_t)d.rs16_stride) && dalloc(b, o.cx16, B * (size_t)d.cx16_stride);
    if (p->method_choice == 2) ok = ok && dalloc(b, o.vox, B * (size_t)d.Hcap) && dalloc(b, *reinterpret_cast<unsigned char **>(&o.vrec), B * 2 * K * 32);
    ok = ok && dalloc(b, o.moving, B * (K / 32 + 2)) && hipMemset(o.moving, 0, B * (K / 32 + 2) * sizeof(unsigned)) == hipSuccess && dalloc(b, o.out_desc, B * T) && hipMemset(o.out_desc, 0, B * T * sizeof(unsigned long long)) == hipSuccess && halloc(b, o.h_nout, B) && halloc(b, o.h_noff, B);   // (the size mirrors of the filtered clouds too: the output kernels of consecutive frames need no order among themselves)
    ok = ok && dalloc(b, o.zmin_i, B * MOR_ZR) && dalloc(b, o.zmax_i, B * MOR_ZR) && hipMemsetD32((hipDeviceptr_t)o.zmin_i, 0x7fffffff, B * MOR_ZR) == hipSuccess && hipMemsetD32((hipDeviceptr_t)o.zmax_i, (int)0x80000000, B * MOR_ZR) == hipSuccess && dalloc(b, o.zorg, B) && dalloc(b, o.zbase, B) && hipMemset(o.zbase, 0, B * sizeof(int)) == hipSuccess && dalloc(b, o.mode_bin, B) && dalloc(b, o.g2_used, B) && dalloc(b, o.g2_tag, B) && hipMemset(o.g2_tag, 0, B * sizeof(int)) == hipSuccess && hipMemsetD32((hipDeviceptr_t)o.g2_used, 0x7fffffff, B) == hipSuccess;
    if (ok) ok = hipMemcpy(o.zorg, z0.data(), B * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
    if (d.gmode == 1) {   // voxel-covariance ground variant: the VoxelGrid sort and the per-voxel results
      for (int i = 0; i < 2; ++i) ok = ok && dalloc(b, o.rkeys[i], B * N) && dalloc(b, o.rvals[i], B * N);
      ok = ok && dalloc(b, o.rhist, B * T * 256) && dalloc(b, o.gnz, B) && dalloc(b, o.vnz, B) && dalloc(b, o.rawbuf, B * N) && dalloc(b, o.is_ground, B * N) && dalloc(b, o.vcent, B * N) && dalloc(b, o.vbin, B * N) && dalloc(b, o.g2_big, B * N) && dalloc(b, o.g2_nbig, B) && hipMemset(o.g2_nbig, 0, B * sizeof(int)) == hipSuccess && dalloc(b, o.g2_open, B * (size_t)256) && dalloc(b, o.g2_nopen, 2) && hipMemset(o.g2_nopen, 0, 2 * sizeof(int)) == hipSuccess && hipMemset(o.is_ground, 0, B * N * sizeof(int)) == hipSuccess;
      o.skey = o.rkeys[d.voxel_passes & 1]; o.sidx = o.rvals[d.voxel_passes & 1];
      o.g2_opencap = (int)(B * 256);
      {   // occupancy bits of the lattice: 64 + 32 bytes per (y,z) row and 512 cells in x — 990 MB per frame in flight at B = 64, ±50 m, 0.2-m leaves, of which a frame touches the rows of its own z layers (25 MB)
        const size_t words = (size_t)B * (size_t)d.gv.nrows * (size_t)d.g2_nch * 8;
        if (!getenv("MOR_G2_NOBITS") && words * 12 <= ((size_t)2 << 30)) ok = ok && dalloc(b, o.g2_bits, words) && dalloc(b, o.g2_dir, words);   // (k_g2_cent writes every row of the stream's layers, every frame: no clearing)
      }
    }
    if (!ok) return fail(set_error(MOR_ERR_HIP, "device allocation failed (B=%d, max_points=%llu, copy %d of %d)", d.B, (unsigned long long)max_points, c + 1, (int)b->pipe_depth));
    b->d_args_s[c] = dargs; o.args = dargs;
    b->dtemp[c] = o;
  }
  b->d = b->dtemp[0];
  b->prev_pose.resize(B); b->last_n.assign(B, 0);
  if (err) *err = MOR_OK;
  return b;
}

int mor_batch_streams(const mor_batch *b) { return b ? b->B : 0; }

int mor_push_batch(mor_batch *b, const mor_cloud_view *clouds, const double *poses) {
  if (!b || !clouds || !poses) return set_error(MOR_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(b->device));
  { const int rcf = flush_track(b); if (rcf != MOR_OK) return rcf; }   // (a push behind a push: the earlier frame's tracking step goes out alone)
  const uint64_t k = b->frame;
  MorDev d = b->dtemp[k % b->pipe_depth]; const int B = d.B;
  uint64_t maxn = 0; size_t max_host_bytes = 0;
  for (int s = 0; s < B; ++s) {
    const mor_cloud_view &c = clouds[s];
    if (c.n_points > b->Nmax) return set_error(MOR_ERR_CAPACITY, "stream %d: %llu points > max_points %llu", s, (unsigned long long)c.n_points, (unsigned long long)b->Nmax);
    if (c.n_points && (!c.data || c.point_step < 12 || c.off_x + 4 > c.point_step || c.off_y + 4 > c.point_step || c.off_z + 4 > c.point_step ||
                       (c.off_intensity != MOR_NO_FIELD && c.off_intensity + 4 > c.point_step)))
      return set_error(MOR_ERR_INVALID, "stream %d: bad blob layout", s);
    maxn = std::max<uint64_t>(maxn, c.n_points);
    if (!c.on_device) max_host_bytes = std::max<size_t>(max_host_bytes, (size_t)c.n_points * c.point_step);
  }
  if (max_host_bytes > b->stage_stride) {   // (re)allocate the staging area for host-resident blobs
    if (b->d_stage) { b->pending = true; int rcw = wait_all_checked(b); if (rcw != MOR_OK) return rcw; HIP_TRY(hipFree(b->d_stage)); b->d_stage = nullptr; }
    b->stage_stride = (max_host_bytes + 255) & ~(size_t)255;
    HIP_TRY(hipMalloc((void **)&b->d_stage, b->stage_stride * B * b->pipe_depth));
  }
  // The frame before this one was pushed but not filtered (push, push): its tracking step and its end get their events now, on its lane
  // — a filterCloud records them itself, and a push that is followed by its filterCloud (the usual case) needs neither
  if (k > 0 && !b->filtered) {
    const uint64_t kq = (k - 1) % MOR_MAX_SLOTS; hipStream_t pl = b->lane_stream(k - 1);
    HIP_TRY(hipEventRecord(b->ev_tpush[kq], pl)); b->last_track = &b->ev_tpush[kq]; b->last_track_st = pl;
    HIP_TRY(hipEventRecord(b->ev_back[kq], pl)); b->ev_back_st[kq] = pl;
  }
  // pinned argument slot of this push: a ring of MOR_ARGS_RING slots; the slot's last user, frame k − MOR_ARGS_RING, must have completed
  const int slot = (int)(k % MOR_ARGS_RING);
  if (k >= MOR_ARGS_RING) HIP_TRY(hipEventSynchronize(b->ev_back[(k - MOR_ARGS_RING) % MOR_MAX_SLOTS]));   // (never recorded for a frame whose lane orders it anyway, below: then the tracking event of that frame, which lies behind its k_split — the slot's reader — on its lane)
  if (k >= MOR_ARGS_RING) HIP_TRY(hipEventSynchronize(b->ev_track[(k - MOR_ARGS_RING) % MOR_MAX_SLOTS]));
  b->h_args = b->h_args_ring + (size_t)slot * B;
  std::vector<PoseTf> cur(B);
  for (int s = 0; s < B; ++s) {
    const mor_cloud_view &c = clouds[s]; MorStreamArgs &a = b->h_args[s];
    a.n = (uint32_t)c.n_points; a.step = c.point_step; a.off_x = c.off_x; a.off_y = c.off_y; a.off_z = c.off_z; a.off_i = c.off_intensity;
    a.data = (c.on_device || c.n_points == 0) ? c.data : (const void *)(b->d_stage + b->stage_stride * ((k % b->pipe_depth) * B + s));
    b->last_n[s] = c.n_points;
    pose_to_tf(poses + 7 * s, cur[s]);
    if (k > 0) relative_transform(cur[s], b->prev_pose[s], a.xf); else memset(a.xf, 0, sizeof a.xf);
    b->prev_pose[s] = cur[s];
  }
  d.tiles = std::max<int>(1, (int)((maxn + MOR_TILE - 1) / MOR_TILE));
  d.split_g = d.tiles;
  {  // cell graph: slabs per stream (k_cg_slab) — enough that a slab's cells fit its LDS with room for imbalance, and that
     // the launch fills the GPU (two 512-thread workgroups per CU)
    uint32_t maxocc = 0, maxloc = 0;
    for (int s = 0; s < B; ++s) { maxocc = std::max(maxocc, k > 0 ? d.h_info[s].n_occ : 0u); maxloc = std::max(maxloc, k > 0 ? d.h_info[s].max_loc : 0u); }
    (void)maxloc;
    const int p_fit = (int)((maxocc * 3ull / 2 + 1087) / 1088), p_par = (512 + B - 1) / B;   // a slab's LDS holds 1360 cells (own + look-ahead; CGS_CAP); two 512-thread workgroups per CU
    d.P = b->env_cg_p > 0 ? b->env_cg_p : std::max(p_fit, p_par);
    d.P = std::max(1, std::min(d.P, std::min(MOR_MAXP, std::max(1, d.g.ny / 2))));
    {  // slabs in proportion to the streams' cell counts: own cells per slab such that the slabs of all streams together are the launch's B·P workgroups
       // (Σ ceil(n_occ / T) ≤ Σ n_occ / T + B); never more than a slab's LDS holds with its look-ahead.  Counts of the latest frame the device reported.
      unsigned long long tot = 0; for (int s = 0; s < B; ++s) tot += k > 0 ? d.h_info[s].n_occ : 0u;
      d.slab_T = (k > 0 && tot > 0 && d.P > 1) ? (int)std::min<unsigned long long>(800, std::max<unsigned long long>(32, (tot + (unsigned long long)B * (d.P - 1) - 1) / ((unsigned long long)B * (d.P - 1)))) : 0;
    }
    {  // grid build, cell pass, output: workgroups per stream of k_gridcount / k_gridplace / k_cellboxes / k_out.  The kernels share the launch out over the
       // streams by their point counts (map_block_work), so the width follows the MEAN cloud the device last reported (+ 15 %, + 1), not the largest:
       // a launch sized for the largest stream was mostly workgroups that found nothing to do (32 per stream for a mean of 5 chunks).
      uint64_t sumM = 0; uint32_t mxM = 0; for (int s = 0; s < B; ++s) { const uint32_t m = k > 0 ? d.h_info[s].M : (uint32_t)maxn; sumM += m; mxM = std::max(mxM, m); }
      const uint64_t meanM = (sumM + B - 1) / B, ref = d.prop_map ? meanM * 23 / 20 : (uint64_t)mxM * 5 / 4;
      { uint64_t sumC = 0; for (int s = 0; s < B; ++s) sumC += k > 0 ? d.h_info[s].C : 0u; d.label_prefill = k > 0 && 2 * sumC < sumM; }
      if (const char *lp = getenv("MOR_LABEL_PREFILL")) d.label_prefill = atoi(lp) != 0;   // test knob: the path is otherwise chosen per push from host mirrors the device updates asynchronously — which one a frame takes depends on timing; 0 / 1 hold each to the oracle   // (either way the labels are the same: a matter of where the −1 of an unclustered point is written)
      const int want = (int)((ref + MOR_GC_CHUNK - 1) / MOR_GC_CHUNK) + (d.prop_map ? 1 : 0);
      d.gc_P = std::max(1, std::min(std::min(want, d.gc_chunks), std::max(1, 2048 / B)));
      d.g_out = std::max(1, std::min(want, d.tiles));
      d.g_box = std::max(2, std::min(32, (int)((ref + 1023) / 1024) + 1));   // (a workgroup of the cell pass takes 1024 positions per round)
    }
    d.cg_fused = (maxocc * 11ull / 10 <= MOR_CGS_FCAP && !getenv("MOR_CG_UNFUSED")) ? 1 : 0;   // (a stream beyond it runs the merge on global-memory arrays: correct, slow — hence the separate kernel when that is foreseeable)
    // (table tier of k_gridhash: −1 = every stream by its own cell count of the latest build; MOR_GH_TIER forces the tier all streams start with)
  }
  d.cur = (int)(k % b->n_slots); d.prev = (int)((k + b->n_slots - 1) % b->n_slots); d.has_prev = k > 0; d.out_ptrs = nullptr; d.out_step32 = 0; d.frame_no = (int)k;
  {  // workgroups for the cloud-sized kernels: 1.25 × the largest cloud / cluster set the device last reported
    uint32_t mx = 0;
    for (int s = 0; s < B; ++s) mx = std::max(mx, std::max(d.h_info[s].M, d.h_info[s].C));
    d.tiles_m = (k > 0 && mx > 0) ? std::min<int>(d.tiles, (int)(((uint64_t)mx * 5 / 4 + MOR_TILE - 1) / MOR_TILE) + 1) : d.tiles;
  }
  // ---- the launches of the push, piece by piece, on the stage streams (frames overlap as a software pipeline: piece p of
  //      frame k runs beside later pieces of frames k−1, k−2).  The first piece must not overwrite what frame k−depth still
  //      uses (same buffer copy; its cluster slot doubles as the `ca` slot of frame k−depth+1)
  hipStream_t lane = b->lane_stream(k);
  const uint64_t depth = b->pipe_depth, ks = k % MOR_MAX_SLOTS, kp = (k + MOR_MAX_SLOTS - 1) % MOR_MAX_SLOTS;
  if (k >= depth && b->ev_back_st[(k - depth) % MOR_MAX_SLOTS] != lane) HIP_TRY(hipStreamWaitEvent(lane, b->ev_back[(k - depth) % MOR_MAX_SLOTS], 0));   // the frame that used this copy of the per-frame arrays (and this cluster slot as its ca) is done (with as many lanes as copies it ran on this very lane)
  // (the staging area of this frame's copy was last read by the split of frame k − depth: covered by the wait above)
  if (max_host_bytes > 0) {   // host-resident blobs are staged through device memory, on the host → device copy stream
    hipStream_t cs = b->s_h2d_[0];
    if (k >= depth) HIP_TRY(hipStreamWaitEvent(cs, b->ev_back[(k - depth) % MOR_MAX_SLOTS], 0));   // (this frame's staging area was read by the split of frame k − depth)
    for (int s = 0; s < B;) {   // runs of blobs that lie back to back in host memory and in the staging area travel as one copy
      const mor_cloud_view &c = clouds[s];
      if (c.on_device || !c.n_points) { ++s; continue; }
      const char *src = (const char *)c.data; char *dst = (char *)b->h_args[s].data; size_t bytes = (size_t)c.n_points * c.point_step; int e = s + 1;
      while (e < B && !clouds[e].on_device && clouds[e].n_points && (const char *)clouds[e].data == src + bytes && (char *)b->h_args[e].data == dst + bytes) { bytes += (size_t)clouds[e].n_points * clouds[e].point_step; ++e; }
      HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, cs));
      s = e;
    }
    HIP_TRY(hipEventRecord(b->ev_h2d[ks], cs));
    HIP_TRY(hipStreamWaitEvent(lane, b->ev_h2d[ks], 0));
  }
  // the per-stream arguments: a one-workgroup kernel reads the page-locked slot (5 KB) — a hipMemcpyAsync here kept the lane idle for ≈ 35 µs
  // in front of every frame's first kernel (copy packet, its signal, the barrier behind it)
  // (crop variant: the frame's first kernel, k_split, reads the slot itself and leaves the device copy behind — nothing at all in front of it)
  d.args_src = nullptr; d.args_out = b->d_args_s[k % b->pipe_depth];
  if (!d.two_pass_split && (d.gmode == 0 || !d.g2_passa2)) d.args_src = b->h_args;   // (… and pass A of the voxel ground variant as the single-read split)
  else mor_launch_copy(b->d_args_s[k % b->pipe_depth], b->h_args, sizeof(MorStreamArgs) * B, lane);
  if (!b->async) HIP_TRY(hipEventRecord(b->ev[0], lane));
  for (int pc = 0; pc < b->n_pieces; ++pc) {
    const int id = b->piece_id[pc];
    const bool trk = pc == b->n_pieces - 1;
    if (id == 3 && k > 0) {   // the cluster piece also transforms frame k − 1's clusters in place (:540-551) and pairs them with this frame's
      HIP_TRY(hipStreamWaitEvent(lane, b->ev_clusters[kp], 0));
      if (d.method == 2) HIP_TRY(hipStreamWaitEvent(lane, b->ev_pairs[kp], 0));   // … which frame k − 1's own voxel probe (method 2) must have finished reading
    }
    if (trk && b->async && b->fuse_track) { b->track_held = true; continue; }   // held back: launched with the loop of the filterCloud that follows (mor_filter_batch), or alone by flush_track
    if (trk && b->last_track && b->last_track_st != lane) HIP_TRY(hipStreamWaitEvent(lane, *b->last_track, 0));   // tracking state: after the previous frame's filterCloud loop
    mor_launch_piece(d, id, lane, &b->timer);
    if (id == 3) HIP_TRY(hipEventRecord(b->ev_clusters[ks], lane));
    if (id == 4 && d.method == 2) HIP_TRY(hipEventRecord(b->ev_pairs[ks], lane));
  }
  if (!b->async) HIP_TRY(hipEventRecord(b->ev[1], lane));
  // (no event for the tracking step or the end of the frame here: the frame's filterCloud follows on this lane and records both;
  //  a push without one gets them at the start of the next push)
  HIP_TRY(hipGetLastError());
  b->d = d; b->frame++; b->filtered = false; b->pending = true;
  if (b->async) return MOR_OK;
  int rc = wait_all_checked(b);
  hipEventElapsedTime(&b->push_ms, b->ev[0], b->ev[1]);
  return rc;
}

int mor_filter_batch(mor_batch *b, void *const *out, int out_on_device, uint64_t *n_out) { return mor_filter_batch_ex(b, out, out_on_device, n_out, 16); }
int mor_filter_batch_ex(mor_batch *b, void *const *out, int out_on_device, uint64_t *n_out, uint32_t out_point_step) {
  if (!b) return set_error(MOR_ERR_INVALID, "null batch");
  if (b->frame == 0) return set_error(MOR_ERR_NOT_READY, "filterCloud before the first pushRawCloudAndPose");
  if (out_point_step != 16 && out_point_step != 32) return set_error(MOR_ERR_INVALID, "out_point_step must be 16 (packed xyzi) or 32 (PointXYZI records)");
  if (out_point_step == 32 && !(out && out_on_device)) return set_error(MOR_ERR_INVALID, "32-byte records are written by the device itself: pass device-accessible pointers (device memory, or page-locked / registered host memory) with out_on_device = 1");
  HIP_TRY(hipSetDevice(b->device));
  MorDev d = b->d; const int B = d.B;
  const uint64_t k = b->frame - 1;
  // (the tracking loop of filterCloud (:630-671) runs on the device — on EVERY call, as in the reference: a second filterCloud on the same
  //  frame walks mo_vec again and moves the confidences again)
  d.filter_epoch = (unsigned)(++b->n_filter_calls); if (d.filter_epoch == 0) d.filter_epoch = (unsigned)(++b->n_filter_calls);
  b->filtered = true;
  d.out_ptrs = nullptr; d.out_step32 = out_point_step == 32 ? 1 : 0;
  hipStream_t fs = b->lane_stream(k);   // behind the frame's push
  // (the tracking step in front of this one — the frame's push, or an earlier filterCloud of the same frame — ran on this very stream)
  // Asynchronous mode with HOST output pointers: the output kernels assemble every stream's filtered cloud in a device staging area of
  // this frame and DMA copies of the stream's input size (an upper bound of the output: the caller's buffers hold n_in points, as the
  // synchronous form requires) carry them out behind the kernels — nothing waits; the sizes are read after the wait (mor_get_output_device).
  const bool host_async = out && !out_on_device && b->async && !n_out;
  if (host_async && !b->d_outstage) HIP_TRY(hipMalloc((void **)&b->d_outstage, sizeof(float4) * (size_t)d.Nmax * B * b->pipe_depth));
  if ((out && out_on_device) || host_async) {
    const int oslot = (int)(b->n_filters++ % MOR_ARGS_RING);   // pinned pointer table of this call (a ring: an earlier table may still be in flight)
    HIP_TRY(hipEventSynchronize(b->outptr_ev[oslot]));
    float4 **hp = b->h_outptrs + (size_t)oslot * B, **dp = b->d_outptrs + (size_t)(k % b->pipe_depth) * B;
    for (int s = 0; s < B; ++s) hp[s] = host_async ? b->d_outstage + ((size_t)(k % b->pipe_depth) * B + s) * d.Nmax : (float4 *)out[s];
    mor_launch_copy(dp, hp, sizeof(float4 *) * B, fs);
    HIP_TRY(hipEventRecord(b->outptr_ev[oslot], fs));
    d.out_ptrs = dp;
  }
  if (!b->async) HIP_TRY(hipEventRecord(b->ev[2], fs));
  if (host_async && k >= b->pipe_depth && b->d2h_used[(k - b->pipe_depth) % MOR_MAX_SLOTS]) {   // the output staging area of this frame's copy has been carried out
    HIP_TRY(hipStreamWaitEvent(fs, b->ev_d2h[(k - b->pipe_depth) % MOR_MAX_SLOTS], 0));
  }
  if (host_async && b->d2h_used[k % MOR_MAX_SLOTS]) HIP_TRY(hipStreamWaitEvent(fs, b->ev_d2h[k % MOR_MAX_SLOTS], 0));   // a second filterCloud on this very frame rewrites its staging area: after the first call's copies
  if (b->track_held) {   // the frame's own tracking step is still to come: one launch for it and this call's loop
    if (b->last_track && b->last_track_st != fs) HIP_TRY(hipStreamWaitEvent(fs, *b->last_track, 0));   // tracking state: after the previous frame's filterCloud loop
    mor_launch_filter(d, fs, &b->timer, 3);
    b->track_held = false;
  } else mor_launch_filter(d, fs, &b->timer, 1);
  HIP_TRY(hipEventRecord(b->ev_track[k % MOR_MAX_SLOTS], fs)); b->last_track = &b->ev_track[k % MOR_MAX_SLOTS]; b->last_track_st = fs;   // the tracking state is settled: the next frame's tracking step may follow
  mor_launch_filter(d, fs, &b->timer, 2);
  if (host_async) HIP_TRY(hipEventRecord(b->ev_out[k % MOR_MAX_SLOTS], fs));   // (the device → host copies below follow it)
  if (host_async) {   // on the device → host copy stream, behind the output kernels
    hipStream_t cs = b->s_d2h_[0];
    HIP_TRY(hipStreamWaitEvent(cs, b->ev_out[k % MOR_MAX_SLOTS], 0));
    for (int s = 0; s < B;) {   // (back-to-back buffers: one copy)
      if (!out[s] || !b->last_n[s]) { ++s; continue; }
      char *dst = (char *)out[s]; const char *src = (const char *)(b->d_outstage + ((size_t)(k % b->pipe_depth) * B + s) * d.Nmax); size_t bytes = b->last_n[s] * sizeof(float4); int e = s + 1;
      while (e < B && out[e] && b->last_n[e] && (char *)out[e] == dst + bytes && (const char *)(b->d_outstage + ((size_t)(k % b->pipe_depth) * B + e) * d.Nmax) == src + bytes) { bytes += b->last_n[e] * sizeof(float4); ++e; }
      HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, cs));
      s = e;
    }
    HIP_TRY(hipEventRecord(b->ev_d2h[k % MOR_MAX_SLOTS], cs));
    b->d2h_used[k % MOR_MAX_SLOTS] = true;
  }
  if (!b->async) HIP_TRY(hipEventRecord(b->ev[3], fs));
  // end of the frame: frame k + depth reuses this frame's copy of the arrays.  With as many lanes as copies that frame runs on this very
  // stream and needs no event; the copy stream of host-resident clouds (its staging area) does
  if (b->lane_stream(k + b->pipe_depth) != fs || b->stage_stride > 0) { HIP_TRY(hipEventRecord(b->ev_back[k % MOR_MAX_SLOTS], fs)); b->ev_back_st[k % MOR_MAX_SLOTS] = fs; }
  else b->ev_back_st[k % MOR_MAX_SLOTS] = fs;
  HIP_TRY(hipGetLastError());
  b->pending = true;
  const bool need_host = n_out != nullptr || (out && !out_on_device && !host_async);
  if (b->async && !need_host) return MOR_OK;
  int rc = wait_all_checked(b);
  hipEventElapsedTime(&b->filter_ms, b->ev[2], b->ev[3]);
  for (int s = 0; s < B; ++s) if (n_out) n_out[s] = d.h_nout[s];
  if (out && !out_on_device) {
    for (int s = 0; s < B; ++s) if (out[s] && d.h_nout[s]) HIP_TRY(hipMemcpyAsync(out[s], d.ground + 2 * (size_t)s * d.Nmax + d.h_noff[s], d.h_nout[s] * sizeof(float4), hipMemcpyDeviceToHost, b->st));
    HIP_TRY(hipStreamSynchronize(b->st));
  }
  return rc;
}

int mor_batch_set_async(mor_batch *b, int on) { if (!b) return MOR_ERR_INVALID; int rc = wait_all_checked(b); b->async = on != 0; return rc; }
int mor_batch_wait(mor_batch *b) { if (!b) return MOR_ERR_INVALID; HIP_TRY(hipSetDevice(b->device)); return wait_all_checked(b); }

const void *mor_get_output_device(const mor_batch *b, int s, uint64_t *n_out) {
  if (!b || s < 0 || s >= b->B) return nullptr;
  if (n_out) *n_out = b->d.h_nout[s];
  return b->d.ground + 2 * (size_t)s * b->d.Nmax + b->d.h_noff[s];   // assembled in place in the frame's ground buffer (one copy per frame in flight: valid for MOR_PIPE_DEPTH − 1 = 3 further pushes by default)
}

// ---- single-stream forms
mor_ctx *mor_create(const mor_params *p, int n_bad, int n_good, uint64_t max_points, int device, int *err) { return mor_batch_create(p, n_bad, n_good, 1, max_points, device, err); }
int mor_push(mor_ctx *c, const void *data, uint64_t n, uint32_t step, uint32_t ox, uint32_t oy, uint32_t oz, uint32_t oi, const double pose[7]) {
  mor_cloud_view v; v.data = data; v.n_points = n; v.point_step = step; v.off_x = ox; v.off_y = oy; v.off_z = oz; v.off_intensity = oi; v.on_device = 0;
  return mor_push_batch(c, &v, pose);
}
int mor_filter(mor_ctx *c, float *out, uint64_t *n_out) { void *o = out; return mor_filter_batch(c, out ? &o : nullptr, 0, n_out); }
void mor_destroy(mor_ctx *c) { mor_batch_destroy(c); }

// ---- read-backs
#define CHECK_STREAM()                                                                     \
  if (!b || s < 0 || s >= b->B) return set_error(MOR_ERR_INVALID, "bad batch/stream");      \
  if (b->frame == 0) return set_error(MOR_ERR_NOT_READY, "no frame pushed yet");            \
  HIP_TRY(hipSetDevice(b->device));                                                         \
  if (b->pending) { const int rc_ = sync_all(const_cast<mor_batch *>(b)); if (rc_ != MOR_OK) return rc_; }   /* (sticky error words are left for the next wait) */ \
  const MorDev &d = b->d; const MorFrameInfo &f = d.h_info[s]; const size_t so = (size_t)s * d.Nmax, ko = (size_t)s * d.Kcap; (void)so; (void)ko; (void)f

int mor_get_counts(const mor_batch *b, int s, mor_counts *o) {
  CHECK_STREAM();
  o->n_in = f.N; o->n_trim = f.T; o->n_cloud = f.M; o->n_ground = f.G; o->n_clusters = f.K; o->n_clustered = f.C;
  o->n_corr = f.n_pairs;
  { int n_mo = 0; HIP_TRY(hipMemcpy(&n_mo, &d.tr[s].n_mo, sizeof(int), hipMemcpyDeviceToHost)); o->n_tracks = (uint32_t)n_mo; }
  return MOR_OK;
}
// the split's class masks (two 64-bit words per 64 input records: cloud, ground) → for every cloud point / ground point, in order, its index in the trimmed cloud
static int trimmed_indices(const mor_batch *b, int s, std::vector<int> *cloud_ti, std::vector<int> *ground_ti) {
  const MorDev &d = b->d; const MorFrameInfo &f = d.h_info[s];
  const uint32_t n_in = d.gmode ? f.T : f.N;   // (voxel ground variant: pass B splits the trimmed cloud)
  const size_t rows = ((size_t)n_in + 63) / 64;
  std::vector<unsigned long long> m(rows * 2);
  if (rows) HIP_TRY(hipMemcpy(m.data(), d.cls_mask + (size_t)s * d.cls_rows * 2, rows * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  if (cloud_ti) cloud_ti->clear();
  if (ground_ti) ground_ti->clear();
  int t = 0;
  for (size_t r = 0; r < rows; ++r) {
    const unsigned long long ng = m[2 * r], g = m[2 * r + 1];
    for (unsigned long long any = ng | g; any; any &= any - 1) {
      const int l = __builtin_ctzll(any);
      if ((ng >> l) & 1ull) { if (cloud_ti) cloud_ti->push_back(t); } else if (ground_ti) ground_ti->push_back(t);
      ++t;
    }
  }
  return MOR_OK;
}
int mor_get_labels(const mor_batch *b, int s, int32_t *lab) {
  CHECK_STREAM();
  std::vector<int> pc(f.M), ti;
  if (f.M) HIP_TRY(hipMemcpy(pc.data(), d.pcid + so, f.M * sizeof(int), hipMemcpyDeviceToHost));
  const int rc = trimmed_indices(b, s, &ti, nullptr);
  if (rc != MOR_OK) return rc;
  if (ti.size() != f.M) return set_error(MOR_ERR_HIP, "stream %d: class masks name %zu cloud points, the frame has %u", s, ti.size(), f.M);
  for (uint32_t i = 0; i < f.T; ++i) lab[i] = -2;
  for (uint32_t i = 0; i < f.M; ++i) lab[ti[i]] = pc[i];
  return MOR_OK;
}
int mor_get_ground_indices(const mor_batch *b, int s, int32_t *idx) {
  CHECK_STREAM();
  std::vector<int> gi;
  const int rc = trimmed_indices(b, s, nullptr, &gi);
  if (rc != MOR_OK) return rc;
  if (gi.size() != f.G) return set_error(MOR_ERR_HIP, "stream %d: class masks name %zu ground points, the frame has %u", s, gi.size(), f.G);
  if (f.G) memcpy(idx, gi.data(), f.G * sizeof(int));
  return MOR_OK;
}
// cluster_indices (:218) in the reference's order — cluster after cluster, ascending cloud index inside a cluster —
// rebuilt from the labels (the device keeps the points of a cluster cell by cell)
static int cluster_lists(const mor_batch *b, int s, std::vector<int> &off, std::vector<int> &idx) {
  CHECK_STREAM();
  off.assign(d.h_cl_off + (size_t)s * (d.Kcap + 1), d.h_cl_off + (size_t)s * (d.Kcap + 1) + f.K + 1);
  idx.assign(f.C, 0);
  std::vector<int> pc(f.M), fill(off.begin(), off.end());
  if (f.M) HIP_TRY(hipMemcpy(pc.data(), d.pcid + so, f.M * sizeof(int), hipMemcpyDeviceToHost));
  for (uint32_t i = 0; i < f.M; ++i) if (pc[i] >= 0) idx[fill[pc[i]]++] = (int)i;
  return MOR_OK;
}
int mor_get_clusters(const mor_batch *b, int s, int32_t *off, int32_t *idx) {
  std::vector<int> o, ix;
  const int rc = cluster_lists(b, s, o, ix);
  if (rc != MOR_OK) return rc;
  memcpy(off, o.data(), o.size() * sizeof(int)); if (!ix.empty()) memcpy(idx, ix.data(), ix.size() * sizeof(int));
  return MOR_OK;
}
int mor_get_centroids(const mor_batch *b, int s, float *xyz) {
  CHECK_STREAM();
  for (uint32_t k = 0; k < f.K; ++k) { xyz[3 * k] = d.h_centroid[ko + k].x; xyz[3 * k + 1] = d.h_centroid[ko + k].y; xyz[3 * k + 2] = d.h_centroid[ko + k].z; }
  return MOR_OK;
}
int mor_get_detection(const mor_batch *b, int s, uint8_t *det) { CHECK_STREAM(); memcpy(det, d.h_det + ko, f.K); return MOR_OK; }
int mor_get_correspondences(const mor_batch *b, int s, int32_t *q, int32_t *m, float *dist, double *score) {
  CHECK_STREAM();
  memcpy(q, d.h_pair_q + ko, f.n_pairs * sizeof(int)); memcpy(m, d.h_pair_m + ko, f.n_pairs * sizeof(int));
  memcpy(dist, d.h_pair_d + ko, f.n_pairs * sizeof(float)); memcpy(score, d.h_score + ko, f.n_pairs * sizeof(double));
  return MOR_OK;
}
int mor_get_tracks(const mor_batch *b, int s, float *xyz, int32_t *conf, int32_t *maxc) {
  CHECK_STREAM();
  std::vector<MorTrackDev> t(1);
  HIP_TRY(hipMemcpy(t.data(), &d.tr[s], sizeof(MorTrackDev), hipMemcpyDeviceToHost));
  for (int i = 0; i < t[0].n_mo; ++i) { if (xyz) memcpy(xyz + 3 * i, t[0].mo_c[i], 3 * sizeof(float)); if (conf) conf[i] = t[0].mo_conf[i]; if (maxc) maxc[i] = t[0].mo_max[i]; }
  return MOR_OK;
}
// the clusters the latest filterCloud's loop over mo_vec matched its tracked centroids to, in loop order (:630-642)
int mor_get_moving_clusters(const mor_batch *b, int s, int32_t *cluster_of_track, uint32_t *n) {
  CHECK_STREAM();
  if (!n) return set_error(MOR_ERR_INVALID, "null argument");
  int cnt = 0;
  HIP_TRY(hipMemcpy(&cnt, d.tr_match + (size_t)s * (MOR_TR_MAXT + 1), sizeof(int), hipMemcpyDeviceToHost));
  if (!b->filtered) cnt = 0;   // no filterCloud on the latest frame yet
  *n = (uint32_t)cnt;
  if (cluster_of_track && cnt > 0) HIP_TRY(hipMemcpy(cluster_of_track, d.tr_match + (size_t)s * (MOR_TR_MAXT + 1) + 1, sizeof(int) * (size_t)cnt, hipMemcpyDeviceToHost));
  return MOR_OK;
}
// cluster_collection (:229): the clustered points (x, y, z, intensity) in the reference's order
static int cluster_points(const mor_batch *b, int s, std::vector<int> &off, std::vector<float4> &pts) {
  std::vector<int> idx;
  const int rc = cluster_lists(b, s, off, idx);
  if (rc != MOR_OK) return rc;
  const MorDev &d = b->d; const uint32_t M = d.h_info[s].M;
  std::vector<float4> cloud(M);
  if (M) HIP_TRY(hipMemcpy(cloud.data(), d.cloud + (size_t)s * d.Nmax, M * sizeof(float4), hipMemcpyDeviceToHost));
  pts.resize(idx.size());
  for (size_t j = 0; j < idx.size(); ++j) pts[j] = cloud[idx[j]];
  return MOR_OK;
}
int mor_get_cluster_collection(const mor_batch *b, int s, float *out) {
  std::vector<int> off; std::vector<float4> pts;
  const int rc = cluster_points(b, s, off, pts);
  if (rc != MOR_OK) return rc;
  if (!pts.empty()) memcpy(out, pts.data(), pts.size() * sizeof(float4));
  return MOR_OK;
}

// mark_cluster (:7-58) for every cluster of the latest frame: position = compute3DCentroid into an Eigen::Vector4f, i.e. a
// sequential FLOAT sum over the cluster's points in order divided by n (:15) — not the fp64 centroid of :239-243 —,
// scale = extent of getMinMax3D (:16, :36-38) with zero extents replaced by 0.1 (:40-47).  A debug read-back (the
// reference builds markers only under VISUALIZE, for tracked clusters): the points come to the host and are summed there
// in the reference's order.
int mor_get_markers(const mor_batch *b, int s, float *pos_K3, float *scale_K3) {
  std::vector<int> off; std::vector<float4> pts;
  const int rc = cluster_points(b, s, off, pts);
  if (rc != MOR_OK) return rc;
  for (size_t k = 0; k + 1 < off.size(); ++k) {
    float sx = 0.f, sy = 0.f, sz = 0.f, mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    const int n = off[k + 1] - off[k];
    for (int j = off[k]; j < off[k + 1]; ++j) {
      const float q[3] = {pts[j].x, pts[j].y, pts[j].z};
      sx += q[0]; sy += q[1]; sz += q[2];
      for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], q[a]); mx[a] = std::max(mx[a], q[a]); }
    }
    pos_K3[3 * k] = sx / (float)n; pos_K3[3 * k + 1] = sy / (float)n; pos_K3[3 * k + 2] = sz / (float)n;
    for (int a = 0; a < 3; ++a) { const float e = mx[a] - mn[a]; scale_K3[3 * k + a] = e == 0.f ? 0.1f : e; }
  }
  return MOR_OK;
}

int mor_get_boxes(const mor_batch *b, int s, float *min_K3, float *max_K3) {
  CHECK_STREAM();
  if (!f.K) return MOR_OK;
  std::vector<float4> lo(f.K), hi(f.K);
  HIP_TRY(hipMemcpy(lo.data(), d.amin[d.cur] + ko, f.K * sizeof(float4), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(hi.data(), d.amax[d.cur] + ko, f.K * sizeof(float4), hipMemcpyDeviceToHost));
  for (uint32_t k = 0; k < f.K; ++k) {
    min_K3[3 * k] = lo[k].x; min_K3[3 * k + 1] = lo[k].y; min_K3[3 * k + 2] = lo[k].z;
    max_K3[3 * k] = hi[k].x; max_K3[3 * k + 1] = hi[k].y; max_K3[3 * k + 2] = hi[k].z;
  }
  return MOR_OK;
}

int mor_get_frame_log(const mor_batch *b, uint64_t frame, int s, int64_t *out) {
  if (!b || s < 0 || s >= b->B || !out) return set_error(MOR_ERR_INVALID, "bad batch/stream");
  HIP_TRY(hipSetDevice(b->device));
  if (b->pending) { const int rc_ = sync_all(const_cast<mor_batch *>(b)); if (rc_ != MOR_OK) return rc_; }
  if (frame >= b->frame || frame + MOR_LOG_CAP < b->frame) return set_error(MOR_ERR_NOT_READY, "frame %llu is not in the log (frames pushed: %llu, log depth %d)", (unsigned long long)frame, (unsigned long long)b->frame, MOR_LOG_CAP);
  const MorFrameLog &L = b->d.h_log[(size_t)(frame % MOR_LOG_CAP) * b->B + s];
  out[0] = L.frame; out[1] = L.K; out[2] = L.C; out[3] = L.n_pairs; out[4] = L.cnt_sum; out[5] = L.det_sum; out[6] = L.n_mo_push; out[7] = L.n_mo_filter; out[8] = (int64_t)L.n_out; out[9] = L.flags;
  return MOR_OK;
}

// Development read-back of a named intermediate device array of stream s (latest frame): copies min(bytes, size of the stream's
// slice) bytes and returns the number copied, or a negative error.  Not part of the product interface (exp/, tests).
long long mor_debug_read(const mor_batch *b, const char *name, int s, void *out, size_t bytes) {
  if (!b || s < 0 || s >= b->B || !name || !out) return set_error(MOR_ERR_INVALID, "bad arguments");
  HIP_TRY(hipSetDevice(b->device));
  if (b->pending) { const int rc_ = sync_all(const_cast<mor_batch *>(b)); if (rc_ != MOR_OK) return rc_; }
  const MorDev &d = b->d; const size_t N = d.Nmax, K = d.Kcap, S = MOR_MAXP + 1;
  struct Ent { const char *n; const void *p; size_t stride; };
  const Ent tab[] = {
      {"ckey", d.ckey, N * 4}, {"cstart", d.cstart, (N + 1) * 4}, {"row_start", d.row_start, ((size_t)d.g.nrows + 1) * 4}, 
      {"pkey", d.pkey, N * 4}, {"sorted", d.sorted, N * 16}, {"cloud", d.cloud, N * 16}, {"cmin", d.cmin, N * 4}, {"cmeta", d.cmeta, 2 * N * 16}, {"crep", d.crep, N * 16},
      {"slab_y", d.slab_y, S * 4}, {"slab_c", d.slab_c, S * 4}, {"slab_e", d.slab_e, S * 4}, {"lroot_a", d.lroot_a, N * 4}, {"lroot_b", d.lroot_b, N * 4},
      {"ccid", d.ccid, N * 4}, {"pcid", d.pcid, N * 4}, {"info", d.info, sizeof(MorFrameInfo)},
      {"xcent", d.xcent, K * 16}, {"xamin", d.xamin, K * 16}, {"xamax", d.xamax, K * 16}, {"xfirst", d.xfirst, K * 16}, {"g2_big", d.g2_big, N * 4}, {"g2_nbig", d.g2_nbig, 4}, {"vbin", d.vbin, N * 4}, {"scell", d.scell, N * 4}, {"cgat", d.cgat, N * 16}, {"csum", d.csum, N * 48}, {"clist", d.clist, N * 4}, {"cl_pts_prev", d.cl_pts[d.prev], N * 16}, {"cl_pts", d.cl_pts[d.cur], N * 16}};
  for (const Ent &e : tab) if (!strcmp(e.n, name)) {
    if (!e.p) return set_error(MOR_ERR_INVALID, "array %s is not allocated in this configuration", name);
    const size_t n = std::min(bytes, e.stride);
    HIP_TRY(hipMemcpy(out, (const char *)e.p + (size_t)s * e.stride, n, hipMemcpyDeviceToHost));
    return (long long)n;
  }
  return set_error(MOR_ERR_INVALID, "unknown array %s", name);
}
int mor_debug_config(const mor_batch *b, int *out, int n) {   // grid geometry and launch configuration of the latest push
  if (!b || !out) return MOR_ERR_INVALID;
  const MorDev &d = b->d;
  const int v[12] = {d.g.nx, d.g.ny, d.g.nz, d.g.nrows, d.P, 1, 1, d.Hcell, d.Kcap, d.tiles_m, d.cur, d.prev};
  for (int i = 0; i < n && i < 12; ++i) out[i] = v[i];
  return MOR_OK;
}

int mor_get_stage_counts(const mor_batch *b, int s, uint32_t *out, int n) {
  CHECK_STREAM();
  const uint32_t v[6] = {f.n_occ, f.n_defer, f.pad0, f.Cprev, f.g2_exact, f.max_loc};   // v[2] = queries left after tier 1; v[4] = voxels whose ordered sums were evaluated (voxel ground variant)
  for (int i = 0; i < n && i < 6; ++i) out[i] = v[i];   // v[5] = cells (own + look-ahead) of the stream's largest slab of the cell graph
  return MOR_OK;
}

// ---- device memory helpers
void *mor_device_alloc(int device, size_t bytes) { void *p = nullptr; if (hipSetDevice(device) != hipSuccess || hipMalloc(&p, bytes) != hipSuccess) { set_error(MOR_ERR_HIP, "hipMalloc(%zu) failed", bytes); return nullptr; } return p; }
void mor_device_free(int device, void *p) { if (hipSetDevice(device) == hipSuccess) hipFree(p); }
void *mor_host_alloc(size_t bytes) { void *p = nullptr; if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { set_error(MOR_ERR_HIP, "hipHostMalloc(%zu) failed", bytes); return nullptr; } return p; }
void mor_host_free(void *p) { if (p) hipHostFree(p); }
// page-locks caller-owned host memory (a std::vector's buffer, say) and maps it for the device: *device_ptr is what kernels and out_on_device = 1 pointers use
int mor_host_register(void *p, size_t bytes, void **device_ptr) {
  if (!p || !bytes || !device_ptr) return set_error(MOR_ERR_INVALID, "null argument");
  HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterMapped));
  void *dp = nullptr;
  if (hipHostGetDevicePointer(&dp, p, 0) != hipSuccess || !dp) { hipHostUnregister(p); return set_error(MOR_ERR_HIP, "hipHostGetDevicePointer failed for registered memory"); }
  *device_ptr = dp;
  return MOR_OK;
}
int mor_host_unregister(void *p) { if (!p) return MOR_OK; HIP_TRY(hipHostUnregister(p)); return MOR_OK; }
int mor_device_upload(int device, void *dst, const void *src, size_t bytes) { HIP_TRY(hipSetDevice(device)); HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return MOR_OK; }
int mor_device_download(int device, void *dst, const void *src, size_t bytes) { HIP_TRY(hipSetDevice(device)); HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return MOR_OK; }
int mor_device_synchronize(int device) { HIP_TRY(hipSetDevice(device)); HIP_TRY(hipDeviceSynchronize()); return MOR_OK; }

// ---- timing
int mor_get_last_timing(const mor_batch *b, float *push_ms, float *filter_ms) {
  if (!b) return MOR_ERR_INVALID;
  if (push_ms) *push_ms = b->push_ms;
  if (filter_ms) *filter_ms = b->filter_ms;
  return MOR_OK;
}
int mor_kernel_timing_enable(mor_batch *b, int enable) { if (!b) return MOR_ERR_INVALID; b->timer.enabled = enable != 0; return MOR_OK; }
// the kernels of the last timed leg as (kernel id, start ms, end ms) triples on one clock: shows how the stage streams overlap
int mor_kernel_timeline_read(mor_batch *b, int *ids, float *t0_ms, float *t1_ms, int max_n) {
  if (!b) return MOR_ERR_INVALID;
  const int n = std::min<int>((int)b->timer.timeline.size(), max_n);
  for (int i = 0; i < n; ++i) { ids[i] = b->timer.timeline[i].id; t0_ms[i] = b->timer.timeline[i].t0; t1_ms[i] = b->timer.timeline[i].t1; }
  return n;
}
int mor_kernel_timing_read(mor_batch *b, int reset, char *names, size_t names_cap, float *ms_total, uint32_t *launches, int max_kernels) {
  if (!b) return MOR_ERR_INVALID;
  std::string all;
  int n = std::min<int>(MK_COUNT, max_kernels);
  for (int i = 0; i < n; ++i) { if (ms_total) ms_total[i] = (float)b->timer.ms[i]; if (launches) launches[i] = b->timer.launches[i]; all += mor_kernel_names[i]; all += (i + 1 < n) ? "," : ""; }
  if (names && names_cap) { strncpy(names, all.c_str(), names_cap - 1); names[names_cap - 1] = 0; }
  if (reset) { memset(b->timer.ms, 0, sizeof b->timer.ms); memset(b->timer.launches, 0, sizeof b->timer.launches); }
  return n;
}

}   // extern "C"
