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
#include <hip/hip_fp16.h>

const char *const mor_kernel_names[MK_COUNT] = {   // in MorKernelId order: "k_" + name = the __global__ function
    "classify", "scatter", "split", "heads_count", "heads_scatter", "cellboxes", "rhist", "rscan", "rscatter",
    "score_fast", "score_nb", "score_pde", "vox_clear", "vox_insert", "vox_probe",
    "out", "g2_cov_mid", "g2_cov", "g2_cov_big", "g2_mode", "g2_mark", "track_push", "track_filter", "gridhash", "gridcount", "gridplace", "cg_slab", "cg_final", "clusters", "g2_cent", "track_push_filter"};

// The kernels by stage (one translation unit; every file is #included exactly here):
#include "kernels_common.h"
#include "kernels_split.h"
#include "kernels_grid.h"
#include "kernels_cellgraph.h"
#include "kernels_radix.h"
#include "kernels_clusters.h"
#include "kernels_scores.h"
#include "kernels_ground_voxel.h"
#include "kernels_track_out.h"

// ------------------------------------------------------------------------------------ launch sequences
// Builds with -DMOR_EXPERIMENTS (exp/shadow.sh; never the product build) can launch ONE kernel of the pipeline twice (MOR_EXP_DUP=<kernel id>, idempotent kernels only): the
// period then grows by what that kernel costs the pipeline, which is not its duration (DESIGN.md §4 "Where the time goes")
#ifdef MOR_EXPERIMENTS
static int mor_exp_dup() { static const int v = getenv("MOR_EXP_DUP") ? atoi(getenv("MOR_EXP_DUP")) : -1; return v; }
#define MOR_EXP_SECOND_LAUNCH(id, kern, grid, threads, ...) if (mor_exp_dup() == (int)(id)) hipLaunchKernelGGL(kern, grid, dim3(threads), 0, st, __VA_ARGS__)
#else
#define MOR_EXP_SECOND_LAUNCH(id, kern, grid, threads, ...) (void)0
#endif
#define MOR_LAUNCH_T(id, kern, grid, threads, ...)                        \
  do {                                                                    \
    mor_timer_begin(tm, id, st);                                          \
    hipLaunchKernelGGL(kern, grid, dim3(threads), 0, st, __VA_ARGS__);    \
    MOR_EXP_SECOND_LAUNCH(id, kern, grid, threads, __VA_ARGS__);          \
    mor_timer_end(tm, id, st);                                            \
  } while (0)
#define MOR_LAUNCH(id, kern, grid, ...) MOR_LAUNCH_T(id, kern, grid, MOR_BT, __VA_ARGS__)

// part: 0 = both, 1 = the split only, 2 = the grid build only (the lane schedule runs them as two pieces)
static void mor_launch_split_and_grid(const MorDev &d, hipStream_t st, MorLaunchTimer *tm, int part = 0) {
  const dim3 gT(d.B * d.tiles), gM(d.B * d.tiles_m), gB(d.B);
  const bool single_read = !d.two_pass_split && !(d.gmode == 1 && d.g2_passa2);   // (pass A of the voxel ground variant: MOR_G2_PASSA2=1 keeps its count pass + scatter pass)
  if (part == 2) goto grid;
  if (single_read) {
    if (d.gmode == 2) MOR_LAUNCH_T(MK_SPLIT, k_split<2>, dim3(d.B * d.sp_g), 64 * MOR_SP_NW, d);   // (pass B of the voxel ground variant: the records carry a ground flag; three instances so that the crop variant does not keep registers for it)
    else if (d.gmode == 1) MOR_LAUNCH_T(MK_SPLIT, k_split<1>, dim3(d.B * d.sp_g), 64 * MOR_SP_NW, d);   // (pass A: x/y trim, z range, packed lattice coordinates — radix pass 0 makes the keys)
    else MOR_LAUNCH_T(MK_SPLIT, k_split<0>, dim3(d.B * d.sp_g), 64 * MOR_SP_NW, d);
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
      MorRadix j = {pass == 0 ? d.pkey : d.rkeys[pass & 1], pass == 0 ? nullptr : d.rvals[pass & 1], d.rkeys[(pass + 1) & 1], d.rvals[(pass + 1) & 1], 8 * pass, 0, 0, nullptr, d.rhist, 0, 1, 1, 1, (pass == 0 && single_read) ? 1 : 0};   // (a stream's last pass — by its own key width — leaves the inverse permutation: k_heads_scatter moves the points)
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
  da.gnz_out = d.gnz; da.gnz = d.vnz; da.vnz_out = d.vnz; da.cg_nz = d.g.nz; da.cg_inv_cs = d.g.inv_cs;   // (pass A's kernels behind the split see the lattice with the stream's own layers through stream_grid)
  da.scell = nullptr;   // (nobody reads the cell of a position of the voxel-ordered cloud: k_heads_scatter skips that store)
  da.skey = d.rkeys[da.cell_passes & 1]; da.sidx = d.rvals[da.cell_passes & 1];
  if (sub == 0) {
    // (z range, ground flags and the queue of big voxels need no clearing launches: their last readers of the previous frame on this copy
    //  leave them ready — publish_split of pass B, the frame tag in is_ground, k_g2_mode)
    mor_launch_split_and_grid(da, st, tm);
  } else if (sub == 1) {
    MOR_LAUNCH_T(MK_G2_CENT, k_g2_cent, dim3(G2_CENT_G * d.B), MOR_BT, da);
    MOR_LAUNCH_T(MK_G2_COV, k_g2_cov, dim3(G2_COV_G * d.B), MOR_BT, da);
  } else if (sub == 2) {
    MOR_LAUNCH(MK_G2_COV_MID, k_g2_cov_mid, dim3(64, d.B), da);
  } else if (sub == 3) {
    MOR_LAUNCH_T(MK_G2_COV_BIG, k_g2_cov_big, dim3(G2_BIG_WG), G2_BIG_T, da);   // (every workgroup is a whole CU's LDS: one per CU, entries of the batch-wide list by ticket)
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
  if (part == 3) { MOR_LAUNCH_T(MK_TRACK_PUSH_FILTER, k_track_push_filter, dim3(d.B), 64, d); return; }   // the tracking step the push held back + the loop of this filterCloud
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
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_split<0>, 64 * MOR_SP_NW, 0) != hipSuccess) n = 2;
  return n < 1 ? 1 : n;
}
