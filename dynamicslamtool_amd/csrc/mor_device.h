// mor_device.h — device-visible descriptors shared by mor_kernels.hip and mor_engine.cpp.
// Vocabulary follows the reference: streams (one MovingObjectRemoval instance each), frames ca/cb,
// `cloud` (non-ground points), `gp_indices`, clusters, centroids, correspondences.
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

#define MOR_TILE 2048   // points per workgroup tile: 4 waves × 8 coalesced 1-KiB rows of float4
#ifndef MOR_SP_ROWS
#define MOR_SP_ROWS 4    // rows of 64 records per wave and tile of the single-read split (k_split): its tiles are MOR_SP_NW waves × MOR_SP_ROWS × 64 records (8, 4 or 2 rows)
#endif
#define MOR_BT 256      // threads per workgroup
#ifndef MOR_SP_NW
#define MOR_SP_NW 8      // waves per workgroup of the single-read split (k_split): a tile is MOR_SP_NW × MOR_SP_ROWS × 64 records.  Round 6: 8 waves × 4 rows — 2048-record tiles at the 32 data registers of the
                        // 1024-record form (half the descriptors, barriers and tickets per byte): k_split alone 97 → 82 µs, hdl64_b64 +1.9 %, agg10_b32 +2.4 % interleaved; 1 / 2 waves (tiles of 256 / 512: the wave-granular form) −6 / −3 %, 16 waves −0.5 % against 8
#endif
#define MOR_CHUNK 2048  // points per work item of the per-cluster reductions
#define MOR_KGRID 128   // workgroups per stream for per-cluster kernels (grid-stride over clusters)

// Uniform grid over the trim box, keyed by linear cell id (a collision-free spatial hash).
// Cell edge s = 0.57·r, so the cell diagonal is < r and ALL points of one cell are mutually within
// the cluster tolerance: a cell is a clique, and Euclidean clustering reduces to connected
// components over occupied CELLS (two cells are adjacent iff some point pair across them has
// d² < r²; only cells ≤ 2 apart per axis can be).  Points are radix-sorted by cell key
// ((z·ny + y)·nx + x); the occupied cells are the distinct keys `ckey` (ascending), an occupied
// cell's compact id is its rank in `ckey`, and `row_start` (dense over the ny·nz rows) finds a
// row's cells with two loads — so an x-run of cells is one contiguous range of the sorted points.
struct MorGrid {
  float ox, oy, oz, inv_cs, cs;
  int nx, ny, nz, nrows, keybits;
  int mode, ibx, iby;   // mode 1 (VoxelGrid lattice): cell = floor(v·inv_cs) − ib, absolute multiples of the leaf; z base per stream
};

struct Red6 { double sx, sy, sz; float mnx, mny, mnz, mxx, mxy, mxz; };   // partial Σxyz (fp64) + AABB

// one LSD radix pass (8-bit digit) of the stable (key, value) sort
struct MorRadix {
  const int *kin, *vin;   // [B][Nmax]; vin == nullptr ⇒ value = element index
  int *kout, *vout;       // kout may be nullptr on a last pass that only needs the values
  int shift, count_sel;   // digit = (key >> shift) & 255; element count: 0 → M, 1 → C
  int drop_negative;      // elements with key < 0 are dropped (unclustered points)
  int *vout2;             // optional second copy of the values (pass 0 of the cluster partition also writes cl_idx)
  int *hist;              // [B][tiles_max][256] histogram / offset scratch of this sort
  int skip_k_le;          // > 0: streams with K ≤ this need no further pass — the kernel returns at once for them
  int fuse;               // the stream's last workgroup of k_rhist scans the histograms (no k_rscan launch)
  int vox;                // keys are voxel keys of the voxel ground variant: a stream takes part in the passes its own key width needs (voxel_passes_of)
  int inverse;            // last pass: vout[value] = position (the inverse permutation) instead of vout[position] = value — the points are then MOVED to their places
                          // by coalesced reads and fire-and-forget writes (k_heads_scatter) instead of gathered by 7 M dependent random 16-byte reads per step
  int unpack;             // pass 0 behind the single-read pass A of the voxel ground variant: kin holds packed lattice coordinates (voxel_pack); the pass sorts — and writes — the stream's linear keys
};

struct MorStreamArgs {       // per stream, per push (host → device, one small copy)
  const void *data;          // incoming blob (device pointer)
  uint32_t n, step, off_x, off_y, off_z, off_i;
  float xf[12];              // row-major 3×4 fp32 transform prev → cur (:536, cast as pcl_ros does)
};

struct MorFrameInfo {        // per stream, produced on device
  uint32_t N, T, M, G, K, C, n_pairs, flags;   // flags bit0: cluster capacity exceeded, bit1: voxel key overflow
  uint32_t Kprev, Cprev, n_keep, n_occ;   // n_occ: occupied grid cells
  uint32_t n_defer, pad0, g2_exact, max_loc;   // n_defer: method-1 queries handed to the wave tier; g2_exact: voxels of the voxel ground variant whose ordered sums had to be evaluated (the screen left them open); max_loc: cells (own + look-ahead) of the largest slab of the cell graph
};

// Temporal logic (T1 + the tracking loop of F1) as device state, one instance per stream.  O(clusters) sequential work
// per frame: it runs in a one-workgroup-per-stream kernel right behind the geometry so that push + filter need no
// host round trip in between.  Mirrors csrc/mor_tracker.cpp (the host version behind the mor_tracker_* C ABI).
#define MOR_MAX_DEPTH 8     // frames in flight in the stage pipeline, at most (one copy of every per-frame array each)
#define MOR_MAX_SLOTS 10     // cluster-array slots (depth + 1 are in use)
#define MOR_ZR 32            // ints between the z-range words of two streams: a 128-byte line each — every wave of k_classify hits them with atomics, and on dense [B] arrays 64 streams are two lines whose atomics go through the L2 one after the other.  (The same spacing for the worklist counters of the scoring tiers, the queue counter of the voxel tiers and the tile tickets — returning atomics of every wave / tile — measured nothing: 219.2 against 218.3 k, 44.1 against 44.5 k.)
#define MOR_MAXP 32        // slabs per stream of the cell graph (k_cg_slab), at most
#define MOR_CGS_FCAP 9400  // cells per stream the fused merge at the tail of k_cg_slab holds in LDS (beyond: its global-memory path; the host then prefers the separate k_cg_final)
#define MOR_GC_CHUNK 2048   // points per chunk of the grid build (k_gridcount / k_gridplace)
#define MOR_CGS_OVF 8192   // overflow entries per slab of its candidate-pair lists (beyond them a pair is settled on the spot)
#define MOR_TR_MAXT 32768  // tracked moving centroids per stream (mo_vec); the reference has no bound — beyond this one the push reports MOR_ERR_CAPACITY
#define MOR_TR_NB 8       // longest supported window (n_bad)
#define MOR_KCAP_MAX 16384 // clusters per stream, at most (Kcap = min(max_points / min_cluster_size + 1, this))
struct MorTrackDev {
  int n_mo, n_corr, n_res, has_cur, K_last, overflow, pad0, pad1;
  int corr_n[MOR_TR_NB], res_n[MOR_TR_NB + 1];
  float mo_c[MOR_TR_MAXT][3];
  int mo_conf[MOR_TR_MAXT], mo_max[MOR_TR_MAXT];
};

// per-frame, per-stream summary written into pinned host memory by the tail / filter kernels
#define MOR_LOG_CAP 64
struct MorFrameLog { int frame, K, C, n_pairs; unsigned cnt_sum, det_sum; int n_mo_push, n_mo_filter; unsigned long long n_out; int flags, pad; };

// per cluster of ca / cb (method 2): its pair and the origin of the pair's voxel lattice as ONE record — the voxel kernels go point → cluster id → record →
// table instead of point → cluster id → pair → previous cluster → its first point → table
struct MorVoxRec { double mn[3]; int pr, pad; };

struct MorDev {
  // ---- static configuration
  int B, s0, Btot;           // streams in this launch, first stream, streams in the batch
  int Nmax, Kcap, tiles_max, radix_passes, cell_passes, Hcap;
  float trim_x, trim_y, trim_z, gp_limit, r2;
  long long min_cs, max_cs;
  float pde_lb, pde_ub;
  double pde_thr, vol_thr, opc_res, opc_inv_res;   // opc_inv_res = 1 / opc_res (fp64): the voxel keys of method 2 by a multiplication where that cannot differ from the division
  int method, opc_norm, score_R, n_rows, t1_budget, vol_abs_int, opc_anchor_half;
  const signed char *row_order; // [n_rows][2] (dy,dz) of the method-1 search stencil, nearest rows first
  MorGrid g;                 // clustering grid (cell edge 0.57·r)
  MorGrid gv;                // VoxelGrid lattice of the voxel-covariance ground removal (cell edge gp_leaf)
  int gmode;                 // 0: crop-box ground removal; 1 / 2: passes A (trim) / B (split by ground flag) of the voxel variant
  int voxel_passes; float leaf_r2; double g2_r, g2_inv_r;   // g2_r = √leaf_r2 · 1.0001 + 1e-6: the radius as the screen of the voxel verdicts bounds coordinates with it; g2_inv_r = 1 / √leaf_r2 · 1.0001
  // ---- per call
  int tiles;                 // ceil(max n_points of this batch / MOR_TILE)
  int tiles_m;               // workgroups per stream for kernels over the non-ground cloud / clusters: an estimate from the
                             // previous frame; those kernels grid-stride over the tiles a stream really has, so any value ≥ 1 is correct
  int split_g;   // workgroups per stream of the two split passes (each walks tiles split_g apart)
  int g_out;                 // workgroups per stream of k_out's compaction (from the streams' mean tile count)
  int g_fast, g_score, g_pde, g_box;   // launch widths, workgroups per stream (shared out by work inside the launch): tier 1 of the scores, the worklist tiers, the wave tier, k_cellboxes
  int label_prefill;         // 1: k_gridplace leaves −1 in every cloud point's label (input order, coalesced) and k_clusters stores the labels of clustered points only (its label stores are scattered); the host sets it when fewer than half of the cloud points of the latest reported frame lay in clusters (the voxel ground variant's pass B: a seventh)
  int prop_map;              // 1 (default): launches share their workgroups out over the streams in proportion to the streams' work (map_block_work); 0: the same number for every stream (MOR_PROP_MAP=0)
  int xcd_map;               // 1: the workgroups of a stream share an XCD (its tables stay in that L2); 0: streams spread over all XCDs
  int sp_g;                  // workgroups per stream of the single-read split (2 … 64)
  int g2_passa2;             // voxel ground variant: pass A as count pass + scatter pass (MOR_G2_PASSA2=1; default: the single-read split, k_split<1>)
  int two_pass_split;        // 0 (default): the single-read split with decoupled look-back (k_split); 1: count pass + scatter pass (MOR_SINGLE_PASS_SPLIT=0, or a batch wider than half the workgroups the GPU holds)
  int cur, prev, has_prev;   // cluster-array slots of cb and ca (four slots rotate: up to three frames are in flight in the stage pipeline); whether ca exists (:534)
  // ---- device arrays (per-stream stride noted)
  const MorStreamArgs *args; // [B]
  const MorStreamArgs *args_src; MorStreamArgs *args_out;   // crop variant with the single-read split: the page-locked host slot k_split reads the arguments from (null: `args` is already filled), and `args` again, writable
  MorFrameInfo *info;        // [B]  this frame (one copy per frame in flight, like every array that crosses a stage boundary)
  int2 *slot_kc[MOR_MAX_SLOTS];          // [B]  (K, C) of the frame that owns the cluster slot: written by the cell graph of that frame, read by the
                             //      next frame's pair stage as ca's K and C (never through another frame's `info` copy, which the grid
                             //      stage of a later frame resets while the pair stage may still be running)
  int *tickets;              // [B][8]  arrival counters of the "last workgroup of the stream" hand-offs (one set per frame in flight)
  unsigned *err;             // [B]  sticky error word per stream: every raised flag is OR-ed in and stays until the host has reported it
  unsigned *h_err;           // [B]  pinned host mirror of `err`, refreshed by the last kernel of every push and filter
  int frame_no;              // index of this frame since the batch was created
  struct MorFrameLog *h_log; // [MOR_LOG_CAP][B]  pinned per-frame summaries (frame k in row k % MOR_LOG_CAP): lets tests compare every frame of an asynchronous run
  int *tile_cnt;             // [B][tiles_max][2]   (non-ground, ground) counts per tile
  unsigned long long *split_desc; // [B][tiles_max]  look-back descriptors of the single-pass split (status | non-ground | ground)
  float4 *cloud;             // [B][Nmax]  non-ground points, input order (`cloud`, :85)
  float4 *ground;            // [B][2·Nmax]  slots [Nmax, Nmax+G): removed points in order (raw_cloud[gp_indices], :683); slots [Nmax − n_keep, Nmax): the kept cloud points after filterCloud — together the filtered cloud, assembled in place
  unsigned long long *cls_mask; int cls_rows;   // [B][cls_rows][2]  per 64 input records of the split: which went to `cloud`, which to the ground — gp_indices (:86) and the cloud points' indices in the trimmed cloud are rebuilt from them on the host
  float4 *rawbuf;            // [B][Nmax]  trimmed cloud (voxel ground variant only)
  int *is_ground;            // [B][Nmax]  per trimmed point
  float4 *vcent; int *vbin;  // [B][Nmax]  voxel centroids (dsc, :113) and bin id of accepted voxels
  int *g2_big, *g2_nbig;     // [B][Nmax], [B]  voxels with more neighbours than the one-wave kernel holds
  int2 *g2_open; int *g2_nopen; int g2_opencap;   // [g2_opencap] (stream, voxel), [2] (entries, entries taken)  what k_g2_cov_mid leaves for k_g2_cov_big, ALL streams in one list: its workgroups — a whole CU's LDS each — take entries by ticket instead of walking their own stream's queue (a handful of open voxels per stream, dozens in a few streams)
  unsigned long long *g2_bits; int *g2_dir; int g2_nch;   // [B][gv.nrows][g2_nch·8]  occupancy bits of the VoxelGrid lattice, a row padded to g2_nch chunks of 512 cells, and the first cell of every word of them (row_cells_bits); null: the kernels search the keys (MOR_G2_NOBITS, or a lattice whose bits would not fit 2 GB per frame in flight)
  int *zmin_i, *zmax_i;      // [B·MOR_ZR]  ordered-int min / max z of the trimmed cloud, stream s at s·MOR_ZR
  float *zorg; int *zbase;   // [B]  z origin of the clustering grid / z base of the voxel lattice
  int *mode_bin;             // [B]  dominant z-bin (:169-178)
  int *g2_pred, *g2_used, *g2_tag;   // [B] the latest mode bin any frame of the stream has reported (ONE array for all copies of the per-frame state); [B] this frame's snapshot of it — the bin its kernels mark speculatively; [B] the tag of this frame's ground marks (k_g2_mode)
  int *pkey;                 // [B][Nmax]  linear cell key per cloud point
  int *pslot;                // [B][Nmax]  grid build: per cloud point, its entry in its chunk's list of cells
  int2 *gc_list, *gc_ent;    // [B][Nmax]  grid build, chunk c at c·GC_CHUNK: (cell key, points) of every cell of the chunk (k_gridcount); (slot, offset) then (compact cell id, first position) of the same entries (k_gridhash)
  int2 *gc_tab; int *gc_tabsel;   // [B][16384 = GH_H], [B]  grid build: the stream's cell table as k_gridhash leaves it (slot → compact id + 1, first position) for k_gridplace; 0 = it is in gh_key / gh_val
  int *gc_n; int gc_chunks, gc_P;   // [B][gc_chunks] entries per chunk; chunks per stream at most; workgroups per stream of k_gridcount / k_gridplace
  int *gh_rowlist, *gh_cells; // [B][Nmax]  hash path, streams beyond the LDS lists: x of the cells of every row (unordered inside the row) then point counts per cell; claimed slots in discovery order
  int *gh_rowfill;           // [B][nrows+1]  hash path: per-row fill cursors when the row table does not fit the LDS copy
  int *gh_key, *gh_val;      // [B][Hcell] hash path: the cell table of streams with more cells than the LDS table holds
  int gh_tier;   // table tier k_gridhash starts with (0 small LDS table, 1 big LDS table, 2 global memory; −1: per stream, from gh_hint)
  int *gh_hint;  // [B] occupied cells of the stream's latest grid build (one array for all copies of the per-frame state): a stream starts at the smallest tier that holds 17/16 of it and moves up when its table overflows
  int *slab_y, *slab_c, *slab_e;    // [B][MOR_MAXP+1]  slabs of the cell graph: first y-slice, first compact cell id, end of the look-ahead (cells of the next two y-slices)
  int g2_exact_only;         // test knob (MOR_G2_EXACT): the voxel ground variant takes no verdict from the screen — every voxel goes through the ordered sums
  int g2_nobet;              // test knob (MOR_G2_NOBET): the voxel ground variant bets on a mode bin no frame has — every frame loses its bet and marks the ground in k_g2_mark
  int cg_slow_tail;          // test knob (MOR_CG_SLOW_TAIL): the merge of the slab forests by the general code (cgf_body) even where the register / LDS form (cgf_fast) applies
  int P, cg_force_global;    // workgroups per stream of k_cg_slab this frame (= slabs per stream when every stream gets the same); test knob: forests in global memory
  int *slab_p; int slab_T;   // [B] slabs of each stream (slab_bounds); own cells per slab the host aims at when the slabs follow the streams' cell counts (0: d.P slabs for every stream)
  int cg_fused;              // the merge of the slab forests (k_cg_final's work) runs in each stream's last slab workgroup of k_cg_slab
  int *lroot_a, *lroot_b;    // [B][Nmax]  per cell: its local root in its own slab / in the previous slab's look-ahead (compact ids)
  int *parent2;              // [B][Nmax]  second global forest (odd slabs when they do not fit LDS)
  int *skey, *sidx;          // aliases of the radix buffers holding the cell-sorted (key, cloud index)
  int *ckey;                 // [B][Nmax]  distinct cell keys, ascending (n_occ of them)
  int *cstart;               // [B][Nmax+1]  first sorted position of each occupied cell
  int *row_start;            // [B][nrows+1]  first occupied cell of each (y,z) row
  int *vnz, *vnz_out;        // [B]  voxel ground variant: z layers of the VoxelGrid lattice per stream (pass A writes it through vnz_out; its later kernels read it as gnz)
  int *gnz, *gnz_out; int cg_nz; float cg_inv_cs;   // voxel ground variant: z layers of the clustering grid per stream (stream_grid); written by pass A through gnz_out
  unsigned short *rs16, *cx16; int rs16_stride, cx16_stride;   // [B][rs16_stride], [B][cx16_stride] (even strides ≥ rows + 1 / Nmax + 1)  16-bit copies of the row table and of the x of every occupied cell: the cell index the method-1 scoring tiers keep in LDS
  int Hcell, use_hash;       // capacity per stream of the grid build's global-memory cell table (power of two ≥ 4·Nmax); whether the 16-bit index is written (method 1)
  int *cmin;                 // [B][Nmax]  smallest cloud index in the cell
  float4 *cmeta;             // [B][2·Nmax]  per occupied cell: low corner of its point box (.w = cluster id bits), high corner
  float4 *crep;              // [B][Nmax]  per occupied cell: its first point (sample for the quick edge test of the cell graph)
  float4 *sorted;            // [B][Nmax]  (x,y,z, bits(cloud index)) in cell order
  int *scell;                // [B][Nmax]  compact cell id per position of `sorted`
  struct MorCellSum *csum;   // [B][Nmax]  per occupied cell: exact fixed-point coordinate sums of its points (cluster centroids are sums over cells)
  int4 *cgat;                // [B][Nmax]  per occupied cell, written by k_cg_final: (first slot in cl_pts − first position in `sorted`, cluster id or −1, cloud index of the cluster's first point if it lies in this cell else −1, –)
  int *clist;                // [B][Nmax]  occupied cells grouped by cluster (any order inside a cluster)
  int *cl_coff;              // [B][Kcap+1]  first entry of each cluster in clist
  int *parent;               // [B][Nmax]  union-find forest over occupied cells (parent ≤ child)
  int *cg_ovf;               // [B][2][MOR_CG_OVF][2]  k_cellgraph: candidate cell pairs beyond its LDS list; undecided big cell pairs
  int *croot;                // [B][Nmax]  flattened root per cell
  int *csize;                // [B][Nmax]  component size (points) at its root cell
  int *compmin;              // [B][Nmax]  smallest cloud index of the component, at its root cell
  int *cid_of_root;          // [B][Nmax]  cluster id of a root cell (−1: not kept)
  int *pcid;                 // [B][Nmax]  cluster id (−1 none) per cloud point
  int *ccid;                 // [B][Nmax]  cluster id per occupied cell (a cell is a clique ⇒ one cluster)
  int *ktile_cnt;            // [B][tiles_max]
  int *kcell, *kroot, *ksize; // [B][Kcap]  kept components: root cell, smallest cloud index, size
  int *csz;                  // [B][Kcap]  sizes in final cluster order
  int *krank_inv;            // [B][Kcap]  final cluster id → entry of the kept-component list
  int *rkeys[2], *rvals[2];  // [B][Nmax]  radix ping-pong of the cell sort (grid stage)
  int *rhist;                // [B][tiles_max][256]
  // frame-slotted (cb / ca)
  float4 *cl_pts[MOR_MAX_SLOTS];         // [B][Nmax]  clusters[k] points (:229), cluster after cluster, cell by cell inside a cluster (.w = bits of the cloud index); ca's are transformed in place (:550)
  int *cl_cid[MOR_MAX_SLOTS];            // [B][Nmax]  cluster id per cl_pts entry
  int *cl_off[MOR_MAX_SLOTS];            // [B][Kcap+1]
  int *chunk_off[MOR_MAX_SLOTS];         // [B][Kcap+1]  first reduction chunk of each cluster
  Red6 *part_back; int Wcap; // [B][Wcap]  per-chunk partials of the transform of ca (k_clusters' transform workgroups)
  float4 *centroid[MOR_MAX_SLOTS];       // [B][Kcap]  centroid_collection (:243)
  float4 *amin[MOR_MAX_SLOTS], *amax[MOR_MAX_SLOTS]; // [B][Kcap]  cluster AABBs (getMinMax3D, :272-274)
  float4 *xcent, *xamin, *xamax; // [B][Kcap]  ca's centroids and AABBs after the transform into cb's frame (:540-550)
  float4 *cl_first[MOR_MAX_SLOTS];       // [B][Kcap]  the point with the smallest cloud index of every cluster (clusters[k][0], the octree anchor of method 2)
  float4 *xfirst;                        // [B][Kcap]  ca's first points after the transform
  // pair stage
  int *nn_fwd, *nn_bwd;      // [B][Kcap]
  float *nn_fwd_d;           // [B][Kcap]
  int *pair_q, *pair_m;      // [B][Kcap]
  float *pair_d;             // [B][Kcap]
  int *pair_cnt;             // [B][Kcap]
  int *pair_of_prev, *pair_of_cur; // [B][Kcap]
  float4 *qrec;              // [B][Kcap][2]  per cluster of the previous frame, for the scoring tiers: (matched cluster's box low corner, pair index as int bits), (high corner, matched cluster) — one record instead of the chain pair_of_prev → pair_m → amin / amax
  int4 *wl; unsigned long long *wl_nb; // [B][Nmax], [B]  method-1 worklists after tier 1 (query, pair, matched cluster, –): E2-known queries from the front, block queries from the back; their counts share a word (low / high half)
  int4 *wl2; int *wl2_n;          // [B][Nmax], [B]  method-1 worklist of the wave tier
  unsigned long long *vox;   // [B][Hcap]
  struct MorVoxRec *vrec;    // [B][2][Kcap]  method 2: per cluster of ca (first half) / cb (second half) its pair and the origin of the pair's voxel lattice
  unsigned char *det;        // [B][Kcap]  detection_results of cb
  // filter stage
  unsigned *moving;          // [B][Kcap/32 + 2]  k_track_filter → k_out: a bit per cluster queued for removal, then the ExtractIndices size-check flag and the number of kept cloud points
  unsigned long long *out_desc;   // [B][tiles_max]  look-back descriptors of the output compaction (call epoch | kept points of the tile)
  unsigned filter_epoch;     // tag of this filterCloud call (never 0, never repeated): tag of k_out's look-back descriptors
  MorTrackDev *tr;           // [B]
  int2 *tr_corr;             // [B][MOR_TR_NB][Kcap]   corrs_vec (query, match), oldest first
  unsigned char *tr_res;     // [B][MOR_TR_NB+1][Kcap] res_vec, oldest first
  unsigned char *tr_lastdet; // [B][Kcap]  detection_results of the previous frame (ca)
  int *tr_match;             // [B][MOR_TR_MAXT+1]  latest filterCloud: number of tracked centroids its loop visited, then the cluster each was matched to, in mo_vec order (the reference's bounding-box markers, :641)
  int moving_confidence, static_confidence; float leave_off, catch_up;
  float4 *const *out_ptrs;   // [B] or null
  int out_step32;            // caller-provided output pointers receive PCL's 32-byte PointXYZI records (x, y, z, 1.0f | intensity, 0, 0, 0: what toPCLPointCloud2<PointXYZI> serialises, :690) instead of packed 16-byte points — the class adapter's output, written by k_out straight into page-locked caller memory
  // ---- pinned host mirrors written by the device (zero-copy summaries)
  MorFrameInfo *h_info;      // [B]
  float4 *h_centroid;        // [B][Kcap]
  int *h_cl_off;             // [B][Kcap+1]
  unsigned char *h_det;      // [B][Kcap]
  int *h_pair_q, *h_pair_m;  // [B][Kcap]
  float *h_pair_d;           // [B][Kcap]
  double *h_score;           // [B][Kcap]
  unsigned long long *h_nout;// [B]  points in the filtered cloud
  int *h_noff;               // [B]  first slot of the filtered cloud in the stream's `ground` buffer (Nmax − n_keep)
};

// Σ of the coordinates of a set of points as exact integers: x = a·2^-24 + b·2^-56 with a = ⌊x·2^24⌋ (|a| < 2^35 inside the largest
// grid) and 0 ≤ b < 2^32, summed separately in 64-bit integers (no carries needed up to 2^26 points).  Integer sums commute,
// so a centroid does not depend on the order in which cells, waves or atomics deliver the points.
struct MorCellSum { long long a[3], b[3]; };
// kernel ids for optional per-kernel event timing
enum MorKernelId {
  MK_CLASSIFY, MK_SCATTER, MK_SPLIT, MK_HEADS_COUNT, MK_HEADS_SCATTER, MK_CELLBOXES, MK_RHIST, MK_RSCAN, MK_RSCATTER,
  MK_SCORE_FAST, MK_SCORE_NB, MK_SCORE1, MK_VOX_CLEAR, MK_VOX_INSERT, MK_VOX_PROBE,
  MK_OUT, MK_G2_COV_MID, MK_G2_COV, MK_G2_COV_BIG, MK_G2_MODE, MK_G2_MARK, MK_TRACK_PUSH, MK_TRACK_FILTER, MK_GRIDHASH, MK_GRIDCOUNT, MK_GRIDPLACE, MK_CG_SLAB, MK_CG_FINAL, MK_CLUSTERS, MK_G2_CENT, MK_TRACK_PUSH_FILTER, MK_COUNT
};
extern const char *const mor_kernel_names[MK_COUNT];

struct MorLaunchTimer;   // engine-owned; records event pairs when enabled
#define MOR_MAX_PIECES 13   // pieces of a push, at most (voxel ground variant: its grid stage is six of them)
// piece ids: 7 split | 8 grid build (crop variant) or 10 … 15 (voxel ground variant) | 1 cell boxes | 2 cell graph | 3 clusters |
//            (3 also: transform of ca, correspondences) | 4 thread tiers of the scores / voxel kernels of method 2 | 5 wave tier | 6 thresholds + tracking
// small host → device copies on a stream as a one-workgroup kernel (the source is page-locked host memory the device reads directly)
void mor_launch_copy(void *dst, const void *src_pinned, size_t bytes, hipStream_t st);
void mor_launch_piece(const MorDev &d, int piece, hipStream_t st, MorLaunchTimer *tm);
void mor_launch_filter(const MorDev &d, hipStream_t st, MorLaunchTimer *tm, int part);
int mor_split_blocks_per_cu();
void mor_timer_begin(MorLaunchTimer *tm, int kernel_id, hipStream_t st);
void mor_timer_end(MorLaunchTimer *tm, int kernel_id, hipStream_t st);
