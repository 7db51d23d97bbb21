/*
 * mor_oracle.h — CPU restatement (the ORACLE) of the reference hot path
 *   MovingObjectRemoval::pushRawCloudAndPose() + filterCloud()
 *   (/root/reference/src/MovingObjectRemoval.cpp:516-611, :613-696).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (libmor_hip.so) never links,
 * loads or calls anything in oracle/.
 *
 * PARITY PINNING: the reference ships no tests, golden vectors or fixtures, and cannot
 * be compiled here (needs ROS + PCL 1.8 + FLANN + Eigen + Boost, none present).  The
 * oracle is therefore pinned by (1) an independent definition-level brute force
 * (numpy + scipy.sparse.csgraph, tests/test_oracle_bruteforce.py), (2) hand-checkable
 * known-answer scenes (tests/test_oracle_known_answers.py), (3) committed fixtures
 * generated after (1) and (2) agree (tests/golden/).  With respect to the real
 * PCL binary it is "parity unpinned" — see DESIGN.md.
 *
 * Style: single-threaded, PCL-like data flow (32-byte PointXYZI points, kd-tree with
 * leaf size 15, BFS flood-fill clustering, one kd-tree per matched cluster pair), so it
 * doubles as the timed CPU baseline ("port") in bench.py.
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).
 */
#ifndef MOR_ORACLE_H
#define MOR_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same field order / layout as include/mor_hip.h:mor_params (checked in tests via
 * oracle_sizeof_params()).  Members mirror include/MOR/MovingObjectRemoval.h:103-105. */
typedef struct oracle_params {
  float gp_limit, gp_leaf, bin_gap;
  int64_t min_cluster_size, max_cluster_size; /* reference: long */
  float volume_constraint, pde_lb, pde_ub;
  float leave_off_distance, catch_up_distance;
  float trim_x, trim_y, trim_z;
  float ec_distance_threshold, pde_distance_threshold;
  int32_t method_choice;            /* 1 = point-distance estimate, 2 = octree change */
  int32_t opc_normalization_factor; /* reference parses with stof into an int (:843) */
  int32_t ground_method;            /* 0 = crop (:526, active), 1 = voxel covariance (:527, intended) */
  float opc_resolution;             /* 0.1f, hard-coded at the call site :575 */
  int32_t volume_abs_int;           /* 0: abs(volp-volc) at :277 is fabs (libstdc++ >= 6); 1: it is C's int abs(int) (truncates first) */
  int32_t opc_anchor;               /* method-2 voxel lattice: 0 = anchored at p0 - res (getKeyBitSize re-centres the first box), 1 = at p0 - res/2 */
} oracle_params;

typedef struct oracle_ctx oracle_ctx;

size_t oracle_sizeof_params(void);

/* ctor — MovingObjectRemoval.cpp:368-391 (n_bad → moving_confidence, n_good → static_confidence) */
oracle_ctx *oracle_create(const oracle_params *p, int n_bad, int n_good);
void oracle_destroy(oracle_ctx *c);

/* pushRawCloudAndPose — :516-611.  `data` is a PCLPointCloud2-style blob: n_points records of
 * point_step bytes with float32 fields at the given byte offsets (off_intensity = 0xFFFFFFFF
 * when the blob has no intensity field ⇒ intensity stays 0, fromPCLPointCloud2 semantics).
 * pose = position xyz + quaternion xyzw.  Returns 0. */
int oracle_push(oracle_ctx *c, const void *data, uint64_t n_points, uint32_t point_step,
                uint32_t off_x, uint32_t off_y, uint32_t off_z, uint32_t off_intensity,
                const double pose_xyz_qxyzw[7]);

/* filterCloud — :613-696.  Writes the filtered cloud as packed (x,y,z,intensity) float32
 * quadruples; out_xyzi must hold ≥ T*4 floats (T = trimmed count of the last push).
 * Returns 0, or -1 when called before any push (the reference would crash). */
int oracle_filter(oracle_ctx *c, float *out_xyzi, uint64_t *n_out);

/* ---- read-backs of the state of the latest frame `cb` (for parity tests) ---- */
typedef struct oracle_counts {
  uint64_t n_in, n_trim, n_cloud, n_ground; /* N, T, M, G */
  uint32_t n_clusters, n_clustered;          /* K, C */
  uint32_t n_corr, n_tracks;
} oracle_counts;
void oracle_get_counts(const oracle_ctx *c, oracle_counts *out);
/* per trimmed point: cluster id ≥0, -1 non-ground unclustered, -2 ground/removed */
void oracle_get_labels(const oracle_ctx *c, int32_t *labels_T);
/* ground indices into the trimmed cloud (gp_indices, :86) in stored order */
void oracle_get_ground_indices(const oracle_ctx *c, int32_t *idx_G);
/* offsets[K+1], indices[C] (indices into `cloud`, ascending inside each cluster) */
void oracle_get_clusters(const oracle_ctx *c, int32_t *offsets, int32_t *indices);
void oracle_get_centroids(const oracle_ctx *c, float *xyz_K3);
void oracle_get_detection(const oracle_ctx *c, uint8_t *det_K);
/* correspondences of the last push: query (prev cluster), match (cur cluster), squared
 * centroid distance, movement score (:564-576) */
void oracle_get_correspondences(const oracle_ctx *c, int32_t *query, int32_t *match,
                                float *dist, double *score);
/* mo_vec: centroid xyz, confidence, max_confidence */
void oracle_get_tracks(const oracle_ctx *c, float *xyz_n3, int32_t *conf, int32_t *max_conf);
/* clusters of the PREVIOUS frame after the in-place transform (:540-551): offsets + xyzi */
uint32_t oracle_get_prev_cluster_count(const oracle_ctx *c);
uint32_t oracle_get_prev_clustered(const oracle_ctx *c);
void oracle_get_prev_transformed(const oracle_ctx *c, float *cent_K3, float *pts_C4);
/* mark_cluster (:7-58) of every cluster of `cb`: fp32-centroid position and box extent (zero extent -> 0.1) */
void oracle_get_markers(const oracle_ctx *c, float *pos_K3, float *scale_K3);
/* clusters the latest oracle_filter's loop over mo_vec matched its tracked centroids to, in loop order (the markers of :641, ids 1, 2, …); returns the count */
uint32_t oracle_get_moving_clusters(const oracle_ctx *c, int32_t *cluster_of_track);

/* Margin census (process-wide counters, see mor_oracle.c): [0] C1 pairs with d² within 4 ulp of r², [1] the same for the voxel ground variant's radius searches,
 * [2] far-subtree visits of radius searches taken only thanks to the pruning slack, [3] neighbours found in them (what FLANN's literal test would lose on this tree),
 * [4] / [5] the same for 1-NN searches (visits, new best found), [6] exact 1-NN distance ties, [7] equal-size clusters next to each other in the order,
 * [8] volume gates within 1e-6 of the constraint, [9] method-1 distances within 4 ulp of a bound, [10] method-2 points within one ulp of a voxel face,
 * [11] voxels with a covariance term within 1e-6 of 0.001, [12] radius queries, [13] 1-NN queries.  oracle_census_read returns the number of counters. */
void oracle_census_reset(void);
int oracle_census_read(unsigned long long *out, int n);
/* 1: the kd-tree prunes with FLANN's literal test (mindist <= worst, no slack); 0 (default): with the 1.0001 slack that misses nothing */
void oracle_set_literal_pruning(int on);

/* wall-clock seconds spent inside oracle_push + oracle_filter since creation */
double oracle_get_busy_seconds(const oracle_ctx *c);

#ifdef __cplusplus
}
#endif
#endif
