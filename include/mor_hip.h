/*
 * mor_hip.h — C ABI of libmor_hip.so: the MI355X-native (HIP, gfx950) implementation of the
 * reference hot path
 *     MovingObjectRemoval::pushRawCloudAndPose()   /root/reference/src/MovingObjectRemoval.cpp:516-611
 *     MovingObjectRemoval::filterCloud()           /root/reference/src/MovingObjectRemoval.cpp:613-696
 * behind the reference's class (include/MOR/MovingObjectRemoval.h:96-168).  The reference has no
 * FFI layer — the class *is* the boundary — so these are the entry points the header-level adapter
 * (include/MOR/MovingObjectRemoval.h in this repo) binds; SURVEY.md §8(b) lists them.
 *
 * Plain pointers and sizes only.  Every function returns MOR_OK (0) or a negative error code;
 * nothing here calls exit() (the reference exit(0)s on config errors, :703-707, :856-860 — that
 * behaviour lives in the adapter).  There is NO CPU fallback: without a usable HIP device
 * mor_batch_create() fails with MOR_ERR_HIP.
 *
 * Threading: like the reference (single-threaded ros::spin, external_sync_test.cpp:39) a batch is
 * not thread-safe; calls on one batch must be strictly ordered push → filter → push → …
 */
#ifndef MOR_HIP_H
#define MOR_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOR_OK 0
#define MOR_ERR_INVALID (-1)   /* bad argument / parameter */
#define MOR_ERR_NOT_READY (-2) /* filter before the first push (the reference would crash, :618/:681) */
#define MOR_ERR_HIP (-3)       /* HIP runtime error or no device; see mor_last_error() */
#define MOR_ERR_CAPACITY (-4)  /* more points than max_points, or more clusters than the cluster capacity */
#define MOR_NO_FIELD 0xFFFFFFFFu

/* The 17 numeric members of MovingObjectRemoval (include/MOR/MovingObjectRemoval.h:103-105; keys of
 * config/MOR_config.txt, parsed at MovingObjectRemoval.cpp:736-855) plus two values the reference
 * fixes in source. */
typedef struct mor_params {
  float gp_limit, gp_leaf, bin_gap;
  int64_t min_cluster_size, max_cluster_size; /* reference: long */
  float volume_constraint, pde_lb, pde_ub;
  float leave_off_distance, catch_up_distance;
  float trim_x, trim_y, trim_z;
  float ec_distance_threshold, pde_distance_threshold;
  int32_t method_choice;            /* 1 = point-distance estimate (:336-366), 2 = octree change (:309-334) */
  int32_t opc_normalization_factor; /* :843 */
  int32_t ground_method;            /* 0 = crop box (:526, the active call), 1 = voxel covariance (:527, intended semantics) */
  float opc_resolution;             /* 0.1f — literal at the call site :575 */
  int32_t volume_abs_int;           /* 0 (default): the unqualified abs(volp-volc) of :277 is fabs, as with libstdc++ >= 6; 1: it is C's int abs(int) — the
                                       difference is truncated towards zero first (what an older libstdc++ can pick); DESIGN.md §2 */
  int32_t opc_anchor;               /* voxel lattice of method 2 (OctreePointCloudChangeDetector, :319-329), anchored at the first point p0 of the previous
                                       cluster: 0 (default) = p0 - res (adoptBoundingBoxToPoint sets p0 +- res/2, getKeyBitSize then widens the empty
                                       tree to two voxels per axis and re-centres it); 1 = p0 - res/2 (no re-centring: SURVEY.md Appendix A's reading).
                                       Neither can be checked against a PCL build here; DESIGN.md §2 */
} mor_params;

/* One incoming cloud: a pcl::PCLPointCloud2-style blob (what fromPCLPointCloud2 consumes at :523).
 * n_points records of point_step bytes; float32 fields at the given byte offsets;
 * off_intensity = MOR_NO_FIELD when the blob has no "intensity" field (intensity stays 0). */
typedef struct mor_cloud_view {
  const void *data;
  uint64_t n_points;
  uint32_t point_step, off_x, off_y, off_z, off_intensity;
  int32_t on_device; /* 0: host memory, staged by the library; 1: device memory on the batch's GPU */
} mor_cloud_view;

/* Sizes of the latest frame `cb` of one stream. */
typedef struct mor_counts {
  uint64_t n_in, n_trim, n_cloud, n_ground; /* N, T (after x/y trim), M (`cloud`), G (`gp_indices`) */
  uint32_t n_clusters, n_clustered;          /* K, C */
  uint32_t n_corr, n_tracks;                 /* |mp| of the last push (:564), |mo_vec| */
} mor_counts;

typedef struct mor_batch mor_batch; /* B independent sensor streams sharing one GPU and one set of launches */

size_t mor_sizeof_params(void);
const char *mor_last_error(void);

/* MovingObjectRemoval::MovingObjectRemoval (:368-391) for n_streams independent instances:
 * n_bad → moving_confidence, n_good → static_confidence.  max_points bounds the points per cloud
 * (buffers are sized once for it; 288 GB of HBM3E makes that cheap).  device = HIP ordinal. */
mor_batch *mor_batch_create(const mor_params *p, int n_bad, int n_good, int n_streams, uint64_t max_points,
                            int device, int *err);
void mor_batch_destroy(mor_batch *b);
int mor_batch_streams(const mor_batch *b);

/* pushRawCloudAndPose (:516-611) for every stream of the batch at once: clouds[n_streams],
 * poses = n_streams × (position xyz, quaternion xyzw) in fp64.  All device work — including the temporal logic
 * (checkMovingClusterChain, :478-514), whose state lives on the device — is enqueued as batched launches; in the
 * default synchronous mode the call returns when they have finished and the per-stream summaries are on the host. */
int mor_push_batch(mor_batch *b, const mor_cloud_view *clouds, const double *poses_xyz_qxyzw);

/* filterCloud (:613-696) for every stream.  out[i] receives stream i's filtered cloud as packed
 * (x,y,z,intensity) float32 records — [cloud minus moving clusters, original order] ++ [ground
 * points in index order] — and n_out[i] its point count.  out[i] must hold n_in points.
 * out_on_device: 0 = host pointers, 1 = device pointers.  out may be NULL: results then stay in the
 * batch's own device buffers (mor_get_output_device). */
int mor_filter_batch(mor_batch *b, void *const *out, int out_on_device, uint64_t *n_out);
/* The same with a choice of the output record: out_point_step = 16 is mor_filter_batch; 32 makes the DEVICE write PCL's PointXYZI records — x@0 y@4 z@8 (1.0f @12)
 * intensity@16, zeros behind it: what toPCLPointCloud2<PointXYZI> serialises at :690 — straight into out[i], which must then be device-accessible memory
 * (out_on_device = 1: device memory, or page-locked host memory of mor_host_alloc / mor_host_register through its device pointer) of n_in x 32 bytes.  This is the class
 * adapter's output path: no 16-byte intermediate, no expansion loop on the host. */
int mor_filter_batch_ex(mor_batch *b, void *const *out, int out_on_device, uint64_t *n_out, uint32_t out_point_step);

/* Asynchronous mode (off by default).  With it on, mor_push_batch and mor_filter_batch (called with n_out == NULL) only
 * enqueue their work — host-resident input blobs are staged by copies that run beside the kernels of the frames in
 * flight, and with host output pointers (out_on_device = 0) every stream's filtered cloud leaves by a DMA copy of the
 * stream's input size behind the kernels (page-locked caller memory: mor_host_alloc; the buffers must not be touched
 * before the wait; the sizes are read after it, mor_get_output_device) — the tracking state lives on the device, so a push + filter
 * pair needs no host round trip, and return at once; mor_batch_wait blocks until everything enqueued has
 * finished and reports the errors any frame has raised since the last report (a sticky error word per stream; reporting
 * clears it).  Every read-back waits by itself but reports nothing: an error raised by an earlier frame stays pending until
 * mor_batch_wait (or the next synchronous push / filter) returns it. */
int mor_batch_set_async(mor_batch *b, int on);
int mor_batch_wait(mor_batch *b);

/* Device-resident result of the last filter for stream i (float4 records).  The filtered cloud is assembled in place in
 * one of the batch's per-frame buffers (the ground points are written there once, at the split; there is one buffer per frame
 * in flight, MOR_PIPE_DEPTH = 4 by default): the pointer stays valid for MOR_PIPE_DEPTH - 1 further pushes (three by default;
 * with MOR_PIPE_DEPTH=1 the very next push overwrites it). */
const void *mor_get_output_device(const mor_batch *b, int stream, uint64_t *n_out);

/* Single-stream forms used by the class adapter (a batch with one stream). */
typedef mor_batch mor_ctx;
mor_ctx *mor_create(const mor_params *p, int n_bad, int n_good, uint64_t max_points, int device, int *err);
int mor_push(mor_ctx *c, const void *data, uint64_t n_points, uint32_t point_step, uint32_t off_x, uint32_t off_y,
             uint32_t off_z, uint32_t off_intensity, const double pose_xyz_qxyzw[7]);
int mor_filter(mor_ctx *c, float *out_xyzi, uint64_t *n_out);
void mor_destroy(mor_ctx *c);

/* ---- read-backs of stream i's latest frame (parity tests, debugging, VISUALIZE side channel) ---- */
int mor_get_counts(const mor_batch *b, int stream, mor_counts *out);
/* per trimmed point: cluster id ≥ 0 (cluster_indices order, :221), -1 non-ground unclustered, -2 ground */
int mor_get_labels(const mor_batch *b, int stream, int32_t *labels_T);
/* gp_indices (:86): indices into the trimmed cloud, ascending */
int mor_get_ground_indices(const mor_batch *b, int stream, int32_t *idx_G);
/* cluster_indices (:218): offsets[K+1] and indices[C] into `cloud`, ascending inside a cluster;
 * clusters ordered by size descending, ties by first index.  (The device keeps a cluster's points cell by cell — nothing it
 * computes depends on their order; this read-back, the cluster collection and the markers rebuild the reference's order on
 * the host from the labels.) */
int mor_get_clusters(const mor_batch *b, int stream, int32_t *offsets, int32_t *indices);
int mor_get_centroids(const mor_batch *b, int stream, float *xyz_K3);  /* centroid_collection (:243) */
int mor_get_detection(const mor_batch *b, int stream, uint8_t *det_K); /* detection_results (:593-604) */
/* axis-aligned boxes of the clusters (getMinMax3D, :16 / :272-274): with the centroids they are the data of the
 * reference's debug markers (mark_cluster, :7-58: CUBE at the centroid, scale = max − min, zero extent → 0.1) */
int mor_get_boxes(const mor_batch *b, int stream, float *min_K3, float *max_K3);
/* the marker data itself, as mark_cluster computes it: position = FLOAT-accumulated centroid of the cluster's points (:15 — not
 * the fp64 centroid of :239-243), scale = box extent with zero extents replaced by 0.1 */
int mor_get_markers(const mor_batch *b, int stream, float *pos_K3, float *scale_K3);
/* the latest filterCloud's loop over mo_vec (:630-671): for every tracked centroid it visited, in order, the cluster of the latest frame it was matched to
 * (the cluster whose bounding box the reference publishes as a marker with id 1, 2, … at :641).  *n = entries (0 before the frame's first filterCloud);
 * cluster_of_track may be null to query the count (at most mor_get_tracks' count before that filterCloud). */
int mor_get_moving_clusters(const mor_batch *b, int stream, int32_t *cluster_of_track, uint32_t *n);
/* correspondence map mp (:564) + movement scores param_vec (:571/:575) of the last push */
int mor_get_correspondences(const mor_batch *b, int stream, int32_t *query, int32_t *match, float *dist, double *score);
/* mo_vec (header :109): centroid xyz, confidence, max_confidence */
int mor_get_tracks(const mor_batch *b, int stream, float *xyz_n3, int32_t *conf, int32_t *max_conf);
/* cluster_collection (:227, :258-260): the clustered points of `cb` concatenated in cluster order
 * (what push writes back into the caller's cloud under VISUALIZE, :553-558); out holds C records */
int mor_get_cluster_collection(const mor_batch *b, int stream, float *out_xyzi);

/* diagnostics of the last push of stream i: out[0] = occupied grid cells, out[1] = method-1 queries that
 * needed the wave tier, out[2] = method-1 queries left after the own-cell tier, out[3] = clustered points of the previous frame,
 * out[4] = voxels of the voxel ground variant whose ordered sums had to be evaluated, out[5] = cells (own + look-ahead) of the stream's largest slab of the cell graph */
int mor_get_stage_counts(const mor_batch *b, int stream, uint32_t *out, int n);

/* Per-frame summary log (the last 64 frames): lets a caller of the asynchronous mode inspect EVERY frame after one wait.
 * out[10] = frame, K, C, |mp|, checksum of the per-pair movement counts, checksum of detection_results, |mo_vec| after
 * the push, |mo_vec| after filterCloud, points in the filtered cloud, device flags of the frame. */
int mor_get_frame_log(const mor_batch *b, uint64_t frame, int stream, int64_t *out10);

/* Development hooks (exp/, tests): raw copy of a named intermediate device array of one stream (returns bytes copied or
 * a negative error), and the grid geometry / launch configuration of the latest push. */
long long mor_debug_read(const mor_batch *b, const char *name, int stream, void *out, size_t bytes);
int mor_debug_config(const mor_batch *b, int *out, int n);

/* ---- device-memory helpers so callers can keep clouds resident in HBM (bench, replay driver) ---- */
void *mor_device_alloc(int device, size_t bytes);
void mor_device_free(int device, void *p);
/* page-locked host memory: clouds handed over (and outputs received) in it cross PCIe by DMA at full rate; pageable
 * memory works too but is staged by the driver at a fraction of it */
void *mor_host_alloc(size_t bytes);
void mor_host_free(void *p);
/* page-lock caller-owned host memory (the buffer of a std::vector, say) and map it for the device; *device_ptr is the address kernels and out_on_device = 1 pointers
 * use for it.  Unregister before the memory is freed. */
int mor_host_register(void *p, size_t bytes, void **device_ptr);
int mor_host_unregister(void *p);
int mor_device_upload(int device, void *dst, const void *src, size_t bytes);
int mor_device_download(int device, void *dst, const void *src, size_t bytes);
int mor_device_synchronize(int device);
int mor_device_count(void);
/* "MOR_SRC_HASH=<24 hex digits>": hash of the sources + compiler flags the library was built from (build.py); "…=unknown" for a build made by hand. */
const char *mor_build_hash(void);
/* Host placement.  The thread that creates a batch and enqueues its frames should run on the NUMA node the GPU is attached to: the
 * command queues and the page-locked rings it fills live where it runs, and from the other socket the same run is 5–6 % slower
 * (measured on a two-socket MI355X host: 166 k against 176 k frame-pairs/s).  mor_device_numa_node returns that node (from the device's
 * PCI address and sysfs; −1 when it cannot be told); mor_bind_thread_to_device_node restricts the CALLING thread to the CPUs of that
 * node it is already allowed on — `share_index` of `share_count` equal slices of them when several processes serve GPUs of one node
 * (0, 1: all of them) — and returns the number of CPUs it kept (0: nothing changed).  The library never changes affinities by itself. */
int mor_device_numa_node(int device);
int mor_bind_thread_to_device_node(int device, int share_index, int share_count);

/* ---- timing hooks (HIP events on the batch's own stream) ---- */
/* milliseconds spent in device work of the last push / filter (event-timed), and the accumulated
 * time + launch count of the dominant kernel (cluster hook) since the last reset. */
int mor_get_last_timing(const mor_batch *b, float *push_ms, float *filter_ms);
int mor_kernel_timing_enable(mor_batch *b, int enable);
int mor_kernel_timing_read(mor_batch *b, int reset, char *names, size_t names_cap, float *ms_total, uint32_t *launches, int max_kernels);
/* the launches of the last timed leg as (index into the names of mor_kernel_timing_read, start ms, end ms) on one clock */
int mor_kernel_timeline_read(mor_batch *b, int *ids, float *t0_ms, float *t1_ms, int max_n);

#ifdef __cplusplus
}
#endif
#endif
