/*
 * mor_oracle.c — single-threaded CPU restatement of the reference hot path.
 * TEST INFRASTRUCTURE ONLY (see mor_oracle.h).  Every function cites the reference
 * lines it follows (paths relative to /root/reference/) and, where the reference
 * delegates to PCL/FLANN/tf, the library semantics recalled in SURVEY.md Appendix A
 * (tagged [PCL-1.8]; those libraries are not in this container — parity with the real
 * binaries is unpinned, parity with the definition-level brute force is tested).
 *
 * Floating point: compile with -ffp-contract=off; all fp32 expressions below are written
 * so that every operation is individually rounded, in the order PCL evaluates them.
 */
#include "mor_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <time.h>

/* ------------------------------------------------------------------ small utilities */
typedef struct { int *d; size_t n, cap; } ivec;
static void iv_push(ivec *v, int x) {
  if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 64; v->d = (int *)realloc(v->d, v->cap * sizeof(int)); }
  v->d[v->n++] = x;
}
static void iv_free(ivec *v) { free(v->d); v->d = NULL; v->n = v->cap = 0; }
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

/* pcl::PointXYZI is 32 bytes: {x,y,z,1.0f | intensity,pad,pad,pad} [PCL-1.8] */
typedef struct { float x, y, z, w, intensity, p1, p2, p3; } opoint;

/* ------------------------------------------------------------------ kd-tree
 * Stands in for pcl::KdTreeFLANN → FLANN KDTreeSingleIndex(leaf 15), L2_Simple<float>
 * [PCL-1.8] (call sites MovingObjectRemoval.cpp:115-125, :213-218, :291-294, :343-350,
 * :618/:636).  Own implementation of the textbook bounding-box kd-tree (midpoint split
 * of the widest side, points reordered for locality, incremental per-axis lower
 * bounds).  Result sets are mathematically defined — radius: all d² < r²; 1-NN: minimum
 * d², ties → lowest index (the reference leaves ties to FLANN's traversal order). */
#define KD_LEAF 15
typedef struct { int dim; int a, b; float lo, hi; } kdnode; /* dim<0: leaf, points [a,b); else children a,b; lo=left.high, hi=right.low */
typedef struct {
  int n; int *perm; float *xyz; /* reordered coords, 3 per point */
  kdnode *nodes; int n_nodes, cap_nodes; float bb[3][2];
} kdtree;

static inline float sqdist3(const float *a, const float *b) {
  /* L2_Simple: diff = a-b; result += diff*diff, sequentially over x,y,z [PCL-1.8] */
  float d0 = a[0] - b[0], d1 = a[1] - b[1], d2 = a[2] - b[2];
  float r = d0 * d0; r = r + d1 * d1; r = r + d2 * d2; return r;
}
static int kd_new_node(kdtree *t) {
  if (t->n_nodes == t->cap_nodes) { t->cap_nodes = t->cap_nodes ? 2 * t->cap_nodes : 256; t->nodes = (kdnode *)realloc(t->nodes, t->cap_nodes * sizeof(kdnode)); }
  return t->n_nodes++;
}
static int kd_build_rec(kdtree *t, const float *src, int stride, int l, int r, float bb[3][2]) {
  int id = kd_new_node(t);
  if (r - l <= KD_LEAF) {
    for (int d = 0; d < 3; ++d) { bb[d][0] = FLT_MAX; bb[d][1] = -FLT_MAX; }
    for (int i = l; i < r; ++i) { const float *p = src + (size_t)t->perm[i] * stride;
      for (int d = 0; d < 3; ++d) { if (p[d] < bb[d][0]) bb[d][0] = p[d]; if (p[d] > bb[d][1]) bb[d][1] = p[d]; } }
    t->nodes[id].dim = -1; t->nodes[id].a = l; t->nodes[id].b = r; return id;
  }
  int dim = 0; float span = bb[0][1] - bb[0][0];
  for (int d = 1; d < 3; ++d) if (bb[d][1] - bb[d][0] > span) { span = bb[d][1] - bb[d][0]; dim = d; }
  float mn = FLT_MAX, mx = -FLT_MAX;
  for (int i = l; i < r; ++i) { float v = src[(size_t)t->perm[i] * stride + dim]; if (v < mn) mn = v; if (v > mx) mx = v; }
  float cut = (bb[dim][0] + bb[dim][1]) * 0.5f; if (cut < mn) cut = mn; if (cut > mx) cut = mx;
  /* three-way partition: < cut | == cut | > cut, then pick the split closest to the middle */
  int i = l, lt = l, gt = r;
  while (i < gt) { float v = src[(size_t)t->perm[i] * stride + dim];
    if (v < cut) { int tmp = t->perm[i]; t->perm[i] = t->perm[lt]; t->perm[lt] = tmp; ++i; ++lt; }
    else if (v > cut) { --gt; int tmp = t->perm[i]; t->perm[i] = t->perm[gt]; t->perm[gt] = tmp; }
    else ++i; }
  int mid = l + (r - l) / 2, split = (lt > mid) ? lt : (gt < mid) ? gt : mid;
  if (split == l) split = l + 1; if (split == r) split = r - 1;
  float lb[3][2], rb[3][2]; memcpy(lb, bb, sizeof lb); memcpy(rb, bb, sizeof rb);
  lb[dim][1] = cut; rb[dim][0] = cut;
  int a = kd_build_rec(t, src, stride, l, split, lb);
  int b = kd_build_rec(t, src, stride, split, r, rb);
  kdnode *nd = &t->nodes[id]; nd->dim = dim; nd->a = a; nd->b = b; nd->lo = lb[dim][1]; nd->hi = rb[dim][0];
  for (int d = 0; d < 3; ++d) { bb[d][0] = lb[d][0] < rb[d][0] ? lb[d][0] : rb[d][0]; bb[d][1] = lb[d][1] > rb[d][1] ? lb[d][1] : rb[d][1]; }
  return id;
}
/* src: n points, `stride` floats apart, xyz first */
static void kd_build(kdtree *t, const float *src, int stride, int n) {
  memset(t, 0, sizeof *t); t->n = n; if (n == 0) return;
  t->perm = (int *)malloc(n * sizeof(int)); for (int i = 0; i < n; ++i) t->perm[i] = i;
  for (int d = 0; d < 3; ++d) { t->bb[d][0] = FLT_MAX; t->bb[d][1] = -FLT_MAX; }
  for (int i = 0; i < n; ++i) for (int d = 0; d < 3; ++d) { float v = src[(size_t)i * stride + d]; if (v < t->bb[d][0]) t->bb[d][0] = v; if (v > t->bb[d][1]) t->bb[d][1] = v; }
  float bb[3][2]; memcpy(bb, t->bb, sizeof bb);
  kd_build_rec(t, src, stride, 0, n, bb);
  memcpy(t->bb, bb, sizeof bb);
  t->xyz = (float *)malloc((size_t)n * 3 * sizeof(float));
  for (int i = 0; i < n; ++i) memcpy(t->xyz + 3 * (size_t)i, src + (size_t)t->perm[i] * stride, 3 * sizeof(float));
}
static void kd_free(kdtree *t) { free(t->perm); free(t->xyz); free(t->nodes); memset(t, 0, sizeof *t); }

/* ------------------------------------------------------------------ margin census + literal pruning (test infrastructure of the test infrastructure)
 * The reference's numbers come out of PCL / FLANN binaries this image cannot build, so the restatement is pinned by definition-level second implementations only.  How much
 * could a real FLANN change?  Two things are counted while the oracle runs (process-wide, oracle_census_*):
 *  - how many decisions sit within a few ulp of their threshold (an edge at d² ≈ r², a 1-NN tie, equal-size clusters, a volume gate at the constraint, a method-1 distance at a
 *    bound, a method-2 point at a voxel face, a covariance term at 0.001): only those can differ between two correct evaluations;
 *  - how many results the kd-tree owes to the SLACK of its pruning test.  FLANN's KDTreeSingleIndex descends into the far child iff `mindistsq * epsError <= worstDist` with
 *    epsError = 1 — the fp32 lower bound itself may round above a true distance, and FLANN then misses the point; this restatement prunes with a 1.0001 slack and misses
 *    nothing.  oracle_set_literal_pruning(1) switches to FLANN's literal test (on THIS tree: FLANN's own split planes differ), and tests/test_margin_census.py holds every
 *    golden stream to identical results in both modes. */
enum { CEN_EDGE_C1 = 0, CEN_EDGE_G2, CEN_RAD_SLACK_VISITS, CEN_RAD_SLACK_HITS, CEN_NN_SLACK_VISITS, CEN_NN_SLACK_WINS, CEN_NN_TIES, CEN_EQUAL_SIZE, CEN_VOLUME_GATE, CEN_PDE_BOUND,
       CEN_VOXEL_FACE, CEN_G2_TERM, CEN_RADIUS_QUERIES, CEN_NN_QUERIES, CEN_COUNT };
static unsigned long long g_cen[CEN_COUNT];
static int g_literal = 0, g_edge_kind = CEN_EDGE_C1;
void oracle_census_reset(void) { memset(g_cen, 0, sizeof g_cen); }
int oracle_census_read(unsigned long long *out, int n) { for (int i = 0; i < n && i < CEN_COUNT; ++i) out[i] = g_cen[i]; return CEN_COUNT; }
void oracle_set_literal_pruning(int on) { g_literal = on ? 1 : 0; }
static inline int within_ulps(float a, float b, int k) {   /* |a − b| ≤ k ulp (same sign, finite) */
  int ia, ib; memcpy(&ia, &a, 4); memcpy(&ib, &b, 4);
  if ((ia < 0) != (ib < 0)) return a == b;
  long long d = (long long)ia - (long long)ib; if (d < 0) d = -d; return d <= k;
}
typedef struct { const kdtree *t; const float *q; float r2; ivec *out; float *out_d; size_t dcap; int slack_depth; } kd_rq;
static void kd_radius_rec(kd_rq *s, int id, float mind, float *dists) {
  const kdnode *nd = &s->t->nodes[id];
  if (nd->dim < 0) {
    for (int i = nd->a; i < nd->b; ++i) { float d = sqdist3(s->q, s->t->xyz + 3 * (size_t)i);
      if (within_ulps(d, s->r2, 4)) g_cen[g_edge_kind]++;
      if (d < s->r2) { iv_push(s->out, s->t->perm[i]); if (s->slack_depth) g_cen[CEN_RAD_SLACK_HITS]++; } } /* strict <, RadiusResultSet [PCL-1.8] */
    return;
  }
  float v = s->q[nd->dim], d1 = v - nd->lo, d2 = v - nd->hi; int near, far; float cut;
  if (d1 + d2 < 0) { near = nd->a; far = nd->b; cut = d2 * d2; } else { near = nd->b; far = nd->a; cut = d1 * d1; }
  kd_radius_rec(s, near, mind, dists);
  float old = dists[nd->dim]; float m2 = mind + cut - old; dists[nd->dim] = cut;
  if (m2 <= s->r2) kd_radius_rec(s, far, m2, dists);   /* FLANN's test (epsError = 1) */
  else if (!g_literal && m2 <= s->r2 * 1.0001f) {   /* slack: the fp32 bound may round above the true distance */
    g_cen[CEN_RAD_SLACK_VISITS]++; s->slack_depth++; kd_radius_rec(s, far, m2, dists); s->slack_depth--; }
  dists[nd->dim] = old;
}
/* appends indices (original numbering, unsorted) of all points with d² < r2 */
static void kd_radius(const kdtree *t, const float *q, float r2, ivec *out) {
  if (t->n == 0) return;
  float dists[3], mind = 0;
  for (int d = 0; d < 3; ++d) { dists[d] = 0; if (q[d] < t->bb[d][0]) { float e = q[d] - t->bb[d][0]; dists[d] = e * e; } else if (q[d] > t->bb[d][1]) { float e = q[d] - t->bb[d][1]; dists[d] = e * e; } mind += dists[d]; }
  g_cen[CEN_RADIUS_QUERIES]++;
  if (!g_literal && mind > r2 * 1.0001f) return;   /* (FLANN has no test at the root) */
  kd_rq s = { t, q, r2, out, NULL, 0, 0 };
  kd_radius_rec(&s, 0, mind, dists);
}
typedef struct { const kdtree *t; const float *q; float best; int besti; int slack_depth; } kd_nq;
static void kd_nn_rec(kd_nq *s, int id, float mind, float *dists) {
  const kdnode *nd = &s->t->nodes[id];
  if (nd->dim < 0) {
    for (int i = nd->a; i < nd->b; ++i) { float d = sqdist3(s->q, s->t->xyz + 3 * (size_t)i); int oi = s->t->perm[i];
      if (d == s->best && oi != s->besti && s->besti >= 0) g_cen[CEN_NN_TIES]++;
      if (d < s->best || (d == s->best && oi < s->besti)) { s->best = d; s->besti = oi; if (s->slack_depth) g_cen[CEN_NN_SLACK_WINS]++; } }
    return;
  }
  float v = s->q[nd->dim], d1 = v - nd->lo, d2 = v - nd->hi; int near, far; float cut;
  if (d1 + d2 < 0) { near = nd->a; far = nd->b; cut = d2 * d2; } else { near = nd->b; far = nd->a; cut = d1 * d1; }
  kd_nn_rec(s, near, mind, dists);
  float old = dists[nd->dim]; float m2 = mind + cut - old; dists[nd->dim] = cut;
  if (m2 <= s->best) kd_nn_rec(s, far, m2, dists);   /* FLANN's test */
  else if (!g_literal && m2 <= s->best * 1.0001f) { g_cen[CEN_NN_SLACK_VISITS]++; s->slack_depth++; kd_nn_rec(s, far, m2, dists); s->slack_depth--; }
  dists[nd->dim] = old;
}
/* 1-NN; returns index or -1 for an empty tree; *d2 = squared fp32 distance */
static int kd_nn(const kdtree *t, const float *q, float *d2) {
  if (t->n == 0) return -1;
  float dists[3], mind = 0;
  for (int d = 0; d < 3; ++d) { dists[d] = 0; if (q[d] < t->bb[d][0]) { float e = q[d] - t->bb[d][0]; dists[d] = e * e; } else if (q[d] > t->bb[d][1]) { float e = q[d] - t->bb[d][1]; dists[d] = e * e; } mind += dists[d]; }
  g_cen[CEN_NN_QUERIES]++;
  kd_nq s = { t, q, INFINITY, -1, 0 };
  kd_nn_rec(&s, 0, mind, dists);
  *d2 = s.best; return s.besti;
}

/* ------------------------------------------------------------------ frame state
 * struct MovingObjectDetectionCloud — include/MOR/MovingObjectRemoval.h:7-56 */
typedef struct {
  opoint *raw; size_t n_raw;        /* raw_cloud after x/y trim (T) */
  opoint *cloud; size_t n_cloud;    /* cloud after ground removal (M) */
  int *cloud_src;                   /* index in raw of each cloud point (not in the reference; for label read-back) */
  int *gp; size_t n_gp;             /* gp_indices (G) */
  int K; int *cl_off; int *cl_idx;  /* cluster_indices */
  opoint **clusters;                /* clusters[k]: copied points (transformed in place when the frame becomes `ca`) */
  float *centroid;                  /* centroid_collection, 3 floats each */
  unsigned char *det;               /* detection_results */
  double R[3][3], o[3];             /* tf::Pose ps */
  int init; size_t n_in;
} frame;

static void frame_free(frame *f) {
  if (!f) return;
  free(f->raw); free(f->cloud); free(f->cloud_src); free(f->gp); free(f->cl_off); free(f->cl_idx);
  if (f->clusters) { for (int k = 0; k < f->K; ++k) free(f->clusters[k]); free(f->clusters); }
  free(f->centroid); free(f->det); free(f);
}

typedef struct { int query, match; float dist; } corr_t;
typedef struct { corr_t *c; int n; } corr_list;
typedef struct { unsigned char *v; int n; } bool_vec;
typedef struct { float c[3]; int confidence, max_confidence; } mo_centroid; /* header :83-94 */

struct oracle_ctx {
  oracle_params p; int moving_confidence, static_confidence;
  frame *ca, *cb;
  /* deques corrs_vec / res_vec (header :112-116) */
  corr_list *corrs_vec; int n_corrs_vec;
  bool_vec *res_vec; int n_res_vec;
  mo_centroid *mo; int n_mo, cap_mo;
  /* last push's correspondences + scores (for read-back) */
  corr_t *last_corr; double *last_score; int n_last_corr;
  ivec marked; /* clusters the latest filterCloud published a bounding-box marker for, in loop order (:641) */
  double busy;
};

size_t oracle_sizeof_params(void) { return sizeof(oracle_params); }

static frame *frame_new(void) { return (frame *)calloc(1, sizeof(frame)); }

oracle_ctx *oracle_create(const oracle_params *p, int n_bad, int n_good) {
  /* MovingObjectRemoval.cpp:368 (moving_confidence(n_bad), static_confidence(n_good)), :387-389 */
  oracle_ctx *c = (oracle_ctx *)calloc(1, sizeof *c);
  c->p = *p; c->moving_confidence = n_bad; c->static_confidence = n_good;
  c->ca = frame_new(); c->cb = frame_new();
  return c;
}
void oracle_destroy(oracle_ctx *c) {
  if (!c) return;
  frame_free(c->ca); frame_free(c->cb);
  for (int i = 0; i < c->n_corrs_vec; ++i) free(c->corrs_vec[i].c);
  for (int i = 0; i < c->n_res_vec; ++i) free(c->res_vec[i].v);
  free(c->corrs_vec); free(c->res_vec); free(c->mo); free(c->last_corr); free(c->last_score); iv_free(&c->marked); free(c);
}

/* ------------------------------------------------------------------ G1
 * MovingObjectDetectionCloud::groundPlaneRemoval(x,y,z) — MovingObjectRemoval.cpp:62-88.
 * PassThrough x then y (finite xyz and min ≤ v ≤ max, order kept) [PCL-1.8], then
 * CropBox(min=(-x,-y,gp_limit), max=(x,y,z)): outside ⇔ any coord < min or > max;
 * removed indices recorded ascending [PCL-1.8]. */
static void ground_removal_crop(frame *f, const oracle_params *p) {
  float X = p->trim_x, Y = p->trim_y, Z = p->trim_z;
  /* :66-70 PassThrough "x" in [-x, x] (in place through a temporary copy) */
  opoint *tmp = (opoint *)malloc((f->n_raw ? f->n_raw : 1) * sizeof(opoint)); size_t m = 0;
  for (size_t i = 0; i < f->n_raw; ++i) { const opoint *q = &f->raw[i];
    if (!isfinite(q->x) || !isfinite(q->y) || !isfinite(q->z)) continue;
    if (q->x < -X || q->x > X) continue; tmp[m++] = *q; }
  /* :71-74 PassThrough "y" in [-y, y] */
  size_t t = 0;
  for (size_t i = 0; i < m; ++i) { const opoint *q = &tmp[i]; if (q->y < -Y || q->y > Y) continue; f->raw[t++] = *q; }
  free(tmp); f->n_raw = t;
  /* :78-86 CropBox; cloud keeps the inside points in order, gp_indices the rest */
  f->cloud = (opoint *)malloc((t ? t : 1) * sizeof(opoint)); f->cloud_src = (int *)malloc((t ? t : 1) * sizeof(int));
  f->gp = (int *)malloc((t ? t : 1) * sizeof(int)); f->n_cloud = 0; f->n_gp = 0;
  float mnx = -X, mny = -Y, mnz = p->gp_limit;
  for (size_t i = 0; i < t; ++i) { const opoint *q = &f->raw[i];
    int outside = (q->x < mnx || q->y < mny || q->z < mnz) || (q->x > X || q->y > Y || q->z > Z);
    if (outside) f->gp[f->n_gp++] = (int)i; else { f->cloud_src[f->n_cloud] = (int)i; f->cloud[f->n_cloud++] = *q; } }
}

/* ------------------------------------------------------------------ G2 (intended semantics)
 * MovingObjectDetectionCloud::groundPlaneRemoval(x,y) — :90-200.  Dead code in the
 * reference (call commented out at :527; dereferences a null shared_ptr at :188).
 * Restated with the deterministic definition from DESIGN.md:
 *   VoxelGrid order inside a voxel = ascending point index (PCL uses an unstable sort);
 *   mode-bin ties → smallest bin key (reference: unordered_map iteration order);
 *   ground index list de-duplicated and sorted (the literal list has duplicates and
 *   would trip ExtractIndices' size check). */
typedef struct { long long idx; int pt; } vg_item;
static int vg_cmp(const void *a, const void *b) { const vg_item *x = (const vg_item *)a, *y = (const vg_item *)b;
  if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1; return x->pt < y->pt ? -1 : (x->pt > y->pt); }
typedef struct { float d; int i; } di_item;
static int di_cmp(const void *a, const void *b) { const di_item *x = (const di_item *)a, *y = (const di_item *)b;
  if (x->d != y->d) return x->d < y->d ? -1 : 1; return x->i < y->i ? -1 : (x->i > y->i); }
static int int_cmp(const void *a, const void *b) { int x = *(const int *)a, y = *(const int *)b; return x < y ? -1 : (x > y); }

static void ground_removal_voxel(frame *f, const oracle_params *p) {
  float X = p->trim_x, Y = p->trim_y;
  /* :94-102 same x/y PassThrough pair */
  size_t t = 0;
  for (size_t i = 0; i < f->n_raw; ++i) { const opoint q = f->raw[i];
    if (!isfinite(q.x) || !isfinite(q.y) || !isfinite(q.z)) continue;
    if (q.x < -X || q.x > X) continue; if (q.y < -Y || q.y > Y) continue; f->raw[t++] = q; }
  f->n_raw = t;
  f->cloud = (opoint *)malloc((t ? t : 1) * sizeof(opoint)); f->cloud_src = (int *)malloc((t ? t : 1) * sizeof(int));
  f->gp = (int *)malloc((t ? t : 1) * sizeof(int)); f->n_cloud = 0; f->n_gp = 0;
  unsigned char *is_ground = (unsigned char *)calloc(t ? t : 1, 1);
  if (t > 0) {
    /* :110-113 VoxelGrid(leaf) → dsc [PCL-1.8]: min_b=floor(min*inv), idx=(floor(x*inv)-min_b.x)+…, centroids in fp32 */
    float leaf = p->gp_leaf, inv = 1.0f / leaf;
    float mn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, mx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (size_t i = 0; i < t; ++i) { const float *q = &f->raw[i].x; for (int d = 0; d < 3; ++d) { if (q[d] < mn[d]) mn[d] = q[d]; if (q[d] > mx[d]) mx[d] = q[d]; } }
    int minb[3], maxb[3]; long long div[3];
    for (int d = 0; d < 3; ++d) { minb[d] = (int)floorf(mn[d] * inv); maxb[d] = (int)floorf(mx[d] * inv); div[d] = (long long)maxb[d] - minb[d] + 1; }
    vg_item *items = (vg_item *)malloc(t * sizeof(vg_item));
    for (size_t i = 0; i < t; ++i) { const opoint *q = &f->raw[i];
      int i0 = (int)(floorf(q->x * inv) - (float)minb[0]), i1 = (int)(floorf(q->y * inv) - (float)minb[1]), i2 = (int)(floorf(q->z * inv) - (float)minb[2]);
      items[i].idx = i0 + i1 * div[0] + i2 * div[0] * div[1]; items[i].pt = (int)i; }
    qsort(items, t, sizeof(vg_item), vg_cmp);
    /* :115-116 kd-tree on the trimmed cloud */
    kdtree tree; kd_build(&tree, &f->raw[0].x, 8, (int)t);
    double rr = (double)leaf; float r2 = (float)(rr * rr);
    /* accepted voxels: bin id + neighbour list */
    int n_acc = 0, cap_acc = 256; int *acc_bin = (int *)malloc(cap_acc * sizeof(int)); ivec *acc_nb = (ivec *)malloc(cap_acc * sizeof(ivec));
    ivec nb = { 0 }; di_item *srt = NULL; size_t srt_cap = 0;
    for (size_t s = 0; s < t;) {
      size_t e = s; float sx = 0, sy = 0, sz = 0;
      while (e < t && items[e].idx == items[s].idx) { const opoint *q = &f->raw[items[e].pt]; sx += q->x; sy += q->y; sz += q->z; ++e; }
      float n = (float)(e - s); float c[3] = { sx / n, sy / n, sz / n }; s = e;
      /* :125 radiusSearch(dsc[i], gp_leaf), sorted by (d², index) [PCL-1.8] */
      nb.n = 0; g_edge_kind = CEN_EDGE_G2; kd_radius(&tree, c, r2, &nb); g_edge_kind = CEN_EDGE_C1;
      if (nb.n <= 3) continue; /* :131 */
      if (nb.n > srt_cap) { srt_cap = nb.n * 2; srt = (di_item *)realloc(srt, srt_cap * sizeof(di_item)); }
      for (size_t j = 0; j < nb.n; ++j) { srt[j].i = nb.d[j]; srt[j].d = sqdist3(c, &f->raw[nb.d[j]].x); }
      qsort(srt, nb.n, sizeof(di_item), di_cmp);
      /* :142 compute3DCentroid (fp32, sequential) */
      float cx = 0, cy = 0, cz = 0;
      for (size_t j = 0; j < nb.n; ++j) { const opoint *q = &f->raw[srt[j].i]; cx += q->x; cy += q->y; cz += q->z; }
      float fn = (float)nb.n; cx /= fn; cy /= fn; cz /= fn;
      /* :144 computeCovarianceMatrix un-normalised, fp32 */
      float c02 = 0, c12 = 0, c22 = 0;
      for (size_t j = 0; j < nb.n; ++j) { const opoint *q = &f->raw[srt[j].i]; float dx = q->x - cx, dy = q->y - cy, dz = q->z - cz;
        c12 += dy * dz; c22 += dz * dz; c02 += dz * dx; }
      if (fabs((double)fabsf(c02) - 0.001) <= 1e-6 || fabs((double)fabsf(c12) - 0.001) <= 1e-6 || fabs((double)fabsf(c22) - 0.001) <= 1e-6) g_cen[CEN_G2_TERM]++;
      if (!((double)fabsf(c02) < 0.001 && (double)fabsf(c12) < 0.001 && (double)fabsf(c22) < 0.001)) continue; /* :145 */
      if (n_acc == cap_acc) { cap_acc *= 2; acc_bin = (int *)realloc(acc_bin, cap_acc * sizeof(int)); acc_nb = (ivec *)realloc(acc_nb, cap_acc * sizeof(ivec)); }
      acc_bin[n_acc] = (int)(c[2] * 10); /* :166 key = (float)((int)(z*10))/bin_gap — grouping is by the int */
      memset(&acc_nb[n_acc], 0, sizeof(ivec)); for (size_t j = 0; j < nb.n; ++j) iv_push(&acc_nb[n_acc], srt[j].i);
      ++n_acc;
    }
    /* :169-178 mode bin; ties → smallest key (documented deviation) */
    if (n_acc > 0) {
      int *bins = (int *)malloc(n_acc * sizeof(int)); memcpy(bins, acc_bin, n_acc * sizeof(int)); qsort(bins, n_acc, sizeof(int), int_cmp);
      int best_bin = bins[0], best_cnt = 0;
      for (int i = 0; i < n_acc;) { int j = i; while (j < n_acc && bins[j] == bins[i]) ++j; if (j - i > best_cnt) { best_cnt = j - i; best_bin = bins[i]; } i = j; }
      free(bins);
      for (int v = 0; v < n_acc; ++v) if (acc_bin[v] == best_bin) for (size_t j = 0; j < acc_nb[v].n; ++j) is_ground[acc_nb[v].d[j]] = 1; /* :184-191 */
    }
    for (int v = 0; v < n_acc; ++v) iv_free(&acc_nb[v]);
    free(acc_bin); free(acc_nb); iv_free(&nb); free(srt); free(items); kd_free(&tree);
  }
  /* :194-198 ExtractIndices(negative) */
  for (size_t i = 0; i < t; ++i) { if (is_ground[i]) f->gp[f->n_gp++] = (int)i; else { f->cloud_src[f->n_cloud] = (int)i; f->cloud[f->n_cloud++] = f->raw[i]; } }
  free(is_ground);
}

/* ------------------------------------------------------------------ C1 + C2
 * MovingObjectDetectionCloud::computeClusters — :202-262.
 * EuclideanClusterExtraction::extract [PCL-1.8]: kd-tree, BFS flood fill over
 * radiusSearch(tolerance) with radius² = (float)((double)tol*(double)tol), strict <;
 * keep min ≤ size ≤ max; indices sorted; clusters sorted by size descending (ties here:
 * ascending first index — the reference's std::sort leaves ties unspecified). */
typedef struct { int off, n, first; } cl_rec;
static int cl_cmp(const void *a, const void *b) { const cl_rec *x = (const cl_rec *)a, *y = (const cl_rec *)b;
  if (x->n != y->n) return x->n > y->n ? -1 : 1; return x->first < y->first ? -1 : (x->first > y->first); }

static void compute_clusters(frame *f, const oracle_params *p) {
  int M = (int)f->n_cloud;
  double tol = (double)p->ec_distance_threshold; float r2 = (float)(tol * tol);
  kdtree tree; kd_build(&tree, M ? &f->cloud[0].x : NULL, 8, M);
  unsigned char *processed = (unsigned char *)calloc(M ? M : 1, 1);
  ivec all = { 0 }; cl_rec *recs = NULL; int n_recs = 0, cap_recs = 0;
  ivec queue = { 0 }, nn = { 0 };
  for (int i = 0; i < M; ++i) {
    if (processed[i]) continue;
    queue.n = 0; iv_push(&queue, i); processed[i] = 1;
    for (size_t s = 0; s < queue.n; ++s) {
      nn.n = 0; kd_radius(&tree, &f->cloud[queue.d[s]].x, r2, &nn);
      for (size_t j = 0; j < nn.n; ++j) { int q = nn.d[j]; if (processed[q]) continue; processed[q] = 1; iv_push(&queue, q); }
    }
    if ((long long)queue.n >= p->min_cluster_size && (long long)queue.n <= p->max_cluster_size) {
      qsort(queue.d, queue.n, sizeof(int), int_cmp);
      if (n_recs == cap_recs) { cap_recs = cap_recs ? 2 * cap_recs : 64; recs = (cl_rec *)realloc(recs, cap_recs * sizeof(cl_rec)); }
      recs[n_recs].off = (int)all.n; recs[n_recs].n = (int)queue.n; recs[n_recs].first = queue.d[0]; ++n_recs;
      for (size_t j = 0; j < queue.n; ++j) iv_push(&all, queue.d[j]);
    }
  }
  qsort(recs, n_recs, sizeof(cl_rec), cl_cmp);
  for (int k = 1; k < n_recs; ++k) if (recs[k].n == recs[k - 1].n) g_cen[CEN_EQUAL_SIZE]++;   /* std::sort leaves their order unspecified (:217) */
  f->K = n_recs; f->cl_off = (int *)malloc((n_recs + 1) * sizeof(int)); f->cl_idx = (int *)malloc((all.n ? all.n : 1) * sizeof(int));
  f->clusters = (opoint **)calloc(n_recs ? n_recs : 1, sizeof(opoint *)); f->centroid = (float *)malloc((n_recs ? n_recs : 1) * 3 * sizeof(float));
  f->det = (unsigned char *)calloc(n_recs ? n_recs : 1, 1); /* :250-254 all false */
  int pos = 0;
  for (int k = 0; k < n_recs; ++k) {
    f->cl_off[k] = pos; int n = recs[k].n;
    /* :221-230 copy member points; :239-243 centroid = Σ(double)p / n, cast to fp32 */
    f->clusters[k] = (opoint *)malloc(n * sizeof(opoint)); double sx = 0, sy = 0, sz = 0;
    for (int j = 0; j < n; ++j) { int idx = all.d[recs[k].off + j]; f->cl_idx[pos + j] = idx; f->clusters[k][j] = f->cloud[idx];
      sx += (double)f->cloud[idx].x; sy += (double)f->cloud[idx].y; sz += (double)f->cloud[idx].z; }
    f->centroid[3 * k + 0] = (float)(sx / (double)n); f->centroid[3 * k + 1] = (float)(sy / (double)n); f->centroid[3 * k + 2] = (float)(sz / (double)n);
    pos += n;
  }
  f->cl_off[n_recs] = pos;
  free(recs); iv_free(&all); iv_free(&queue); iv_free(&nn); free(processed); kd_free(&tree);
}

/* ------------------------------------------------------------------ pose (tf) [tf]
 * tf::poseMsgToTF (:524): Quaternion(x,y,z,w) renormalised only when |len²-1| > 0.1;
 * Matrix3x3::setRotation with s = 2/len². */
static void pose_to_tf(const double p[7], double R[3][3], double o[3]) {
  double x = p[3], y = p[4], z = p[5], w = p[6];
  double l2 = x * x + y * y + z * z + w * w;
  if (fabs(l2 - 1.0) > 0.1) { double l = sqrt(l2); x /= l; y /= l; z /= l; w /= l; l2 = x * x + y * y + z * z + w * w; }
  double s = 2.0 / l2, xs = x * s, ys = y * s, zs = z * s;
  double wx = w * xs, wy = w * ys, wz = w * zs, xx = x * xs, xy = x * ys, xz = x * zs, yy = y * ys, yz = y * zs, zz = z * zs;
  R[0][0] = 1.0 - (yy + zz); R[0][1] = xy - wz; R[0][2] = xz + wy;
  R[1][0] = xy + wz; R[1][1] = 1.0 - (xx + zz); R[1][2] = yz - wx;
  R[2][0] = xz - wy; R[2][1] = yz + wx; R[2][2] = 1.0 - (xx + yy);
  o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
}
/* t = cb.ps.inverseTimes(ca.ps) (:536) = (Rbᵀ·Ra, Rbᵀ·(oa − ob)) in fp64, then cast to a
 * row-major 3x4 fp32 matrix (pcl_ros transformAsMatrix) [tf/pcl_ros] */
static void relative_transform(const frame *cb, const frame *ca, float m[12]) {
  double v[3] = { ca->o[0] - cb->o[0], ca->o[1] - cb->o[1], ca->o[2] - cb->o[2] };
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) m[4 * i + j] = (float)(cb->R[0][i] * ca->R[0][j] + cb->R[1][i] * ca->R[1][j] + cb->R[2][i] * ca->R[2][j]);
    m[4 * i + 3] = (float)(cb->R[0][i] * v[0] + cb->R[1][i] * v[1] + cb->R[2][i] * v[2]);
  }
}
/* pcl::transformPointCloud(Matrix4f): x' = ((m00·x + m01·y) + m02·z) + m03 in fp32 [PCL-1.8] */
static inline void xform3(const float m[12], float *x, float *y, float *z) {
  float a = *x, b = *y, c = *z;
  *x = ((m[0] * a + m[1] * b) + m[2] * c) + m[3];
  *y = ((m[4] * a + m[5] * b) + m[6] * c) + m[7];
  *z = ((m[8] * a + m[9] * b) + m[10] * c) + m[11];
}

/* ------------------------------------------------------------------ P2
 * MovingObjectDetectionMethods::volumeConstraint — :264-283 (getMinMax3D fp32, product in
 * fp32, ratio in fp64; NaN when both volumes are 0 ⇒ rejected) */
static int volume_constraint(const opoint *fp, int np, const opoint *fc, int nc, double threshold, int abs_int) {
  float mn[3], mx[3]; double vol[2];
  for (int w = 0; w < 2; ++w) { const opoint *c = w ? fc : fp; int n = w ? nc : np;
    for (int d = 0; d < 3; ++d) { mn[d] = FLT_MAX; mx[d] = -FLT_MAX; }
    for (int i = 0; i < n; ++i) { const float *q = &c[i].x; for (int d = 0; d < 3; ++d) { if (q[d] < mn[d]) mn[d] = q[d]; if (q[d] > mx[d]) mx[d] = q[d]; } }
    float v = (mx[0] - mn[0]) * (mx[1] - mn[1]); v = v * (mx[2] - mn[2]); vol[w] = (double)v; }
  /* :277 is an UNQUALIFIED abs(volp-volc) on doubles.  With libstdc++ >= 6 (<cstdlib>/<stdlib.h> export the floating overloads
   * into the global namespace; Ubuntu 18.04 / GCC 7, the toolchain of PCL 1.8) it is fabs — the default here.  An older
   * libstdc++ can resolve it to C's int abs(int): the difference is truncated towards zero first, so volumes less than 1 m3
   * apart always pass the gate.  volume_abs_int = 1 restates that reading (tested both ways). */
  double diff = vol[0] - vol[1];
  double ad = abs_int ? (double)abs((int)diff) : fabs(diff);
  if (fabs(ad / (vol[0] + vol[1]) - threshold) <= 1e-6) g_cen[CEN_VOLUME_GATE]++;
  return (ad / (vol[0] + vol[1])) < threshold;
}
/* calculateCorrespondenceCentroid — :285-307.  determineReciprocalCorrespondences [PCL-1.8]:
 * per source index in order: 1-NN in target, 1-NN of that in source, keep iff same index. */
static corr_list correspondence_centroid(const frame *ca, const frame *cb, double vc, int abs_int) {
  corr_list out = { NULL, 0 };
  if (ca->K == 0 || cb->K == 0) return out; /* empty kd-tree: defined as no correspondences */
  kdtree tp, tc; kd_build(&tp, ca->centroid, 3, ca->K); kd_build(&tc, cb->centroid, 3, cb->K);
  out.c = (corr_t *)malloc(ca->K * sizeof(corr_t));
  for (int i = 0; i < ca->K; ++i) {
    float d, dr; int j = kd_nn(&tc, ca->centroid + 3 * i, &d);
    int back = kd_nn(&tp, cb->centroid + 3 * j, &dr);
    if (back != i) continue;
    int np = ca->cl_off[i + 1] - ca->cl_off[i], nc = cb->cl_off[j + 1] - cb->cl_off[j];
    if (!volume_constraint(ca->clusters[i], np, cb->clusters[j], nc, vc, abs_int)) continue; /* :300 */
    out.c[out.n].query = i; out.c[out.n].match = j; out.c[out.n].dist = d; ++out.n;
  }
  kd_free(&tp); kd_free(&tc); return out;
}

/* ------------------------------------------------------------------ P3 (method 1)
 * getPointDistanceEstimateVector — :336-366.  determineCorrespondences: 1-NN per source
 * point, squared fp32 distance; count lb < d² < ub (:356, thresholds un-squared — quirk
 * kept); score = count / ((n1+n2)/2) with integer halving (:361). */
static double score_point_distance(const opoint *c1, int n1, const opoint *c2, int n2, float lb, float ub) {
  kdtree t; kd_build(&t, &c2[0].x, 8, n2);
  double count = 0;
  for (int i = 0; i < n1; ++i) { float d; kd_nn(&t, &c1[i].x, &d); if (within_ulps(d, lb, 4) || within_ulps(d, ub, 4)) g_cen[CEN_PDE_BOUND]++; if (d > lb && d < ub) count++; }
  kd_free(&t);
  return count / (double)(((size_t)n1 + (size_t)n2) / 2);
}

/* ------------------------------------------------------------------ P4 (method 2)
 * getClusterPointcloudChangeVector — :309-334, resolution 0.1f at :575.
 * OctreePointCloudChangeDetector [PCL-1.8] (recalled from octree_pointcloud.hpp:
 * adoptBoundingBoxToPoint + getKeyBitSize; SURVEY App. A omits the getKeyBitSize step):
 *   first inserted point p0: min = p0 − res/2, max = p0 + res/2 (fp64), then getKeyBitSize()
 *   forces ≥ 2 voxels per axis ⇒ depth 1, side 2·res, and — the tree being empty — re-centres
 *   the box: min −= (side − (max−min))/2 ⇒ min = p0 − res;
 *   a later point outside [min,max) adds a root level: min −= side on every axis whose UPPER
 *   bound is not violated, depth++, max = min + 2^depth·res − FLT_EPSILON;
 *   leaf key = (unsigned)((p − min)/res).
 * Every shift of min is a whole multiple of res, so the voxel lattice is min₀ + k·res for the
 * whole life of the tree; keys are re-expressed relative to min₀ by the integer shift.
 * Score = number of c2 points whose leaf holds no c1 point (getPointIndicesFromNewVoxels,
 * min_points_per_leaf = 0). */
typedef struct { long long k[3]; } vkey;
static int vkey_cmp(const void *a, const void *b) { const vkey *x = (const vkey *)a, *y = (const vkey *)b;
  for (int d = 0; d < 3; ++d) if (x->k[d] != y->k[d]) return x->k[d] < y->k[d] ? -1 : 1; return 0; }
typedef struct { double mn[3], mx[3]; long long shift[3]; int depth; int defined; double res; int anchor_half; } octbox;
static void oct_adopt(octbox *b, const float *pt) {
  const double eps = (double)FLT_EPSILON;
  for (;;) {
    int lo[3], hi[3], any = 0;
    for (int d = 0; d < 3; ++d) { lo[d] = b->defined ? ((double)pt[d] < b->mn[d]) : 1; hi[d] = b->defined ? ((double)pt[d] >= b->mx[d]) : 1; any |= lo[d] | hi[d]; }
    if (!any) return;
    if (!b->defined) {
      for (int d = 0; d < 3; ++d) { b->mn[d] = (double)pt[d] - b->res / 2; b->mx[d] = (double)pt[d] + b->res / 2; }
      /* opc_anchor = 1 (SURVEY App. A's reading): the first box stays p0 ± res/2 — one voxel, depth 0, no re-centring */
      if (b->anchor_half) { b->depth = 0; b->defined = 1; continue; }
      /* getKeyBitSize() on an empty tree */
      unsigned mk = 2;
      for (int d = 0; d < 3; ++d) { unsigned k = (unsigned)ceil((b->mx[d] - b->mn[d] - eps) / b->res); if (k > mk) mk = k; }
      b->depth = (int)ceil(log2((double)mk) - eps); if (b->depth < 0) b->depth = 0;
      double side = (double)(1LL << b->depth) * b->res;
      for (int d = 0; d < 3; ++d) { double over = (side - (b->mx[d] - b->mn[d])) / 2.0; if (over > eps) { b->mn[d] -= over; b->mx[d] += over; } }
      b->defined = 1; continue;
    }
    double side = (double)(1LL << b->depth) * b->res;
    for (int d = 0; d < 3; ++d) if (!hi[d]) { b->mn[d] -= side; b->shift[d] += (1LL << b->depth); }
    b->depth++;
    side = (double)(1LL << b->depth) * b->res - eps;
    for (int d = 0; d < 3; ++d) b->mx[d] = b->mn[d] + side;
  }
}
static double score_octree_change(const opoint *c1, int n1, const opoint *c2, int n2, float resolution, int anchor_half) {
  octbox b; memset(&b, 0, sizeof b); b.res = (double)resolution; b.anchor_half = anchor_half;
  vkey *k1 = (vkey *)malloc((n1 ? n1 : 1) * sizeof(vkey));
  for (int i = 0; i < n1; ++i) { oct_adopt(&b, &c1[i].x);
    for (int d = 0; d < 3; ++d) k1[i].k[d] = (long long)(unsigned)(((double)(&c1[i].x)[d] - b.mn[d]) / b.res) - b.shift[d]; }
  qsort(k1, n1, sizeof(vkey), vkey_cmp);
  int changed = 0;
  for (int i = 0; i < n2; ++i) { oct_adopt(&b, &c2[i].x); vkey k;
    for (int d = 0; d < 3; ++d) k.k[d] = (long long)(unsigned)(((double)(&c2[i].x)[d] - b.mn[d]) / b.res) - b.shift[d];
    for (int d = 0; d < 3; ++d) { const float v = (&c2[i].x)[d], lo = nextafterf(v, -INFINITY), hi = nextafterf(v, INFINITY);
      if ((long long)(unsigned)(((double)lo - b.mn[d]) / b.res) - b.shift[d] != k.k[d] || (long long)(unsigned)(((double)hi - b.mn[d]) / b.res) - b.shift[d] != k.k[d]) { g_cen[CEN_VOXEL_FACE]++; break; } }
    if (!bsearch(&k, k1, n1, sizeof(vkey), vkey_cmp)) ++changed; }
  free(k1);
  return (double)changed;
}

/* ------------------------------------------------------------------ T1
 * recurseFindClusterChain — :415-453 */
static int recurse_find_chain(const oracle_ctx *c, int col, int track) {
  if (col == c->n_corrs_vec) return track;
  const corr_list *m = &c->corrs_vec[col];
  for (int j = 0; j < m->n; ++j) {
    if (m->c[j].query == track) {
      if (c->res_vec[col + 1].v[m->c[j].match]) return recurse_find_chain(c, col + 1, m->c[j].match);
      return -1;
    }
  }
  return -1;
}
/* pushCentroid — :455-476 (true Euclidean distance in fp64 via sqrt(pow+pow+pow)) */
static void push_centroid(oracle_ctx *c, const float pt[3]) {
  for (int i = 0; i < c->n_mo; ++i) {
    double dx = pt[0] - c->mo[i].c[0], dy = pt[1] - c->mo[i].c[1], dz = pt[2] - c->mo[i].c[2]; /* float - float → float, then pow(double) */
    double dist = sqrt(dx * dx + dy * dy + dz * dz);
    if (dist < (double)c->p.catch_up_distance) return;
  }
  if (c->n_mo == c->cap_mo) { c->cap_mo = c->cap_mo ? 2 * c->cap_mo : 16; c->mo = (mo_centroid *)realloc(c->mo, c->cap_mo * sizeof(mo_centroid)); }
  mo_centroid *m = &c->mo[c->n_mo++]; memcpy(m->c, pt, 3 * sizeof(float));
  m->confidence = c->static_confidence + 1; m->max_confidence = c->static_confidence + 1; /* header :91 */
}
/* checkMovingClusterChain — :478-514 */
static void check_moving_cluster_chain(oracle_ctx *c, corr_list mp, const frame *ca, const frame *cb) {
  c->corrs_vec = (corr_list *)realloc(c->corrs_vec, (c->n_corrs_vec + 1) * sizeof(corr_list));
  corr_list cp = { (corr_t *)malloc((mp.n ? mp.n : 1) * sizeof(corr_t)), mp.n }; if (mp.n) memcpy(cp.c, mp.c, mp.n * sizeof(corr_t));
  c->corrs_vec[c->n_corrs_vec++] = cp;
  int add = (c->n_res_vec == 0) ? 2 : 1;
  c->res_vec = (bool_vec *)realloc(c->res_vec, (c->n_res_vec + add) * sizeof(bool_vec));
  if (c->n_res_vec == 0) { bool_vec v = { (unsigned char *)malloc(ca->K ? ca->K : 1), ca->K }; if (ca->K) memcpy(v.v, ca->det, ca->K); c->res_vec[c->n_res_vec++] = v; } /* :484-488 */
  { bool_vec v = { (unsigned char *)malloc(cb->K ? cb->K : 1), cb->K }; if (cb->K) memcpy(v.v, cb->det, cb->K); c->res_vec[c->n_res_vec++] = v; } /* :490 */
  if (c->n_res_vec >= c->moving_confidence) { /* :492 */
    for (int i = 0; i < c->res_vec[0].n; ++i) {
      if (c->res_vec[0].v[i]) { int found = recurse_find_chain(c, 0, i); if (found != -1) push_centroid(c, cb->centroid + 3 * found); } /* :501-508 */
    }
    free(c->corrs_vec[0].c); memmove(c->corrs_vec, c->corrs_vec + 1, (c->n_corrs_vec - 1) * sizeof(corr_list)); c->n_corrs_vec--; /* :511 */
    free(c->res_vec[0].v); memmove(c->res_vec, c->res_vec + 1, (c->n_res_vec - 1) * sizeof(bool_vec)); c->n_res_vec--;       /* :512 */
  }
}

/* ------------------------------------------------------------------ pushRawCloudAndPose — :516-611 */
int oracle_push(oracle_ctx *c, const void *data, uint64_t n_points, uint32_t point_step, uint32_t off_x, uint32_t off_y,
                uint32_t off_z, uint32_t off_i, const double pose[7]) {
  double t0 = now_s();
  frame_free(c->ca); c->ca = c->cb; c->cb = frame_new(); /* :520-521 */
  frame *cb = c->cb, *ca = c->ca;
  /* :523 fromPCLPointCloud2: named float32 fields → PointXYZI; missing intensity stays 0 [PCL-1.8] */
  cb->n_in = n_points; cb->n_raw = n_points; cb->raw = (opoint *)malloc((n_points ? n_points : 1) * sizeof(opoint));
  const unsigned char *src = (const unsigned char *)data;
  for (uint64_t i = 0; i < n_points; ++i) { opoint q; memset(&q, 0, sizeof q); q.w = 1.0f; const unsigned char *r = src + i * point_step;
    memcpy(&q.x, r + off_x, 4); memcpy(&q.y, r + off_y, 4); memcpy(&q.z, r + off_z, 4); if (off_i != 0xFFFFFFFFu) memcpy(&q.intensity, r + off_i, 4);
    cb->raw[i] = q; }
  pose_to_tf(pose, cb->R, cb->o); /* :524 */
  if (c->p.ground_method == 1) ground_removal_voxel(cb, &c->p); else ground_removal_crop(cb, &c->p); /* :526 / :527 */
  compute_clusters(cb, &c->p); /* :529 */
  cb->init = 1;                /* :532 */
  free(c->last_corr); free(c->last_score); c->last_corr = NULL; c->last_score = NULL; c->n_last_corr = 0;
  if (ca->init && cb->init) {  /* :534 */
    float m[12]; relative_transform(cb, ca, m); /* :536 */
    for (int k = 0; k < ca->K; ++k) xform3(m, &ca->centroid[3 * k], &ca->centroid[3 * k + 1], &ca->centroid[3 * k + 2]); /* :540-541 */
    for (int k = 0; k < ca->K; ++k) { int n = ca->cl_off[k + 1] - ca->cl_off[k]; for (int j = 0; j < n; ++j) { opoint *q = &ca->clusters[k][j]; xform3(m, &q->x, &q->y, &q->z); } } /* :544-551 */
    corr_list mp = correspondence_centroid(ca, cb, (double)c->p.volume_constraint, c->p.volume_abs_int); /* :564 */
    double *score = (double *)malloc((mp.n ? mp.n : 1) * sizeof(double));
    for (int j = 0; j < mp.n; ++j) {
      int q = mp.c[j].query, mt = mp.c[j].match; int n1 = ca->cl_off[q + 1] - ca->cl_off[q], n2 = cb->cl_off[mt + 1] - cb->cl_off[mt];
      if (c->p.method_choice == 1) score[j] = score_point_distance(ca->clusters[q], n1, cb->clusters[mt], n2, c->p.pde_lb, c->p.pde_ub); /* :571 */
      else if (c->p.method_choice == 2) score[j] = score_octree_change(ca->clusters[q], n1, cb->clusters[mt], n2, c->p.opc_resolution, c->p.opc_anchor != 0); /* :575 */
      else score[j] = 0; /* reference: param_vec stays empty ⇒ UB; defined as 0 */
      double threshold = 0;
      if (c->p.method_choice == 1) threshold = (double)c->p.pde_distance_threshold; /* :586 */
      else if (c->p.method_choice == 2) threshold = (double)(((size_t)n1 + (size_t)n2) / (size_t)c->p.opc_normalization_factor); /* :590 integer division */
      cb->det[mt] = score[j] > threshold; /* :593-604 */
    }
    c->last_corr = mp.c; c->last_score = score; c->n_last_corr = mp.n;
    check_moving_cluster_chain(c, mp, ca, cb); /* :608 */
  }
  c->busy += now_s() - t0;
  return 0;
}

/* ------------------------------------------------------------------ filterCloud — :613-696 */
int oracle_filter(oracle_ctx *c, float *out, uint64_t *n_out) {
  double t0 = now_s();
  frame *cb = c->cb; if (!cb->init) { *n_out = 0; return -1; }
  kdtree tree; kd_build(&tree, cb->centroid, 3, cb->K); /* :618 */
  ivec moving = { 0 };
  c->marked.n = 0;
  for (int i = 0; i < c->n_mo; ++i) { /* :630 */
    float d; int nn = kd_nn(&tree, c->mo[i].c, &d); /* :636 */
    if (nn < 0) continue; /* empty centroid set: nearestKSearch returns 0 (defined) */
    iv_push(&c->marked, nn); /* :641 marker_pub.publish(mark_cluster(cb->clusters[nn], id, …)); id++ at :669 */
    for (int j = cb->cl_off[nn]; j < cb->cl_off[nn + 1]; ++j) iv_push(&moving, cb->cl_idx[j]); /* :644-648, before the distance test */
    if (!cb->det[nn] || d > c->p.leave_off_distance) { /* :650 squared vs un-squared: quirk kept */
      if (--c->mo[i].confidence == 0) { memmove(&c->mo[i], &c->mo[i + 1], (c->n_mo - i - 1) * sizeof(mo_centroid)); c->n_mo--; i--; } /* :655-660 */
    } else {
      memcpy(c->mo[i].c, cb->centroid + 3 * nn, 3 * sizeof(float)); /* :664 */
      if (c->mo[i].confidence < c->mo[i].max_confidence) c->mo[i].confidence++; /* :667 */
    }
  }
  /* :673-678 ExtractIndices(negative) [PCL-1.8]: more indices than points ⇒ error, empty output */
  uint64_t n = 0;
  if (moving.n <= cb->n_cloud) {
    unsigned char *rm = (unsigned char *)calloc(cb->n_cloud ? cb->n_cloud : 1, 1);
    for (size_t j = 0; j < moving.n; ++j) rm[moving.d[j]] = 1;
    for (size_t i = 0; i < cb->n_cloud; ++i) if (!rm[i]) { const opoint *q = &cb->cloud[i]; out[4 * n] = q->x; out[4 * n + 1] = q->y; out[4 * n + 2] = q->z; out[4 * n + 3] = q->intensity; ++n; }
    free(rm);
  }
  for (size_t i = 0; i < cb->n_gp; ++i) { const opoint *q = &cb->raw[cb->gp[i]]; out[4 * n] = q->x; out[4 * n + 1] = q->y; out[4 * n + 2] = q->z; out[4 * n + 3] = q->intensity; ++n; } /* :681-684 */
  *n_out = n;
  iv_free(&moving); kd_free(&tree);
  c->busy += now_s() - t0;
  return 0;
}

/* ------------------------------------------------------------------ read-backs */
void oracle_get_counts(const oracle_ctx *c, oracle_counts *o) {
  const frame *f = c->cb; memset(o, 0, sizeof *o);
  o->n_in = f->n_in; o->n_trim = f->n_raw; o->n_cloud = f->n_cloud; o->n_ground = f->n_gp;
  o->n_clusters = (uint32_t)f->K; o->n_clustered = f->cl_off ? (uint32_t)f->cl_off[f->K] : 0;
  o->n_corr = (uint32_t)c->n_last_corr; o->n_tracks = (uint32_t)c->n_mo;
}
uint32_t oracle_get_moving_clusters(const oracle_ctx *c, int32_t *cluster_of_track) {
  if (cluster_of_track) for (size_t i = 0; i < c->marked.n; ++i) cluster_of_track[i] = c->marked.d[i];
  return (uint32_t)c->marked.n;
}
void oracle_get_labels(const oracle_ctx *c, int32_t *lab) {
  const frame *f = c->cb;
  for (size_t i = 0; i < f->n_raw; ++i) lab[i] = -2;
  for (size_t i = 0; i < f->n_cloud; ++i) lab[f->cloud_src[i]] = -1;
  for (int k = 0; k < f->K; ++k) for (int j = f->cl_off[k]; j < f->cl_off[k + 1]; ++j) lab[f->cloud_src[f->cl_idx[j]]] = k;
}
void oracle_get_ground_indices(const oracle_ctx *c, int32_t *idx) { const frame *f = c->cb; for (size_t i = 0; i < f->n_gp; ++i) idx[i] = f->gp[i]; }
void oracle_get_clusters(const oracle_ctx *c, int32_t *off, int32_t *idx) {
  const frame *f = c->cb; if (!f->cl_off) { off[0] = 0; return; }
  for (int k = 0; k <= f->K; ++k) off[k] = f->cl_off[k];
  for (int j = 0; j < f->cl_off[f->K]; ++j) idx[j] = f->cl_idx[j];
}
void oracle_get_centroids(const oracle_ctx *c, float *xyz) { const frame *f = c->cb; if (f->K) memcpy(xyz, f->centroid, (size_t)f->K * 3 * sizeof(float)); }
void oracle_get_detection(const oracle_ctx *c, uint8_t *det) { const frame *f = c->cb; if (f->K) memcpy(det, f->det, f->K); }
void oracle_get_correspondences(const oracle_ctx *c, int32_t *q, int32_t *m, float *d, double *s) {
  for (int j = 0; j < c->n_last_corr; ++j) { q[j] = c->last_corr[j].query; m[j] = c->last_corr[j].match; d[j] = c->last_corr[j].dist; s[j] = c->last_score[j]; }
}
void oracle_get_tracks(const oracle_ctx *c, float *xyz, int32_t *conf, int32_t *maxc) {
  for (int i = 0; i < c->n_mo; ++i) { memcpy(xyz + 3 * i, c->mo[i].c, 3 * sizeof(float)); conf[i] = c->mo[i].confidence; maxc[i] = c->mo[i].max_confidence; }
}
uint32_t oracle_get_prev_cluster_count(const oracle_ctx *c) { return (uint32_t)c->ca->K; }
/* P1 read-back: `ca` after the in-place transform of :540-551 — centroids (K_prev × 3) and the cluster points in cluster
 * order (C_prev × xyzi), as pcl_ros::transformPointCloud left them */
void oracle_get_prev_transformed(const oracle_ctx *c, float *cent_K3, float *pts_C4) {
  const frame *f = c->ca; if (!f || !f->init) return;
  if (cent_K3 && f->K) memcpy(cent_K3, f->centroid, (size_t)f->K * 3 * sizeof(float));
  if (pts_C4) { size_t o = 0; for (int k = 0; k < f->K; ++k) { int n = f->cl_off[k + 1] - f->cl_off[k]; for (int j = 0; j < n; ++j, ++o) { const opoint *q = &f->clusters[k][j]; pts_C4[4 * o] = q->x; pts_C4[4 * o + 1] = q->y; pts_C4[4 * o + 2] = q->z; pts_C4[4 * o + 3] = q->intensity; } } }
}
uint32_t oracle_get_prev_clustered(const oracle_ctx *c) { const frame *f = c->ca; return (f && f->init && f->K) ? (uint32_t)f->cl_off[f->K] : 0u; }
/* mark_cluster (:7-58) for every cluster of `cb`: position = compute3DCentroid into an Eigen::Vector4f — a sequential FLOAT
 * sum over the cluster's points in order, divided by n (:15) —, scale = getMinMax3D extent (:16, :36-38), zero extents
 * replaced by 0.1 (:40-47).  (The reference builds the marker only for tracked clusters inside filterCloud, :641.) */
void oracle_get_markers(const oracle_ctx *c, float *pos_K3, float *scale_K3) {
  const frame *f = c->cb;
  for (int k = 0; k < f->K; ++k) {
    int n = f->cl_off[k + 1] - f->cl_off[k]; float sx = 0.f, sy = 0.f, sz = 0.f, mn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, mx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (int j = 0; j < n; ++j) { const float *q = &f->clusters[k][j].x; sx += q[0]; sy += q[1]; sz += q[2]; for (int d = 0; d < 3; ++d) { if (q[d] < mn[d]) mn[d] = q[d]; if (q[d] > mx[d]) mx[d] = q[d]; } }
    pos_K3[3 * k] = sx / (float)n; pos_K3[3 * k + 1] = sy / (float)n; pos_K3[3 * k + 2] = sz / (float)n;
    for (int d = 0; d < 3; ++d) { float e = mx[d] - mn[d]; scale_K3[3 * k + d] = e == 0.f ? 0.1f : e; }
  }
}
double oracle_get_busy_seconds(const oracle_ctx *c) { return c->busy; }
