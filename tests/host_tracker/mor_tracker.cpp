// mor_tracker.cpp — see mor_tracker.h.  Plain host C++, no GPU dependency.
#include "mor_tracker.h"
#include <cmath>
#include <cstring>
#include <limits>

int mor_tracker::recurse_find_cluster_chain(int col, int track) const {
  // walk the correspondence maps oldest → newest; every hop must land on a cluster flagged moving
  if (col == (int)corrs_vec.size()) return track;
  for (const MorCorr &c : corrs_vec[col]) {
    if (c.query != track) continue;
    return res_vec[col + 1][c.match] ? recurse_find_cluster_chain(col + 1, c.match) : -1;
  }
  return -1;
}

void mor_tracker::push_centroid(const float *pt) {
  for (const MorMovingCentroid &m : mo_vec) {
    // reference: sqrt(pow(float-float,2)+…) — fp32 differences, fp64 norm (:462)
    double dx = (double)(pt[0] - m.c[0]), dy = (double)(pt[1] - m.c[1]), dz = (double)(pt[2] - m.c[2]);
    if (std::sqrt(dx * dx + dy * dy + dz * dz) < (double)p.catch_up_distance) return;
  }
  MorMovingCentroid m;
  std::memcpy(m.c, pt, sizeof m.c);
  m.confidence = m.max_confidence = static_confidence + 1;   // header :91
  mo_vec.push_back(m);
}

void mor_tracker::push(int K, const float *centroids, const uint8_t *det, int n_pairs, const int32_t *query, const int32_t *match) {
  prev_det.swap(cur_det);
  cur_det.assign(det, det + K);
  cur_centroids.assign(centroids, centroids + 3 * (size_t)K);
  bool had_prev = has_cur;
  has_cur = true;
  if (n_pairs < 0 || !had_prev) return;   // first frame: ca->init is false (:534)
  // checkMovingClusterChain (:478-514)
  std::vector<MorCorr> mp((size_t)n_pairs);
  for (int j = 0; j < n_pairs; ++j) { mp[j].query = query[j]; mp[j].match = match[j]; }
  corrs_vec.push_back(std::move(mp));
  if (res_vec.empty()) res_vec.push_back(prev_det);
  res_vec.push_back(cur_det);
  if ((int)res_vec.size() >= moving_confidence) {
    const std::vector<uint8_t> &oldest = res_vec[0];
    for (int i = 0; i < (int)oldest.size(); ++i) {
      if (!oldest[i]) continue;
      int found = recurse_find_cluster_chain(0, i);
      if (found != -1) push_centroid(&cur_centroids[3 * (size_t)found]);
    }
    corrs_vec.pop_front();
    res_vec.pop_front();
  }
}

void mor_tracker::filter(const int32_t *cluster_sizes, uint8_t *moving, uint64_t *n_moving_idx) {
  const int K = (int)cur_det.size();
  std::memset(moving, 0, (size_t)K);
  uint64_t total = 0;
  for (int i = 0; i < (int)mo_vec.size(); ++i) {
    if (K == 0) continue;   // empty centroid set: nearestKSearch finds nothing (defined; the reference is UB here)
    // 1-NN among the current centroids, squared fp32 distance, ties → lowest index (:636)
    int best = -1; float bd = std::numeric_limits<float>::infinity();
    for (int k = 0; k < K; ++k) {
      const float *c = &cur_centroids[3 * (size_t)k];
      float d0 = mo_vec[i].c[0] - c[0], d1 = mo_vec[i].c[1] - c[1], d2 = mo_vec[i].c[2] - c[2];
      float d = d0 * d0; d = d + d1 * d1; d = d + d2 * d2;
      if (d < bd) { bd = d; best = k; }
    }
    moving[best] = 1;                       // whole cluster queued for removal before any test (:644-648)
    total += (uint64_t)cluster_sizes[best];
    if (!cur_det[best] || bd > p.leave_off_distance) {   // squared vs un-squared: reference quirk kept (:650)
      if (--mo_vec[i].confidence == 0) { mo_vec.erase(mo_vec.begin() + i); --i; }
    } else {
      std::memcpy(mo_vec[i].c, &cur_centroids[3 * (size_t)best], sizeof mo_vec[i].c);   // :664
      if (mo_vec[i].confidence < mo_vec[i].max_confidence) ++mo_vec[i].confidence;       // :667
    }
  }
  *n_moving_idx = total;
}

extern "C" {
mor_tracker *mor_tracker_create(const mor_params *p, int n_bad, int n_good) { return p ? new mor_tracker(*p, n_bad, n_good) : nullptr; }
void mor_tracker_destroy(mor_tracker *t) { delete t; }
int mor_tracker_push(mor_tracker *t, int K, const float *c, const uint8_t *det, int n_pairs, const int32_t *q, const int32_t *m) {
  if (!t || K < 0) return MOR_ERR_INVALID;
  t->push(K, c, det, n_pairs, q, m);
  return MOR_OK;
}
int mor_tracker_filter(mor_tracker *t, const int32_t *sizes, uint8_t *moving, uint64_t *n_idx) {
  if (!t || !t->has_cur) return MOR_ERR_NOT_READY;
  t->filter(sizes, moving, n_idx);
  return MOR_OK;
}
int mor_tracker_get(const mor_tracker *t, float *xyz, int32_t *conf, int32_t *maxc, int max_n) {
  if (!t) return MOR_ERR_INVALID;
  int n = (int)t->mo_vec.size();
  for (int i = 0; i < n && i < max_n; ++i) {
    if (xyz) std::memcpy(xyz + 3 * i, t->mo_vec[i].c, 3 * sizeof(float));
    if (conf) conf[i] = t->mo_vec[i].confidence;
    if (maxc) maxc[i] = t->mo_vec[i].max_confidence;
  }
  return n;
}
}
