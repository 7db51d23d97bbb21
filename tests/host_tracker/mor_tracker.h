// mor_tracker.h — the temporal logic of the hot path (SURVEY.md §8a rows T1 and the tracking loop of F1) as plain host
// C++.  TEST-ONLY: the product runs this logic on the device (k_track_push / k_track_filter in csrc/mor_kernels.hip);
// this second, independent statement of the same rules lets CPU-only tests drive scripted cluster / score sequences
// (tests/test_tracker_host.py) and is not part of libmor_hip.so.  Follows /root/reference/src/MovingObjectRemoval.cpp:415-514 and :630-671,
// struct MovingObjectCentroid at include/MOR/MovingObjectRemoval.h:83-94.
#pragma once
#include <cstdint>
#include <deque>
#include <vector>
#include "../../include/mor_hip.h"   // mor_params, error codes

struct MorCorr { int32_t query, match; };

struct MorMovingCentroid {   // MovingObjectCentroid, header :83-94
  float c[3];
  int confidence, max_confidence;
};

struct mor_tracker {
  mor_params p;
  int moving_confidence, static_confidence;            // n_bad, n_good (:368)
  std::deque<std::vector<MorCorr>> corrs_vec;           // header :112
  std::deque<std::vector<uint8_t>> res_vec;             // header :115
  std::vector<MorMovingCentroid> mo_vec;                // header :109
  // latest frame summary (cb) and the previous frame's detection results (ca)
  std::vector<float> cur_centroids;                     // K×3
  std::vector<uint8_t> cur_det, prev_det;
  bool has_cur = false;

  mor_tracker(const mor_params &pp, int n_bad, int n_good) : p(pp), moving_confidence(n_bad), static_confidence(n_good) {}

  // rotate cb→ca, store the new frame; when n_pairs >= 0 run checkMovingClusterChain (:608)
  void push(int K, const float *centroids, const uint8_t *det, int n_pairs, const int32_t *query, const int32_t *match);
  // filterCloud's loop over mo_vec (:630-671)
  void filter(const int32_t *cluster_sizes, uint8_t *moving, uint64_t *n_moving_idx);

 private:
  int recurse_find_cluster_chain(int col, int track) const;   // :415-453
  void push_centroid(const float *pt);                        // :455-476
};

extern "C" {
mor_tracker *mor_tracker_create(const mor_params *p, int n_bad, int n_good);
void mor_tracker_destroy(mor_tracker *t);
/* feed one frame's cluster summary: K centroids, detection flags, and the correspondence pairs to the previous frame
 * (n_pairs < 0: first frame, no pair stage) — runs checkMovingClusterChain */
int mor_tracker_push(mor_tracker *t, int K, const float *centroids_K3, const uint8_t *det_K, int n_pairs, const int32_t *query, const int32_t *match);
/* filterCloud's tracking loop (:630-671): fills moving_K (1 = cluster removed) and *n_moving_idx = total indices pushed incl. duplicates */
int mor_tracker_filter(mor_tracker *t, const int32_t *cluster_sizes_K, uint8_t *moving_K, uint64_t *n_moving_idx);
int mor_tracker_get(const mor_tracker *t, float *xyz_n3, int32_t *conf, int32_t *max_conf, int max_n);
}
