// mor_tracker.h — host-side temporal logic of the hot path (SURVEY.md §8a rows T1 and the
// tracking loop of F1).  O(clusters) per frame, strictly sequential, stays on the CPU exactly as
// in the reference.  Follows /root/reference/src/MovingObjectRemoval.cpp:415-514 and :630-671,
// struct MovingObjectCentroid at include/MOR/MovingObjectRemoval.h:83-94.
#pragma once
#include <cstdint>
#include <deque>
#include <vector>
#include "../../include/mor_hip.h"

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
