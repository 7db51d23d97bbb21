#pragma once
// message_filters::Synchronizer<Policy>: constructor (policy, f0, f1), registerCallback(&Class::method, obj) — the call forms the
// reference uses (src/MovingObjectRemoval.cpp:379-385).  The stand-in pairs the k-th message of one input with the k-th of the other
// (what ApproximateTime yields for two streams published in lock step); it is NOT the approximate-time algorithm.
#include <deque>
#include <functional>
#include <message_filters/subscriber.h>
namespace message_filters {
template <class Policy> class Synchronizer {
 public:
  typedef typename Policy::M0 M0; typedef typename Policy::M1 M1;
  typedef boost::shared_ptr<M0 const> M0ConstPtr; typedef boost::shared_ptr<M1 const> M1ConstPtr;
  template <class F0, class F1> Synchronizer(const Policy &policy, F0 &f0, F1 &f1) : policy_(policy) {
    f0.registerCallback([this](const M0ConstPtr &m) { q0_.push_back(m); fire(); });
    f1.registerCallback([this](const M1ConstPtr &m) { q1_.push_back(m); fire(); });
  }
  template <class C> void registerCallback(void (C::*fp)(const M0ConstPtr &, const M1ConstPtr &), C *obj) {
    cb_ = [fp, obj](const M0ConstPtr &a, const M1ConstPtr &b) { (obj->*fp)(a, b); };
  }
 private:
  void fire() {
    while (!q0_.empty() && !q1_.empty()) { M0ConstPtr a = q0_.front(); M1ConstPtr b = q1_.front(); q0_.pop_front(); q1_.pop_front(); if (cb_) cb_(a, b); }
    while (q0_.size() > policy_.queue_size) q0_.pop_front();
    while (q1_.size() > policy_.queue_size) q1_.pop_front();
  }
  Policy policy_;
  std::deque<M0ConstPtr> q0_; std::deque<M1ConstPtr> q1_;
  std::function<void(const M0ConstPtr &, const M1ConstPtr &)> cb_;
};
}  // namespace message_filters
