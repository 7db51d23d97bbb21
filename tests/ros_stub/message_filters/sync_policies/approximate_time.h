#pragma once
#include <cstddef>
#include <cstdint>
namespace message_filters { namespace sync_policies {
template <class A, class B> struct ApproximateTime {
  typedef A M0; typedef B M1;
  explicit ApproximateTime(uint32_t queue_size_) : queue_size(queue_size_) {}
  size_t queue_size;
};
} }  // namespace message_filters::sync_policies
