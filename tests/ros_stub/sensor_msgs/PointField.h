#pragma once
#include <cstdint>
#include <memory>
#include <string>
namespace sensor_msgs {
template <class A> struct PointField_ {
  PointField_() : name(), offset(0), datatype(0), count(0) {}
  std::string name; uint32_t offset; uint8_t datatype; uint32_t count;
  enum { INT8 = 1u, UINT8 = 2u, INT16 = 3u, UINT16 = 4u, INT32 = 5u, UINT32 = 6u, FLOAT32 = 7u, FLOAT64 = 8u };
};
typedef PointField_<std::allocator<void>> PointField;
}  // namespace sensor_msgs
