// ros_pcl_types.h — minimal stand-ins for the few ROS / PCL message types that appear in the
// signature of MovingObjectRemoval (reference include/MOR/MovingObjectRemoval.h:158-167), used ONLY
// when the adapter is built without ROS/PCL (this image has neither).  Field names and meanings
// follow the real message definitions so code written against them compiles unchanged against
// <pcl/PCLPointCloud2.h>, <sensor_msgs/PointCloud2.h>, <geometry_msgs/Pose.h> when
// MOR_WITH_ROS_PCL is defined.  These are data carriers for the adapter and its tests — they are
// not used to build the reference.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace pcl {
struct PCLHeader { uint32_t seq = 0; uint64_t stamp = 0; std::string frame_id; };
struct PCLPointField {
  std::string name; uint32_t offset = 0; uint8_t datatype = 0; uint32_t count = 0;
  enum PointFieldTypes { INT8 = 1, UINT8 = 2, INT16 = 3, UINT16 = 4, INT32 = 5, UINT32 = 6, FLOAT32 = 7, FLOAT64 = 8 };
};
struct PCLPointCloud2 {
  PCLHeader header; uint32_t height = 0, width = 0; std::vector<PCLPointField> fields;
  uint8_t is_bigendian = 0; uint32_t point_step = 0, row_step = 0; std::vector<uint8_t> data; uint8_t is_dense = 0;
};
}  // namespace pcl

// ros::Time as far as the adapter and its callers use it: integer seconds + nanoseconds, no implicit conversion from a number
// (the real constructor from double is explicit), fromNSec / toNSec / toSec, and the stream form "sec.nnnnnnnnn"
namespace ros {
struct Time {
  uint32_t sec = 0, nsec = 0;
  Time() {}
  Time(uint32_t s, uint32_t ns) : sec(s), nsec(ns) {}
  explicit Time(double t) : sec((uint32_t)t), nsec((uint32_t)((t - (double)(uint32_t)t) * 1e9)) {}
  Time &fromNSec(uint64_t t) { sec = (uint32_t)(t / 1000000000ull); nsec = (uint32_t)(t % 1000000000ull); return *this; }
  uint64_t toNSec() const { return (uint64_t)sec * 1000000000ull + nsec; }
  double toSec() const { return (double)sec + 1e-9 * (double)nsec; }
  bool operator==(const Time &o) const { return sec == o.sec && nsec == o.nsec; }
};
template <class OS> OS &operator<<(OS &os, const Time &t) {
  char b[32]; int n = 0; uint32_t ns = t.nsec;
  for (int i = 8; i >= 0; --i) { b[i] = (char)('0' + ns % 10); ns /= 10; ++n; }
  b[9] = 0; (void)n;
  os << t.sec << "." << b;
  return os;
}
}  // namespace ros
namespace std_msgs { struct Header { uint32_t seq = 0; ros::Time stamp; std::string frame_id; }; }
// pcl_conversions::fromPCL(const pcl::PCLHeader &, std_msgs::Header &): pcl stamps are microseconds since the epoch
namespace pcl_conversions {
inline void fromPCL(const pcl::PCLHeader &pcl_header, std_msgs::Header &header) {
  header.stamp.fromNSec(pcl_header.stamp * 1000ull); header.seq = pcl_header.seq; header.frame_id = pcl_header.frame_id;
}
}  // namespace pcl_conversions
namespace geometry_msgs {
struct Point { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Pose { Point position; Quaternion orientation; };
}  // namespace geometry_msgs
namespace sensor_msgs {
struct PointField { std::string name; uint32_t offset = 0; uint8_t datatype = 0; uint32_t count = 0; enum { FLOAT32 = 7 }; };
struct PointCloud2 {
  std_msgs::Header header; uint32_t height = 0, width = 0; std::vector<PointField> fields;
  uint8_t is_bigendian = 0; uint32_t point_step = 0, row_step = 0; std::vector<uint8_t> data; uint8_t is_dense = 0;
};
}  // namespace sensor_msgs
namespace ros { class NodeHandle {}; }
