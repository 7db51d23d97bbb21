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

namespace std_msgs { struct Header { uint32_t seq = 0; double stamp = 0; std::string frame_id; }; }
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
