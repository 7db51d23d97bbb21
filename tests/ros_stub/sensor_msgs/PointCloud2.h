#pragma once
#include <vector>
#include <boost/shared_ptr.hpp>
#include <sensor_msgs/PointField.h>
#include <std_msgs/Header.h>
namespace sensor_msgs {
template <class A> struct PointCloud2_ {
  PointCloud2_() : header(), height(0), width(0), fields(), is_bigendian(false), point_step(0), row_step(0), data(), is_dense(false) {}
  ::std_msgs::Header_<A> header;
  uint32_t height, width;
  std::vector<::sensor_msgs::PointField_<A>> fields;
  uint8_t is_bigendian;
  uint32_t point_step, row_step;
  std::vector<uint8_t> data;
  uint8_t is_dense;
};
typedef PointCloud2_<std::allocator<void>> PointCloud2;
typedef boost::shared_ptr<PointCloud2> PointCloud2Ptr;
typedef boost::shared_ptr<PointCloud2 const> PointCloud2ConstPtr;
}  // namespace sensor_msgs
