#pragma once
#include <cstdint>
#include <string>
#include <vector>
namespace pcl {
typedef ::uint8_t uint8_t; typedef ::uint32_t uint32_t; typedef ::uint64_t uint64_t;
struct PCLHeader {
  PCLHeader() : seq(0), stamp(), frame_id() {}
  pcl::uint32_t seq; pcl::uint64_t stamp; std::string frame_id;
};
struct PCLPointField {
  PCLPointField() : name(), offset(0), datatype(0), count(0) {}
  std::string name; pcl::uint32_t offset; pcl::uint8_t datatype; pcl::uint32_t count;
  enum PointFieldTypes { INT8 = 1, UINT8 = 2, INT16 = 3, UINT16 = 4, INT32 = 5, UINT32 = 6, FLOAT32 = 7, FLOAT64 = 8 };
};
struct PCLPointCloud2 {
  PCLPointCloud2() : header(), height(0), width(0), fields(), is_bigendian(false), point_step(0), row_step(0), data(), is_dense(false) {}
  ::pcl::PCLHeader header;
  pcl::uint32_t height, width;
  std::vector<::pcl::PCLPointField> fields;
  pcl::uint8_t is_bigendian;
  pcl::uint32_t point_step, row_step;
  std::vector<pcl::uint8_t> data;
  pcl::uint8_t is_dense;
};
}  // namespace pcl
