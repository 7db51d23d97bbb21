#pragma once
#include <pcl/PCLPointCloud2.h>
#include <sensor_msgs/PointCloud2.h>
#include <std_msgs/Header.h>
namespace pcl_conversions {
inline void fromPCL(const pcl::uint64_t &pcl_stamp, ros::Time &stamp) { stamp.fromNSec(pcl_stamp * 1000ull); }
inline void fromPCL(const pcl::PCLHeader &pcl_header, std_msgs::Header &header) {
  fromPCL(pcl_header.stamp, header.stamp); header.seq = pcl_header.seq; header.frame_id = pcl_header.frame_id;
}
inline void toPCL(const ros::Time &stamp, pcl::uint64_t &pcl_stamp) { pcl_stamp = stamp.toNSec() / 1000ull; }
inline void toPCL(const std_msgs::Header &header, pcl::PCLHeader &pcl_header) { toPCL(header.stamp, pcl_header.stamp); pcl_header.seq = header.seq; pcl_header.frame_id = header.frame_id; }
inline void toPCL(const sensor_msgs::PointField &pf, pcl::PCLPointField &pcl_pf) { pcl_pf.name = pf.name; pcl_pf.offset = pf.offset; pcl_pf.datatype = pf.datatype; pcl_pf.count = pf.count; }
inline void toPCL(const sensor_msgs::PointCloud2 &pc2, pcl::PCLPointCloud2 &pcl_pc2) {   // header, layout, fields and a copy of the data
  toPCL(pc2.header, pcl_pc2.header);
  pcl_pc2.height = pc2.height; pcl_pc2.width = pc2.width; pcl_pc2.fields.resize(pc2.fields.size());
  for (size_t i = 0; i < pc2.fields.size(); ++i) toPCL(pc2.fields[i], pcl_pc2.fields[i]);
  pcl_pc2.is_bigendian = pc2.is_bigendian; pcl_pc2.point_step = pc2.point_step; pcl_pc2.row_step = pc2.row_step; pcl_pc2.is_dense = pc2.is_dense;
  pcl_pc2.data = pc2.data;
}
}  // namespace pcl_conversions
