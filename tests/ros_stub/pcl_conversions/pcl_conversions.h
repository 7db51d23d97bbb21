#pragma once
#include <pcl/PCLPointCloud2.h>
#include <sensor_msgs/PointCloud2.h>
#include <std_msgs/Header.h>
namespace pcl_conversions {
inline void fromPCL(const pcl::uint64_t &pcl_stamp, ros::Time &stamp) { stamp.fromNSec(pcl_stamp * 1000ull); }
inline void fromPCL(const pcl::PCLHeader &pcl_header, std_msgs::Header &header) {
  fromPCL(pcl_header.stamp, header.stamp); header.seq = pcl_header.seq; header.frame_id = pcl_header.frame_id;
}
}  // namespace pcl_conversions
