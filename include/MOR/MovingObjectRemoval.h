// MovingObjectRemoval.h — the reference's public class, backed by libmor_hip.so (MI355X / HIP).
//
// Drop-in for /root/reference/include/MOR/MovingObjectRemoval.h:96-168: same class name, same
// constructor and method signatures, same public member `output`, same call protocol
// (README.md:16-29, src/external_sync_test.cpp:14-17).  All geometry runs on the GPU through the C
// ABI in include/mor_hip.h; this class only parses the config file, maps the PCLPointCloud2 blob
// fields and re-expands the 16-byte device points into PCL's 32-byte PointXYZI records.
#pragma once
#include "MOR/IncludeAll.h"
#include "mor_hip.h"

#ifdef MOR_WITH_ROS_PCL
typedef message_filters::sync_policies::ApproximateTime<sensor_msgs::PointCloud2, nav_msgs::Odometry> MySyncPolicy;   // reference header :2
#endif

class MovingObjectRemoval {
 public:
  sensor_msgs::PointCloud2 output;   // filtered cloud of the last filterCloud() (reference header :159)

  // config_path: key:value file, grammar of setVariables (reference .cpp:698-864); n_bad / n_good as
  // in the reference (:368).  Extra knobs come from the environment so the signature stays intact:
  // MOR_DEVICE (HIP ordinal, default 0), MOR_MAX_POINTS (capacity per cloud, default 2^20).
  MovingObjectRemoval(ros::NodeHandle nh, std::string config_path, int n_bad, int n_good);
  ~MovingObjectRemoval();
  MovingObjectRemoval(const MovingObjectRemoval &) = delete;
  MovingObjectRemoval &operator=(const MovingObjectRemoval &) = delete;

  // Input (reference :163 / .cpp:516-611).  Under MOR_VISUALIZE the caller's cloud and `output` are
  // overwritten with the clustered points of the new frame from the second call on (:553-558).
  void pushRawCloudAndPose(pcl::PCLPointCloud2 &cloud, geometry_msgs::Pose pose);

  // Output (reference :166 / .cpp:613-696): `cloud` and `output` receive the latest cloud minus the
  // tracked moving clusters, ground points appended; PointXYZI layout (x@0,y@4,z@8,intensity@16,
  // point_step 32), width = n, height = 1, is_dense = true, output.header.frame_id = f_id.
  // Returns true (the reference cannot fail); false only if the GPU call failed or the preceding push was refused.
  bool filterCloud(pcl::PCLPointCloud2 &cloud, std::string f_id);

  // not in the reference's public interface: the data of its debug bounding-box markers (mark_cluster, .cpp:7-58 —
  // a CUBE at the cluster centroid, scale = max − min of its points, zero extents replaced by 0.1), one entry per
  // cluster of the latest frame in cluster order; `moving` = detection_results of that cluster.  A ROS build turns
  // them into visualization_msgs::Marker with header.frame_id = debug_fid(), ns "bounding_box", lifetime 2 s.
  struct BoxMarker { int id; float position[3], scale[3]; bool moving; };
  std::vector<BoxMarker> clusterMarkers() const;

  // not in the reference: parameters as parsed (for tools/tests) and the last error text
  const mor_params &params() const { return params_; }
  const std::string &debug_fid() const { return debug_fid_; }
  const std::string &output_fid() const { return output_fid_; }
  // the tracked moving clusters of the latest filterCloud, in mo_vec order: one marker per tracked centroid visited by the loop (.cpp:630-642), id 1, 2, …
  std::vector<BoxMarker> movingMarkers() const;

 private:
  void setVariables(const std::string &config_file_path);
  mor_params params_;
  std::string output_topic_, debug_topic_, marker_topic_, input_pointcloud_topic_, input_odometry_topic_, output_fid_, debug_fid_;
  mor_ctx *ctx_ = nullptr;
  uint64_t pushes_ = 0, last_n_ = 0;
  bool push_ok_ = false;            // the latest pushRawCloudAndPose reached the device
  pcl::PCLHeader in_header_;        // header of the latest incoming cloud
  std::vector<float> scratch_;
  std::vector<uint8_t> rows_;       // de-padded rows of an organised cloud
  uint8_t *in_pinned_ = nullptr; size_t in_cap_ = 0;   // page-locked bounce buffer of the incoming blob (MOR_CLASS_INPUT)
  int input_mode_ = 0;              // 0 pageable (the caller's memory as it is), 1 bounce + DMA, 2 bounce + the split kernel reads it over PCIe
  bool output_direct_ = false;      // MOR_CLASS_OUTPUT=direct: the device writes the PointXYZI records straight into `output.data` (page-locked while its buffer stays in place)
  uint8_t *out_reg_ = nullptr; void *out_dev_ = nullptr; size_t out_reg_bytes_ = 0;   // the registered range of output.data and its device address
#ifdef MOR_WITH_ROS_PCL
  // ---- ROS plumbing of the reference's constructor (.cpp:372-385) and of its internal-sync callback (.cpp:393-413)
  ros::NodeHandle nh_;              // (the reference keeps a reference to its by-value constructor argument, header :131; a copy here)
#ifdef MOR_VISUALIZE
  ros::Publisher pub_, debug_pub_, marker_pub_;   // output_topic, debug_topic, marker_topic of the config file (.cpp:374-376)
  visualization_msgs::Marker toMarker(const BoxMarker &m, int id) const;   // mark_cluster (.cpp:7-58)
#endif
#ifdef INTERNAL_SYNC
  message_filters::Subscriber<sensor_msgs::PointCloud2> pc_sub_;
  message_filters::Subscriber<nav_msgs::Odometry> odom_sub_;
  std::unique_ptr<message_filters::Synchronizer<MySyncPolicy>> sync_;
  void movingCloudObjectSubscriber(const sensor_msgs::PointCloud2ConstPtr &input, const nav_msgs::OdometryConstPtr &odm);
#endif
#endif
};
