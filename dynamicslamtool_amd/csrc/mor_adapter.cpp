// mor_adapter.cpp — implementation of the drop-in class in include/MOR/MovingObjectRemoval.h on top
// of the C ABI (include/mor_hip.h).  Host-only C++; link with libmor_hip.so.
#include "MOR/MovingObjectRemoval.h"
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iostream>

namespace {
// PointXYZI record as toPCLPointCloud2<PointXYZI> serialises it (reference .cpp:690): 32 bytes,
// x@0 y@4 z@8 (pad 1.0f @12) intensity@16
constexpr uint32_t kStep = 32, kOffI = 16;

template <class Cloud, class Field> void describe_xyzi(Cloud &c, size_t n) {
  c.fields.clear();
  const char *names[4] = {"x", "y", "z", "intensity"}; const uint32_t offs[4] = {0, 4, 8, kOffI};
  for (int i = 0; i < 4; ++i) { Field f; f.name = names[i]; f.offset = offs[i]; f.datatype = 7 /*FLOAT32*/; f.count = 1; c.fields.push_back(f); }
  c.height = 1; c.width = (uint32_t)n; c.point_step = kStep; c.row_step = kStep * (uint32_t)n; c.is_bigendian = 0; c.is_dense = 1;
}
void expand(const float *xyzi, size_t n, std::vector<uint8_t> &blob) {
  blob.assign(n * kStep, 0);
  const float one = 1.0f;
  for (size_t i = 0; i < n; ++i) {
    uint8_t *r = blob.data() + i * kStep;
    std::memcpy(r, xyzi + 4 * i, 12); std::memcpy(r + 12, &one, 4); std::memcpy(r + kOffI, xyzi + 4 * i + 3, 4);
  }
}
}  // namespace

#ifdef MOR_WITH_ROS_PCL
MovingObjectRemoval::MovingObjectRemoval(ros::NodeHandle nh, std::string config_path, int n_bad, int n_good) : nh_(nh) {
#else
MovingObjectRemoval::MovingObjectRemoval(ros::NodeHandle, std::string config_path, int n_bad, int n_good) {
#endif
  std::memset(&params_, 0, sizeof params_);
  params_.opc_resolution = 0.1f;   // literal at the reference call site (.cpp:575)
  params_.ground_method = 0;       // .cpp:526 is the active call
  setVariables(config_path);
#ifdef MOR_WITH_ROS_PCL
  /*ROS setup* (.cpp:372-385): topics from the config file, same queue sizes */
#ifdef MOR_VISUALIZE
  pub_ = nh_.advertise<sensor_msgs::PointCloud2>(output_topic_, 10);
  debug_pub_ = nh_.advertise<sensor_msgs::PointCloud2>(debug_topic_, 10);
  marker_pub_ = nh_.advertise<visualization_msgs::Marker>(marker_topic_, 10);
#endif
#ifdef INTERNAL_SYNC
  pc_sub_.subscribe(nh_, input_pointcloud_topic_, 1);
  odom_sub_.subscribe(nh_, input_odometry_topic_, 1);
  sync_.reset(new message_filters::Synchronizer<MySyncPolicy>(MySyncPolicy(10), pc_sub_, odom_sub_));   // ROS Approximate Time policy (.cpp:381)
  sync_->registerCallback(&MovingObjectRemoval::movingCloudObjectSubscriber, this);
#endif
#endif
  const char *dev = std::getenv("MOR_DEVICE"), *cap = std::getenv("MOR_MAX_POINTS");
  int err = 0;
  // MOR_BIND_NUMA=1: keep the constructing thread (the one that will call push / filter) on the CPUs of the GPU's NUMA node — opt-in,
  // because a library should not move an application's threads unasked (5–6 % from the wrong socket)
  if (const char *bn = std::getenv("MOR_BIND_NUMA")) if (std::atoi(bn)) mor_bind_thread_to_device_node(dev ? std::atoi(dev) : 0, 0, 1);
  // How the incoming blob reaches the device and the filtered cloud leaves it.  Round 6 built the page-locked forms and measured them through mor_replay (bench.py's
  // class_latency_ms; one 120 000-point stream, median of 22 frames, MI355X host): pageable in + staged out 0.597 ms, bounce + direct 0.619, pageable + direct 0.610,
  // zerocopy + direct 0.606, bounce + staged 0.626 — against 0.438 ms for the C ABI from page-locked memory: the adapter's copies are 0.16 ms on this host whichever way
  // they are made (the device-side chain of thirteen launches is the latency), so the default stays the simplest form and the others are opt-in:
  //   MOR_CLASS_INPUT=pageable (default)  the caller's memory as it is, staged by the driver
  //   MOR_CLASS_INPUT=bounce              one memcpy into a page-locked buffer of the class, DMA from there
  //   MOR_CLASS_INPUT=zerocopy            the same buffer, read by the split kernel itself over PCIe (no staging copy on the device)
  //   MOR_CLASS_OUTPUT=staged (default)   16-byte points to a scratch vector, expanded to PointXYZI records on the host, copied to `output`
  //   MOR_CLASS_OUTPUT=direct             the device writes the 32-byte records straight into `output.data` (page-locked in place), ONE host copy to the caller's cloud
  if (const char *im = std::getenv("MOR_CLASS_INPUT")) input_mode_ = !std::strcmp(im, "bounce") ? 1 : !std::strcmp(im, "zerocopy") ? 2 : 0;
  if (const char *om = std::getenv("MOR_CLASS_OUTPUT")) output_direct_ = !std::strcmp(om, "direct");
  ctx_ = mor_create(&params_, n_bad, n_good, cap ? std::strtoull(cap, nullptr, 10) : (1ull << 20), dev ? std::atoi(dev) : 0, &err);
  if (!ctx_) { std::cerr << "MovingObjectRemoval: mor_create failed (" << err << "): " << mor_last_error() << std::endl; std::exit(1); }
}

MovingObjectRemoval::~MovingObjectRemoval() {
  if (out_reg_) mor_host_unregister(out_reg_);
  if (in_pinned_) mor_host_free(in_pinned_);
  mor_destroy(ctx_);
}

// same file format and the same echo as the reference's setVariables (.cpp:698-864): `key:value`,
// '#' comment lines and lines shorter than 3 characters skipped, every ':' dropped, no trimming;
// unreadable file or unknown key ⇒ message on stdout and exit(0).
void MovingObjectRemoval::setVariables(const std::string &path) {
  std::ifstream in(path);
  if (!in.is_open()) { std::cout << "Couldnt open the file\n"; std::exit(0); }
  struct FKey { const char *name; float mor_params::*field; };
  static const FKey fkeys[] = {
      {"gp_limit", &mor_params::gp_limit}, {"gp_leaf", &mor_params::gp_leaf}, {"bin_gap", &mor_params::bin_gap},
      {"volume_constraint", &mor_params::volume_constraint}, {"pde_lb", &mor_params::pde_lb}, {"pde_ub", &mor_params::pde_ub},
      {"leave_off_distance", &mor_params::leave_off_distance}, {"catch_up_distance", &mor_params::catch_up_distance},
      {"trim_x", &mor_params::trim_x}, {"trim_y", &mor_params::trim_y}, {"trim_z", &mor_params::trim_z},
      {"ec_distance_threshold", &mor_params::ec_distance_threshold}, {"pde_distance_threshold", &mor_params::pde_distance_threshold}};
  struct SKey { const char *name; std::string MovingObjectRemoval::*field; };
  static const SKey skeys[] = {
      {"output_topic", &MovingObjectRemoval::output_topic_}, {"debug_topic", &MovingObjectRemoval::debug_topic_},
      {"marker_topic", &MovingObjectRemoval::marker_topic_}, {"input_pointcloud_topic", &MovingObjectRemoval::input_pointcloud_topic_},
      {"input_odometry_topic", &MovingObjectRemoval::input_odometry_topic_}, {"output_fid", &MovingObjectRemoval::output_fid_},
      {"debug_fid", &MovingObjectRemoval::debug_fid_}};
  std::string line;
  while (std::getline(in, line)) {
    if (line.empty() || line[0] == '#' || line.length() < 3) continue;
    std::string key, val; bool in_key = true;
    for (char ch : line) { if (ch == ':') { in_key = false; continue; } (in_key ? key : val).push_back(ch); }
    std::cout << key << ":";
    bool known = false;
    for (const FKey &k : fkeys) if (key == k.name) { params_.*(k.field) = std::stof(val); std::cout << params_.*(k.field); known = true; }
    for (const SKey &k : skeys) if (key == k.name) { this->*(k.field) = val; std::cout << val; known = true; }
    if (key == "min_cluster_size") { params_.min_cluster_size = std::stol(val); std::cout << params_.min_cluster_size; known = true; }
    else if (key == "max_cluster_size") { params_.max_cluster_size = std::stol(val); std::cout << params_.max_cluster_size; known = true; }
    else if (key == "method_choice") { params_.method_choice = std::stoi(val); std::cout << params_.method_choice; known = true; }
    else if (key == "opc_normalization_factor") { params_.opc_normalization_factor = (int)std::stof(val); std::cout << params_.opc_normalization_factor; known = true; }   // stof into an int (.cpp:843)
    else if (key == "ground_method") { params_.ground_method = std::stoi(val); std::cout << params_.ground_method; known = true; }   // extension key: 0 crop box, 1 voxel covariance
    else if (key == "opc_anchor") { params_.opc_anchor = std::stoi(val); std::cout << params_.opc_anchor; known = true; }   // extension key: method-2 voxel lattice anchored at p0 − res (0) or p0 − res/2 (1)
    else if (key == "volume_abs_int") { params_.volume_abs_int = std::stoi(val); std::cout << params_.volume_abs_int; known = true; }   // extension key: 1 = the abs() of .cpp:277 truncates to int first (old libstdc++)
    if (!known) { std::cout << "Invalid parameter found in config file\n"; std::exit(0); }
    std::cout << std::endl;
  }
}

void MovingObjectRemoval::pushRawCloudAndPose(pcl::PCLPointCloud2 &cloud, geometry_msgs::Pose pose) {
  push_ok_ = false;
  // fromPCLPointCloud2 (.cpp:523): fields matched by name; a missing intensity stays 0
  uint32_t off[4] = {MOR_NO_FIELD, MOR_NO_FIELD, MOR_NO_FIELD, MOR_NO_FIELD};
  for (const auto &f : cloud.fields) {
    if (f.datatype != 7 /*FLOAT32*/) continue;
    if (f.name == "x") off[0] = f.offset; else if (f.name == "y") off[1] = f.offset; else if (f.name == "z") off[2] = f.offset; else if (f.name == "intensity") off[3] = f.offset;
  }
  const uint64_t n = (uint64_t)cloud.width * cloud.height;
  if (off[0] == MOR_NO_FIELD || off[1] == MOR_NO_FIELD || off[2] == MOR_NO_FIELD) { std::cerr << "MovingObjectRemoval: cloud has no float32 x/y/z fields" << std::endl; return; }
  // the blob must hold what its header promises (fromPCLPointCloud2 would read past the end otherwise); rows of an
  // organised cloud may be padded (row_step > width·point_step): the records are then gathered row by row
  const uint64_t row_bytes = (uint64_t)cloud.width * cloud.point_step;
  const uint64_t row_step = (cloud.height > 1 && cloud.row_step > row_bytes) ? cloud.row_step : row_bytes;
  const uint64_t need = cloud.height ? (uint64_t)(cloud.height - 1) * row_step + row_bytes : 0;
  if (cloud.data.size() < need) { std::cerr << "MovingObjectRemoval: cloud.data holds " << cloud.data.size() << " bytes, header needs " << need << std::endl; return; }
  const uint8_t *records = cloud.data.data();
  if (row_step != row_bytes) {
    rows_.resize((size_t)(n * cloud.point_step));
    for (uint32_t r = 0; r < cloud.height; ++r) std::memcpy(rows_.data() + (size_t)r * row_bytes, cloud.data.data() + (size_t)r * row_step, (size_t)row_bytes);
    records = rows_.data();
  }
  const double p[7] = {pose.position.x, pose.position.y, pose.position.z, pose.orientation.x, pose.orientation.y, pose.orientation.z, pose.orientation.w};
  mor_cloud_view view; view.data = records; view.n_points = n; view.point_step = cloud.point_step; view.off_x = off[0]; view.off_y = off[1]; view.off_z = off[2]; view.off_intensity = off[3]; view.on_device = 0;
  const size_t blob_bytes = (size_t)(n * cloud.point_step);
  if (input_mode_ != 0 && blob_bytes) {   // page-locked bounce buffer: grows to the largest blob seen
    if (in_cap_ < blob_bytes) {
      if (in_pinned_) mor_host_free(in_pinned_);
      in_cap_ = blob_bytes + blob_bytes / 4; in_pinned_ = static_cast<uint8_t *>(mor_host_alloc(in_cap_));
      if (!in_pinned_) { in_cap_ = 0; std::cerr << "MovingObjectRemoval: no page-locked memory for the input (" << mor_last_error() << "), using the caller's" << std::endl; }
    }
    if (in_pinned_) { std::memcpy(in_pinned_, records, blob_bytes); view.data = in_pinned_; view.on_device = input_mode_ == 2 ? 1 : 0; }
  }
  int rc = mor_push_batch(ctx_, &view, p);
  if (rc != MOR_OK) { std::cerr << "MovingObjectRemoval: mor_push failed (" << rc << "): " << mor_last_error() << std::endl; return; }
  push_ok_ = true;
  in_header_ = cloud.header;   // raw_cloud->header (fromPCLPointCloud2 copies it, .cpp:523); filterCloud's outputs carry it (.cpp:690-691)
  last_n_ = n; ++pushes_;
#ifdef MOR_VISUALIZE
  if (pushes_ >= 2) {   // inside `if(ca->init && cb->init)` (.cpp:534, :553-558)
    mor_counts c; mor_get_counts(ctx_, 0, &c);
    scratch_.resize(4 * (size_t)c.n_clustered + 4);
    mor_get_cluster_collection(ctx_, 0, scratch_.data());
    expand(scratch_.data(), c.n_clustered, cloud.data);
    describe_xyzi<pcl::PCLPointCloud2, pcl::PCLPointField>(cloud, c.n_clustered);
    cloud.header = pcl::PCLHeader();   // cluster_collection is a fresh PointCloud: default header (.cpp:554)
    if (out_reg_ && output.data.capacity() < cloud.data.size()) { mor_host_unregister(out_reg_); out_reg_ = nullptr; out_dev_ = nullptr; out_reg_bytes_ = 0; }   // (the assignment below would move the registered buffer)
    output.data = cloud.data;
    describe_xyzi<sensor_msgs::PointCloud2, sensor_msgs::PointField>(output, c.n_clustered);
    output.header = std_msgs::Header();   // fromPCL copies the (default) header (.cpp:555) …
    output.header.frame_id = debug_fid_;  // … then .cpp:556
#ifdef MOR_WITH_ROS_PCL
    debug_pub_.publish(output);           // .cpp:557
#endif
  }
#endif
}

#if defined(MOR_WITH_ROS_PCL) && defined(INTERNAL_SYNC)
// subscriber for internal sync (.cpp:393-413): toPCL → push → filter → publish, CPU time of the iteration printed in ms between two rules
void MovingObjectRemoval::movingCloudObjectSubscriber(const sensor_msgs::PointCloud2ConstPtr &input, const nav_msgs::OdometryConstPtr &odm) {
  clock_t begin_time = clock();
  std::cout << "-----------------------------------------------------\n";
  pcl::PCLPointCloud2 cloud;
  pcl_conversions::toPCL(*input, cloud);
  pushRawCloudAndPose(cloud, odm->pose.pose);
  if (filterCloud(cloud, output_fid_)) {
#ifdef MOR_VISUALIZE
    pub_.publish(output);
#endif
  }
  std::cout << 1000.0 * (clock() - begin_time) / CLOCKS_PER_SEC << std::endl;
  std::cout << "-----------------------------------------------------\n";
}
#endif

#if defined(MOR_WITH_ROS_PCL) && defined(MOR_VISUALIZE)
// mark_cluster (.cpp:7-58): CUBE at the float-accumulated centroid, scale = box extent (0 → 0.1), colour and lifetime of the caller at .cpp:623/:641
visualization_msgs::Marker MovingObjectRemoval::toMarker(const BoxMarker &m, int id) const {
  visualization_msgs::Marker marker;
  marker.header.frame_id = debug_fid_;
  marker.header.stamp = ros::Time::now();
  marker.ns = "bounding_box";
  marker.id = id;
  marker.type = visualization_msgs::Marker::CUBE;
  marker.action = visualization_msgs::Marker::ADD;
  marker.pose.position.x = m.position[0]; marker.pose.position.y = m.position[1]; marker.pose.position.z = m.position[2];
  marker.pose.orientation.x = 0.0; marker.pose.orientation.y = 0.0; marker.pose.orientation.z = 0.0; marker.pose.orientation.w = 1.0;
  marker.scale.x = m.scale[0]; marker.scale.y = m.scale[1]; marker.scale.z = m.scale[2];
  marker.color.r = 0.8f; marker.color.g = 0.1f; marker.color.b = 0.4f; marker.color.a = 0.5f;   // rd, gd, bd of .cpp:623; opacity .cpp:53
  marker.lifetime = ros::Duration(2);
  return marker;
}
#endif

std::vector<MovingObjectRemoval::BoxMarker> MovingObjectRemoval::movingMarkers() const {
  std::vector<BoxMarker> out;
  uint32_t n = 0;
  if (!ctx_ || mor_get_moving_clusters(ctx_, 0, nullptr, &n) != MOR_OK || n == 0) return out;
  std::vector<int32_t> cl(n);
  if (mor_get_moving_clusters(ctx_, 0, cl.data(), &n) != MOR_OK) return out;
  const std::vector<BoxMarker> all = clusterMarkers();
  for (uint32_t i = 0; i < n; ++i) if (cl[i] >= 0 && (size_t)cl[i] < all.size()) { BoxMarker m = all[(size_t)cl[i]]; m.id = (int)i + 1; out.push_back(m); }   // id starts at 1, one per visited tracked centroid (.cpp:623, :669)
  return out;
}

std::vector<MovingObjectRemoval::BoxMarker> MovingObjectRemoval::clusterMarkers() const {
  std::vector<BoxMarker> out;
  mor_counts c;
  if (!ctx_ || mor_get_counts(ctx_, 0, &c) != MOR_OK || c.n_clusters == 0) return out;
  std::vector<float> pos(3 * (size_t)c.n_clusters), scale(3 * (size_t)c.n_clusters);
  std::vector<uint8_t> det(c.n_clusters);
  if (mor_get_markers(ctx_, 0, pos.data(), scale.data()) != MOR_OK || mor_get_detection(ctx_, 0, det.data()) != MOR_OK) return out;
  out.resize(c.n_clusters);
  for (uint32_t k = 0; k < c.n_clusters; ++k) {
    BoxMarker &m = out[k];
    m.id = (int)k; m.moving = det[k] != 0;
    for (int a = 0; a < 3; ++a) { m.position[a] = pos[3 * k + a]; m.scale[a] = scale[3 * k + a]; }   // float-accumulated centroid (.cpp:15), extent with 0 → 0.1 (.cpp:36-47)
  }
  return out;
}

bool MovingObjectRemoval::filterCloud(pcl::PCLPointCloud2 &out_cloud, std::string f_id) {
  // the reference cannot fail here (.cpp:695); this implementation can when the preceding push was refused (capacity,
  // malformed blob, GPU error): the frame never reached the device, so there is nothing to filter — report it
  // instead of re-emitting the previous frame's cloud
  if (!push_ok_) { std::cerr << "MovingObjectRemoval: filterCloud without a successful pushRawCloudAndPose" << std::endl; return false; }
  uint64_t n = 0;
  bool direct = false;
  if (output_direct_ && last_n_) {
    // toPCLPointCloud2 (.cpp:690) on the device: k_out writes the 32-byte PointXYZI records straight into `output.data` — the class's own vector, page-locked
    // and mapped for the device while its buffer stays where it is (re-registered when it has grown or the caller has moved it away) — so there is no 16-byte
    // intermediate, no expansion loop and ONE host copy (output → out_cloud) instead of two
    const size_t need = (size_t)last_n_ * kStep;
    if (out_reg_ != output.data.data() || out_reg_bytes_ < need || output.data.capacity() < need) {
      if (out_reg_) { mor_host_unregister(out_reg_); out_reg_ = nullptr; out_dev_ = nullptr; out_reg_bytes_ = 0; }
      if (output.data.capacity() < need) { output.data.clear(); output.data.reserve(need + need / 4); }
      void *dp = nullptr;
      if (mor_host_register(output.data.data(), output.data.capacity(), &dp) == MOR_OK) { out_reg_ = output.data.data(); out_dev_ = dp; out_reg_bytes_ = output.data.capacity(); }
    }
    if (out_reg_) {
      output.data.resize(need);   // (within the capacity: the buffer does not move; what grows is zero-filled, a few per cent of the cloud from one frame to the next)
      void *o = out_dev_;
      int rc = mor_filter_batch_ex(ctx_, &o, 1, &n, kStep);
      if (rc != MOR_OK) { std::cerr << "MovingObjectRemoval: mor_filter failed (" << rc << "): " << mor_last_error() << std::endl; return false; }
      output.data.resize((size_t)n * kStep);
      out_cloud.data.assign(output.data.begin(), output.data.end());   // pcl_conversions::fromPCL (.cpp:691) copies in the other direction; both hold the records afterwards either way
      direct = true;
    }
  }
  if (!direct) {
    scratch_.resize(4 * (size_t)last_n_ + 4);
    int rc = mor_filter(ctx_, scratch_.data(), &n);
    if (rc != MOR_OK) { std::cerr << "MovingObjectRemoval: mor_filter failed (" << rc << "): " << mor_last_error() << std::endl; return false; }
    expand(scratch_.data(), n, out_cloud.data);   // toPCLPointCloud2 (.cpp:690)
    output.data = out_cloud.data;                 // pcl_conversions::fromPCL (.cpp:691)
  }
  describe_xyzi<pcl::PCLPointCloud2, pcl::PCLPointField>(out_cloud, n);
  // f_cloud is created empty and filled by ExtractIndices::filter(f_cloud), which copies the header of its input cloud
  // cb->cloud (= the incoming cloud's header, carried through fromPCLPointCloud2 and the filters); toPCLPointCloud2
  // (.cpp:690) and fromPCL (.cpp:691) pass it on, .cpp:692 then overwrites frame_id
  out_cloud.header = in_header_;
  describe_xyzi<sensor_msgs::PointCloud2, sensor_msgs::PointField>(output, n);
  pcl_conversions::fromPCL(in_header_, output.header);   // seq, stamp.fromNSec(pcl stamp [µs] · 1000), frame_id — integer arithmetic, as .cpp:691 does
  output.header.frame_id = f_id;                // .cpp:692
#if defined(MOR_WITH_ROS_PCL) && defined(MOR_VISUALIZE)
  for (const BoxMarker &m : movingMarkers()) marker_pub_.publish(toMarker(m, m.id));   // the reference publishes them inside its loop over mo_vec (.cpp:641)
#endif
  return true;
}
