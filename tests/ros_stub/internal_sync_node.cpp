// internal_sync_node.cpp — test counterpart of the reference's src/internal_sync_test.cpp:1-10 (a node that only constructs the class and
// spins; the class subscribes, synchronises, runs push + filter and publishes by itself under INTERNAL_SYNC, src/MovingObjectRemoval.cpp:379-413).
// Instead of ros::spin() over TCPROS this driver publishes the recorded clouds and poses into the class's subscriptions through the in-process
// bus of tests/ros_stub/ros/ros.h and writes what the class published:
//   filtered_%04d.bin         output_topic  (sensor_msgs::PointCloud2 → packed xyzi float32)
//   debug_%04d.bin            debug_topic   (the clustered points of the frame, from the second frame on; packed xyzi)
//   moving_markers_%04d.txt   marker_topic  (id px py pz sx sy sz r g b a lifetime ns frame_id)
//   internal_sync_node <config> <poses.txt> <out_dir> <cloud0.bin> ...     (same command line as mor_replay)
// Built by tests/test_adapter.py with -DMOR_WITH_ROS_PCL -DINTERNAL_SYNC -I tests/ros_stub.  TEST INFRASTRUCTURE.
#include "MOR/MovingObjectRemoval.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

std::ostream &ros::operator<<(std::ostream &os, const ros::Time &t) { char b[16]; std::snprintf(b, sizeof b, "%09u", t.nsec); return os << t.sec << "." << b; }

static bool config_value(const char *path, const std::string &key, std::string &val) {
  std::ifstream in(path); std::string line;
  while (std::getline(in, line)) if (line.compare(0, key.size() + 1, key + ":") == 0) { val = line.substr(key.size() + 1); return true; }
  return false;
}
static void write_packed(const sensor_msgs::PointCloud2 &m, const std::string &path) {
  std::vector<float> packed(4 * (size_t)m.width);
  for (size_t k = 0; k < m.width; ++k) { const uint8_t *r = m.data.data() + k * m.point_step; std::memcpy(&packed[4 * k], r, 12); std::memcpy(&packed[4 * k + 3], r + 16, 4); }
  std::ofstream o(path, std::ios::binary); o.write((const char *)packed.data(), (std::streamsize)(packed.size() * sizeof(float)));
}

int main(int argc, char **argv) {
  if (argc < 5) { std::fprintf(stderr, "usage: %s <config> <poses.txt> <out_dir> <cloud.bin>...\n", argv[0]); return 2; }
  setenv("GPU_MAX_HW_QUEUES", "8", 0);
  ros::NodeHandle nh;
  MovingObjectRemoval mor(nh, argv[1], 4, 3);   // internal_sync_test.cpp:8
  std::string t_in, t_odom, t_out, t_dbg, t_mk;
  if (!config_value(argv[1], "input_pointcloud_topic", t_in) || !config_value(argv[1], "input_odometry_topic", t_odom) || !config_value(argv[1], "output_topic", t_out) ||
      !config_value(argv[1], "debug_topic", t_dbg) || !config_value(argv[1], "marker_topic", t_mk)) { std::fprintf(stderr, "topics missing from the config file\n"); return 1; }
  auto &bus = ros_stub::bus();
  if (bus[t_in].subscribers.size() != 1 || bus[t_odom].subscribers.size() != 1) { std::fprintf(stderr, "the class did not subscribe to %s / %s\n", t_in.c_str(), t_odom.c_str()); return 1; }
  if (bus[t_out].advertised != 1 || bus[t_dbg].advertised != 1 || bus[t_mk].advertised != 1) { std::fprintf(stderr, "the class did not advertise its three topics\n"); return 1; }
  ros::Publisher cloud_pub = nh.advertise<sensor_msgs::PointCloud2>(t_in, 1), odom_pub = nh.advertise<nav_msgs::Odometry>(t_odom, 1);   // the sensor drivers' side
  std::ifstream poses(argv[2]);
  const std::string out_dir = argv[3];
  for (int i = 4; i < argc; ++i) {
    sensor_msgs::PointCloud2 msg;
    { std::ifstream f(argv[i], std::ios::binary | std::ios::ate); if (!f) { std::fprintf(stderr, "cannot read %s\n", argv[i]); return 1; }
      std::streamsize n = f.tellg(); f.seekg(0); msg.data.resize((size_t)n); f.read((char *)msg.data.data(), n); }
    const char *names[4] = {"x", "y", "z", "intensity"};
    for (int k = 0; k < 4; ++k) { sensor_msgs::PointField f; f.name = names[k]; f.offset = 4 * k; f.datatype = sensor_msgs::PointField::FLOAT32; f.count = 1; msg.fields.push_back(f); }
    msg.point_step = 16; msg.width = (uint32_t)(msg.data.size() / 16); msg.height = 1; msg.row_step = 16 * msg.width; msg.is_dense = 1;
    msg.header.seq = (uint32_t)(i - 4); msg.header.stamp.fromNSec(1000000000ull * (uint64_t)(100 + i - 4)); msg.header.frame_id = "/velodyne";
    nav_msgs::Odometry od; std::string line;
    if (!std::getline(poses, line)) { std::fprintf(stderr, "poses file too short\n"); return 1; }
    std::istringstream ls(line);
    ls >> od.pose.pose.position.x >> od.pose.pose.position.y >> od.pose.pose.position.z >> od.pose.pose.orientation.x >> od.pose.pose.orientation.y >> od.pose.pose.orientation.z >> od.pose.pose.orientation.w;
    od.header = msg.header;
    const size_t n_out0 = bus[t_out].published.size(), n_dbg0 = bus[t_dbg].published.size(), n_mk0 = bus[t_mk].published.size();
    cloud_pub.publish(msg);      // first half of the pair: nothing may happen yet
    if (bus[t_out].published.size() != n_out0) { std::fprintf(stderr, "the callback ran before the odometry arrived\n"); return 1; }
    odom_pub.publish(od);        // the synchroniser fires movingCloudObjectSubscriber: toPCL → push → filter → publish
    if (bus[t_out].published.size() != n_out0 + 1) { std::fprintf(stderr, "frame %d: nothing published on %s\n", i - 4, t_out.c_str()); return 1; }
    char name[64];
    const auto out = std::static_pointer_cast<const sensor_msgs::PointCloud2>(bus[t_out].published.back().second);
    std::snprintf(name, sizeof name, "/filtered_%04d.bin", i - 4); write_packed(*out, out_dir + name);
    std::cout << "frame " << (i - 4) << ": published " << out->width << " pts, frame_id " << out->header.frame_id << ", seq " << out->header.seq << ", stamp " << out->header.stamp.toNSec() << " ns, debug clouds "
              << (bus[t_dbg].published.size() - n_dbg0) << ", markers " << (bus[t_mk].published.size() - n_mk0) << std::endl;
    if (bus[t_dbg].published.size() > n_dbg0) {
      const auto dbg = std::static_pointer_cast<const sensor_msgs::PointCloud2>(bus[t_dbg].published.back().second);
      std::snprintf(name, sizeof name, "/debug_%04d.bin", i - 4); write_packed(*dbg, out_dir + name);
      if (dbg->header.frame_id.empty()) { std::fprintf(stderr, "debug cloud without frame_id\n"); return 1; }
    }
    std::snprintf(name, sizeof name, "/moving_markers_%04d.txt", i - 4);
    std::ofstream mk(out_dir + name); mk.precision(9);
    for (size_t j = n_mk0; j < bus[t_mk].published.size(); ++j) {
      const auto m = std::static_pointer_cast<const visualization_msgs::Marker>(bus[t_mk].published[j].second);
      mk << m->id << ' ' << m->pose.position.x << ' ' << m->pose.position.y << ' ' << m->pose.position.z << ' ' << m->scale.x << ' ' << m->scale.y << ' ' << m->scale.z << ' '
         << m->color.r << ' ' << m->color.g << ' ' << m->color.b << ' ' << m->color.a << ' ' << m->lifetime.toSec() << ' ' << m->ns << ' ' << m->header.frame_id << ' ' << m->type << ' ' << m->action << '\n';
    }
  }
  return 0;
}
