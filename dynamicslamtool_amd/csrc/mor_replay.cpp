// mor_replay.cpp — ROS-free replay driver: the counterpart of the reference's demo node
// src/external_sync_test.cpp:7-22 (callback: toPCL → pushRawCloudAndPose → filterCloud → publish).
// Reads a sequence of clouds (KITTI-style .bin: packed float32 x,y,z,intensity) and poses
// (text, one "x y z qx qy qz qw" line per frame), drives the drop-in class exactly as the demo node
// does, and writes every filtered cloud (`output`) next to the input as packed xyzi float32.
//   mor_replay <config.txt> <poses.txt> <out_dir> <cloud0.bin> <cloud1.bin> ...
#include "MOR/MovingObjectRemoval.h"
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <cstdlib>
#include <sstream>

static bool read_file(const std::string &path, std::vector<uint8_t> &out) {
  std::ifstream f(path, std::ios::binary | std::ios::ate);
  if (!f) return false;
  std::streamsize n = f.tellg(); f.seekg(0);
  out.resize((size_t)n);
  return (bool)f.read((char *)out.data(), n);
}

#ifndef MOR_SRC_HASH_STR
#define MOR_SRC_HASH_STR "MOR_SRC_HASH=unknown"
#endif

int main(int argc, char **argv) {
  if (argc < 5) { std::fprintf(stderr, "usage: %s <config> <poses.txt> <out_dir> <cloud.bin>...\n(%s)\n", argv[0], MOR_SRC_HASH_STR); return 2; }
  setenv("GPU_MAX_HW_QUEUES", "8", 0);   // the application's choice: one hardware queue per stage stream of the library's frame pipeline
  ros::NodeHandle nh;
  MovingObjectRemoval mor(nh, argv[1], 4, 3);   // n_bad = 4, n_good = 3 as in external_sync_test.cpp:37
  std::ifstream poses(argv[2]);
  const std::string out_dir = argv[3];
  for (int i = 4; i < argc; ++i) {
    pcl::PCLPointCloud2 cloud;   // what pcl_conversions::toPCL would hand over for a packed xyzi cloud
    if (!read_file(argv[i], cloud.data)) { std::fprintf(stderr, "cannot read %s\n", argv[i]); return 1; }
    const char *names[4] = {"x", "y", "z", "intensity"};
    for (int k = 0; k < 4; ++k) { pcl::PCLPointField f; f.name = names[k]; f.offset = 4 * k; f.datatype = pcl::PCLPointField::FLOAT32; f.count = 1; cloud.fields.push_back(f); }
    cloud.point_step = 16; cloud.width = (uint32_t)(cloud.data.size() / 16); cloud.height = 1; cloud.row_step = 16 * cloud.width; cloud.is_dense = 1;
    cloud.header.seq = (uint32_t)(i - 4); cloud.header.stamp = 1000000ull * (uint64_t)(100 + i - 4); cloud.header.frame_id = "/velodyne";   // what toPCL copies from the sensor message
    geometry_msgs::Pose pose; std::string line;
    if (!std::getline(poses, line)) { std::fprintf(stderr, "poses file too short\n"); return 1; }
    std::istringstream ls(line);
    ls >> pose.position.x >> pose.position.y >> pose.position.z >> pose.orientation.x >> pose.orientation.y >> pose.orientation.z >> pose.orientation.w;
    auto t0 = std::chrono::steady_clock::now();
    mor.pushRawCloudAndPose(cloud, pose);
#ifdef MOR_VISUALIZE
    {  // under VISUALIZE the push overwrites the caller's cloud and `output` with the clustered points (.cpp:553-558): dump what the caller now holds
      char pn[64]; std::snprintf(pn, sizeof pn, "/pushed_%04d.bin", i - 4);
      std::ofstream po(out_dir + pn, std::ios::binary); po.write((const char *)cloud.data.data(), (std::streamsize)cloud.data.size());
      std::cout << "pushed " << (i - 4) << ": caller cloud width " << cloud.width << " point_step " << cloud.point_step << " output.width " << mor.output.width << " output.frame_id " << mor.output.header.frame_id << std::endl;
    }
#endif
    bool ok = mor.filterCloud(cloud, "/filtered");
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (!ok) return 1;
    if (std::getenv("MOR_REPLAY_NO_OUTPUT")) {   // timing runs (bench.py's class_latency_ms): the per-frame line only
      std::cout << "frame " << (i - 4) << ": " << cloud.width << " pts in filtered cloud, " << ms << " ms" << std::endl;
      continue;
    }
    // "publish": unpack the 32-byte PointXYZI records of `output` back to packed xyzi
    const size_t n = mor.output.width;
    std::vector<float> packed(4 * n);
    for (size_t k = 0; k < n; ++k) { const uint8_t *r = mor.output.data.data() + k * mor.output.point_step; std::memcpy(&packed[4 * k], r, 12); std::memcpy(&packed[4 * k + 3], r + 16, 4); }
    char name[64]; std::snprintf(name, sizeof name, "/filtered_%04d.bin", i - 4);
    std::ofstream o(out_dir + name, std::ios::binary); o.write((const char *)packed.data(), packed.size() * sizeof(float));
    // the debug bounding boxes (what the reference publishes as markers under VISUALIZE): id px py pz sx sy sz moving
    std::snprintf(name, sizeof name, "/markers_%04d.txt", i - 4);
    std::ofstream mk(out_dir + name); mk.precision(9);
    const auto markers = mor.clusterMarkers();
    for (const auto &m : markers) mk << m.id << ' ' << m.position[0] << ' ' << m.position[1] << ' ' << m.position[2] << ' ' << m.scale[0] << ' ' << m.scale[1] << ' ' << m.scale[2] << ' ' << (m.moving ? 1 : 0) << '\n';
    std::cout << "frame " << (i - 4) << ": " << cloud.width << " pts in filtered cloud, " << markers.size() << " cluster boxes, frame_id " << mor.output.header.frame_id << ", seq " << mor.output.header.seq << ", stamp " << mor.output.header.stamp << " (" << mor.output.header.stamp.toNSec() << " ns), cloud seq " << cloud.header.seq << " stamp " << cloud.header.stamp << ", " << ms << " ms" << std::endl;
  }
  return 0;
}
