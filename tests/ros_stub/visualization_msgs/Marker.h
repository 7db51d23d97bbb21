#pragma once
#include <string>
#include <ros/ros.h>
#include <geometry_msgs/Pose.h>
#include <std_msgs/Header.h>
namespace std_msgs { template <class A> struct ColorRGBA_ { ColorRGBA_() : r(0.f), g(0.f), b(0.f), a(0.f) {} float r, g, b, a; }; }
namespace visualization_msgs {
template <class A> struct Marker_ {
  Marker_() : header(), ns(), id(0), type(0), action(0), pose(), scale(), color(), lifetime(), frame_locked(false) {}
  enum { ARROW = 0u, CUBE = 1u, SPHERE = 2u, CYLINDER = 3u };
  enum { ADD = 0u, MODIFY = 0u, DELETE = 2u, DELETEALL = 3u };
  ::std_msgs::Header_<A> header;
  std::string ns;
  int32_t id, type, action;
  ::geometry_msgs::Pose_<A> pose;
  struct Scale { Scale() : x(0.0), y(0.0), z(0.0) {} double x, y, z; } scale;   // geometry_msgs::Vector3
  ::std_msgs::ColorRGBA_<A> color;
  ros::Duration lifetime;
  uint8_t frame_locked;
};
typedef Marker_<std::allocator<void>> Marker;
}  // namespace visualization_msgs
