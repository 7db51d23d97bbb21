#pragma once
#include <string>
#include <boost/shared_ptr.hpp>
#include <geometry_msgs/Pose.h>
#include <std_msgs/Header.h>
namespace geometry_msgs {
template <class A> struct PoseWithCovariance_ { PoseWithCovariance_() : pose(), covariance() {} Pose_<A> pose; double covariance[36]; };
template <class A> struct Vector3_ { Vector3_() : x(0.0), y(0.0), z(0.0) {} double x, y, z; };
template <class A> struct Twist_ { Vector3_<A> linear, angular; };
template <class A> struct TwistWithCovariance_ { TwistWithCovariance_() : twist(), covariance() {} Twist_<A> twist; double covariance[36]; };
}  // namespace geometry_msgs
namespace nav_msgs {
template <class A> struct Odometry_ {
  ::std_msgs::Header_<A> header;
  std::string child_frame_id;
  ::geometry_msgs::PoseWithCovariance_<A> pose;
  ::geometry_msgs::TwistWithCovariance_<A> twist;
};
typedef Odometry_<std::allocator<void>> Odometry;
typedef boost::shared_ptr<Odometry> OdometryPtr;
typedef boost::shared_ptr<Odometry const> OdometryConstPtr;
}  // namespace nav_msgs
