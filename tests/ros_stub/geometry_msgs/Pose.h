#pragma once
#include <memory>
namespace geometry_msgs {
template <class A> struct Point_ { Point_() : x(0.0), y(0.0), z(0.0) {} double x, y, z; };
template <class A> struct Quaternion_ { Quaternion_() : x(0.0), y(0.0), z(0.0), w(0.0) {} double x, y, z, w; };
template <class A> struct Pose_ { Point_<A> position; Quaternion_<A> orientation; };
typedef Point_<std::allocator<void>> Point;
typedef Quaternion_<std::allocator<void>> Quaternion;
typedef Pose_<std::allocator<void>> Pose;
}  // namespace geometry_msgs
