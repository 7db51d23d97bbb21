#pragma once
// boost::shared_ptr as far as roscpp's message pointers need it here: the std one under boost's name (ROS 1 message ConstPtr types are
// boost::shared_ptr<M const>).  Stand-in for compile / in-process tests only (tests/ros_stub/README.md).
#include <memory>
namespace boost {
template <class T> using shared_ptr = std::shared_ptr<T>;
template <class T, class... A> shared_ptr<T> make_shared(A &&...a) { return std::make_shared<T>(std::forward<A>(a)...); }
}  // namespace boost
