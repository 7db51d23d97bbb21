#pragma once
#include <memory>
#include <string>
#include <ros/ros.h>
namespace std_msgs {
template <class ContainerAllocator> struct Header_ {
  typedef Header_<ContainerAllocator> Type;
  Header_() : seq(0), stamp(), frame_id() {}
  uint32_t seq;
  ros::Time stamp;
  std::basic_string<char, std::char_traits<char>, typename std::allocator_traits<ContainerAllocator>::template rebind_alloc<char>> frame_id;
};
typedef ::std_msgs::Header_<std::allocator<void>> Header;
}  // namespace std_msgs
