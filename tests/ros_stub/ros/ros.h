#pragma once
// ROS-shaped stand-ins (see README.md): ros::Time / Duration / NodeHandle / Publisher with the member names and call forms of roscpp.
// Messages travel over an in-process bus (ros_stub::bus) instead of TCPROS, so a test can publish into the subscriptions of the class
// under test and read what it published — no master, no network, no threads.
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <ostream>
#include <string>
#include <typeindex>
#include <vector>
#include <boost/shared_ptr.hpp>
namespace ros {
class Time {
 public:
  uint32_t sec, nsec;
  Time() : sec(0), nsec(0) {}
  Time(uint32_t _sec, uint32_t _nsec) : sec(_sec), nsec(_nsec) {}
  explicit Time(double t) : sec((uint32_t)t), nsec((uint32_t)((t - (double)(uint32_t)t) * 1e9)) {}
  Time &fromNSec(uint64_t t) { sec = (uint32_t)(t / 1000000000ull); nsec = (uint32_t)(t % 1000000000ull); return *this; }
  uint64_t toNSec() const { return (uint64_t)sec * 1000000000ull + (uint64_t)nsec; }
  double toSec() const { return (double)sec + 1e-9 * (double)nsec; }
  static Time now() { return Time(); }   // (no clock in the stub)
};
class Duration {
 public:
  int32_t sec, nsec;
  Duration() : sec(0), nsec(0) {}
  explicit Duration(double t) : sec((int32_t)t), nsec((int32_t)((t - (double)(int32_t)t) * 1e9)) {}
  double toSec() const { return (double)sec + 1e-9 * (double)nsec; }
};
std::ostream &operator<<(std::ostream &os, const Time &rhs);
}  // namespace ros

namespace ros_stub {
struct Topic {
  std::vector<std::pair<std::type_index, boost::shared_ptr<const void>>> published;   // every message published on the topic, in order
  std::vector<std::pair<std::type_index, std::function<void(const boost::shared_ptr<const void> &)>>> subscribers;
  int advertised = 0, queue_size = 0;
};
inline std::map<std::string, Topic> &bus() { static std::map<std::string, Topic> b; return b; }
template <class M> void deliver(const std::string &topic, const boost::shared_ptr<const M> &msg) {
  Topic &t = bus()[topic];
  t.published.emplace_back(std::type_index(typeid(M)), boost::shared_ptr<const void>(msg));
  for (auto &s : t.subscribers) if (s.first == std::type_index(typeid(M))) s.second(boost::shared_ptr<const void>(msg));
}
}  // namespace ros_stub

namespace ros {
class Publisher {
 public:
  Publisher() {}
  explicit Publisher(const std::string &topic) : topic_(topic) {}
  template <class M> void publish(const M &message) const { ros_stub::deliver<M>(topic_, boost::shared_ptr<const M>(new M(message))); }
  std::string getTopic() const { return topic_; }
 private:
  std::string topic_;
};
class NodeHandle {
 public:
  NodeHandle(const std::string &ns = std::string()) { (void)ns; }
  template <class M> Publisher advertise(const std::string &topic, uint32_t queue_size, bool latch = false) {
    (void)latch; ros_stub::Topic &t = ros_stub::bus()[topic]; ++t.advertised; t.queue_size = (int)queue_size; return Publisher(topic);
  }
};
}  // namespace ros
