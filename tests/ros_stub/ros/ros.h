#pragma once
#include <cstdint>
#include <ostream>
#include <string>
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
};
std::ostream &operator<<(std::ostream &os, const Time &rhs);
class NodeHandle {
 public:
  NodeHandle(const std::string &ns = std::string()) { (void)ns; }
};
}  // namespace ros
