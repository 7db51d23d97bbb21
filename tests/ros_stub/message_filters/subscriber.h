#pragma once
// message_filters::Subscriber<M>: subscribe(nh, topic, queue_size) + registerCallback — over the in-process bus of ros/ros.h.
#include <functional>
#include <string>
#include <vector>
#include <ros/ros.h>
namespace message_filters {
template <class M> class Subscriber {
 public:
  typedef boost::shared_ptr<M const> MConstPtr;
  Subscriber() {}
  Subscriber(ros::NodeHandle &nh, const std::string &topic, uint32_t queue_size) { subscribe(nh, topic, queue_size); }
  void subscribe(ros::NodeHandle &, const std::string &topic, uint32_t queue_size) {
    topic_ = topic; (void)queue_size;
    ros_stub::bus()[topic].subscribers.emplace_back(std::type_index(typeid(M)), [this](const boost::shared_ptr<const void> &v) {
      MConstPtr m = std::static_pointer_cast<M const>(v);
      for (auto &cb : callbacks_) cb(m);
    });
  }
  void registerCallback(const std::function<void(const MConstPtr &)> &cb) { callbacks_.push_back(cb); }
  std::string getTopic() const { return topic_; }
 private:
  std::string topic_;
  std::vector<std::function<void(const MConstPtr &)>> callbacks_;
};
}  // namespace message_filters
