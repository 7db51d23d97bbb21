// IncludeAll.h — include hub of the MI355X-native adapter (counterpart of the reference's
// include/MOR/IncludeAll.h:1-39, which pulls in ROS, PCL, FLANN, tf).  The GPU path needs none of
// PCL's algorithms; only the message types of the public signature remain.
#pragma once
#ifdef MOR_WITH_ROS_PCL
#include <geometry_msgs/Pose.h>
#include <pcl/PCLPointCloud2.h>
#include <pcl_conversions/pcl_conversions.h>
#include <ros/ros.h>
#include <sensor_msgs/PointCloud2.h>
#else
#include "MOR/shim/ros_pcl_types.h"
#endif
#include <memory>
#include <string>

// The reference compiles with VISUALIZE defined (IncludeAll.h:32): pushRawCloudAndPose then
// overwrites the caller's cloud and `output` with the concatenated clusters (:553-558).  The
// adapter reproduces that side effect when MOR_VISUALIZE is defined (default, as in the reference);
// the RViz marker / debug publishers themselves need ROS and are out of scope.
#ifndef MOR_NO_VISUALIZE
#define MOR_VISUALIZE
#endif
