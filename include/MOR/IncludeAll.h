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
#include <message_filters/subscriber.h>
#include <message_filters/synchronizer.h>
#include <message_filters/sync_policies/approximate_time.h>
#include <nav_msgs/Odometry.h>
#include <visualization_msgs/Marker.h>
#else
#include "MOR/shim/ros_pcl_types.h"
#endif
#include <memory>
#include <string>

// The reference compiles with VISUALIZE defined (IncludeAll.h:32): pushRawCloudAndPose then
// overwrites the caller's cloud and `output` with the concatenated clusters (:553-558).  The
// adapter reproduces that side effect when MOR_VISUALIZE is defined (default, as in the reference);
// the RViz marker / debug publishers themselves exist in a ROS build only (MOR_WITH_ROS_PCL).
#ifndef MOR_NO_VISUALIZE
#define MOR_VISUALIZE
#endif

// The reference's INTERNAL_SYNC (IncludeAll.h:36, off by default): the class subscribes to the config file's input topics itself,
// synchronises point clouds with odometry (ApproximateTime) and runs push + filter + publish in its own callback
// (src/MovingObjectRemoval.cpp:379-385, :393-413).  Same macro name here; it needs roscpp + message_filters, i.e. MOR_WITH_ROS_PCL.
#if defined(INTERNAL_SYNC) && !defined(MOR_WITH_ROS_PCL)
#error "INTERNAL_SYNC needs roscpp and message_filters: build with -DMOR_WITH_ROS_PCL"
#endif
