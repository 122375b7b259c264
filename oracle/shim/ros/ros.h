// oracle/shim/ros/ros.h -- TEST INFRASTRUCTURE ONLY. include/parameters.h only names ros::NodeHandle
// in a method declaration; src/parameters.cpp is never compiled by the oracle.
#pragma once
namespace ros {
class NodeHandle;
}
