// stand-in for <ros/ros.h>: only the logging macros the filters use
#pragma once
#include <cstdio>
#define ROS_ERROR(...) do { std::fprintf(stderr, "[ERROR] "); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); } while (0)
#define ROS_WARN(...) do { std::fprintf(stderr, "[WARN] "); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); } while (0)
#define ROS_DEBUG(...) do { } while (0)
#define ROS_INFO(...) do { } while (0)
