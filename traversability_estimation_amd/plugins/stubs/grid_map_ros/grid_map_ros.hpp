#pragma once
#include <grid_map_core/GridMap.hpp>
#include <ros/ros.h>
