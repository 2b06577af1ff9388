// include/flame/utils/triangulator.h -- flame_ros includes this header for flame::Triangle /
// flame::Edge (reference src/utils.h:36,95; src/utils.cc:170,224-226).  Upstream's header also
// holds its Delaunay wrapper; triangulation itself is upstream of the regulariser path and is
// handed to flame::Flame through FrontEnd::triangulate (flame.h).
#pragma once
#include "../types.h"
