// include/flame/utils/triangulator.h -- flame_ros includes this header for flame::Triangle /
// flame::Edge (reference src/utils.h:36,95; src/utils.cc:170,224-226).  Upstream's header also
// holds its Delaunay wrapper (around Shewchuk's Triangle); here the triangulator is delaunay.h, the
// default of flame::Flame's FrontEnd::triangulate (flame.h), on the host, off the regulariser path.
#pragma once
#include "../types.h"
#include "delaunay.h"  // flame::utils::delaunay(): the build's own exact triangulator
