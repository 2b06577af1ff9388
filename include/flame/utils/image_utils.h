// include/flame/utils/image_utils.h -- the scalar helpers flame_ros takes from
// flame::utils (reference src/flame_offline_tum.cc:608 fast_abs, :688-689 fast_roundf,
// src/utils.cc:346,356).
#pragma once
#include <cmath>

#include "../types.h"

namespace flame {
namespace utils {

inline float fast_abs(float v) { return v < 0.0f ? -v : v; }
// round half away from zero, as an int (pixel coordinates)
inline int fast_roundf(float v) { return static_cast<int>(v + (v >= 0.0f ? 0.5f : -0.5f)); }
inline int fast_floor(float v) { const int i = static_cast<int>(v); return i - (static_cast<float>(i) > v ? 1 : 0); }
inline int fast_ceil(float v) { const int i = static_cast<int>(v); return i + (static_cast<float>(i) < v ? 1 : 0); }

}  // namespace utils
}  // namespace flame
