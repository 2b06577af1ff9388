// include/flame/utils/visualization.h -- jet colormap and applyColorMap as flame_ros calls them
// (reference src/flame_offline_tum.cc:337-342: colormap lambda (float v, cv::Vec3b c) ->
// cv::Vec3b built on flame::utils::jet(v, 0.0f, 0.35f); applyColorMap<float>(idepth_error,
// colormap, debug_img)).  Channel order is BGR (the images are published as "bgr8",
// src/flame_offline_tum.cc:730).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>

#include "../types.h"

namespace flame {
namespace utils {

// MATLAB-style jet: vmin -> blue, middle -> green/yellow, vmax -> red.  Values outside are clamped.
inline Vec3b jet(float v, float vmin, float vmax) {
  float t = (vmax > vmin) ? (v - vmin) / (vmax - vmin) : 0.0f;
  t = std::max(0.0f, std::min(1.0f, t));
  auto ramp = [](float x) { return std::max(0.0f, std::min(1.0f, x)); };
  const float r = ramp(1.5f - std::fabs(4.0f * t - 3.0f));
  const float g = ramp(1.5f - std::fabs(4.0f * t - 2.0f));
  const float b = ramp(1.5f - std::fabs(4.0f * t - 1.0f));
  return Vec3b(static_cast<uint8_t>(255.0f * b + 0.5f), static_cast<uint8_t>(255.0f * g + 0.5f),
               static_cast<uint8_t>(255.0f * r + 0.5f));
}

// out(i, j) = colormap(in(i, j), out(i, j)) for every pixel; `out` must have the size of `in`
// (the callers pre-fill it with the grayscale image, so the colormap can keep that pixel).
template <typename T, typename ColorMap>
inline void applyColorMap(const
#ifdef FLAME_HAVE_OPENCV
                          cv::Mat_<T>&
#else
                          ImageT<T>&
#endif
                              in,
                          ColorMap colormap, Image3b* out) {
  for (int i = 0; i < in.rows; ++i)
    for (int j = 0; j < in.cols; ++j) (*out)(i, j) = colormap(in(i, j), (*out)(i, j));
}

}  // namespace utils
}  // namespace flame
