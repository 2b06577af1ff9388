// include/flame/utils/assert.h -- FLAME_ASSERT as flame_ros uses it: crash handlers call
// FLAME_ASSERT(false) (reference src/flame_offline_tum.cc:74-77, src/flame_nodelet.cc:85-87), bounds
// checks when scattering raw idepths (src/flame_offline_tum.cc:691-694), resize_factor == 1 (:255).
// A failed assertion reports where and aborts; it is active in every build type.
#pragma once
#include <cstdio>
#include <cstdlib>

#define FLAME_ASSERT(cond)                                                                   \
  do {                                                                                       \
    if (!(cond)) {                                                                           \
      std::fprintf(stderr, "FLAME_ASSERT failed: %s (%s:%d, %s)\n", #cond, __FILE__, __LINE__, \
                   __func__);                                                                \
      std::abort();                                                                          \
    }                                                                                        \
  } while (0)
