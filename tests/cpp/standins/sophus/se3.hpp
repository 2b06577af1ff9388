// TEST STAND-IN for <sophus/se3.hpp>, see ../README.md.
#pragma once
#include <Eigen/Core>

namespace Sophus {

class SE3f {
 public:
  SE3f() {}
  SE3f(const Eigen::Quaternionf& q, const Eigen::Vector3f& t) : q_(q), t_(t) {}
  const Eigen::Quaternionf& unit_quaternion() const { return q_; }
  const Eigen::Vector3f& translation() const { return t_; }

 private:
  Eigen::Quaternionf q_;
  Eigen::Vector3f t_;
};

}  // namespace Sophus
