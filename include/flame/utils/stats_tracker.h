// include/flame/utils/stats_tracker.h -- flame::utils::StatsTracker as flame_ros consumes it:
// tick/tock/set, stats(name) -> double, timings(name) -> double in MILLISECONDS, stats()/
// timings() -> unordered_map<string,double>; a missing key reads as 0 (the frontends test
// `stats_.stats("fps_max") <= 0.0f` on the first frame).  Reference call sites:
// src/flame_offline_tum.cc:380-392,418-430,504-522,706-707; src/utils.h:72-83.
#pragma once
#include <chrono>
#include <string>
#include <unordered_map>

namespace flame {
namespace utils {

class StatsTracker {
 public:
  void tick(const std::string& name) { starts_[name] = std::chrono::steady_clock::now(); }
  double tock(const std::string& name) {
    auto it = starts_.find(name);
    if (it == starts_.end()) return 0.0;
    const double ms =
        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - it->second).count();
    timings_[name] = ms;
    starts_.erase(it);  // a second tock without a tick is a no-op
    return ms;
  }
  void set(const std::string& name, double v) { stats_[name] = v; }
  void setTiming(const std::string& name, double ms) { timings_[name] = ms; }
  double stats(const std::string& name) const {
    auto it = stats_.find(name);
    return it == stats_.end() ? 0.0 : it->second;
  }
  double timings(const std::string& name) const {
    auto it = timings_.find(name);
    return it == timings_.end() ? 0.0 : it->second;
  }
  const std::unordered_map<std::string, double>& stats() const { return stats_; }
  const std::unordered_map<std::string, double>& timings() const { return timings_; }

 private:
  std::unordered_map<std::string, double> stats_, timings_;
  std::unordered_map<std::string, std::chrono::steady_clock::time_point> starts_;
};

}  // namespace utils
}  // namespace flame
