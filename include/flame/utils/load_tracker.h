// include/flame/utils/load_tracker.h -- flame::utils::LoadTracker / Load as flame_ros uses them:
// member `fu::LoadTracker load_` constructed with getpid() (reference src/flame_offline_tum.cc:100),
// move-assigned (src/flame_nodelet.cc:153), sampled every load_integration_factor frames with
// load_.get(&max_load, &sys_load, &pid_load), fields cpu / mem / swap (src/flame_offline_tum.cc:
// 529-543).  Linux /proc based: cpu = busy fraction since the previous get(), mem / swap = used
// fraction of the system (sys) or of the system's memory used by the process (pid).
#pragma once
#include <sys/types.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>
#include <string>

namespace flame {
namespace utils {

struct Load {
  float cpu = 0.0f;   // fraction of all cores, 0..1
  float mem = 0.0f;   // fraction of physical memory
  float swap = 0.0f;  // fraction of swap
};

class LoadTracker {
 public:
  LoadTracker() : LoadTracker(getpid()) {}
  explicit LoadTracker(pid_t pid) : pid_(pid) { sample(&prev_total_, &prev_busy_, &prev_proc_); }
  LoadTracker(LoadTracker&&) = default;
  LoadTracker& operator=(LoadTracker&&) = default;
  LoadTracker(const LoadTracker&) = default;
  LoadTracker& operator=(const LoadTracker&) = default;

  // max_load: the running maximum of sys_load over the calls so far
  void get(Load* max_load, Load* sys_load, Load* pid_load) {
    unsigned long long total = 0, busy = 0, proc = 0;
    sample(&total, &busy, &proc);
    Load sys, pid;
    const double dt = static_cast<double>(total - prev_total_);
    if (dt > 0.0) {
      sys.cpu = static_cast<float>((busy - prev_busy_) / dt);
      pid.cpu = static_cast<float>((proc - prev_proc_) / dt);
    }
    prev_total_ = total; prev_busy_ = busy; prev_proc_ = proc;
    unsigned long long mem_total = 0, mem_avail = 0, swap_total = 0, swap_free = 0, rss = 0, vmswap = 0;
    meminfo("/proc/meminfo", "MemTotal:", &mem_total);
    meminfo("/proc/meminfo", "MemAvailable:", &mem_avail);
    meminfo("/proc/meminfo", "SwapTotal:", &swap_total);
    meminfo("/proc/meminfo", "SwapFree:", &swap_free);
    const std::string st = "/proc/" + std::to_string(static_cast<long long>(pid_)) + "/status";
    meminfo(st.c_str(), "VmRSS:", &rss);
    meminfo(st.c_str(), "VmSwap:", &vmswap);
    if (mem_total) { sys.mem = static_cast<float>(1.0 - static_cast<double>(mem_avail) / mem_total); pid.mem = static_cast<float>(static_cast<double>(rss) / mem_total); }
    if (swap_total) { sys.swap = static_cast<float>(1.0 - static_cast<double>(swap_free) / swap_total); pid.swap = static_cast<float>(static_cast<double>(vmswap) / swap_total); }
    if (sys.cpu > max_.cpu) max_.cpu = sys.cpu;
    if (sys.mem > max_.mem) max_.mem = sys.mem;
    if (sys.swap > max_.swap) max_.swap = sys.swap;
    if (max_load) *max_load = max_;
    if (sys_load) *sys_load = sys;
    if (pid_load) *pid_load = pid;
  }

 private:
  static void meminfo(const char* path, const char* key, unsigned long long* kb) {
    FILE* f = std::fopen(path, "r");
    if (!f) return;
    char line[256];
    const size_t n = std::strlen(key);
    while (std::fgets(line, sizeof(line), f))
      if (std::strncmp(line, key, n) == 0) { std::sscanf(line + n, "%llu", kb); break; }
    std::fclose(f);
  }
  void sample(unsigned long long* total, unsigned long long* busy, unsigned long long* proc) const {
    *total = *busy = *proc = 0;
    FILE* f = std::fopen("/proc/stat", "r");
    if (f) {
      unsigned long long v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (std::fscanf(f, "cpu %llu %llu %llu %llu %llu %llu %llu %llu", &v[0], &v[1], &v[2], &v[3], &v[4],
                      &v[5], &v[6], &v[7]) >= 4) {
        for (int k = 0; k < 8; ++k) *total += v[k];
        *busy = *total - v[3] - v[4];  // minus idle and iowait
      }
      std::fclose(f);
    }
    const std::string st = "/proc/" + std::to_string(static_cast<long long>(pid_)) + "/stat";
    f = std::fopen(st.c_str(), "r");
    if (f) {
      char buf[1024];
      if (std::fgets(buf, sizeof(buf), f)) {
        const char* p = std::strrchr(buf, ')');  // the command name may contain spaces
        unsigned long long ut = 0, stt = 0;
        if (p && std::sscanf(p + 1, " %*c %*d %*d %*d %*d %*d %*u %*u %*u %*u %*u %llu %llu", &ut, &stt) == 2)
          *proc = ut + stt;
      }
      std::fclose(f);
    }
  }

  pid_t pid_;
  unsigned long long prev_total_ = 0, prev_busy_ = 0, prev_proc_ = 0;
  Load max_;
};

}  // namespace utils
}  // namespace flame
