// tools/pin_upstream/fldump.h -- tiny header-only writer of the FLDUMP1 container: named dense
// arrays of float32 / int32 with up to 3 dimensions.  No dependencies (C++11), so it can be dropped
// next to a robustrobotics/flame checkout.  tools/pin_upstream/convert_dump.py turns a dump into
// the tests/golden/upstream_<tag>.npz layout that tests/test_upstream_pin.py consumes.
//
//   file   := "FLDUMP1\n" record*
//   record := name '\n' dtype('f'|'i') ndim(uint32) shape(uint32 x ndim) data(little endian)
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace fldump {

class Writer {
 public:
  explicit Writer(const std::string& path) : f_(std::fopen(path.c_str(), "wb")) {
    if (f_) std::fputs("FLDUMP1\n", f_);
  }
  ~Writer() { if (f_) std::fclose(f_); }
  bool ok() const { return f_ != nullptr; }
  void floats(const std::string& name, const std::vector<float>& v, uint32_t d0, uint32_t d1 = 0, uint32_t d2 = 0) {
    put(name, 'f', v.data(), v.size(), d0, d1, d2);
  }
  void ints(const std::string& name, const std::vector<int32_t>& v, uint32_t d0, uint32_t d1 = 0, uint32_t d2 = 0) {
    put(name, 'i', v.data(), v.size(), d0, d1, d2);
  }

 private:
  void put(const std::string& name, char dtype, const void* data, size_t n, uint32_t d0, uint32_t d1, uint32_t d2) {
    if (!f_) return;
    uint32_t shape[3] = {d0, d1, d2};
    uint32_t ndim = d2 ? 3 : (d1 ? 2 : 1);
    size_t want = d0;
    if (d1) want *= d1;
    if (d2) want *= d2;
    if (want != n) { std::fprintf(stderr, "fldump: %s has %zu values, shape says %zu\n", name.c_str(), n, want); return; }
    std::fputs(name.c_str(), f_);
    std::fputc('\n', f_);
    std::fputc(dtype, f_);
    std::fwrite(&ndim, 4, 1, f_);
    std::fwrite(shape, 4, ndim, f_);
    std::fwrite(data, 4, n, f_);
  }
  FILE* f_;
};

}  // namespace fldump
