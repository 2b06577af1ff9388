// include/flame_ros/dataset_streams.h -- ROS-free dataset harness: the index
// parsing, timestamp association and pose-frame conversions of flame_ros' offline streams, without
// ROS / OpenCV / Eigen / yaml-cpp / Boost.  What it restates (behaviour, not text):
//   TUM   reference src/ros_sensor_streams/tum_rgbd_offline_stream.cc:124-195 (pose -> optical
//         frame for RDF / FLU / FRD / RDF_IN_FLU / RDF_IN_FRD inputs), :248-300 (index line:
//         pose_time tx ty tz qx qy qz qw rgb_time rgb_file [depth_time depth_file]; the rgb time is
//         the frame time; fewer than 11 tokens = no depth image)
//   ASL   reference src/ros_sensor_streams/asl_rgbd_offline_stream.cc:62-147 (sensor.yaml:
//         resolution, intrinsics, distortion_coefficients, T_BS, depth_scale_factor), :152-203
//         (rgb / depth / pose association), :205-275 (pose sensor -> body -> camera chain, world
//         frame RDF / FLU / FRD / RFU); src/dataset_utils/asl/types.h:37-120 (csv records),
//         src/dataset_utils/utils.h:50-93 (greedy closest-first association, max_diff 0.02 s),
//         src/dataset_utils/asl/dataset.h:83-103 (folder = sensor.yaml + data.csv, first line =
//         column names)
// Frames carry the image FILE PATHS, the depth scale factor and the camera pose; the pixels come
// through image_io.h (PNG / PNM decode, gray conversion, plumb-bob rectification, depth scaling --
// what the reference does with cv::imread / rectifyImage / cv::undistort): loadFramePixels() below
// turns a frame record into the cv::Mat1b / cv::Mat1f contents flame::Flame::update() receives.
// Header-only, C++11.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <tuple>
#include <unordered_set>
#include <vector>

#include "image_io.h"

namespace flame_ros {
namespace datasets {

// ---- small quaternion / rotation algebra (w, x, y, z), double precision ----
struct Quat {
  double w, x, y, z;
  Quat() : w(1), x(0), y(0), z(0) {}
  Quat(double w_, double x_, double y_, double z_) : w(w_), x(x_), y(y_), z(z_) {}
  Quat operator*(const Quat& o) const {
    return Quat(w * o.w - x * o.x - y * o.y - z * o.z, w * o.x + x * o.w + y * o.z - z * o.y,
                w * o.y - x * o.z + y * o.w + z * o.x, w * o.z + x * o.y - y * o.x + z * o.w);
  }
  Quat inverse() const { const double n = w * w + x * x + y * y + z * z; return Quat(w / n, -x / n, -y / n, -z / n); }
  void normalize() { const double n = std::sqrt(w * w + x * x + y * y + z * z); w /= n; x /= n; y /= n; z /= n; }
  // rotate a vector: q v q^-1 for a unit quaternion
  void rotate(const double v[3], double out[3]) const {
    const double tx = 2 * (y * v[2] - z * v[1]), ty = 2 * (z * v[0] - x * v[2]), tz = 2 * (x * v[1] - y * v[0]);
    out[0] = v[0] + w * tx + (y * tz - z * ty);
    out[1] = v[1] + w * ty + (z * tx - x * tz);
    out[2] = v[2] + w * tz + (x * ty - y * tx);
  }
  // from a row-major rotation matrix (the branch on the largest diagonal term keeps it stable)
  static Quat fromMatrix(const double R[9]) {
    Quat q;
    const double tr = R[0] + R[4] + R[8];
    if (tr > 0) {
      const double s = std::sqrt(tr + 1.0) * 2;
      q = Quat(0.25 * s, (R[7] - R[5]) / s, (R[2] - R[6]) / s, (R[3] - R[1]) / s);
    } else if (R[0] > R[4] && R[0] > R[8]) {
      const double s = std::sqrt(1.0 + R[0] - R[4] - R[8]) * 2;
      q = Quat((R[7] - R[5]) / s, 0.25 * s, (R[1] + R[3]) / s, (R[2] + R[6]) / s);
    } else if (R[4] > R[8]) {
      const double s = std::sqrt(1.0 + R[4] - R[0] - R[8]) * 2;
      q = Quat((R[2] - R[6]) / s, (R[1] + R[3]) / s, 0.25 * s, (R[5] + R[7]) / s);
    } else {
      const double s = std::sqrt(1.0 + R[8] - R[0] - R[4]) * 2;
      q = Quat((R[3] - R[1]) / s, (R[2] + R[6]) / s, (R[5] + R[7]) / s, 0.25 * s);
    }
    return q;
  }
};

struct Pose {  // camera in world
  Quat q;
  double t[3];
  Pose() : q() { t[0] = t[1] = t[2] = 0; }
};

// world/body frame conventions of the inputs (the outputs are always optical = Right-Down-Forward)
enum Frame { RDF, FLU, FRD, RDF_IN_FLU, RDF_IN_FRD, RFU };

inline Quat qFluToRdf() { return Quat(-0.5, -0.5, 0.5, -0.5); }
inline Quat qFrdToRdf() { const double R[9] = {0, 1, 0, 0, 0, 1, 1, 0, 0}; return Quat::fromMatrix(R); }
inline Quat qRfuToRdf() { const double R[9] = {1, 0, 0, 0, 0, -1, 0, 1, 0}; return Quat::fromMatrix(R); }

// TUM convention (tum_rgbd_offline_stream.cc:145-194): FLU / FRD conjugate the rotation (the pose is
// expressed in that convention on BOTH sides), *_IN_* only re-expresses the world side.
inline bool tumToOptical(Frame f, const Pose& in, Pose* out) {
  Quat c;
  bool conj;
  switch (f) {
    case RDF: *out = in; return true;
    case FLU: c = qFluToRdf(); conj = true; break;
    case FRD: c = qFrdToRdf(); conj = true; break;
    case RDF_IN_FLU: c = qFluToRdf(); conj = false; break;
    case RDF_IN_FRD: c = qFrdToRdf(); conj = false; break;
    default: return false;
  }
  out->q = conj ? c * in.q * c.inverse() : c * in.q;
  c.rotate(in.t, out->t);
  return true;
}

// ---- text helpers (dataset_utils/utils.h: readLines, split) ----
inline std::vector<std::string> readLines(const std::string& file) {
  std::vector<std::string> out;
  std::ifstream f(file.c_str());
  std::string line;
  while (std::getline(f, line)) {
    if (!line.empty() && line[line.size() - 1] == '\r') line.erase(line.size() - 1);
    out.push_back(line);
  }
  return out;
}
inline std::vector<std::string> split(const std::string& s, char delim = ' ') {
  std::vector<std::string> out;
  std::stringstream ss(s);
  std::string item;
  while (std::getline(ss, item, delim)) out.push_back(item);
  return out;
}
inline std::string parentDir(const std::string& path) {
  const size_t k = path.find_last_of('/');
  return k == std::string::npos ? std::string(".") : path.substr(0, k);
}

// ---- TUM index (one line per frame) ----
struct TumFrame {
  double time;  // = rgb time
  double pose_time, rgb_time, depth_time;
  Pose pose_raw;      // as written in the file (normalised quaternion)
  Pose pose_optical;  // camera in world, optical convention
  std::string rgb_file, depth_file;  // relative to the index file; depth_file empty = no depth
  bool has_depth;
};

inline bool parseTumLine(const std::string& line, Frame input_frame, TumFrame* fr) {
  std::istringstream ss(line);
  double v[8];
  int tokens = 0;
  if (ss >> fr->pose_time) ++tokens;
  for (int k = 0; k < 7; ++k) { v[k] = 0; if (ss >> v[k]) ++tokens; }
  fr->rgb_time = fr->depth_time = 0;
  fr->rgb_file.clear(); fr->depth_file.clear();
  if (ss >> fr->rgb_time) ++tokens;
  if (ss >> fr->rgb_file) ++tokens;
  if (ss >> fr->depth_time) ++tokens;
  if (ss >> fr->depth_file) ++tokens;
  if (tokens < 10) return false;  // no pose + rgb entry on this line
  fr->time = fr->rgb_time;
  fr->pose_raw.t[0] = v[0]; fr->pose_raw.t[1] = v[1]; fr->pose_raw.t[2] = v[2];
  fr->pose_raw.q = Quat(v[6], v[3], v[4], v[5]);  // file order: qx qy qz qw
  fr->pose_raw.q.normalize();
  fr->has_depth = tokens >= 11 && !fr->depth_file.empty();
  if (!fr->has_depth) fr->depth_file.clear();
  return tumToOptical(input_frame, fr->pose_raw, &fr->pose_optical);
}

class TumIndex {
 public:
  // depth_scale_factor: raw uint16 depth / factor = metres (cfg: 5000 for TUM)
  TumIndex(const std::string& index_file, Frame input_frame, float depth_scale_factor = 5000.0f)
      : base_dir_(parentDir(index_file)), depth_scale_factor_(depth_scale_factor), next_(0) {
    const std::vector<std::string> lines = readLines(index_file);
    for (size_t k = 0; k < lines.size(); ++k) {
      if (lines[k].empty() || lines[k][0] == '#') continue;
      TumFrame fr;
      if (parseTumLine(lines[k], input_frame, &fr)) frames_.push_back(fr);
    }
  }
  bool empty() const { return next_ >= frames_.size(); }
  size_t size() const { return frames_.size(); }
  const TumFrame& frame(size_t k) const { return frames_[k]; }
  // next frame: id = running index (the reference's img_id), paths made absolute
  bool get(uint32_t* id, TumFrame* fr) {
    if (empty()) return false;
    *id = static_cast<uint32_t>(next_);
    *fr = frames_[next_++];
    fr->rgb_file = base_dir_ + "/" + fr->rgb_file;
    if (fr->has_depth) fr->depth_file = base_dir_ + "/" + fr->depth_file;
    return true;
  }
  float depthScaleFactor() const { return depth_scale_factor_; }

 private:
  std::string base_dir_;
  float depth_scale_factor_;
  size_t next_;
  std::vector<TumFrame> frames_;
};

// ---- ASL csv records (dataset_utils/asl/types.h) ----
struct AslPose { uint64_t timestamp; double trans[3]; double quat[4]; /* x y z w */ };
struct AslFile { uint64_t timestamp; std::string filename; };

inline bool parseAslPose(const std::string& csv, AslPose* p) {  // timestamp,tx,ty,tz,qw,qx,qy,qz
  const std::vector<std::string> t = split(csv, ',');
  if (t.size() < 8) return false;
  p->timestamp = std::strtoull(t[0].c_str(), nullptr, 10);
  for (int k = 0; k < 3; ++k) p->trans[k] = std::atof(t[1 + k].c_str());
  p->quat[3] = std::atof(t[4].c_str());
  for (int k = 0; k < 3; ++k) p->quat[k] = std::atof(t[5 + k].c_str());
  return true;
}
inline bool parseAslFile(const std::string& csv, AslFile* f) {  // timestamp,filename
  const std::vector<std::string> t = split(csv, ',');
  if (t.size() < 2) return false;
  f->timestamp = std::strtoull(t[0].c_str(), nullptr, 10);
  f->filename = t[1];
  while (!f->filename.empty() && (f->filename[0] == ' ')) f->filename.erase(0, 1);
  return true;
}

// ---- the subset of YAML an ASL sensor.yaml uses: `key: scalar`, `key: [a, b, ...]` (may span
// lines), and one nested level (`T_BS:` followed by indented `rows:`, `cols:`, `data: [...]`) ----
class SensorYaml {
 public:
  explicit SensorYaml(const std::string& file) {
    const std::vector<std::string> lines = readLines(file);
    std::string prefix;
    for (size_t k = 0; k < lines.size(); ++k) {
      std::string line = lines[k];
      const size_t hash = line.find('#');
      if (hash != std::string::npos) line.erase(hash);
      if (line.find_first_not_of(" \t") == std::string::npos) continue;
      const bool indented = line[0] == ' ' || line[0] == '\t';
      const size_t colon = line.find(':');
      if (colon == std::string::npos) continue;
      std::string key = trim(line.substr(0, colon)), val = trim(line.substr(colon + 1));
      if (!indented) prefix.clear();
      if (val.empty()) { prefix = key + "."; continue; }  // a nested map starts
      if (val[0] == '[') {
        while (val.find(']') == std::string::npos && k + 1 < lines.size()) val += " " + trim(lines[++k]);
        val = val.substr(1, val.find(']') - 1);
      }
      map_[(indented ? prefix : std::string()) + key] = val;
    }
  }
  bool has(const std::string& key) const { return map_.count(key) != 0; }
  std::vector<double> numbers(const std::string& key) const {
    std::vector<double> out;
    std::map<std::string, std::string>::const_iterator it = map_.find(key);
    if (it == map_.end()) return out;
    const std::vector<std::string> t = split(it->second, ',');
    for (size_t k = 0; k < t.size(); ++k)
      if (!trim(t[k]).empty()) out.push_back(std::atof(trim(t[k]).c_str()));
    return out;
  }
  // dataset_utils/utils.h readMatrix: rows / cols must match
  bool matrix(const std::string& name, int rows, int cols, double* out) const {
    const std::vector<double> r = numbers(name + ".rows"), c = numbers(name + ".cols"), d = numbers(name + ".data");
    if (r.size() != 1 || c.size() != 1 || static_cast<int>(r[0]) != rows || static_cast<int>(c[0]) != cols ||
        static_cast<int>(d.size()) != rows * cols)
      return false;
    std::copy(d.begin(), d.end(), out);
    return true;
  }

 private:
  static std::string trim(const std::string& s) {
    const size_t a = s.find_first_not_of(" \t"), b = s.find_last_not_of(" \t");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
  }
  std::map<std::string, std::string> map_;
};

// dataset_utils/utils.h associate(): every pair closer than max_diff is a candidate; candidates are
// taken closest first, each element at most once; the selected index lists come back sorted.
// Ties in distance are broken by (a index, b index) so the result is deterministic.
inline void associate(const std::vector<uint64_t>& a_ns, const std::vector<uint64_t>& b_ns,
                      std::vector<size_t>* aidxs, std::vector<size_t>* bidxs, float max_diff = 0.02f) {
  std::vector<std::tuple<float, size_t, size_t> > cand;
  for (size_t i = 0; i < a_ns.size(); ++i)
    for (size_t j = 0; j < b_ns.size(); ++j) {
      const float d = static_cast<float>(std::fabs(static_cast<double>(a_ns[i]) * 1e-9 - static_cast<double>(b_ns[j]) * 1e-9));
      if (d < max_diff) cand.push_back(std::make_tuple(d, i, j));
    }
  std::sort(cand.begin(), cand.end());
  std::unordered_set<size_t> as, bs;
  aidxs->clear(); bidxs->clear();
  for (size_t k = 0; k < cand.size(); ++k) {
    const size_t i = std::get<1>(cand[k]), j = std::get<2>(cand[k]);
    if (!as.count(i) && !bs.count(j)) { aidxs->push_back(i); bidxs->push_back(j); as.insert(i); bs.insert(j); }
  }
  std::sort(aidxs->begin(), aidxs->end());
  std::sort(bidxs->begin(), bidxs->end());
}

struct AslFrame {
  double time;  // rgb timestamp in seconds
  Pose pose_optical;
  std::string rgb_file, depth_file;
  bool has_depth;
};

class AslDataset {
 public:
  // pose_path / rgb_path / depth_path: ASL sensor folders (sensor.yaml + data.csv [+ data/]);
  // depth_path may be empty.
  AslDataset(const std::string& pose_path, const std::string& rgb_path, const std::string& depth_path,
             Frame world_frame)
      : ok_(false), world_frame_(world_frame), next_(0), width_(0), height_(0), depth_scale_(0) {
    const std::string pp = strip(pose_path), rp = strip(rgb_path), dp = strip(depth_path);
    readCsv(pp + "/data.csv", &poses_);
    readCsv(rp + "/data.csv", &rgbs_);
    if (!dp.empty()) readCsv(dp + "/data.csv", &depths_);
    rgb_dir_ = rp; depth_dir_ = dp;
    SensorYaml ry(rp + "/sensor.yaml"), py(pp + "/sensor.yaml");
    const std::vector<double> res = ry.numbers("resolution"), in = ry.numbers("intrinsics"),
                              dc = ry.numbers("distortion_coefficients");
    if (res.size() < 2 || in.size() < 4) return;
    width_ = static_cast<int>(res[0]); height_ = static_cast<int>(res[1]);
    for (int k = 0; k < 9; ++k) K_[k] = 0;
    K_[0] = in[0]; K_[4] = in[1]; K_[2] = in[2]; K_[5] = in[3]; K_[8] = 1;
    for (int k = 0; k < 5; ++k) D_[k] = k < static_cast<int>(dc.size()) && k < 4 ? dc[k] : 0.0;  // k3 = 0
    if (!dp.empty()) {
      SensorYaml dy(dp + "/sensor.yaml");
      const std::vector<double> f = dy.numbers("depth_scale_factor");
      if (f.size() == 1) depth_scale_ = f[0];
    }
    double Tp[16], Tc[16];
    if (!py.matrix("T_BS", 4, 4, Tp) || !ry.matrix("T_BS", 4, 4, Tc)) return;
    splitT(Tp, &q_pose_in_body_, t_pose_in_body_);
    splitT(Tc, &q_cam_in_body_, t_cam_in_body_);
    associateAll();
    ok_ = true;
  }
  bool ok() const { return ok_; }
  bool empty() const { return next_ >= pose_idxs_.size(); }
  size_t size() const { return pose_idxs_.size(); }
  int width() const { return width_; }
  int height() const { return height_; }
  const double* K() const { return K_; }
  const double* D() const { return D_; }
  double depthScaleFactor() const { return depth_scale_; }
  const std::vector<size_t>& poseIdxs() const { return pose_idxs_; }
  const std::vector<size_t>& rgbIdxs() const { return rgb_idxs_; }
  const std::vector<size_t>& depthIdxs() const { return depth_idxs_; }

  bool frame(size_t k, AslFrame* fr) const {
    if (k >= pose_idxs_.size()) return false;
    const AslFile& rgb = rgbs_[rgb_idxs_[k]];
    const AslPose& p = poses_[pose_idxs_[k]];
    fr->time = static_cast<double>(rgb.timestamp) * 1e-9;
    Quat q_pose_in_world(p.quat[3], p.quat[0], p.quat[1], p.quat[2]);
    q_pose_in_world.normalize();
    // pose sensor in world -> body in world -> camera in world
    const Quat q_body_in_pose = q_pose_in_body_.inverse();
    double tmp[3], t_body_in_pose[3];
    q_body_in_pose.rotate(t_pose_in_body_, tmp);
    for (int i = 0; i < 3; ++i) t_body_in_pose[i] = -tmp[i];
    const Quat q_body_in_world = q_pose_in_world * q_body_in_pose;
    double t_body_in_world[3];
    q_pose_in_world.rotate(t_body_in_pose, tmp);
    for (int i = 0; i < 3; ++i) t_body_in_world[i] = tmp[i] + p.trans[i];
    Pose cam;
    cam.q = q_body_in_world * q_cam_in_body_;
    q_body_in_world.rotate(t_cam_in_body_, tmp);
    for (int i = 0; i < 3; ++i) cam.t[i] = tmp[i] + t_body_in_world[i];
    // world convention -> optical: only the world side is re-expressed
    Quat c;
    switch (world_frame_) {
      case RDF: fr->pose_optical = cam; c = Quat(); break;
      case FLU: c = qFluToRdf(); break;
      case FRD: c = qFrdToRdf(); break;
      case RFU: c = qRfuToRdf(); break;
      default: return false;
    }
    if (world_frame_ != RDF) { fr->pose_optical.q = c * cam.q; c.rotate(cam.t, fr->pose_optical.t); }
    fr->rgb_file = rgb_dir_ + "/data/" + rgb.filename;
    fr->has_depth = !depth_dir_.empty();
    fr->depth_file = fr->has_depth ? depth_dir_ + "/data/" + depths_[depth_idxs_[k]].filename : std::string();
    return true;
  }
  bool get(uint32_t* id, AslFrame* fr) {
    if (empty()) return false;
    *id = static_cast<uint32_t>(next_);
    return frame(next_++, fr);
  }

 private:
  static std::string strip(const std::string& p) { return (!p.empty() && p[p.size() - 1] == '/') ? p.substr(0, p.size() - 1) : p; }
  static void readCsv(const std::string& file, std::vector<AslPose>* out) {
    const std::vector<std::string> lines = readLines(file);
    for (size_t k = 1; k < lines.size(); ++k) { AslPose p; if (parseAslPose(lines[k], &p)) out->push_back(p); }
  }
  static void readCsv(const std::string& file, std::vector<AslFile>* out) {
    const std::vector<std::string> lines = readLines(file);
    for (size_t k = 1; k < lines.size(); ++k) { AslFile f; if (parseAslFile(lines[k], &f)) out->push_back(f); }
  }
  static void splitT(const double T[16], Quat* q, double t[3]) {
    const double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    *q = Quat::fromMatrix(R);
    t[0] = T[3]; t[1] = T[7]; t[2] = T[11];
  }
  template <class A> static std::vector<uint64_t> stamps(const std::vector<A>& v) {
    std::vector<uint64_t> s(v.size());
    for (size_t k = 0; k < v.size(); ++k) s[k] = v[k].timestamp;
    return s;
  }
  void associateAll() {
    std::vector<size_t> pose_rgb, rgb_pose, pose_depth, depth_pose;
    associate(stamps(rgbs_), stamps(poses_), &rgb_pose, &pose_rgb);
    if (!depth_dir_.empty()) associate(stamps(depths_), stamps(poses_), &depth_pose, &pose_depth);
    else { pose_depth = pose_rgb; depth_pose = rgb_pose; }
    pose_idxs_.clear();
    std::set_intersection(pose_rgb.begin(), pose_rgb.end(), pose_depth.begin(), pose_depth.end(),
                          std::back_inserter(pose_idxs_));
    const std::unordered_set<size_t> keep(pose_idxs_.begin(), pose_idxs_.end());
    rgb_idxs_.clear(); depth_idxs_.clear();
    for (size_t k = 0; k < pose_rgb.size(); ++k) if (keep.count(pose_rgb[k])) rgb_idxs_.push_back(rgb_pose[k]);
    if (!depth_dir_.empty())
      for (size_t k = 0; k < pose_depth.size(); ++k) if (keep.count(pose_depth[k])) depth_idxs_.push_back(depth_pose[k]);
  }

  bool ok_;
  Frame world_frame_;
  size_t next_;
  int width_, height_;
  double K_[9], D_[5], depth_scale_;
  std::string rgb_dir_, depth_dir_;
  std::vector<AslPose> poses_;
  std::vector<AslFile> rgbs_, depths_;
  std::vector<size_t> pose_idxs_, rgb_idxs_, depth_idxs_;
  Quat q_pose_in_body_, q_cam_in_body_;
  double t_pose_in_body_[3], t_cam_in_body_[3];
};

// The pixels of one frame as the frontends hand them to flame::Flame::update(): the 8-bit gray
// image (undistorted when `cam` is given: reference tum_rgbd_offline_stream.cc:196-200 rectifies both
// images, asl_rgbd_offline_stream.cc:285-287 the colour image only) and, when the frame has one, the
// depth image in metres (raw / depth_scale_factor; `rectify_depth` as the TUM stream does).
// Returns false with *err set when a file is missing or not decodable.
inline bool loadFramePixels(const std::string& rgb_file, const std::string& depth_file, float depth_scale_factor,
                            const images::PlumbBob* cam, bool rectify_depth, int* width, int* height,
                            std::vector<uint8_t>* gray, std::vector<float>* depth_m, std::string* err = nullptr) {
  images::Image rgb;
  if (!images::readImage(rgb_file, &rgb, err)) return false;
  std::vector<uint8_t> g;
  if (!images::toGray8(rgb, &g)) { if (err) *err = "unsupported channel count in " + rgb_file; return false; }
  *width = rgb.width; *height = rgb.height;
  if (cam) {
    gray->resize(g.size());
    images::undistort<uint8_t>(g.data(), rgb.width, rgb.height, 1, *cam, gray->data());
  } else {
    gray->swap(g);
  }
  depth_m->clear();
  if (depth_file.empty()) return true;
  images::Image d;
  if (!images::readImage(depth_file, &d, err)) return false;
  if (d.bit_depth != 16 || d.channels != 1 || d.width != rgb.width || d.height != rgb.height) {
    if (err) *err = "depth image must be 16-bit gray of the colour image's size: " + depth_file;
    return false;
  }
  std::vector<uint16_t> raw;
  if (cam && rectify_depth) {
    raw.resize(d.u16.size());
    images::undistort<uint16_t>(d.u16.data(), d.width, d.height, 1, *cam, raw.data());
  } else {
    raw.swap(d.u16);
  }
  depth_m->resize(raw.size());
  images::depthToFloat(raw.data(), raw.size(), depth_scale_factor, depth_m->data());
  return true;
}

}  // namespace datasets
}  // namespace flame_ros
