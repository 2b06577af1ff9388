// flame_ros_amd/csrc/sync.cpp -- see sync.h.
#include "sync.h"

#include <algorithm>
#include <cmath>

#include "../../include/flame_hip.h"

namespace flamehip {

int graph_sync_host(const flame_hip_sync_params& sp, int32_t V, int32_t T, const float* pos,
                    const float* mu, const float* var, const int32_t* tris, const float* prediction,
                    SyncOut* out) {
  // ---- unique undirected edges of the triangulation, i < j, lexicographic.  Linear time: bucket
  // the 3T (min, max) pairs by min (counting sort), then sort + unique each small bucket. ----
  std::vector<int32_t>& cnt = out->scratch_cnt;
  std::vector<int32_t>& hi = out->scratch_hi;
  cnt.assign((size_t)V + 1, 0);
  for (int32_t t = 0; t < T; ++t) {
    const int32_t a = tris[3 * t], b = tris[3 * t + 1], c = tris[3 * t + 2];
    if (a < 0 || b < 0 || c < 0 || a >= V || b >= V || c >= V || a == b || b == c || a == c)
      return FLAME_HIP_ERR_ARG;
    cnt[std::min(a, b) + 1]++; cnt[std::min(b, c) + 1]++; cnt[std::min(c, a) + 1]++;
  }
  for (int32_t v = 0; v < V; ++v) cnt[v + 1] += cnt[v];
  hi.resize(3 * (size_t)T);
  {
    std::vector<int32_t>& fill = out->scratch_fill;
    fill.assign(cnt.begin(), cnt.end() - 1);
    for (int32_t t = 0; t < T; ++t)
      for (int k = 0; k < 3; ++k) {
        const int32_t a = tris[3 * t + k], b = tris[3 * t + (k + 1) % 3];
        hi[fill[std::min(a, b)]++] = std::max(a, b);
      }
  }
  out->edges.clear();
  out->edges.reserve(3 * (size_t)T);
  for (int32_t v = 0; v < V; ++v) {
    int32_t* b0 = hi.data() + cnt[v];
    int32_t* b1 = hi.data() + cnt[v + 1];
    std::sort(b0, b1);
    b1 = std::unique(b0, b1);
    for (int32_t* p = b0; p < b1; ++p) { out->edges.push_back(v); out->edges.push_back(*p); }
  }
  const int32_t E = (int32_t)(out->edges.size() / 2);
  out->alpha.resize(E);
  const bool custom = sync_weights_custom(sp);
  out->beta.clear();
  if (custom) out->beta.resize(E);
  const int32_t rule = sp.edge_weight_rule;
  for (int32_t e = 0; e < E; ++e) {
    const int32_t i = out->edges[2 * e], j = out->edges[2 * e + 1];
    const float dx = pos[2 * i] - pos[2 * j], dy = pos[2 * i + 1] - pos[2 * j + 1];
    // this file is compiled with -ffp-contract=off: dx*dx + dy*dy is two roundings, as the
    // oracle states it
    const float inv = 1.0f / std::sqrt(dx * dx + dy * dy);
    if (!custom) { out->alpha[e] = inv; continue; }
    float a = (rule == 1 || rule == 3) ? 1.0f : inv;
    float b = (rule == 1 || rule == 2) ? 1.0f : inv;
    if (sp.alpha_gain != 0.0f) a *= sp.alpha_gain;
    if (sp.beta_gain != 0.0f) b *= sp.beta_gain;
    out->alpha[e] = a;
    out->beta[e] = b;
  }
  float scale = 1.0f;
  if (sp.rescale_data && V > 0) {
    double s = 0.0;
    for (int32_t v = 0; v < V; ++v) s += (double)mu[v];
    scale = (float)(s / (double)V);
    if (!(scale > 0.0f)) scale = 1.0f;
  }
  out->z.resize(V); out->wgt.resize(V); out->x0.resize(V);
  for (int32_t v = 0; v < V; ++v) {
    out->z[v] = mu[v] / scale;
    out->wgt[v] = sp.adaptive_data_weights ? 1.0f / var[v] : 1.0f;
    out->x0[v] = (sp.init_with_prediction && prediction && std::isfinite(prediction[v]))
                     ? prediction[v] / scale : out->z[v];
  }
  out->scale = scale;
  return 0;
}

}  // namespace flamehip
