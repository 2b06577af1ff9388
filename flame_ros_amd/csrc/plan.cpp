// flame_ros_amd/csrc/plan.cpp -- see plan.h.
#include "plan.h"

#include <algorithm>
#include <cmath>
#include <numeric>

#include "../../include/flame_hip.h"

namespace flamehip {
namespace {

// Recursive coordinate bisection of idx[lo,hi) into `leaves` parts of near-equal size; parts are
// emitted in recursion order, which keeps spatial neighbours close in the tile order.
void rcb(const float* pos, std::vector<int32_t>& idx, int lo, int hi, int leaves,
         std::vector<int32_t>* leaf_start) {
  if (leaves <= 1 || hi - lo <= 1) {
    leaf_start->push_back(lo);
    for (int k = 1; k < leaves; ++k) leaf_start->push_back(hi);  // empty extra leaves
    return;
  }
  float mn[2] = {INFINITY, INFINITY}, mx[2] = {-INFINITY, -INFINITY};
  for (int k = lo; k < hi; ++k)
    for (int a = 0; a < 2; ++a) {
      const float p = pos[2 * idx[k] + a];
      mn[a] = std::min(mn[a], p);
      mx[a] = std::max(mx[a], p);
    }
  const int axis = (mx[1] - mn[1] > mx[0] - mn[0]) ? 1 : 0;
  const int l1 = leaves / 2, l2 = leaves - l1;
  const int mid = lo + (int)(((int64_t)(hi - lo) * l1) / leaves);
  std::nth_element(idx.begin() + lo, idx.begin() + mid, idx.begin() + hi,
                   [&](int32_t a, int32_t b) {
                     const float pa = pos[2 * a + axis], pb = pos[2 * b + axis];
                     return pa < pb || (pa == pb && a < b);
                   });
  rcb(pos, idx, lo, mid, l1, leaf_start);
  rcb(pos, idx, mid, hi, l2, leaf_start);
}

struct TileCfg { int nt, ept, vpt; };
// instantiated kernel configurations (must match kernels.hip)
const TileCfg kCfgs[] = {
    {256, 2, 1},  {256, 3, 1},  {256, 4, 1},  {256, 6, 1},  {256, 4, 2},  {256, 6, 2},
    {512, 2, 1},  {512, 3, 1},  {512, 4, 1},  {512, 6, 1},  {512, 4, 2},  {512, 6, 2},
    {1024, 2, 1}, {1024, 3, 1}, {1024, 4, 1}, {1024, 6, 1}, {1024, 4, 2}, {1024, 6, 2},
};

bool pick_cfg(int want_nt, int e_max, int upd_max, TileCfg* out) {
  // first pass: the smallest workgroup that keeps <= 4 edges per thread (no spills, short serial
  // chains); second pass: anything that fits
  for (int pass = 0; pass < 2; ++pass)
    for (const TileCfg& c : kCfgs) {
      if (want_nt && c.nt != want_nt) continue;
      if (pass == 0 && c.ept > 4) continue;
      if ((int64_t)c.nt * c.ept >= e_max && (int64_t)c.nt * c.vpt >= upd_max) {
        *out = c;
        return true;
      }
    }
  return false;
}

}  // namespace

int build_plan(const PlanOptions& opt, int32_t V, int32_t E, int32_t T, const float* pos,
               const int32_t* edges, const float* alpha, const float* beta, const int32_t* tris,
               Plan* out) {
  Plan& P = *out;
  P = Plan();
  P.V = V; P.E = E; P.T = tris ? T : 0;
  for (int32_t e = 0; e < E; ++e) {
    const int32_t i = edges[2 * e], j = edges[2 * e + 1];
    if (i < 0 || j < 0 || i >= V || j >= V || i == j) return FLAME_HIP_ERR_ARG;
  }
  for (int32_t t = 0; t < P.T; ++t)
    for (int k = 0; k < 3; ++k)
      if (tris[3 * t + k] < 0 || tris[3 * t + k] >= V) return FLAME_HIP_ERR_ARG;

  // ---- tile sizing ----
  const int64_t lds_cap = opt.lds_bytes;
  // an isolated single tile holds the whole graph when it fits the largest kernel config
  const bool single_fits = V <= 2048 && E <= 6144 && ((int64_t)V * 16 + (int64_t)E * 24) <= lds_cap;
  // Auto sizing (measured on MI355X, DESIGN.md "Tile sizing"): one tile per CU when the graph
  // allows it (256 CUs), never below 32 own vertices (halo overhead) or above 196 (LDS / threads);
  // when there are more tiles than CUs prefer the shallower halo whose tiles co-reside on a CU.
  const int auto_own = std::max(32, std::min(196, (V + 255) / 256));
  const int auto_depth = (V + auto_own - 1) / auto_own > 256 ? 3 : 4;
  int tile_own = opt.tile_own > 0 ? opt.tile_own : auto_own;
  int depth = opt.tile_depth > 0 ? std::min(opt.tile_depth, kMaxDepth) : auto_depth;
  bool single = (opt.tile_own <= 0 || opt.tile_own >= V) && single_fits;
  if (single) { tile_own = std::max(V, 1); depth = 0; }

  std::vector<int32_t> deg_o(V, 0);
  for (int32_t e = 0; e < E; ++e) { deg_o[edges[2 * e]]++; deg_o[edges[2 * e + 1]]++; }

  for (int attempt = 0; attempt < 6; ++attempt) {
    const int ntiles = V == 0 ? 0 : (V + tile_own - 1) / tile_own;
    // ---- vertex order: RCB leaves = tiles ----
    std::vector<int32_t> idx(V);
    std::iota(idx.begin(), idx.end(), 0);
    std::vector<int32_t> leaf_start;
    if (V > 0) rcb(pos, idx, 0, V, ntiles, &leaf_start);
    leaf_start.push_back(V);
    // inside a tile the order is free (everything lives in LDS): sort by degree so the lanes of a
    // wave walk incidence lists of similar length
    for (int t = 0; t < ntiles; ++t)
      std::sort(idx.begin() + leaf_start[t], idx.begin() + leaf_start[t + 1],
                [&](int32_t a, int32_t b) { return deg_o[a] != deg_o[b] ? deg_o[a] > deg_o[b] : a < b; });
    P.v_i2o = idx;
    P.v_o2i.assign(V, 0);
    for (int32_t k = 0; k < V; ++k) P.v_o2i[idx[k]] = k;
    std::vector<int32_t> tile_of(V);
    for (int t = 0; t < ntiles; ++t)
      for (int k = leaf_start[t]; k < leaf_start[t + 1]; ++k) tile_of[k] = t;

    // ---- edge order: (owner tile of the source, level 0 before level 1, original id) ----
    std::vector<int32_t> eorder(E);
    std::iota(eorder.begin(), eorder.end(), 0);
    auto ekey = [&](int32_t e) {
      const int32_t i = P.v_o2i[edges[2 * e]], j = P.v_o2i[edges[2 * e + 1]];
      const int64_t own = tile_of[i];
      const int64_t lvl = (tile_of[j] == tile_of[i]) ? 0 : 1;
      return (own << 33) | (lvl << 32) | (int64_t)e;
    };
    {
      std::vector<int64_t> keys(E);
      for (int32_t e = 0; e < E; ++e) keys[e] = ekey(e);
      std::sort(keys.begin(), keys.end());
      for (int32_t k = 0; k < E; ++k) eorder[k] = (int32_t)(keys[k] & 0xffffffffll);
    }
    P.e_i2o = eorder;
    P.e_o2i.assign(E, 0);
    for (int32_t k = 0; k < E; ++k) P.e_o2i[eorder[k]] = k;
    P.eij.resize(E);
    P.ew.resize(E);
    for (int32_t k = 0; k < E; ++k) {
      const int32_t e = eorder[k];
      const int32_t io = edges[2 * e], jo = edges[2 * e + 1];
      P.eij[k] = {P.v_o2i[io], P.v_o2i[jo]};
      P.ew[k] = {alpha[e], beta[e], pos[2 * io] - pos[2 * jo], pos[2 * io + 1] - pos[2 * jo + 1]};
    }
    // ---- incidence CSR, ascending ORIGINAL edge id per vertex ----
    P.grow.assign(V + 1, 0);
    for (int32_t e = 0; e < E; ++e) { P.grow[P.eij[e].x + 1]++; P.grow[P.eij[e].y + 1]++; }
    for (int32_t v = 0; v < V; ++v) P.grow[v + 1] += P.grow[v];
    P.ginc.assign(2 * (size_t)E, 0);
    {
      std::vector<int32_t> fill(P.grow.begin(), P.grow.end() - 1);
      for (int32_t eo = 0; eo < E; ++eo) {  // original order => each list ascending in original id
        const int32_t k = P.e_o2i[eo];
        P.ginc[fill[P.eij[k].x]++] = k;
        P.ginc[fill[P.eij[k].y]++] = k | (int32_t)0x80000000;
      }
    }
    // ---- triangles ----
    P.tris.clear(); P.trow.clear(); P.tinc.clear();
    if (P.T > 0) {
      P.tris.resize(3 * (size_t)P.T);
      for (size_t k = 0; k < P.tris.size(); ++k) P.tris[k] = P.v_o2i[tris[k]];
      P.trow.assign(V + 1, 0);
      for (size_t k = 0; k < P.tris.size(); ++k) P.trow[P.tris[k] + 1]++;
      for (int32_t v = 0; v < V; ++v) P.trow[v + 1] += P.trow[v];
      P.tinc.assign(P.tris.size(), 0);
      std::vector<int32_t> fill(P.trow.begin(), P.trow.end() - 1);
      for (int32_t t = 0; t < P.T; ++t)
        for (int k = 0; k < 3; ++k) P.tinc[fill[P.tris[3 * t + k]]++] = t;
    }
    if (opt.path == FLAME_HIP_PATH_GLOBAL) { P.note = "global path requested"; return 0; }

    // ---- tiles ----
    P.tiles.assign(ntiles, TileDesc());
    P.t_vmap.clear(); P.t_emap.clear(); P.t_eij.clear(); P.t_ew.clear(); P.t_srow.clear();
    std::vector<int32_t> stamp(V, -1), ring(V, 0), lidx(V, 0);
    int e_max = 0, upd_max = 0;
    int64_t lds_max = 0;
    bool ok = true;
    // internal edge ranges per owner tile
    std::vector<int32_t> estart(ntiles + 1, 0);
    for (int32_t k = 0; k < E; ++k) estart[tile_of[P.eij[k].x] + 1]++;
    for (int t = 0; t < ntiles; ++t) estart[t + 1] += estart[t];

    std::vector<int32_t> ext, frontier, next;
    struct LE { int32_t level, notown, orig, k; };
    std::vector<LE> les;
    for (int t = 0; t < ntiles && ok; ++t) {
      TileDesc& D = P.tiles[t];
      D.vstart = leaf_start[t];
      D.n_own = leaf_start[t + 1] - leaf_start[t];
      D.depth = depth;
      ext.clear();
      frontier.clear();
      for (int32_t v = D.vstart; v < D.vstart + D.n_own; ++v) {
        stamp[v] = t; ring[v] = 0; lidx[v] = (int32_t)ext.size(); ext.push_back(v); frontier.push_back(v);
      }
      D.ring_end[0] = (int32_t)ext.size();
      for (int r = 1; r <= kMaxDepth; ++r) {
        if (r <= depth) {
          next.clear();
          for (int32_t v : frontier)
            for (int32_t s = P.grow[v]; s < P.grow[v + 1]; ++s) {
              const int32_t k = P.ginc[s] & 0x7fffffff;
              const int32_t u = (P.ginc[s] < 0) ? P.eij[k].x : P.eij[k].y;
              if (stamp[u] != t) { stamp[u] = t; ring[u] = r; next.push_back(u); }
            }
          std::sort(next.begin(), next.end(), [&](int32_t a, int32_t b) {
            const int32_t da = P.grow[a + 1] - P.grow[a], db = P.grow[b + 1] - P.grow[b];
            return da != db ? da > db : a < b;
          });
          for (int32_t u : next) { lidx[u] = (int32_t)ext.size(); ext.push_back(u); }
          frontier.swap(next);
        }
        D.ring_end[r] = (int32_t)ext.size();
      }
      D.n_ext = (int32_t)ext.size();
      // a tile whose halo swallowed nothing (isolated component) behaves like depth 0
      D.n_upd = depth == 0 ? D.n_ext : D.ring_end[depth - 1];
      if (D.n_ext > 65535) { ok = false; break; }
      D.vmap_off = (int32_t)P.t_vmap.size();
      for (int32_t k = 0; k < D.n_ext; ++k) P.t_vmap.push_back(ext[k]);
      // local edges: visit each ext vertex's outgoing (source-role) incidences
      les.clear();
      for (int32_t lv = 0; lv < D.n_ext; ++lv) {
        const int32_t v = ext[lv];
        for (int32_t s = P.grow[v]; s < P.grow[v + 1]; ++s) {
          if (P.ginc[s] < 0) continue;  // v is the target; the source adds it
          const int32_t k = P.ginc[s];
          const int32_t u = P.eij[k].y;
          if (stamp[u] != t) continue;
          const int32_t lvl = std::max(ring[v], ring[u]);
          if (depth > 0 && std::min(ring[v], ring[u]) >= depth) continue;  // feeds no updated vertex
          les.push_back({lvl, ring[v] == 0 ? 0 : 1, P.e_i2o[k], k});
        }
      }
      std::sort(les.begin(), les.end(), [](const LE& a, const LE& b) {
        if (a.level != b.level) return a.level < b.level;
        if (a.level <= 1 && a.notown != b.notown) return a.notown < b.notown;
        return a.orig < b.orig;
      });
      D.e_loc = (int32_t)les.size();
      D.estart = estart[t];
      D.e_own = estart[t + 1] - estart[t];
      // owned edges must be exactly the prefix and in internal order
      for (int32_t le = 0; le < D.e_own; ++le)
        if (le >= D.e_loc || les[le].k != D.estart + le) { ok = false; P.note = "edge order invariant"; }
      if (!ok) break;
      {
        int32_t le = 0;
        for (int l = 0; l <= kMaxDepth; ++l) {
          while (le < D.e_loc && les[le].level <= l) ++le;
          D.level_end[l] = le;
        }
      }
      D.emap_off = (int32_t)P.t_emap.size();
      for (int32_t le = 0; le < D.e_loc; ++le) P.t_emap.push_back(les[le].k);
      // incidence slots of updated vertices, ascending original edge id
      D.srow_off = (int32_t)P.t_srow.size();
      // slot assignment: walk updated vertices, their ginc lists are already in original order
      std::vector<uint16_t> slot_src(D.e_loc, 0xffff), slot_dst(D.e_loc, 0xffff);
      {
        // map internal edge id -> local id through a small sorted table
        std::vector<std::pair<int32_t, int32_t>> tab(D.e_loc);
        for (int32_t le = 0; le < D.e_loc; ++le) tab[le] = {les[le].k, le};
        std::sort(tab.begin(), tab.end());
        // Transposed incidence slots: the 64 vertices a wave updates together form a group; the
        // j-th incidence of lane l lives at base + 64 j + l, so phase P reads are conflict-free.
        int32_t base = 0;
        for (int32_t g0 = 0; g0 < D.n_upd && ok; g0 += 64) {
          const int32_t g1 = std::min(g0 + 64, D.n_upd);
          int32_t width = 0;
          for (int32_t lv = g0; lv < g1; ++lv) width = std::max(width, P.grow[ext[lv] + 1] - P.grow[ext[lv]]);
          if (base + 64 * width + kDummySlots > 65535) { ok = false; break; }
          for (int32_t lv = g0; lv < g1; ++lv) {
            const int32_t v = ext[lv];
            const int32_t deg = P.grow[v + 1] - P.grow[v];
            const int32_t s0 = base + (lv - g0);
            P.t_srow.push_back((uint32_t)s0 | ((uint32_t)deg << 16));
            int32_t j = 0;
            for (int32_t s = P.grow[v]; s < P.grow[v + 1]; ++s, ++j) {
              const int32_t k = P.ginc[s] & 0x7fffffff;
              auto it = std::lower_bound(tab.begin(), tab.end(), std::make_pair(k, (int32_t)-1));
              if (it == tab.end() || it->first != k) { ok = false; P.note = "halo closure invariant"; break; }
              const uint16_t slot = (uint16_t)(s0 + 64 * j);
              if (P.ginc[s] < 0) slot_dst[it->second] = slot; else slot_src[it->second] = slot;
            }
            if (!ok) break;
          }
          base += 64 * width;
        }
        D.nslots = base;
      }
      if (!ok) break;
      D.erec_off = (int32_t)P.t_eij.size();
      for (int32_t le = 0; le < D.e_loc; ++le) {
        const int32_t k = les[le].k;
        const uint32_t li = (uint32_t)lidx[P.eij[k].x], lj = (uint32_t)lidx[P.eij[k].y];
        P.t_eij.push_back({li | (lj << 16), (uint32_t)slot_src[le] | ((uint32_t)slot_dst[le] << 16)});
        P.t_ew.push_back(P.ew[k]);
      }
      e_max = std::max(e_max, D.e_loc);
      upd_max = std::max(upd_max, D.n_ext);  // every local vertex gets a register slot
      lds_max = std::max<int64_t>(lds_max, (int64_t)D.n_ext * 16 + (int64_t)(D.nslots + kDummySlots) * 12);
    }
    TileCfg cfg{};
    if (ok && lds_max > lds_cap) ok = false;
    if (ok && !pick_cfg(opt.tile_threads, e_max, upd_max, &cfg)) ok = false;
    if (ok) {
      P.has_tiles = true;
      P.tile_threads = cfg.nt; P.tile_ept = cfg.ept; P.tile_vpt = cfg.vpt;
      P.tile_depth = depth;
      P.tile_lds_bytes = lds_max;
      return 0;
    }
    // did not fit: shrink the tiles (a single tile becomes a halo'd partition) and retry
    P.tiles.clear();
    if (single) { single = false; tile_own = opt.tile_own > 0 ? opt.tile_own : auto_own; depth = opt.tile_depth > 0 ? std::min(opt.tile_depth, kMaxDepth) : auto_depth; }
    else tile_own = std::max(16, tile_own / 2);
  }
  P.has_tiles = false;
  if (P.note.empty()) P.note = "no tile configuration fits";
  return 0;
}

}  // namespace flamehip
