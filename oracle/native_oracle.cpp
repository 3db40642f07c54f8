// oracle/native_oracle.cpp -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement (written from the algorithm, not copied) of the two native functions on
// the D3Feat hot path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg may load this library; the product (libd3feat_hip.so) never does.
//
//   orc_radius_*        restates batch_nanoflann_neighbors
//                       (reference cpp_wrappers/cpp_neighbors/neighbors/neighbors.cpp:211-333)
//                       with nanoflann's distance arithmetic
//                       (cpp_wrappers/cpp_utils/nanoflann/nanoflann.hpp:433-441: d2 accumulates
//                       (dx*dx), then += (dy*dy), then += (dz*dz) in float32, no FMA) and result
//                       rule (nanoflann.hpp:249-251,1361: accept iff d2 < r2 strictly;
//                       :208-214,1287: rows sorted by d2).  The kd-tree is NOT restated: the
//                       accepted SET equals brute force; order among equal d2 is unspecified in
//                       the reference (std::sort on distance only) and canonicalised here to
//                       (d2, index) ascending.
//   orc_grid_subsample  restates grid_subsampling / batch_grid_subsampling
//                       (cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp:5-106,
//                       109-211), points-only branch, INCLUDING the row order, which is the
//                       iteration order of a std::unordered_map<size_t, ...> filled in point
//                       order (grid_subsampling.cpp:48,59-70,85).
//
// Build: g++ -O2 -ffp-contract=off (see oracle/Makefile) so no fused multiply-add is formed,
// like the reference's x86-64 distutils build.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <unordered_map>
#include <utility>
#include <vector>

namespace {

struct P3 { float x, y, z; };

inline float sqdist(const P3& a, const P3& b) {
  // nanoflann L2_Simple_Adaptor::evalMetric, dim = 3 (nanoflann.hpp:433-441)
  float d2 = 0.0f;
  float dx = a.x - b.x; d2 += dx * dx;
  float dy = a.y - b.y; d2 += dy * dy;
  float dz = a.z - b.z; d2 += dz * dz;
  return d2;
}

typedef std::pair<float, int> DI;  // (d2, local support index): std::pair's operator< is the canonical order

void rows_brute(const P3* q, int nq, const P3* s, int ns, float r2, std::vector<std::vector<DI>>& rows, int row0) {
  for (int i = 0; i < nq; ++i) {
    std::vector<DI>& row = rows[row0 + i];
    for (int j = 0; j < ns; ++j) {
      float d2 = sqdist(q[i], s[j]);
      if (d2 < r2) row.push_back(DI(d2, j));
    }
    std::sort(row.begin(), row.end());
  }
}

// Uniform cell list with cell edge slightly above the radius; a 27-cell scan then provably
// contains every support with float d2 < r2 (|dx| <= r(1+2e-7) < cell edge, and the rounding of
// the cell coordinate is ~1e-5 cells at these extents).
void rows_grid(const P3* q, int nq, const P3* s, int ns, float radius, float r2,
               std::vector<std::vector<DI>>& rows, int row0) {
  if (ns == 0 || nq == 0) return;
  const double cell = (double)radius * (1.0 + 1e-4);
  double mn[3] = {s[0].x, s[0].y, s[0].z}, mx[3] = {s[0].x, s[0].y, s[0].z};
  for (int j = 0; j < ns; ++j) {
    const float c[3] = {s[j].x, s[j].y, s[j].z};
    for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], (double)c[a]); mx[a] = std::max(mx[a], (double)c[a]); }
  }
  long long dim[3];
  for (int a = 0; a < 3; ++a) dim[a] = (long long)std::floor((mx[a] - mn[a]) / cell) + 1;
  if ((double)dim[0] * (double)dim[1] * (double)dim[2] > 6.4e7) { rows_brute(q, nq, s, ns, r2, rows, row0); return; }
  const long long ncell = dim[0] * dim[1] * dim[2];
  auto cidx = [&](const P3& p, long long c[3]) {
    const float v[3] = {p.x, p.y, p.z};
    for (int a = 0; a < 3; ++a) c[a] = (long long)std::floor(((double)v[a] - mn[a]) / cell);
  };
  std::vector<int> start(ncell + 1, 0), order(ns);
  std::vector<long long> cof(ns);
  for (int j = 0; j < ns; ++j) {
    long long c[3]; cidx(s[j], c);
    cof[j] = c[0] + dim[0] * (c[1] + dim[1] * c[2]);
    start[cof[j] + 1]++;
  }
  for (long long c = 0; c < ncell; ++c) start[c + 1] += start[c];
  std::vector<int> fill(start.begin(), start.end() - 1);
  for (int j = 0; j < ns; ++j) order[fill[cof[j]]++] = j;
  for (int i = 0; i < nq; ++i) {
    std::vector<DI>& row = rows[row0 + i];
    long long c[3]; cidx(q[i], c);
    for (long long z = c[2] - 1; z <= c[2] + 1; ++z) {
      if (z < 0 || z >= dim[2]) continue;
      for (long long y = c[1] - 1; y <= c[1] + 1; ++y) {
        if (y < 0 || y >= dim[1]) continue;
        for (long long x = c[0] - 1; x <= c[0] + 1; ++x) {
          if (x < 0 || x >= dim[0]) continue;
          const long long cc = x + dim[0] * (y + dim[1] * z);
          for (int t = start[cc]; t < start[cc + 1]; ++t) {
            const int j = order[t];
            const float d2 = sqdist(q[i], s[j]);
            if (d2 < r2) row.push_back(DI(d2, j));
          }
        }
      }
    }
    std::sort(row.begin(), row.end());
  }
}

int radius_impl(const float* queries, int Nq, const float* supports, int Ns, const int* q_batches,
                const int* s_batches, int B, float radius, int use_grid, int max_neighbors,
                int** out_idx, float** out_d2, int* out_width, int* out_counts) {
  const P3* q = (const P3*)queries;
  const P3* s = (const P3*)supports;
  const float r2 = radius * radius;  // float32 product, neighbors.cpp:226
  std::vector<std::vector<DI>> rows((size_t)Nq);
  int qo = 0, so = 0;
  std::vector<int> soff((size_t)Nq, 0);
  for (int b = 0; b < B; ++b) {
    if (use_grid) rows_grid(q + qo, q_batches[b], s + so, s_batches[b], radius, r2, rows, qo);
    else rows_brute(q + qo, q_batches[b], s + so, s_batches[b], r2, rows, qo);
    for (int i = 0; i < q_batches[b]; ++i) soff[qo + i] = so;
    qo += q_batches[b];
    so += s_batches[b];
  }
  size_t max_count = 0;
  for (int i = 0; i < Nq; ++i) {
    max_count = std::max(max_count, rows[i].size());
    if (out_counts) out_counts[i] = (int)rows[i].size();
  }
  if (!out_idx) { if (out_width) *out_width = (int)max_count; return 0; }
  // neighbors.cpp:304-326: width = global max count; python then keeps the first max_neighbors
  // columns (datasets/dataloader.py:64-65).
  size_t width = max_count;
  if (max_neighbors > 0 && (size_t)max_neighbors < width) width = (size_t)max_neighbors;
  *out_width = (int)width;
  if (max_count == 0) { *out_idx = nullptr; if (out_d2) *out_d2 = nullptr; return -1; }  // wrapper.cpp:201-205
  *out_idx = (int*)std::malloc(sizeof(int) * (size_t)Nq * width);
  if (out_d2) *out_d2 = (float*)std::malloc(sizeof(float) * (size_t)Nq * width);
  for (int i = 0; i < Nq; ++i) {
    for (size_t j = 0; j < width; ++j) {
      const bool has = j < rows[i].size();
      (*out_idx)[(size_t)i * width + j] = has ? rows[i][j].second + soff[i] : Ns;  // neighbors.cpp:322,324
      if (out_d2) (*out_d2)[(size_t)i * width + j] = has ? rows[i][j].first : INFINITY;
    }
  }
  return 0;
}

struct Acc { int count; float sx, sy, sz; int first; };

}  // namespace

extern "C" {

// method: 0 = brute force, 1 = cell list.  max_neighbors <= 0: full width.
int orc_radius_neighbors(const float* queries, int Nq, const float* supports, int Ns,
                         const int* q_batches, const int* s_batches, int B, float radius, int method,
                         int max_neighbors, int** out_idx, float** out_d2, int* out_width) {
  return radius_impl(queries, Nq, supports, Ns, q_batches, s_batches, B, radius, method, max_neighbors,
                     out_idx, out_d2, out_width, nullptr);
}

// per-query neighbor counts (uncapped) -- what calibrate_neighbors histograms
// (datasets/dataloader.py:203-205).
int orc_radius_counts(const float* queries, int Nq, const float* supports, int Ns, const int* q_batches,
                      const int* s_batches, int B, float radius, int* out_counts) {
  int w = 0;
  return radius_impl(queries, Nq, supports, Ns, q_batches, s_batches, B, radius, 1, 0, nullptr, nullptr, &w,
                     out_counts);
}

// out_points: caller buffer [N*3]; out_batches [B]; optional out_first [N] = index (global) of the first
// input point that fell into the emitted cell, optional out_key [N] = cell key (per element).
// Optional features [N,fdim] / classes [N,ldim] with outputs out_features [N,fdim] / out_classes [N,ldim]: the
// reference's update_all / update_features / update_classes (grid_subsampling.h:42-73) and its emission loop
// (grid_subsampling.cpp:89-102).  Classes of cloud b are sliced at (off*ldim, (off+n)*ldim) -- the reference's own
// end iterator (.cpp:157-158) is wrong for ldim > 1 and b > 0 (undefined behaviour), so parity is anchored at
// ldim = 1 or B = 1.
int orc_grid_subsample_ex(const float* points, int N, const int* batches, int B, float dl, int max_p,
                          const float* features, int fdim, const int* classes, int ldim,
                          float* out_points, int* out_n, int* out_batches, int* out_first, int64_t* out_key,
                          float* out_features, int* out_classes) {
  const P3* p = (const P3*)points;
  int off = 0, n_out = 0;
  for (int b = 0; b < B; ++b) {
    const int n = batches[b];
    const P3* c = p + off;
    if (n == 0) { out_batches[b] = 0; continue; }
    // min_point / max_point: strict comparisons seeded with point 0 (cloud.cpp:27-66)
    P3 mn = c[0], mx = c[0];
    for (int i = 0; i < n; ++i) {
      if (c[i].x < mn.x) mn.x = c[i].x;
      if (c[i].y < mn.y) mn.y = c[i].y;
      if (c[i].z < mn.z) mn.z = c[i].z;
      if (c[i].x > mx.x) mx.x = c[i].x;
      if (c[i].y > mx.y) mx.y = c[i].y;
      if (c[i].z > mx.z) mx.z = c[i].z;
    }
    // origin = floor(min * (1/dl)) * dl, all float32 (grid_subsampling.cpp:27)
    const float inv = 1 / dl;
    P3 org;
    org.x = std::floor(mn.x * inv) * dl;
    org.y = std::floor(mn.y * inv) * dl;
    org.z = std::floor(mn.z * inv) * dl;
    const size_t nX = (size_t)std::floor((mx.x - org.x) / dl) + 1;  // :30
    const size_t nY = (size_t)std::floor((mx.y - org.y) / dl) + 1;  // :31
    std::unordered_map<size_t, Acc> cells;                          // :48
    std::unordered_map<size_t, std::vector<float>> fsum;            // SampledData::features  (grid_subsampling.h:19)
    std::unordered_map<size_t, std::vector<std::unordered_map<int, int>>> votes;   // SampledData::labels (:20)
    for (int i = 0; i < n; ++i) {
      const size_t iX = (size_t)std::floor((c[i].x - org.x) / dl);  // :53-55
      const size_t iY = (size_t)std::floor((c[i].y - org.y) / dl);
      const size_t iZ = (size_t)std::floor((c[i].z - org.z) / dl);
      const size_t key = iX + nX * iY + nX * nY * iZ;               // :56
      auto it = cells.find(key);
      if (it == cells.end()) it = cells.emplace(key, Acc{0, 0.0f, 0.0f, 0.0f, off + i}).first;  // :59-60
      Acc& a = it->second;                                          // grid_subsampling.h:74-79
      a.count += 1;
      a.sx += c[i].x; a.sy += c[i].y; a.sz += c[i].z;
      if (features) {                                               // grid_subsampling.h:50 (float +=, input order)
        std::vector<float>& f = fsum[key];
        if (f.empty()) f.assign((size_t)fdim, 0.0f);
        for (int d = 0; d < fdim; ++d) f[d] += features[(size_t)(off + i) * fdim + d];
      }
      if (classes) {                                                // grid_subsampling.h:51-56
        std::vector<std::unordered_map<int, int>>& v = votes[key];
        if (v.empty()) v.resize((size_t)ldim);
        for (int d = 0; d < ldim; ++d) v[d][classes[(size_t)(off + i) * ldim + d]] += 1;
      }
    }
    int emitted = 0;
    const int limit = (max_p < 1) ? N : max_p;                      // :134-135 (max_p<1 -> N of the whole call)
    for (auto& kv : cells) {                                        // :85 map iteration order
      if (emitted >= limit) break;                                  // :181-204 keep the first max_p rows
      const Acc& a = kv.second;
      const float w = (float)(1.0 / a.count);                       // :87 double reciprocal narrowed by operator*(PointXYZ,float)
      out_points[3 * (size_t)n_out + 0] = a.sx * w;
      out_points[3 * (size_t)n_out + 1] = a.sy * w;
      out_points[3 * (size_t)n_out + 2] = a.sz * w;
      if (features) {                                               // :89-95  f / (float)count
        const float cnt = (float)a.count;
        const std::vector<float>& f = fsum[kv.first];
        for (int d = 0; d < fdim; ++d) out_features[(size_t)n_out * fdim + d] = f[d] / cnt;
      }
      if (classes) {                                                // :97-102 first maximum in map iteration order
        const std::vector<std::unordered_map<int, int>>& v = votes[kv.first];
        for (int d = 0; d < ldim; ++d)
          out_classes[(size_t)n_out * ldim + d] =
              std::max_element(v[d].begin(), v[d].end(), [](const std::pair<const int, int>& x,
                                                            const std::pair<const int, int>& y) {
                return x.second < y.second;
              })->first;
      }
      if (out_first) out_first[n_out] = a.first;
      if (out_key) out_key[n_out] = (int64_t)kv.first;
      ++n_out; ++emitted;
    }
    out_batches[b] = emitted;
    off += n;
  }
  *out_n = n_out;
  return n_out < 1 ? -1 : 0;  // cpp_subsampling/wrapper.cpp:266: empty result is an error
}

int orc_grid_subsample(const float* points, int N, const int* batches, int B, float dl, int max_p,
                       float* out_points, int* out_n, int* out_batches, int* out_first, int64_t* out_key) {
  return orc_grid_subsample_ex(points, N, batches, B, dl, max_p, nullptr, 0, nullptr, 0, out_points, out_n, out_batches,
                               out_first, out_key, nullptr, nullptr);
}

void orc_free(void* p) { std::free(p); }

}  // extern "C"
