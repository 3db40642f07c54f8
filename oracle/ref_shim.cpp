// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// A plain-C entry surface over the REFERENCE's own native sources, which are compiled
// where they lie under /root/reference (see oracle/Makefile: target `ref`).  Nothing of
// the reference is copied into this repository; this file only #includes its headers at
// build time and forwards two calls:
//
//   ref_batch_query      -> batch_nanoflann_neighbors
//                           (cpp_wrappers/cpp_neighbors/neighbors/neighbors.cpp:211-333),
//                           the function cpp_neighbors/wrapper.cpp:198 actually calls.
//   ref_subsample_batch  -> batch_grid_subsampling
//                           (cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.cpp:109-211),
//                           points-only branch, as used by datasets/dataloader.py:16-22.
//
// The marshalling mirrors what the CPython wrappers do (copy into std::vector<PointXYZ>,
// cpp_neighbors/wrapper.cpp:188-191) minus the NumPy C-API, which does not build against
// NumPy 2.x headers.  Output buffers are malloc'd here and released with ref_free().
#include <cstdlib>
#include <cstring>
#include <vector>

#include "cpp_wrappers/cpp_neighbors/neighbors/neighbors.h"
#include "cpp_wrappers/cpp_subsampling/grid_subsampling/grid_subsampling.h"

extern "C" {

// returns 0 on success; -1 if the reference would have raised RuntimeError("Error")
// (empty result, cpp_neighbors/wrapper.cpp:201-205).
int ref_batch_query(const float* queries, int Nq, const float* supports, int Ns,
                    const int* q_batches, const int* s_batches, int B, float radius,
                    int** out_idx, int* out_width) {
  std::vector<PointXYZ> q((const PointXYZ*)queries, (const PointXYZ*)queries + Nq);
  std::vector<PointXYZ> s((const PointXYZ*)supports, (const PointXYZ*)supports + Ns);
  std::vector<int> qb(q_batches, q_batches + B), sb(s_batches, s_batches + B);
  std::vector<int> idx;
  batch_nanoflann_neighbors(q, s, qb, sb, idx, radius);
  if (idx.size() < 1) { *out_idx = nullptr; *out_width = 0; return -1; }
  *out_width = (int)(idx.size() / (size_t)Nq);
  *out_idx = (int*)std::malloc(idx.size() * sizeof(int));
  std::memcpy(*out_idx, idx.data(), idx.size() * sizeof(int));
  return 0;
}

// points-only subsample_batch; out_points is malloc'd [n_out*3], out_batches is caller's [B].
int ref_subsample_batch(const float* points, int N, const int* batches, int B, float sampleDl,
                        int max_p, float** out_points, int* out_n, int* out_batches) {
  std::vector<PointXYZ> p((const PointXYZ*)points, (const PointXYZ*)points + N);
  std::vector<int> b(batches, batches + B);
  std::vector<PointXYZ> sp;
  std::vector<float> f, sf;
  std::vector<int> c, sc, sb;
  batch_grid_subsampling(p, sp, f, sf, c, sc, b, sb, sampleDl, max_p);
  if (sp.size() < 1) { *out_points = nullptr; *out_n = 0; return -1; }
  *out_n = (int)sp.size();
  *out_points = (float*)std::malloc(sp.size() * 3 * sizeof(float));
  std::memcpy(*out_points, sp.data(), sp.size() * 3 * sizeof(float));
  for (int i = 0; i < B; ++i) out_batches[i] = sb[i];
  return 0;
}

// subsample_batch with features [N,fdim] and / or classes [N,ldim] (either may be NULL); outputs malloc'd.
int ref_subsample_batch_ex(const float* points, int N, const int* batches, int B, float sampleDl, int max_p,
                           const float* features, int fdim, const int* classes, int ldim, float** out_points,
                           int* out_n, int* out_batches, float** out_features, int** out_classes) {
  std::vector<PointXYZ> p((const PointXYZ*)points, (const PointXYZ*)points + N);
  std::vector<int> b(batches, batches + B);
  std::vector<PointXYZ> sp;
  std::vector<float> f, sf;
  std::vector<int> c, sc, sb;
  if (features) f.assign(features, features + (size_t)N * fdim);
  if (classes) c.assign(classes, classes + (size_t)N * ldim);
  batch_grid_subsampling(p, sp, f, sf, c, sc, b, sb, sampleDl, max_p);
  *out_points = nullptr; *out_features = nullptr; *out_classes = nullptr;
  if (sp.size() < 1) { *out_n = 0; return -1; }
  *out_n = (int)sp.size();
  *out_points = (float*)std::malloc(sp.size() * 3 * sizeof(float));
  std::memcpy(*out_points, sp.data(), sp.size() * 3 * sizeof(float));
  if (features) {
    *out_features = (float*)std::malloc(sf.size() * sizeof(float));
    std::memcpy(*out_features, sf.data(), sf.size() * sizeof(float));
  }
  if (classes) {
    *out_classes = (int*)std::malloc(sc.size() * sizeof(int));
    std::memcpy(*out_classes, sc.data(), sc.size() * sizeof(int));
  }
  for (int i = 0; i < B; ++i) out_batches[i] = sb[i];
  return 0;
}

void ref_free(void* p) { std::free(p); }

}  // extern "C"
