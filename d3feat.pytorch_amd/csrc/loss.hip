// Fused descriptor-matching loss: all-pairs L2 distance + circle loss + detector loss, forward and backward.
//
// Replaces reference utils/loss.py: cdist 'euclidean' (:8-44), CircleLoss.forward (:111-141) and
// DetLoss.forward (:149-158) -- ~40 small PyTorch launches and two host syncs per step in the reference.
// M (sampled correspondences) is 128 in training / 64 in validation (config.py:78), so the whole problem
// (M*M = 16k distances) lives in ONE workgroup: distances go to the caller's `dists` buffer (an output of the
// reference API as well), row/column statistics to LDS, and a single launch produces both losses + metrics.
//
// Semantics reproduced literally (including the reference's quirk that masked entries contribute exp(-0) = 1 to
// the log-sum-exp because their detached weight is clamped to 0):
//   D      = sqrt(sum_c (a_i - p_j)^2 + 1e-12)
//   pos    = D - 1e5*neg_mask,  pw = max(0, pos - pos_margin)       T+ = s*(pos - pos_margin)*pw
//   neg    = D + 1e5*(1-neg_mask), nw = max(0, neg_margin - neg)    T- = s*(neg_margin - neg)*nw
//   desc   = mean_i softplus(lse_j T+ + lse_j T-)/s + mean_j softplus(lse_i T+ + lse_i T-)/s
//   det    = mean_i (D_ii - min_j (D + 1e5*I)_ij) * (sa_i + sp_i)
#include "common.hpp"

namespace {

constexpr int kThreads = 1024;
constexpr int kMaxM = 1024;

struct LossParams {
  float s, safe_radius, pos_margin, neg_margin;
};

__device__ __forceinline__ float softplus_t(float x) { return x > 20.0f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoid_sp(float x) { return x > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ void terms(float d, bool negm, const LossParams& P, float& tpos, float& pw, float& tneg,
                                      float& nw) {
  const float pos = d - 1e5f * (negm ? 1.0f : 0.0f);
  pw = fmaxf(0.0f, pos - P.pos_margin);
  tpos = P.s * (pos - P.pos_margin) * pw;
  const float neg = d + 1e5f * (negm ? 0.0f : 1.0f);
  nw = fmaxf(0.0f, P.neg_margin - neg);
  tneg = P.s * (P.neg_margin - neg) * nw;
}

// wave-cooperative statistics of one line of D: element t of the line is D[base + t*stride] (mask alike)
__device__ void line_stats(const float* __restrict__ D, long dbase, long dstride, const uint8_t* __restrict__ negm,
                           long nbase, long nstride, int M, int self, const LossParams& P, float& lse_p, float& lse_n,
                           float& sumd, float& cmin, int& carg) {
  const int lane = threadIdx.x & 63;
  float mp = -INFINITY, mn = -INFINITY, sd = 0.0f, cm = INFINITY;
  int ca = 0x7fffffff;
  for (int t = lane; t < M; t += 64) {
    const float d = D[dbase + t * dstride];
    float tp, pw, tn, nw;
    terms(d, negm[nbase + t * nstride] != 0, P, tp, pw, tn, nw);
    mp = fmaxf(mp, tp);
    mn = fmaxf(mn, tn);
    sd += d;
    const float dm = d + (t == self ? 1e5f : 0.0f);
    if (dm < cm) { cm = dm; ca = t; }
  }
  mp = d3f::wave_max(mp);
  mn = d3f::wave_max(mn);
  sd = d3f::wave_sum(sd);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {  // (value, index) lexicographic min -> first minimal index like torch.min
    const float ov = __shfl_xor(cm, o, 64);
    const int oa = __shfl_xor(ca, o, 64);
    if (ov < cm || (ov == cm && oa < ca)) { cm = ov; ca = oa; }
  }
  float ep = 0.0f, en = 0.0f;
  for (int t = lane; t < M; t += 64) {
    const float d = D[dbase + t * dstride];
    float tp, pw, tn, nw;
    terms(d, negm[nbase + t * nstride] != 0, P, tp, pw, tn, nw);
    ep += expf(tp - mp);
    en += expf(tn - mn);
  }
  ep = d3f::wave_sum(ep);
  en = d3f::wave_sum(en);
  lse_p = mp + logf(ep);
  lse_n = mn + logf(en);
  sumd = sd;
  cmin = cm;
  carg = ca;
}

__device__ float block_sum(float v, float* sh) {
  v = d3f::wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.0f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
  return t;
}

// LDS plan of the cached variants (M <= 128, C <= 64; everything the workgroup touches more than once):
//   fwd: region0 = descriptors a, p as [M][C+1] each (later reused for the mask bytes), region1 = D as [M][M+1]
//   bwd: region0 = a, p as [M][C+1] each, region1 = G as [M][M+1]
// Without CACHE (larger problems) the same code runs on the global buffers.
__host__ __device__ inline size_t cached_lds_bytes(int M, int C) {
  return sizeof(float) * ((size_t)2 * M * (C + 1) + (size_t)M * (M + 1));
}
static bool cache_ok(int M, int C) { return M <= 128 && C <= 64; }

// stats layout (floats): [0,M) lse_pr  [M,2M) lse_nr  [2M,3M) lse_pc  [3M,4M) lse_nc  [4M,5M) cn  [5M,6M) cnarg(int)
template <bool CACHE>
__global__ __launch_bounds__(kThreads) void loss_fwd_kernel(
    const float* __restrict__ a, const float* __restrict__ p, int M, int C, const uint8_t* __restrict__ negm,
    const float* __restrict__ sa, const float* __restrict__ sp, LossParams P, float* __restrict__ D,
    float* __restrict__ fp_out, float* __restrict__ avgneg_out, float* __restrict__ scalars,
    float* __restrict__ stats) {
  __shared__ float sh[16];
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nw = blockDim.x >> 6;
  const int CS = C + 1, DS = M + 1;
  float* al = lds;
  float* pl = lds + (size_t)M * CS;
  float* Dl = lds + (size_t)2 * M * CS;
  uint8_t* nl = (uint8_t*)lds;  // overlays al/pl once the distances exist
  if (CACHE) {
    for (int t = tid; t < M * C; t += blockDim.x) {
      al[(t / C) * CS + t % C] = a[t];
      pl[(t / C) * CS + t % C] = p[t];
    }
    __syncthreads();
  }
  for (int t = tid; t < M * M; t += blockDim.x) {
    const int i = t / M, j = t % M;
    float acc = 0.0f;
    for (int c = 0; c < C; ++c) {
      const float df = CACHE ? al[i * CS + c] - pl[j * CS + c] : a[(size_t)i * C + c] - p[(size_t)j * C + c];
      acc += df * df;
    }
    const float d = sqrtf(acc + 1e-12f);
    D[t] = d;
    if (CACHE) Dl[i * DS + j] = d;
  }
  __threadfence_block();
  __syncthreads();
  if (CACHE) {
    for (int t = tid; t < M * M; t += blockDim.x) nl[t] = negm[t];
    __syncthreads();
  }
  const float* Dr = CACHE ? Dl : D;
  const long ld = CACHE ? DS : M;
  const uint8_t* nr = CACHE ? nl : negm;
  for (int i = wave; i < M; i += nw) {
    float lp, ln, sd, cm; int ca;
    line_stats(Dr, (long)i * ld, 1, nr, (long)i * M, 1, M, i, P, lp, ln, sd, cm, ca);
    if (lane == 0) {
      stats[i] = lp;
      stats[M + i] = ln;
      stats[4 * M + i] = cm;
      ((int*)stats)[5 * M + i] = ca;
      const float fp = Dr[(long)i * ld + i];  // max_j D*I = D_ii (D > 0)
      fp_out[i] = fp;
      avgneg_out[i] = (sd - fp) / (float)(M - 1);
    }
  }
  for (int j = wave; j < M; j += nw) {
    float lp, ln, sd, cm; int ca;
    line_stats(Dr, (long)j, ld, nr, (long)j, M, M, j, P, lp, ln, sd, cm, ca);
    if (lane == 0) {
      stats[2 * M + j] = lp;
      stats[3 * M + j] = ln;
    }
  }
  __threadfence_block();
  __syncthreads();
  float l = 0.0f, dt = 0.0f, ac = 0.0f, fps = 0.0f, ans = 0.0f;
  for (int i = tid; i < M; i += blockDim.x) {
    l += softplus_t(stats[i] + stats[M + i]) / P.s + softplus_t(stats[2 * M + i] + stats[3 * M + i]) / P.s;
    const float diff = fp_out[i] - stats[4 * M + i];
    dt += diff * (sa[i] + sp[i]);
    ac += diff < 0.0f ? 1.0f : 0.0f;
    fps += fp_out[i];
    ans += avgneg_out[i];
  }
  l = block_sum(l, sh);
  dt = block_sum(dt, sh);
  ac = block_sum(ac, sh);
  fps = block_sum(fps, sh);
  ans = block_sum(ans, sh);
  if (tid == 0) {
    scalars[0] = l / (float)M;
    scalars[1] = dt / (float)M;
    scalars[2] = ac * 100.0f / (float)M;
    scalars[3] = fps / (float)M;
    scalars[4] = ans / (float)M;
    scalars[5] = l / (float)M + dt / (float)M;  // desc + det: the step's loss with unit weights (trainer.py:98)
  }
}

template <bool CACHE>
__global__ __launch_bounds__(kThreads) void loss_bwd_kernel(
    const float* __restrict__ a, const float* __restrict__ p, int M, int C, const uint8_t* __restrict__ negm,
    const float* __restrict__ sa, const float* __restrict__ sp, LossParams P, const float* __restrict__ D,
    const float* __restrict__ stats, const float* __restrict__ g_desc, const float* __restrict__ g_det,
    float* __restrict__ G, float* __restrict__ ga, float* __restrict__ gp, float* __restrict__ gsa,
    float* __restrict__ gsp) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int CS = C + 1, DS = M + 1;
  float* al = lds;
  float* pl = lds + (size_t)M * CS;
  float* Gl = lds + (size_t)2 * M * CS;
  const float gd = g_desc ? *g_desc : 0.0f, gt = g_det ? *g_det : 0.0f;
  const float invM = 1.0f / (float)M;
  if (CACHE)
    for (int t = tid; t < M * C; t += blockDim.x) {
      al[(t / C) * CS + t % C] = a[t];
      pl[(t / C) * CS + t % C] = p[t];
    }
  for (int t = tid; t < M * M; t += blockDim.x) {
    const int i = t / M, j = t % M;
    const float d = D[t];
    float tp, pw, tn, nw;
    terms(d, negm[t] != 0, P, tp, pw, tn, nw);
    const float sr = sigmoid_sp(stats[i] + stats[M + i]);
    const float sc = sigmoid_sp(stats[2 * M + j] + stats[3 * M + j]);
    float g = gd * invM * (sr * (expf(tp - stats[i]) * pw - expf(tn - stats[M + i]) * nw) +
                           sc * (expf(tp - stats[2 * M + j]) * pw - expf(tn - stats[3 * M + j]) * nw));
    const float w = gt * invM * (sa[i] + sp[i]);
    if (j == i) g += w;
    if (j == ((const int*)stats)[5 * M + i]) g -= w;
    if (CACHE) Gl[i * DS + j] = g / d;
    else G[t] = g / d;
  }
  __threadfence_block();
  __syncthreads();
  for (int t = tid; t < M * C; t += blockDim.x) {
    const int i = t / C, c = t % C;
    float acc = 0.0f;
    if (CACHE) {
      const float ai = al[i * CS + c];
      for (int j = 0; j < M; ++j) acc += Gl[i * DS + j] * (ai - pl[j * CS + c]);
    } else {
      const float ai = a[t];
      for (int j = 0; j < M; ++j) acc += G[(size_t)i * M + j] * (ai - p[(size_t)j * C + c]);
    }
    ga[t] = acc;
  }
  for (int t = tid; t < M * C; t += blockDim.x) {
    const int j = t / C, c = t % C;
    float acc = 0.0f;
    if (CACHE) {
      const float pj = pl[j * CS + c];
      for (int i = 0; i < M; ++i) acc += Gl[i * DS + j] * (pj - al[i * CS + c]);
    } else {
      const float pj = p[t];
      for (int i = 0; i < M; ++i) acc += G[(size_t)i * M + j] * (pj - a[(size_t)i * C + c]);
    }
    gp[t] = acc;
  }
  for (int i = tid; i < M; i += blockDim.x) {
    const float v = gt * invM * (D[(size_t)i * M + i] - stats[4 * M + i]);
    if (gsa) gsa[i] = v;
    if (gsp) gsp[i] = v;
  }
}

// ---- tiled variants (M <= 128, C <= 64: the training / validation shapes) ------------------------------------------
// The one-workgroup kernels above spend 71 + 55 us per step on ONE of 256 CUs (16k distances x 32 channels through
// LDS, then 256 wave-serial line statistics).  Here the matrix is cut into strips of kLossStrip lines: workgroup g owns
// rows [g L, (g+1) L) AND columns [g L, (g+1) L) -- it computes its row strip and its column strip of D itself (the
// same expression in the same order, so both copies of an element are bit-identical), takes the row statistics from
// the one and the column statistics from the other, and never needs another workgroup's data.  Forward: the scalar
// sums over all lines are left to a second, tiny launch (loss_finalize_kernel) -- a dependent launch costs 1.7 us here
// (profiles/r03_launch_floor.txt), an in-kernel rendezvous of 8 workgroups more.  Backward: ga needs row strips of G, gp
// column strips: no cross-workgroup dependency at all.
constexpr int kLossStrip = 8;

__host__ __device__ inline size_t tiled_lds_bytes(int M, int C) {
  // a, p [M][C+1]; row strip [L][M+1]; column strip [M][L+1]; mask bytes of both strips
  return sizeof(float) * ((size_t)2 * M * (C + 1) + (size_t)kLossStrip * (M + 1) + (size_t)M * (kLossStrip + 1)) +
         (size_t)2 * kLossStrip * M;
}

__global__ __launch_bounds__(256) void loss_fwd_tile_kernel(const float* __restrict__ a, const float* __restrict__ p,
                                                            int M, int C, const uint8_t* __restrict__ negm, LossParams P,
                                                            float* __restrict__ D, float* __restrict__ fp_out,
                                                            float* __restrict__ avgneg_out, float* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int L = kLossStrip;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int CS = C + 1, RS = M + 1, KS = L + 1;
  float* al = lds;
  float* pl = al + (size_t)M * CS;
  float* Dr = pl + (size_t)M * CS;            // [L][RS]  rows l0 .. l0+L-1
  float* Dc = Dr + (size_t)L * RS;            // [M][KS]  columns l0 .. l0+L-1
  uint8_t* mr = (uint8_t*)(Dc + (size_t)M * KS);   // [L][M]
  uint8_t* mc = mr + (size_t)L * M;                // [M][L]
  const int l0 = blockIdx.x * L, nl = min(L, M - l0);
  {  // blockIdx.y = fragment pair of a stacked batch (d3f_circle_det_loss_forward_pairs): its own M x M problem
    const size_t pr = blockIdx.y;
    a += pr * M * C; p += pr * M * C; negm += pr * M * M; D += pr * M * M;
    fp_out += pr * M; avgneg_out += pr * M; stats += pr * 6 * M;
  }
  for (int t = tid; t < M * C; t += 256) {
    al[(t / C) * CS + t % C] = a[t];
    pl[(t / C) * CS + t % C] = p[t];
  }
  for (int t = tid; t < nl * M; t += 256) {
    mr[t] = negm[(size_t)(l0 + t / M) * M + t % M];
    const int i = t / nl, c = t % nl;
    mc[i * L + c] = negm[(size_t)i * M + l0 + c];
  }
  __syncthreads();
  for (int t = tid; t < nl * M; t += 256) {
    {  // row strip
      const int r = t / M, j = t % M;
      float acc = 0.0f;
      for (int c = 0; c < C; ++c) {
        const float df = al[(l0 + r) * CS + c] - pl[j * CS + c];
        acc += df * df;
      }
      const float d = sqrtf(acc + 1e-12f);
      Dr[r * RS + j] = d;
      D[(size_t)(l0 + r) * M + j] = d;
    }
    {  // column strip
      const int i = t / nl, cc = t % nl;
      float acc = 0.0f;
      for (int c = 0; c < C; ++c) {
        const float df = al[i * CS + c] - pl[(l0 + cc) * CS + c];
        acc += df * df;
      }
      Dc[i * KS + cc] = sqrtf(acc + 1e-12f);
    }
  }
  __syncthreads();
  for (int r = wave; r < nl; r += 4) {
    const int i = l0 + r;
    float lp, ln, sd, cm; int ca;
    line_stats(Dr, (long)r * RS, 1, mr, (long)r * M, 1, M, i, P, lp, ln, sd, cm, ca);
    if (lane == 0) {
      stats[i] = lp;
      stats[M + i] = ln;
      stats[4 * M + i] = cm;
      ((int*)stats)[5 * M + i] = ca;
      const float fp = Dr[r * RS + i];  // max_j D*I = D_ii (D > 0)
      fp_out[i] = fp;
      avgneg_out[i] = (sd - fp) / (float)(M - 1);
    }
  }
  for (int cc = wave; cc < nl; cc += 4) {
    const int j = l0 + cc;
    float lp, ln, sd, cm; int ca;
    line_stats(Dc, (long)cc, KS, mc, (long)cc, L, M, j, P, lp, ln, sd, cm, ca);
    if (lane == 0) {
      stats[2 * M + j] = lp;
      stats[3 * M + j] = ln;
    }
  }
}

// scalars of the loss from the per-line statistics (second launch of the tiled forward).  One workgroup walks the
// `pairs` problems of a stacked batch one after the other (pairs <= 32, M <= 128: nothing to spread) and leaves
// total = sum_p (w_desc desc_p + w_det det_p) -- the step's loss, whose gradient is the SUM of the pairs' gradients.
__global__ __launch_bounds__(256) void loss_finalize_kernel(int M, const float* __restrict__ sa, const float* __restrict__ sp,
                                                            LossParams P, const float* __restrict__ fp_out,
                                                            const float* __restrict__ avgneg_out,
                                                            const float* __restrict__ stats, float* __restrict__ scalars,
                                                            int pairs, float w_desc, float w_det,
                                                            float* __restrict__ total) {
  __shared__ float sh[16];
  const int tid = threadIdx.x;
  float tot = 0.0f;
  for (int pr = 0; pr < pairs; ++pr, sa += M, sp += M, fp_out += M, avgneg_out += M, stats += 6 * M, scalars += 6) {
    float l = 0.0f, dt = 0.0f, ac = 0.0f, fps = 0.0f, ans = 0.0f;
    for (int i = tid; i < M; i += blockDim.x) {
      l += softplus_t(stats[i] + stats[M + i]) / P.s + softplus_t(stats[2 * M + i] + stats[3 * M + i]) / P.s;
      const float diff = fp_out[i] - stats[4 * M + i];
      dt += diff * (sa[i] + sp[i]);
      ac += diff < 0.0f ? 1.0f : 0.0f;
      fps += fp_out[i];
      ans += avgneg_out[i];
    }
    l = block_sum(l, sh);
    dt = block_sum(dt, sh);
    ac = block_sum(ac, sh);
    fps = block_sum(fps, sh);
    ans = block_sum(ans, sh);
    if (tid == 0) {
      scalars[0] = l / (float)M;
      scalars[1] = dt / (float)M;
      scalars[2] = ac * 100.0f / (float)M;
      scalars[3] = fps / (float)M;
      scalars[4] = ans / (float)M;
      scalars[5] = l / (float)M + dt / (float)M;  // desc + det: the step's loss with unit weights (trainer.py:98)
    }
    tot += w_desc * (l / (float)M) + w_det * (dt / (float)M);
  }
  if (tid == 0 && total) *total = tot;
}

__device__ __forceinline__ float loss_grad_entry(float d, bool negm, int i, int j, int M, const LossParams& P,
                                                 const float* __restrict__ stats, float gdM, float w) {
  float tp, pw, tn, nw;
  terms(d, negm, P, tp, pw, tn, nw);
  const float sr = sigmoid_sp(stats[i] + stats[M + i]);
  const float sc = sigmoid_sp(stats[2 * M + j] + stats[3 * M + j]);
  float g = gdM * (sr * (expf(tp - stats[i]) * pw - expf(tn - stats[M + i]) * nw) +
                   sc * (expf(tp - stats[2 * M + j]) * pw - expf(tn - stats[3 * M + j]) * nw));
  if (j == i) g += w;
  if (j == ((const int*)stats)[5 * M + i]) g -= w;
  return g / d;
}

__global__ __launch_bounds__(256) void loss_bwd_tile_kernel(const float* __restrict__ a, const float* __restrict__ p,
                                                            int M, int C, const uint8_t* __restrict__ negm,
                                                            const float* __restrict__ sa, const float* __restrict__ sp,
                                                            LossParams P, const float* __restrict__ D,
                                                            const float* __restrict__ stats,
                                                            const float* __restrict__ g_desc, const float* __restrict__ g_det,
                                                            float* __restrict__ ga, float* __restrict__ gp,
                                                            float* __restrict__ gsa, float* __restrict__ gsp,
                                                            float w_desc, float w_det) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int L = kLossStrip;
  const int tid = threadIdx.x;
  const int CS = C + 1, RS = M + 1, KS = L + 1;
  float* al = lds;
  float* pl = al + (size_t)M * CS;
  float* Gr = pl + (size_t)M * CS;            // [L][RS]  rows of G = dL/dD / D
  float* Gc = Gr + (size_t)L * RS;            // [M][KS]  columns of G
  const int l0 = blockIdx.x * L, nl = min(L, M - l0);
  {  // blockIdx.y = fragment pair of a stacked batch
    const size_t pr = blockIdx.y;
    a += pr * M * C; p += pr * M * C; negm += pr * M * M; D += pr * M * M; stats += pr * 6 * M;
    sa += pr * M; sp += pr * M; ga += pr * M * C; gp += pr * M * C;
    if (gsa) gsa += pr * M;
    if (gsp) gsp += pr * M;
  }
  const float gd = (g_desc ? *g_desc : 0.0f) * w_desc, gt = (g_det ? *g_det : 0.0f) * w_det;
  const float invM = 1.0f / (float)M;
  for (int t = tid; t < M * C; t += 256) {
    al[(t / C) * CS + t % C] = a[t];
    pl[(t / C) * CS + t % C] = p[t];
  }
  for (int t = tid; t < nl * M; t += 256) {
    {
      const int r = t / M, j = t % M, i = l0 + r;
      const size_t e = (size_t)i * M + j;
      Gr[r * RS + j] = loss_grad_entry(D[e], negm[e] != 0, i, j, M, P, stats, gd * invM, gt * invM * (sa[i] + sp[i]));
    }
    {
      const int i = t / nl, cc = t % nl, j = l0 + cc;
      const size_t e = (size_t)i * M + j;
      Gc[i * KS + cc] = loss_grad_entry(D[e], negm[e] != 0, i, j, M, P, stats, gd * invM, gt * invM * (sa[i] + sp[i]));
    }
  }
  __syncthreads();
  for (int t = tid; t < nl * C; t += 256) {
    {
      const int r = t / C, c = t % C, i = l0 + r;
      const float ai = al[i * CS + c];
      float acc = 0.0f;
      for (int j = 0; j < M; ++j) acc += Gr[r * RS + j] * (ai - pl[j * CS + c]);
      ga[(size_t)i * C + c] = acc;
    }
    {
      const int cc = t / C, c = t % C, j = l0 + cc;
      const float pj = pl[j * CS + c];
      float acc = 0.0f;
      for (int i = 0; i < M; ++i) acc += Gc[i * KS + cc] * (pj - al[i * CS + c]);
      gp[(size_t)j * C + c] = acc;
    }
  }
  for (int r = tid; r < nl; r += 256) {
    const int i = l0 + r;
    const float v = gt * invM * (D[(size_t)i * M + i] - stats[4 * M + i]);
    if (gsa) gsa[i] = v;
    if (gsp) gsp[i] = v;
  }
}

// ---- keypoint selection + L2 normalisation of the selected descriptors -------------------------------------------
// The reference normalises ALL N descriptors (architectures.py:318, F.normalize) and then indexes the M sampled
// correspondences out of them and out of the scores (trainer.py:91-94): ~10 PyTorch launches forward and ~15 backward
// (two zero-filled [N,C] scatter targets, their sum, the normalisation's backward over all N rows).  Training only
// ever looks at the 2M selected rows, so one launch gathers and normalises them and one scatters the gradient back.
//   out[m,:] = x[idx[m],:] / max(||x[idx[m],:]||, 1e-12)          (torch F.normalize semantics)
__global__ __launch_bounds__(256) void select_normalize_fwd_kernel(const float* __restrict__ x,
                                                                   const float* __restrict__ scores, int N, int C,
                                                                   const int64_t* __restrict__ idx_a,
                                                                   const int64_t* __restrict__ idx_p, int M,
                                                                   const int32_t* __restrict__ p_offset,
                                                                   float* __restrict__ out_a, float* __restrict__ out_p,
                                                                   float* __restrict__ sa, float* __restrict__ sp,
                                                                   int idx_stride, const int32_t* __restrict__ pair_len,
                                                                   int M_pair) {
  const int lane = threadIdx.x & 63;
  const int m2 = blockIdx.x * 4 + (threadIdx.x >> 6);  // 0..2M-1: anchors then positives
  if (m2 >= 2 * M) return;
  const bool pos = m2 >= M;
  const int m = pos ? m2 - M : m2;
  long row = pos ? idx_p[(size_t)m * idx_stride] + (p_offset ? (long)*p_offset : 0) : idx_a[(size_t)m * idx_stride];
  if (pair_len) {  // stacked pairs: rows of pair m / M_pair are local to its own two clouds (2p, 2p + 1 of the stack)
    const int pr = m / M_pair;
    row += d3f::batch_offset(pair_len, 2 * pr + (pos ? 1 : 0));
  }
  row = row < 0 ? 0 : (row >= N ? N - 1 : row);
  float ss = 0.0f;
  for (int c = lane; c < C; c += 64) {
    const float v = x[row * C + c];
    ss += v * v;
  }
  ss = d3f::wave_sum(ss);
  const float denom = fmaxf(sqrtf(ss), 1e-12f);
  float* o = (pos ? out_p : out_a) + (size_t)m * C;
  for (int c = lane; c < C; c += 64) o[c] = x[row * C + c] / denom;
  if (lane == 0) (pos ? sp : sa)[m] = scores[row];
}

// grad_x[row,:] += g/n - y (y.g)/n  with y = x/n the normalised row (n > eps); grad_scores[row] += g_s
__global__ __launch_bounds__(256) void select_normalize_bwd_kernel(const float* __restrict__ x, int N, int C,
                                                                   const int64_t* __restrict__ idx_a,
                                                                   const int64_t* __restrict__ idx_p, int M,
                                                                   const int32_t* __restrict__ p_offset,
                                                                   const float* __restrict__ g_a,
                                                                   const float* __restrict__ g_p,
                                                                   const float* __restrict__ g_sa,
                                                                   const float* __restrict__ g_sp,
                                                                   float* __restrict__ grad_x,
                                                                   float* __restrict__ grad_s, int idx_stride,
                                                                   const int32_t* __restrict__ pair_len, int M_pair) {
  const int lane = threadIdx.x & 63;
  const int m2 = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m2 >= 2 * M) return;
  const bool pos = m2 >= M;
  const int m = pos ? m2 - M : m2;
  long row = pos ? idx_p[(size_t)m * idx_stride] + (p_offset ? (long)*p_offset : 0) : idx_a[(size_t)m * idx_stride];
  if (pair_len) {  // stacked pairs: rows of pair m / M_pair are local to its own two clouds (2p, 2p + 1 of the stack)
    const int pr = m / M_pair;
    row += d3f::batch_offset(pair_len, 2 * pr + (pos ? 1 : 0));
  }
  row = row < 0 ? 0 : (row >= N ? N - 1 : row);
  const float* g = (pos ? g_p : g_a) + (size_t)m * C;
  float ss = 0.0f, dot = 0.0f;
  for (int c = lane; c < C; c += 64) {
    const float v = x[row * C + c];
    ss += v * v;
    dot += v * (g ? g[c] : 0.0f);
  }
  ss = d3f::wave_sum(ss);
  dot = d3f::wave_sum(dot);
  const float n = sqrtf(ss);
  if (g) {
    for (int c = lane; c < C; c += 64) {
      const float v = x[row * C + c];
      // n <= eps: the forward divided by the constant eps
      const float gv = n > 1e-12f ? (g[c] - v * dot / ss) / n : g[c] / 1e-12f;
      atomicAdd(&grad_x[row * C + c], gv);
    }
  }
  const float* gs = pos ? g_sp : g_sa;
  if (lane == 0 && gs && grad_s) atomicAdd(&grad_s[row], gs[m]);
}

}  // namespace

extern "C" {

size_t d3f_circle_det_loss_stats_floats(int M) { return 6 * (size_t)(M > 0 ? M : 1); }
size_t d3f_circle_det_loss_ws_bytes(int M) { return sizeof(float) * (size_t)(M > 0 ? M : 1) * (size_t)(M > 0 ? M : 1); }

static int loss_forward_impl(const float* anchor, const float* positive, int M, int C, int pairs, const uint8_t* neg_mask,
                             const float* anc_score, const float* pos_score, LossParams P, float w_desc, float w_det,
                             float* dists, float* furthest_positive, float* average_negative, float* out_scalars,
                             float* out_total, float* stats, void* stream) {
  if (!anchor || !positive || !neg_mask || !anc_score || !pos_score || !dists || !furthest_positive ||
      !average_negative || !out_scalars || !stats || M < 2 || M > kMaxM || C < 1 || pairs < 1 || pairs > 32)
    return D3F_EINVAL;
  if (cache_ok(M, C)) {
    loss_fwd_tile_kernel<<<dim3(d3f::cdiv(M, kLossStrip), pairs), 256, tiled_lds_bytes(M, C), (hipStream_t)stream>>>(
        anchor, positive, M, C, neg_mask, P, dists, furthest_positive, average_negative, stats);
    loss_finalize_kernel<<<1, 256, 0, (hipStream_t)stream>>>(M, anc_score, pos_score, P, furthest_positive,
                                                             average_negative, stats, out_scalars, pairs, w_desc, w_det,
                                                             out_total);
  } else {
    if (pairs != 1 || out_total) return D3F_EINVAL;   // the one-workgroup form (M > 128) serves a single pair
    loss_fwd_kernel<false><<<1, kThreads, 0, (hipStream_t)stream>>>(anchor, positive, M, C, neg_mask, anc_score,
                                                                     pos_score, P, dists, furthest_positive,
                                                                     average_negative, out_scalars, stats);
  }
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

int d3f_circle_det_loss_forward(const float* anchor, const float* positive, int M, int C, const uint8_t* neg_mask,
                                const float* anc_score, const float* pos_score, float log_scale, float safe_radius,
                                float pos_margin, float neg_margin, float* dists, float* furthest_positive,
                                float* average_negative, float* out_scalars, float* stats, void* stream) {
  LossParams P = {log_scale, safe_radius, pos_margin, neg_margin};
  return loss_forward_impl(anchor, positive, M, C, 1, neg_mask, anc_score, pos_score, P, 1.0f, 1.0f, dists,
                           furthest_positive, average_negative, out_scalars, nullptr, stats, stream);
}

int d3f_circle_det_loss_forward_pairs(const float* anchor, const float* positive, int M, int C, int pairs,
                                      const uint8_t* neg_mask, const float* anc_score, const float* pos_score,
                                      float log_scale, float safe_radius, float pos_margin, float neg_margin,
                                      float w_desc, float w_det, float* dists, float* furthest_positive,
                                      float* average_negative, float* out_scalars, float* out_total, float* stats,
                                      void* stream) {
  if (!out_total || !cache_ok(M, C)) return D3F_EINVAL;
  LossParams P = {log_scale, safe_radius, pos_margin, neg_margin};
  return loss_forward_impl(anchor, positive, M, C, pairs, neg_mask, anc_score, pos_score, P, w_desc, w_det, dists,
                           furthest_positive, average_negative, out_scalars, out_total, stats, stream);
}

int d3f_circle_det_loss_backward(const float* anchor, const float* positive, int M, int C, const uint8_t* neg_mask,
                                 const float* anc_score, const float* pos_score, float log_scale, float safe_radius,
                                 float pos_margin, float neg_margin, const float* dists, const float* stats,
                                 const float* grad_desc, const float* grad_det, float* grad_anchor,
                                 float* grad_positive, float* grad_anc_score, float* grad_pos_score, void* ws,
                                 size_t ws_bytes, void* stream) {
  if (!anchor || !positive || !neg_mask || !anc_score || !pos_score || !dists || !stats || !grad_anchor ||
      !grad_positive || !ws || M < 2 || M > kMaxM || C < 1)
    return D3F_EINVAL;
  if (ws_bytes < d3f_circle_det_loss_ws_bytes(M)) return D3F_EWORKSPACE;
  LossParams P = {log_scale, safe_radius, pos_margin, neg_margin};
  if (cache_ok(M, C))
    loss_bwd_tile_kernel<<<d3f::cdiv(M, kLossStrip), 256, tiled_lds_bytes(M, C), (hipStream_t)stream>>>(
        anchor, positive, M, C, neg_mask, anc_score, pos_score, P, dists, stats, grad_desc, grad_det, grad_anchor,
        grad_positive, grad_anc_score, grad_pos_score, 1.0f, 1.0f);
  else
    loss_bwd_kernel<false><<<1, kThreads, 0, (hipStream_t)stream>>>(anchor, positive, M, C, neg_mask, anc_score,
                                                                     pos_score, P, dists, stats, grad_desc, grad_det,
                                                                     (float*)ws, grad_anchor, grad_positive,
                                                                     grad_anc_score, grad_pos_score);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

int d3f_circle_det_loss_backward_pairs(const float* anchor, const float* positive, int M, int C, int pairs,
                                       const uint8_t* neg_mask, const float* anc_score, const float* pos_score,
                                       float log_scale, float safe_radius, float pos_margin, float neg_margin,
                                       float w_desc, float w_det, const float* dists, const float* stats,
                                       const float* grad_total, float* grad_anchor, float* grad_positive,
                                       float* grad_anc_score, float* grad_pos_score, void* stream) {
  if (!anchor || !positive || !neg_mask || !anc_score || !pos_score || !dists || !stats || !grad_total || !grad_anchor ||
      !grad_positive || M < 2 || C < 1 || pairs < 1 || pairs > 32 || !cache_ok(M, C))
    return D3F_EINVAL;
  LossParams P = {log_scale, safe_radius, pos_margin, neg_margin};
  loss_bwd_tile_kernel<<<dim3(d3f::cdiv(M, kLossStrip), pairs), 256, tiled_lds_bytes(M, C), (hipStream_t)stream>>>(
      anchor, positive, M, C, neg_mask, anc_score, pos_score, P, dists, stats, grad_total, grad_total, grad_anchor,
      grad_positive, grad_anc_score, grad_pos_score, w_desc, w_det);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

/* Sampled-correspondence front end of the loss -- replaces F.normalize over all N descriptors
 * (models/architectures.py:318) followed by the four index selections of trainer.py:91-94.
 * idx_a / idx_p: int64 row indices, element m at idx[m * idx_stride] (idx_stride = 2: the two columns of the [M,2]
 * correspondence table read in place); idx_p is offset by *p_offset, a device int32 = points of the first cloud, or
 * NULL.  out_a/out_p [M,C] normalised descriptors, sa/sp [M] scores. */
int d3f_select_normalize_forward(const float* x, const float* scores, int N, int C, const int64_t* idx_a,
                                 const int64_t* idx_p, int idx_stride, int M, const int32_t* p_offset, float* out_a,
                                 float* out_p, float* sa, float* sp, void* stream) {
  if (!x || !scores || !idx_a || !idx_p || !out_a || !out_p || !sa || !sp || N < 1 || C < 1 || M < 1 || idx_stride < 1)
    return D3F_EINVAL;
  select_normalize_fwd_kernel<<<d3f::cdiv(2 * M, 4), 256, 0, (hipStream_t)stream>>>(x, scores, N, C, idx_a, idx_p, M,
                                                                                      p_offset, out_a, out_p, sa, sp,
                                                                                      idx_stride, nullptr, 1);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

/* grad_x [N,C] and grad_scores [N] must be ONE allocation of N*(C+1) floats starting at grad_x (cleared here with a
 * single fill); g_* may be NULL for outputs that received no gradient. */
int d3f_select_normalize_backward(const float* x, int N, int C, const int64_t* idx_a, const int64_t* idx_p,
                                  int idx_stride, int M, const int32_t* p_offset, const float* g_a, const float* g_p,
                                  const float* g_sa, const float* g_sp, float* grad_x, float* grad_scores,
                                  void* stream) {
  if (!x || !idx_a || !idx_p || !grad_x || !grad_scores || N < 1 || C < 1 || M < 1 || idx_stride < 1 ||
      grad_scores != grad_x + (size_t)N * C)
    return D3F_EINVAL;
  if (d3f::zero_async(grad_x, sizeof(float) * (size_t)N * (C + 1), (hipStream_t)stream) != hipSuccess)
    return D3F_ELAUNCH;
  select_normalize_bwd_kernel<<<d3f::cdiv(2 * M, 4), 256, 0, (hipStream_t)stream>>>(x, N, C, idx_a, idx_p, M, p_offset,
                                                                                      g_a, g_p, g_sa, g_sp, grad_x,
                                                                                      grad_scores, idx_stride, nullptr, 1);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

/* The same for `pairs` fragment pairs stacked into one batch (clouds 2p, 2p+1 of the stack = pair p; len [2 pairs] the
 * level-0 stack lengths on the device): corr [pairs*M, 2] int64 holds every pair's own table (cloud-local rows, as the
 * dataset yields them, trainer.py:91-94), outputs are [pairs*M, ...]. */
int d3f_select_normalize_forward_pairs(const float* x, const float* scores, int N, int C, const int64_t* corr, int M,
                                       int pairs, const int32_t* len, float* out_a, float* out_p, float* sa, float* sp,
                                       void* stream) {
  if (!x || !scores || !corr || !len || !out_a || !out_p || !sa || !sp || N < 1 || C < 1 || M < 1 || pairs < 1 ||
      2 * pairs > D3F_MAX_BATCH)
    return D3F_EINVAL;
  const int T = pairs * M;
  select_normalize_fwd_kernel<<<d3f::cdiv(2 * T, 4), 256, 0, (hipStream_t)stream>>>(x, scores, N, C, corr, corr + 1, T,
                                                                                      nullptr, out_a, out_p, sa, sp, 2,
                                                                                      len, M);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

int d3f_select_normalize_backward_pairs(const float* x, int N, int C, const int64_t* corr, int M, int pairs,
                                        const int32_t* len, const float* g_a, const float* g_p, const float* g_sa,
                                        const float* g_sp, float* grad_x, float* grad_scores, void* stream) {
  if (!x || !corr || !len || !grad_x || !grad_scores || N < 1 || C < 1 || M < 1 || pairs < 1 ||
      2 * pairs > D3F_MAX_BATCH || grad_scores != grad_x + (size_t)N * C)
    return D3F_EINVAL;
  if (d3f::zero_async(grad_x, sizeof(float) * (size_t)N * (C + 1), (hipStream_t)stream) != hipSuccess)
    return D3F_ELAUNCH;
  const int T = pairs * M;
  select_normalize_bwd_kernel<<<d3f::cdiv(2 * T, 4), 256, 0, (hipStream_t)stream>>>(x, N, C, corr, corr + 1, T, nullptr,
                                                                                      g_a, g_p, g_sa, g_sp, grad_x,
                                                                                      grad_scores, 2, len, M);
  D3F_LAUNCH_CHECK();
  return D3F_OK;
}

}  // extern "C"
