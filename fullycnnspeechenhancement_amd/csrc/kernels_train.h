// Training step kernels (model_utils/trainer.py:143-192 over the is_training=True
// graph).  Correctness-first, layer by layer, fp32 data with fp64 reductions; the convolutions reuse
// conv_layer_generic (forward, and dgrad with flipped / transposed weights).  Not tuned: the hot path of
// this repo is the inference forward; this exists so that config 5 (fwd + bwd + Adam) runs and is checked.
#pragma once
#include <hip/hip_runtime.h>

namespace rced {
namespace train {

constexpr int kThreads = 256;
constexpr int kMaxC = 32;

// ---- per-channel reductions over pixels --------------------------------------------------------
// For every channel c:  S1 = sum_p a[p,c],  S2 = sum_p a[p,c] * (b[p,c] - mu[c]) * rstd[c].
//   statistics of z:  a = b = z, mu = 0, rstd = 1   ->  S1 = sum z, S2 = sum z^2
//   BN backward:      a = d_u, b = z                 ->  S1 = sum d_u, S2 = sum d_u * zhat
// part [grid][C][2] doubles.  Threads keep a fixed channel (tid % C); fp64 accumulation.
__global__ __launch_bounds__(kThreads) void chan_reduce(const float* __restrict__ a, const float* __restrict__ b,
                                                         const float* __restrict__ mu, const float* __restrict__ rstd,
                                                         size_t P, int C, double* __restrict__ part) {
  __shared__ double s1[kThreads], s2[kThreads];
  const int tid = threadIdx.x;
  const int lanes = (kThreads / C) * C;          // threads that own a channel
  const int c = tid % C;
  double acc1 = 0.0, acc2 = 0.0;
  if (tid < lanes) {
    const float m = mu ? mu[c] : 0.f, r = rstd ? rstd[c] : 1.f;
    const size_t rows_per_iter = (size_t)(lanes / C) * gridDim.x;
    size_t p = (size_t)blockIdx.x * (lanes / C) + tid / C;
    double b1[4] = {0, 0, 0, 0}, b2[4] = {0, 0, 0, 0};   // four rows in flight per thread
    for (; p + 3 * rows_per_iter < P; p += 4 * rows_per_iter) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const size_t q = (p + u * rows_per_iter) * C + c;
        const float av = a[q], bv = b[q];
        b1[u] += (double)av;
        b2[u] += (double)av * (double)((bv - m) * r);
      }
    }
    for (; p < P; p += rows_per_iter) {
      const float av = a[p * C + c], bv = b[p * C + c];
      b1[0] += (double)av;
      b2[0] += (double)av * (double)((bv - m) * r);
    }
    acc1 = (b1[0] + b1[1]) + (b1[2] + b1[3]);
    acc2 = (b2[0] + b2[1]) + (b2[2] + b2[3]);
  }
  s1[tid] = acc1;
  s2[tid] = acc2;
  __syncthreads();
  if (tid < C) {
    double t1 = 0.0, t2 = 0.0;
    for (int j = tid; j < lanes; j += C) { t1 += s1[j]; t2 += s2[j]; }
    part[((size_t)blockIdx.x * C + tid) * 2 + 0] = t1;
    part[((size_t)blockIdx.x * C + tid) * 2 + 1] = t2;
  }
}

// sums[c][2] = sum over partials.  One workgroup per output (launch with 2*C workgroups).
// only_if: a device flag; the launch is a no-op when it is given and zero (the conditional exact recomputation behind
// sums_fix_x, train_api.hip).
__global__ __launch_bounds__(kThreads) void reduce_finish(const double* __restrict__ part, int nparts, int C,
                                                           double* __restrict__ sums, const int* __restrict__ only_if = nullptr) {
  if (only_if && *only_if == 0) return;
  __shared__ double s[kThreads];
  const int tid = threadIdx.x, n = 2 * C, i = blockIdx.x;
  double t = 0.0;
  for (int k = tid; k < nparts; k += kThreads) t += part[(size_t)k * n + i];
  s[tid] = t;
  __syncthreads();
  for (int k = kThreads / 2; k > 0; k >>= 1) {
    if (tid < k) s[tid] += s[tid + k];
    __syncthreads();
  }
  if (tid == 0) sums[i] = s[0];
}

// Batch statistics from (sum z, sum z^2): mu, rstd = 1/sqrt(var_biased + eps); moving statistics with
// momentum 0.99 (moving_variance takes the unbiased batch variance, as TF's fused batch norm does).
__global__ void bn_stats_finish(const double* __restrict__ sums, double P, int C, float eps, float momentum,
                                float* __restrict__ mu, float* __restrict__ rstd, float* __restrict__ moving_mean,
                                float* __restrict__ moving_var) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double m = sums[2 * c] / P;
  double var = sums[2 * c + 1] / P - m * m;
  if (var < 0.0) var = 0.0;
  mu[c] = (float)m;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  const double unbiased = P > 1.0 ? var * P / (P - 1.0) : var;
  moving_mean[c] = (float)(momentum * (double)moving_mean[c] + (1.0 - momentum) * m);
  moving_var[c] = (float)(momentum * (double)moving_var[c] + (1.0 - momentum) * unbiased);
}

// (mean, biased variance) of the batch from (sum z, sum z^2): out[c] = mean, out[C + c] = variance -- what TF's UPDATE_OPS
// would fold into the moving statistics (rced_conv_bn_relu_train hands them to the caller).
__global__ void batch_mean_var(const double* __restrict__ sums, double P, int C, float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double m = sums[2 * c] / P;
  double var = sums[2 * c + 1] / P - m * m;
  if (var < 0.0) var = 0.0;
  out[c] = (float)m;
  out[C + c] = (float)var;
}

// ---- forward elementwise: out = act(gamma*(z-mu)*rstd + beta + skip_pre) + skip_post -------------
__global__ __launch_bounds__(kThreads) void bn_act_fwd(const float* __restrict__ z, const float* __restrict__ mu,
                                                        const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ skip_pre,
                                                        const float* __restrict__ skip_post, int use_act, size_t n, int C,
                                                        float* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kThreads) {
    const int c = (int)(i % C);
    float v = z[i];
    if (mu) v = gamma[c] * ((v - mu[c]) * rstd[c]) + beta[c];
    if (skip_pre) v += skip_pre[i];
    if (use_act) v = fmaxf(v, 0.f);
    if (skip_post) v += skip_post[i];
    out[i] = v;
  }
}

// ---- loss = sum (y - pred)^2 / bs ; g_pred = -2 (y - pred) / bs ----------------------------------
__global__ __launch_bounds__(kThreads) void loss_fwd_bwd(const float* __restrict__ pred, const float* __restrict__ y,
                                                          size_t n, float inv_bs, float* __restrict__ g,
                                                          double* __restrict__ part) {
  __shared__ double s[kThreads];
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kThreads) {
    const float d = y[i] - pred[i];
    acc += (double)d * (double)d;
    g[i] = -2.f * d * inv_bs;
  }
  s[threadIdx.x] = acc;
  __syncthreads();
  for (int k = kThreads / 2; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) s[threadIdx.x] += s[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = s[0];
}

// ---- backward elementwise 1: route the incoming gradient ------------------------------------------
//   g_out = gradient w.r.t. this layer's output.  d_u = g_out * [v > 0] (v recomputed exactly as forward);
//   g_skip_post += g_out;  g_skip_pre += d_u;  d (temp) = d_u.
__global__ __launch_bounds__(kThreads) void bwd_route(const float* __restrict__ g_out, const float* __restrict__ z,
                                                       const float* __restrict__ mu, const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ skip_pre, int use_act, size_t n, int C,
                                                       float* __restrict__ g_skip_pre, float* __restrict__ g_skip_post,
                                                       float* __restrict__ d) {
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kThreads) {
    const int c = (int)(i % C);
    const float go = g_out[i];
    float du = go;
    if (use_act) {
      float v = z[i];
      if (mu) v = gamma[c] * ((v - mu[c]) * rstd[c]) + beta[c];
      if (skip_pre) v += skip_pre[i];
      du = v > 0.f ? go : 0.f;
    }
    if (g_skip_post) g_skip_post[i] += go;
    if (g_skip_pre) g_skip_pre[i] += du;
    d[i] = du;
  }
}

// ---- backward elementwise 2 (BN): dz = gamma*rstd*(d_u - S1/P - zhat*S2/P), in place on d ----------
__global__ __launch_bounds__(kThreads) void bn_bwd_apply(float* __restrict__ d, const float* __restrict__ z,
                                                          const float* __restrict__ mu, const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma, const double* __restrict__ sums,
                                                          double P, size_t n, int C) {
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kThreads) {
    const int c = (int)(i % C);
    const float zh = (z[i] - mu[c]) * rstd[c];
    const float m1 = (float)(sums[2 * c] / P), m2 = (float)(sums[2 * c + 1] / P);
    d[i] = gamma[c] * rstd[c] * (d[i] - m1 - zh * m2);
  }
}

// ---- channel-aligned float2 variants of the three elementwise kernels (even C) ----------------------
// A thread owns one channel pair, c2 = tid % (C/2), and walks pixels row = tid / (C/2) + k * rows: no
// per-element modulo, 8-byte accesses, the per-channel constants live in registers, and the BN-backward
// sums (S1 = sum d_u, S2 = sum d_u * zhat) fall out of bwd_route2 without another pass over d and z.
constexpr int kPairUnroll = 4;   // pixels in flight per thread

struct PairLane {
  int C2, c2, row, rows;
  bool active;
  __device__ PairLane(int C, int tid) : C2(C >> 1), c2(tid % (C >> 1)), row(tid / (C >> 1)), rows(kThreads / (C >> 1)) {
    active = row < rows;
  }
};

__global__ __launch_bounds__(kThreads) void bn_act_fwd2(const float2* __restrict__ z, const float* __restrict__ mu,
                                                         const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float2* __restrict__ skip_pre,
                                                         const float2* __restrict__ skip_post, int use_act, size_t P, int C,
                                                         float2* __restrict__ out, const float* __restrict__ smu = nullptr,
                                                         const float* __restrict__ srstd = nullptr,
                                                         const float* __restrict__ sgamma = nullptr,
                                                         const float* __restrict__ sbeta = nullptr) {
  // smu: skip_post holds the PRE-BatchNorm output of the skip's producer (a plain conv+BN+ReLU layer whose output tensor
  // was never written, train_api.hip `virt`); the skip value is relu(a z + b) with that layer's statistics
  const PairLane L(C, threadIdx.x);
  if (!L.active) return;
  float fa[2] = {1.f, 1.f}, fb[2] = {0.f, 0.f};   // folded BatchNorm: a*z + b (the form every consumer of z uses)
  float ga[2] = {1.f, 1.f}, gb[2] = {0.f, 0.f};
  if (mu)
    for (int k = 0; k < 2; ++k) { const int c = 2 * L.c2 + k; fa[k] = gamma[c] * rstd[c]; fb[k] = beta[c] - fa[k] * mu[c]; }
  if (smu)
    for (int k = 0; k < 2; ++k) { const int c = 2 * L.c2 + k; ga[k] = sgamma[c] * srstd[c]; gb[k] = sbeta[c] - ga[k] * smu[c]; }
  const size_t stride = (size_t)gridDim.x * L.rows;
  for (size_t p = (size_t)blockIdx.x * L.rows + L.row; p < P; p += kPairUnroll * stride) {
    float2 zv[kPairUnroll], s1[kPairUnroll], s2[kPairUnroll];
#pragma unroll
    for (int u = 0; u < kPairUnroll; ++u) {
      const size_t q = (p + u * stride) * L.C2 + L.c2;
      const bool ok = p + u * stride < P;
      zv[u] = ok ? z[q] : float2{0.f, 0.f};
      s1[u] = (ok && skip_pre) ? skip_pre[q] : float2{0.f, 0.f};
      s2[u] = (ok && skip_post) ? skip_post[q] : float2{0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < kPairUnroll; ++u) {
      if (p + u * stride >= P) break;
      float v[2] = {zv[u].x, zv[u].y};
      const float a1[2] = {s1[u].x, s1[u].y}, a2[2] = {s2[u].x, s2[u].y};
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        if (mu) v[k] = fmaf(fa[k], v[k], fb[k]);
        if (skip_pre) v[k] += a1[k];
        if (use_act) v[k] = fmaxf(v[k], 0.f);
        if (skip_post) v[k] += smu ? fmaxf(fmaf(ga[k], a2[k], gb[k]), 0.f) : a2[k];
      }
      out[(p + u * stride) * L.C2 + L.c2] = float2{v[0], v[1]};
    }
  }
}

// part [grid][C][2] doubles (as chan_reduce); pass part = nullptr for layers without BN
__global__ __launch_bounds__(kThreads) void bwd_route2(const float2* __restrict__ g_out, const float2* __restrict__ z,
                                                        const float* __restrict__ mu, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float2* __restrict__ skip_pre, int use_act, size_t P, int C,
                                                        float2* __restrict__ g_skip_pre, float2* __restrict__ g_skip_post,
                                                        float2* __restrict__ d, double* __restrict__ part,
                                                        const int* __restrict__ only_if = nullptr, int skip_first = 0) {
  // skip_first: this launch is the FIRST writer of the skip source's gradient tensor -- it stores instead of adding, so
  // the tensor needs neither a memset nor this kernel's read of it
  if (only_if && *only_if == 0) return;   // wave-uniform: see reduce_finish
  __shared__ double red[kThreads * 4];
  const PairLane L(C, threadIdx.x);
  double acc[2][2] = {{0.0, 0.0}, {0.0, 0.0}};   // [k][S1, S2]
  if (L.active) {
    float m[2] = {0.f, 0.f}, r[2] = {1.f, 1.f}, fa[2] = {1.f, 1.f}, fb[2] = {0.f, 0.f};
    if (mu)
      for (int k = 0; k < 2; ++k) {
        const int c = 2 * L.c2 + k;
        m[k] = mu[c]; r[k] = rstd[c];
        fa[k] = gamma[c] * rstd[c]; fb[k] = beta[c] - fa[k] * mu[c];   // folded form, as in the forward
      }
    const size_t stride = (size_t)gridDim.x * L.rows;
    for (size_t p = (size_t)blockIdx.x * L.rows + L.row; p < P; p += kPairUnroll * stride) {
      float2 gv[kPairUnroll], zv[kPairUnroll], sp[kPairUnroll], gp[kPairUnroll], gq[kPairUnroll];
#pragma unroll
      for (int u = 0; u < kPairUnroll; ++u) {
        const size_t q = (p + u * stride) * L.C2 + L.c2;
        const bool ok = p + u * stride < P;
        gv[u] = ok ? g_out[q] : float2{0.f, 0.f};
        zv[u] = ok ? z[q] : float2{0.f, 0.f};
        sp[u] = (ok && skip_pre) ? skip_pre[q] : float2{0.f, 0.f};
        gp[u] = (ok && g_skip_pre && !skip_first) ? g_skip_pre[q] : float2{0.f, 0.f};
        gq[u] = (ok && g_skip_post && !skip_first) ? g_skip_post[q] : float2{0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < kPairUnroll; ++u) {
        if (p + u * stride >= P) break;
        const size_t q = (p + u * stride) * L.C2 + L.c2;
        const float go[2] = {gv[u].x, gv[u].y}, zz[2] = {zv[u].x, zv[u].y}, sk[2] = {sp[u].x, sp[u].y};
        float du[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          du[k] = go[k];
          if (use_act) {
            float v = zz[k];
            if (mu) v = fmaf(fa[k], v, fb[k]);
            if (skip_pre) v += sk[k];
            du[k] = v > 0.f ? go[k] : 0.f;
          }
          acc[k][0] += (double)du[k];
          acc[k][1] += (double)du[k] * (double)((zz[k] - m[k]) * r[k]);
        }
        if (g_skip_post) g_skip_post[q] = float2{gq[u].x + go[0], gq[u].y + go[1]};
        if (g_skip_pre) g_skip_pre[q] = float2{gp[u].x + du[0], gp[u].y + du[1]};
        if (d) d[q] = float2{du[0], du[1]};      // d == nullptr: the consumers apply the mask themselves
      }
    }
  }
  if (!part) return;
  red[threadIdx.x * 4 + 0] = acc[0][0];
  red[threadIdx.x * 4 + 1] = acc[0][1];
  red[threadIdx.x * 4 + 2] = acc[1][0];
  red[threadIdx.x * 4 + 3] = acc[1][1];
  __syncthreads();
  if ((int)threadIdx.x < L.C2) {
    double t[4] = {0.0, 0.0, 0.0, 0.0};
    for (int rr = 0; rr < L.rows; ++rr)
      for (int j = 0; j < 4; ++j) t[j] += red[(rr * L.C2 + threadIdx.x) * 4 + j];
    double* o = part + ((size_t)blockIdx.x * C + 2 * threadIdx.x) * 2;   // channels 2*c2, 2*c2+1: (S1, S2) each
    o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; o[3] = t[3];
  }
}

__global__ __launch_bounds__(kThreads) void bn_bwd_apply2(float2* __restrict__ d, const float2* __restrict__ z,
                                                           const float* __restrict__ mu, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const double* __restrict__ sums,
                                                           double P, size_t Ppx, int C) {
  const PairLane L(C, threadIdx.x);
  if (!L.active) return;
  float m[2], r[2], gr[2], m1[2], m2[2];
  for (int k = 0; k < 2; ++k) {
    const int c = 2 * L.c2 + k;
    m[k] = mu[c]; r[k] = rstd[c]; gr[k] = gamma[c] * rstd[c];
    m1[k] = (float)(sums[2 * c] / P); m2[k] = (float)(sums[2 * c + 1] / P);
  }
  const size_t stride = (size_t)gridDim.x * L.rows;
  for (size_t p = (size_t)blockIdx.x * L.rows + L.row; p < Ppx; p += kPairUnroll * stride) {
    float2 dv[kPairUnroll], zv[kPairUnroll];
#pragma unroll
    for (int u = 0; u < kPairUnroll; ++u) {
      const size_t q = (p + u * stride) * L.C2 + L.c2;
      const bool ok = p + u * stride < Ppx;
      dv[u] = ok ? d[q] : float2{0.f, 0.f};
      zv[u] = ok ? z[q] : float2{0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < kPairUnroll; ++u) {
      if (p + u * stride >= Ppx) break;
      const float dd[2] = {dv[u].x, dv[u].y}, zz[2] = {zv[u].x, zv[u].y};
      float o[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float zh = (zz[k] - m[k]) * r[k];
        o[k] = gr[k] * (dd[k] - m1[k] - zh * m2[k]);
      }
      d[(p + u * stride) * L.C2 + L.c2] = float2{o[0], o[1]};
    }
  }
}

// ---- weight gradient: dW[i,j,ci,co] = sum_p x[p + (i,j) - pad, ci] * dz[p, co]  (+= via atomics) ----
// One workgroup per group of `frames_per_wg` frames; per frame the kh input rows and the dz row are
// staged in LDS; thread o owns outputs o, o+256, ... of the [kh*kw*cin][cout] gradient and walks the
// 129 bins.  dW is accumulated in registers over the workgroup's frames, then one atomicAdd each.
constexpr int kWgradMaxOut = 28;   // outputs per thread: kh*kw*cin*cout <= 28*256 = 7168 (V2 encode_8: 6325)
__global__ __launch_bounds__(kThreads) void conv_wgrad(const float* __restrict__ x, const float* __restrict__ dz, int T,
                                                        int F, int cin, int cout, int kh, int kw, int frames,
                                                        int frames_per_wg, float* __restrict__ dW) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // xs [kh][F+kw-1][cin], ds [F][cout]
  const int pt = (kh - 1) / 2, pl = (kw - 1) / 2;
  const int W = F + kw - 1, row = W * cin;
  float* xs = lds;
  float* ds = lds + kh * row;
  const int nout = kh * kw * cin * cout;
  float acc[kWgradMaxOut];
#pragma unroll
  for (int k = 0; k < kWgradMaxOut; ++k) acc[k] = 0.f;
  const int fr0 = blockIdx.x * frames_per_wg;
  for (int fr = fr0; fr < min(fr0 + frames_per_wg, frames); ++fr) {
    const int n = fr / T, t = fr - n * T;
    __syncthreads();
    for (int e = threadIdx.x; e < kh * row; e += kThreads) {
      const int i = e / row, r = e - i * row, fw = r / cin, ci = r - fw * cin;
      const int tt = t + i - pt, ff = fw - pl;
      xs[e] = (tt >= 0 && tt < T && ff >= 0 && ff < F) ? x[(((size_t)n * T + tt) * F + ff) * cin + ci] : 0.f;
    }
    for (int e = threadIdx.x; e < F * cout; e += kThreads) ds[e] = dz[(size_t)fr * F * cout + e];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kWgradMaxOut; ++k) {
      const int o = threadIdx.x + k * kThreads;
      if (o < nout) {
        const int co = o % cout, q = o / cout;          // q = (i*kw + j)*cin + ci
        const int ci = q % cin, ij = q / cin, j = ij % kw, i = ij / kw;
        const float* xp = xs + i * row + j * cin + ci;  // bin f reads column f + j
        const float* dp = ds + co;
        float a = 0.f;
        for (int f = 0; f < F; ++f) a = fmaf(xp[f * cin], dp[f * cout], a);
        acc[k] += a;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < kWgradMaxOut; ++k) {
    const int o = threadIdx.x + k * kThreads;
    if (o < nout) atomicAdd(dW + o, acc[k]);
  }
}

// ---- TF-form Adam on the trainable entries of the flat variable blob ------------------------------
__global__ __launch_bounds__(kThreads) void adam_step(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v,
                                                       const unsigned char* __restrict__ trainable, size_t n, float lr_t,
                                                       float b1, float b2, float eps) {
  const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n || !trainable[i]) return;
  const float gi = g[i];
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  p[i] -= lr_t * mi / (sqrtf(vi) + eps);
}

// ---- weight layouts for conv_layer_generic ---------------------------------------------------------
// forward: [K][cout] -> [K][cout4] (zero padded).   dgrad: w_t[(kh-1-i), (kw-1-j), co, ci4] = w[i,j,ci,co].
__global__ void repack_fwd(const float* __restrict__ w, int K, int cout, int cout4, float* __restrict__ out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= K * cout4) return;
  const int k = e / cout4, c = e - k * cout4;
  out[e] = c < cout ? w[k * cout + c] : 0.f;
}
__global__ void repack_dgrad(const float* __restrict__ w, int kh, int kw, int cin, int cout, int cin4,
                             float* __restrict__ out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= kh * kw * cout * cin4) return;
  const int ci = e % cin4, r = e / cin4, co = r % cout, ij = r / cout, j = ij % kw, i = ij / kw;
  out[e] = ci < cin ? w[(((kh - 1 - i) * kw + (kw - 1 - j)) * cin + ci) * cout + co] : 0.f;
}

}  // namespace train
}  // namespace rced
