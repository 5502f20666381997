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

// sums[c][2] = sum over partials
__global__ __launch_bounds__(kThreads) void reduce_finish(const double* __restrict__ part, int nparts, int C,
                                                           double* __restrict__ sums) {
  __shared__ double s[kThreads];
  const int tid = threadIdx.x, n = 2 * C;            // n <= 64 outputs
  const int lanes = (kThreads / n) * n, i = tid % n, j0 = tid / n, stride = lanes / n;
  double t = 0.0;
  if (tid < lanes)
    for (int k = j0; k < nparts; k += stride) t += part[(size_t)k * n + i];   // coalesced across i
  s[tid] = t;
  __syncthreads();
  if (tid < n) {
    double r = 0.0;
    for (int j = tid; j < lanes; j += n) r += s[j];
    sums[tid] = r;
  }
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
