// Generic direct-convolution layer kernel (any kh, kw, cin, cout): the layerwise path and the
// single-op entry point rced_conv_bn_relu.  One workgroup = one time frame of one utterance;
// the kh input rows it needs are staged in LDS with their SAME-padding zeros; a work item is
// (frequency bin, group of 4 output channels).  This is the simple, always-correct path -- the
// fused MFMA kernels (kernels_fused_*.h) are the fast one.
//
// Semantics: model_utils/module.py:11-34 with BatchNorm (inference) folded by the host into
//   w' = w * gamma/sqrt(var+eps),  shift = (bias - mean) * gamma/sqrt(var+eps) + beta
// so the kernel computes  y = relu?( conv(x, w') + shift + skip_pre ) + skip_post.
#pragma once
#include <hip/hip_runtime.h>

namespace rced {

constexpr int kGenericThreads = 256;

// x [N,T,F,cin], w [kh,kw,cin,cout4] (cout padded to a multiple of 4, zero filled),
// shift [cout4], y [N,T,F,cout].
static __global__ __launch_bounds__(kGenericThreads) void conv_layer_generic(
    const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ w,
    const float* __restrict__ shift, const float* __restrict__ skip_pre,
    const float* __restrict__ skip_post, int T, int F, int cin, int cout, int cout4, int kh,
    int kw, int use_act, int pt, int pl) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [kh][F + kw - 1][cin]
  const int frame = blockIdx.x;  // n*T + t
  const int n = frame / T, t = frame - n * T;
  // pt / pl: zero rows / columns before the window.  TF SAME forward: floor((k-1)/2); the transposed
  // (dgrad) use of this kernel passes the other half.
  const int W = F + kw - 1;
  const int row_elems = W * cin;
  // stage: rows t-pt .. t-pt+kh-1, zero outside [0,T) and in the frequency halo
  for (int e = threadIdx.x; e < kh * row_elems; e += kGenericThreads) {
    const int i = e / row_elems, r = e - i * row_elems;
    const int fw = r / cin, ci = r - fw * cin;
    const int tt = t + i - pt, ff = fw - pl;
    float v = 0.f;
    if (tt >= 0 && tt < T && ff >= 0 && ff < F) v = x[(((size_t)n * T + tt) * F + ff) * cin + ci];
    lds[e] = v;
  }
  __syncthreads();
  const int groups = cout4 >> 2;
  const int items = F * groups;
  for (int it = threadIdx.x; it < items; it += kGenericThreads) {
    const int f = it / groups, g = it - f * groups;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int i = 0; i < kh; ++i) {
      const float* xr = lds + i * row_elems + f * cin;         // window start (f - pl + pl)
      const float* wr = w + (size_t)i * kw * cin * cout4 + g * 4;
      const int klen = kw * cin;                                // contiguous (tap, ci) window
      for (int k = 0; k < klen; ++k) {
        const float xv = xr[k];
        const float4 wv = *reinterpret_cast<const float4*>(wr + (size_t)k * cout4);
        a0 = fmaf(xv, wv.x, a0);
        a1 = fmaf(xv, wv.y, a1);
        a2 = fmaf(xv, wv.z, a2);
        a3 = fmaf(xv, wv.w, a3);
      }
    }
    const float4 sh = *reinterpret_cast<const float4*>(shift + g * 4);
    float v[4] = {a0 + sh.x, a1 + sh.y, a2 + sh.z, a3 + sh.w};
    const size_t o = ((size_t)frame * F + f) * cout + g * 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (g * 4 + c < cout) {
        float r = v[c];
        if (skip_pre) r += skip_pre[o + c];
        if (use_act) r = fmaxf(r, 0.f);
        if (skip_post) r += skip_post[o + c];
        y[o + c] = r;
      }
    }
  }
}

}  // namespace rced
