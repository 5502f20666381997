// fp32 GEMM arithmetic on the bf16 matrix pipe (gfx950): x = h + m + l with three bf16 parts (8 significand bits each), the
// product of two such operands as 6 bf16 MFMAs (hh, hm, mh, hl, lh, mm; error ~2^-24 per product, as fp32) or 3 (hh, hm, mh;
// ~2^-16).  Measures (1) the error of a K = 288 dot product against fp64 for the fp32 MFMA, the 6-product and the 3-product
// forms, and (2) the register-only issue rate of each form per K = 32 chunk.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/f32_on_bf16 tools/micro/f32_on_bf16.hip && /tmp/f32_on_bf16
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int K = 288;

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  const float r1 = x - (float)h;
  m = (__bf16)r1;
  l = (__bf16)(r1 - (float)m);
}
// A [16][K] row-major, B [K][16] row-major, D [16][16]; one wave.  mode 0: fp32 MFMA, 1: 6 products, 2: 3 products
__global__ void gemm(const float* A, const float* B, float* D, int mode) {
  const int lane = threadIdx.x, i = lane & 15, kq = lane >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (mode == 0) {
    for (int s = 0; s < K / 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * K + 4 * s + kq], B[(4 * s + kq) * 16 + i], acc, 0, 0, 0);
  } else {
    for (int s = 0; s < K / 32; ++s) {
      bf16x8 ah, am, al, bh, bm, bl;
      for (int e = 0; e < 8; ++e) {
        const int k = 32 * s + 8 * kq + e;
        __bf16 h, m, l;
        split3(A[i * K + k], h, m, l); ah[e] = h; am[e] = m; al[e] = l;
        split3(B[k * 16 + i], h, m, l); bh[e] = h; bm[e] = m; bl[e] = l;
      }
      // smallest terms first
      if (mode == 1) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
    }
  }
  for (int r = 0; r < 4; ++r) D[(4 * kq + r) * 16 + i] = acc[r];
}

template <int MODE>
__global__ void rate(float* out, int iters) {
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a[3], b[3];
  for (int p = 0; p < 3; ++p)
    for (int e = 0; e < 8; ++e) { a[p][e] = (__bf16)(float)(threadIdx.x + e + p); b[p][e] = (__bf16)(float)(e + p); }
  const float af = (float)threadIdx.x, bf = 2.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {     // four independent K = 32 chunks (accumulation chains)
      if constexpr (MODE == 0) {
#pragma unroll
        for (int s = 0; s < 8; ++s) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[c], 0, 0, 0);
      } else {
        if constexpr (MODE == 1) {
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc[c], 0, 0, 0);
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc[c], 0, 0, 0);
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc[c], 0, 0, 0);
        }
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc[c], 0, 0, 0);
      }
    }
  }
  f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
  if (s.x == 12345.f) out[threadIdx.x] = s.x + s.y + s.z + s.w;
}
template <int MODE>
void run_rate(float* d, const char* name) {
  const int cus = 256, threads = 512, iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  rate<MODE><<<cus, threads>>>(d, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  rate<MODE><<<cus, threads>>>(d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double chunks = (double)cus * threads / 64 * iters * 4;          // K = 32 chunks of a 16 x 16 tile
  printf("%-28s %.2f ms: %.1f fp32-equivalent TFLOP/s, %.1f cycles per K = 32 chunk per SIMD at 2.4 GHz\n", name, ms,
         chunks * 16 * 16 * 32 * 2 / ms / 1e9, ms * 1e-3 * 2.4e9 / (chunks / (cus * 4)));
}

int main() {
  std::mt19937 g(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> A(16 * K), B(K * 16), D(256);
  for (auto& v : A) v = nd(g);
  for (auto& v : B) v = nd(g) * 0.1f;
  std::vector<double> ref(256);
  double scale = 0;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double s = 0;
      for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * (double)B[k * 16 + j];
      ref[i * 16 + j] = s;
      scale = std::fmax(scale, std::fabs(s));
    }
  float *dA, *dB, *dD;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 1024);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  const char* names[3] = {"fp32 MFMA 16x16x4", "bf16 x 6 products (3 parts)", "bf16 x 3 products (2 parts)"};
  for (int mode = 0; mode < 3; ++mode) {
    gemm<<<1, 64>>>(dA, dB, dD, mode);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    double mx = 0, rms = 0;
    for (int e = 0; e < 256; ++e) { const double d = D[e] - ref[e]; mx = std::fmax(mx, std::fabs(d)); rms += d * d; }
    printf("%-28s K = %d dot products: max error %.2e, rms %.2e of the largest output\n", names[mode], K, mx / scale, std::sqrt(rms / 256) / scale);
  }
  run_rate<0>(dD, names[0]);
  run_rate<1>(dD, names[1]);
  run_rate<2>(dD, names[2]);
  return 0;
}
