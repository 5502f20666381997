// Microbenchmark: issue rate of the bf16 MFMAs on gfx950 (no memory traffic).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int KIND>
__global__ void k(float* out, int iters) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  s16x4 a4 = {(short)threadIdx.x, 1, 2, 3}, b4 = {4, 5, 6, (short)threadIdx.x};
  bf16x8 a8, b8;
  for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(float)(threadIdx.x + i); b8[i] = (__bf16)(float)i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if constexpr (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[i], 0, 0, 0);
      }
  }
  f32x4 s = acc[0];
  for (int i = 1; i < 8; ++i) s += acc[i];
  if (s.x == 12345.f) out[threadIdx.x] = s.x + s.y + s.z + s.w;
}
template <int KIND>
void run(int threads, float* d) {
  const int cus = 256, iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<KIND><<<cus, threads>>>(d, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<KIND><<<cus, threads>>>(d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double waves = (double)cus * threads / 64;
  const double kk = KIND == 0 ? 16 : 32;
  const double n = waves * iters * 32.0;
  printf("%s threads/WG=%4d: %.2f ms  %.1f TFLOP/s  %.1f cycles/MFMA/SIMD at 2.4 GHz\n", KIND == 0 ? "16x16x16bf16_1k" : "16x16x32_bf16  ",
         threads, ms, n * 16 * 16 * kk * 2 / ms / 1e9, ms * 1e-3 * 2.4e9 / (n / (cus * 4)));
}
int main() {
  float* d;
  hipMalloc(&d, 4096);
  run<0>(256, d); run<0>(512, d); run<1>(256, d); run<1>(512, d);
  return 0;
}
