// Microbenchmark: sustained issue rate of v_mfma_f32_16x16x4_f32 with no memory traffic, for 1/2/4 waves per SIMD
// and 2..8 independent accumulators per wave.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k(float* out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x * 1e-9f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  f32x4 s = acc[0];
  for (int i = 1; i < NACC; ++i) s += acc[i];
  if (s.x == 12345.f) out[threadIdx.x] = s.x + s.y + s.z + s.w;
}
template <int NACC>
void run(int threads, int wgs_per_cu, float* d) {
  const int cus = 256, iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC><<<cus * wgs_per_cu, threads>>>(d, 100, 1.f, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NACC><<<cus * wgs_per_cu, threads>>>(d, iters, 1.f, 1.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double waves = (double)cus * wgs_per_cu * threads / 64;
  const double flops = waves * iters * 8.0 * NACC * 2048.0;
  printf("acc=%d threads/WG=%4d WG/CU=%d (waves/SIMD=%.1f): %.2f ms  %.1f TFLOP/s\n", NACC, threads, wgs_per_cu,
         threads / 64.0 * wgs_per_cu / 4, ms, flops / ms / 1e9);
}
int main() {
  float* d;
  hipMalloc(&d, 4096);
  run<8>(256, 1, d); run<8>(512, 1, d); run<8>(1024, 1, d); run<8>(256, 8, d);
  run<4>(256, 1, d); run<4>(512, 1, d); run<4>(1024, 1, d);
  run<2>(256, 1, d); run<2>(512, 1, d); run<2>(1024, 1, d);
  run<1>(256, 1, d); run<1>(512, 1, d); run<1>(1024, 1, d);
  return 0;
}
