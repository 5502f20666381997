// Microbenchmark: v_mfma_f32_16x16x4_f32 fed from LDS the way the fused CR-CED kernel's slot streams are:
// per slot R ds_read_b64 (A fragments + B window, conflict-free addresses) prefetched D slots ahead, then M MFMAs
// on NACC accumulation chains, sched_barrier around each slot.  1 or 2 waves per SIMD, one workgroup per CU.
// Answers: does the stream shape itself reach the MFMA issue rate (157.3 TFLOP/s = 100 %)?
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o mfma_lds_rate mfma_lds_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void pin() { __builtin_amdgcn_sched_barrier(0); }
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
// R reads per slot: read 0 = A (weights region), reads 1.. = B windows.  M MFMAs per slot = 2 * (R - 1) * ... simplified:
// M MFMAs use operand (r % R) in turn.
template <int R, int M, int NACC, int D, int VALU>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 32768; i += blockDim.x) lds[i] = 1e-3f * (i & 7);
  __syncthreads();
  constexpr int RING = D + 1, SLOTS = 24;
  const f32x2* base = reinterpret_cast<const f32x2*>(lds) + lane + wave * 64 * R;
  f32x2 op[RING][R];
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  int junk = lane;
  auto load = [&](auto ic) {
    constexpr int i = decltype(ic)::value;
#pragma unroll
    for (int r = 0; r < R; ++r) op[i % RING][r] = base[(i % SLOTS) * 64 * 8 + r * 64];
  };
  for (int it = 0; it < iters; ++it) {
    static_for<0, D>(load);
    static_for<0, SLOTS>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if constexpr (i + D < SLOTS) load(std::integral_constant<int, i + D>{});
      pin();
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const f32x2 a = op[i % RING][0], b = op[i % RING][R > 1 ? 1 + (m / 2) % (R - 1) : 0];
        acc[m % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(m & 1 ? a.y : a.x, m & 1 ? b.y : b.x, acc[m % NACC], 0, 0, 0);
      }
#pragma unroll
      for (int v = 0; v < VALU; ++v) junk = (junk * 3 + v) ^ i;
      pin();
    });
  }
  f32x4 s = acc[0];
  for (int i = 1; i < NACC; ++i) s += acc[i];
  if (s.x == 12345.f || junk == 0x7fffffff) out[threadIdx.x] = s.x + s.y + s.z + s.w + junk;
}
template <int R, int M, int NACC, int D, int VALU>
void run(int threads, float* d) {
  const int cus = 256, iters = 4000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<R, M, NACC, D, VALU>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<R, M, NACC, D, VALU><<<cus, threads, 131072>>>(d, 50);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<R, M, NACC, D, VALU><<<cus, threads, 131072>>>(d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)cus * threads / 64 * iters * 24.0 * M * 2048.0;
  printf("reads/slot=%d mfma/slot=%d chains=%d depth=%d valu/slot=%d waves/SIMD=%d: %.2f ms  %.1f TFLOP/s (%.1f %%)\n", R, M, NACC,
         D, VALU, threads / 256, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
}
int main() {
  float* d;
  hipMalloc(&d, 4096);
  run<3, 4, 2, 2, 0>(512, d); run<3, 4, 2, 2, 0>(256, d);     // layer-2-like slot
  run<3, 4, 2, 4, 0>(512, d); run<3, 4, 2, 4, 0>(256, d);
  run<2, 2, 2, 2, 0>(512, d); run<2, 2, 2, 2, 0>(256, d);     // layer-1-like slot
  run<2, 2, 2, 6, 0>(512, d); run<2, 2, 2, 6, 0>(256, d);
  run<3, 4, 4, 2, 0>(512, d); run<3, 4, 4, 2, 0>(256, d);
  run<6, 16, 8, 1, 0>(512, d); run<6, 16, 8, 1, 0>(256, d);   // round-1 lockstep slot
  run<3, 4, 2, 2, 4>(512, d); run<3, 4, 2, 2, 8>(512, d);     // with VALU riding along
  run<1, 4, 2, 2, 0>(512, d); run<1, 8, 4, 2, 0>(512, d);
  return 0;
}
