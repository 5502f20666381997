// Microbenchmark: does it pay to run the two waves of a SIMD half a layer out of phase?
//
// A "layer" of the fused CR-CED kernel, per wave: head (a block of address VALU, a b128 LDS read, D slots of operand
// prefetch), a stream of NS slots (3 ds_read_b64 + 4 MFMAs on four chains each), tail (ReLU VALU on two accumulator
// tiles, four ds_write_b64, wait), workgroup barrier.  In the product kernel all eight waves do head and tail at the
// same time, so nothing covers them (s_memtime stamps: ~1.6 k cycles per layer beyond the MFMA time).
//   LOCK : as the product kernel: [head, stream, tail, barrier] on all waves.
//   SKEW : waves 4..7 lag by half a layer: every wave also executes a raw s_barrier in the middle of its stream, and
//          waves 4..7 start with one extra barrier -- a wave's end-of-layer barrier is its SIMD partner's mid-stream one.
//   PRIO : SKEW + s_setprio 3 during tail and head (the partner's MFMA stream otherwise starves their VALU instructions).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o skew_barrier skew_barrier.hip
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
__device__ __forceinline__ float relu1(float v) {
  const int b = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, b > 0 ? b : 0);
}

template <int NS, int MODE, int HEADVALU>   // MODE 0 LOCK, 1 SKEW, 2 SKEW + PRIO, 3 LOCK as two independent 4-wave workgroups per CU
__global__ __launch_bounds__(512) void k(float* out, int iters, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < (MODE == 3 ? 16384 : 32768); i += blockDim.x) lds[i] = 1e-3f * (i & 7);
  __syncthreads();
  constexpr int D = 2, RING = D + 1, MID = NS / 2;
  const f32x2* abase = reinterpret_cast<const f32x2*>(lds) + lane;                       // "weights": same for all waves
  const f32x2* bbase = reinterpret_cast<const f32x2*>(lds + 6144) + lane + wave * 1024;   // "activations"
  f32x2* wbase = reinterpret_cast<f32x2*>(lds + 6144) + lane + wave * 1024 + 576;
  const f32x4* shp = reinterpret_cast<const f32x4*>(lds + 4096) + (lane >> 4);
  int junk = lane;
  const unsigned long long t0 = __builtin_readcyclecounter();
  if ((MODE == 1 || MODE == 2) && wave >= 4) __builtin_amdgcn_s_barrier();
  for (int it = 0; it < iters; ++it) {
    // ---- head
    if (MODE == 2) __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int v = 0; v < HEADVALU; ++v) junk = (junk * 3 + v) ^ it;
    const f32x4 sh = *shp;
    f32x2 a[RING], b[RING][2];
    f32x4 acc[2], accb[2];
    auto load = [&](auto ic) {
      constexpr int i = decltype(ic)::value, r = i % RING;
      a[r] = abase[(i % 32) * 64];
      b[r][0] = bbase[(i % 8) * 64 + (junk & 1)];
      b[r][1] = bbase[(i % 8) * 64 + 512 + (junk & 1)];
    };
    static_for<0, D>(load);
    if (MODE == 2) __builtin_amdgcn_s_setprio(0);
    // ---- stream
    static_for<0, NS>([&](auto ic) {
      constexpr int i = decltype(ic)::value, r = i % RING;
      if constexpr (i + D < NS) load(std::integral_constant<int, i + D>{});
      pin();
      if constexpr (i == 0) { acc[0] = acc[1] = sh; accb[0] = accb[1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      if constexpr ((MODE == 1 || MODE == 2) && i == MID) __builtin_amdgcn_s_barrier();
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r].x, b[r][0].x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r].x, b[r][1].x, acc[1], 0, 0, 0);
      accb[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r].y, b[r][0].y, accb[0], 0, 0, 0);
      accb[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r].y, b[r][1].y, accb[1], 0, 0, 0);
      pin();
    });
    // ---- tail
    if (MODE == 2) __builtin_amdgcn_s_setprio(3);
    acc[0] += accb[0];
    acc[1] += accb[1];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      wbase[t * 128] = f32x2{relu1(acc[t].x), relu1(acc[t].y)};
      wbase[t * 128 + 64] = f32x2{relu1(acc[t].z), relu1(acc[t].w)};
    }
    __syncthreads();
  }
  if (MODE == 2) __builtin_amdgcn_s_setprio(0);
  if ((MODE == 1 || MODE == 2) && wave < 4) __builtin_amdgcn_s_barrier();
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (junk == 0x7fffffff) out[threadIdx.x] = junk;
  if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}
template <int NS, int MODE, int HEADVALU>
void run(float* d, unsigned long long* dc) {
  const int cus = MODE == 3 ? 512 : 256, iters = 2000, threads = MODE == 3 ? 256 : 512, ldsb = MODE == 3 ? 65536 : 131072;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<NS, MODE, HEADVALU>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<NS, MODE, HEADVALU><<<cus, threads, ldsb>>>(d, 50, dc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NS, MODE, HEADVALU><<<cus, threads, ldsb>>>(d, iters, dc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c = 0;
  hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
  const double flops = (double)cus * (threads / 64) * iters * NS * 4 * 2048.0;
  const char* names[4] = {"LOCK", "SKEW", "SKEW+PRIO", "2 x 4 waves"};
  printf("slots=%d headvalu=%d %-9s: %.2f ms  %.1f TFLOP/s (%.1f %% of the MFMA rate); per layer %.0f s_memtime ticks, MFMA time alone = %d cycles\n",
         NS, HEADVALU, names[MODE], ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100, (double)c / iters, NS * 4 * 2 * 32);
}
int main() {
  float* d;
  unsigned long long* dc;
  hipMalloc(&d, 4096);
  hipMalloc(&dc, 8);
  run<38, 0, 40>(d, dc); run<38, 1, 40>(d, dc); run<38, 2, 40>(d, dc);    // layer-3-like: 152 MFMAs per wave and layer
  run<24, 0, 40>(d, dc); run<24, 1, 40>(d, dc); run<24, 2, 40>(d, dc);    // shorter layers (layer 1)
  run<38, 0, 0>(d, dc); run<38, 1, 0>(d, dc); run<38, 3, 0>(d, dc);
  run<38, 3, 40>(d, dc); run<24, 3, 40>(d, dc); run<24, 0, 0>(d, dc); run<24, 3, 0>(d, dc);
  return 0;
}
