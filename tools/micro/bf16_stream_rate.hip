// Microbenchmark for the fused CR-CED form's bf16 streams: per slot R conflict-free ds_read_b128 (prefetched one slot ahead), M
// v_mfma_f32_16x16x32_bf16 on two accumulation chains, V dependent-free VALU (v_add_f32_dpp on the chains' results of the previous
// slot), sched_barrier around each slot; W waves per workgroup (8 = 2 per SIMD, the kernel's; 12 = 3 per SIMD; 16 = 4), one workgroup
// per CU.  100 % = one MFMA per 16 cycles.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o bf16_stream_rate bf16_stream_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void pin() { __builtin_amdgcn_sched_barrier(0); }
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
template <int CTRL>
__device__ __forceinline__ float dpp0(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int R, int M, int V, int W = 8, bool PLAIN = false>   // PLAIN: v_add_f32 instead of v_add_f32_dpp
__global__ __launch_bounds__(64 * W) __attribute__((target("no-packed-fp32-ops"))) void k(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 32768; i += blockDim.x) lds[i] = 1e-3f * (i & 7);
  __syncthreads();
  constexpr int SLOTS = 16, RR = R > 0 ? R : 1;
  const s16x8* base = reinterpret_cast<const s16x8*>(lds) + lane + (wave & 7) * 64;
  s16x8 op[2][RR];
  f32x4 c[2][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};   // [slot & 1][chain]: never reset, so nothing is dead
  float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < RR; ++r) op[0][r] = op[1][r] = base[r * 512];
  auto load = [&](auto ic) {
    constexpr int i = decltype(ic)::value;
#pragma unroll
    for (int r = 0; r < R; ++r) op[i & 1][r] = base[((i * R + r) % 12) * 512];
  };
  for (int it = 0; it < iters; ++it) {
    static_for<0, SLOTS>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      load(std::integral_constant<int, i + 1>{});
      pin();
      f32x4 a0 = c[i & 1][0], a1 = c[i & 1][1];
#pragma unroll
      for (int m = 0; m < M; m += 2) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, op[i & 1][m % RR]), __builtin_bit_cast(bf16x8, op[i & 1][(m + 1) % RR]), a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, op[i & 1][(m + 1) % RR]), __builtin_bit_cast(bf16x8, op[i & 1][m % RR]), a1, 0, 0, 0);
      }
#pragma unroll
      for (int v = 0; v < V; ++v) {
        if constexpr (PLAIN) {
          float t = c[(i & 1) ^ 1][(v >> 2) & 1][v & 3];
          asm volatile("" : "+v"(t));          // (no re-association into fewer adds)
          o[v & 3] += t;
        } else {
          o[v & 3] += dpp0<0x111>(c[(i & 1) ^ 1][(v >> 2) & 1][v & 3]);
        }
      }
      c[i & 1][0] = a0;
      c[i & 1][1] = a1;
      pin();
    });
  }
  const float s = o[0] + o[1] + o[2] + o[3] + c[0][0].x + c[0][1].y + c[1][0].z + c[1][1].w;
  if (s == 12345.f) out[threadIdx.x] = s;
}
template <int R, int M, int V, int W = 8, bool PLAIN = false>
void run(float* d) {
  const int cus = 256, iters = 2000 * 8 / W;       // the same work per SIMD for every W
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<R, M, V, W, PLAIN>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<R, M, V, W, PLAIN><<<cus, 64 * W, 131072>>>(d, 50);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<R, M, V, W, PLAIN><<<cus, 64 * W, 131072>>>(d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfmas = (double)iters * 16.0 * M * (W / 4);   // per SIMD (W / 4 waves)
  const double cyc = ms * 1e-3 * 2.4e9;
  printf("waves/SIMD=%d b128 reads/slot=%2d mfma/slot=%2d %s/slot=%2d: %.2f ms  %.1f cycles per MFMA (16 = peak; at 2.4 GHz)\n", W / 4, R, M, PLAIN ? "v_add_f32" : "dpp-adds", V, ms, cyc / mfmas);
}
int main() {
  float* d;
  hipMalloc(&d, 4096);
  run<0, 12, 4>(d); run<3, 12, 4>(d); run<6, 12, 4>(d); run<9, 12, 4>(d); run<12, 12, 4>(d);
  run<0, 12, 0>(d); run<0, 12, 12>(d); run<0, 12, 24>(d); run<6, 12, 12>(d); run<6, 12, 24>(d); run<12, 12, 24>(d);
  run<0, 24, 0>(d); run<6, 24, 8>(d); run<12, 24, 24>(d);
  // three and four waves per SIMD on the kernel's mix (9 .. 12 reads, 12 MFMAs, 24 VALU per slot = 2 per MFMA)
  run<9, 12, 24, 8>(d); run<9, 12, 24, 12>(d); run<9, 12, 24, 16>(d);
  run<12, 12, 24, 8>(d); run<12, 12, 24, 12>(d); run<12, 12, 24, 16>(d);
  run<0, 12, 24, 8>(d); run<0, 12, 24, 12>(d); run<0, 12, 24, 16>(d);
  // plain VALU instead of DPP
  run<0, 12, 12, 8, true>(d); run<0, 12, 24, 8, true>(d); run<9, 12, 24, 8, true>(d); run<9, 12, 24, 12, true>(d); run<0, 12, 48, 8, true>(d);
  return 0;
}
