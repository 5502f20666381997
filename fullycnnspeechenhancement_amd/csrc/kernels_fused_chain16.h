// The 1x129 output layer of the bf16 mode of R-CED V1 / V2 on the bf16 MFMA (BASELINE config 2).  The layers in front of it
// are kernels_frame16.h (round 6; this file used to hold the 3-frame-tile kernel it replaced).
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_fused_chain.h"

namespace rced {
namespace chain16 {

using chain::f32x2;
using chain::f32x4;
using chain::kF;

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(s16x4 a, s16x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}
// four floats -> four bf16 (round to nearest even, hardware conversion), as the 8-byte value stored in LDS
__device__ __forceinline__ s16x4 to_bf16x4(f32x4 v) {
  const bf16x4 h = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
  return __builtin_bit_cast(s16x4, h);
}

// ---------------------------------------------------------------------------------------------
// The 1x129 output layer of the bf16 variant on the bf16 MFMA (RCED_C16_FINAL16=0: the fp32 kernel of
// kernels_fused_chain.h instead).  Same Toeplitz GEMM  y[f, frame] = b + sum_k A[f, k] h[frame, k],  k = f'*CH + ci,
// with A rounded to bf16 (packed [step][M-tile][lane] x 4 bf16, k = 16 S + 4 kq + j, zero past K) and the hand-off
// tensor h -- fp32 in memory, but every value is already a bf16 (the last fused layer rounds its output) -- converted
// exactly while it is staged: 64 frames x 128 k per chunk arrive as coalesced fp32 pieces one chunk ahead in registers
// and are committed as bf16 into a two-buffer LDS ping-pong with a row stride of 132 bf16 (264 B: the 16 frames of a
// ds_read_b64 start on 16 different even banks).  A streams from L2 two steps ahead.  One step = K 16 = 12 MFMAs of
// 16 cycles per wave (the fp32 kernel: 2 x K 4 = 24 MFMAs of 32 cycles for K 8).
// ---------------------------------------------------------------------------------------------
template <int CH>
struct Final16 {
  static constexpr int kK = kF * CH;
  static constexpr int kSteps = (kK + 15) / 16;
  static constexpr int kMT = 9;
  static constexpr int kPack16 = kSteps * kMT * 64 * 4;                 // bf16 elements
  static constexpr int kChunk = 128, kRow = kChunk + 4, kStepsPer = kChunk / 16;
  static constexpr int kChunks = (kSteps + kStepsPer - 1) / kStepsPer;
  static constexpr int kPiece = kK % 4 == 0 ? 4 : 2;                    // floats per global load (row alignment 16 / 8 B)
  static constexpr int kPieces = kChunk / kPiece;
  static constexpr int kVec = chain::kFinFrames * kPieces;
  static constexpr int kPer = (kVec + chain::kFinThreads - 1) / chain::kFinThreads;
  static_assert(kK % kPiece == 0, "pieces end with the row");
};

template <int CH>
__global__ __launch_bounds__(chain::kFinThreads) void final_gemm16_kernel(const float* __restrict__ h,
                                                                           const unsigned short* __restrict__ apack16, float bias,
                                                                           float* __restrict__ y, int frames) {
  using G = Final16<CH>;
  constexpr int kFrames = chain::kFinFrames, kThr = chain::kFinThreads;
  __shared__ __attribute__((aligned(16))) unsigned short bs[2][kFrames * G::kRow];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, kq = lane >> 4;
  const int f0 = blockIdx.x * kFrames;
  const s16x4* ap = reinterpret_cast<const s16x4*>(apack16) + (wave * 3) * 64 + lane;
  float r[G::kPer][G::kPiece];
  auto fetch = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < G::kPer; ++i) {
      const int q = tid + i * kThr;
      const int fr = f0 + q / G::kPieces, k = chunk * G::kChunk + G::kPiece * (q % G::kPieces);
#pragma unroll
      for (int j = 0; j < G::kPiece; ++j) r[i][j] = 0.f;
      if (q < G::kVec && fr < frames && k < G::kK) {
        const float* src = h + (size_t)fr * G::kK + k;
        if constexpr (G::kPiece == 4) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(src);
          r[i][0] = v.x; r[i][1] = v.y; r[i][2] = v.z; r[i][3] = v.w;
        } else {
          const f32x2 v = *reinterpret_cast<const f32x2*>(src);
          r[i][0] = v.x; r[i][1] = v.y;
        }
      }
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < G::kPer; ++i) {
      const int q = tid + i * kThr;
      if (q < G::kVec) {
        unsigned short* d = bs[buf] + (q / G::kPieces) * G::kRow + G::kPiece * (q % G::kPieces);
        if constexpr (G::kPiece == 4) {
          *reinterpret_cast<s16x4*>(d) = to_bf16x4(f32x4{r[i][0], r[i][1], r[i][2], r[i][3]});
        } else {
          const s16x4 v = to_bf16x4(f32x4{r[i][0], r[i][1], 0.f, 0.f});
          typedef short s16x2 __attribute__((ext_vector_type(2)));
          *reinterpret_cast<s16x2*>(d) = s16x2{v.x, v.y};
        }
      }
    }
  };
  f32x4 acc[4][3];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int m = 0; m < 3; ++m) acc[t][m] = f32x4{bias, bias, bias, bias};
  fetch(0);
  commit(0);
  s16x4 a[3], an[3], an2[3];            // A fragments of steps S, S+1, S+2
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    a[m] = ap[m * 64];
    an[m] = ap[(G::kMT + m) * 64];
    an2[m] = an[m];
  }
  __syncthreads();
  for (int c = 0; c < G::kChunks; ++c) {
    if (c + 1 < G::kChunks) fetch(c + 1);
    const unsigned short* bb = bs[c & 1] + n * G::kRow + 4 * kq;
    const int left = G::kSteps - G::kStepsPer * c;
    const int ns = left < G::kStepsPer ? left : G::kStepsPer;
#pragma unroll
    for (int s = 0; s < G::kStepsPer; ++s) {
      if (s < ns) {
        const int S = G::kStepsPer * c + s;
        if (S + 2 < G::kSteps) {
#pragma unroll
          for (int m = 0; m < 3; ++m) an2[m] = ap[((S + 2) * G::kMT + m) * 64];
        }
        s16x4 b[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) b[t] = *reinterpret_cast<const s16x4*>(bb + 16 * t * G::kRow + 16 * s);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int m = 0; m < 3; ++m) acc[t][m] = mfma16(a[m], b[t], acc[t][m]);
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          a[m] = an[m];
          an[m] = an2[m];
        }
      }
    }
    if (c + 1 < G::kChunks) commit((c + 1) & 1);
    __syncthreads();
  }
  // D row = f = 16*(3*wave+m) + 4*kq + j, column = frame
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int fr = f0 + 16 * t + n;
    if (fr >= frames) continue;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const int f = 16 * (3 * wave + m) + 4 * kq;
      float* yp = y + (size_t)fr * kF + f;
      const f32x4 v = acc[t][m];
      if (f + 0 < kF) yp[0] = v.x;
      if (f + 1 < kF) yp[1] = v.y;
      if (f + 2 < kF) yp[2] = v.z;
      if (f + 3 < kF) yp[3] = v.w;
    }
  }
}

}  // namespace chain16
}  // namespace rced
