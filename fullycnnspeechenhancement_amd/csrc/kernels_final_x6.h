// The 1x129 output layer (decode_5 / decode_8 / decode_final: CH -> 1, model.py:24,56,89) in fp32 QUALITY on the bf16 matrix
// pipe.  Same Toeplitz GEMM as chain::final_gemm_lds_kernel,
//     y[frame, f] = b + sum_k A[f, k] h[frame, k],   k = f' * CH + ci,   A[f, k] = W[f' - f + 64, ci],
// but every operand is three bf16 parts, x = h + m + l (8 significand bits each: exact to 2^-24), and a product is the six
// bf16 MFMAs (m m, l h, h l, m h, h m, h h -- smallest first) of K = 32 where the fp32 pipe needs eight of K = 4:
// tools/micro/f32_on_bf16.hip measures 386 fp32-equivalent TFLOP/s against 152.5 for v_mfma_f32_16x16x4_f32, and on K = 288
// dot products an error of 3.9e-7 (max) / 8.9e-8 (rms) of the largest output against 5.6e-7 / 1.2e-7 for the fp32
// instruction.  This kernel is the first product kernel built on it (the layer is a plain dense GEMM bound by the matrix pipe).
//   A: packed [step S][M-tile][part][lane] x 8 bf16, k = 32 S + 8 kq + e, zero past K (host: pack_final_x6 in
//      kernels_fused.hip; training: x6::pack_final_x6_dev every step);
//   B: the 64 frames' h rows arrive as coalesced fp32 pieces one chunk (kChunk k) ahead in registers and are split while
//      they are committed into three bf16 planes of a two-buffer LDS ping-pong; a lane's B fragment is one 16-byte read.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_fused_chain.h"

#ifndef RCED_X6_CHUNK
#define RCED_X6_CHUNK 32     // k per staged chunk (multiple of 32); measured 32 / 64 / 128: 0.317 / 0.327 / 0.440 ms (R-CED V2, batch 256 x 512)
#endif
#ifndef RCED_X6_OCC
#define RCED_X6_OCC 2        // workgroups per CU the register allocator leaves room for
#endif
#ifndef RCED_X6_PREF
#define RCED_X6_PREF 1       // A fragments of step S + 1 fetched into a second register set during step S
#endif

namespace rced {
namespace x6 {
using chain::f32x2;
using chain::f32x4;
using chain::kF;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 mfma32(s16x8 a, s16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// four floats -> their three bf16 parts (round to nearest even each time; x - (h + m + l) is below 2^-24 |x|)
__device__ __forceinline__ void split3(f32x4 v, s16x4& h, s16x4& m, s16x4& l) {
  const bf16x4 bh = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
  const f32x4 r1 = v - f32x4{(float)bh.x, (float)bh.y, (float)bh.z, (float)bh.w};
  const bf16x4 bm = {(__bf16)r1.x, (__bf16)r1.y, (__bf16)r1.z, (__bf16)r1.w};
  const f32x4 r2 = r1 - f32x4{(float)bm.x, (float)bm.y, (float)bm.z, (float)bm.w};
  const bf16x4 bl = {(__bf16)r2.x, (__bf16)r2.y, (__bf16)r2.z, (__bf16)r2.w};
  h = __builtin_bit_cast(s16x4, bh);
  m = __builtin_bit_cast(s16x4, bm);
  l = __builtin_bit_cast(s16x4, bl);
}

template <int CH>
struct FinalX6 {
  static constexpr int kK = kF * CH;
  static constexpr int kSteps = (kK + 31) / 32;
  static constexpr int kMT = 9;
  static constexpr int kPackShorts = kSteps * kMT * 3 * 64 * 8;        // bf16 elements of the A pack
  static constexpr int kChunk = RCED_X6_CHUNK, kRow = kChunk + 8, kStepsPer = kChunk / 32;   // row stride: 16-byte aligned, 16 B off a bank period
  static constexpr int kChunks = (kSteps + kStepsPer - 1) / kStepsPer;
  static constexpr int kPiece = kK % 4 == 0 ? 4 : 2;                     // floats per global load (row alignment 16 / 8 B)
  static constexpr int kPieces = kChunk / kPiece;
  static constexpr int kVec = chain::kFinFrames * kPieces;
  static constexpr int kPer = (kVec + chain::kFinThreads - 1) / chain::kFinThreads;
  static constexpr int kPlane = chain::kFinFrames * kRow;                // bf16 elements of one part of one buffer
  static_assert(kK % kPiece == 0 && kChunk % 32 == 0, "pieces end with the row; whole K = 32 steps per chunk");
};

// w [129][CH] (TF [1,129,CH,1]) -> the three-part A pack, on the device (the training step's weights move every step)
static __global__ void pack_final_x6_dev(const float* __restrict__ w, int CH, unsigned short* __restrict__ pack) {
  const int K = kF * CH, steps = (K + 31) / 32, total = steps * 9 * 64 * 8;      // one thread per (S, mt, lane, e)
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int e = idx & 7, lane = (idx >> 3) & 63, r = idx >> 9, mt = r % 9, S = r / 9;
  const int k = 32 * S + 8 * (lane >> 4) + e, f = 16 * mt + (lane & 15), fp = k / CH, ci = k - fp * CH, tap = fp - f + 64;
  const float v = (k < K && f < kF && tap >= 0 && tap < kF) ? w[tap * CH + ci] : 0.f;
  const __bf16 h = (__bf16)v;
  const float r1 = v - (float)h;
  const __bf16 m = (__bf16)r1;
  const __bf16 l = (__bf16)(r1 - (float)m);
  const size_t base = ((size_t)(S * 9 + mt) * 3) * 512 + lane * 8 + e;
  pack[base] = __builtin_bit_cast(unsigned short, h);
  pack[base + 512] = __builtin_bit_cast(unsigned short, m);
  pack[base + 1024] = __builtin_bit_cast(unsigned short, l);
}

template <int CH>
__global__ __launch_bounds__(chain::kFinThreads, RCED_X6_OCC) void final_gemm_x6_kernel(const float* __restrict__ h,
                                                                            const unsigned short* __restrict__ apack, float bias,
                                                                            float* __restrict__ y, int frames,
                                                                            const float* __restrict__ bias_dev = nullptr) {
  using G = FinalX6<CH>;
  constexpr int kFrames = chain::kFinFrames, kThr = chain::kFinThreads;
  if (bias_dev) bias = *bias_dev;      // the training step's bias is a device variable
  __shared__ __attribute__((aligned(16))) unsigned short bs[2][3 * G::kPlane];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, kq = lane >> 4;
  const int f0 = blockIdx.x * kFrames;
  // A fragment (S, M-tile 3 wave + m, part p): 16 bytes per lane
  const s16x8* ap = reinterpret_cast<const s16x8*>(apack) + (size_t)(wave * 3) * 3 * 64 + lane;
  f32x4 r[G::kPer];
  auto fetch = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < G::kPer; ++i) {
      const int q = tid + i * kThr;
      const int fr = f0 + q / G::kPieces, k = chunk * G::kChunk + G::kPiece * (q % G::kPieces);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (q < G::kVec && fr < frames && k < G::kK) {
        const float* src = h + (size_t)fr * G::kK + k;
        if constexpr (G::kPiece == 4) {
          v = *reinterpret_cast<const f32x4*>(src);
        } else {
          const f32x2 w2 = *reinterpret_cast<const f32x2*>(src);
          v.x = w2.x;
          v.y = w2.y;
        }
      }
      r[i] = v;
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < G::kPer; ++i) {
      const int q = tid + i * kThr;
      if (q < G::kVec) {
        s16x4 ph, pm, pl;
        split3(r[i], ph, pm, pl);
        unsigned short* d = bs[buf] + (q / G::kPieces) * G::kRow + G::kPiece * (q % G::kPieces);
        if constexpr (G::kPiece == 4) {
          *reinterpret_cast<s16x4*>(d) = ph;
          *reinterpret_cast<s16x4*>(d + G::kPlane) = pm;
          *reinterpret_cast<s16x4*>(d + 2 * G::kPlane) = pl;
        } else {
          typedef short s16x2 __attribute__((ext_vector_type(2)));
          *reinterpret_cast<s16x2*>(d) = s16x2{ph.x, ph.y};
          *reinterpret_cast<s16x2*>(d + G::kPlane) = s16x2{pm.x, pm.y};
          *reinterpret_cast<s16x2*>(d + 2 * G::kPlane) = s16x2{pl.x, pl.y};
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
  s16x8 a[3][3], an[3][3];            // [M-tile][part] of steps S and S + 1
#pragma unroll
  for (int m = 0; m < 3; ++m)
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      a[m][p] = ap[(m * 3 + p) * 64];
      an[m][p] = a[m][p];
    }
  __syncthreads();
  for (int c = 0; c < G::kChunks; ++c) {
    if (c + 1 < G::kChunks) fetch(c + 1);
    const unsigned short* bb = bs[c & 1] + n * G::kRow + 8 * kq;
#pragma unroll
    for (int s = 0; s < G::kStepsPer; ++s) {
      const int S = G::kStepsPer * c + s;
      if (S < G::kSteps) {          // (wave-uniform; the last chunk may hold fewer steps)
        if (RCED_X6_PREF && S + 1 < G::kSteps) {
#pragma unroll
          for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int p = 0; p < 3; ++p) an[m][p] = ap[(((size_t)(S + 1) * G::kMT + m) * 3 + p) * 64];
        }
        if (!RCED_X6_PREF && S > 0) {
#pragma unroll
          for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int p = 0; p < 3; ++p) a[m][p] = ap[(((size_t)S * G::kMT + m) * 3 + p) * 64];
        }
        s16x8 b[4][3];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int p = 0; p < 3; ++p) b[t][p] = *reinterpret_cast<const s16x8*>(bb + 16 * t * G::kRow + 32 * s + p * G::kPlane);
        // six products per (frame tile, M-tile), smallest first: (m,m) (l,h) (h,l) (m,h) (h,m) (h,h)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int m = 0; m < 3; ++m) {
            f32x4 v = acc[t][m];
            v = mfma32(a[m][1], b[t][1], v);
            v = mfma32(a[m][2], b[t][0], v);
            v = mfma32(a[m][0], b[t][2], v);
            v = mfma32(a[m][1], b[t][0], v);
            v = mfma32(a[m][0], b[t][1], v);
            v = mfma32(a[m][0], b[t][0], v);
            acc[t][m] = v;
          }
        if (RCED_X6_PREF) {
#pragma unroll
          for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int p = 0; p < 3; ++p) a[m][p] = an[m][p];
        }
      }
    }
    if (c + 1 < G::kChunks) commit((c + 1) & 1);
    __syncthreads();
  }
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

// ---------------------------------------------------------------------------------------------
// Output-layer dgrad (1x129, CH -> 1; tmm::final_dgrad's GEMM) in the same arithmetic:
//     dx[frame, f', ci] = sum_f dz[frame, f] * W[f' - f + 64, ci]  =  D[m = f' CH + ci][frame] = sum_f A[m, f] dz[frame, f],
// K = 129 padded to 160 (five K = 32 steps; the fp32 kernel runs 33 steps of K = 4).  A [M-tile][step][part][lane] x 8 bf16 is
// rebuilt on the device every step (pack_dgrad_x6_dev) and streams from L2; B = the 64 frames' dz rows, split into three bf16
// planes while they are staged.  One workgroup = 4 waves = 64 frames; wave w owns M-tiles w, w + 4, ... kMc at a time.
// ---------------------------------------------------------------------------------------------
constexpr int kDgX6Steps = 5, kDgX6Frames = 64, kDgX6Threads = 256, kDgX6Mc = 2;
constexpr int kDgX6Row = 32 * kDgX6Steps + 8;          // bf16 per staged row: 16-byte aligned, 16 B off a bank period
static __global__ void pack_dgrad_x6_dev(const float* __restrict__ w, int CH, unsigned short* __restrict__ pack) {
  const int M = kF * CH, MT = (M + 15) / 16, total = MT * kDgX6Steps * 64 * 8;      // one thread per (mt, S, lane, e)
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int e = idx & 7, lane = (idx >> 3) & 63, r = idx >> 9, S = r % kDgX6Steps, mt = r / kDgX6Steps;
  const int m = 16 * mt + (lane & 15), f = 32 * S + 8 * (lane >> 4) + e;
  const int fp = m / CH, ci = m - fp * CH, tap = fp - f + 64;
  const float v = (m < M && f < kF && tap >= 0 && tap < kF) ? w[tap * CH + ci] : 0.f;
  const __bf16 h = (__bf16)v;
  const float r1 = v - (float)h;
  const __bf16 mm = (__bf16)r1;
  const __bf16 l = (__bf16)(r1 - (float)mm);
  const size_t base = ((size_t)(mt * kDgX6Steps + S) * 3) * 512 + lane * 8 + e;
  pack[base] = __builtin_bit_cast(unsigned short, h);
  pack[base + 512] = __builtin_bit_cast(unsigned short, mm);
  pack[base + 1024] = __builtin_bit_cast(unsigned short, l);
}

template <int CH>
__global__ __launch_bounds__(kDgX6Threads, 2) void final_dgrad_x6_kernel(const float* __restrict__ dz,
                                                                          const unsigned short* __restrict__ apack,
                                                                          float* __restrict__ dx, int frames) {
  constexpr int M = kF * CH, MT = (M + 15) / 16;            // 1032 rows -> 65 M-tiles (CH 8)
  constexpr int kPlane = kDgX6Frames * kDgX6Row;
  __shared__ __attribute__((aligned(16))) unsigned short rows[3 * kPlane];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, kq = lane >> 4;
  const int f0 = blockIdx.x * kDgX6Frames;
  // stage: pieces of 4 consecutive f of one frame (rows of 129 floats are only 4-byte aligned: scalar loads)
  for (int q = tid; q < kDgX6Frames * (kDgX6Row / 4); q += kDgX6Threads) {
    const int fr = q / (kDgX6Row / 4), f = 4 * (q - fr * (kDgX6Row / 4));
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (f0 + fr < frames) {
      const float* src = dz + (size_t)(f0 + fr) * kF + f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (f + j < kF) v[j] = src[j];
    }
    s16x4 ph, pm, pl;
    split3(v, ph, pm, pl);
    unsigned short* d = rows + fr * kDgX6Row + f;
    *reinterpret_cast<s16x4*>(d) = ph;
    *reinterpret_cast<s16x4*>(d + kPlane) = pm;
    *reinterpret_cast<s16x4*>(d + 2 * kPlane) = pl;
  }
  __syncthreads();
  // B fragment (step S, frame tile t, part p): eight consecutive f of frame 16 t + n
  const unsigned short* bp = rows + n * kDgX6Row + 8 * kq;
  for (int m0 = wave * kDgX6Mc; m0 < MT; m0 += 4 * kDgX6Mc) {
    f32x4 acc[kDgX6Mc][4];
#pragma unroll
    for (int c = 0; c < kDgX6Mc; ++c)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const s16x8* ap = reinterpret_cast<const s16x8*>(apack) + (size_t)m0 * kDgX6Steps * 3 * 64 + lane;
#pragma unroll 1     // (unrolled, hipcc hoists all five steps' operand loads: 256 VGPRs and scratch)
    for (int S = 0; S < kDgX6Steps; ++S) {
      s16x8 a[kDgX6Mc][3], b[4][3];
#pragma unroll
      for (int c = 0; c < kDgX6Mc; ++c)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          a[c][p] = (m0 + c < MT) ? ap[((c * kDgX6Steps + S) * 3 + p) * 64] : s16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int p = 0; p < 3; ++p) b[t][p] = *reinterpret_cast<const s16x8*>(bp + 16 * t * kDgX6Row + 32 * S + p * kPlane);
#pragma unroll
      for (int c = 0; c < kDgX6Mc; ++c)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          f32x4 v = acc[c][t];
          v = mfma32(a[c][1], b[t][1], v);
          v = mfma32(a[c][2], b[t][0], v);
          v = mfma32(a[c][0], b[t][2], v);
          v = mfma32(a[c][1], b[t][0], v);
          v = mfma32(a[c][0], b[t][1], v);
          v = mfma32(a[c][0], b[t][0], v);
          acc[c][t] = v;
        }
    }
    // D row = m = 16 (m0 + c) + 4 kq + r (four consecutive floats of the frame's [129 CH] row), column = frame
#pragma unroll
    for (int c = 0; c < kDgX6Mc; ++c) {
      const int mrow = 16 * (m0 + c) + 4 * kq;
      if (m0 + c >= MT) continue;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int fr = f0 + 16 * t + n;
        if (fr >= frames) continue;
        float* op = dx + (size_t)fr * M + mrow;
        const f32x4 v = acc[c][t];
        if (mrow + 3 < M) {
          *reinterpret_cast<f32x2*>(op) = f32x2{v.x, v.y};          // M even (CH even): 8-byte aligned
          *reinterpret_cast<f32x2*>(op + 2) = f32x2{v.z, v.w};
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (mrow + r < M) op[r] = v[r];
        }
      }
    }
  }
}

}  // namespace x6
}  // namespace rced
