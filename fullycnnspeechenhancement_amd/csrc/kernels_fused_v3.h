// Fused CR-CED (V3) forward: layers 0..14 of model_utils/model.py:64-96 in ONE kernel, fp32 MFMA.
//
// Why this shape (DESIGN.md has the long form):
//   * only the first conv (8x9) looks along time; everything after is 1xk along frequency, so a
//     tile of frames runs through all 15 conv+BN+ReLU layers without leaving the CU;
//   * per layer the conv is an implicit GEMM  D[cout, pixel] = sum_k W[cout, k] * X[k, pixel]
//     with k = (tap, cin).  Activations live in LDS as [pixel][channel] with the channel stride
//     EXACTLY cin, so the im2col row of a pixel is one contiguous window of taps*cin floats:
//     the B operand of v_mfma_f32_16x16x4_f32 is read straight out of LDS with ds_read_b64,
//     no im2col copy, no shuffles;
//   * weights are pre-packed on the host into MFMA A-fragment order, BN folded, and streamed
//     L2 -> LDS one layer ahead (ping-pong);
//   * cout sits on the MFMA M axis (16 rows).  18 and 30 pad to 2 M-tiles; the 30->8 layers use
//     two pixel phases as rows (8 cout x 2 adjacent pixels = 16 rows, K = 10 taps instead of 9),
//     which is 90 % efficient instead of 50 %;
//   * the two CR-CED block skips (model.py:75-76, added after ReLU) never touch LDS: they stay in
//     the accumulator registers of the wave that produced them.
//
// Pixel space of a tile: kTF frames, frame i at flat pixels [i*kS, i*kS+129); the kS-129 = 4 gap
// pixels between frames are always zero and serve as the SAME-padding halo of both neighbours.
#pragma once
#include <hip/hip_runtime.h>

namespace rced {
namespace v3 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kF = 129;
constexpr int kTF = 4;                       // frames per tile
constexpr int kS = 133;                      // pixel stride of a frame (129 + 4 zero gap)
constexpr int kNPX = kTF * kS;               // 532 pixels per tile
constexpr int kTiles16 = (kNPX + 15) / 16;   // 34 N-tiles of 16 pixels
constexpr int kTiles32 = (kNPX + 31) / 32;   // 17 N-tiles of 16 pixel PAIRS
constexpr int kPX = kTiles16 * 16;           // 544 pixels computed (the tail past 532 is masked)
constexpr int kWaves = 8;
constexpr int kThreads = kWaves * 64;
constexpr int kHCh = 8;                      // channels of the tensor handed to the final layer

// ---- LDS map, in floats -----------------------------------------------------------------
constexpr int kB8Pad = 4, kB18Pad = 2, kB30Pad = 4;       // leading zero rows (pixels -pad..-1)
constexpr int kB8Rows = kB8Pad + kPX + 4;                 // 9-tap windows reach pixel 547
constexpr int kB18Rows = kB18Pad + kPX + 2;               // 5-tap windows reach pixel 545
constexpr int kB30Rows = kB30Pad + kPX + 4;               // 10-tap pair windows reach pixel 547
constexpr int kB8Off = 0;
constexpr int kB18Off = kB8Off + kB8Rows * 8;
constexpr int kB30Off = kB18Off + kB18Rows * 18;
constexpr int kWRegion = 38 * 128;                        // largest layer: 30->8, 38 b64-steps
constexpr int kWOff = kB30Off + kB30Rows * 30;
constexpr int kLdsFloats = kWOff + 2 * kWRegion;
constexpr int kLdsBytes = kLdsFloats * 4;                 // 162,272 B of the 163,840 B
static_assert(kLdsBytes <= 160 * 1024, "LDS budget");
// input rows of the 8x9 first layer alias the (not yet live) B30 buffer, from its pixel-0 row on
constexpr int kX0Rows = kTF + 7;
constexpr int kX0Floats = ((kX0Rows * kS + 24 + 3) / 4) * 4;   // 1488: max index 543 + 7*133 + 8
constexpr int kX0Off = kB30Off + kB30Pad * 30;
static_assert(kX0Floats <= 60 * 30, "X0 must sit inside rows that layer 2 rewrites");

// ---- packed weight stream (floats), per layer --------------------------------------------
//  first layer (8x9x1->18): 18 k-steps x 2 M-tiles x 64 lanes, one float per lane (b32 steps)
//  L1 (1x9, 8->18):  9 b64-steps x 2 M-tiles x 64 lanes x 2
//  L2 (1x5, 18->30): 12 b64-steps x 2 M-tiles x 64 x 2   (K = 90, last step 2 valid)
//  L3 (1x9, 30->8):  38 b64-steps x 1 M-tile x 64 x 2    (K = 300 = 10 taps x 30, pixel pairs)
constexpr int kWFirst = 18 * 2 * 64;      // 2304
constexpr int kW1 = 9 * 2 * 128;          // 2304
constexpr int kW2 = 12 * 2 * 128;         // 3072
constexpr int kW3 = 38 * 1 * 128;         // 4864
constexpr int kWBlock = kW1 + kW2 + kW3;  // block 0 uses kWFirst in place of kW1 (same size)
static_assert(kWFirst == kW1, "block 0 and blocks 1..4 share one stream layout");
constexpr int kWTotal = 5 * kWBlock;
constexpr int kShiftPerLayer = 32;        // shift[co], zero padded

struct Params {
  const float* x;       // [N, T, 129]
  float* h;             // [N*T, 129, 8]  output of CD2 (input of decode_final)
  const float* wpack;   // kWTotal floats
  const float* shifts;  // 15 x 32 floats
  int N, T;
  int tiles_per_utt;    // ceil(T / kTF)
  int total_tiles;      // N * tiles_per_utt
};

__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Issue the global loads of a packed layer (<= 3 float4 per thread; out-of-range threads re-read
// the last float4 so that every register is defined) ...
struct WStage {
  f32x4 v0, v1, v2;
};
template <int NFLOATS>
__device__ __forceinline__ WStage wstage_load(const float* __restrict__ src, int tid) {
  constexpr int n4 = NFLOATS / 4;
  const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
  WStage st;
  st.v0 = s4[tid < n4 ? tid : n4 - 1];
  st.v1 = st.v0;
  st.v2 = st.v0;
  if constexpr (n4 > kThreads) st.v1 = s4[tid + kThreads < n4 ? tid + kThreads : n4 - 1];
  if constexpr (n4 > 2 * kThreads) st.v2 = s4[tid + 2 * kThreads < n4 ? tid + 2 * kThreads : n4 - 1];
  return st;
}
// ... and park them in the other weight region once the current layer's math is issued.
template <int NFLOATS>
__device__ __forceinline__ void wstage_store(const WStage& st, float* dst, int tid) {
  constexpr int n4 = NFLOATS / 4;
  f32x4* d4 = reinterpret_cast<f32x4*>(dst);
  if (tid < n4) d4[tid] = st.v0;
  if constexpr (n4 > kThreads)
    if (tid + kThreads < n4) d4[tid + kThreads] = st.v1;
  if constexpr (n4 > 2 * kThreads)
    if (tid + 2 * kThreads < n4) d4[tid + 2 * kThreads] = st.v2;
}

// One implicit-GEMM pass over NT N-tiles with b64 steps.
//   act  : LDS buffer base (float index of pixel 0, channel 0)
//   boff : per slot, float offset of this lane's window start (+ 2*kq)
//   tail : float delta applied in the last step so that lanes past the window re-read in-window
//          data (their weights are zero); keeps every read inside the pixel's own window
//   w    : LDS weight region, [step][mt][lane][2]
template <int NT, int MT, int STEPS>
__device__ __forceinline__ void gemm_pass(const float* act, const int (&boff)[NT], int tail, const float* w,
                                          int lane, f32x4 (&acc)[NT][MT]) {
  const f32x2* wp = reinterpret_cast<const f32x2*>(w) + lane;
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    f32x2 a[MT], b[NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a[mt] = wp[(s * MT + mt) * 64];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int off = boff[t] + 8 * s + (s == STEPS - 1 ? tail : 0);
      b[t] = *reinterpret_cast<const f32x2*>(act + off);
    }
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[t][mt] = mfma(a[mt][e], b[t][e], acc[t][mt]);
  }
}

// First layer, 8x9 kernel on the 1-channel input: k-step s = ih*9 + j, lane kq <-> time tap 4*ih+kq.
template <int NT>
__device__ __forceinline__ void first_pass(const float* x0, const int (&boff)[NT], const float* w, int lane,
                                           f32x4 (&acc)[NT][2]) {
  const float* wp = w + lane;
#pragma unroll
  for (int ih = 0; ih < 2; ++ih)
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      const int s = ih * 9 + j;
      const float a0 = wp[(s * 2 + 0) * 64], a1 = wp[(s * 2 + 1) * 64];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float b = x0[boff[t] + ih * 4 * kS + j];
        acc[t][0] = mfma(a0, b, acc[t][0]);
        acc[t][1] = mfma(a1, b, acc[t][1]);
      }
    }
}

__device__ __forceinline__ f32x4 relu4(f32x4 v) {
  f32x4 r;
  r.x = fmaxf(v.x, 0.f);
  r.y = fmaxf(v.y, 0.f);
  r.z = fmaxf(v.z, 0.f);
  r.w = fmaxf(v.w, 0.f);
  return r;
}

// Epilogue of a P = 1 pass (rows = 16*mt + 4*kq + j output channels, column = pixel):
// ReLU, zero the gap pixels, store [pixel][COUT] with 8-byte stores.
template <int NT, int COUT>
__device__ __forceinline__ void store_p1(float* out, const f32x4 (&acc)[NT][2], const int (&px)[NT],
                                         unsigned valid_bits, int kq) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const bool ok = (valid_bits >> t) & 1u;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int co0 = 16 * mt + 4 * kq;
      f32x4 v = relu4(acc[t][mt]);
      if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
      float* p = out + px[t] * COUT + co0;
      if (co0 + 1 < COUT) *reinterpret_cast<f32x2*>(p) = f32x2{v.x, v.y};
      if (co0 + 3 < COUT) *reinterpret_cast<f32x2*>(p + 2) = f32x2{v.z, v.w};
    }
  }
}

template <int NT16, int NT32>
__device__ __forceinline__ void run_tile(const Params& P, float* lds, int tile, int& wcur, int tid, int lane,
                                         int wave) {
  const int n = lane & 15, kq = lane >> 4;
  float* b8 = lds + kB8Off + kB8Pad * 8;
  float* b18 = lds + kB18Off + kB18Pad * 18;
  float* b30 = lds + kB30Off + kB30Pad * 30;
  float* x0 = lds + kX0Off;
  float* const wbase = lds + kWOff;
#define WREG(i) (wbase + (i) * kWRegion)

  const int utt = tile / P.tiles_per_utt;
  const int t0 = (tile - utt * P.tiles_per_utt) * kTF;   // first frame of the tile

  // ---- per-lane tile geometry (tile-invariant except the T edge) ---------------------------
  int px16[NT16], px32[NT32];
  unsigned ok16 = 0, ok32 = 0, st32 = 0;   // valid pixel / valid pixel AND frame < T (global store)
  int hidx[NT32];
#pragma unroll
  for (int t = 0; t < NT16; ++t) {
    px16[t] = 16 * (wave + kWaves * t) + n;
    const int fr = px16[t] / kS, f = px16[t] - fr * kS;
    if (px16[t] < kNPX && f < kF) ok16 |= 1u << t;
  }
#pragma unroll
  for (int t = 0; t < NT32; ++t) {
    px32[t] = 2 * (16 * (wave + kWaves * t) + n) + (kq >> 1);
    const int fr = px32[t] / kS, f = px32[t] - fr * kS;
    const bool v = px32[t] < kNPX && f < kF;
    if (v) ok32 |= 1u << t;
    if (v && t0 + fr < P.T) st32 |= 1u << t;
    hidx[t] = (fr * kF + f) * kHCh + 4 * (kq & 1);
  }
  int off_b8[NT16], off_b18[NT16], off_b30[NT32], off_x0[NT16];
#pragma unroll
  for (int t = 0; t < NT16; ++t) {
    off_b8[t] = (px16[t] - 4) * 8 + 2 * kq;
    off_b18[t] = (px16[t] - 2) * 18 + 2 * kq;
    off_x0[t] = px16[t] + kq * kS;
  }
#pragma unroll
  for (int t = 0; t < NT32; ++t) off_b30[t] = (2 * (16 * (wave + kWaves * t) + n) - 4) * 30 + 2 * kq;
  const int tail2 = 2 * (0 - kq);                      // K = 90: only pair 0 of the last step is real
  const int tail3 = 2 * ((kq < 2 ? kq : kq - 2) - kq);  // K = 300: pairs 0,1 of the last step are real

  // ---- stage the 11 input rows of the tile (zero outside [0,T) and in the gaps) -------------
  {
    const float* xu = P.x + (size_t)utt * P.T * kF;
    for (int e = tid; e < kX0Floats; e += kThreads) {
      const int q = e - 4;
      const int r = q >= 0 ? q / kS : -1;
      const int f = q - r * kS;
      const int tt = t0 + r - 3;
      float v = 0.f;
      if (q >= 0 && r < kX0Rows && f < kF && tt >= 0 && tt < P.T) v = xu[(size_t)tt * kF + f];
      x0[e] = v;
    }
  }
  __syncthreads();

  f32x4 skip_ce1[NT32], skip_ce2[NT32];
#pragma unroll
  for (int t = 0; t < NT32; ++t) skip_ce1[t] = skip_ce2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* wsrc = P.wpack;
  const float* shsrc = P.shifts;

#pragma unroll 1
  for (int blk = 0; blk < 5; ++blk) {
    // ======== layer 1 of the block: (8x9, 1->18) for block 0, (1x9, 8->18) otherwise =========
    {
      const WStage st = wstage_load<kW2>(wsrc + kW1, tid);
      f32x4 acc[NT16][2];
      f32x4 sh[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) sh[mt] = *reinterpret_cast<const f32x4*>(shsrc + 16 * mt + 4 * kq);
#pragma unroll
      for (int t = 0; t < NT16; ++t) { acc[t][0] = sh[0]; acc[t][1] = sh[1]; }
      if (blk == 0) {
        first_pass<NT16>(x0, off_x0, WREG(wcur), lane, acc);
      } else {
        gemm_pass<NT16, 2, 9>(b8, off_b8, 0, WREG(wcur), lane, acc);
      }
      store_p1<NT16, 18>(b18, acc, px16, ok16, kq);
      wstage_store<kW2>(st, WREG(wcur ^ 1), tid);
      wcur ^= 1;
      __syncthreads();
    }
    // ======== layer 2: (1x5, 18->30) ==========================================================
    {
      const WStage st = wstage_load<kW3>(wsrc + kW1 + kW2, tid);
      f32x4 acc[NT16][2];
      f32x4 sh[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        sh[mt] = *reinterpret_cast<const f32x4*>(shsrc + kShiftPerLayer + 16 * mt + 4 * kq);
#pragma unroll
      for (int t = 0; t < NT16; ++t) { acc[t][0] = sh[0]; acc[t][1] = sh[1]; }
      gemm_pass<NT16, 2, 12>(b18, off_b18, tail2, WREG(wcur), lane, acc);
      store_p1<NT16, 30>(b30, acc, px16, ok16, kq);
      wstage_store<kW3>(st, WREG(wcur ^ 1), tid);
      wcur ^= 1;
      __syncthreads();
    }
    // ======== layer 3: (1x9, 30->8) on pixel pairs; block skips; hand-off =====================
    {
      // next: layer 1 of the next block, or of block 0 of the next tile (stream wraps around)
      const float* nxt = (blk == 4) ? P.wpack : wsrc + kWBlock;
      const WStage st = wstage_load<kW1>(nxt, tid);
      f32x4 acc[NT32][1];
      const f32x4 sh = *reinterpret_cast<const f32x4*>(shsrc + 2 * kShiftPerLayer + 4 * (kq & 1));
#pragma unroll
      for (int t = 0; t < NT32; ++t) acc[t][0] = sh;
      gemm_pass<NT32, 1, 38>(b30, off_b30, tail3, WREG(wcur), lane, acc);
#pragma unroll
      for (int t = 0; t < NT32; ++t) {
        f32x4 v = relu4(acc[t][0]);
        if (blk == 3) v += skip_ce2[t];   // CD1 + CE2 (model.py:87, 75-76: after the ReLU)
        if (blk == 4) v += skip_ce1[t];   // CD2 + CE1 (model.py:88)
        if (!((ok32 >> t) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
        skip_ce1[t] = (blk == 0) ? v : skip_ce1[t];
        skip_ce2[t] = (blk == 1) ? v : skip_ce2[t];
        if (blk < 4) {
          *reinterpret_cast<f32x4*>(b8 + px32[t] * 8 + 4 * (kq & 1)) = v;
        } else if ((st32 >> t) & 1u) {
          float* hp = P.h + ((size_t)utt * P.T + t0) * (kF * kHCh) + hidx[t];
          *reinterpret_cast<f32x4*>(hp) = v;
        }
      }
      wstage_store<kW1>(st, WREG(wcur ^ 1), tid);
      wcur ^= 1;
      __syncthreads();
    }
    wsrc += kWBlock;
    shsrc += 3 * kShiftPerLayer;
  }
#undef WREG
}

__global__ __launch_bounds__(kThreads) void fused_v3_kernel(Params P) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // zero all of LDS once: gap pixels and margins are never written afterwards
  for (int e = tid; e < kLdsFloats; e += kThreads) lds[e] = 0.f;
  __syncthreads();
  // weights of the very first layer into region 0
  {
    const WStage st = wstage_load<kW1>(P.wpack, tid);
    wstage_store<kW1>(st, lds + kWOff, tid);
  }
  int wcur = 0;
  __syncthreads();

  for (int tile = blockIdx.x; tile < P.total_tiles; tile += gridDim.x) {
    // waves 0,1 own five 16-pixel tiles (others four); wave 0 owns three pair tiles (others two)
    if (wave == 0) run_tile<5, 3>(P, lds, tile, wcur, tid, lane, wave);
    else if (wave == 1) run_tile<5, 2>(P, lds, tile, wcur, tid, lane, wave);
    else run_tile<4, 2>(P, lds, tile, wcur, tid, lane, wave);
  }
}

// ---------------------------------------------------------------------------------------------
// decode_final (1x129, 8->1, no BN, no ReLU; model.py:89-90) as a dense Toeplitz GEMM:
//   y[frame, f] = b + sum_{f', ci} h[frame, f', ci] * W[f' - f + 64, ci]
//   D[f (M: 9 tiles of 16), frame (N)] = sum_k A[f, k] * B[k, frame],  k = f'*8 + ci, K = 1032.
// A (Toeplitz-expanded, A-fragment order) streams from L2; B is the h row of a frame, contiguous.
// One workgroup = 3 waves = 64 frames; wave w owns M-tiles 3w..3w+2.
// ---------------------------------------------------------------------------------------------
constexpr int kFinK = kF * kHCh;          // 1032
constexpr int kFinSteps = kFinK / 8;      // 129 b64-steps
constexpr int kFinMT = 9;
constexpr int kFinPack = kFinSteps * kFinMT * 128;   // floats
constexpr int kFinFrames = 64;
constexpr int kFinThreads = 192;

__global__ __launch_bounds__(kFinThreads) void final_gemm_kernel(const float* __restrict__ h,
                                                                  const float* __restrict__ apack, float bias,
                                                                  float* __restrict__ y, int frames) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, kq = lane >> 4;
  const int f0 = blockIdx.x * kFinFrames;
  const f32x2* ap = reinterpret_cast<const f32x2*>(apack) + (wave * 3) * 64 + lane;
  const float* bp[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    int fr = f0 + 16 * t + n;
    if (fr >= frames) fr = frames - 1;   // clamp: computed, never stored
    bp[t] = h + (size_t)fr * kFinK + 2 * kq;
  }
  f32x4 acc[4][3];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int m = 0; m < 3; ++m) acc[t][m] = f32x4{bias, bias, bias, bias};
#pragma unroll 2
  for (int s = 0; s < kFinSteps; ++s) {
    f32x2 a[3], b[4];
#pragma unroll
    for (int m = 0; m < 3; ++m) a[m] = ap[(s * kFinMT + m) * 64];
#pragma unroll
    for (int t = 0; t < 4; ++t) b[t] = *reinterpret_cast<const f32x2*>(bp[t] + 8 * s);
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int m = 0; m < 3; ++m) acc[t][m] = mfma(a[m][e], b[t][e], acc[t][m]);
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

}  // namespace v3
}  // namespace rced
