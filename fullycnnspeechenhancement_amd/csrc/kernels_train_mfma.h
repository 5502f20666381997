// MFMA kernels for the training step's 1xk convolutions (forward z = conv + bias, dgrad, wgrad),
// single layer, global tensors in the reference's [pixel][channel] layout.  Same implicit-GEMM
// construction as the inference kernels (kernels_fused_chain.h): a tile of frames is staged in LDS as
// [pixel][even-padded channels] with zero gaps between frames, the B operand of v_mfma_f32_16x16x4_f32 is
// a ds_read_b64 out of that buffer, cout sits on the 16-row M axis.  All 1xk kernels of the three nets
// have odd k, so forward and dgrad share the geometry and differ only in the packed weights.
//   conv1xk_mfma : out[px, co] (=|+=) shift[co] + sum_k W[co, k] * in[window(px), k]
//   wgrad1xk_mfma: dW[k, co] += sum_px in[window(px), k] * dz[px, co]      (K dimension = pixels)
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_fused_chain.h"

namespace rced {
namespace tmm {

using chain::f32x2;
using chain::f32x4;
using chain::mfma;
using chain::pin;

constexpr int kF = 129;
constexpr int kTF = 2;                 // frames per tile
constexpr int kWaves = 4, kThreads = 256;

template <int CIN, int TAPS, int COUT>
struct Geo {
  static constexpr int kCinP = (CIN + 1) & ~1;
  static constexpr int kCoutP = (COUT + 1) & ~1;
  static constexpr int kG = (TAPS - 1) / 2;              // halo = gap between frames
  static constexpr int kS = kF + kG;
  static constexpr int kNPX = kTF * kS;
  static constexpr int kTiles = (kNPX + 15) / 16;
  static constexpr int kRegular = kTiles / kWaves, kExtra = kTiles - kRegular * kWaves;
  static constexpr int kK = TAPS * kCinP;
  static constexpr int kMT = (COUT + 15) / 16;
  static constexpr int kNB64 = kK / 8, kNTail = (kK % 8 + 3) / 4;
  static constexpr int kData = kNB64 * kMT * 128 + kNTail * kMT * 64;
  static constexpr int kPacket = kData + 32;             // + shift[32]
  static constexpr int kInRows = kG + 16 * kTiles + kG;
  static constexpr int kInFloats = ((kInRows * kCinP + 3) / 4) * 4;
  static constexpr int kLdsFloats = kInFloats + kPacket;
};

// Pack the layer's weights (TF layout [TAPS][CIN][COUT] in the variable blob) into the A-fragment packet.
// transpose = 0: forward  (rows = COUT_L outputs, k = tap*cinp + ci).
// transpose = 1: dgrad    (rows = CIN_L outputs; the packet is for a conv whose input has COUT_L channels:
//                          W_t[tap'][co_l][ci_l] = w[TAPS-1-tap'][ci_l][co_l]).
// cin / cout below are those of the conv being PACKED (dgrad: cin = COUT_L, cout = CIN_L).
__global__ void pack_packet(const float* __restrict__ w, const float* __restrict__ shift, int taps, int cin, int cout,
                            int transpose, float* __restrict__ packet) {
  const int cinp = (cin + 1) & ~1, K = taps * cinp, MT = (cout + 15) / 16;
  const int NB = K / 8, NTL = (K % 8 + 3) / 4, data = NB * MT * 128 + NTL * MT * 64;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= data + 32) return;
  if (e >= data) {
    const int c = e - data;
    packet[e] = (shift && c < cout) ? shift[c] : 0.f;
    return;
  }
  int k, co;
  if (e < NB * MT * 128) {
    const int s = e / (MT * 128), r = e - s * MT * 128, mt = r / 128, q = r - mt * 128, lane = q >> 1, ee = q & 1;
    k = 8 * s + 2 * (lane >> 4) + ee;
    co = 16 * mt + (lane & 15);
  } else {
    const int r = e - NB * MT * 128, j = r / (MT * 64), q = r - j * MT * 64, mt = q / 64, lane = q - mt * 64;
    k = 8 * NB + 4 * j + (lane >> 4);
    co = 16 * mt + (lane & 15);
  }
  float v = 0.f;
  const int tap = k / cinp, ci = k - tap * cinp;
  if (k < K && co < cout && ci < cin)
    v = transpose ? w[((taps - 1 - tap) * cout + co) * cin + ci]    // w[tap_l][ci_l = co][co_l = ci], layer dims (cout, cin)
                  : w[(tap * cin + ci) * cout + co];
  packet[e] = v;
}


// ---- tile staging: kTF frames are contiguous in global ([frame][bin][C], C even => 16-byte aligned tile start and
// a whole number of float4).  A thread keeps PER float4 in flight (fetch), and writes them to LDS later (commit) as
// float2 pieces, which never straddle a frame or a pixel because C is even.  The fetch of the NEXT tile is issued
// before the MFMA work of the current one so HBM latency hides behind it.
template <int C>
struct Stage {
  static constexpr int kFrame = kF * C, kElems = kTF * kFrame, kVec = kElems / 4;
  static constexpr int kPer = (kVec + kThreads - 1) / kThreads;
  static_assert(C % 2 == 0 && kElems % 4 == 0, "wide staging needs an even channel count");
};
template <int C>
__device__ __forceinline__ void tile_fetch(const float* __restrict__ base, int frame0, int frames, int tid,
                                           f32x4 (&pre)[Stage<C>::kPer]) {
  using St = Stage<C>;
  const float* src = base + (size_t)frame0 * St::kFrame;
  const int left = frames - frame0;
  const int nvalid = (left < kTF ? left : kTF) * St::kFrame;
#pragma unroll
  for (int i = 0; i < St::kPer; ++i) {
    const int q = tid + i * kThreads;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (q < St::kVec) {
      if (4 * q + 4 <= nvalid) {
        v = *reinterpret_cast<const f32x4*>(src + 4 * q);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (4 * q + j < nvalid) v[j] = src[4 * q + j];
      }
    }
    pre[i] = v;
  }
}
// dst index of element (frame fr, offset r inside the frame) = base_row(fr) * ROWSTRIDE-style mapping given by MAP
template <int C, class MAP>
__device__ __forceinline__ void tile_commit(float* lds, int tid, const f32x4 (&pre)[Stage<C>::kPer], MAP map) {
  using St = Stage<C>;
#pragma unroll
  for (int i = 0; i < St::kPer; ++i) {
    const int q = tid + i * kThreads;
    if (q < St::kVec) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int e = 4 * q + 2 * h;
        const int fr = e / St::kFrame, r = e - fr * St::kFrame;
        *reinterpret_cast<f32x2*>(lds + map(fr, r)) = f32x2{pre[i][2 * h], pre[i][2 * h + 1]};
      }
    }
  }
}

template <int CIN, int TAPS, int COUT, bool ACCUM, bool STATS, int NX>
__device__ __forceinline__ void conv_tile(const float* lds_in, const float* lds_w, float* __restrict__ out, int frame0,
                                          int frames, int wave, int lane,
                                          double (&st1)[Geo<CIN, TAPS, COUT>::kMT][4], double (&st2)[Geo<CIN, TAPS, COUT>::kMT][4]) {
  using G = Geo<CIN, TAPS, COUT>;
  constexpr int NR = G::kRegular, NT = NR + NX, MT = G::kMT;
  const int n = lane & 15, kq = lane >> 4;
  const float* in = lds_in + G::kG * G::kCinP;
  const int xtile = NR * kWaves + wave;
  const int px0 = 16 * wave + n, pxx = 16 * xtile + n;
  f32x4 acc[NT][MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const f32x4 sh = *reinterpret_cast<const f32x4*>(lds_w + G::kData + 16 * mt + 4 * kq);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t][mt] = sh;
  }
  chain::gemm_pass<NR, NX, MT, G::kK, 64 * G::kCinP, 2>(in, (px0 - G::kG) * G::kCinP + 2 * kq,
                                                        (pxx - G::kG) * G::kCinP + 2 * kq, lds_w, lane, acc);
  float p1[MT][4], p2[MT][4];   // this tile's share of sum z, sum z^2 (<= NT values each, fp32)
  if constexpr (STATS) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < 4; ++j) p1[mt][j] = p2[mt][j] = 0.f;
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int px = t < NR ? px0 + 64 * t : pxx;
    const int fr = px / G::kS, f = px - fr * G::kS;
    if (px >= G::kNPX || f >= kF || frame0 + fr >= frames) continue;
    if constexpr (STATS) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v = acc[t][mt][j];
          p1[mt][j] += v;
          p2[mt][j] = fmaf(v, v, p2[mt][j]);
        }
    }
    float* op = out + ((size_t)(frame0 + fr) * kF + f) * COUT;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int co0 = 16 * mt + 4 * kq;
      f32x4 v = acc[t][mt];
      if (co0 + 1 < COUT || (co0 < COUT && (COUT & 1))) {
        // 8-byte pieces where the row stride allows it (COUT even), else scalars
        if constexpr ((COUT & 1) == 0) {
          if (co0 + 1 < COUT) {
            f32x2* p = reinterpret_cast<f32x2*>(op + co0);
            f32x2 r = {v.x, v.y};
            if (ACCUM) { const f32x2 o = *p; r.x += o.x; r.y += o.y; }
            *p = r;
          }
          if (co0 + 3 < COUT) {
            f32x2* p = reinterpret_cast<f32x2*>(op + co0 + 2);
            f32x2 r = {v.z, v.w};
            if (ACCUM) { const f32x2 o = *p; r.x += o.x; r.y += o.y; }
            *p = r;
          }
        } else {
          const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (co0 + j < COUT) op[co0 + j] = (ACCUM ? op[co0 + j] : 0.f) + vv[j];
        }
      }
    }
  }
  if constexpr (STATS) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        st1[mt][j] += (double)p1[mt][j];
        st2[mt][j] += (double)p2[mt][j];
      }
  }
}

// in [frames][129][CIN], out [frames][129][COUT].  Persistent over tiles of kTF frames.
// STATS: also emit this workgroup's per-channel (sum z, sum z^2) of what it wrote, as doubles, into
// part[blockIdx.x][COUT][2] -- the batch-norm statistics, without another pass over z.
template <int CIN, int TAPS, int COUT, bool ACCUM, bool STATS>
__global__ __launch_bounds__(kThreads) void conv1xk_mfma(const float* __restrict__ in, const float* __restrict__ packet,
                                                          float* __restrict__ out, int frames, double* __restrict__ part) {
  using G = Geo<CIN, TAPS, COUT>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* lin = lds;
  float* lw = lds + G::kInFloats;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int e = tid; e < G::kLdsFloats; e += kThreads) lds[e] = e < G::kInFloats ? 0.f : packet[e - G::kInFloats];
  __syncthreads();
  double st1[G::kMT][4], st2[G::kMT][4];
#pragma unroll
  for (int mt = 0; mt < G::kMT; ++mt)
#pragma unroll
    for (int j = 0; j < 4; ++j) st1[mt][j] = st2[mt][j] = 0.0;
  const int ntiles = (frames + kTF - 1) / kTF;
  if constexpr (CIN % 2 == 0) {
    f32x4 pre[Stage<CIN>::kPer];
    if ((int)blockIdx.x < ntiles) tile_fetch<CIN>(in, blockIdx.x * kTF, frames, tid, pre);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      const int frame0 = tile * kTF;
      tile_commit<CIN>(lin, tid, pre, [](int fr, int r) { return (G::kG + fr * G::kS) * CIN + r; });
      __syncthreads();
      if (tile + (int)gridDim.x < ntiles) tile_fetch<CIN>(in, (tile + gridDim.x) * kTF, frames, tid, pre);
      pin();
      if (wave < G::kExtra) conv_tile<CIN, TAPS, COUT, ACCUM, STATS, 1>(lin, lw, out, frame0, frames, wave, lane, st1, st2);
      else conv_tile<CIN, TAPS, COUT, ACCUM, STATS, 0>(lin, lw, out, frame0, frames, wave, lane, st1, st2);
      __syncthreads();
    }
  } else {
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      const int frame0 = tile * kTF;
      // odd channel count (the 1-channel dz of decode_final): element-wise staging, channel stride CinP
      const float* src = in + (size_t)frame0 * kF * CIN;
      for (int e = tid; e < kTF * kF * CIN; e += kThreads) {
        const int fr = e / (kF * CIN), r = e - fr * (kF * CIN), f = r / CIN, ci = r - f * CIN;
        lin[(G::kG + fr * G::kS + f) * G::kCinP + ci] = (frame0 + fr < frames) ? src[e] : 0.f;
      }
      __syncthreads();
      if (wave < G::kExtra) conv_tile<CIN, TAPS, COUT, ACCUM, STATS, 1>(lin, lw, out, frame0, frames, wave, lane, st1, st2);
      else conv_tile<CIN, TAPS, COUT, ACCUM, STATS, 0>(lin, lw, out, frame0, frames, wave, lane, st1, st2);
      __syncthreads();
    }
  }
  if constexpr (STATS) {
    // lane (n, kq) holds channels 16*mt + 4*kq + j of its pixels: add over the 16 pixel lanes, then over the waves
    double* red = reinterpret_cast<double*>(lds);             // [wave][32 channels][2]; the tile loop is over
#pragma unroll
    for (int mt = 0; mt < G::kMT; ++mt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        double a = st1[mt][j], b = st2[mt][j];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          a += __shfl_xor(a, o, 64);
          b += __shfl_xor(b, o, 64);
        }
        if ((lane & 15) == 0) {
          const int c = 16 * mt + 4 * (lane >> 4) + j;
          red[(wave * 32 + c) * 2 + 0] = a;
          red[(wave * 32 + c) * 2 + 1] = b;
        }
      }
    __syncthreads();
    if (tid < 2 * COUT) {
      const int c = tid >> 1, k = tid & 1;
      double t = 0.0;
      for (int w = 0; w < kWaves; ++w) t += red[(w * 32 + c) * 2 + k];
      part[((size_t)blockIdx.x * COUT + c) * 2 + k] = t;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// wgrad: dW[tap][ci][co] += sum_px x[px + tap - G][ci] * dz[px][co].
// MFMA roles: M = k (window index tap*CinP + ci, 16 per tile), N = co, K = pixels (4 per MFMA).
//   A[i = k row][kq = pixel]  = xin[(px0 + kq - G) * CinP + 16*kt + i]
//   B[kq = pixel][j = co]     = dz[(px0 + kq) * CoutP + 16*nt + j]
// Each wave walks the pixel groups g = wave, wave+4, ... of every tile the workgroup owns, keeps the
// whole [K][COUT] partial in accumulators, and the workgroup adds it to dW with atomics at the end.
// ---------------------------------------------------------------------------------------------
template <int CIN, int TAPS, int COUT>
__global__ __launch_bounds__(kThreads) void wgrad1xk_mfma(const float* __restrict__ x, const float* __restrict__ dz,
                                                           float* __restrict__ dW, float* __restrict__ dbias, int frames) {
  using G = Geo<CIN, TAPS, COUT>;
  static_assert(CIN % 2 == 0 && COUT % 2 == 0, "wgrad1xk_mfma stages float4 / float2 pieces");
  // one spare k row carries a constant 1, so its output row is sum_px dz = dbias
  constexpr int KT = (G::kK + 1 + 15) / 16, NTo = G::kMT;
  constexpr int kOneTile = G::kK / 16, kOneRow = G::kK % 16;
  constexpr int kDzRows = 16 * G::kTiles + 4;
  constexpr int kDzStride = 32;                        // floats per pixel row of the dz tile (>= 16*NTo, bank friendly)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* lin = lds;                                    // [kInRows + tail][CinP]
  float* ldz = lds + G::kInFloats + 64;                // [kDzRows][32]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kq = lane >> 4;
  for (int e = tid; e < G::kInFloats + 64 + kDzRows * kDzStride; e += kThreads) lds[e] = 0.f;
  f32x4 acc[KT][NTo];
#pragma unroll
  for (int a = 0; a < KT; ++a)
#pragma unroll
    for (int b = 0; b < NTo; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ntiles = (frames + kTF - 1) / kTF;
  f32x4 prex[Stage<CIN>::kPer], prez[Stage<COUT>::kPer];
  if ((int)blockIdx.x < ntiles) {
    tile_fetch<CIN>(x, blockIdx.x * kTF, frames, tid, prex);
    tile_fetch<COUT>(dz, blockIdx.x * kTF, frames, tid, prez);
  }
  __syncthreads();
  const float* ain = lin + kq * G::kCinP + i;          // window start of pixel (px0 + kq) is row (px0 + kq) of lin
  const float* bin = ldz + kq * kDzStride + i;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    tile_commit<CIN>(lin, tid, prex, [](int fr, int r) { return (G::kG + fr * G::kS) * CIN + r; });
    tile_commit<COUT>(ldz, tid, prez, [](int fr, int r) {
      const int f = r / COUT, co = r - f * COUT;
      return (fr * G::kS + f) * kDzStride + co;
    });
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) {
      tile_fetch<CIN>(x, (tile + gridDim.x) * kTF, frames, tid, prex);
      tile_fetch<COUT>(dz, (tile + gridDim.x) * kTF, frames, tid, prez);
    }
    pin();
    for (int g = wave; g < G::kNPX / 4 + 1; g += kWaves) {
      const int px0 = 4 * g;
      float a[KT], b[NTo];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) a[kt] = ain[px0 * G::kCinP + 16 * kt];
      if (i == kOneRow) a[kOneTile] = 1.f;
#pragma unroll
      for (int nt = 0; nt < NTo; ++nt) b[nt] = bin[px0 * kDzStride + 16 * nt];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int nt = 0; nt < NTo; ++nt) acc[kt][nt] = mfma(a[kt], b[nt], acc[kt][nt]);
    }
    __syncthreads();
  }
  // D row = k = 16*kt + 4*kq + r, column = co = 16*nt + i
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int nt = 0; nt < NTo; ++nt) {
      const int co = 16 * nt + i;
      const float vv[4] = {acc[kt][nt].x, acc[kt][nt].y, acc[kt][nt].z, acc[kt][nt].w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = 16 * kt + 4 * kq + r;
        const int tap = k / G::kCinP, ci = k - tap * G::kCinP;
        if (k < G::kK && ci < CIN && co < COUT) atomicAdd(dW + (tap * CIN + ci) * COUT + co, vv[r]);
        if (k == G::kK && co < COUT && dbias) atomicAdd(dbias + co, vv[r]);
      }
    }
}

}  // namespace tmm
}  // namespace rced
