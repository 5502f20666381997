// CR-CED (V3) fused forward, SIXTEEN-WAVE variant of kernels_fused_v3.h (read that file first).
//
// Why: experiments on the eight-wave kernel show that its idle MFMA cycles are not a
// synchronisation effect -- removing every barrier, or running the two waves of a SIMD half a layer apart, gains
// 2 % -- but an occupancy one: one wave per SIMD keeps the pipe 65 % busy, two waves 80 %.  LDS (one tile of 4
// frames = 160 KB) fixes the workgroup count at one per CU, so the only way to more waves per SIMD is a bigger
// workgroup: 1024 threads = 16 waves = 4 per SIMD, each with half the tiles and at most 128 VGPRs.
//
// Same LDS layout, packets, passes and epilogues as the eight-wave kernel; what changes is the tile map:
//   16-pixel tiles 0..32: wave w owns tiles w and w + 16; tile 32 is the extra one.
//   layer 1: extra main tile 32 -> wave 0; remainder tiles 0..3 -> waves 1..4, remainder tile 4 -> wave 7
//            (MFMAs per SIMD 194 / 176 / 176 / 208).
//   layer 2: tile 32 cut in four (M-tile x K-half) on waves 0..3, as in the eight-wave kernel.
//   layer 3: pair tile w -> wave w; pair tile 16 cut in four along K on waves 0..3.
#pragma once
#include <hip/hip_runtime.h>

#include "lds_dma.h"

#include "kernels_fused_v3.h"

namespace rced {
namespace v3w {

using v3::f32x2;
using v3::f32x4;
using v3::kB18Off;
using v3::kB18Pad;
using v3::kB30Off;
using v3::kB30Pad;
using v3::kB8Off;
using v3::kB8Pad;
using v3::kB8S;
using v3::kF;
using v3::kHCh;
using v3::kL2Steps;
using v3::kL3Steps;
using v3::kLdsBytes;
using v3::kLdsFloats;
using v3::kNPX;
using v3::kS;
using v3::kTF;
using v3::kW1;
using v3::kW1Data;
using v3::kW1Main;
using v3::kW2;
using v3::kW2Data;
using v3::kW3;
using v3::kW3Data;
using v3::kWBlock;
using v3::kWOff;
using v3::kWRegion;
using v3::kX0Floats;
using v3::kX0Off;
using v3::kX0Rows;
using v3::lds_peek;
using v3::lds_poke;
using v3::mfma;
using v3::Params;
using v3::pin;
using v3::px_valid;
using v3::relu4;
using v3::span_has_gap;
using v3::store_p1;
using v3::store_p1_mt;
using v3::store_rem;

constexpr int kWaves = 16;
constexpr int kThreads = kWaves * 64;
constexpr int kSlot = 16 * kWaves;   // pixels between a wave's two regular tiles

__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

template <int NFLOATS>
__device__ __forceinline__ void packet_dma(const float* __restrict__ src, float* dst, int wave, int lane) {
  constexpr int n4 = NFLOATS / 4;
  constexpr int chunks = (n4 + 63) / 64;
#pragma unroll
  for (int i = 0; i < (chunks + kWaves - 1) / kWaves; ++i) {
    const int c = wave + i * kWaves;
    if (c < chunks) {
      const int idx = c * 64 + lane;
      if (idx < n4)
        lds_dma16(src + (size_t)idx * 4, dst + c * 256);
    }
  }
}
__device__ __forceinline__ void layer_end_sync() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// Layer 1 of blocks 1..4 (see v3::l1_pass): main pass on NM = NMR + NMX tiles and remainder pass on NR (0/1) tiles
// as one pipelined stream.
template <int NMR, int NMX, int NR>
__device__ __forceinline__ void l1_pass(const float* b8, int offm0, int offmx, int offr, const float* w, int lane,
                                        f32x4 (&accm)[NMR + NMX][1], f32x4 (&accr)[2]) {
  constexpr int NM = NMR + NMX, DEPTH = 2, RING = DEPTH + 1;
  constexpr int SLOTS = NR > 0 ? 16 : 9;
  const f32x2* wm = reinterpret_cast<const f32x2*>(w) + lane;
  const f32x2* wr = reinterpret_cast<const f32x2*>(w + kW1Main) + lane;
  f32x2 am[RING], bm[RING][NM], ar[RING], br[RING];
  auto main_step = [](int i) { return NR > 0 ? ((i * 9) / 16 != ((i + 1) * 9) / 16 ? (i * 9) / 16 : -1) : i; };
  auto load = [&](int i, int buf) {
    if constexpr (NR > 0) {
      ar[buf] = wr[i * 64];
      br[buf] = *reinterpret_cast<const f32x2*>(b8 + offr + kB8S * i);
    }
    const int m = main_step(i);
    if (m >= 0) {
      am[buf] = wm[m * 64];
#pragma unroll
      for (int t = 0; t < NMR; ++t) bm[buf][t] = *reinterpret_cast<const f32x2*>(b8 + offm0 + t * kSlot * kB8S + kB8S * m);
      if constexpr (NMX > 0) bm[buf][NMR] = *reinterpret_cast<const f32x2*>(b8 + offmx + kB8S * m);
    }
  };
#pragma unroll
  for (int i = 0; i < DEPTH; ++i) load(i, i % RING);
#pragma unroll
  for (int i = 0; i < SLOTS; ++i) {
    if (i + DEPTH < SLOTS) load(i + DEPTH, (i + DEPTH) % RING);
    pin();
    const int buf = i % RING;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      if constexpr (NR > 0) accr[i & 1] = mfma(ar[buf][e], br[buf][e], accr[i & 1]);
      if (main_step(i) >= 0) {
#pragma unroll
        for (int t = 0; t < NM; ++t) accm[t][0] = mfma(am[buf][e], bm[buf][t][e], accm[t][0]);
      }
    }
    pin();
  }
}

// Block 0 (8x9 kernel on the 1-channel input rows, b32 steps), see v3::l1_first_pass.
template <int NMR, int NMX, int NR>
__device__ __forceinline__ void l1_first_pass(const float* x0, int offm0, int offmx, int offr, const float* w, int lane,
                                              f32x4 (&accm)[NMR + NMX][1], f32x4 (&accr)[2]) {
  constexpr int NM = NMR + NMX, DEPTH = 3, RING = DEPTH + 1;
  constexpr int SLOTS = NR > 0 ? 32 : 18;
  const float* wm = w + lane;
  const float* wr = w + kW1Main + lane;
  float am[RING], bm[RING][NM], ar[RING], br[RING];
  auto main_step = [](int i) { return NR > 0 ? ((i * 18) / 32 != ((i + 1) * 18) / 32 ? (i * 18) / 32 : -1) : i; };
  auto load = [&](int i, int buf) {
    if constexpr (NR > 0) {
      ar[buf] = wr[i * 64];
      br[buf] = x0[offr + (i / 16) * 4 * kS + (i % 16)];
    }
    const int m = main_step(i);
    if (m >= 0) {
      am[buf] = wm[m * 64];
      const int d = (m / 9) * 4 * kS + (m % 9);
#pragma unroll
      for (int t = 0; t < NMR; ++t) bm[buf][t] = x0[offm0 + t * kSlot + d];
      if constexpr (NMX > 0) bm[buf][NMR] = x0[offmx + d];
    }
  };
#pragma unroll
  for (int i = 0; i < DEPTH; ++i) load(i, i % RING);
#pragma unroll
  for (int i = 0; i < SLOTS; ++i) {
    if (i + DEPTH < SLOTS) load(i + DEPTH, (i + DEPTH) % RING);
    pin();
    const int buf = i % RING;
    if constexpr (NR > 0) accr[i & 1] = mfma(ar[buf], br[buf], accr[i & 1]);
    if (main_step(i) >= 0) {
#pragma unroll
      for (int t = 0; t < NM; ++t) accm[t][0] = mfma(am[buf], bm[buf][t], accm[t][0]);
    }
    pin();
  }
}

// ---- the three layer kinds ---------------------------------------------------------------------
// NMX: this wave also has main tile 32.  NR: this wave also has remainder tile xr.
template <int NMX, int NR>
__device__ __forceinline__ void layer1(float* lds, const float* w, bool first, int wave, int lane, int xr) {
  constexpr int NMR = 2, NM = NMR + NMX;
  lane = opaque(lane);
  const int n = lane & 15, kq = lane >> 4;
  float* b8 = lds + kB8Off + kB8Pad * kB8S;
  float* b18 = lds + kB18Off + kB18Pad * 18;
  const float* x0 = lds + kX0Off;
  const int px0 = 16 * wave + n, pxx = 16 * 32 + n;
  const int pxr = 8 * (16 * xr + n);
  f32x4 accm[NM][1], accr[2];
  const f32x4 sh = *reinterpret_cast<const f32x4*>(w + kW1Data + 4 * kq);
  const f32x2 s2 = *reinterpret_cast<const f32x2*>(w + kW1Data + 16);
#pragma unroll
  for (int t = 0; t < NM; ++t) accm[t][0] = sh;
  accr[0] = f32x4{s2.x, s2.y, s2.x, s2.y};
  accr[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (first)
    l1_first_pass<NMR, NMX, NR>(x0, px0 + kq * kS, pxx + kq * kS, pxr + kq * kS, w, lane, accm, accr);
  else
    l1_pass<NMR, NMX, NR>(b8, (px0 - 4) * kB8S + 2 * kq, (pxx - 4) * kB8S + 2 * kq, (pxr - 4) * kB8S + 2 * kq, w, lane,
                          accm, accr);
#pragma unroll
  for (int t = 0; t < NMR; ++t)
    store_p1<1, 18>(b18, accm[t], px0 + kSlot * t, kq, span_has_gap(16 * (wave + kWaves * t), 16));
  if constexpr (NMX > 0) store_p1<1, 18>(b18, accm[NMR], pxx, kq, span_has_gap(16 * 32, 16));
  if constexpr (NR > 0) store_rem(b18, accr[0] + accr[1], pxr, kq, span_has_gap(128 * xr, 128));
}

constexpr int kL2Plain = 0, kL2Reducer = 1, kL2Helper = 2;
using v3::kFlag2Off;
using v3::kL2Cut;
using v3::kScratch2Off;

template <int ROLE, int XMTP>
__device__ __forceinline__ void layer2(float* lds, const float* w, int wave, int lane, unsigned tag) {
  constexpr int NR = 2, NX = ROLE == kL2Plain ? 0 : 1, NT = NR + NX;
  constexpr int XMT = ROLE == kL2Plain ? -1 : XMTP;
  lane = opaque(lane);
  const int n = lane & 15, kq = lane >> 4;
  const float* b18 = lds + kB18Off + kB18Pad * 18;
  float* b30 = lds + kB30Off + kB30Pad * 30;
  const int px0 = 16 * wave + n, pxx = 16 * 32 + n;
  f32x4 acc[NT][2];
  f32x4 sh[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) sh[mt] = *reinterpret_cast<const f32x4*>(w + kW2Data + 16 * mt + 4 * kq);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const bool partial = t == NR && ROLE == kL2Helper;
    acc[t][0] = partial ? f32x4{0.f, 0.f, 0.f, 0.f} : sh[0];
    acc[t][1] = partial ? f32x4{0.f, 0.f, 0.f, 0.f} : sh[1];
  }
  const int tailoff = (kq < 1 ? kq : 1) - 2 * kq;
  {
    constexpr int XS0 = ROLE == kL2Reducer ? kL2Cut : 0;
    constexpr int XS1 = ROLE == kL2Helper ? kL2Cut : kL2Steps;
    v3::gemm_pass<NR, NX, 2, XMT, kL2Steps, XS0, XS1, ROLE == kL2Reducer, kSlot * 18, 1>(
        b18, (px0 - 2) * 18 + 2 * kq, (pxx - 2) * 18 + 2 * kq, tailoff, w, lane, acc);
  }
  if constexpr (ROLE == kL2Helper) {
    *reinterpret_cast<f32x4*>(lds + kScratch2Off + XMTP * 256 + 4 * lane) = acc[NR][XMTP];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) lds_poke(lds + kFlag2Off + XMTP, tag);
  }
#pragma unroll
  for (int t = 0; t < NR; ++t)
    store_p1<2, 30>(b30, acc[t], px0 + kSlot * t, kq, span_has_gap(16 * (wave + kWaves * t), 16));
  if constexpr (ROLE == kL2Reducer) {
    for (int spin = 0; spin < (1 << 22); ++spin) {
      if (__builtin_amdgcn_readfirstlane(lds_peek(lds + kFlag2Off + XMTP)) == tag) break;
      __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const f32x4 v = acc[NR][XMTP] + *reinterpret_cast<const f32x4*>(lds + kScratch2Off + XMTP * 256 + 4 * lane);
    store_p1_mt<30>(b30, v, pxx, kq, XMTP);
  }
}

constexpr int kRolePlain = 0, kRoleReducer = 1, kRoleHelper = 2;
using v3::kFlagOff;
using v3::kL3Cut1;
using v3::kL3Cut2;
using v3::kL3Cut3;
using v3::kScratchOff;

template <int ROLE, int HID>
__device__ __forceinline__ void layer3(const Params& P, float* lds, const float* w, int blk, int wave, int lane,
                                       unsigned tag, int utt, int t0, f32x4 (&skip_ce1)[2], f32x4 (&skip_ce2)[2]) {
  constexpr int NR = 1, NX = ROLE == kRolePlain ? 0 : 1, NT = NR + NX;
  constexpr int NEPI = ROLE == kRoleHelper ? NR : NT;
  lane = opaque(lane);
  const int n = lane & 15, kq = lane >> 4;
  const float* b30 = lds + kB30Off + kB30Pad * 30;
  float* b8 = lds + kB8Off + kB8Pad * kB8S;
  const int q0 = 16 * wave + n, qx = 16 * 16 + n;   // pixel pair indices
  f32x4 acc[NT][1];
  const f32x4 sh = *reinterpret_cast<const f32x4*>(w + kW3Data + 4 * (kq & 1));
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t][0] = (t == NR && ROLE == kRoleHelper) ? f32x4{0.f, 0.f, 0.f, 0.f} : sh;
  {
    constexpr int XS0 = ROLE != kRoleHelper ? 0 : HID == 1 ? kL3Cut1 : HID == 2 ? kL3Cut2 : kL3Cut3;
    constexpr int XS1 = ROLE == kRoleReducer ? kL3Cut1 : ROLE != kRoleHelper ? kL3Steps
                        : HID == 1 ? kL3Cut2 : HID == 2 ? kL3Cut3 : kL3Steps;
    constexpr bool XT = ROLE == kRoleHelper && HID == 3;
    v3::gemm_pass<NR, NX, 1, -1, kL3Steps, XS0, XS1, XT, 0, 2>(b30, (2 * q0 - 4) * 30 + 2 * kq, (2 * qx - 4) * 30 + 2 * kq,
                                                               -kq, w, lane, acc);
  }
  if constexpr (ROLE == kRoleHelper) {
    *reinterpret_cast<f32x4*>(lds + kScratchOff + (HID - 1) * 256 + 4 * lane) = acc[NR][0];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) lds_poke(lds + kFlagOff + (HID - 1), tag);
  }
  if constexpr (ROLE == kRoleReducer) {
#pragma unroll
    for (int h = 0; h < 3; ++h) {
      for (int spin = 0; spin < (1 << 22); ++spin) {
        if (__builtin_amdgcn_readfirstlane(lds_peek(lds + kFlagOff + h)) == tag) break;
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
    for (int h = 0; h < 3; ++h) acc[NR][0] += *reinterpret_cast<const f32x4*>(lds + kScratchOff + h * 256 + 4 * lane);
  }
#pragma unroll
  for (int t = 0; t < NEPI; ++t) {
    const int q = (t < NR) ? q0 : qx;
    const int px = 2 * q + (kq >> 1);
    f32x4 v = relu4(acc[t][0]);
    if (blk == 3) v += skip_ce2[t];   // CD1 + CE2 (model.py:87, 75-76: after the ReLU)
    if (blk == 4) v += skip_ce1[t];   // CD2 + CE1 (model.py:88)
    const bool gap = span_has_gap(32 * (t < NR ? wave : 16), 32);
    const int fr = px / kS, f = px - fr * kS;
    const bool ok = gap ? (px < kNPX && f < kF) : true;
    if (gap && !ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
    skip_ce1[t] = (blk == 0) ? v : skip_ce1[t];
    skip_ce2[t] = (blk == 1) ? v : skip_ce2[t];
    if (blk < 4) {
      if (gap && px >= kNPX) continue;
      float* bp = b8 + px * kB8S + 4 * (kq & 1);
      *reinterpret_cast<f32x2*>(bp) = f32x2{v.x, v.y};
      *reinterpret_cast<f32x2*>(bp + 2) = f32x2{v.z, v.w};
    } else if (ok && t0 + fr < P.T) {
      float* hp = P.h + (((size_t)utt * P.T + t0 + fr) * kF + f) * kHCh + 4 * (kq & 1);
      *reinterpret_cast<f32x4*>(hp) = v;
    }
  }
}

// input rows of the next tile: 2 floats per thread, loaded one tile ahead
struct XStage {
  float v0, v1;
};
static_assert(kX0Floats <= 2 * kThreads, "XStage holds 2 floats per thread");
__device__ __forceinline__ XStage xstage_load(const Params& P, int tile, int tid) {
  const bool live = tile < P.total_tiles;
  const int utt = live ? tile / P.tiles_per_utt : 0;
  const int t0 = live ? (tile - utt * P.tiles_per_utt) * kTF : 0;
  const float* xu = P.x + (size_t)utt * P.T * kF;
  XStage st;
  st.v0 = v3::xstage_one(P, live, xu, t0, tid);
  st.v1 = v3::xstage_one(P, live, xu, t0, tid + kThreads);
  return st;
}
__device__ __forceinline__ void xstage_store(const XStage& st, float* x0, int tid) {
  x0[tid] = st.v0;
  if (tid + kThreads < kX0Floats) x0[tid + kThreads] = st.v1;
}

__global__ __launch_bounds__(kThreads) void fused_v3w_kernel(Params P) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* const wbase = lds + kWOff;
#define WREG(i) (wbase + (i) * kWRegion)
  for (int e = tid; e < kLdsFloats; e += kThreads) lds[e] = 0.f;
  __syncthreads();
  packet_dma<kW1>(P.wpack, WREG(0), wave, lane);
  int wcur = 0;
  unsigned epoch = 0;
  XStage xst = v3w::xstage_load(P, blockIdx.x, tid);
  layer_end_sync();
  // layer-1 remainder tile of this wave: waves 1..4 -> tiles 0..3, wave 7 -> tile 4
  const int xr = wave == 7 ? 4 : wave - 1;
  const bool has_rem = (wave >= 1 && wave <= 4) || wave == 7;

  for (int tile = blockIdx.x; tile < P.total_tiles; tile += gridDim.x) {
    const int utt = tile / P.tiles_per_utt;
    const int t0 = (tile - utt * P.tiles_per_utt) * kTF;
    v3w::xstage_store(xst, lds + kX0Off, tid);
    __syncthreads();

    f32x4 skip_ce1[2], skip_ce2[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) skip_ce1[t] = skip_ce2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* wsrc = P.wpack;

#pragma unroll 1
    for (int blk = 0; blk < 5; ++blk) {
      {  // ---- layer 1
        packet_dma<kW2>(wsrc + kW1, WREG(wcur ^ 1), wave, lane);
        const float* w = WREG(wcur);
        if (wave == 0) layer1<1, 0>(lds, w, blk == 0, wave, lane, 0);
        else if (has_rem) layer1<0, 1>(lds, w, blk == 0, wave, lane, xr);
        else layer1<0, 0>(lds, w, blk == 0, wave, lane, 0);
        wcur ^= 1;
        layer_end_sync();
      }
      {  // ---- layer 2
        packet_dma<kW3>(wsrc + kW1 + kW2, WREG(wcur ^ 1), wave, lane);
        const float* w = WREG(wcur);
        const unsigned tag2 = 0xC0000000u | (epoch + 1u);
        if (wave == 0) layer2<kL2Helper, 0>(lds, w, wave, lane, tag2);
        else if (wave == 1) layer2<kL2Helper, 1>(lds, w, wave, lane, tag2);
        else if (wave == 2) layer2<kL2Reducer, 0>(lds, w, wave, lane, tag2);
        else if (wave == 3) layer2<kL2Reducer, 1>(lds, w, wave, lane, tag2);
        else layer2<kL2Plain, 0>(lds, w, wave, lane, tag2);
        wcur ^= 1;
        layer_end_sync();
      }
      {  // ---- layer 3
        packet_dma<kW1>(blk == 4 ? P.wpack : wsrc + kWBlock, WREG(wcur ^ 1), wave, lane);
        if (blk == 4) xst = v3w::xstage_load(P, tile + gridDim.x, tid);
        const float* w = WREG(wcur);
        ++epoch;
        const unsigned tag = 0x80000000u | epoch;
        if (wave == 0) layer3<kRoleReducer, 0>(P, lds, w, blk, wave, lane, tag, utt, t0, skip_ce1, skip_ce2);
        else if (wave == 1) layer3<kRoleHelper, 1>(P, lds, w, blk, wave, lane, tag, utt, t0, skip_ce1, skip_ce2);
        else if (wave == 2) layer3<kRoleHelper, 2>(P, lds, w, blk, wave, lane, tag, utt, t0, skip_ce1, skip_ce2);
        else if (wave == 3) layer3<kRoleHelper, 3>(P, lds, w, blk, wave, lane, tag, utt, t0, skip_ce1, skip_ce2);
        else layer3<kRolePlain, 0>(P, lds, w, blk, wave, lane, tag, utt, t0, skip_ce1, skip_ce2);
        wcur ^= 1;
        layer_end_sync();
      }
      wsrc += kWBlock;
    }
  }
#undef WREG
}

}  // namespace v3w
}  // namespace rced
