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
//     L2 -> LDS by LDS-DMA one layer ahead (ping-pong), each packet carrying its 32 shift values;
//   * cout sits on the MFMA M axis (16 rows).  30 pads to 2 M-tiles; the 30->8 layers use two
//     pixel phases as rows (8 cout x 2 adjacent pixels = 16 rows, K = 10 taps instead of 9); the
//     ->18 layers run channels 0..15 as one M-tile plus a remainder pass that computes channels
//     16,17 for 8 adjacent pixels at once (rows = 8 phases x 2 channels, K = 16 taps);
//   * the two CR-CED block skips (model.py:75-76, added after ReLU) never touch LDS: they stay in
//     the accumulator registers of the wave that produced them;
//   * operands are software-pipelined by hand (read step s+1 or s+2, then the MFMAs of step s):
//     left alone, hipcc issues each LDS read right before its use and the MFMA pipe starves.
//
// Pixel space of a tile: kTF frames, frame i at flat pixels [i*kS, i*kS+129); the kS-129 = 4 gap
// pixels between frames are always zero and serve as the SAME-padding halo of both neighbours.
#pragma once
#include <hip/hip_runtime.h>

#include "lds_dma.h"

#ifndef RCED_EXP_NOBAR
#define RCED_EXP_NOBAR 0  // timing experiment only (results wrong): 1 = no per-layer barrier, no hand-off waits (waves drift
                          // freely inside a tile); >= 2 = no barrier at all and waves 4..7 start ~4 k cycles * value late
#endif
#ifndef RCED_EXP_NOEPI
#define RCED_EXP_NOEPI 0  // timing experiments only: bit0/1/2 drop the epilogue (ReLU + LDS stores) of L1/L2/L3 (results wrong)
#endif
#ifndef RCED_EXP_SKIP
#define RCED_EXP_SKIP 0   // timing experiments only: bit0 skip L1 math, bit1 L2, bit2 L3 (results wrong)
#endif
#ifndef RCED_D1
#define RCED_D1 2   // operand prefetch depth (steps) of the layer-1 / layer-2 / layer-3 passes
#endif
#ifndef RCED_D2
#define RCED_D2 1
#endif
#ifndef RCED_D3
#define RCED_D3 2
#endif
#ifndef RCED_EXP_WGLOBAL
#define RCED_EXP_WGLOBAL 0   // experiment: A fragments straight from global/L2 instead of the LDS packet
#endif
#ifndef RCED_STAMPS
#define RCED_STAMPS 0     // diagnostic build: s_memtime stamps around every layer's math and barrier
#endif

namespace rced {
namespace v3 {

#if RCED_STAMPS
__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define STAMP_BEGIN() const unsigned long long st_a_ = stamp()
#define STAMP_MATH(i) const unsigned long long st_b_ = stamp(); tsum[i] += st_b_ - st_a_
#define STAMP_WAIT(i) tsum[(i) + ((i) < 3 ? 3 : 1)] += stamp() - st_b_
#else
#define STAMP_BEGIN()
#define STAMP_MATH(i)
#define STAMP_WAIT(i)
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kF = 129;
constexpr int kTF = 4;                       // frames per tile
constexpr int kS = 133;                      // pixel stride of a frame (129 + 4 zero gap)
constexpr int kNPX = kTF * kS;               // 532 pixels per tile
constexpr int kWaves = 8;
constexpr int kThreads = kWaves * 64;
constexpr int kHCh = 8;                      // channels of the tensor handed to the final layer

// ---- LDS map, in floats -----------------------------------------------------------------
// B8 keeps its 8 channels at a pixel stride of 10 floats: with 8 channels one b64-step of the
// 9-tap window is exactly one pixel, so the stride is free, and 10 turns the remainder pass's
// 16-way bank conflict (column stride 8 pixels x 8 floats = 64 banks) into 4-way.
// Rows past the last pixel any VALID output needs are not allocated: reads that run past a buffer
// land in the next one (always finite floats) and only feed masked outputs.
constexpr int kB8S = 10;                                  // B8 pixel stride (floats)
constexpr int kB8Pad = 4, kB18Pad = 2, kB30Pad = 4;       // leading zero rows (pixels -pad..-1)
constexpr int kB8Rows = kB8Pad + kNPX + 4;                // pixels -4 .. 535
constexpr int kB18Rows = kB18Pad + kNPX;                  // pixels -2 .. 531
constexpr int kB30Rows = kB30Pad + kNPX;                  // pixels -4 .. 531
constexpr int kB8Off = 0;
constexpr int kB18Off = kB8Off + kB8Rows * kB8S;
constexpr int kB30Off = kB18Off + kB18Rows * 18;
constexpr int kWRegion = 37 * 128 + 64 + 32;              // largest packet: 30->8 (b64 steps + tail + shifts)
constexpr int kWOff = kB30Off + kB30Rows * 30;
constexpr int kLdsFloats = kWOff + 2 * kWRegion;
constexpr int kLdsBytes = kLdsFloats * 4;
static_assert(kLdsBytes <= 160 * 1024, "LDS budget");
static_assert((kWOff * 4) % 16 == 0 && (kWRegion * 4) % 16 == 0, "LDS-DMA destinations are 16-byte aligned");
static_assert((kB18Off % 2) == 0 && (kB30Off % 2) == 0, "8-byte aligned buffers");
// input rows of the 8x9 first layer alias the (not yet live) B30 buffer, from its pixel-0 row on
constexpr int kX0Rows = kTF + 7;
constexpr int kX0Floats = ((kX0Rows * kS + 24 + 3) / 4) * 4;   // 1488: max main-pass index 527 + 7*133 + 8
constexpr int kX0Off = kB30Off + kB30Pad * 30;
static_assert(kX0Floats <= 60 * 30, "X0 must sit inside rows that layer 2 rewrites");

// ---- packed weight stream (floats), per layer --------------------------------------------
//  first layer main (8x9x1 -> ch 0..15): 18 k-steps x 64 lanes (b32 steps, k = (time tap, freq tap))
//  first layer rem  (ch 16,17 x 8 phases): 32 k-steps x 64 lanes (k = (time tap, 16 freq taps))
//  L1 main (1x9, 8 -> ch 0..15):  9 b64-steps x 64 lanes x 2
//  L1 rem  (ch 16,17 x 8 phases): 16 b64-steps x 64 x 2   (K = 16 taps x 8)
//  L2 (1x5, 18->30): 11 b64-steps x 2 M-tiles x 64 x 2 + a b32 tail step (k = 88 + kq; K = 90)
//  L3 (1x9, 30->8):  37 b64-steps x 64 x 2 + a b32 tail step (k = 296 + kq; K = 300 = 10 taps x 30,
//                    rows = 2 pixel phases x 8 channels)
// Every packet ends with its 32 shift values (bias + folded BatchNorm).
constexpr int kShiftPerLayer = 32;
constexpr int kW1Main = 9 * 128;          // 1152 (= 18 * 64 for the first layer)
constexpr int kW1Rem = 16 * 128;          // 2048 (= 32 * 64 for the first layer)
constexpr int kW1Data = kW1Main + kW1Rem; // 3200
constexpr int kL2Steps = 11, kL3Steps = 37;                  // b64 steps; each pass ends with one b32 step
constexpr int kW2Data = kL2Steps * 2 * 128 + 2 * 64;   // 2944
constexpr int kW3Data = kL3Steps * 128 + 64;           // 4800
constexpr int kW1 = kW1Data + kShiftPerLayer;
constexpr int kW2 = kW2Data + kShiftPerLayer;
constexpr int kW3 = kW3Data + kShiftPerLayer;
constexpr int kWBlock = kW1 + kW2 + kW3;
constexpr int kWTotal = 5 * kWBlock;

// ---- tile -> wave assignment ---------------------------------------------------------------
// 16-pixel tiles 0..32 (pixels 0..527; 528..531 is gap): wave w owns tiles w + 8*slot, slot < 4
// ("regular": one address register per wave, everything else immediates); tile 32 and, in layer 1,
// tile 31 are handed out as "extra" tiles.  Pair tiles 0..16: w + 8*slot, slot < 2; extra 16.
// Waves w and w+4 share a SIMD, and the barrier that ends a layer waits for the most loaded SIMD, so
// what is balanced is each LAYER's MFMA count per SIMD:
//   layer 1: remainder tiles (128 pixels) 0..4 go one each to waves 4,5,6 and two to wave 7, which gives up
//            main tile 31 (to wave 1; tile 32 to wave 0): 194, 194, 176, 190 per SIMD (18 per main tile,
//            32 per remainder tile).
//   layer 2: tile 32 is cut in four, M-tile x K-half, one piece on each of waves 0..3: 379.5 everywhere.
//   layer 3: pair tile 16 is cut in four along K on waves 0..3: 318.75 everywhere.
// The cut tiles are put back together through LDS scratch + tagged flag words (pairwise, no extra barrier).

struct Params {
  const float* x;       // [N, T, 129]
  float* h;             // [N*T, 129, 8]  output of CD2 (input of decode_final)
  const float* wpack;   // kWTotal floats
  int N, T;
  int tiles_per_utt;    // ceil(T / kTF)
  int total_tiles;      // N * tiles_per_utt
  unsigned long long* stamps;  // diagnostic builds only (RCED_STAMPS): [wave][8] cycle sums of workgroup 0
};

__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void pin() { __builtin_amdgcn_sched_barrier(0); }

// Volatile accesses to LDS words used for wave-to-wave signalling.  Through a generic pointer hipcc emits
// flat_load/flat_store (both memory pipes, s_waitcnt vmcnt(0) in every poll); typed as LDS they are ds_read/ds_write.
__device__ __forceinline__ unsigned lds_peek(const void* p) {
  return *(volatile __attribute__((address_space(3))) const unsigned*)p;
}
__device__ __forceinline__ void lds_poke(void* p, unsigned v) {
  *(volatile __attribute__((address_space(3))) unsigned*)p = v;
}

// Experiment switch: re-derive lane coordinates behind an opaque barrier in every layer, so that hipcc
// does not hoist every layer's address arithmetic out of the loops (VGPRs 238 -> 136, but the
// recomputation costs 3 % here; the chain kernel needs it to avoid spills).
// Always-on variant for the few values only the split-tile code needs: recomputed where used (3-4 VALU)
// instead of being hoisted out of the tile loop by LICM and kept live in VGPRs across all fifteen layers.
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}
#ifdef RCED_V3_OPAQUE
#define OPAQUE_LANE(l) asm volatile("" : "+v"(l))
#else
#define OPAQUE_LANE(l)
#endif

// Stream one weight packet global -> LDS with LDS-DMA (no VGPR staging, no ds_write): wave w copies
// the 1-KiB chunks w, w+8, w+16; lane l of a chunk moves 16 bytes.  Completion: vmcnt(0) + barrier
// at the end of the layer during which it was issued.
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
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA pieces have landed
  if (!RCED_EXP_NOBAR) __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// Implicit-GEMM pass: NB64 b64 steps (k = 8s + 2kq + e) plus one b32 tail step (k = 8*NB64 + kq),
// hand-pipelined DEPTH steps ahead.
//   NR regular slots at float offsets off0 + t*STRIDE run every step and every M-tile.
//   NX (0/1) extra slot at offx runs b64 steps [XS0, XS1), the tail iff XTAIL, and only M-tile XMT
//   (XMT < 0: all) -- this is how an odd tile is shared between two waves (by M-tile or along K).
//   w: LDS packet [NB64][MT][lane][2] then tail [MT][lane].  tailoff: per-lane float delta of the
//   tail read relative to off + 8*NB64 (kq_eff - 2*kq: lanes past K re-read in-window data, their
//   weights are zero).
// ---------------------------------------------------------------------------------------------
template <int NR, int NX, int MT, int XMT, int NB64, int XS0, int XS1, bool XTAIL, int STRIDE, int DEPTH>
__device__ __forceinline__ void gemm_pass(const float* act, int off0, int offx, int tailoff, const float* w, int lane,
                                          f32x4 (&acc)[NR + NX][MT]) {
  constexpr int NT = NR + NX;
  constexpr int RING = DEPTH + 1;
  const f32x2* wp = reinterpret_cast<const f32x2*>(w) + lane;
  const float* wt = w + NB64 * MT * 128 + lane;
  f32x2 a[RING][MT], b[RING][NT];
  float at[MT], bt[NT];
  auto xlive = [](int s) { return NX > 0 && s >= XS0 && s < XS1; };
  auto xmt = [](int mt) { return XMT < 0 || mt == XMT; };
  auto load = [&](int s, f32x2(&as)[MT], f32x2(&bs)[NT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) as[mt] = wp[(s * MT + mt) * 64];
#pragma unroll
    for (int t = 0; t < NR; ++t) bs[t] = *reinterpret_cast<const f32x2*>(act + off0 + t * STRIDE + 8 * s);
    if constexpr (NX > 0)
      if (xlive(s)) bs[NR] = *reinterpret_cast<const f32x2*>(act + offx + 8 * s);
  };
  // tail operands first: they are needed last and cost MT + NT registers
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) at[mt] = wt[mt * 64];
#pragma unroll
  for (int t = 0; t < NR; ++t) bt[t] = act[off0 + t * STRIDE + 8 * NB64 + tailoff];
  if constexpr (NX > 0 && XTAIL) bt[NR] = act[offx + 8 * NB64 + tailoff];
#pragma unroll
  for (int s = 0; s < DEPTH && s < NB64; ++s) load(s, a[s % RING], b[s % RING]);
#pragma unroll
  for (int s = 0; s < NB64; ++s) {
    if (s + DEPTH < NB64) load(s + DEPTH, a[(s + DEPTH) % RING], b[(s + DEPTH) % RING]);
    pin();
#pragma unroll
    for (int e = 0; e < 2; ++e) {
#pragma unroll
      for (int t = 0; t < NR; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[t][mt] = mfma(a[s % RING][mt][e], b[s % RING][t][e], acc[t][mt]);
      if constexpr (NX > 0)
        if (xlive(s)) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            if (xmt(mt)) acc[NR][mt] = mfma(a[s % RING][mt][e], b[s % RING][NR][e], acc[NR][mt]);
        }
    }
    pin();
  }
#pragma unroll
  for (int t = 0; t < NR; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[t][mt] = mfma(at[mt], bt[t], acc[t][mt]);
  if constexpr (NX > 0 && XTAIL) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      if (xmt(mt)) acc[NR][mt] = mfma(at[mt], bt[NR], acc[NR][mt]);
  }
}

// Layer 1 of blocks 1..4: main pass (NM = NMR regular + NMX extra tiles, 9 b64-steps) and remainder
// pass (NR tiles of 8-pixel columns, 16 b64-steps) issued as ONE pipelined stream of 16 slots; the
// remainder accumulates even / odd steps in two independent chains.
template <int NMR, int NMX, int NR>
__device__ __forceinline__ void l1_pass(const float* b8, int offm0, int offmx, const int (&offr)[NR == 0 ? 1 : NR],
                                        const float* w, int lane, f32x4 (&accm)[NMR + NMX][1],
                                        f32x4 (&accr)[NR == 0 ? 1 : NR][2]) {
  constexpr int NM = NMR + NMX, NRA = NR == 0 ? 1 : NR, DEPTH = RCED_D1, RING = DEPTH + 1;
  constexpr int SLOTS = NR > 0 ? 16 : 9;
  const f32x2* wm = reinterpret_cast<const f32x2*>(w) + lane;
  const f32x2* wr = reinterpret_cast<const f32x2*>(w + kW1Main) + lane;
  f32x2 am[RING], bm[RING][NM], ar[RING], br[RING][NRA];
  // main step issued in slot i (or -1): 9 main steps spread over the 16 remainder steps
  auto main_step = [](int i) { return NR > 0 ? ((i * 9) / 16 != ((i + 1) * 9) / 16 ? (i * 9) / 16 : -1) : i; };
  auto load = [&](int i, int buf) {
    if constexpr (NR > 0) {
      ar[buf] = wr[i * 64];
#pragma unroll
      for (int t = 0; t < NR; ++t) br[buf][t] = *reinterpret_cast<const f32x2*>(b8 + offr[t] + kB8S * i);
    }
    const int m = main_step(i);
    if (m >= 0) {
      am[buf] = wm[m * 64];
#pragma unroll
      for (int t = 0; t < NMR; ++t)
        bm[buf][t] = *reinterpret_cast<const f32x2*>(b8 + offm0 + t * 128 * kB8S + kB8S * m);
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
      if constexpr (NR > 0) {
#pragma unroll
        for (int t = 0; t < NR; ++t) accr[t][i & 1] = mfma(ar[buf][e], br[buf][t][e], accr[t][i & 1]);
      }
      if (main_step(i) >= 0) {
#pragma unroll
        for (int t = 0; t < NM; ++t) accm[t][0] = mfma(am[buf][e], bm[buf][t][e], accm[t][0]);
      }
    }
    pin();
  }
}

// The same for block 0 (8x9 kernel on the 1-channel input, b32 steps): main 18 k-steps (ih, j<9),
// remainder 32 k-steps (ih, u<16); lane kq <-> time tap 4*ih + kq.
template <int NMR, int NMX, int NR>
__device__ __forceinline__ void l1_first_pass(const float* x0, int offm0, int offmx,
                                              const int (&offr)[NR == 0 ? 1 : NR], const float* w, int lane,
                                              f32x4 (&accm)[NMR + NMX][1], f32x4 (&accr)[NR == 0 ? 1 : NR][2]) {
  constexpr int NM = NMR + NMX, NRA = NR == 0 ? 1 : NR, DEPTH = 3, RING = DEPTH + 1;
  constexpr int SLOTS = NR > 0 ? 32 : 18;
  const float* wm = w + lane;
  const float* wr = w + kW1Main + lane;
  float am[RING], bm[RING][NM], ar[RING], br[RING][NRA];
  auto main_step = [](int i) { return NR > 0 ? ((i * 18) / 32 != ((i + 1) * 18) / 32 ? (i * 18) / 32 : -1) : i; };
  auto load = [&](int i, int buf) {
    if constexpr (NR > 0) {
      ar[buf] = wr[i * 64];
#pragma unroll
      for (int t = 0; t < NR; ++t) br[buf][t] = x0[offr[t] + (i / 16) * 4 * kS + (i % 16)];
    }
    const int m = main_step(i);
    if (m >= 0) {
      am[buf] = wm[m * 64];
      const int d = (m / 9) * 4 * kS + (m % 9);
#pragma unroll
      for (int t = 0; t < NMR; ++t) bm[buf][t] = x0[offm0 + t * 128 + d];
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
    if constexpr (NR > 0) {
#pragma unroll
      for (int t = 0; t < NR; ++t) accr[t][i & 1] = mfma(ar[buf], br[buf][t], accr[t][i & 1]);
    }
    if (main_step(i) >= 0) {
#pragma unroll
      for (int t = 0; t < NM; ++t) accm[t][0] = mfma(am[buf], bm[buf][t], accm[t][0]);
    }
    pin();
  }
}

// ReLU as ONE integer max per element: for IEEE floats max(bits, 0) == bits of relu(x) (negative
// floats are negative ints).  fmaxf() on an MFMA result makes hipcc add a canonicalising
// v_max_f32 v,v,v in front (2 VALU per element); an inline-asm v_max_f32 is NOT an option: hipcc pads
// no MFMA -> VALU wait states inside asm, and a ReLU issued right behind the last MFMA then reads the
// accumulator before it lands (seen as rare wrong values in each wave's first tile).
__device__ __forceinline__ float relu1(float v) {
  const int b = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, b > 0 ? b : 0);
}
__device__ __forceinline__ f32x4 relu4(f32x4 v) {
  return f32x4{relu1(v.x), relu1(v.y), relu1(v.z), relu1(v.w)};
}

__device__ __forceinline__ bool px_valid(int px) {   // a real frequency bin (not gap, not past the tile)
  const int fr = px / kS;
  return px < kNPX && (px - fr * kS) < kF;
}

// Wave-uniform: does the pixel span [p0, p0+len) contain a gap pixel or run past the tile?  Only 3 of
// the 33 sixteen-pixel tiles and 4 of the 17 pair tiles do, so the masking VALU is behind a scalar branch.
__device__ __forceinline__ bool span_has_gap(int p0, int len) {
  const int fr = p0 / kS;
  return p0 + len > kNPX || (p0 - fr * kS) + len > kF;
}

// Epilogue of a P = 1 pass for one slot (rows = 16*mt + 4*kq + j output channels, column = pixel):
// ReLU, zero the gap pixels, store [pixel][COUT] with 8-byte stores.
template <int MT, int COUT>
__device__ __forceinline__ void store_p1(float* out, const f32x4 (&acc)[MT], int px, int kq, bool gap) {
  const bool ok = gap ? px_valid(px) : true;   // `gap` is wave-uniform
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int co0 = 16 * mt + 4 * kq;
    f32x4 v = relu4(acc[mt]);
    if (gap && !ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
    float* p = out + px * COUT + co0;
    if (co0 + 1 < COUT) *reinterpret_cast<f32x2*>(p) = f32x2{v.x, v.y};
    if (co0 + 3 < COUT) *reinterpret_cast<f32x2*>(p + 2) = f32x2{v.z, v.w};
  }
}

// One M-tile of the same (used where a tile's two M-tiles belong to different waves).
template <int COUT>
__device__ __forceinline__ void store_p1_mt(float* out, f32x4 acc, int px, int kq, int mt) {
  const int co0 = 16 * mt + 4 * kq;   // only used for tile 32 (pixels 512..527): no gap inside
  f32x4 v = relu4(acc);
  float* p = out + px * COUT + co0;
  if (co0 + 1 < COUT) *reinterpret_cast<f32x2*>(p) = f32x2{v.x, v.y};
  if (co0 + 3 < COUT) *reinterpret_cast<f32x2*>(p + 2) = f32x2{v.z, v.w};
}

// Epilogue of the layer-1 remainder pass: rows 4*kq+j = (phase 2*kq + (j>>1), channel 16 + (j&1)),
// column = pixel octet.  Two 8-byte stores per lane: channels 16,17 of two adjacent pixels.
__device__ __forceinline__ void store_rem(float* b18, f32x4 acc, int px0, int kq, bool gap) {
  const f32x4 v = relu4(acc);
  const int pa = px0 + 2 * kq;
  if (!gap || px_valid(pa)) *reinterpret_cast<f32x2*>(b18 + pa * 18 + 16) = f32x2{v.x, v.y};
  if (!gap || px_valid(pa + 1)) *reinterpret_cast<f32x2*>(b18 + (pa + 1) * 18 + 16) = f32x2{v.z, v.w};
}

// The 11 input rows of a tile (frames t0-3 .. t0+7 of one utterance): 3 floats per thread, loaded one
// tile ahead into registers, written to the X0 area (aliasing B30) once B30 is dead.
struct XStage {
  float v0, v1, v2;
};
__device__ __forceinline__ float xstage_one(const Params& P, bool live, const float* xu, int t0, int e) {
  const int q = e - 4;
  const int r = q >= 0 ? q / kS : -1;
  const int f = q - r * kS;
  const int tt = t0 + r - 3;
  float v = 0.f;
  if (live && e < kX0Floats && q >= 0 && r < kX0Rows && f < kF && tt >= 0 && tt < P.T) v = xu[(size_t)tt * kF + f];
  return v;
}
__device__ __forceinline__ XStage xstage_load(const Params& P, int tile, int tid) {
  const bool live = tile < P.total_tiles;
  const int utt = live ? tile / P.tiles_per_utt : 0;
  const int t0 = live ? (tile - utt * P.tiles_per_utt) * kTF : 0;
  const float* xu = P.x + (size_t)utt * P.T * kF;
  XStage st;
  st.v0 = xstage_one(P, live, xu, t0, tid);
  st.v1 = xstage_one(P, live, xu, t0, tid + kThreads);
  st.v2 = xstage_one(P, live, xu, t0, tid + 2 * kThreads);
  return st;
}
__device__ __forceinline__ void xstage_store(const XStage& st, float* x0, int tid) {
  x0[tid] = st.v0;
  x0[tid + kThreads] = st.v1;
  if (tid + 2 * kThreads < kX0Floats) x0[tid + 2 * kThreads] = st.v2;
}
static_assert(kX0Floats <= 3 * kThreads, "XStage holds 3 floats per thread");

// ---- the three layer kinds, templated on the calling wave's tile signature -------------------
template <int NMR, int NMX, int NR>
__device__ __forceinline__ void layer1(float* lds, const float* w, bool first, int wave, int lane, int xm, int xr0,
                                       int xr1) {
  constexpr int NM = NMR + NMX, NRA = NR == 0 ? 1 : NR;
  OPAQUE_LANE(lane);
  const int n = lane & 15, kq = lane >> 4;
  float* b8 = lds + kB8Off + kB8Pad * kB8S;
  float* b18 = lds + kB18Off + kB18Pad * 18;
  const float* x0 = lds + kX0Off;
  const int px0 = 16 * wave + n, pxx = 16 * xm + n;
  int pxr[NRA] = {8 * (16 * xr0 + n)};
  if constexpr (NR > 1) pxr[1] = 8 * (16 * xr1 + n);
  int offr[NRA];
  f32x4 accm[NM][1], accr[NRA][2];
  const f32x4 sh = *reinterpret_cast<const f32x4*>(w + kW1Data + 4 * kq);
  const f32x2 s2 = *reinterpret_cast<const f32x2*>(w + kW1Data + 16);
#pragma unroll
  for (int t = 0; t < NM; ++t) accm[t][0] = sh;
#pragma unroll
  for (int t = 0; t < NRA; ++t) {
    accr[t][0] = f32x4{s2.x, s2.y, s2.x, s2.y};
    accr[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if (!(RCED_EXP_SKIP & 1)) {
    if (first) {
#pragma unroll
      for (int t = 0; t < NRA; ++t) offr[t] = pxr[t] + kq * kS;
      l1_first_pass<NMR, NMX, NR>(x0, px0 + kq * kS, pxx + kq * kS, offr, w, lane, accm, accr);
    } else {
#pragma unroll
      for (int t = 0; t < NRA; ++t) offr[t] = (pxr[t] - 4) * kB8S + 2 * kq;
      l1_pass<NMR, NMX, NR>(b8, (px0 - 4) * kB8S + 2 * kq, (pxx - 4) * kB8S + 2 * kq, offr, w, lane, accm, accr);
    }
  }
#pragma unroll
  for (int t = 0; t < NMR; ++t)
    if (!(RCED_EXP_NOEPI & 1) || accm[t][0].x == 12345.678f)
      store_p1<1, 18>(b18, accm[t], px0 + 128 * t, kq, span_has_gap(16 * (wave + 8 * t), 16));
  if constexpr (NMX > 0)
    if (!(RCED_EXP_NOEPI & 1) || accm[NMR][0].x == 12345.678f) store_p1<1, 18>(b18, accm[NMR], pxx, kq, span_has_gap(16 * xm, 16));
  if constexpr (NR > 0) {
#pragma unroll
    for (int t = 0; t < NR; ++t)
      if (!(RCED_EXP_NOEPI & 1) || accr[t][0].x + accr[t][1].x == 12345.678f)
        store_rem(b18, accr[t][0] + accr[t][1], pxr[t], kq, span_has_gap(128 * (t == 0 ? xr0 : xr1), 128));
  }
}

#if RCED_STAMPS
__device__ unsigned long long g_fine[8][4];   // [wave][prologue, gemm, epilogue, -] of layer 2, workgroup 0
#endif

// Layer 2.  Every wave has four regular tiles.  Tile 32 (pixels 512..527) is cut in four equal pieces, one per
// SIMD, so that the layer's MFMA count is the same on every SIMD: M-tile XMT x K-half.  Waves 0 / 1 are the
// helpers (steps [0, kL2Cut) of M-tile 0 / 1), waves 2 / 3 the reducers (steps [kL2Cut, 11) + tail of M-tile
// 0 / 1; they add the helper's partial sums and own the epilogue).  Hand-off as in layer 3, through scratch
// in the B8 buffer, which is dead during layer 2 (layer 3 rewrites every real pixel of it).
constexpr int kL2Cut = 6;
constexpr int kScratch2Off = kB8Off + kB8Pad * kB8S + 8 * kB8S;   // B8 rows 8.. of frame 0 (2 x 256 floats + 2 flags)
constexpr int kFlag2Off = kScratch2Off + 2 * 256;
static_assert((kScratch2Off * 4) % 16 == 0, "scratch is read/written with b128");
static_assert(8 + (2 * 256 + 2 + kB8S - 1) / kB8S <= kF, "layer-2 scratch stays inside frame 0's real pixels");
constexpr int kL2Plain = 0, kL2Reducer = 1, kL2Helper = 2;

template <int ROLE, int XMTP>   // XMTP: the M-tile of tile 32 this wave works on (plain: unused)
__device__ __forceinline__ void layer2(float* lds, const float* w, int wave, int lane, unsigned tag) {
  constexpr int NX = ROLE == kL2Plain ? 0 : 1, NT = 4 + NX;
  constexpr int XMT = ROLE == kL2Plain ? -1 : XMTP;
  OPAQUE_LANE(lane);
  const int n = lane & 15, kq = lane >> 4;
  const float* b18 = lds + kB18Off + kB18Pad * 18;
  float* b30 = lds + kB30Off + kB30Pad * 30;
#if RCED_STAMPS
  const unsigned long long f0 = stamp();
#endif
  const int lx = NX > 0 ? opaque(lane) : lane;   // lane copy for the split tile's addresses (see opaque())
  const int px0 = 16 * wave + n, pxx = 16 * 32 + (lx & 15), kqx = lx >> 4;
  f32x4 acc[NT][2];
  f32x4 sh[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) sh[mt] = *reinterpret_cast<const f32x4*>(w + kW2Data + 16 * mt + 4 * kq);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const bool partial = t == 4 && ROLE == kL2Helper;   // the helper's share starts from zero, the reducer's from the shift
    acc[t][0] = partial ? f32x4{0.f, 0.f, 0.f, 0.f} : sh[0];
    acc[t][1] = partial ? f32x4{0.f, 0.f, 0.f, 0.f} : sh[1];
  }
  const int tailoff = (kq < 1 ? kq : 1) - 2 * kq;   // K = 90: tail k = 88 + kq is real for kq < 2
#if RCED_STAMPS
  const unsigned long long f1 = stamp();
#endif
  if (!(RCED_EXP_SKIP & 2)) {
    constexpr int XS0 = ROLE == kL2Reducer ? kL2Cut : 0;
    constexpr int XS1 = ROLE == kL2Helper ? kL2Cut : kL2Steps;
    gemm_pass<4, NX, 2, XMT, kL2Steps, XS0, XS1, ROLE == kL2Reducer, 128 * 18, RCED_D2>(
        b18, (px0 - 2) * 18 + 2 * kq, (pxx - 2) * 18 + 2 * kqx, tailoff, w, lane, acc);
  }
#if RCED_STAMPS
  const unsigned long long f2 = stamp();
#endif
  if constexpr (ROLE == kL2Helper) {
    *reinterpret_cast<f32x4*>(lds + kScratch2Off + XMTP * 256 + 4 * lx) = acc[4][XMTP];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lx == 0) lds_poke(lds + kFlag2Off + XMTP, tag);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
    if (!(RCED_EXP_NOEPI & 2) || acc[t][0].x + acc[t][1].x == 12345.678f)
      store_p1<2, 30>(b30, acc[t], px0 + 128 * t, kq, span_has_gap(16 * (wave + 8 * t), 16));
  if constexpr (ROLE == kL2Reducer) {   // after the own tiles' epilogue: the helper has had time to finish
    for (int spin = 0; spin < (RCED_EXP_NOBAR ? 0 : (1 << 22)); ++spin) {
      if (__builtin_amdgcn_readfirstlane(lds_peek(lds + kFlag2Off + XMTP)) == tag) break;
      __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const f32x4 v = acc[4][XMTP] + *reinterpret_cast<const f32x4*>(lds + kScratch2Off + XMTP * 256 + 4 * lx);
    store_p1_mt<30>(b30, v, pxx, kqx, XMTP);
  }
#if RCED_STAMPS
  const unsigned long long f3 = stamp();
  if (blockIdx.x == 0 && lane == 0) {
    g_fine[wave][0] += f1 - f0;
    g_fine[wave][1] += f2 - f1;
    g_fine[wave][2] += f3 - f2;
  }
#endif
}

// Layer 3 roles: every wave has two regular pair tiles; pair tile 16 is split ALONG K in four, one part per
// SIMD (waves 0..3), so that every SIMD carries the same MFMA count in this layer (the barrier that ends the
// layer waits for the most loaded SIMD): the reducer (wave 0: steps [0,10), owns the epilogue and the skip
// registers) and three helpers (waves 1..3: steps [10,19), [19,28), [28,37) + tail), which hand their partial
// sums over through 1-KiB scratch areas in the (dead during layer 3) B18 buffer and tagged flag words --
// pairwise hand-offs, no extra barrier.
constexpr int kRolePlain = 0, kRoleReducer = 1, kRoleHelper = 2;
constexpr int kL3Cut1 = 10, kL3Cut2 = 19, kL3Cut3 = 28;
constexpr int kScratchOff = kB18Off + kB18Pad * 18 + 8 * 18;   // pixels 8..51 of frame 0: always rewritten by layer 1
constexpr int kFlagOff = kScratchOff + 3 * 256;
static_assert((kScratchOff * 4) % 16 == 0, "scratch is read/written with b128");
static_assert(8 + (3 * 256 + 3 + 17) / 18 <= kF, "scratch + flags stay inside frame 0's real pixels");

template <int ROLE, int HID>   // HID: helper number 1..3 (0 otherwise)
__device__ __forceinline__ void layer3(const Params& P, float* lds, const float* w, int blk, int wave, int lane,
                                       unsigned tag, int utt, int t0, f32x4 (&skip_ce1)[3], f32x4 (&skip_ce2)[3]) {
  constexpr int NX = ROLE == kRolePlain ? 0 : 1, NT = 2 + NX;
  constexpr int NEPI = ROLE == kRoleHelper ? 2 : NT;   // tiles this wave finishes
  OPAQUE_LANE(lane);
  const int n = lane & 15, kq = lane >> 4;
  const float* b30 = lds + kB30Off + kB30Pad * 30;
  float* b8 = lds + kB8Off + kB8Pad * kB8S;
  const int lx = NX > 0 ? opaque(lane) : lane;   // lane copy for the split tile's addresses (see opaque())
  const int q0 = 16 * wave + n, qx = 16 * 16 + (lx & 15), kqx = lx >> 4;   // pixel pair indices
  f32x4 acc[NT][1];
  const f32x4 sh = *reinterpret_cast<const f32x4*>(w + kW3Data + 4 * (kq & 1));
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t][0] = (t == 2 && ROLE == kRoleHelper) ? f32x4{0.f, 0.f, 0.f, 0.f} : sh;
  const int tailoff = -kq;   // K = 300: tail k = 296 + kq, all four real
  if (!(RCED_EXP_SKIP & 4)) {
    constexpr int XS0 = ROLE != kRoleHelper ? 0 : HID == 1 ? kL3Cut1 : HID == 2 ? kL3Cut2 : kL3Cut3;
    constexpr int XS1 = ROLE == kRoleReducer ? kL3Cut1 : ROLE != kRoleHelper ? kL3Steps
                        : HID == 1 ? kL3Cut2 : HID == 2 ? kL3Cut3 : kL3Steps;
    constexpr bool XT = ROLE == kRoleHelper && HID == 3;
    gemm_pass<2, NX, 1, -1, kL3Steps, XS0, XS1, XT, 128 * 60, RCED_D3>(
        b30, (2 * q0 - 4) * 30 + 2 * kq, (2 * qx - 4) * 30 + 2 * kqx, tailoff, w, lane, acc);
  }
  if constexpr (ROLE == kRoleHelper) {   // publish the partial sums of pair tile 16
    *reinterpret_cast<f32x4*>(lds + kScratchOff + (HID - 1) * 256 + 4 * lx) = acc[2][0];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lx == 0) lds_poke(lds + kFlagOff + (HID - 1), tag);
  }
  if constexpr (ROLE == kRoleReducer) {  // collect them (bounded spins: all waves are resident)
#pragma unroll
    for (int h = 0; h < 3; ++h) {
      for (int spin = 0; spin < (RCED_EXP_NOBAR ? 0 : (1 << 22)); ++spin) {
        if (__builtin_amdgcn_readfirstlane(lds_peek(lds + kFlagOff + h)) == tag) break;
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
    for (int h = 0; h < 3; ++h) acc[2][0] += *reinterpret_cast<const f32x4*>(lds + kScratchOff + h * 256 + 4 * lx);
  }
#pragma unroll
  for (int t = 0; t < NEPI; ++t) {
    if ((RCED_EXP_NOEPI & 4) && acc[t][0].x != 12345.678f) continue;
    const int q = (t < 2) ? q0 + 128 * t : qx;
    const int px = 2 * q + (kq >> 1);   // this lane's output pixel (phase = kq >> 1)
    f32x4 v = relu4(acc[t][0]);
    if (blk == 3) v += skip_ce2[t];   // CD1 + CE2 (model.py:87, 75-76: after the ReLU)
    if (blk == 4) v += skip_ce1[t];   // CD2 + CE1 (model.py:88)
    const bool gap = span_has_gap(32 * (t < 2 ? wave + 8 * t : 16), 32);   // wave-uniform
    const int fr = px / kS, f = px - fr * kS;
    const bool ok = gap ? (px < kNPX && f < kF) : true;
    if (gap && !ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
    skip_ce1[t] = (blk == 0) ? v : skip_ce1[t];
    skip_ce2[t] = (blk == 1) ? v : skip_ce2[t];
    if (blk < 4) {
      if (gap && px >= kNPX) continue;   // rows past the tile are not allocated
      float* bp = b8 + px * kB8S + 4 * (kq & 1);   // 8-byte aligned (stride 10): two b64 stores
      *reinterpret_cast<f32x2*>(bp) = f32x2{v.x, v.y};
      *reinterpret_cast<f32x2*>(bp + 2) = f32x2{v.z, v.w};
    } else if (ok && t0 + fr < P.T) {
      float* hp = P.h + (((size_t)utt * P.T + t0 + fr) * kF + f) * kHCh + 4 * (kq & 1);
      *reinterpret_cast<f32x4*>(hp) = v;
    }
  }
}

__global__ __launch_bounds__(kThreads) void fused_v3_kernel(Params P) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* const wbase = lds + kWOff;
#define WREG(i) (wbase + (i) * kWRegion)

  // zero all of LDS once: gap pixels and margins are never written afterwards
  for (int e = tid; e < kLdsFloats; e += kThreads) lds[e] = 0.f;
  __syncthreads();
  // packet of the very first layer into region 0; input rows of the first tile into registers
  packet_dma<kW1>(P.wpack, WREG(0), wave, lane);
  int wcur = 0;
  unsigned epoch = 0;   // layer-3 instances so far (tags the K-split hand-off)
  XStage xst = xstage_load(P, blockIdx.x, tid);
#if RCED_STAMPS
  unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  layer_end_sync();

  // extra tiles of this wave (see the assignment comment above)
  const int xm = wave == 0 ? 32 : 31;                           // layer 1 main, waves 0 and 1
  const int xr0 = wave == 7 ? 3 : wave - 4, xr1 = 4;            // layer 1 remainder, waves 4..7

  if (RCED_EXP_NOBAR >= 2 && wave >= 4)   // experiment: put the second wave of every SIMD half a layer behind the first
    for (int i = 0; i < RCED_EXP_NOBAR; ++i) __builtin_amdgcn_s_sleep(64);   // ~4 k cycles each
  for (int tile = blockIdx.x; tile < P.total_tiles; tile += gridDim.x) {
    const int utt = tile / P.tiles_per_utt;
    const int t0 = (tile - utt * P.tiles_per_utt) * kTF;   // first frame of the tile
    // input rows prefetched during the previous tile -> X0 (B30 is dead: its last reader finished
    // before the barrier that ended the previous tile)
    xstage_store(xst, lds + kX0Off, tid);
    if (RCED_EXP_NOBAR < 2) __syncthreads();

    f32x4 skip_ce1[3], skip_ce2[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) skip_ce1[t] = skip_ce2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* wsrc = P.wpack;

#pragma unroll 1
    for (int blk = 0; blk < 5; ++blk) {
      {  // ---- layer 1: (8x9, 1->18) for block 0, (1x9, 8->18) otherwise
        STAMP_BEGIN();
        if (!RCED_EXP_WGLOBAL) packet_dma<kW2>(wsrc + kW1, WREG(wcur ^ 1), wave, lane);
        const float* w = RCED_EXP_WGLOBAL ? wsrc : WREG(wcur);
        if (wave < 2) layer1<4, 1, 0>(lds, w, blk == 0, wave, lane, xm, 0, 0);
        else if (wave < 4) layer1<4, 0, 0>(lds, w, blk == 0, wave, lane, 0, 0, 0);
        else if (wave < 7) layer1<4, 0, 1>(lds, w, blk == 0, wave, lane, 0, xr0, 0);
        else layer1<3, 0, 2>(lds, w, blk == 0, wave, lane, 0, xr0, xr1);
        wcur ^= 1;
#if RCED_STAMPS
        const unsigned long long st_b_ = stamp();
        tsum[blk == 0 ? 6 : 0] += st_b_ - st_a_;
        layer_end_sync();
        tsum[blk == 0 ? 7 : 3] += stamp() - st_b_;
#else
        layer_end_sync();
#endif
      }
      {  // ---- layer 2: (1x5, 18->30)
        STAMP_BEGIN();
        if (!RCED_EXP_WGLOBAL) packet_dma<kW3>(wsrc + kW1 + kW2, WREG(wcur ^ 1), wave, lane);
        const float* w = RCED_EXP_WGLOBAL ? wsrc + kW1 : WREG(wcur);
        const unsigned tag2 = 0xC0000000u | (epoch + 1u);   // distinct from layer 3's tags (0x8.......)
        if (wave == 0) layer2<kL2Helper, 0>(lds, w, wave, lane, tag2);
        else if (wave == 1) layer2<kL2Helper, 1>(lds, w, wave, lane, tag2);
        else if (wave == 2) layer2<kL2Reducer, 0>(lds, w, wave, lane, tag2);
        else if (wave == 3) layer2<kL2Reducer, 1>(lds, w, wave, lane, tag2);
        else layer2<kL2Plain, 0>(lds, w, wave, lane, tag2);
        wcur ^= 1;
        STAMP_MATH(1);
        layer_end_sync();
        STAMP_WAIT(1);
      }
      {  // ---- layer 3: (1x9, 30->8) on pixel pairs; block skips; hand-off
        STAMP_BEGIN();
        // next packet: layer 1 of the next block, or of block 0 of the next tile (the stream wraps)
        if (!RCED_EXP_WGLOBAL) packet_dma<kW1>(blk == 4 ? P.wpack : wsrc + kWBlock, WREG(wcur ^ 1), wave, lane);
        if (blk == 4) xst = xstage_load(P, tile + gridDim.x, tid);   // next tile's input rows
        const float* w = RCED_EXP_WGLOBAL ? wsrc + kW1 + kW2 : WREG(wcur);
        ++epoch;
        const unsigned tag = 0x80000000u | epoch;   // sign bit set: never the bits of a ReLU output
        if (wave == 0) layer3<kRoleReducer, 0>(P, lds, w, blk, wave, lane, tag, utt, t0, skip_ce1, skip_ce2);
        else if (wave == 1) layer3<kRoleHelper, 1>(P, lds, w, blk, wave, lane, tag, utt, t0, skip_ce1, skip_ce2);
        else if (wave == 2) layer3<kRoleHelper, 2>(P, lds, w, blk, wave, lane, tag, utt, t0, skip_ce1, skip_ce2);
        else if (wave == 3) layer3<kRoleHelper, 3>(P, lds, w, blk, wave, lane, tag, utt, t0, skip_ce1, skip_ce2);
        else layer3<kRolePlain, 0>(P, lds, w, blk, wave, lane, tag, utt, t0, skip_ce1, skip_ce2);
        wcur ^= 1;
        STAMP_MATH(2);
        layer_end_sync();
        STAMP_WAIT(2);
      }
      wsrc += kWBlock;
    }
  }
#undef WREG
#if RCED_STAMPS
  if (P.stamps && blockIdx.x == 0 && lane == 0)
    for (int i = 0; i < 8; ++i) P.stamps[wave * 8 + i] = tsum[i];
  if (P.stamps && blockIdx.x == 0 && lane == 0)
    for (int i = 0; i < 3; ++i) P.stamps[64 + wave * 3 + i] = g_fine[wave][i];
#endif
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

// The same GEMM with the B operand staged through LDS: the 64 frames' h rows arrive in chunks of 64 k as coalesced
// 256-byte row pieces (read once per workgroup instead of once per wave, 16 cache lines per load before), one chunk
// ahead in registers, into a two-buffer ping-pong with frame stride 66 floats (2 mod 32: the 16 frames of a
// ds_read_b64 fall on 16 distinct bank pairs).  A still streams from L2, two steps ahead.
template <int NT, int CHUNK>
struct FinLds {
  static constexpr int kFrames = 16 * NT;                             // frames per workgroup
  static constexpr int kRow = CHUNK + 2;                              // 2 mod 32
  static constexpr int kStepsPer = CHUNK / 8;
  static constexpr int kChunks = (kFinK + CHUNK - 1) / CHUNK;
  static constexpr int kPieces = CHUNK / 4;                           // float4 pieces per frame per chunk
  static constexpr int kVec = kFrames * kPieces;
  static constexpr int kPer = (kVec + kFinThreads - 1) / kFinThreads;
  static_assert(kFinK % 4 == 0 && CHUNK % 8 == 0 && CHUNK % 32 == 0, "float4 pieces never straddle the end of a row");
};

template <int NT, int CHUNK>
__global__ __launch_bounds__(kFinThreads) void final_gemm_lds_kernel(const float* __restrict__ h,
                                                                      const float* __restrict__ apack, float bias,
                                                                      float* __restrict__ y, int frames) {
  using G = FinLds<NT, CHUNK>;
  __shared__ __attribute__((aligned(16))) float bs[2][G::kFrames * G::kRow];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, kq = lane >> 4;
  const int f0 = blockIdx.x * G::kFrames;
  const f32x2* ap = reinterpret_cast<const f32x2*>(apack) + (wave * 3) * 64 + lane;
  auto fetch = [&](int chunk, f32x4(&r)[G::kPer]) {
#pragma unroll
    for (int i = 0; i < G::kPer; ++i) {
      const int q = tid + i * kFinThreads;
      const int fr = f0 + q / G::kPieces, k = chunk * CHUNK + 4 * (q % G::kPieces);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (q < G::kVec && fr < frames && k < kFinK) v = *reinterpret_cast<const f32x4*>(h + (size_t)fr * kFinK + k);
      r[i] = v;
    }
  };
  auto commit = [&](int buf, const f32x4(&r)[G::kPer]) {
#pragma unroll
    for (int i = 0; i < G::kPer; ++i) {
      const int q = tid + i * kFinThreads;
      if (q < G::kVec) {
        float* d = bs[buf] + (q / G::kPieces) * G::kRow + 4 * (q % G::kPieces);
        *reinterpret_cast<f32x2*>(d) = f32x2{r[i].x, r[i].y};
        *reinterpret_cast<f32x2*>(d + 2) = f32x2{r[i].z, r[i].w};
      }
    }
  };
  f32x4 acc[NT][3];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int m = 0; m < 3; ++m) acc[t][m] = f32x4{bias, bias, bias, bias};
  f32x4 r[G::kPer];
  fetch(0, r);
  commit(0, r);
  f32x2 a[3], an[3], an2[3];            // A fragments of steps S, S+1, S+2
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    a[m] = ap[m * 64];
    an[m] = ap[(kFinMT + m) * 64];
  }
  __syncthreads();
  for (int c = 0; c < G::kChunks; ++c) {
    if (c + 1 < G::kChunks) fetch(c + 1, r);
    const float* bb = bs[c & 1] + n * G::kRow + 2 * kq;
    const int left = kFinSteps - G::kStepsPer * c;
    const int ns = left < G::kStepsPer ? left : G::kStepsPer;
#pragma unroll
    for (int s = 0; s < G::kStepsPer; ++s) {
      if (s < ns) {
        const int S = G::kStepsPer * c + s;
        if (S + 2 < kFinSteps) {
#pragma unroll
          for (int m = 0; m < 3; ++m) an2[m] = ap[((S + 2) * kFinMT + m) * 64];
        }
        f32x2 b[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) b[t] = *reinterpret_cast<const f32x2*>(bb + 16 * t * G::kRow + 8 * s);
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int m = 0; m < 3; ++m) acc[t][m] = mfma(a[m][e], b[t][e], acc[t][m]);
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          a[m] = an[m];
          an[m] = an2[m];
        }
      }
    }
    if (c + 1 < G::kChunks) commit((c + 1) & 1, r);
    __syncthreads();
  }
  // D row = f = 16*(3*wave+m) + 4*kq + j, column = frame
#pragma unroll
  for (int t = 0; t < NT; ++t) {
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
