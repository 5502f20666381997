// Fused R-CED (V1 / V2) forward with bf16 activations and weights on the bf16 MFMA (BASELINE config 2:
// "R-CED V2 forward, batch 64, 129x512, bf16").  Same construction as kernels_fused_chain.h (read that first):
// a tile of frames lives in LDS as [pixel][channel], a 1xk conv is an implicit GEMM whose B operand is a
// ds_read_b64 out of that buffer, cout sits on the 16-row M axis, packets arrive by LDS-DMA one layer ahead.
// What changes:
//   * activations are bf16 in LDS, channel stride = cout rounded up to 4 (one ds_read_b64 = 4 consecutive k);
//     v_mfma_f32_16x16x16_bf16 consumes K = 16 per instruction at 16 cycles (vs 4 x 32 for the fp32 MFMA);
//   * every layer's output is rounded to bf16 (round to nearest even) after bias/BatchNorm shift (fp32), skip add and
//     ReLU; the skip fragments kept in the global scratch and the hand-off tensor hold those rounded values;
//   * the first layer (8 x k on the fp32 input) and the final 1x129 layer stay on the fp32 MFMA with fp32 weights
//     (their inputs / outputs are the network's fp32 boundary); their activations on the inside are bf16 values.
// Precision contract: DESIGN.md 3.3b and tests/test_forward_gpu.py (the GPU result is tested against an emulation that
// rounds at the same places) -- this path is NOT within the 1e-4 fp32 bar, it exists because config 2 names bf16.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_fused_chain.h"

#ifndef RCED_C16_TF
#define RCED_C16_TF 3        // frames per tile of the bf16 kernel (3: two workgroups per CU; 6: one, experiment)
#endif
#if RCED_C16_TF == 3
#define RCED_C16_LDS_KB 80
#define RCED_C16_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))
#else
#define RCED_C16_LDS_KB 160
#define RCED_C16_ATTR
#endif
#ifndef RCED_C16_K32
#define RCED_C16_K32 0     // 1: inner layers on v_mfma_f32_16x16x32_bf16 (K = 32 per instruction in 16 cycles: the full bf16
                           // rate, half the matrix-pipe time); 0: v_mfma_f32_16x16x16_bf16 (K = 16 in the same 16 cycles).
                           // Measured, round 3 (A/B on one box, bench.py --dtype bf16): R-CED V2 batch 64 (BASELINE config 2)
                           // fused kernel 0.890 vs 0.897 ms, batch 256 3.60 vs 3.67 ms; R-CED V1 batch 64 0.805 vs 0.700 ms
                           // (K rounds up to 32 per layer).  Halving the MFMA cycles buys < 1 %: this kernel is not bound by
                           // the matrix pipe -- so the default stays the K = 16 form
#endif
#ifndef RCED_C16_EMU_X6
#define RCED_C16_EMU_X6 0  // timing experiment only (wrong results): the COST of a three-part (six-product) form of this kernel -- three reads per
                           // operand fragment, six MFMAs per product, a three-part split and three stores per output fragment
#endif
#if RCED_C16_EMU_X6 && !defined(RCED_TIMING_ONLY)
#error "RCED_C16_EMU_X6 computes wrong results: timing experiments only (-DRCED_TIMING_ONLY)"
#endif
#ifndef RCED_C16_DEPTH
#define RCED_C16_DEPTH 1   // operand prefetch depth of the bf16 pass (steps)
#endif

namespace rced {
namespace chain16 {

using chain::f32x2;
using chain::f32x4;
using chain::kF;
using chain::kThreads;
using chain::kWaves;
using chain::LayerDesc;
using chain::Params;
using chain::pin;
using chain::u32x4;

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int round4(int c) { return (c + 3) & ~3; }

// The same net with more frames per tile (bf16 activations need half the LDS): geometry only.
using chain::WithTF;

template <class N>
struct Geo {
  using G32 = chain::Geo<N>;
  static constexpr int kS = G32::kS, kNPX = G32::kNPX, kTiles = G32::kTiles, kRegular = G32::kRegular, kExtra = G32::kExtra;
  static constexpr int kPad = G32::kPad, kRows = G32::kRows;
  // channel stride (bf16 elements) of layer l's output: cout rounded up to 4.  A stride of 16 (= 8 dwords) puts a
  // tile's 16 pixels on 8 bank groups (2-way conflicts on every B read): those get 4 more, zero-weight, k per tap.
  // (32 would deserve the same, but 36 does not leave room for two workgroups per CU.)
  static constexpr int cp(int l) { return round4(N::layer[l].cout) == 16 ? 20 : round4(N::layer[l].cout); }
  static constexpr int chmax(int parity) {
    int m = 0;
    for (int l = parity; l < N::kLayers; l += 2) m = cp(l) > m ? cp(l) : m;
    return m;
  }
  static constexpr int kChX = chmax(0), kChY = chmax(1);                      // X holds outputs of even layers
  // LDS map in floats (4-byte units); bf16 buffers take rows * ch / 2 floats
  static constexpr int kSlack = RCED_C16_K32 ? 16 : 8;                        // floats: reads of the zero-weight K padding
  static constexpr int kXFloats = ((kRows * kChX / 2 + kSlack + 3) / 4) * 4;
  static constexpr int kYFloats = ((kRows * kChY / 2 + kSlack + 3) / 4) * 4;
  static constexpr int kXOff = 0, kYOff = kXOff + kXFloats, kWOff = kYOff + kYFloats;
  // packets: layer 0 as in the fp32 kernel (b32 steps); layers >= 1: [step][mt][lane] x 4 bf16, then 32 shifts
  static constexpr int K(int l) { return N::layer[l].taps * cp(l - 1); }
  static constexpr int kKStep = RCED_C16_K32 ? 32 : 16;                       // K per MFMA
  static constexpr int steps(int l) { return (K(l) + kKStep - 1) / kKStep; }
  static constexpr int MT(int l) { return (N::layer[l].cout + 15) / 16; }
  static constexpr int data(int l) { return l == 0 ? G32::data(0) : steps(l) * MT(l) * 64 * (kKStep / 8); }   // 2 bytes per k and lane
  static constexpr int packet(int l) { return data(l) + 32; }
  static constexpr int packet_off(int l) {
    int o = 0;
    for (int i = 0; i < l; ++i) o += packet(i);
    return o;
  }
  static constexpr int kWTotal = packet_off(N::kLayers);
  static constexpr int maxpacket() {
    int m = 0;
    for (int l = 0; l < N::kLayers; ++l) m = packet(l) > m ? packet(l) : m;
    return m;
  }
  static constexpr int kWRegion = ((maxpacket() + 3) / 4) * 4;
  static constexpr int kLdsFloats = kWOff + 2 * kWRegion;
  static constexpr int kLdsBytes = kLdsFloats * 4;
  static_assert(kLdsBytes <= RCED_C16_LDS_KB * 1024, "LDS budget (80 KB = two workgroups per CU)");
  // fp32 input rows of the first layer alias buffer Y (dead until layer 1 writes it)
  static constexpr int kX0Off = kYOff + (kPad * kChY) / 2;
  static_assert((kPad * kChY) % 2 == 0 && G32::kX0Floats <= kYFloats - (kPad * kChY) / 2, "X0 fits in buffer Y");
  // skip scratch: units = (layer that saves, slot, mt), each kThreads x float4 (this kernel's own M-tile counts: the
  // fp32 kernel runs some layers as one M-tile + a remainder pass and numbers its units differently)
  static constexpr int skip_unit(int l) {
    int u = 0;
    for (int i = 0; i < l; ++i)
      if (N::layer[i].saves_skip) u += (kRegular + 1) * MT(i);
    return u;
  }
  static constexpr size_t kScratchFloatsPerWg = (size_t)skip_unit(N::kLayers) * kThreads * 4;
};

__device__ __forceinline__ f32x4 mfma16(s16x4 a, s16x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}
// four floats -> four bf16 (round to nearest even, hardware conversion), as the 8-byte value stored in LDS
__device__ __forceinline__ s16x4 to_bf16x4(f32x4 v) {
  const bf16x4 h = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
  return __builtin_bit_cast(s16x4, h);
}
__device__ __forceinline__ f32x4 from_bf16x4(s16x4 s) {
  const bf16x4 h = __builtin_bit_cast(bf16x4, s);
  return f32x4{(float)h.x, (float)h.y, (float)h.z, (float)h.w};
}

// K = 32 per instruction: lane kq supplies k = 32 s + 8 kq .. + 7 -- eight consecutive bf16 of the pixel's im2col window,
// which starts 8-byte aligned (channel strides are multiples of 4), so the B operand is two 8-byte halves (hipcc fuses them
// into one ds_read_b128 at an 8-byte-aligned address; the LDS takes that) and the A fragment one 16-byte read of the packet.
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 mfma32(s16x8 a, s16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <int NR, int NX, int MT, int STEPS, int STRIDE, int DEPTH, class Pre>
__device__ __forceinline__ void pass32(const __bf16* act, int off0, int offx, const float* w, int lane,
                                       f32x4 (&acc)[NR + NX][MT], Pre pre) {
  constexpr int NT = NR + NX, RING = DEPTH + 1;
  const s16x8* wp = reinterpret_cast<const s16x8*>(w) + lane;
  s16x8 a[RING][MT], b[RING][NT];
  int offs[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    offs[t] = t < NR ? off0 + t * STRIDE : offx;
    asm volatile("" : "+v"(offs[t]));
  }
#if RCED_C16_EMU_X6
  s16x8 a1[RING][MT], a2[RING][MT], b1[RING][NT], b2[RING][NT];
#endif
  auto load = [&](int s, int buf) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a[buf][mt] = wp[(s * MT + mt) * 64];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const s16x4 lo = *reinterpret_cast<const s16x4*>(act + offs[t] + 32 * s);
      const s16x4 hi = *reinterpret_cast<const s16x4*>(act + offs[t] + 32 * s + 4);
      b[buf][t] = s16x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    }
#if RCED_C16_EMU_X6
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      a1[buf][mt] = wp[(s * MT + mt) * 64 + 1];     // (another lane's fragment: a different address, the same cost)
      a2[buf][mt] = wp[(s * MT + mt) * 64 + 2];
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const s16x4 lo1 = *reinterpret_cast<const s16x4*>(act + offs[t] + 32 * s + 64), hi1 = *reinterpret_cast<const s16x4*>(act + offs[t] + 32 * s + 68);
      const s16x4 lo2 = *reinterpret_cast<const s16x4*>(act + offs[t] + 32 * s + 128), hi2 = *reinterpret_cast<const s16x4*>(act + offs[t] + 32 * s + 132);
      b1[buf][t] = s16x8{lo1.x, lo1.y, lo1.z, lo1.w, hi1.x, hi1.y, hi1.z, hi1.w};
      b2[buf][t] = s16x8{lo2.x, lo2.y, lo2.z, lo2.w, hi2.x, hi2.y, hi2.z, hi2.w};
    }
#endif
  };
#pragma unroll
  for (int s = 0; s < DEPTH && s < STEPS; ++s) load(s, s % RING);
  pin();
  pre();
  pin();
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    if (s + DEPTH < STEPS) load(s + DEPTH, (s + DEPTH) % RING);
    pin();
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int r = s % RING;
        acc[t][mt] = mfma32(a[r][mt], b[r][t], acc[t][mt]);
#if RCED_C16_EMU_X6
        acc[t][mt] = mfma32(a1[r][mt], b1[r][t], acc[t][mt]);
        acc[t][mt] = mfma32(a2[r][mt], b[r][t], acc[t][mt]);
        acc[t][mt] = mfma32(a[r][mt], b2[r][t], acc[t][mt]);
        acc[t][mt] = mfma32(a1[r][mt], b[r][t], acc[t][mt]);
        acc[t][mt] = mfma32(a[r][mt], b1[r][t], acc[t][mt]);
#endif
      }
    pin();
  }
}

// Implicit-GEMM pass on bf16: STEPS steps of K = 16 (lane kq supplies k = 16 s + 4 kq .. +3), NT = NR + NX slots.
// `pre` runs once the first operand reads are in flight (the next packet's LDS-DMA: see chain::gemm_pass).
template <int NR, int NX, int MT, int STEPS, int STRIDE, int DEPTH, class Pre>
__device__ __forceinline__ void pass16(const __bf16* act, int off0, int offx, const float* w, int lane,
                                       f32x4 (&acc)[NR + NX][MT], Pre pre) {
  constexpr int NT = NR + NX, RING = DEPTH + 1;
  const s16x4* wp = reinterpret_cast<const s16x4*>(w) + lane;
  s16x4 a[RING][MT], b[RING][NT];
  int offs[NT];   // one hidden base register per tile: no ds_read2 fusing, no per-step re-basing adds (see chain::gemm_pass)
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    offs[t] = t < NR ? off0 + t * STRIDE : offx;
    asm volatile("" : "+v"(offs[t]));
  }
  auto load = [&](int s, int buf) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a[buf][mt] = wp[(s * MT + mt) * 64];
#pragma unroll
    for (int t = 0; t < NT; ++t) b[buf][t] = *reinterpret_cast<const s16x4*>(act + offs[t] + 16 * s);
  };
#pragma unroll
  for (int s = 0; s < DEPTH && s < STEPS; ++s) load(s, s % RING);
  pin();
  pre();
  pin();
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    if (s + DEPTH < STEPS) load(s + DEPTH, (s + DEPTH) % RING);
    pin();
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[t][mt] = mfma16(a[s % RING][mt], b[s % RING][t], acc[t][mt]);
    pin();
  }
}

template <class N, int L, int NX, class Dma>
__device__ __forceinline__ void run_layer(const Params& P, float* lds, const float* w, __amdgpu_buffer_rsrc_t scratch,
                                          int wave, int lane, int tid, int utt, int t0, Dma dma) {
  using G = Geo<N>;
  constexpr LayerDesc D = N::layer[L];
  constexpr int NR = G::kRegular, NT = NR + NX, MT = G::MT(L);
  constexpr bool kLast = (L == N::kLayers - 1);
  constexpr int cpo = G::cp(L);
  asm volatile("" : "+v"(lane), "+v"(tid));   // see chain::run_layer: no hoisting of every layer's addresses
  const int n = lane & 15, kq = lane >> 4;
  __bf16* bufx = reinterpret_cast<__bf16*>(lds + G::kXOff) + G::kPad * G::kChX;
  __bf16* bufy = reinterpret_cast<__bf16*>(lds + G::kYOff) + G::kPad * G::kChY;
  const __bf16* in = (L % 2 == 1) ? bufx : bufy;
  __bf16* out = (L % 2 == 0) ? bufx : bufy;
  const int xtile = G::kRegular * kWaves + wave;
  const int px0 = 16 * wave + n, pxx = 16 * xtile + n;

  f32x4 acc[NT][MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const f32x4 sh = *reinterpret_cast<const f32x4*>(w + G::data(L) + 16 * mt + 4 * kq);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t][mt] = sh;
  }
  if constexpr (L == 0) {
    chain::first_pass<NR, NX, D.taps, G::kS, 2>(lds + G::kX0Off, px0 + kq * G::kS, pxx + kq * G::kS, w, lane, acc, dma);
  } else {
    constexpr int padl = (D.taps - 1) / 2, cpi = G::cp(L - 1);
    if constexpr (RCED_C16_K32)
      pass32<NR, NX, MT, G::steps(L), 128 * cpi, RCED_C16_DEPTH>(in, (px0 - padl) * cpi + 8 * kq, (pxx - padl) * cpi + 8 * kq, w, lane, acc, dma);
    else
      pass16<NR, NX, MT, G::steps(L), 128 * cpi, RCED_C16_DEPTH>(in, (px0 - padl) * cpi + 4 * kq, (pxx - padl) * cpi + 4 * kq, w, lane, acc, dma);
  }
  // skip fragments of the matching encoder layer (own stores of an earlier layer; L2-resident).  Loaded here, not
  // before the pass: 32 fewer live VGPRs during the pass keep the kernel at 128 and two workgroups on a CU, whose
  // MFMAs hide this latency.
  f32x4 skip[D.skip_from >= 0 ? NT : 1][D.skip_from >= 0 ? MT : 1];
  if constexpr (D.skip_from >= 0) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        skip[t][mt] = __builtin_bit_cast(
            f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                       scratch, tid * 16, (G::skip_unit(D.skip_from) + t * MT + mt) * kThreads * 16, 0));
  }
  // Layers that store to global memory in the epilogue (skip fragments, the hand-off tensor) wait HERE for the
  // next packet's LDS-DMA (issued a whole pass ago) and end on a bare barrier, so the stores stay in flight across
  // it; the other layers wait at their end (chain::layer_end_sync).  See chain::run_layer.
  if constexpr (D.saves_skip || kLast) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int tile = t < NR ? wave + kWaves * t : xtile;
    const int px = t < NR ? px0 + 128 * t : pxx;
    const bool gap = chain::span_has_gap<N>(16 * tile, 16);   // wave-uniform: 3 of the 26 tiles
    bool ok = true;
    if (gap) {
      asm volatile("" ::: "memory");   // a wave-uniform BRANCH, not selects (see chain::run_layer)
      ok = chain::px_valid<N>(px);
    }
    const int fr = px / G::kS, f = px - fr * G::kS;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      f32x4 v = acc[t][mt];
      if constexpr (D.skip_from >= 0) v += skip[t][mt];   // module.py:30-31: before the ReLU
      v = chain::relu4(v);
      if (gap) {
        asm volatile("" ::: "memory");
        if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      const s16x4 h = to_bf16x4(v);                       // the layer's output IS this rounded value
#if RCED_C16_EMU_X6
      s16x4 hm, hl;
      {
        const f32x4 r1 = v - from_bf16x4(h);
        hm = to_bf16x4(r1);
        const f32x4 r2 = r1 - from_bf16x4(hm);
        hl = to_bf16x4(r2);
      }
#endif
      if constexpr (D.saves_skip || kLast) v = from_bf16x4(h);
      if constexpr (D.saves_skip)
      {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), scratch, tid * 16,
                                               (G::skip_unit(L) + t * MT + mt) * kThreads * 16, 0);
        store_wait_state();
      }
      const int co0 = 16 * mt + 4 * kq;
      if constexpr (!kLast) {
        if (co0 < cpo) *reinterpret_cast<s16x4*>(out + px * cpo + co0) = h;
#if RCED_C16_EMU_X6
        if (co0 < cpo && hm.x == 0x7fff) *reinterpret_cast<s16x4*>(out + px * cpo + co0) = hm;   // (never true for real data; keeps the split alive)
        if (co0 < cpo && hl.x == 0x7fff) *reinterpret_cast<s16x4*>(out + px * cpo + co0) = hl;
#endif
        // padding channels past the last M-tile (stride 20 for 16 channels): keep them zero -- stale bits of another
        // layer's layout could read as bf16 NaN, and NaN x 0 weight is not 0
        if constexpr (cpo > 16 * MT) {
          static_assert(cpo - 16 * MT == 4, "one 8-byte store clears the padding");
          if (mt == MT - 1 && kq == 0) *reinterpret_cast<s16x4*>(out + px * cpo + 16 * MT) = s16x4{0, 0, 0, 0};
        }
      } else if (ok && px < G::kNPX && f < kF && t0 + fr < P.T) {
        float* hp = P.h + (((size_t)utt * P.T + t0 + fr) * kF + f) * N::kFinalCh + co0;
        if (co0 + 1 < N::kFinalCh) *reinterpret_cast<f32x2*>(hp) = f32x2{v.x, v.y};
        if (co0 + 3 < N::kFinalCh) *reinterpret_cast<f32x2*>(hp + 2) = f32x2{v.z, v.w};
      }
    }
  }
}

template <int NFLOATS>
__device__ __forceinline__ void packet_dma(const float* __restrict__ src, float* dst, int wave, int lane) {
  chain::packet_dma<NFLOATS, true>(src, dst, wave, lane);
}

template <class N, int L>
__device__ __forceinline__ void run_layers(const Params& P, float* lds, __amdgpu_buffer_rsrc_t scratch, int& wcur,
                                           chain::XStage& xst, int tile, int wave, int lane, int tid, int utt, int t0) {
  using G = Geo<N>;
  if constexpr (L < N::kLayers) {
    float* const wbase = lds + G::kWOff;
    constexpr int nxt = (L + 1 < N::kLayers) ? L + 1 : 0;
    float* const wdst = wbase + (wcur ^ 1) * G::kWRegion;
    auto dma = [&] { packet_dma<G::packet(nxt)>(P.wpack + G::packet_off(nxt), wdst, wave, lane); };   // issued inside the pass
    if constexpr (L == N::kLayers - 1) xst = chain::xstage_load<N>(P, tile + gridDim.x, tid);
    const float* w = wbase + wcur * G::kWRegion;
    if (wave < G::kExtra) chain16::run_layer<N, L, 1>(P, lds, w, scratch, wave, lane, tid, utt, t0, dma);
    else chain16::run_layer<N, L, 0>(P, lds, w, scratch, wave, lane, tid, utt, t0, dma);
    wcur ^= 1;
    if constexpr (N::layer[L].saves_skip || L == N::kLayers - 1) __syncthreads();
    else chain::layer_end_sync();
    chain16::run_layers<N, L + 1>(P, lds, scratch, wcur, xst, tile, wave, lane, tid, utt, t0);
  }
}

template <class N>
__global__ __launch_bounds__(kThreads) RCED_C16_ATTR void fused_chain16_kernel(Params P) {
  using G = Geo<N>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int e = tid; e < G::kLdsFloats; e += kThreads) lds[e] = 0.f;
  __syncthreads();
  packet_dma<G::packet(0)>(P.wpack, lds + G::kWOff, wave, lane);
  int wcur = 0;
  chain::XStage xst = chain::xstage_load<N>(P, blockIdx.x, tid);
  const __amdgpu_buffer_rsrc_t scratch = __builtin_amdgcn_make_buffer_rsrc(
      P.scratch + (size_t)blockIdx.x * G::kScratchFloatsPerWg, 0, (int)(G::kScratchFloatsPerWg * 4), 0x00020000);
  chain::layer_end_sync();
  for (int tile = blockIdx.x; tile < P.total_tiles; tile += gridDim.x) {
    const int utt = tile / P.tiles_per_utt;
    const int t0 = (tile - utt * P.tiles_per_utt) * N::kTF;
    chain::xstage_store<N>(xst, lds + G::kX0Off, tid);
    __syncthreads();
    chain16::run_layers<N, 0>(P, lds, scratch, wcur, xst, tile, wave, lane, tid, utt, t0);
  }
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
