// R-CED V1 / V2 forward in bf16 (BASELINE config 2: "R-CED V2 forward, batch 64, 129x512, bf16"): every layer but the
// 1x129 output layer in ONE kernel on v_mfma_f32_16x16x32_bf16.  Round 6's rebuild of kernels_fused_chain16.h, whose
// 3-frame tiles crossed fifteen workgroup barriers with 4-17 half-rate MFMAs per tile and layer between them (0.107 of
// the bf16 peak for four rounds).  Reference: model_utils/model.py:6-61 over module.py:11-34 (conv -> BN -> +skip -> ReLU).
//
// A WAVE OWNS A FRAME.  Only the first conv (8 x k) looks along time; every later layer is 1 x k along frequency, so a
// frame runs through all the layers without ever reading another frame's activations.  A workgroup is four waves (one
// per SIMD), two workgroups share a CU; a work item is four consecutive frames of one utterance, wave w takes frame
// t0 + w.  No activation ever crosses a wave: the only thing the waves of a workgroup share is the weight stream, and the
// one barrier per layer exists for that alone (it meets four waves that have done exactly the same work).
//
//   * Pixel space of a frame: bin f at row f + 8 of a 160-row image; rows 0..7 and 137..159 stay zero for the kernel's
//     lifetime (the SAME padding of every layer).  Nine 16-pixel tiles (the ninth holds bin 128 alone).
//   * Activations are bf16 PLANES  [octet of channels][row][8 channels]  = 16-byte rows, 2,560 bytes per plane (a
//     multiple of 256: the lane groups of a ds_read_b128 -- {n 0-3, 12-15 of k-quad kq, n 4-11 of kq + 1} -- land on
//     sixteen different 16-byte bank slots).  A layer works IN PLACE: all nine tiles' accumulators are in registers
//     (<= 72) before the first output row is stored, so one 10-KB image per frame is all the LDS a frame needs.
//   * A conv is an implicit GEMM, cout on the M axis, pixels on N, K = (tap, octet) slots of 8 channels: lane (kq, n) of
//     K-step s reads slot j = 4 s + kq = (tap j / OCT, octet j % OCT) of pixel n's window -- ONE aligned ds_read_b128
//     out of the image, no im2col copy.  (The old kernel's [pixel][channel] rows gave 8-byte-aligned 16-byte reads,
//     which the LDS replays: its K = 32 switch bought nothing.)
//   * The first layer (8 x k on the 1-channel input) is the same code: the wave lays its eight input rows out as ONE
//     plane [row f + 8][8 time rows] (an im2col along time only), the input cast to bf16 ("C2 ... (cast
//     bf16)"), weights packed with the time row in the channel slot.
//   * Epilogue per fragment: two v_cvt_pk_bf16_f32, two v_pk_max_i16 (ReLU on the rounded value: the same result as
//     rounding the ReLU), one ds_write_b64.
//   * Skips (module.py:30-31: decoder layer += encoder output BEFORE the ReLU; 72 / 114 channels): the encoder layer's
//     packed bf16 fragment -- lane (kq, n): channels 4 kq .. + 3 of pixel n -- is exactly the B operand of a
//     v_mfma_f32_16x16x16_bf16 whose k is the channel, so the decoder adds it with ONE MFMA against an identity
//     A fragment: no unpacking, no VALU.  The fragments wait in a per-wave global scratch (8 bytes per lane, 512-byte
//     wave stores, L2 / MALL resident), loaded at the start of the decoder layer and used after its last K-step.
//   * Weights: per layer a packet of 1-KiB A fragments [step][M-tile][lane] x 8 bf16 + 32 fp32 shifts, LDS-DMA'd one
//     layer ahead into a two-packet ring.
// Precision contract: tests/test_forward_gpu.py against oracle/rced_np.forward_bf16, which rounds at the same places
// (input, every folded kernel, every layer's output).  NOT within the fp32 path's 1e-4 bar: opt-in (option "bf16").
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_fused_chain.h"
#include "lds_dma.h"

#ifndef RCED_F16_EXP
#define RCED_F16_EXP 0   // timing experiments only (wrong results): 1 = no skip stores, 2 = no skip loads / adds, 4 = no barriers,
                         // 8 = A fragments read for the first K-step only, 16 = no input-row loads, 64 = no packet DMA
#endif
#if RCED_F16_EXP != 0 && !defined(RCED_TIMING_ONLY)
#error "RCED_F16_EXP builds compute wrong results: timing experiments only (-DRCED_TIMING_ONLY)"
#endif

#ifndef RCED_F16_STAMPS
#define RCED_F16_STAMPS 0   // diagnostic build: s_memtime stamps of workgroup 0 / wave 0 on its second tile (tools/stamps16.py)
#endif

namespace rced {
namespace frame16 {

using chain::f32x2;
using chain::f32x4;
using chain::kF;
using chain::LayerDesc;
using chain::pin;

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kWaves = 4;                 // one per SIMD; two workgroups per CU
constexpr int kThreads = kWaves * 64;
constexpr int kRowPad = 8;                // bin f lives at row f + kRowPad
constexpr int kPlanes = 4;                // 32 channels at most (V1's 24 -> 32 layer, V2's 23 -> 25)
constexpr int kTiles = 9;                 // 16-pixel tiles per frame

struct Params {
  const float* x;            // [N, T, 129]
  float* h;                  // [N*T, 129, kFinalCh] fp32: input of the output layer's kernel (values are bf16-exact)
  const unsigned* wpack;     // Geo::kWBytes
  unsigned* scratch;         // gridDim.x * kWaves * Geo::kScratchBytesPerWave (skip fragments)
  int N, T;
  int tiles_per_utt;         // ceil(T / 4)
  int total_tiles;
  unsigned long long* stamps;   // RCED_F16_STAMPS builds only
};
#if RCED_F16_STAMPS
#define F16_STAMP(on, i) do { if ((on) && (lane & 63) == 0) P.stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define F16_STAMP(on, i) do { } while (0)
#endif

template <class N>
struct Geo {
  static constexpr int kLayers = N::kLayers;
  static constexpr int oct_out(int l) { return (N::layer[l].cout + 7) / 8; }
  static constexpr int oct_in(int l) { return l == 0 ? 1 : oct_out(l - 1); }
  static constexpr int MT(int l) { return (N::layer[l].cout + 15) / 16; }
  static constexpr int slots(int l) { return N::layer[l].taps * oct_in(l); }   // K slots of 8 (layer 0: 8 time rows per tap)
  static constexpr int steps(int l) { return (slots(l) + 3) / 4; }
  static constexpr int frags(int l) { return steps(l) * MT(l); }
  static constexpr int packet_bytes(int l) { return frags(l) * 1024; }
  static constexpr int packet_off(int l) {
    int o = 0;
    for (int i = 0; i < l; ++i) o += packet_bytes(i);
    return o;
  }
  static constexpr int kShiftOff = packet_off(kLayers);         // 32 fp32 shifts per layer behind the packets
  static constexpr int kShiftBytes = kLayers * 128;
  static constexpr int kWBytes = kShiftOff + kShiftBytes;
  static constexpr int maxpacket() {
    int m = 0;
    for (int l = 0; l < kLayers; ++l) m = packet_bytes(l) > m ? packet_bytes(l) : m;
    return m;
  }
  static constexpr int kWRegion = maxpacket();
  static constexpr int pad(int l) { return (N::layer[l].taps - 1) / 2; }
  // rows per plane: 8 zero rows, 129 bins, then zero rows up to the last row any window reaches -- tile 8's pixel 143 at the
  // last slot of a layer's last K-step (zero-weight pad slots included: they read up to three taps past the window) --
  // rounded up to 16 rows, so that a plane is a multiple of 256 bytes: 160 rows (V2), 176 (V1: thirteen taps over one octet)
  static constexpr int last_row() {
    int m = 0;
    for (int l = 0; l < kLayers; ++l) {
      const int r = 16 * (kTiles - 1) + 15 + (4 * steps(l) - 1) / oct_in(l) - pad(l) + kRowPad;
      m = r > m ? r : m;
    }
    return m;
  }
  static constexpr int kRows = (last_row() + 1 + 15) / 16 * 16;
  static constexpr int kPlane = kRows * 16;         // bytes
  static constexpr int kRegion = kPlanes * kPlane;  // one frame's image
  static constexpr int kActBytes = kWaves * kRegion;
  static constexpr int kWOff = kActBytes;
  static constexpr int kSOff = kWOff + 2 * kWRegion;            // every layer's shifts, resident (loaded once per workgroup)
  static constexpr int kLdsBytes = kSOff + kShiftBytes;
  static_assert(kLdsBytes <= 80 * 1024, "two workgroups per CU");
  static constexpr bool pads_ok() {
    for (int l = 0; l < kLayers; ++l)
      if (pad(l) > kRowPad) return false;
    return true;
  }
  static_assert(pads_ok(), "the widest kernel's left halo fits the leading zero rows");
  // skip scratch, per wave: per (saving layer, M-tile) four 1-KiB units -- the fragments of tiles (0,1) .. (6,7), 16 bytes per
  // lane -- and 512 bytes for tile 8.  Only the lanes whose four channels exist are stored and loaded (k-quads 0 .. quads - 1:
  // whole 256-byte runs), so an M-tile with 3 real channels moves a quarter of its unit.
  static constexpr int kSkipSet = 4 * 1024 + 512;
  static constexpr int skip_off(int l, int mt) {
    int u = 0;
    for (int i = 0; i < l; ++i)
      if (N::layer[i].saves_skip) u += MT(i);
    return (u + mt) * kSkipSet;
  }
  static constexpr int skip_quads(int l, int mt) {   // k-quads of M-tile mt that hold real channels
    const int q = (N::layer[l].cout + 3) / 4 - 4 * mt;
    return q > 4 ? 4 : q;
  }
  static constexpr size_t kScratchBytesPerWave = (size_t)skip_off(kLayers, 0);
};

__device__ __forceinline__ f32x4 mfma32(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(u32x2 a, u32x2 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
}
// two floats -> packed bf16 (round to nearest even; v_cvt_pk_bf16_f32 keeps a NaN a NaN)
// (as a vector conversion: element-wise casts followed by an integer use of the pair came out as two conversions + v_perm_b32)
__device__ __forceinline__ unsigned pack2(float a, float b) {
  const bf16x2 h = __builtin_convertvector(f32x2{a, b}, bf16x2);
  return __builtin_bit_cast(unsigned, h);
}
// ReLU on two packed bf16: a signed 16-bit max with zero (negative values, -0 and NaNs with the sign bit set become +0)
__device__ __forceinline__ unsigned relu2(unsigned v) {
  const s16x2 z = {0, 0};
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), z));
}

// LDS-DMA of one packet: whole 1-KiB pieces dealt round-robin to the four waves.  The piece's source is a wave-uniform
// base + lane * 16, its LDS address goes through M0 (readfirstlane: hipcc is free to compute a uniform address on the VALU,
// and an "s" operand of an asm statement does not make it move the value).
template <int BYTES>
__device__ __forceinline__ void packet_dma(const unsigned* __restrict__ src, char* dst, int wave, int lane) {
  static_assert(BYTES % 1024 == 0, "whole pieces");
  constexpr int chunks = BYTES / 1024;
  const unsigned d0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)dst);
#pragma unroll
  for (int i = 0; i < (chunks + kWaves - 1) / kWaves; ++i) {
    const int c = wave + i * kWaves;
    if (c < chunks) {
      const unsigned m0v = d0 + c * 1024;
      const unsigned* sp = src + c * 256;
      unsigned saved;   // m0 is saved and restored inside the statement, so the compiler's view of it stays valid
      asm volatile(
          "s_mov_b32 %0, m0\n\t"
          "s_mov_b32 m0, %3\n\t"
          "s_nop 3\n\t"
          "global_load_lds_dwordx4 %1, %2\n\t"
          "s_mov_b32 m0, %0"
          : "=&s"(saved)
          : "v"((unsigned)lane * 16u), "s"(sp), "s"(m0v)
          : "memory");
    }
  }
}

// The eight input rows t - 3 .. t + 4 of one frame, three 64-bin columns per lane, fetched one tile ahead
struct XRows {
  float v[3][8];
};
// Buffer loads over the utterance's [T, 129] floats: a row in front of the first or behind the last frame (TF 'SAME' for the
// 8-tall kernel: 3 rows before, 4 after) is out of the descriptor's range and reads as zero -- no per-row predicates, one
// scalar offset per row.  Lanes past bin 128 of the third column read the next row's bins; x_store drops them.
__device__ __forceinline__ XRows x_load(const Params& P, int tile, int wave, int lane) {
  XRows r;
  const bool live = tile < P.total_tiles;
  const int utt = live ? tile / P.tiles_per_utt : 0;
  const int t = live ? (tile - utt * P.tiles_per_utt) * kWaves + wave : 0;
  const __amdgpu_buffer_rsrc_t xu = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(P.x) + (size_t)utt * P.T * kF, 0, live ? P.T * kF * 4 : 0, 0x00020000);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int row = (t + k - 3) * (kF * 4);           // negative: wraps past the range
#pragma unroll
    for (int i = 0; i < 3; ++i)
      r.v[i][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xu, lane * 4 + 256 * i, row, 0));
  }
  return r;
}
__device__ __forceinline__ void x_store(const XRows& r, char* region, int lane) {   // plane 0 (its stride does not matter)
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int p = lane + 64 * i;
    if (p < kF) {
      const u32x4 q = {pack2(r.v[i][0], r.v[i][1]), pack2(r.v[i][2], r.v[i][3]), pack2(r.v[i][4], r.v[i][5]), pack2(r.v[i][6], r.v[i][7])};
      *reinterpret_cast<u32x4*>(region + (p + kRowPad) * 16) = q;
    }
  }
}

// One layer of one frame (one wave).  `w` = the layer's packet in LDS; `pre` = issued once the first operand reads are in
// flight (the next packet's LDS-DMA, the next tile's input rows).
template <class N, int L, class Pre>
__device__ __forceinline__ void run_layer(const Params& P, char* region, const char* w, const char* shifts, __amdgpu_buffer_rsrc_t scratch, int lane,
                                          long long hrow /* first float of this frame's hand-off rows, < 0: no frame */, Pre pre,
                                          bool stamp = false) {
  using G = Geo<N>;
  F16_STAMP(stamp, 4 * L + 0);
  constexpr LayerDesc D = N::layer[L];
  constexpr int OCT = G::oct_in(L), OCTO = G::oct_out(L), MT = G::MT(L), STEPS = G::steps(L), PADL = G::pad(L);
  constexpr bool kLast = (L == N::kLayers - 1);
  constexpr int NB = OCT < STEPS ? OCT : STEPS;     // per-lane window bases: slot j + 4 OCT is the same octet four taps on
  asm volatile("" : "+v"(lane));                    // no hoisting of every layer's address arithmetic out of the tile loop
  const int n = lane & 15, kq = lane >> 4;

  // skip fragments of the matching encoder layer: issued now, used after the last K-step (lanes without real channels: zero)
  u32x2 skip[D.skip_from >= 0 ? kTiles : 1][D.skip_from >= 0 ? MT : 1];
  if constexpr (D.skip_from >= 0 && !(RCED_F16_EXP & 2)) {
    static_assert(N::layer[D.skip_from >= 0 ? D.skip_from : 0].cout == D.cout, "skip shapes match");
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      constexpr int SF = D.skip_from >= 0 ? D.skip_from : 0;
      const int so = G::skip_off(SF, mt);
      const bool real = kq < G::skip_quads(SF, mt);
#pragma unroll
      for (int t = 0; t < kTiles; ++t) skip[t][mt] = u32x2{0u, 0u};
      if (real) {
#pragma unroll
        for (int t = 0; t + 1 < kTiles; t += 2) {
          const u32x4 q = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(scratch, lane * 16, so + (t / 2) * 1024, 0));
          skip[t][mt] = u32x2{q.x, q.y};
          skip[t + 1][mt] = u32x2{q.z, q.w};
        }
        skip[kTiles - 1][mt] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(scratch, lane * 8, so + 4096, 0));
      }
    }
  }

  int base[NB];
#pragma unroll
  for (int r = 0; r < NB; ++r) {
    const int j = 4 * r + kq;
    base[r] = (j % OCT) * G::kPlane + (n + j / OCT - PADL + kRowPad) * 16;
    asm volatile("" : "+v"(base[r]));
  }
  int wl = lane * 16;
  asm volatile("" : "+v"(wl));
  const char* wp = w + wl;

  f32x4 acc[kTiles][MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const f32x4 sh = *reinterpret_cast<const f32x4*>(shifts + (32 * L + 16 * mt + 4 * kq) * 4);
#pragma unroll
    for (int t = 0; t < kTiles; ++t) acc[t][mt] = sh;
  }
  u32x4 a[2][MT], b[2][kTiles];
  auto load = [&](int s, int buf) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      if (!(RCED_F16_EXP & 8) || s == 0) a[buf][mt] = *reinterpret_cast<const u32x4*>(wp + (s * MT + mt) * 1024);
      else a[buf][mt] = a[buf ^ 1][mt];
#pragma unroll
    for (int t = 0; t < kTiles; ++t)
      b[buf][t] = *reinterpret_cast<const u32x4*>(region + base[s % OCT % NB] + (s / OCT) * 64 + t * 256);
  };
  load(0, 0);
  pin();
  pre();
  pin();
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    if (s + 1 < STEPS) load(s + 1, (s + 1) & 1);
    pin();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < kTiles; ++t) acc[t][mt] = mfma32(a[s & 1][mt], b[s & 1][t], acc[t][mt]);
    pin();
  }
  if constexpr (D.skip_from >= 0 && !(RCED_F16_EXP & 2)) {
    // identity A fragment of the K = 16 instruction: A[m][4 kq + i] = (m == 4 kq + i)
    const int d = n - 4 * kq;
    const u32x2 eye = {d == 0 ? 0x3F80u : d == 1 ? 0x3F800000u : 0u, d == 2 ? 0x3F80u : d == 3 ? 0x3F800000u : 0u};
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int t = 0; t < kTiles; ++t) acc[t][mt] = mfma16(eye, skip[t][mt], acc[t][mt]);
  }
  // ---- epilogue: round to bf16, ReLU, store in place (every read of this layer has been consumed by an MFMA above)
  // The next packet's LDS-DMA (issued a whole K loop ago) and the skip loads have landed: wait for them HERE, in front of the
  // epilogue's global stores (skip fragments, the hand-off tensor), and end the layer on a bare s_barrier -- the stores stay in
  // flight across it instead of exposing their latency at every layer's end.  Nobody reads them before a later layer's wait
  // at this place has retired them (a skip fragment is read two layers later at the earliest).
  F16_STAMP(stamp, 4 * L + 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0x0f70);                // ... and hipcc's wait-count pass knows it (vmcnt(0), nothing else)
  F16_STAMP(stamp, 4 * L + 2);
  // A fragment (tile t, M-tile mt) goes to octet 2 mt + (kq >> 1), bytes 8 (kq & 1) .. + 7 of pixel 16 t + n's row.  An M-tile whose
  // upper octet lies past the layer's last one stores it all the same (zeros: those rows of the packet are zero): the plane is
  // dead, and an unconditional store is one instruction where a lane mask is four.
  char* const out = region + (n + kRowPad) * 16 + (kq >> 1) * G::kPlane + (kq & 1) * 8;
  u32x2 prev[MT];                                   // the even tile of a pair (skip stores are 16 bytes per lane), tile 8 at the end
#pragma unroll
  for (int t = 0; t < kTiles; ++t) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const f32x4 v = acc[t][mt];
      const u32x2 hq = {relu2(pack2(v.x, v.y)), relu2(pack2(v.z, v.w))};
      if constexpr (!kLast) {
        if (t < kTiles - 1) *reinterpret_cast<u32x2*>(out + 2 * mt * G::kPlane + t * 256) = hq;
      } else {
        const int f = 16 * t + n, co0 = 16 * mt + 4 * kq;
        if (hrow >= 0 && f < kF) {
          float* hp = P.h + hrow + (size_t)f * N::kFinalCh + co0;
          if (co0 + 1 < N::kFinalCh) *reinterpret_cast<f32x2*>(hp) = f32x2{__builtin_bit_cast(float, hq.x << 16), __builtin_bit_cast(float, hq.x & 0xffff0000u)};
          if (co0 + 3 < N::kFinalCh) *reinterpret_cast<f32x2*>(hp + 2) = f32x2{__builtin_bit_cast(float, hq.y << 16), __builtin_bit_cast(float, hq.y & 0xffff0000u)};
        }
      }
      if constexpr (D.saves_skip && !(RCED_F16_EXP & 1)) {
        if ((t & 1) && kq < G::skip_quads(L, mt)) {
          __builtin_amdgcn_raw_buffer_store_b128(u32x4{prev[mt].x, prev[mt].y, hq.x, hq.y}, scratch, lane * 16, G::skip_off(L, mt) + (t / 2) * 1024, 0);
          store_wait_state();
        }
      }
      if ((t & 1) == 0) prev[mt] = hq;
    }
  }
  // tile 8: bin 128 alone (lane n = 0) -- the other rows stay zero
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    if constexpr (D.saves_skip && !(RCED_F16_EXP & 1)) {
      if (kq < G::skip_quads(L, mt)) {
        __builtin_amdgcn_raw_buffer_store_b64(prev[mt], scratch, lane * 8, G::skip_off(L, mt) + 4096, 0);
        store_wait_state();
      }
    }
    if constexpr (!kLast) {
      if (n == 0) *reinterpret_cast<u32x2*>(out + 2 * mt * G::kPlane + (kTiles - 1) * 256) = prev[mt];
    }
  }
  F16_STAMP(stamp, 4 * L + 3);
}

// A bare s_barrier (not __syncthreads(), whose release fence waits for every global store in flight): what the four waves hand
// each other is the weight ring alone -- this wave's pieces of the next packet have landed (run_layer's wait in front of its
// epilogue), and it has read the last fragment of the packet that the next layer's DMA will overwrite.
__device__ __forceinline__ void layer_end_sync() {
  if (!(RCED_F16_EXP & 4)) asm volatile("s_barrier" ::: "memory");
}

template <class N, int L>
__device__ __forceinline__ void run_layers(const Params& P, char* lds, char* region, __amdgpu_buffer_rsrc_t scratch, int& wcur,
                                           XRows& xr, int tile, int wave, int lane, long long hrow, bool stamp) {
  using G = Geo<N>;
  if constexpr (L < N::kLayers) {
    constexpr int nxt = (L + 1 < N::kLayers) ? L + 1 : 0;   // the stream wraps: the next tile's first packet
    char* const wbase = lds + G::kWOff;
    char* const wdst = wbase + (wcur ^ 1) * G::kWRegion;
    auto pre = [&] {
      if (!(RCED_F16_EXP & 64)) packet_dma<G::packet_bytes(nxt)>(P.wpack + G::packet_off(nxt) / 4, wdst, wave, lane);
      if constexpr (L == N::kLayers - 3 && !(RCED_F16_EXP & 16)) xr = x_load(P, tile + 1, wave, lane);   // the next tile's input rows, two layers early
    };
    run_layer<N, L>(P, region, wbase + wcur * G::kWRegion, lds + G::kSOff, scratch, lane, hrow, pre, stamp);
    wcur ^= 1;
    layer_end_sync();
    run_layers<N, L + 1>(P, lds, region, scratch, wcur, xr, tile, wave, lane, hrow, stamp);
  }
}

template <class N>
__global__ __launch_bounds__(kThreads, 2) void frame16_kernel(Params P) {
  using G = Geo<N>;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int e = tid; e < G::kActBytes / 16; e += kThreads) reinterpret_cast<u32x4*>(lds)[e] = u32x4{0u, 0u, 0u, 0u};
  for (int e = tid; e < G::kShiftBytes / 4; e += kThreads) reinterpret_cast<unsigned*>(lds + G::kSOff)[e] = P.wpack[G::kShiftOff / 4 + e];
  __syncthreads();
  // a workgroup walks a CONTIGUOUS range of tiles: consecutive frames share seven of their eight input rows (L1 / L2 hits)
  const int per = (P.total_tiles + gridDim.x - 1) / gridDim.x;
  const int first = blockIdx.x * per;
  const int last = first + per < P.total_tiles ? first + per : P.total_tiles;
  if (first >= last) return;
  char* const region = lds + wave * G::kRegion;
  packet_dma<G::packet_bytes(0)>(P.wpack, lds + G::kWOff, wave, lane);
  int wcur = 0;
  XRows xr = x_load(P, first, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const __amdgpu_buffer_rsrc_t scratch = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(P.scratch) + ((size_t)blockIdx.x * kWaves + wave) * G::kScratchBytesPerWave, 0,
      (int)G::kScratchBytesPerWave, 0x00020000);
  layer_end_sync();
  for (int tile = first; tile < last; ++tile) {
    const int utt = tile / P.tiles_per_utt;
    const int t = (tile - utt * P.tiles_per_utt) * kWaves + wave;
    const long long hrow = t < P.T ? ((long long)utt * P.T + t) * kF * N::kFinalCh : -1;
    x_store(xr, region, lane);   // plane 0 of the wave's own image: its last reader was this wave's previous layer 1
    const bool stamp = RCED_F16_STAMPS && P.stamps && blockIdx.x == 0 && wave == 0 && tile == first + 1;
    run_layers<N, 0>(P, lds, region, scratch, wcur, xr, tile, wave, lane, hrow, stamp);
    F16_STAMP(stamp, 4 * N::kLayers);
  }
}

}  // namespace frame16
}  // namespace rced
