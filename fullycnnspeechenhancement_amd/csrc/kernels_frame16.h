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
#define RCED_F16_EXP 0   // timing experiments only (wrong results): 1 = no skip stores, 2 = no skip loads / adds, 4 = no barriers
#endif
#if RCED_F16_EXP != 0 && !defined(RCED_TIMING_ONLY)
#error "RCED_F16_EXP builds compute wrong results: timing experiments only (-DRCED_TIMING_ONLY)"
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
};

template <class N>
struct Geo {
  static constexpr int kLayers = N::kLayers;
  static constexpr int oct_out(int l) { return (N::layer[l].cout + 7) / 8; }
  static constexpr int oct_in(int l) { return l == 0 ? 1 : oct_out(l - 1); }
  static constexpr int MT(int l) { return (N::layer[l].cout + 15) / 16; }
  static constexpr int slots(int l) { return N::layer[l].taps * oct_in(l); }   // K slots of 8 (layer 0: 8 time rows per tap)
  static constexpr int steps(int l) { return (slots(l) + 3) / 4; }
  static constexpr int frags(int l) { return steps(l) * MT(l); }
  static constexpr int packet_bytes(int l) { return frags(l) * 1024 + 128; }
  static constexpr int packet_off(int l) {
    int o = 0;
    for (int i = 0; i < l; ++i) o += packet_bytes(i);
    return o;
  }
  static constexpr int kWBytes = packet_off(kLayers);
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
  static constexpr int kLdsBytes = kWOff + 2 * kWRegion;
  static_assert(kLdsBytes <= 80 * 1024, "two workgroups per CU");
  static constexpr bool pads_ok() {
    for (int l = 0; l < kLayers; ++l)
      if (pad(l) > kRowPad) return false;
    return true;
  }
  static_assert(pads_ok(), "the widest kernel's left halo fits the leading zero rows");
  // skip scratch, per wave: units of 512 bytes (64 lanes x 8), unit = (saving layer, tile, M-tile)
  static constexpr int skip_unit(int l) {
    int u = 0;
    for (int i = 0; i < l; ++i)
      if (N::layer[i].saves_skip) u += kTiles * MT(i);
    return u;
  }
  static constexpr int kSkipUnits = skip_unit(kLayers);
  static constexpr size_t kScratchBytesPerWave = (size_t)kSkipUnits * 512;
};

__device__ __forceinline__ f32x4 mfma32(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(u32x2 a, u32x2 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
}
// two floats -> packed bf16 (round to nearest even; v_cvt_pk_bf16_f32 keeps a NaN a NaN)
__device__ __forceinline__ unsigned pack2(float a, float b) {
  const bf16x2 h = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, h);
}
// ReLU on two packed bf16: a signed 16-bit max with zero (negative values, -0 and NaNs with the sign bit set become +0)
__device__ __forceinline__ unsigned relu2(unsigned v) {
  const s16x2 z = {0, 0};
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), z));
}

// LDS-DMA of one packet, 1-KiB pieces dealt round-robin to the four waves (the last piece is the 128 bytes of shifts)
template <int BYTES>
__device__ __forceinline__ void packet_dma(const unsigned* __restrict__ src, char* dst, int wave, int lane) {
  static_assert(BYTES % 16 == 0, "16-byte pieces");
  constexpr int n16 = BYTES / 16, chunks = (n16 + 63) / 64;
#pragma unroll
  for (int i = 0; i < (chunks + kWaves - 1) / kWaves; ++i) {
    const int c = wave + i * kWaves;
    if (c < chunks) {
      if (c * 64 + lane < n16)
        lds_dma16s(reinterpret_cast<const float*>(src) + c * 256, (unsigned)lane * 16u, reinterpret_cast<float*>(dst + c * 1024));
    }
  }
}

// The eight input rows t - 3 .. t + 4 of one frame, three 64-bin columns per lane, fetched one tile ahead
struct XRows {
  float v[3][8];
};
__device__ __forceinline__ XRows x_load(const Params& P, int tile, int wave, int lane) {
  XRows r;
  const bool live = tile < P.total_tiles;
  const int utt = live ? tile / P.tiles_per_utt : 0;
  const int t = live ? (tile - utt * P.tiles_per_utt) * kWaves + wave : 0;
  const float* xu = P.x + (size_t)utt * P.T * kF;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int p = lane + 64 * i;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int tt = t + k - 3;                       // TF 'SAME' for the 8-tall kernel: 3 rows before, 4 after
      float v = 0.f;
      if (live && p < kF && t < P.T && tt >= 0 && tt < P.T) v = xu[(size_t)tt * kF + p];
      r.v[i][k] = v;
    }
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
__device__ __forceinline__ void run_layer(const Params& P, char* region, const char* w, __amdgpu_buffer_rsrc_t scratch, int lane,
                                          long long hrow /* first float of this frame's hand-off rows, < 0: no frame */, Pre pre) {
  using G = Geo<N>;
  constexpr LayerDesc D = N::layer[L];
  constexpr int OCT = G::oct_in(L), OCTO = G::oct_out(L), MT = G::MT(L), STEPS = G::steps(L), PADL = G::pad(L);
  constexpr bool kLast = (L == N::kLayers - 1);
  constexpr int NB = OCT < STEPS ? OCT : STEPS;     // per-lane window bases: slot j + 4 OCT is the same octet four taps on
  asm volatile("" : "+v"(lane));                    // no hoisting of every layer's address arithmetic out of the tile loop
  const int n = lane & 15, kq = lane >> 4;

  // skip fragments of the matching encoder layer: issued now, used after the last K-step
  u32x2 skip[D.skip_from >= 0 ? kTiles : 1][D.skip_from >= 0 ? MT : 1];
  if constexpr (D.skip_from >= 0 && !(RCED_F16_EXP & 2)) {
    static_assert(N::layer[D.skip_from >= 0 ? D.skip_from : 0].cout == D.cout, "skip shapes match");
#pragma unroll
    for (int t = 0; t < kTiles; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        skip[t][mt] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(
                                                    scratch, lane * 8, (G::skip_unit(D.skip_from) + t * MT + mt) * 512, 0));
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
    const f32x4 sh = *reinterpret_cast<const f32x4*>(w + G::frags(L) * 1024 + (16 * mt + 4 * kq) * 4);
#pragma unroll
    for (int t = 0; t < kTiles; ++t) acc[t][mt] = sh;
  }
  u32x4 a[2][MT], b[2][kTiles];
  auto load = [&](int s, int buf) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a[buf][mt] = *reinterpret_cast<const u32x4*>(wp + (s * MT + mt) * 1024);
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
  char* const out = region + (n + kRowPad) * 16 + (kq >> 1) * G::kPlane + (kq & 1) * 8;
#pragma unroll
  for (int t = 0; t < kTiles; ++t) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const f32x4 v = acc[t][mt];
      const u32x2 hq = {relu2(pack2(v.x, v.y)), relu2(pack2(v.z, v.w))};
      if constexpr (D.saves_skip && !(RCED_F16_EXP & 1)) {
        __builtin_amdgcn_raw_buffer_store_b64(hq, scratch, lane * 8, (G::skip_unit(L) + t * MT + mt) * 512, 0);
        store_wait_state();
      }
      if constexpr (!kLast) {
        // octet 2 mt + (kq >> 1); the planes past the layer's last octet are not written (nobody reads them)
        const bool oct_ok = 2 * mt + 1 < OCTO || kq < 2;
        if (oct_ok && (t < kTiles - 1 || n == 0)) *reinterpret_cast<u32x2*>(out + 2 * mt * G::kPlane + t * 256) = hq;
      } else {
        const int f = 16 * t + n, co0 = 16 * mt + 4 * kq;
        if (hrow >= 0 && f < kF) {
          float* hp = P.h + hrow + (size_t)f * N::kFinalCh + co0;
          if (co0 + 1 < N::kFinalCh) *reinterpret_cast<f32x2*>(hp) = f32x2{__builtin_bit_cast(float, hq.x << 16), __builtin_bit_cast(float, hq.x & 0xffff0000u)};
          if (co0 + 3 < N::kFinalCh) *reinterpret_cast<f32x2*>(hp + 2) = f32x2{__builtin_bit_cast(float, hq.y << 16), __builtin_bit_cast(float, hq.y & 0xffff0000u)};
        }
      }
    }
  }
}

__device__ __forceinline__ void layer_end_sync() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the next packet have landed; its skip stores are out
  __builtin_amdgcn_s_waitcnt(0x0f70);                // ... and hipcc's wait-count pass knows it (vmcnt(0), nothing else)
  if (!(RCED_F16_EXP & 4)) __syncthreads();
}

template <class N, int L>
__device__ __forceinline__ void run_layers(const Params& P, char* lds, char* region, __amdgpu_buffer_rsrc_t scratch, int& wcur,
                                           XRows& xr, int tile, int wave, int lane, long long hrow) {
  using G = Geo<N>;
  if constexpr (L < N::kLayers) {
    constexpr int nxt = (L + 1 < N::kLayers) ? L + 1 : 0;   // the stream wraps: the next tile's first packet
    char* const wbase = lds + G::kWOff;
    char* const wdst = wbase + (wcur ^ 1) * G::kWRegion;
    auto pre = [&] {
      packet_dma<G::packet_bytes(nxt)>(P.wpack + G::packet_off(nxt) / 4, wdst, wave, lane);
      if constexpr (L == N::kLayers - 3) xr = x_load(P, tile + 1, wave, lane);   // the next tile's input rows, two layers early
    };
    run_layer<N, L>(P, region, wbase + wcur * G::kWRegion, scratch, lane, hrow, pre);
    wcur ^= 1;
    layer_end_sync();
    run_layers<N, L + 1>(P, lds, region, scratch, wcur, xr, tile, wave, lane, hrow);
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
  const __amdgpu_buffer_rsrc_t scratch = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(P.scratch) + ((size_t)blockIdx.x * kWaves + wave) * G::kScratchBytesPerWave, 0,
      (int)G::kScratchBytesPerWave, 0x00020000);
  layer_end_sync();
  for (int tile = first; tile < last; ++tile) {
    const int utt = tile / P.tiles_per_utt;
    const int t = (tile - utt * P.tiles_per_utt) * kWaves + wave;
    const long long hrow = t < P.T ? ((long long)utt * P.T + t) * kF * N::kFinalCh : -1;
    x_store(xr, region, lane);   // plane 0 of the wave's own image: its last reader was this wave's previous layer 1
    run_layers<N, 0>(P, lds, region, scratch, wcur, xr, tile, wave, lane, hrow);
  }
}

}  // namespace frame16
}  // namespace rced
