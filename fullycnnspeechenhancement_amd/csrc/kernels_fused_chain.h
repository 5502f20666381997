// Fused R-CED (V1: model_utils/model.py:6-29, V2: model.py:32-61) forward: every layer but the last
// 1x129 one in ONE kernel, fp32 MFMA; generic over a compile-time layer table.
//
// Same construction as the CR-CED kernel (kernels_fused_v3.h, which carries the long explanation):
// a tile of frames lives in LDS as [pixel][channel] with the channel stride equal to the (even-padded)
// channel count, a 1xk conv is an implicit GEMM whose B operand is a ds_read_b64 out of that buffer,
// cout sits on the 16-row MFMA M axis, weights arrive as pre-packed A-fragment packets by LDS-DMA
// one layer ahead.  Differences:
//   * the nets are plain encoder/decoder chains, so activations ping-pong between two buffers;
//   * module.py:30-31 skips (decoder layer += encoder output, BEFORE the ReLU) are too many to hold in
//     registers (V2: 114 channels), so an encoder layer that feeds a skip also stores its output
//     fragments, in MFMA D-layout order, to a per-workgroup scratch in global memory (L2/MALL
//     resident, fully coalesced 1 KiB per wave-store); the matching decoder layer has the same cout,
//     hence the same tile -> wave map and fragment layout, and simply loads them back (issued at the
//     start of the layer, consumed in its epilogue).  A wave only ever re-reads its own stores.
//   * layers with 17..24 output channels (V1: 20, 24; V2: 19, 21, 23) would fill two 16-row M-tiles 53-75 %: they run
//     channels 0..15 as ONE M-tile (the main pass) plus a REMAINDER pass that computes channels 16.. for P adjacent
//     pixels at once -- rows = (pixel phase p < P, channel 16 + c), P = 16 / (cout - 16), K = (taps + P - 1) * cin, one
//     column per group of P pixels -- as the CR-CED kernel does for its ->18 layers: 16 % (V1) / 14 % (V2) fewer MFMAs.
#pragma once
#include <hip/hip_runtime.h>

#include "lds_dma.h"

#ifndef RCED_CHAIN_EXP
#define RCED_CHAIN_EXP 0     // timing experiments only (wrong results): 1 = no skip-fragment stores, 2 = no skip-fragment loads,
                             // 4 = the two-M-tile layers issue half of their MFMAs (the bound on a bf16-pipe form of those layers),
                             // 8 = every wave runs the extra-tile copy of a layer's code: ONE copy per layer, 49 / 61 KB of code instead of
                             //     88 / 109 KB; 24 = the same work from THREE marked copies (125 / 155 KB).  Measured (A/B in one call): V1
                             //     11.63 ms with one copy, 11.48 with three; V2 13.92 / 13.80 -- the instruction cache (64 KB per two CUs)
                             //     is NOT what these kernels wait for
                             // 32 = no layer-end wait and no barrier at all, 64 = the wait but no barrier: V2 12.01 -> 11.08 / 11.06 ms, V1 10.20 ->
                             //     9.83 / 9.84: the fifteen (nine) barriers cost 7.9 % (3.6 %), the wait for the next packet nothing
#endif
#if RCED_CHAIN_EXP != 0 && !defined(RCED_TIMING_ONLY)
#error "RCED_CHAIN_EXP builds compute wrong results: timing experiments only (tools/mkexp.sh ... -DRCED_TIMING_ONLY -DRCED_CHAIN_EXP=...)"
#endif
#ifndef RCED_CHAIN_DEPTH
#define RCED_CHAIN_DEPTH 1   // operand prefetch depth (b64 steps) of the fp32 R-CED passes
#endif

namespace rced {
namespace chain {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kF = 129;
constexpr int kWaves = 8;
constexpr int kThreads = kWaves * 64;
constexpr int kMaxLayers = 15;

struct LayerDesc {
  int cin, cinp;    // true / even-padded input channels (cinp = channel stride of the input buffer)
  int taps;         // kernel width along frequency
  int cout, coutp;  // true / even-padded output channels
  int skip_from;    // layer whose output is added before the ReLU, or -1
  int saves_skip;   // 1 if a later layer adds this layer's output
};

constexpr int even(int c) { return (c + 1) & ~1; }

// ---- the two nets (first layer: 8 x taps kernel on the 1-channel input) ------------------------
struct NetV1 {
  static constexpr int kVariant = 1;
  static constexpr int kLayers = 9;          // decode_5 (1x129) is the separate final GEMM
  static constexpr int kTF = 3;              // frames per tile
  static constexpr int kGap = 6;             // >= half width of the widest kernel (13)
  static constexpr int kFinalCh = 12;
  static constexpr LayerDesc layer[kMaxLayers] = {
      // channel strides 16 and 32 would put a tile's 16 pixels on 4 resp. 2 distinct LDS bank groups (4- / 8-way
      // conflicts on every B-operand read); 18 and 34 spread them over all banks at the price of 2 zero-weight
      // k per tap
      {1, 1, 13, 12, 12, -1, 1},  {12, 12, 11, 16, 18, -1, 1}, {16, 18, 9, 20, 20, -1, 1},
      {20, 20, 7, 24, 24, -1, 1}, {24, 24, 7, 32, 34, -1, 0},  {32, 34, 7, 24, 24, 3, 0},
      {24, 24, 9, 20, 20, 2, 0},  {20, 20, 11, 16, 18, 1, 0},  {16, 18, 13, 12, 12, 0, 0}};
};
struct NetV2 {
  static constexpr int kVariant = 2;
  static constexpr int kLayers = 15;         // decode_8 (1x129) is the separate final GEMM
  static constexpr int kTF = 3;
  static constexpr int kGap = 5;             // widest kernel 11
  static constexpr int kFinalCh = 10;
  static constexpr LayerDesc layer[kMaxLayers] = {
      {1, 1, 11, 10, 10, -1, 1},   {10, 10, 7, 12, 12, -1, 1},  {12, 12, 5, 14, 14, -1, 1},
      {14, 14, 5, 15, 16, -1, 1},  {15, 16, 5, 19, 20, -1, 1},  {19, 20, 5, 21, 22, -1, 1},
      {21, 22, 7, 23, 24, -1, 1},  {23, 24, 11, 25, 26, -1, 0}, {25, 26, 7, 23, 24, 6, 0},
      {23, 24, 5, 21, 22, 5, 0},   {21, 22, 5, 19, 20, 4, 0},   {19, 20, 5, 15, 16, 3, 0},
      {15, 16, 5, 14, 14, 2, 0},   {14, 14, 7, 12, 12, 1, 0},   {12, 12, 11, 10, 10, 0, 0}};
};

// The same net with another number of frames per tile: geometry only (the packets do not depend on it).  Used by the fp32 kernel's
// latency form (one-frame tiles when a call has fewer tiles than the part has CUs).
template <class N, int TF>
struct WithTF : N {
  static constexpr int kTF = TF;
};

// ---- derived geometry ------------------------------------------------------------------------
template <class N>
struct Geo {
  static constexpr int kS = kF + N::kGap;                 // pixel stride of a frame
  static constexpr int kNPX = N::kTF * kS;                // pixels per tile
  static constexpr int kTiles = (kNPX + 15) / 16;         // every pixel of the tile is (re)written each layer
  static constexpr int kRegular = kTiles / kWaves;        // slots every wave has
  static constexpr int kExtra = kTiles - kRegular * kWaves;   // waves 0..kExtra-1 have one more
  static_assert(kExtra <= 4, "extra tiles go to waves 0..3 (one per SIMD)");
  // Layers with two M-tiles cut each extra tile in two (by M-tile) when that gives at most one piece per SIMD:
  // piece u = wave (< 2*kExtra) is M-tile u / kExtra of extra tile u % kExtra.  The barrier that ends a layer
  // waits for the most loaded SIMD, so 6.5 tiles on each beats 7 / 7 / 6 / 6.
  static constexpr bool kSplitExtra = kExtra > 0 && 2 * kExtra <= 4;
  static constexpr int kPad = N::kGap;                    // leading zero rows of each buffer
  static constexpr int kRows = kPad + 16 * kTiles;        // pixels -pad .. 16*tiles-1 (reads past land in the next buffer)
  static_assert(kPad == (N::layer[0].taps - 1) / 2, "X0 indexing assumes pad == half width of the first kernel");
  static constexpr int chmax(int parity) {
    int m = 0;
    for (int l = parity; l < N::kLayers; l += 2) m = N::layer[l].coutp > m ? N::layer[l].coutp : m;
    return m;
  }
  static constexpr int kChX = chmax(0), kChY = chmax(1);  // buffer X holds outputs of even layers
  static constexpr int kXOff = 0;
  static constexpr int kYOff = kXOff + kRows * kChX;
  static constexpr int kWOff = ((kYOff + kRows * kChY + 3) / 4) * 4;
  // remainder pass of layer l (0 = none): R channels past the first M-tile, P pixel phases per column, K, tiles
  static constexpr int R(int l) {
    const int c = N::layer[l].cout;
    return (l > 0 && c > 16 && c - 16 <= 8) ? c - 16 : 0;
  }
  static constexpr int PH(int l) { return R(l) ? 16 / R(l) : 0; }
  static constexpr int KR(int l) { return (N::layer[l].taps + PH(l) - 1) * N::layer[l].cinp; }
  // Remainder columns are FRAME-ALIGNED: column c = (frame c / CPF, pixels PH (c % CPF) .. + PH - 1 of it), CPF = ceil(kS / PH) columns per
  // frame: bins AND gap pixels (a layer rewrites every pixel of its buffer, whose previous contents are another layer's layout; the gap
  // pixels get zeros), the last column's pixels past the frame's stride dropped -- so a bin's phase in its column, and with it the
  // grouping of its taps into K-steps, does not depend on where the frame sits in the tile: a frame's result is the same bits in a tile
  // of one frame (the latency form) and of three.  As many tiles as flat columns over the tile's pixels took.
  static constexpr int CPF(int l) { return R(l) ? (kS + PH(l) - 1) / PH(l) : 0; }
  static constexpr int NRT(int l) { return R(l) ? (N::kTF * CPF(l) + 15) / 16 : 0; }
  static constexpr int kRemSlots = 2;                       // remainder tiles per wave, at most (waves 7..2 first, see rem_tile)
  // packet of layer l (floats): main pass b64 steps, b32 tails; remainder pass likewise; 32 shifts; 16 remainder-row shifts
  static constexpr int K(int l) { return (l == 0 ? 8 : 1) * N::layer[l].taps * N::layer[l].cinp; }
  static constexpr int MT(int l) { return R(l) ? 1 : (N::layer[l].cout + 15) / 16; }   // M-tiles of the MAIN pass
  static constexpr int NB64(int l) { return l == 0 ? 0 : K(l) / 8; }
  static constexpr int NTAIL(int l) { return l == 0 ? 2 * N::layer[0].taps : (K(l) % 8 + 3) / 4; }   // b32 steps
  static constexpr int main_data(int l) { return NB64(l) * MT(l) * 128 + NTAIL(l) * MT(l) * 64; }
  static constexpr int rem_data(int l) { return R(l) ? (KR(l) / 8) * 128 + ((KR(l) % 8 + 3) / 4) * 64 : 0; }
  static constexpr int data(int l) { return main_data(l) + rem_data(l); }
  static constexpr int packet(int l) { return data(l) + 32 + (R(l) ? 16 : 0); }
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
  static constexpr bool kFitsLds = kLdsBytes <= 160 * 1024;   // asserted where the fp32 kernel is instantiated
  // input rows of the first layer alias buffer Y (dead until layer 1 writes it)
  static constexpr int kX0Rows = N::kTF + 7;
  static constexpr int kX0Floats = ((kX0Rows * kS + 32 + 3) / 4) * 4;
  static constexpr int kX0Off = kYOff + kPad * kChY;
  static_assert(kX0Floats <= 4 * kThreads, "XStage holds 4 floats per thread");
  static constexpr bool kX0Fits = kX0Floats <= kRows * kChY - kPad * kChY;
  // skip scratch: units = (layer that saves, slot, mt), each kThreads x float4
  static constexpr int skip_unit(int l) {   // first unit index of saving layer l
    int u = 0;
    for (int i = 0; i < l; ++i)
      if (N::layer[i].saves_skip) u += (kRegular + 1) * MT(i) + (R(i) ? kRemSlots : 0);
    return u;
  }
  static constexpr int skip_unit_rem(int l) { return skip_unit(l) + (kRegular + 1) * MT(l); }   // its remainder-tile units
  static constexpr int kSkipUnits = skip_unit(N::kLayers);
  static constexpr size_t kScratchFloatsPerWg = (size_t)kSkipUnits * kThreads * 4;
};

// Remainder tile j (0 / 1) of wave w: tiles 0..5 go to waves 7..2 (waves 0, 1 carry the two odd main tiles), tiles 6, 7 to
// waves 0, 1, tiles 8..13 to waves 7..2 again; >= the layer's tile count: none.  (Per-SIMD MFMA counts, V1's 16 -> 20
// layer: 148 / 94 / 108 / 108 on top of six main tiles each; the barrier that ends the layer waits for the fullest.)
__device__ __forceinline__ int rem_tile(int wave, int j) { return wave >= 2 ? (j == 0 ? 7 - wave : 15 - wave) : (j == 0 ? 6 + wave : 99); }

struct Params {
  const float* x;       // [N, T, 129]
  float* h;             // [N*T, 129, kFinalCh]: input of the final 1x129 layer
  const float* wpack;   // Geo::kWTotal floats
  float* scratch;       // gridDim.x * Geo::kScratchFloatsPerWg floats (skip fragments)
  int N, T;
  int tiles_per_utt;
  int total_tiles;
};

__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void pin() { __builtin_amdgcn_sched_barrier(0); }
__device__ __forceinline__ float relu1(float v) {   // integer max: see kernels_fused_v3.h relu1
  const int b = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, b > 0 ? b : 0);
}
__device__ __forceinline__ f32x4 relu4(f32x4 v) { return f32x4{relu1(v.x), relu1(v.y), relu1(v.z), relu1(v.w)}; }

// SBASE: chunk base in SGPRs + one 32-bit lane offset (lds_dma16s) instead of a 64-bit address per lane and chunk.  It
// frees ~45 VGPRs here but the per-layer chunk bases then live in (spilled) SGPRs: A/B -1 % (V1), -2 % (V2), -0.3 %
// (CR-CED); only the bf16 kernel, capped at 128 VGPRs for two workgroups per CU, gains (no scratch any more, +0.5-1 %).
template <int NFLOATS, bool SBASE = false>
__device__ __forceinline__ void packet_dma(const float* __restrict__ src, float* dst, int wave, int lane) {
  constexpr int n4 = NFLOATS / 4;
  constexpr int chunks = (n4 + 63) / 64;
#pragma unroll
  for (int i = 0; i < (chunks + kWaves - 1) / kWaves; ++i) {
    const int c = wave + i * kWaves;
    if (c < chunks) {
      const int idx = c * 64 + lane;
      if (idx < n4) {
        if constexpr (SBASE) lds_dma16s(src + c * 256, (unsigned)lane * 16u, dst + c * 256);
        else lds_dma16(src + (size_t)idx * 4, dst + c * 256);
      }
    }
  }
}
__device__ __forceinline__ void layer_end_sync() {
  if (RCED_CHAIN_EXP & 32) return;   // timing experiment (wrong results): no wait for the next packet, no barrier (the skip-saving layers keep theirs)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (RCED_CHAIN_EXP & 64) return;   // ... the wait, but no barrier
  __syncthreads();
}

// Implicit-GEMM pass: NB64 b64 steps (k = 8s + 2kq + e) then NTAIL b32 steps (k = 8*NB64 + 4j + kq),
// NT = NR regular slots (offsets off0 + t*STRIDE) + NX extra slot (offx), every M-tile.
// Lanes whose tail k is past K re-read in-window data (their weights are zero).
// XMT >= 0: the extra slot only computes M-tile XMT (the tile's other M-tile belongs to another wave).
// `pre` runs once the first operand reads are in flight: a layer's main pass issues the next packet's LDS-DMA there
// (~30 mostly scalar instructions in the shadow of the reads' latency instead of between the barrier and the first read).
// `each(s)` runs in front of b64 step s's MFMAs (s is a constant after unrolling): the training kernels issue the next
// tile's global loads there, one 16-byte piece per step, instead of in a burst that stalls on the memory pipeline's queues.
struct NoPre {
  __device__ __forceinline__ void operator()() const {}
};
struct NoEach {
  __device__ __forceinline__ void operator()(int) const {}
};
template <int NR, int NX, int MT, int K, int STRIDE, int DEPTH, int XMT = -1, class Pre = NoPre, class Each = NoEach>
__device__ __forceinline__ void gemm_pass(const float* act, int off0, int offx, const float* w, int lane,
                                          f32x4 (&acc)[NR + NX][MT], Pre pre = Pre(), Each each = Each()) {
  constexpr int NT = NR + NX, RING = DEPTH + 1;
  constexpr int NB64 = K / 8, NTAIL = (K % 8 + 3) / 4;
  const int kq = lane >> 4;
  const f32x2* wp = reinterpret_cast<const f32x2*>(w) + lane;
  const float* wt = w + NB64 * MT * 128 + lane;
  f32x2 a[RING][MT], b[RING][NT];
  float at[NTAIL == 0 ? 1 : NTAIL][MT], bt[NTAIL == 0 ? 1 : NTAIL][NT];
  // One base register per tile, hidden from the optimiser: with one base and NT immediates hipcc fuses the tiles' reads
  // of a step into ds_read2st64_b64 pairs, each of which needs a v_add for its re-based address -- VALU work that is
  // not hidden behind fp32 MFMAs (they share the vector ALU) -- and takes the LDS cycles of two reads anyway.
  int offs[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    offs[t] = t < NR ? off0 + t * STRIDE : offx;
    asm volatile("" : "+v"(offs[t]));
  }
#pragma unroll
  for (int j = 0; j < NTAIL; ++j) {
    constexpr int dummy = 0;
    (void)dummy;
    const int valid = K - 8 * NB64 - 4 * j;                  // real k in this tail step (1..4, or more)
    const int kqe = valid >= 4 ? kq : (kq < valid ? kq : valid - 1);
    const int d = 8 * NB64 + 4 * j + kqe - 2 * kq;            // off0 already carries + 2*kq
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) at[j][mt] = wt[(j * MT + mt) * 64];
#pragma unroll
    for (int t = 0; t < NT; ++t) bt[j][t] = act[offs[t] + d];
  }
  auto load = [&](int s, f32x2(&as)[MT], f32x2(&bs)[NT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) as[mt] = wp[(s * MT + mt) * 64];
#pragma unroll
    for (int t = 0; t < NT; ++t) bs[t] = *reinterpret_cast<const f32x2*>(act + offs[t] + 8 * s);
  };
#pragma unroll
  for (int s = 0; s < DEPTH && s < NB64; ++s) load(s, a[s % RING], b[s % RING]);
  pin();
  pre();
  pin();
#pragma unroll
  for (int s = 0; s < NB64; ++s) {
    if (s + DEPTH < NB64) load(s + DEPTH, a[(s + DEPTH) % RING], b[(s + DEPTH) % RING]);
    each(s);
    pin();
#pragma unroll
    for (int e = 0; e < ((RCED_CHAIN_EXP & 4) && MT == 2 ? 1 : 2); ++e)      // (EXP 4: half of the two-M-tile layers' MFMAs)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          if (t < NR || XMT < 0 || mt == XMT) acc[t][mt] = mfma(a[s % RING][mt][e], b[s % RING][t][e], acc[t][mt]);
    pin();
  }
#pragma unroll
  for (int j = 0; j < NTAIL; ++j)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        if (t < NR || XMT < 0 || mt == XMT) acc[t][mt] = mfma(at[j][mt], bt[j][t], acc[t][mt]);
}

// First layer: 8 x TAPS kernel on the 1-channel input, b32 steps; step s = ih*TAPS + j, lane kq <->
// time tap 4*ih + kq.  x0 index of (input row r, bin f') is PADL + r*S + f'  =>  pixel + ti*S + j.
template <int NR, int NX, int TAPS, int S, int DEPTH, class Pre>
__device__ __forceinline__ void first_pass(const float* x0, int off0, int offx, const float* w, int lane,
                                           f32x4 (&acc)[NR + NX][1], Pre pre) {
  constexpr int NT = NR + NX, RING = DEPTH + 1, STEPS = 2 * TAPS;
  const float* wp = w + lane;
  float a[RING], b[RING][NT];
  auto load = [&](int s, int buf) {
    a[buf] = wp[s * 64];
    const int d = (s / TAPS) * 4 * S + (s % TAPS);
#pragma unroll
    for (int t = 0; t < NR; ++t) b[buf][t] = x0[off0 + t * 128 + d];
    if constexpr (NX > 0) b[buf][NR] = x0[offx + d];
  };
#pragma unroll
  for (int s = 0; s < DEPTH; ++s) load(s, s % RING);
  pin();
  pre();
  pin();
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    if (s + DEPTH < STEPS) load(s + DEPTH, (s + DEPTH) % RING);
    pin();
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t][0] = mfma(a[s % RING], b[s % RING][t], acc[t][0]);
    pin();
  }
}

template <class N>
__device__ __forceinline__ bool px_valid(int px) {
  using G = Geo<N>;
  const int fr = px / G::kS;
  return px < G::kNPX && (px - fr * G::kS) < kF;
}
template <class N>
__device__ __forceinline__ bool span_has_gap(int p0, int len) {   // wave-uniform arguments
  using G = Geo<N>;
  const int fr = p0 / G::kS;
  return p0 + len > G::kNPX || (p0 - fr * G::kS) + len > kF;
}

struct XStage {
  float v0, v1, v2, v3;
};
template <class N>
__device__ __forceinline__ float xstage_one(const Params& P, bool live, const float* xu, int t0, int e) {
  using G = Geo<N>;
  const int q = e - G::kPad;
  const int r = q >= 0 ? q / G::kS : -1;
  const int f = q - r * G::kS;
  const int tt = t0 + r - 3;
  float v = 0.f;
  if (live && e < G::kX0Floats && q >= 0 && r < G::kX0Rows && f < kF && tt >= 0 && tt < P.T)
    v = xu[(size_t)tt * kF + f];
  return v;
}
template <class N>
__device__ __forceinline__ XStage xstage_load(const Params& P, int tile, int tid) {
  const bool live = tile < P.total_tiles;
  const int utt = live ? tile / P.tiles_per_utt : 0;
  const int t0 = live ? (tile - utt * P.tiles_per_utt) * N::kTF : 0;
  const float* xu = P.x + (size_t)utt * P.T * kF;
  XStage st;
  st.v0 = xstage_one<N>(P, live, xu, t0, tid);
  st.v1 = xstage_one<N>(P, live, xu, t0, tid + kThreads);
  st.v2 = xstage_one<N>(P, live, xu, t0, tid + 2 * kThreads);
  st.v3 = xstage_one<N>(P, live, xu, t0, tid + 3 * kThreads);
  return st;
}
template <class N>
__device__ __forceinline__ void xstage_store(const XStage& st, float* x0, int tid) {
  using G = Geo<N>;
  x0[tid] = st.v0;
  if (tid + kThreads < G::kX0Floats) x0[tid + kThreads] = st.v1;
  if (tid + 2 * kThreads < G::kX0Floats) x0[tid + 2 * kThreads] = st.v2;
  if (tid + 3 * kThreads < G::kX0Floats) x0[tid + 3 * kThreads] = st.v3;
}

// Per-lane window / output offsets of a wave's regular tile `wave`, every layer's, computed ONCE per kernel and kept in
// registers: re-derived in every layer they were ~20 VALU instructions in front of each layer's first LDS read, and VALU
// work is not hidden behind fp32 MFMAs.  A layer takes its two values through an opaque copy, so that nothing DERIVED from
// them is hoisted out of the tile loop as well (hoisted wholesale, the layers' address arithmetic spilled).
template <class N>
struct LaneTab {
  int in[N::kLayers];    // (px0 - padl) * cinp + 2 kq: B-operand window start
  int out[N::kLayers];   // px0 * coutp + 4 kq: this lane's output channels 4kq.. of pixel px0 (M-tile 0)
};
template <class N>
__device__ __forceinline__ LaneTab<N> make_lane_tab(int wave, int lane) {
  LaneTab<N> T;
  const int n = lane & 15, kq = lane >> 4, px0 = 16 * wave + n;
#pragma unroll
  for (int l = 0; l < N::kLayers; ++l) {
    T.in[l] = (px0 - (N::layer[l].taps - 1) / 2) * N::layer[l].cinp + 2 * kq;
    T.out[l] = px0 * N::layer[l].coutp + 4 * kq;
    asm volatile("" : "+v"(T.in[l]), "+v"(T.out[l]));
  }
  return T;
}

// One layer for one wave.  NX = 1 for the waves that own an extra tile (XMT < 0) or one M-tile of it (XMT = 0/1).
template <class N, int L, int NX, int XMT = -1, class Dma>
__device__ __forceinline__ void run_layer(const Params& P, float* lds, const float* w, __amdgpu_buffer_rsrc_t scratch,
                                          int wave, int lane, int tid, int utt, int t0, Dma dma, const LaneTab<N>& LT) {
  using G = Geo<N>;
  int lt_in = LT.in[L], lt_out = LT.out[L];
  asm volatile("" : "+v"(lt_in), "+v"(lt_out));
  constexpr LayerDesc D = N::layer[L];
  constexpr int NR = G::kRegular, NT = NR + NX, MT = G::MT(L);
  constexpr bool kLast = (L == N::kLayers - 1);
  // Re-derive the lane coordinates behind an opaque barrier in EVERY layer: otherwise hipcc hoists the
  // address arithmetic of all 9-15 layers out of the tile loop, keeps ~100 values live and spills.
  asm volatile("" : "+v"(lane), "+v"(tid));
  const int n = lane & 15, kq = lane >> 4;
  float* bufx = lds + G::kXOff + G::kPad * G::kChX;
  float* bufy = lds + G::kYOff + G::kPad * G::kChY;
  const float* in = (L % 2 == 1) ? bufx : bufy;          // layer L reads what layer L-1 wrote
  float* out = (L % 2 == 0) ? bufx : bufy;
  const int xtile = G::kRegular * kWaves + (XMT < 0 ? wave : wave % (G::kExtra > 0 ? G::kExtra : 1));   // this wave's extra tile
  const int px0 = 16 * wave + n, pxx = 16 * xtile + n;

  // skip fragments of the matching encoder layer: issue the loads now, use them in the epilogue
  f32x4 skip[D.skip_from >= 0 ? NT : 1][D.skip_from >= 0 ? MT : 1];
  if constexpr (D.skip_from >= 0) {
    static_assert(N::layer[D.skip_from >= 0 ? D.skip_from : 0].cout == D.cout, "skip shapes match");
    // buffer loads: one descriptor (SGPRs) + tid*16 (one VGPR) + a scalar unit offset; plain pointers
    // made hipcc keep a 64-bit address per unit live across the whole tile loop and spill them
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        if (t < NR || XMT < 0 || mt == XMT)
        skip[t][mt] = (RCED_CHAIN_EXP & 2) ? f32x4{0.f, 0.f, 0.f, 0.f} : __builtin_bit_cast(
            f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                       scratch, tid * 16, (G::skip_unit(D.skip_from) + t * MT + mt) * kThreads * 16, 0));
  }

  f32x4 acc[NT][MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const f32x4 sh = *reinterpret_cast<const f32x4*>(w + G::data(L) + 16 * mt + 4 * kq);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t][mt] = sh;
  }
  if constexpr (L == 0) {
    first_pass<NR, NX, D.taps, G::kS, 2>(lds + G::kX0Off, px0 + kq * G::kS, pxx + kq * G::kS, w, lane, acc, dma);
  } else {
    gemm_pass<NR, NX, MT, G::K(L), 128 * D.cinp, RCED_CHAIN_DEPTH, XMT>(in, lt_in, lt_in + 16 * (xtile - wave) * D.cinp,
                                                                        w, lane, acc, dma);
  }
  // ---- epilogue: (+skip) -> ReLU -> zero the gap pixels -> LDS (or the hand-off tensor)
  // A layer that stores to global memory here (skip fragments, the hand-off tensor) waits NOW for the next packet's
  // LDS-DMA -- issued a whole pass ago, so this costs nothing -- and ends on a bare barrier: its stores stay in flight
  // across the barrier instead of exposing their latency in front of it (vmcnt counts loads and stores alike).  Nobody
  // reads them before a later layer's vmcnt(0) has retired them.  The other layers wait at their end (layer_end_sync).
  if constexpr (D.saves_skip || kLast) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // One fragment (tile slot t, M-tile mt): (+skip) -> ReLU -> (gap pixels: zero) -> skip scratch -> LDS / the hand-off tensor
  auto fragment = [&](int t, int mt, int px, bool masked, bool ok) {
    if (t >= NR && XMT >= 0 && mt != XMT) return;          // the other M-tile of the split extra tile is another wave's
    const int fr = px / G::kS, f = px - fr * G::kS;
    f32x4 v = acc[t][mt];
    if constexpr (D.skip_from >= 0) v += skip[t][mt];   // module.py:30-31: before the ReLU
    v = relu4(v);
    if (masked && !ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (D.saves_skip && !(RCED_CHAIN_EXP & 1))
    {
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), scratch, tid * 16,
                                             (G::skip_unit(L) + t * MT + mt) * kThreads * 16, 0);
      store_wait_state();
    }
    const int co0 = 16 * mt + 4 * kq;
    if constexpr (!kLast) {
      float* p = out + lt_out + (t < NR ? 128 * t : 16 * (xtile - wave)) * D.coutp + 16 * mt;
      if (16 * mt + 16 <= D.coutp || co0 + 1 < D.coutp) *reinterpret_cast<f32x2*>(p) = f32x2{v.x, v.y};
      if (16 * mt + 16 <= D.coutp || co0 + 3 < D.coutp) *reinterpret_cast<f32x2*>(p + 2) = f32x2{v.z, v.w};
    } else if (ok && px < G::kNPX && f < kF && t0 + fr < P.T) {
      float* hp = P.h + (((size_t)utt * P.T + t0 + fr) * kF + f) * N::kFinalCh + co0;
      if (co0 + 1 < N::kFinalCh) *reinterpret_cast<f32x2*>(hp) = f32x2{v.x, v.y};
      if (co0 + 3 < N::kFinalCh) *reinterpret_cast<f32x2*>(hp + 2) = f32x2{v.z, v.w};
    }
  };
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int tile = t < NR ? wave + kWaves * t : xtile;
    const int px = t < NR ? px0 + 128 * t : pxx;
    // Wave-uniform: 3 of the 26 tiles contain gap pixels.  ONE branch per tile with the tile's fragments in both arms
    // (as selects, every tile paid the validity arithmetic and four v_cndmask per fragment; as a branch per fragment the
    // condition went through a VGPR and back at every join).
    if (span_has_gap<N>(16 * tile, 16)) {
      asm volatile("");   // keeps the arms apart
      const bool ok = px_valid<N>(px);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) fragment(t, mt, px, true, ok);
    } else {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) fragment(t, mt, px, false, true);
    }
  }
  // ---- remainder pass: channels 16.. of P adjacent pixels per column (rows 4kq+j = (phase i / R, channel 16 + i % R))
  if constexpr (G::R(L) > 0) {
    static_assert(!kLast && MT == 1, "remainder layers are inner layers with one main M-tile");
    constexpr int R = G::R(L), PHS = G::PH(L), KR = G::KR(L), NRT = G::NRT(L), padl = (D.taps - 1) / 2;
    static_assert(NRT <= 14, "rem_tile hands out at most 14 remainder tiles");
    const float* wr = w + G::main_data(L);
    const f32x4 rsh = *reinterpret_cast<const f32x4*>(w + G::data(L) + 32 + 4 * kq);
#pragma unroll 1
    for (int jj = 0; jj < G::kRemSlots; ++jj) {
      const int rt = rem_tile(wave, jj);            // wave-uniform
      if (rt >= NRT) break;
      constexpr int CPF = G::CPF(L);
      static_assert(CPF > 16, "a tile of 16 columns touches two frames at most");
      const int fr0 = (16 * rt) / CPF;              // wave-uniform: the frame of the tile's first column
      const int col = 16 * rt + n - fr0 * CPF;      // this lane's column, counted from frame fr0's first
      const bool nextf = col >= CPF;
      const int fbase = nextf ? (fr0 + 1) * G::kS : fr0 * G::kS;   // its frame's first pixel
      const int pb = fbase + (nextf ? col - CPF : col) * PHS;      // the column's first pixel (frame-aligned columns: Geo::CPF)
      // stores stop at the frame's stride (a last column's pixels past it are the next frame's first bins) and at the buffer's last row
      const int plim = fbase + G::kS < 16 * G::kTiles ? fbase + G::kS : 16 * G::kTiles;
      f32x4 sk = {0.f, 0.f, 0.f, 0.f};
      if constexpr (D.skip_from >= 0 && !(RCED_CHAIN_EXP & 2))
        sk = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                           scratch, tid * 16, (G::skip_unit_rem(D.skip_from) + jj) * kThreads * 16, 0));
      f32x4 racc[1][1] = {{rsh}};
      gemm_pass<1, 0, 1, KR, 0, RCED_CHAIN_DEPTH>(in, (pb - padl) * D.cinp + 2 * kq, 0, wr, lane, racc);
      f32x4 v = racc[0][0];
      if constexpr (D.skip_from >= 0) v += sk;      // module.py:30-31: before the ReLU
      v = relu4(v);
      // wave-uniform: does the tile hold a column with a gap pixel (from the column of bin 129 to the frame's last), or columns past the
      // tile's frames (they follow the last frame's)?
      const bool gap = 16 * rt + 15 >= fr0 * CPF + kF / PHS;
      float vv[4] = {v.x, v.y, v.z, v.w};
      if (gap) {
        asm volatile("" ::: "memory");   // a wave-uniform branch (see the main epilogue)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int i = 4 * kq + j, px = pb + i / R;
          if (!px_valid<N>(px)) vv[j] = 0.f;   // gap pixels hold zeros (the next layer's SAME padding)
        }
      }
      if constexpr (D.saves_skip && !(RCED_CHAIN_EXP & 1))
      {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{vv[0], vv[1], vv[2], vv[3]}), scratch, tid * 16,
                                               (G::skip_unit_rem(L) + jj) * kThreads * 16, 0);
        store_wait_state();
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = 4 * kq + j, px = pb + i / R;
        if (i < PHS * R && px < plim) out[px * D.coutp + 16 + i % R] = vv[j];
      }
    }
  }
}

template <class N, int L>
__device__ __forceinline__ void run_layers(const Params& P, float* lds, __amdgpu_buffer_rsrc_t scratch, int& wcur, XStage& xst,
                                           int tile, int wave, int lane, int tid, int utt, int t0, const LaneTab<N>& LT) {
  using G = Geo<N>;
  if constexpr (L < N::kLayers) {
    float* const wbase = lds + G::kWOff;
    // next packet: layer L+1, or layer 0 for the next tile (the stream wraps)
    constexpr int nxt = (L + 1 < N::kLayers) ? L + 1 : 0;
    float* const wdst = wbase + (wcur ^ 1) * G::kWRegion;
    auto dma = [&] { packet_dma<G::packet(nxt)>(P.wpack + G::packet_off(nxt), wdst, wave, lane); };   // issued inside the main pass
    if constexpr (L == N::kLayers - 1) xst = xstage_load<N>(P, tile + gridDim.x, tid);
    const float* w = wbase + wcur * G::kWRegion;
    if constexpr ((RCED_CHAIN_EXP & 24) == 24) {   // ... the same work from THREE copies of the code (what the instruction cache is worth)
      if (wave < 2) { asm volatile("; copy 0"); run_layer<N, L, 1>(P, lds, w, scratch, wave, lane, tid, utt, t0, dma, LT); }
      else if (wave < 4) { asm volatile("; copy 1"); run_layer<N, L, 1>(P, lds, w, scratch, wave, lane, tid, utt, t0, dma, LT); }
      else { asm volatile("; copy 2"); run_layer<N, L, 1>(P, lds, w, scratch, wave, lane, tid, utt, t0, dma, LT); }
    } else if constexpr ((RCED_CHAIN_EXP & 8) != 0) {   // timing experiment (wrong results): ONE copy of the layer's code for every wave
      run_layer<N, L, 1>(P, lds, w, scratch, wave, lane, tid, utt, t0, dma, LT);
    } else if constexpr (L > 0 && G::MT(L) == 2 && G::kSplitExtra) {
      if (wave < G::kExtra) run_layer<N, L, 1, 0>(P, lds, w, scratch, wave, lane, tid, utt, t0, dma, LT);
      else if (wave < 2 * G::kExtra) run_layer<N, L, 1, 1>(P, lds, w, scratch, wave, lane, tid, utt, t0, dma, LT);
      else run_layer<N, L, 0>(P, lds, w, scratch, wave, lane, tid, utt, t0, dma, LT);
    } else {
      if (wave < G::kExtra) run_layer<N, L, 1>(P, lds, w, scratch, wave, lane, tid, utt, t0, dma, LT);
      else run_layer<N, L, 0, -1>(P, lds, w, scratch, wave, lane, tid, utt, t0, dma, LT);
    }
    wcur ^= 1;
    if constexpr (N::layer[L].saves_skip || L == N::kLayers - 1) {
      if (!(RCED_CHAIN_EXP & 96)) __syncthreads();
    } else {
      layer_end_sync();
    }
    run_layers<N, L + 1>(P, lds, scratch, wcur, xst, tile, wave, lane, tid, utt, t0, LT);
  }
}

template <class N>
__global__ __launch_bounds__(kThreads) void fused_chain_kernel(Params P) {
  using G = Geo<N>;
  static_assert(G::kFitsLds, "LDS budget");
  static_assert(G::kX0Fits, "X0 fits in buffer Y");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int e = tid; e < G::kLdsFloats; e += kThreads) lds[e] = 0.f;
  __syncthreads();
  packet_dma<G::packet(0)>(P.wpack, lds + G::kWOff, wave, lane);
  int wcur = 0;
  XStage xst = xstage_load<N>(P, blockIdx.x, tid);
  const __amdgpu_buffer_rsrc_t scratch = __builtin_amdgcn_make_buffer_rsrc(
      P.scratch + (size_t)blockIdx.x * G::kScratchFloatsPerWg, 0, (int)(G::kScratchFloatsPerWg * 4), 0x00020000);
  const LaneTab<N> LT = make_lane_tab<N>(wave, lane);
  layer_end_sync();
  for (int tile = blockIdx.x; tile < P.total_tiles; tile += gridDim.x) {
    const int utt = tile / P.tiles_per_utt;
    const int t0 = (tile - utt * P.tiles_per_utt) * N::kTF;
    xstage_store<N>(xst, lds + G::kX0Off, tid);   // buffer Y is dead: its last reader finished before the last barrier
    __syncthreads();
    run_layers<N, 0>(P, lds, scratch, wcur, xst, tile, wave, lane, tid, utt, t0, LT);
  }
}

// ---------------------------------------------------------------------------------------------
// Final 1x129 layer (decode_5 / decode_8: CH -> 1, no BN, no ReLU) as a Toeplitz GEMM, see
// v3::final_gemm_kernel.  K = 129*CH = 8*NB64 + one b32 tail step.
// ---------------------------------------------------------------------------------------------
template <int CH>
struct FinalGeo {
  static constexpr int kK = kF * CH;
  static constexpr int kNB64 = kK / 8;
  static constexpr int kTailValid = kK - 8 * kNB64;   // 4 (CH = 12), 2 (CH = 10) or 0 (CH = 8: the training step's CR-CED output layer)
  static_assert(kTailValid >= 0 && kTailValid <= 4, "at most one b32 tail step");
  static constexpr int kMT = 9;
  static constexpr int kPack = kNB64 * kMT * 128 + (kTailValid ? kMT * 64 : 0);
};
constexpr int kFinFrames = 64;
constexpr int kFinThreads = 192;

template <int CH>
__global__ __launch_bounds__(kFinThreads) void final_gemm_kernel(const float* __restrict__ h,
                                                                  const float* __restrict__ apack, float bias,
                                                                  float* __restrict__ y, int frames) {
  using FG = FinalGeo<CH>;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, kq = lane >> 4;
  const int f0 = blockIdx.x * kFinFrames;
  const f32x2* ap = reinterpret_cast<const f32x2*>(apack) + (wave * 3) * 64 + lane;
  const float* at = apack + FG::kNB64 * FG::kMT * 128 + (wave * 3) * 64 + lane;
  const float* bp[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    int fr = f0 + 16 * t + n;
    if (fr >= frames) fr = frames - 1;   // clamp: computed, never stored
    bp[t] = h + (size_t)fr * FG::kK;
  }
  f32x4 acc[4][3];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int m = 0; m < 3; ++m) acc[t][m] = f32x4{bias, bias, bias, bias};
#pragma unroll 4
  for (int s = 0; s < FG::kNB64; ++s) {
    f32x2 a[3], b[4];
#pragma unroll
    for (int m = 0; m < 3; ++m) a[m] = ap[(s * FG::kMT + m) * 64];
#pragma unroll
    for (int t = 0; t < 4; ++t) b[t] = *reinterpret_cast<const f32x2*>(bp[t] + 8 * s + 2 * kq);
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int m = 0; m < 3; ++m) acc[t][m] = mfma(a[m][e], b[t][e], acc[t][m]);
  }
  {
    const int kqe = kq < FG::kTailValid ? kq : FG::kTailValid - 1;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float b = bp[t][8 * FG::kNB64 + kqe];
#pragma unroll
      for (int m = 0; m < 3; ++m) acc[t][m] = mfma(at[m * 64], b, acc[t][m]);
    }
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

// The same GEMM with the B operand staged through LDS (see v3::final_gemm_lds_kernel): the 64 frames' h rows arrive in
// chunks of 32 k as coalesced pieces, one chunk ahead in registers, into a ping-pong with frame stride 34 floats;
// A streams from L2 two steps ahead.  k >= K is staged as zero, so the b32 tail step needs no lane clamp.
template <int CH>
struct FinalLds {
  using FG = FinalGeo<CH>;
  static constexpr int kChunk = 32, kRow = kChunk + 2, kStepsPer = kChunk / 8;
  static constexpr int kPiece = FG::kK % 4 == 0 ? 4 : 2;              // floats per global load: rows are 16- or 8-byte aligned
  static constexpr int kPieces = kChunk / kPiece;
  static constexpr int kVec = kFinFrames * kPieces;
  static constexpr int kPer = (kVec + kFinThreads - 1) / kFinThreads;
  static constexpr int kTailChunk = (8 * FG::kNB64) / kChunk;          // chunk that holds the tail step's k
  static constexpr int kTailOff = 8 * FG::kNB64 - kTailChunk * kChunk; // its offset inside that chunk
  static constexpr int kChunks = kTailChunk + 1;
  static_assert(FG::kK % kPiece == 0 && kTailOff + 4 <= kChunk, "pieces end with the row; the tail step sits in one chunk");
};

template <int CH>
__global__ __launch_bounds__(kFinThreads) void final_gemm_lds_kernel(const float* __restrict__ h,
                                                                      const float* __restrict__ apack, float bias,
                                                                      float* __restrict__ y, int frames,
                                                                      const float* __restrict__ bias_dev = nullptr) {
  using FG = FinalGeo<CH>;
  using G = FinalLds<CH>;
  if (bias_dev) bias = *bias_dev;      // the training step's bias is a device variable (it moves every step)
  __shared__ __attribute__((aligned(16))) float bs[2][kFinFrames * G::kRow];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, kq = lane >> 4;
  const int f0 = blockIdx.x * kFinFrames;
  const f32x2* ap = reinterpret_cast<const f32x2*>(apack) + (wave * 3) * 64 + lane;
  const float* at = apack + FG::kNB64 * FG::kMT * 128 + (wave * 3) * 64 + lane;
  auto fetch = [&](int chunk, f32x4(&r)[G::kPer]) {
#pragma unroll
    for (int i = 0; i < G::kPer; ++i) {
      const int q = tid + i * kFinThreads;
      const int fr = f0 + q / G::kPieces, k = chunk * G::kChunk + G::kPiece * (q % G::kPieces);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (q < G::kVec && fr < frames && k < FG::kK) {
        const float* src = h + (size_t)fr * FG::kK + k;
        if constexpr (G::kPiece == 4) {
          v = *reinterpret_cast<const f32x4*>(src);
        } else {
          const f32x2 w = *reinterpret_cast<const f32x2*>(src);
          v.x = w.x;
          v.y = w.y;
        }
      }
      r[i] = v;
    }
  };
  auto commit = [&](int buf, const f32x4(&r)[G::kPer]) {
#pragma unroll
    for (int i = 0; i < G::kPer; ++i) {
      const int q = tid + i * kFinThreads;
      if (q < G::kVec) {
        float* d = bs[buf] + (q / G::kPieces) * G::kRow + G::kPiece * (q % G::kPieces);
        *reinterpret_cast<f32x2*>(d) = f32x2{r[i].x, r[i].y};
        if constexpr (G::kPiece == 4) *reinterpret_cast<f32x2*>(d + 2) = f32x2{r[i].z, r[i].w};
      }
    }
  };
  f32x4 acc[4][3];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int m = 0; m < 3; ++m) acc[t][m] = f32x4{bias, bias, bias, bias};
  f32x4 r[G::kPer];
  fetch(0, r);
  commit(0, r);
  f32x2 a[3], an[3], an2[3];            // A fragments of steps S, S+1, S+2
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    a[m] = ap[m * 64];
    an[m] = ap[(FG::kMT + m) * 64];
  }
  __syncthreads();
  for (int c = 0; c < G::kChunks; ++c) {
    if (c + 1 < G::kChunks) fetch(c + 1, r);
    const float* bb = bs[c & 1] + n * G::kRow;
    const int left = FG::kNB64 - G::kStepsPer * c;
    const int ns = left < G::kStepsPer ? left : G::kStepsPer;
#pragma unroll
    for (int s = 0; s < G::kStepsPer; ++s) {
      if (s < ns) {
        const int S = G::kStepsPer * c + s;
        if (S + 2 < FG::kNB64) {
#pragma unroll
          for (int m = 0; m < 3; ++m) an2[m] = ap[((S + 2) * FG::kMT + m) * 64];
        }
        f32x2 b[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) b[t] = *reinterpret_cast<const f32x2*>(bb + 16 * t * G::kRow + 8 * s + 2 * kq);
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int m = 0; m < 3; ++m) acc[t][m] = mfma(a[m][e], b[t][e], acc[t][m]);
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          a[m] = an[m];
          an[m] = an2[m];
        }
      }
    }
    if (FG::kTailValid > 0 && c == G::kTailChunk) {   // b32 step: lane kq <-> k = 8*NB64 + kq (zero past K)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float b = bb[16 * t * G::kRow + G::kTailOff + kq];
#pragma unroll
        for (int m = 0; m < 3; ++m) acc[t][m] = mfma(at[m * 64], b, acc[t][m]);
      }
    }
    if (c + 1 < G::kChunks) commit((c + 1) & 1, r);
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

}  // namespace chain
}  // namespace rced
