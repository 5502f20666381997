// R-CED V1 / V2 forward in bf16 (BASELINE config 2: "R-CED V2 forward, batch 64, 129x512, bf16"): ALL layers, the 1x129 output
// layer included, in ONE kernel on v_mfma_f32_16x16x32_bf16.  Round 6's rebuild of kernels_fused_chain16.h, whose 3-frame tiles
// crossed fifteen workgroup barriers with 4-17 half-rate MFMAs per tile and layer between them (0.107 of the bf16 peak for four
// rounds; this kernel: 0.225).  Reference: model_utils/model.py:6-61 over module.py:11-34 (conv -> BN -> +skip -> ReLU).
//
// A WAVE OWNS A FRAME.  Only the first conv (8 x k) looks along time; every later layer is 1 x k along frequency, so a frame runs
// through all the layers without ever reading another frame's activations.  A workgroup's W waves take W consecutive frames of one
// utterance, wave w frame t0 + w; the kernel is a template over W (frame16_kernel<N, W>, bit-identical results): W = 4, one wave
// per SIMD and two workgroups per CU, for calls of few tiles; W = 8, one workgroup per CU, for the others -- every packet then
// feeds eight frames, and the LDS that frees holds one skip (below).  No activation ever crosses a wave: the only thing the waves of
// a workgroup share is the weight stream, and the barriers -- one per GROUP of layers whose packets share a ring slot, 8 per tile
// for V2 -- exist for that alone (they meet waves that have done exactly the same work).
//
//   * Pixel space of a frame: bin f at row f + 8 of a 160-row image; rows 0..7 and 137..159 stay zero for the kernel's lifetime
//     (the SAME padding of every layer).  Nine 16-pixel tiles (the ninth holds bin 128 alone).
//   * Activations are bf16 PLANES  [octet of channels][row][8 channels]  = 16-byte rows, 2,560 bytes per plane (a multiple of
//     256: the lane groups of a ds_read_b128 -- {n 0-3, 12-15 of k-quad kq, n 4-11 of kq + 1} -- land on sixteen different
//     16-byte bank slots).  A layer works IN PLACE: every tile's accumulators are in registers before the rows its successors
//     read are overwritten (run_layer), so one 10-KB image per frame is all the LDS a frame's activations need.
//   * A conv is an implicit GEMM, cout on the M axis, pixels on N, K = (tap, octet) slots of 8 channels: lane (kq, n) of K-step s
//     reads slot j = 4 s + kq = (tap j / OCT, octet j % OCT) of pixel n's window -- ONE aligned ds_read_b128 out of the image, no
//     im2col copy.  (The old kernel's [pixel][channel] rows gave 8-byte-aligned 16-byte reads, which the LDS replays.)
//   * The first layer (8 x k on the 1-channel input) is the same code: the wave lays its eight input rows out as ONE plane
//     [row f + 8][8 time rows] (an im2col along time only), the input cast to bf16 ("C2 ... (cast bf16)"), weights
//     packed with the time row in the channel slot.
//   * Epilogue per fragment: two v_cvt_pk_bf16_f32, two v_pk_max_i16 (ReLU on the rounded value: the same result as rounding
//     the ReLU), one ds_write_b64; it rides between the MFMAs of the next group of tiles.
//   * Skips (module.py:30-31: decoder layer += encoder output BEFORE the ReLU; 72 / 114 channels): the encoder layer's packed
//     bf16 fragment -- lane (kq, n): channels 4 kq .. + 3 of pixel n -- is exactly the B operand of a v_mfma_f32_16x16x16_bf16
//     whose k is the channel, so the decoder adds it with ONE MFMA against an identity A fragment: no unpacking, no VALU.  The
//     one-M-tile encoder layers' fragments wait in REGISTERS (Res); the two-M-tile ones' in a per-wave global scratch (compact:
//     8 bytes per lane whose four channels exist), except -- W = 8 -- M-tile 0 of the first of them, which waits in LDS.
//   * Weights: per layer a packet of 1-KiB A fragments [step][M-tile][lane] x 8 bf16 (+ 32 fp32 shifts per layer, resident),
//     LDS-DMA'd a group ahead into a two-slot ring; a layer reads its fragments ONCE, into registers.
//   * The output layer (run_final) is a Toeplitz GEMM over tap tables in its packet and images the last hidden layer writes.
// Precision contract: tests/test_forward_gpu.py against the test suite's bf16 emulation, which rounds at the same places (input,
// every folded kernel, every layer's output), and tests/tools/fuzz_bf16.py.  NOT within the fp32 path's 1e-4 bar: opt-in
// (option "bf16").
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_fused_chain.h"
#include "lds_dma.h"

#ifndef RCED_F16_EXP
#define RCED_F16_EXP 0   // timing experiments only (wrong results): 1 = no skip stores, 2 = no skip loads / adds (128: loads issued, data dropped at the layer's end), 4 = no barriers,
                         // 8 = A fragments read for the first K-step only, 16 = no input-row loads, 64 = no packet DMA
#endif
#if RCED_F16_EXP != 0 && !defined(RCED_TIMING_ONLY)
#error "RCED_F16_EXP builds compute wrong results: timing experiments only (-DRCED_TIMING_ONLY)"
#endif

#ifndef RCED_F16_AREG
#define RCED_F16_AREG 18  // A fragments of a layer kept in registers when it has at most this many (0 = never)
#endif
#ifndef RCED_F16_DBGEXPOSE
#define RCED_F16_DBGEXPOSE 0   // debugging: bit L = layer L keeps all three groups' accumulators and runs its whole epilogue behind the K loop
#endif
#ifndef RCED_F16_GROUPS
#define RCED_F16_GROUPS 1   // small layers' packets travel together, no barrier between them (0: one packet, one barrier per layer)
#endif
#ifndef RCED_F16_SKIP_LDS
#define RCED_F16_SKIP_LDS 1   // eight-wave form only: one skip fragment set in LDS (Geo::kSkipLdsLayer)
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

// Frames (= waves) per workgroup, a template parameter W of everything below: 4 (one wave per SIMD, two workgroups per CU: the form
// for calls of few tiles) or 8 (one workgroup per CU: every packet feeds eight frames, and the LDS that frees holds a skip: Geo)
constexpr int kRowPad = 8;                // bin f lives at row f + kRowPad
constexpr int kPlanes = 4;                // 32 channels at most (V1's 24 -> 32 layer, V2's 23 -> 25)
constexpr int kTiles = 9;                 // 16-pixel tiles per frame

struct Params {
  const float* x;            // [N, T, 129]
  float* y;                  // [N, T, 129]: the masks
  float fin_bias;            // the output layer's bias
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

template <class N, int W = 4>
struct Geo {
  static_assert(W == 4 || W == 8, "four or eight frames per workgroup");
  static constexpr int kWaves = W, kThreads = W * 64;
  static constexpr int kLayers = N::kLayers;
  static constexpr int oct_out(int l) { return (N::layer[l].cout + 7) / 8; }
  static constexpr int oct_in(int l) { return l == 0 ? 1 : oct_out(l - 1); }
  static constexpr int MT(int l) { return (N::layer[l].cout + 15) / 16; }
  static constexpr int slots(int l) { return N::layer[l].taps * oct_in(l); }   // K slots of 8 (layer 0: 8 time rows per tap)
  static constexpr int steps(int l) { return (slots(l) + 3) / 4; }
  static constexpr int frags(int l) { return steps(l) * MT(l); }
  // Packet kLayers is the output layer's (1 x 129, CH -> 1; run_final reads its Toeplitz A operand straight out of it).  Octet 0
  // (channels 0 .. 7) has a K slot per window position; of octet 1 only CB channels exist (V2: 2, V1: 4), so its slots hold RB =
  // 8 / CB consecutive window positions x CB channels: 36 + 36 / RB K-steps per frame instead of 72.  The packet:
  //   TA[4 copies kq][160 rows][8 channels]: row r of copy kq = tap r + kq - 15 (zero outside taps 0 .. 128) -- the four k-quads
  //     of a read land 2,560 bytes apart (the same bank slots: the lane groups of a ds_read_b128 stay conflict-free);
  //   TB[RB copies c][k][RB positions j][CB channels]: tap RB k + c + j - 15 of channels 8 .. 8 + CB - 1 -- copy c serves the
  //     lanes whose bin phase leaves the remainder c, so that every read is 16-byte aligned.
  static constexpr int kFinCB = N::kFinalCh <= 10 ? 2 : 4;
  static constexpr int kFinRB = 8 / kFinCB;
  static constexpr int kFinStepsA = 36, kFinStepsB = 36 / kFinRB;
  static constexpr int kTARows = 160, kTACopy = kTARows * 16;
  static constexpr int kTBOff = 4 * kTACopy;
  static constexpr int kTBRows = 4 * (kFinStepsB - 1) + 3 + 15 / kFinRB + 1;     // k values a read reaches
  static constexpr int kTBCopy = kFinRB == 4 ? 704 : 1408;                        // 12 / 8 bank slots (mod 16) from copy to copy
  static constexpr int kFinPacket = 13 * 1024;
  static_assert(N::kFinalCh > 8 && N::kFinalCh <= 12 && kTBRows * 16 <= kTBCopy && kTBOff + kFinRB * kTBCopy <= kFinPacket, "tap tables fit");
  static constexpr int packet_bytes(int l) { return l == kLayers ? kFinPacket : frags(l) * 1024; }
  static constexpr int packet_off(int l) {
    int o = 0;
    for (int i = 0; i < l; ++i) o += packet_bytes(i);
    return o;
  }
  static constexpr int kShiftOff = packet_off(kLayers + 1);     // 32 fp32 shifts per layer behind the packets
  static constexpr int kShiftBytes = kLayers * 128;
  static constexpr int kWBytes = kShiftOff + kShiftBytes;
  static constexpr int maxpacket() {
    int m = 0;
    for (int l = 0; l <= kLayers; ++l) m = packet_bytes(l) > m ? packet_bytes(l) : m;
    return m;
  }
  static constexpr int kWRegion = maxpacket();
  // Consecutive layers whose packets fit one ring slot TOGETHER travel as one transfer and run back to back: the barrier at a
  // layer's end is the weight ring's alone (who has read the slot the next transfer overwrites, whose pieces have landed), so inside
  // a group there is none.  Greedy from layer 0 -- V2: {0..3} {4,5} {6} {7} {8} {9,10} {11..14} {output layer}: 8 barriers per tile
  // instead of 16.
  static constexpr int group_first(int l) {
    int f = 0, bytes = 0;
    for (int i = 0; i <= l; ++i) {
      if (i > f && (bytes + packet_bytes(i) > kWRegion || !RCED_F16_GROUPS)) {
        f = i;
        bytes = 0;
      }
      bytes += packet_bytes(i);
    }
    return f;
  }
  static constexpr int group_last(int l) {
    int j = l;
    while (j + 1 <= kLayers && group_first(j + 1) == group_first(l)) ++j;
    return j;
  }
  static constexpr int group_bytes(int f) { return packet_off(group_last(f) + 1) - packet_off(f); }
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
  // Eight-wave workgroups leave LDS free: M-tile 0 of the FIRST two-M-tile encoder layer's skip -- the longest-lived one, the one
  // that never survives in the L2 -- waits there, 9 tiles x 512 bytes per wave, instead of in the global scratch
  static constexpr int skip_lds_layer() {
    if (W != 8 || !RCED_F16_SKIP_LDS) return -1;
    for (int l = 0; l < kLayers; ++l)
      if (N::layer[l].saves_skip && MT(l) == 2) return l;   // (a two-M-tile layer is never register-resident: Res::slot)
    return -1;
  }
  static constexpr int kSkipLdsLayer = skip_lds_layer();
  static constexpr int kSkipLdsBytes = kSkipLdsLayer >= 0 ? kTiles * 512 : 0;
  static constexpr int kSkipLOff = kSOff + kShiftBytes;
  static constexpr int kLdsBytes = kSkipLOff + kWaves * kSkipLdsBytes;
  static_assert(kLdsBytes <= (kWaves == 4 ? 80 : 160) * 1024, "two workgroups per CU (four waves each), or one of eight");
  static constexpr bool pads_ok() {
    for (int l = 0; l < kLayers; ++l)
      if (pad(l) > kRowPad) return false;
    return true;
  }
  static_assert(pads_ok(), "the widest kernel's left halo fits the leading zero rows");
  // The last fused layer writes its output for the output layer as two images (it reads planes 0, 1 only):
  //   H over plane 2: octet 0, 16-byte rows, bin f at row f + 2 (f >> 4) -- two pad rows per 16 bins, so that the nine 16-bin
  //     blocks a ds_read_b128 of run_final touches start 288 bytes apart (with the k-quads 16 bytes apart: sixteen different
  //     bank slots per lane group); block 8's rows behind bin 128 are the plane's own zero rows;
  //   H2 over plane 3's bin rows: channels 8 .. 8 + CB - 1, 2 CB bytes per bin, blocks kH2Stride apart (6 / 10 bank slots);
  //   out-of-range window positions read plane 3's trailing zero rows (kZOff), which nothing ever writes.
  static constexpr int kHOff = 2 * kPlane;
  static constexpr int kHStride = 18 * 16;
  static constexpr int kH2Off = 3 * kPlane + kRowPad * 16;
  static constexpr int kH2Stride = kFinRB == 4 ? 96 : 160;
  static constexpr int kH2Block = 16 * 2 * kFinCB;    // bytes of a block's sixteen bins
  static constexpr int kZOff = 3 * kPlane + (kRowPad + kF) * 16;
  static_assert(8 * kHStride + 16 * 16 <= kPlane && kH2Off + 8 * kH2Stride + kH2Block <= kZOff && kZOff + 64 <= kRegion &&
                    oct_in(kLayers - 1) <= 2 && MT(kLayers - 1) == 1,
                "H, H2 and the zero rows fit planes 2, 3; the last layer reads planes 0, 1 only");
  // skip scratch, per wave: per (saving layer, M-tile) and group of three tiles a 1-KiB unit -- the group's first two fragments,
  // 16 bytes per lane -- and a 512-byte one for the third.  Only the lanes whose four channels exist are stored and loaded
  // (k-quads 0 .. quads - 1: whole 256-byte runs), so an M-tile with 3 real channels moves a quarter of its unit.
  static constexpr int kSkipSet = 3 * (1024 + 512);
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
template <int BYTES, int kWaves>
__device__ __forceinline__ void packet_dma(const unsigned* __restrict__ src, char* dst, int wave, int lane) {
  static_assert(BYTES % 1024 == 0, "whole pieces");
  constexpr int chunks = BYTES / 1024;
  const unsigned d0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)dst;
#pragma unroll
  for (int i = 0; i < (chunks + kWaves - 1) / kWaves; ++i) {
    const int c = wave + i * kWaves;
    if (c < chunks) {
      const unsigned m0v = __builtin_amdgcn_readfirstlane(d0 + c * 1024);
      const unsigned long long sa = (unsigned long long)(size_t)(src + c * 256);
      // (the builtin returns a SIGNED int: widened as it stands, a low half with bit 31 set turns the pointer into 0xffffffff........)
      const unsigned long long sp = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)sa) |
                                    ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(sa >> 32)) << 32);
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

// The eight input rows t - 3 .. t + 4 of one frame, fetched one tile ahead: bins 4 l .. 4 l + 3 of every row in lane l <= 32, ONE
// 16-byte load per row (a vector-memory instruction costs its wave ~100 cycles of issue in this kernel whatever it moves: eight
// of them, not twenty-four four-byte ones -- 4 % of the kernel's time)
struct XRows {
  f32x4 v[8];
};
// Buffer loads over the utterance's [T, 129] floats (a row is 516 bytes: dword-aligned, which is all a buffer load asks for).  A row
// in front of the first or behind the last frame (TF 'SAME' for the 8-tall kernel: 3 rows before, 4 after) is skipped by a
// wave-uniform branch and stays zero: offsets are never negative (a "negative" offset is a huge unsigned one to the range check
// but, with a positive immediate folded in behind it, an address 4 GB away to the address unit -- measured: memory faults).  Lanes
// 33 .. 63 carry an offset past the descriptor's range (zero, no traffic); lane 32 fetches bins 125 .. 128 -- its sixteen bytes end
// with the row, so the utterance's last row is in range whatever the range check does with a partly covered access.
template <int kWaves>
__device__ __forceinline__ XRows x_load(const Params& P, int tile, int wave, int lane) {
  XRows r;
  const bool live = tile < P.total_tiles;
  const int utt = live ? tile / P.tiles_per_utt : 0;
  const int t = live ? (tile - utt * P.tiles_per_utt) * kWaves + wave : 0;
  const __amdgpu_buffer_rsrc_t xu = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(P.x) + (size_t)utt * P.T * kF, 0, P.T * kF * 4, 0x00020000);
  const int vo = lane < kF / 4 ? lane * 16 : lane == kF / 4 ? (kF - 4) * 4 : 0x40000000;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int tt = t + k - 3;
    r.v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (live && tt >= 0 && tt < P.T)
      r.v[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xu, vo, tt * (kF * 4), 0));
  }
  return r;
}
__device__ __forceinline__ void x_store(const XRows& r, char* region, int lane) {   // plane 0 (its stride does not matter)
  if (lane <= kF / 4) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = (lane < kF / 4 ? 4 * lane : kF - 4) + i;
      const u32x4 q = {pack2(r.v[0][i], r.v[1][i]), pack2(r.v[2][i], r.v[3][i]), pack2(r.v[4][i], r.v[5][i]), pack2(r.v[6][i], r.v[7][i])};
      if (i == 3 || lane < kF / 4) *reinterpret_cast<u32x4*>(region + (p + kRowPad) * 16) = q;
    }
  }
}

// Skip fragments that never leave the register file: the one-M-tile encoder layers' (V2: encode_1..4, V1: encode_1..2),
// two registers per tile, 72 / 36 in all.  The other saving layers' fragments go through the global scratch.
#ifndef RCED_F16_RESMASK
#define RCED_F16_RESMASK 0xffff
#endif
#ifndef RCED_F16_MAXRES
#define RCED_F16_MAXRES 4
#endif
constexpr int kMaxRes = RCED_F16_MAXRES;
template <class N>
struct Res {
  static constexpr int slot(int l) {   // register slot of saving layer l, or -1
    if (l < 0 || !N::layer[l].saves_skip || Geo<N>::MT(l) != 1 || !((RCED_F16_RESMASK >> l) & 1)) return -1;
    int k = 0;
    for (int i = 0; i < l; ++i)
      if (N::layer[i].saves_skip && Geo<N>::MT(i) == 1 && ((RCED_F16_RESMASK >> i) & 1)) ++k;
    return k < kMaxRes ? k : -1;
  }
  static constexpr int count() {
    int k = 0;
    for (int i = 0; i < N::kLayers; ++i)
      if (slot(i) >= 0) ++k;
    return k;
  }
  u32x2 v[kMaxRes > 0 ? kMaxRes : 1][kTiles];
};

// One layer of one frame (one wave).  `w` = the layer's packet in LDS; `pre` = issued once the first operand reads are in
// flight (the next packet's LDS-DMA, the next tile's input rows).
//
// The nine tiles run as THREE GROUPS of three through one software pipeline of 3 x STEPS slots (operands of slot i + 1 are read
// while the MFMAs of slot i issue; a group's accumulators are the registers of the group before last).  The epilogue of group
// g -- round, ReLU, LDS store, skip store: 4 VALU + 1-2 stores per fragment -- rides between the MFMAs of group g + 1, one
// fragment per two MFMAs; only the last group's is exposed.  In place: group g + 1 reads pixels 48 (g + 1) - 6 and up, so group
// g's first two tiles (pixels 48 g .. 48 g + 31) may be stored while it does; its third tile waits for group g + 1's last slot,
// whose reads were issued a slot earlier (the LDS serves a wave's requests in order).
template <class N, int W, int L, class Pre>
__device__ __forceinline__ void run_layer(const Params& P, char* region, const char* w, const char* shifts, __amdgpu_buffer_rsrc_t scratch, char* skl, int lane,
                                          Pre pre, Res<N>& res, bool stamp = false) {
  using G = Geo<N, W>;
  F16_STAMP(stamp, 4 * L + 0);
  constexpr LayerDesc D = N::layer[L];
  constexpr int OCT = G::oct_in(L), MT = G::MT(L), STEPS = G::steps(L), PADL = G::pad(L);
  constexpr bool kLast = (L == N::kLayers - 1);
  constexpr int NB = OCT < STEPS ? OCT : STEPS;     // per-lane window bases: slot j + 4 OCT is the same octet four taps on
  constexpr int GT = 3, NG = kTiles / GT;           // tiles per group, groups
  constexpr int SF = D.skip_from >= 0 ? D.skip_from : 0;
  constexpr int kResIn = D.skip_from >= 0 ? Res<N>::slot(SF) : -1;     // the skip comes out of registers
  constexpr int kResOut = D.saves_skip ? Res<N>::slot(L) : -1;         // ... goes into registers
  constexpr bool kSkipMem = D.skip_from >= 0 && kResIn < 0 && !(RCED_F16_EXP & 2);
  constexpr bool kSaveMem = D.saves_skip && kResOut < 0 && !(RCED_F16_EXP & 1);
  constexpr bool kSaveLds = kSaveMem && L == G::kSkipLdsLayer;             // ... its M-tile 0 goes to LDS
  constexpr bool kSkipLds = kSkipMem && SF == G::kSkipLdsLayer && G::kSkipLdsLayer >= 0;
  constexpr int kMemMT = MT - (kSaveLds ? 1 : 0);                          // M-tiles whose skip fragments go to the global scratch
  static_assert(STEPS >= 2 && kTiles == GT * NG, "the third tile of a group is stored in the next group's last slot");
  constexpr int kPer = (2 * MT + STEPS - 2) / (STEPS - 1);   // fragments of the previous group per slot (slots 0 .. STEPS - 2)
  static_assert(kPer <= (GT * MT + 1) / 2 && MT <= (GT * MT + 1) / 2, "one fragment behind every second MFMA of a slot");
  asm volatile("" : "+v"(lane));                    // no hoisting of every layer's address arithmetic out of the tile loop
  const int n = lane & 15, kq = lane >> 4;

  // skip fragments of the matching encoder layer that wait in the scratch: issued now, used after their group's last K-step.
  // A lane whose four channels do not exist gets an offset past the descriptor's range: its load returns zero, its store is
  // dropped -- no lane mask, no branch.
  constexpr int kOob = 0x40000000;
  u32x2 skip[kSkipMem ? kTiles : 1][kSkipMem ? MT : 1];
  if constexpr (kSkipMem) {
    static_assert(N::layer[SF].cout == D.cout, "skip shapes match");
#pragma unroll
    for (int mt = kSkipLds ? 1 : 0; mt < MT; ++mt) {
      const int so = G::skip_off(SF, mt);
      const bool real = kq < G::skip_quads(SF, mt);
      const int o16 = real ? lane * 16 : kOob, o8 = real ? lane * 8 : kOob;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const u32x4 q = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(scratch, o16, so + g * 1536, 0));
        skip[GT * g][mt] = u32x2{q.x, q.y};
        skip[GT * g + 1][mt] = u32x2{q.z, q.w};
        skip[GT * g + 2][mt] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(scratch, o8, so + g * 1536 + 1024, 0));
      }
    }
  }

  if constexpr (kSkipLds) {
#pragma unroll
    for (int t = 0; t < kTiles; ++t) skip[t][0] = *reinterpret_cast<const u32x2*>(skl + t * 512 + lane * 8);
  }
  if constexpr (kLast) {   // plane 3 holds an earlier layer's activations: H2's block 8 is zero behind bin 128
    if (lane < G::kH2Block / 16) *reinterpret_cast<u32x4*>(region + G::kH2Off + 8 * G::kH2Stride + lane * 16) = u32x4{0u, 0u, 0u, 0u};
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
  f32x4 sh[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) sh[mt] = *reinterpret_cast<const f32x4*>(shifts + (32 * L + 16 * mt + 4 * kq) * 4);
  // identity A fragment of the K = 16 instruction: A[m][4 kq + i] = (m == 4 kq + i)
  const int dg = n - 4 * kq;
  const u32x2 eye = {dg == 0 ? 0x3F80u : dg == 1 ? 0x3F800000u : 0u, dg == 2 ? 0x3F80u : dg == 3 ? 0x3F800000u : 0u};

  // ---- one fragment's epilogue (tile t, M-tile mt): the layer's output IS the rounded value.  It goes to octet 2 mt + (kq >> 1),
  // bytes 8 (kq & 1) .. + 7 of pixel 16 t + n's row; an M-tile whose upper octet lies past the layer's last one stores it all the
  // same (zeros: those rows of the packet are zero) -- the plane is dead, and an unconditional store is one instruction.
  char* const out = region + (n + kRowPad) * 16 + (kq >> 1) * G::kPlane + (kq & 1) * 8;
  int so16[MT], so8[MT];
  if constexpr (kSaveMem) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const bool real = kq < G::skip_quads(L, mt);
      so16[mt] = real ? lane * 16 : kOob;
      so8[mt] = real ? lane * 8 : kOob;
    }
  }
  u32x2 pair[MT];                                   // a group's first tile, until its second completes the 16-byte skip store
  auto fragment = [&](const f32x4 v, int t, int mt) {
    const u32x2 hq = {relu2(pack2(v.x, v.y)), relu2(pack2(v.z, v.w))};
    if constexpr (kResOut >= 0) res.v[kResOut >= 0 ? kResOut : 0][t] = hq;
    if constexpr (!kLast) {
      if (t < kTiles - 1) *reinterpret_cast<u32x2*>(out + 2 * mt * G::kPlane + t * 256) = hq;
      else if (n == 0) *reinterpret_cast<u32x2*>(out + 2 * mt * G::kPlane + t * 256) = hq;   // tile 8: bin 128 alone, the other rows stay zero
    } else {
      // the output layer's images (Geo::kHOff, kH2Off): k-quads 0, 1 hold octet 0 of bin 16 t + n, k-quad 2 channels 8 .. 11
      if (t < kTiles - 1 || n == 0) {
        if (kq < 2) *reinterpret_cast<u32x2*>(region + G::kHOff + t * G::kHStride + n * 16 + kq * 8) = hq;
        else if (kq == 2) {
          char* const h2 = region + G::kH2Off + t * G::kH2Stride + n * (2 * G::kFinCB);
          if constexpr (G::kFinCB == 2) *reinterpret_cast<unsigned*>(h2) = hq.x;
          else *reinterpret_cast<u32x2*>(h2) = hq;
        }
      }
    }
    if (kSaveLds && mt == 0) {
      *reinterpret_cast<u32x2*>(skl + t * 512 + lane * 8) = hq;
    } else if constexpr (kSaveMem) {
      const int g = t / GT, j = t % GT, so = G::skip_off(L, mt) + g * 1536;
      if (j == 0) pair[mt] = hq;
      if (j == 1) __builtin_amdgcn_raw_buffer_store_b128(u32x4{pair[mt].x, pair[mt].y, hq.x, hq.y}, scratch, so16[mt], so, 0);
      if (j == 2) __builtin_amdgcn_raw_buffer_store_b64(hq, scratch, so8[mt], so + 1024, 0);
      if (j > 0) store_wait_state();
    }
  };

  constexpr bool kExpose = (RCED_F16_DBGEXPOSE >> L) & 1;
  constexpr int AM = kExpose ? 3 : 1;               // accumulator sets: g & AM
  f32x4 acc[kExpose ? NG : 2][GT][MT];
  // A fragments: a layer of at most RCED_F16_AREG fragments reads them ONCE, into registers (the three groups would read them
  // three times: the LDS is this kernel's busiest unit); a larger one reads its step's with the step's B fragments
  constexpr bool kAReg = G::frags(L) <= RCED_F16_AREG;
  u32x4 areg[kAReg ? STEPS : 1][kAReg ? MT : 1];
  if constexpr (kAReg) {
#pragma unroll
    for (int s = 0; s < STEPS; ++s)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) areg[s][mt] = *reinterpret_cast<const u32x4*>(wp + (s * MT + mt) * 1024);
  }
  u32x4 a[2][kAReg ? 1 : MT], b[2][GT];
  auto load = [&](int slot, int buf) {
    const int g = slot / STEPS, s = slot % STEPS;
    if constexpr (!kAReg) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a[buf][mt] = *reinterpret_cast<const u32x4*>(wp + (s * MT + mt) * 1024);
    }
#pragma unroll
    for (int j = 0; j < GT; ++j)
      b[buf][j] = *reinterpret_cast<const u32x4*>(region + base[s % OCT % NB] + (s / OCT) * 64 + (GT * g + j) * 256);
  };
  load(0, 0);
  pin();
  pre();
  pin();
#pragma unroll
  for (int g = 0; g < NG; ++g) {
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      const int slot = g * STEPS + s, buf = slot & 1;
      if (slot + 1 < NG * STEPS) load(slot + 1, buf ^ 1);
      pin();
      // this slot's share of the previous group's epilogue: tiles 0 and 1 of that group spread over slots 0 .. STEPS - 2, tile 2
      // in the last slot
      int q = 0;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < GT; ++j) {
          acc[g & AM][j][mt] = mfma32(kAReg ? areg[kAReg ? s : 0][kAReg ? mt : 0] : a[buf][kAReg ? 0 : mt], b[buf][j], s == 0 ? sh[mt] : acc[g & AM][j][mt]);
          if (g > 0 && (q & 1) == 0 && !kExpose) {
            // behind every second MFMA one fragment of the previous group: pieces p = jj * MT + mm (tile jj < 2 of that group, in
            // this order: a skip store pairs a group's first tile with its second), kPer per slot; tile 2 in the last slot
            const int k = q / 2;
            if (s < STEPS - 1) {
              const int p = s * kPer + k;
              if (k < kPer && p < 2 * MT) {
                pin();
                fragment(acc[(g - 1) & 1][p / MT][p % MT], GT * (g - 1) + p / MT, p % MT);
                pin();
              }
            } else if (k < MT) {
              pin();
              fragment(acc[(g - 1) & 1][2][k], GT * (g - 1) + 2, k);
              pin();
            }
          }
          ++q;
        }
      pin();
    }
    if constexpr (D.skip_from >= 0 && !(RCED_F16_EXP & (2 | 128))) {
      // HAZARD (found the hard way, round 6): a v_mfma_f32_16x16x16_bf16 issued DIRECTLY behind the v_mfma_f32_16x16x32_bf16
      // that wrote its srcC -- same registers as vdst, the ordinary accumulation chain, but two opcodes of different pass
      // counts -- read a stale accumulator on this part (one tile's skip landed on the previous K-step's sum; deterministic),
      // and hipcc pads no wait state between them.  Left to itself the scheduler does produce that pair (it reorders these
      // three to six MFMAs freely), so their order is pinned to the order of the last K-step's: every accumulator is two
      // MFMAs old at least when its skip arrives.  tools/isa_lint.py scans every build for the pair.
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < GT; ++j) {
          pin();
          if constexpr (kResIn >= 0) acc[g & AM][j][mt] = mfma16(eye, res.v[kResIn >= 0 ? kResIn : 0][GT * g + j], acc[g & AM][j][mt]);
          else acc[g & AM][j][mt] = mfma16(eye, skip[kSkipMem ? GT * g + j : 0][kSkipMem ? mt : 0], acc[g & AM][j][mt]);
        }
      pin();
    }
  }
  if constexpr (kSkipMem && (RCED_F16_EXP & 128)) {   // timing only: the skip loads issued, their data waited for here and dropped
#pragma unroll
    for (int t = 0; t < kTiles; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) asm volatile("" ::"v"(skip[t][mt]));
  }
  // The next packet's LDS-DMA was issued in front of every global store of this layer; the vector-memory counter retires in
  // order, so waiting until only the stores issued since (the first two groups' skip fragments: exactly 4 per M-tile -- their
  // masked lanes are out-of-range offsets, not skipped instructions) are outstanding means the packet has landed.  The stores
  // themselves stay in flight across the barrier: nobody reads them before a later layer's wait at this place has retired them.
  F16_STAMP(stamp, 4 * L + 1);
  if constexpr (kSaveMem && !kExpose) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * kMemMT) : "memory");
  else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0f70);              // ... and hipcc's wait-count pass knows it (vmcnt(0), nothing else)
  }
  F16_STAMP(stamp, 4 * L + 2);
  if constexpr (kExpose) {
#pragma unroll
    for (int g = 0; g + 1 < NG; ++g)
#pragma unroll
      for (int j = 0; j < GT; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) fragment(acc[g][j][mt], GT * g + j, mt);
  }
  // ---- the last group's epilogue
#pragma unroll
  for (int j = 0; j < GT; ++j)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) fragment(acc[(NG - 1) & AM][j][mt], GT * (NG - 1) + j, mt);
  F16_STAMP(stamp, 4 * L + 3);
}

// The 1 x 129 output layer (decode_5 / decode_8: CH -> 1, no BatchNorm, no ReLU; model.py:24,55) of one frame, inside the kernel:
// a GEMM with the 16 bin phases m on the M axis, the nine 16-bin blocks n of the frame on N and K = (window position u < 144,
// channel):  y[16 n + m] = sum_u sum_c W[u - m][c] H[16 n + u - 64][c].  Octet 0: lane (kq, .) of K-step s holds position u =
// 4 s + kq (36 steps); channels 8 ..: lane (kq, .) of step s' holds positions RB (4 s' + kq) .. + RB - 1 x CB channels (36 / RB
// steps; Geo's packet comment).  The A fragment is 16 bytes of a tap table (out-of-range taps are its zero rows), the B fragment 16
// bytes of H / H2 at bin f' = 16 n + u - 64 -- for the steps of a 16-bin block q one per-lane base + immediates (f' >> 4 = n - 4 + q
// is the same for all of them), out-of-range blocks read zero rows.  45 MFMAs per frame (V2; nine of sixteen columns used) against
// 1,026 for the layers in front of it; no hand-off tensor in HBM, no second launch.
template <class N, int W, class Pre>
__device__ __forceinline__ void run_final(const Params& P, const char* region, const char* tt, int lane, long long yrow, Pre pre) {
  using G = Geo<N, W>;
  asm volatile("" : "+v"(lane));
  const int n = lane & 15, kq = lane >> 4;
  constexpr int RB = G::kFinRB, NA = G::kFinStepsA, NT = NA + G::kFinStepsB;
  int ta = kq * G::kTACopy + (15 - n) * 16;
  int tb = G::kTBOff + ((15 - n) % RB) * G::kTBCopy + (kq + (15 - n) / RB) * 16;
  asm volatile("" : "+v"(ta), "+v"(tb));
  const char* const ap = tt + ta;
  const char* const bp = tt + tb;
  f32x4 acc[2] = {f32x4{P.fin_bias, P.fin_bias, P.fin_bias, P.fin_bias}, f32x4{0.f, 0.f, 0.f, 0.f}};
  constexpr int SS = NT % 6 == 0 ? 6 : 5, NS = NT / SS;   // steps per slot (<= 12 reads in flight: the LDS counter holds 15), slots
  static_assert(NT % SS == 0, "whole slots");
  u32x4 a[2][SS], b[2][SS];
  auto load = [&](int slot, int buf) {
#pragma unroll
    for (int e = 0; e < SS; ++e) {
      const int s = SS * slot + e;
      const bool first = s < NA;                      // octet 0 / the packed channels
      const int t = first ? s : s - NA, spb = first ? 4 : 4 / RB;   // steps per 16-bin block
      const int q = t / spb, i = t % spb;
      const int blk = n - 4 + q;
      const bool ok = n <= 8 && blk >= 0 && blk <= 8;
      const int off = (ok ? (first ? G::kHOff + blk * G::kHStride : G::kH2Off + blk * G::kH2Stride) : G::kZOff) + kq * 16;
      a[buf][e] = *reinterpret_cast<const u32x4*>((first ? ap : bp) + t * 64);
      b[buf][e] = *reinterpret_cast<const u32x4*>(region + off + i * 64);
    }
  };
  load(0, 0);
  pin();
  pre();
  pin();
#pragma unroll
  for (int slot = 0; slot < NS; ++slot) {
    if (slot + 1 < NS) load(slot + 1, (slot + 1) & 1);
    pin();
#pragma unroll
    for (int e = 0; e < SS; ++e) acc[e & 1] = mfma32(a[slot & 1][e], b[slot & 1][e], acc[e & 1]);
    pin();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next tile's first packet has landed (see run_layer); the mask stores follow
  __builtin_amdgcn_s_waitcnt(0x0f70);
  // H lay over plane 2, zero rows included (8 in front of the bins, 23 behind: every layer's SAME padding): put the zeros back --
  // 31 rows of 16 bytes (V2), one store (the LDS serves this wave's reads above first)
  {
    constexpr int PR = kRowPad + G::kRows - (kRowPad + kF);   // zero rows of a plane
    static_assert(PR <= 64, "one store");
    if (lane < PR) *reinterpret_cast<u32x4*>(const_cast<char*>(region) + 2 * G::kPlane + (lane < kRowPad ? lane : kF + lane) * 16) = u32x4{0u, 0u, 0u, 0u};
  }
  const f32x4 v = acc[0] + acc[1];
  if (yrow >= 0) {
    float* yp = P.y + yrow + 16 * n + 4 * kq;
    if (n < 8) *reinterpret_cast<f32x4*>(yp) = v;
    else if (n == 8 && kq == 0) *yp = v.x;
  }
}

// A bare s_barrier (not __syncthreads(), whose release fence waits for every global store in flight): what the four waves hand
// each other is the weight ring alone -- this wave's pieces of the next packet have landed (run_layer's wait in front of its
// epilogue), and it has read the last fragment of the packet that the next layer's DMA will overwrite.
__device__ __forceinline__ void layer_end_sync() {
  if (!(RCED_F16_EXP & 4)) asm volatile("s_barrier" ::: "memory");
}

template <class N, int W, int L>
__device__ __forceinline__ void run_layers(const Params& P, char* lds, char* region, __amdgpu_buffer_rsrc_t scratch, int& wcur,
                                           XRows& xr, Res<N>& res, int tile, int wave, int lane, long long yrow, bool stamp) {
  using G = Geo<N, W>;
  if constexpr (L <= N::kLayers) {
    constexpr int gf = G::group_first(L), gl = G::group_last(L);
    constexpr int nxt = gl + 1 <= N::kLayers ? gl + 1 : 0;   // the next group's first layer; the stream wraps to the next tile's first group
    char* const wbase = lds + G::kWOff;
    char* const wdst = wbase + (wcur ^ 1) * G::kWRegion;
    const char* const wl = wbase + wcur * G::kWRegion + (G::packet_off(L) - G::packet_off(gf));
    auto pre = [&] {
      if constexpr (L == N::kLayers - 3 && !(RCED_F16_EXP & 16)) xr = x_load<W>(P, tile + 1, wave, lane);   // the next tile's input rows, three layers early
      // the next group's packets, issued by the group's FIRST layer: its last vector-memory issue in front of its stores
      if constexpr (L == gf && !(RCED_F16_EXP & 64)) packet_dma<G::group_bytes(nxt), W>(P.wpack + G::packet_off(nxt) / 4, wdst, wave, lane);
    };
    if constexpr (L < N::kLayers) run_layer<N, W, L>(P, region, wl, lds + G::kSOff, scratch, lds + G::kSkipLOff + wave * G::kSkipLdsBytes, lane, pre, res, stamp);
    else run_final<N, W>(P, region, wl, lane, yrow, pre);
    if constexpr (L == gl) {
      wcur ^= 1;
      layer_end_sync();
    }
    run_layers<N, W, L + 1>(P, lds, region, scratch, wcur, xr, res, tile, wave, lane, yrow, stamp);
  }
}

template <class N, int W>
__global__ __launch_bounds__(W * 64, W == 4 ? 2 : 1) void frame16_kernel(Params P) {
  using G = Geo<N, W>;
  constexpr int kWaves = W, kThreads = W * 64;
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
  packet_dma<G::group_bytes(0), W>(P.wpack, lds + G::kWOff, wave, lane);
  int wcur = 0;
  XRows xr = x_load<W>(P, first, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const __amdgpu_buffer_rsrc_t scratch = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(P.scratch) + ((size_t)blockIdx.x * kWaves + wave) * G::kScratchBytesPerWave, 0,
      (int)G::kScratchBytesPerWave, 0x00020000);
  layer_end_sync();
  for (int tile = first; tile < last; ++tile) {
    const int utt = tile / P.tiles_per_utt;
    const int t = (tile - utt * P.tiles_per_utt) * kWaves + wave;
    const long long yrow = t < P.T ? ((long long)utt * P.T + t) * kF : -1;
    x_store(xr, region, lane);   // plane 0 of the wave's own image: its last reader was this wave's previous layer 1
    const bool stamp = RCED_F16_STAMPS && P.stamps && blockIdx.x == 0 && wave == 0 && tile == first + 1;
    Res<N> res;
    run_layers<N, W, 0>(P, lds, region, scratch, wcur, xr, res, tile, wave, lane, yrow, stamp);
    F16_STAMP(stamp, 4 * N::kLayers);
  }
}

}  // namespace frame16
}  // namespace rced
