// Fused CR-CED (V3) forward: all 16 layers of model_utils/model.py:64-96 in ONE kernel.
//
// Why this shape (DESIGN.md has the long form):
//   * only the first conv (8x9) looks along time; everything after is 1xk along frequency, so a
//     tile of frames runs through all 15 conv+BN+ReLU layers (and decode_final) without leaving the CU;
//   * per layer the conv is an implicit GEMM  D[cout, pixel] = sum_k W[cout, k] * X[k, pixel]
//     with k = (tap, cin).  Activations live in LDS as [pixel][channel] with the channel stride
//     EXACTLY cin, so the im2col row of a pixel is one contiguous window of taps*cin values:
//     the B operand of the MFMA is read straight out of LDS, no im2col copy, no shuffles;
//   * cout sits on the MFMA M axis (16 rows).  30 pads to 2 M-tiles; the 30->8 layers use two
//     pixel phases as rows (8 cout x 2 adjacent pixels = 16 rows, K = 10 taps instead of 9); the
//     ->18 layers run channels 0..15 as one M-tile plus a remainder pass that computes channels
//     16,17 for 8 adjacent pixels at once (rows = 8 phases x 2 channels, K = 16 taps);
//   * the two CR-CED block skips (model.py:75-76, added after ReLU) never touch LDS: they stay in
//     the accumulator registers of the wave that produced them;
//   * operands are software-pipelined by hand (read slot i+2, then the MFMAs of slot i): left alone, hipcc issues
//     each LDS read right before its use and the MFMA pipe starves;
//   * a wave walks its tiles one after the other and stores tile j's results between the MFMAs of tile j+1
//     ("slot streams", below): the epilogues' LDS stores no longer sit on every layer's tail.
//
// THREE FORMS of the kernel, one template (`Map<FORM>`), all in the library (option "v3_l2x6" = FORM):
//   FORM 2 (the product, "fused"): as FORM 1, and the 30 -> 8 layers too on the bf16 pipe, computed tap by tap from layer 2's accumulators
//     in the same stream (kernels_fused_v3_l23.h; the note at Map<2> below): the 30-channel tensor is never stored.  Blocks 1..4's
//     layer 1 (8 -> 18) runs on the bf16 pipe as well, from bf16 planes of the 8-channel tensor (RCED_T_L1X6; layer1_x6l).
//   FORM 1 ("X6"): the 18 -> 30 layers -- 43 % of the net's multiply-adds and the only ones whose B fragment
//     feeds two M-tiles -- run at fp32 quality on the bf16 matrix pipe: every operand as three bf16 parts (x = h + m + l,
//     exact to 2^-24), every product as six v_mfma_f32_16x16x32_bf16 (m.m, l.h, h.l, m.h, h.m, h.h -- smallest first --
//     into the fp32 accumulator).  The activation is split ONCE, where it is produced: layer 1's epilogue writes the
//     18-channel tensor as bf16 planes (h, m, l) and layer 2 reads its fragments with ds_read_b128 and issues only
//     MFMAs (round 3 split every fragment in the consumer, 36 VALU per 12 MFMAs, each element five times over).  The
//     planes are 19 KB more than the fp32 buffer; the LDS for them comes from the weights: layer 1's and layer 2's
//     A fragments (50 + 72 registers) are loaded from global memory into VGPRs one layer ahead and only layer 3's
//     packet still goes through LDS (one region, no ping-pong).
//   FORM 0 ("F32"): every layer on v_mfma_f32_16x16x4_f32 (bit-for-bit an fp32 fmaf chain), all weight packets streamed
//     L2 -> LDS by LDS-DMA one layer ahead (ping-pong).  Rounds 1 / 2's kernel; kept as the in-build comparator of the
//     X6 arithmetic (tests/test_forward_gpu.py holds the two against each other and both against the fp64 restatement).
//
// Pixel space of a tile: kTF frames, frame i at flat pixels [i*kS, i*kS+129); the kS-129 = 4 gap
// pixels between frames are always zero and serve as the SAME-padding halo of both neighbours.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "lds_dma.h"

#ifndef RCED_V3_LEGACY_FORMS
#define RCED_V3_LEGACY_FORMS 0   // 1: forms 1 and 2 of the kernel (rounds 3 / 4: kernels_fused_v3_legacy*.h) are compiled too -- history runs through
                                 // tools/ab.sh only; the product library holds form 3 (the product) and form 0 (the bit-exact fp32-MFMA comparator)
#endif
#ifndef RCED_D1
#define RCED_D1 1   // operand prefetch depth (slots) of the layer-1 / layer-2 / layer-3 jobs (1..4 measured: +-0.5 %)
#endif
#ifndef RCED_D2
#define RCED_D2 2
#endif
#ifndef RCED_D2X
#define RCED_D2X 1  // ... of layer 2 in the X6 form (a slot = one K = 32 chunk of one tile: 12 MFMAs = 192 cycles; depth 2 spills)
#endif
#ifndef RCED_D3
#define RCED_D3 2
#endif
#ifndef RCED_LANE_OPAQUE
#define RCED_LANE_OPAQUE 1   // all per-lane addresses hidden from the optimiser (A/B: -0.5 %)
#endif
#ifndef RCED_L1_ORDER
#define RCED_L1_ORDER 1   // layer 1's job order: 1 = pairs, single tile, remainder tiles; 0 = the reverse (see layer1)
#endif
#ifndef RCED_L3_CHAINS
#define RCED_L3_CHAINS 2   // accumulation chains of layer 3's regular job: 4 = (tile, k-quad of the slot), 2 = one per tile (A/B: -0.3 %)
#endif
#ifndef RCED_L2_BOTH
#define RCED_L2_BOTH 1    // X6 form, layer 2: 1 = a wave computes both M-tiles of tiles w + 8t (72 registers of A fragments; every B fragment read
                          // once); 0 = one M-tile of tiles j + 4t (36 registers, every B fragment read by two waves: the layer is bound by the
                          // LDS then -- A/B on one box 8.36 against 8.18 ms)
#endif
#ifndef RCED_T_L1X6
#define RCED_T_L1X6 1     // fused form: blocks 1..4's layer 1 (8 -> 18) on the bf16 pipe too, its input as bf16 planes with 16-byte rows (0: on the fp32
                          // MFMA as the first layer, reading the 8-channel tensor as fp32 [pixel][10]: 6.43 against 6.15 ms)
#endif
#ifndef RCED_T_L1SWAP
#define RCED_T_L1SWAP 1
#endif
#ifndef RCED_V3_LB
#define RCED_V3_LB kThreads   // (diagnostic: 256 lifts the 256-register cap, to see what the allocator would like to have)
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
// finer stamps inside the layer functions: td[k] += cycles since the previous DET / DET_BEGIN of this function
#define DET_ARG , unsigned long long (&td)[16]
#define DET_PASS , tdet
#define DET_BEGIN() unsigned long long dt_ = stamp()
#define DET(k) do { const unsigned long long n_ = stamp(); td[k] += n_ - dt_; dt_ = n_; } while (0)
// RCED_STAMPS == 2: the bf16-pipe layer 1's own breakdown (slot by slot) in td[8..15]
#if RCED_STAMPS == 2
#define DETX(k) DET(8 + (k))
#else
#define DETX(k)
#endif
#else
#define DETX(k)
#define STAMP_BEGIN()
#define STAMP_MATH(i)
#define STAMP_WAIT(i)
#define DET_ARG
#define DET_PASS
#define DET_BEGIN()
#define DET(k)
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kF = 129;
constexpr int kTF = 4;                       // frames per tile
constexpr int kS = 133;                      // pixel stride of a frame (129 + 4 zero gap)
constexpr int kNPX = kTF * kS;               // 532 pixels per tile
constexpr int kWaves = 8;
constexpr int kThreads = kWaves * 64;

// ---- LDS map ----------------------------------------------------------------------------
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
constexpr int kWRegion = 37 * 128 + 64 + 32;              // largest packet: 30->8 (b64 steps + tail + shifts)
constexpr int kX0Rows = kTF + 7;                          // input rows of the 8x9 first layer (they alias B30)
constexpr int kX0Floats = ((kX0Rows * kS + 24 + 3) / 4) * 4;   // 1488: max main-pass index 527 + 7*133 + 8
constexpr int kHS = 10;                                   // decode_final's H image: channel stride (floats): 8-byte aligned b64 reads
constexpr int kHFrame = 193;                              // H pixels per frame: 64 zero + 129 bins
constexpr int kHPix = kTF * kHFrame + 64;                 // 836
constexpr int kFinU = 144, kFinRun = kFinU / kWaves;      // decode_final: window taps; K-steps per wave
constexpr int kFinA = kFinU * 128;                        // floats of its A fragments: [u][lane][2]
constexpr int kFin128 = 80 * 8;                           // bin 128: W[t][c] for taps t = 0..64 (window bins 64..128), zero for t = 65..79
constexpr int kFinPack = kFinA + kFin128;

// The 18-channel tensor between layers 1 and 2 ("B18"):
//   F32 form: fp32 [pixel][18].
//   X6 form:  three bf16 planes (h, m, l) of channels 0..15, [pixel][16] = 32-byte rows -- a lane's fragment of a
//             K = 32 chunk is ONE 16-byte-aligned ds_read_b128 (tap 2c + (kq >> 1), channels 8 (kq & 1)..+7), bank-conflict
//             free in the instruction's lane groups -- and the two remainder channels 16, 17 beside them as rows of
//             [h16 h17 m16 m17] (8 bytes) and [l16 l17] (4 bytes): their five taps ride in the last chunk's upper lanes.
// FORM: 0 = F32, 1 = X6, 2 = X6 with layers 2 and 3 fused (its map is the specialisation below)
template <int FORM>
struct Map {
  static constexpr bool X6 = FORM != 0;
  static constexpr bool kX6 = X6, kFused = false, kAllX6 = false;
  static constexpr int kB8Off = 0;                                     // all offsets in floats unless named *Bytes
  static constexpr int kB18Off = kB8Off + kB8Rows * kB8S;
  static constexpr int kPlaneBytes = kB18Rows * 32;                    // one bf16 plane of 16 channels: 17,088
  static constexpr int kRemRows = kB18Rows + 4;                        // the last chunk's zero-weight slots read 3 rows past a window
  static constexpr int kRemHMBytes = 3 * kPlaneBytes;                  // byte offset of the [h16 h17 m16 m17] rows from the B18 base
  static constexpr int kRemLBytes = kRemHMBytes + kRemRows * 8;        // ... of the [l16 l17] rows
  static constexpr int kB18Bytes = X6 ? ((kRemLBytes + kRemRows * 4 + 15) / 16) * 16 : kB18Rows * 18 * 4;
  static constexpr int kB30Off = kB18Off + kB18Bytes / 4;
  static constexpr int kWRegions = X6 ? 1 : 2;                         // X6: only layer 3's packet goes through LDS
  static constexpr int kWOff = kB30Off + kB30Rows * 30;
  static constexpr int kLdsFloats = kWOff + kWRegions * kWRegion;
  static constexpr int kLdsBytes = kLdsFloats * 4;
  static_assert(kLdsBytes <= 160 * 1024, "LDS budget");
  static_assert((kWOff * 4) % 16 == 0 && (kWRegion * 4) % 16 == 0, "LDS-DMA destinations are 16-byte aligned");
  static_assert((kB18Off * 4) % 16 == 0 && (kB30Off % 2) == 0, "aligned buffers");
  // input rows of the 8x9 first layer alias the (not yet live) B30 buffer, from its pixel-0 row on
  static constexpr int kX0Off = kB30Off + kB30Pad * 30;
  static_assert(kX0Floats <= 60 * 30, "X0 must sit inside rows that layer 2 rewrites");
  // decode_final's H image aliases the (by then dead) B18 buffer, behind layer 3's hand-off scratch (B18 floats 180..951)
  static constexpr int kHOff = kB18Off + 1200;
  static_assert(kHOff + kHPix * kHS <= kB30Off && (kHOff % 2) == 0, "H fits inside B18");
  // decode_final's bin-128 weights: F32 form: behind a layer-1 packet in its weight region; X6 form: behind the H image in
  // rows of the l plane that belong to real pixels of frame 1 (rewritten by every layer 1, never a gap row)
  static constexpr int kFin128Off = kB18Off + (2 * kPlaneBytes + (kS + kB18Pad) * 32) / 4;
  static_assert(!X6 || (kFin128Off >= kHOff + kHPix * kHS && (kFin128Off * 4) % 16 == 0 &&
                        (kFin128Off + kFin128) * 4 <= kB18Off * 4 + 2 * kPlaneBytes + (kS + kF + kB18Pad) * 32),
                "bin-128 weights: behind H, 16-byte aligned, inside frame 1's real rows of the l plane");
  // decode_final's partial sums: 8 waves x 256 floats per column tile, in two stretches of B30 (dead by then) that contain NO
  // gap pixel (B30's gap rows sit at floats 3990.., 7980.., 11970.. and must stay zero) and are clear of X0
  static constexpr int kFinScr0 = kB30Off + 4112, kFinScr1 = kB30Off + 8112;
  static_assert(kFinScr0 >= kX0Off + kX0Floats && ((kFinScr0 * 4) % 16) == 0 && ((kFinScr1 * 4) % 16) == 0, "16-byte aligned, clear of X0");
  static_assert(kFinScr0 >= kB30Off + (kB30Pad + kF + 4) * 30 && kFinScr0 + 2048 <= kB30Off + (kB30Pad + kS + kF) * 30, "no gap row");
  static_assert(kFinScr1 >= kB30Off + (kB30Pad + kS + kF + 4) * 30 && kFinScr1 + 2048 <= kB30Off + (kB30Pad + 2 * kS + kF) * 30, "no gap row");
  static constexpr int finscr0(int w) { return 4 * kFinScr0 + w * 1024; }   // byte offset of wave w's partial sums of column tile 0
  static constexpr int kFinScrCt = 4 * (kFinScr1 - kFinScr0);               // ... + this for column tile 1
  // layer 2's split-tile hand-off: in the B8 buffer, which is dead during layer 2 (layer 3 rewrites every real pixel of it)
  static constexpr int kScr2Slots = X6 ? 4 : 2;                            // helpers of the split tile (1 KiB of partial sums each)
  static constexpr int kScratch2Off = kB8Off + kB8Pad * kB8S + 8 * kB8S;   // B8 rows 8.. of frame 0 (slots x 256 floats + flags)
  static constexpr int kFlag2Off = kScratch2Off + kScr2Slots * 256;
  static_assert((kScratch2Off * 4) % 16 == 0, "scratch is read/written with b128");
  static_assert(8 + (kScr2Slots * 257 + kB8S - 1) / kB8S <= kF, "layer-2 scratch stays inside frame 0's real pixels");
  // layer 3's: in the B18 buffer (dead during layer 3), in rows of frame 0's real pixels: always rewritten by layer 1
  static constexpr int kScratchOff = kB18Off + 180;
  static constexpr int kFlagOff = kScratchOff + 3 * 256;
  static_assert((kScratchOff * 4) % 16 == 0, "scratch is read/written with b128");
  static_assert(X6 ? (kScratchOff - kB18Off) * 4 >= (8 + kB18Pad) * 32 && (kFlagOff + 3 - kB18Off) * 4 <= (kF + kB18Pad) * 32
                   : kScratchOff - kB18Off >= (8 + kB18Pad) * 18 && kFlagOff + 3 - kB18Off <= (kF + kB18Pad) * 18,
                "scratch + flags stay inside frame 0's real pixels");
  static_assert(kFlagOff + 3 <= kHOff, "layer 3's hand-off scratch and the H image do not overlap");
  // byte strides between a wave's regular tiles (8 tiles = 128 pixels apart)
  static constexpr int kT1R = 128 * kB8S * 4, kT1W = X6 ? 128 * 32 : 128 * 18 * 4;
  static constexpr int kT2R = !X6 ? 128 * 18 * 4 : RCED_L2_BOTH ? 128 * 32 : 64 * 32;   // X6, one M-tile per wave: its tiles are 4 apart
  static constexpr int kT2W = X6 && !RCED_L2_BOTH ? 64 * 30 * 4 : 128 * 30 * 4;
  static constexpr int kT3R = 128 * 60 * 4, kT3W = 256 * kB8S * 4;
  static constexpr int kTileB18 = X6 ? 16 * 32 : 16 * 18 * 4;          // ... between adjacent 16-pixel tiles of B18
};
typedef Map<0> MapF32;
#if RCED_V3_LEGACY_FORMS
typedef Map<1> MapX6;
#endif

// ---- packed weight streams (floats), per block -----------------------------------------------
//  first layer main (8x9x1 -> ch 0..15): 18 k-steps x 64 lanes (b32 steps, k = (time tap, freq tap))
//  first layer rem  (ch 16,17 x 8 phases): 32 k-steps x 64 lanes (k = (time tap, 16 freq taps))
//  L1 main (1x9, 8 -> ch 0..15):  9 b64-steps x 64 lanes x 2
//  L1 rem  (ch 16,17 x 8 phases): 16 b64-steps x 64 x 2   (K = 16 taps x 8)
//  L2 (1x5, 18->30), F32 form: 11 b64-steps x 2 M-tiles x 64 x 2 + a b32 tail step (k = 88 + kq; K = 90)
//  L2, X6 form: [chunk 3][M-tile 2][part 3][lane 64] x 8 bf16 (K = 96 slots: three K = 32 chunks, see pack_v3)
//  L3 (1x9, 30->8):  37 b64-steps x 64 x 2 + a b32 tail step (k = 296 + kq; K = 300 = 10 taps x 30,
//                    rows = 2 pixel phases x 8 channels)
// F32 form: one packet per layer, each ending with its 32 shift values (bias + folded BatchNorm), streamed into LDS.
// X6 form: layer 1's and layer 2's weights are laid out for 16-byte-per-lane global loads into registers
// ([j][lane][4 floats]: main j < 5, remainder j < 8), each followed by its 32 shifts; layer 3's packet as in the F32 form.
constexpr int kShiftPerLayer = 32;
constexpr int kW1Main = 9 * 128;          // 1152 (= 18 * 64 for the first layer)
constexpr int kW1Rem = 16 * 128;          // 2048 (= 32 * 64 for the first layer)
constexpr int kW1Data = kW1Main + kW1Rem; // 3200
constexpr int kL2Steps = 11, kL3Steps = 37;                  // b64 steps; each pass ends with one b32 step
constexpr int kL2Chunks = 3;                                  // X6 form: K = 90 in three K = 32 chunks
constexpr int kW2Data = kL2Steps * 2 * 128 + 2 * 64;   // 2944
constexpr int kW3Data = kL3Steps * 128 + 64;           // 4800
constexpr int kW1 = kW1Data + kShiftPerLayer;
constexpr int kW2 = kW2Data + kShiftPerLayer;
constexpr int kW3 = kW3Data + kShiftPerLayer;
constexpr int kWBlock = kW1 + kW2 + kW3;
constexpr int kWTotal = 5 * kWBlock;
constexpr int kG1Main = 5 * 256, kG1Rem = 8 * 256;                        // X6 form: register images of layer 1's A fragments
constexpr int kG1 = kG1Main + kG1Rem + kShiftPerLayer;                    // 3360
constexpr int kG2Data = kL2Chunks * 2 * 3 * 64 * 4;                       // 4608 floats
constexpr int kG2 = kG2Data + kShiftPerLayer;                             // 4640
constexpr int kGBlock = kG1 + kG2 + kW3;
constexpr int kGTotal = 5 * kGBlock;
static_assert(kG1 % 4 == 0 && kG2 % 4 == 0 && kW3 % 4 == 0, "16-byte aligned pieces");
static_assert(kW1 + kFin128 <= kWRegion && (kW1 % 4) == 0, "F32 form: the bin-128 weights fit behind a layer-1 packet in its LDS region");

// ---- layers 2 AND 3 as one stream on the bf16 pipe, the 30-channel tensor never stored (the product form, Map<3>; legacy form 2) ----
// Layer 3 (1x9, 30 -> 8) is computed TAP BY TAP from layer 2's accumulators: a wave that has layer 2's two M-tiles of a 16-pixel
// tile in registers (lane (kq, n): channels 4kq..+3 and 16+4kq..+3 of pixel n) applies the ReLU, splits them into three bf16 parts
// -- which IS the B fragment of a K = 32 MFMA whose k-slot 8kq + e is channel (e < 4 ? 4kq + e : 16 + 4kq + e - 4) -- and
// multiplies by five M-tiles of layer-3 weights, rows (cout 2q + (i >> 1), tap 2j + (i & 1)) for row 4q + i of M-tile j: 30 bf16
// MFMAs give, per lane, P[cout 2kq, 2kq+1][tap t] at INPUT pixel n for the ten taps.  out[cout][p] = sum_t P_t[cout][p + t - 4]:
// every P value is added into the output accumulators of this tile and of one neighbour with DPP row shifts (v_add_f32_dpp,
// 34 per lane and tile: beside bf16 MFMAs nearly free).  What the fp32 form spends on layer 3 -- 75 fp32 MFMAs of 32 cycles
// per 32 pixels, the [pixel][30] stores and their re-reads, a barrier -- becomes 30 bf16 MFMAs of 16 cycles per 16 pixels.
// Tiles are FRAME-ALIGNED here (frame f = waves f and f + 4: tiles 0..3 and 4..8 of its 129 bins; tile 8 has one real pixel):
// no tile straddles a gap, the 4 zero gap pixels absorb every shift across frames, and the only partial sums that cross
// waves are the two at the middle of each frame (LDS scratch + tagged flags, as the split tiles of the other forms).
constexpr int kL3MT = 5;                                   // M-tiles of layer-3 weights (ten taps, the tenth zero)
constexpr int kW3TData = kL3MT * 3 * 256;                  // floats: [M-tile][part h, m, l][lane] x 8 bf16
constexpr int kW3T = kW3TData + kShiftPerLayer;            // 3872
constexpr int kTBlock = kG1 + kG2 + kW3T;
// layer 1 of blocks 1..4 in the three-part form (RCED_T_L1X6): main pass [chunk 3][part 3][lane] x 8 bf16 (k-slot 8kq + e = tap 4c + kq,
// channel e; taps 9..11 zero), remainder pass [chunk 4][part 3][lane] x 8 bf16 (rows = 8 phases x channels 16, 17; k-slot = window tap
// 4c + kq, channel e), 32 shifts
constexpr int kG1XMain = 9 * 256, kG1XRem = 12 * 256;
constexpr int kG1X = kG1XMain + kG1XRem + kShiftPerLayer;
constexpr int kTBlockX = kG1X + kG2 + kW3T;                 // blocks 1..4 (block 0's first layer keeps the fp32 images)
constexpr int kTTotal = RCED_T_L1X6 ? kTBlock + 4 * kTBlockX : 5 * kTBlock;
constexpr int kB8PlaneBytes = kB8Rows * 16;                 // one bf16 plane of the 8-channel tensor: [pixel][8] = 16-byte rows
static_assert(kW3T % 4 == 0 && kG2 + kW3T <= 2 * kWRegion, "layer 3's fused-form packet: 16-byte pieces, inside a weight region");
#if RCED_V3_LEGACY_FORMS
#include "kernels_fused_v3_legacy_map.h"
#endif
// ---- ALL-X6 form (Map<3>, the product): as the fused form, and the FIRST layer and decode_final on the bf16 pipe too ------------
// * The first layer (8x9, 1 -> 18) is layer1_x6l like blocks 1..4's: its K axis (8 time rows x 9 frequency taps) becomes "8 channels x 9
//   taps" once the input rows of a tile are laid out as bf16 planes [pixel (frame i, bin f)][8] with entry r = x[t0 + i + r - 3][f] -- an
//   im2col along time only, written once per tile by convert_x0 (8 LDS reads, 4 splits, 3 sixteen-byte stores per pixel).  All five
//   blocks then run ONE copy of layer-1 code, and no layer's weights live in registers.
// * decode_final (1x129, 8 -> 1) reads block 4's output as three bf16 planes too.  GEMM as in the other forms -- rows = 16 bin phases m,
//   columns = (frame, 16-bin block j), K = (144 window taps u) x 8 channels in 36 chunks of 4 taps, A[m][(u, c)] = W[u - m][c] -- but
//     - the A fragment of lane (kq, m) for chunk q is W[4q + kq - m][0..7]: sixteen contiguous bytes of a [tap][8] table with zero rows
//       around taps 0..128.  The whole Toeplitz operand is that 7.5-KB table, resident in LDS for the kernel's lifetime (the fp32 form
//       streams 73.7 KB of fragments from the L2 into registers per tile);
//     - the image ("H'") keeps bin b of frame f at row 140 f + b + (b >> 4): a pad row per 16 bins.  The 16 columns of a read are bins
//       16 rows apart in up to four frames; at a plain stride of 16 rows = 256 bytes the four blocks of a frame would meet on one bank
//       slot (4-way conflicts: the phase would be bound by the LDS), with 17 they sit on neighbouring slots and the four frames (140 = 12
//       mod 16) four slots apart: one two-way meeting per read.  b >> 4 of a window tap is (j - 4) + (q >> 2): the same for every lane of a
//       chunk, so the pad rows cost no per-lane arithmetic;
//     - there are no zero pads around a frame: a lane whose tap lies outside bins 0..128 reads a zero row instead (one compare + select
//       per chunk and column tile on the row address);
//     - bin 128 is a third column tile (block 8: one row of 16 used, chunks 0..16 only) instead of a dot product on one wave's VALU.
//   The image lies over the 8-channel planes (dead once block 4's layer 1 has run); convert_x0 rewrites them for the next tile and puts
//   the zeros back into their gap rows.  K is cut in eight runs, one per wave (4 chunks x 3 column tiles on waves 0..3, 5 chunks x 2
//   on waves 4..7: 22 / 23 six-MFMA sets per SIMD); partial sums meet in LDS (rows of the 18-channel planes, dead by then).
constexpr int kHFr = 140;                        // H' rows per frame (137 used: bin 128 sits at row 136; rows 137..139 spare)
constexpr int kHRows = kTF * kHFr;               // 560
constexpr int kHPlaneBytes = kHRows * 16;        // one part of the image: 8,960
constexpr int kHZeroRow = kHRows - 1;            // the zero row out-of-range taps read (frame 3's last spare row; zeroed once per tile)
constexpr int kFinTRows = 160;                   // table rows: tap t at row t + 15, t = -15 .. 144
constexpr int kFinTPart = kFinTRows * 16;        // bytes per part
constexpr int kFinTFloats = 3 * kFinTPart / 4;   // 1,920
constexpr int kATotal = 5 * kTBlockX;            // the weight stream of this form: every block in the bf16-pipe format
template <>
struct Map<3> {
  static constexpr bool X6 = true, kX6 = true, kFused = true, kAllX6 = true, kL1X6 = true;
  static_assert(RCED_T_L1X6 != 0, "the all-x6 form is built on layer1_x6l");
  static constexpr int kB8Off = 0;                                              // the 8-channel planes (3 x 8,640 B) / decode_final's image (3 x 8,960 B)
  static constexpr int kB8RegionBytes = 3 * (kHPlaneBytes > kB8PlaneBytes ? kHPlaneBytes : kB8PlaneBytes);
  static constexpr int kB18Off = kB8Off + kB8RegionBytes / 4;
  // B18 here: per part (h, m, l) TWO half-planes [pixel][8 channels] (16-byte rows: channels 0..7 / 8..15) and the remainder channels' rows
  // [c16 c17].  A lane of layer 1's epilogue owns 4 channels of a pixel = 8 bytes; with [pixel][16] rows of 32 bytes the 16 lanes one
  // ds_write_b64 group serves sat 32 bytes apart -- on 4 of the 16 eight-byte bank slots, 4-way conflicts on every one of the layer's 99
  // plane stores (1 k of its 6 k cycles); with 16-byte rows they sit 16 bytes apart: 2-way, which a ds_write_b64 absorbs.  Layer 2's
  // fragment (8 channels of a pixel) is still one aligned ds_read_b128; half-planes a multiple of 256 bytes apart keep the lane groups
  // of that read conflict-free.
  static constexpr int kHalfBytes = ((kB18Rows * 16 + 255) / 256) * 256;       // 8,704
  static constexpr int kRemOff = 2 * kHalfBytes;
  static constexpr int kRemRows = kB18Rows + 6;
  static constexpr int kPlaneBytes = ((kRemOff + kRemRows * 4 + 15) / 16) * 16;   // stride between the parts: 19,568
  static constexpr int kRemHMBytes = 0, kRemLBytes = 0;
  static constexpr int kB18Bytes = 3 * kPlaneBytes;
  static constexpr int kWOff = kB18Off + kB18Bytes / 4;                         // layer 2's + layer 3's images of the block
  static constexpr int kW3TOff = kWOff + kG2;
  static constexpr int kWRegions = 1;
  static constexpr int kB30Off = kWOff;
  static constexpr int kW1Off = kW3TOff + kW3T;                                 // layer 1's image of the block (21 pieces + shifts)
  static constexpr int kFinTOff = kW1Off + kG1X;                                // decode_final's tap table
  static constexpr int kX0Off = kFinTOff + kFinTFloats;                         // the next tile's input rows, fp32 (convert_x0's source)
  static constexpr int kHOff = kB8Off, kFin128Off = kB8Off;                     // (other forms' buffers: make_lane's unused addresses)
  static constexpr int kEdgeOff = kX0Off + kX0Floats;
  static constexpr int kEdgeFlagOff = kEdgeOff + 4 * 2 * 128;
  static constexpr int kLdsFloats = kEdgeFlagOff + 8;
  static constexpr int kLdsBytes = kLdsFloats * 4;
  static_assert(kLdsBytes <= 160 * 1024, "LDS budget");
  static_assert((kB18Off * 4) % 16 == 0 && (kWOff * 4) % 16 == 0 && (kW3TOff * 4) % 16 == 0 && (kW1Off * 4) % 16 == 0 && (kFinTOff * 4) % 16 == 0 &&
                (kX0Off * 4) % 16 == 0 && (kEdgeOff * 4) % 16 == 0, "aligned buffers");
  // decode_final's partial sums: 8 waves x 3 column tiles x 1 KiB in rows of the 18-channel planes' REAL pixels (their gap rows must stay
  // zero): column tile ct in part ct, wave w in the first 64 rows of frame w & 3 of half-plane w >> 2
  static constexpr int finscr(int w, int ct) { return kB18Off * 4 + ct * kPlaneBytes + (w >> 2) * kHalfBytes + (kB18Pad + kS * (w & 3)) * 16; }
  static constexpr int kT1R = 128 * kB8S * 4, kT1W = 128 * 16;
  static constexpr int kT2R = 0, kT2W = 0, kT3R = 0, kT3W = 0;
  static constexpr int kTileB18 = 16 * 16;
};
typedef Map<3> MapA;
static_assert(MapA::finscr(7, 2) % 16 == 0 && MapA::finscr(3, 0) + 1024 <= MapA::kB18Off * 4 + (kB18Pad + kS * 3 + kF) * 16, "partial sums: aligned, inside real rows");

// ---- decode_final (1x129, 8 -> 1, no BN, no ReLU; model.py:89-90) inside the kernel ----------------------------
// The CD2 output of a tile never leaves the CU: block 4's layer 3 stores it to H, an LDS image that aliases the (by then
// dead) B18 buffer: H pixel 193*i + 64 + f holds bin f of frame i, channel stride kHS; the 64 pixels in front of every
// frame (and behind the last) are zero -- the SAME padding of the 129-tap kernel (64 taps each side).
// GEMM: rows = 16 pixel phases r, columns = (frame i, block j) with output bin f = 16j + r (bins 0..127: 32 columns =
// two column tiles: tile ct holds blocks 4ct..4ct+3 of all four frames, which spreads a read's 16 columns over more LDS
// banks than 8 blocks of 2 frames), K = (144 window taps u) x 8 channels, window start = bin 16j - 64, A[r][(u, c)] = W[u - r][c]
// (zero outside 0..128).  The 144 b64 K-steps are cut in 8 runs of 18, one per wave (both column tiles: 72 MFMAs per
// wave); the eight partial sums meet in LDS scratch (B30 is dead) and waves 0 / 1 finish column tile 0 / 1 in a fixed
// order (deterministic).  Bin 128 (one output per frame, 65 x 8 taps) is a dot product on the VALU of wave 2.
// A fragments (73.7 KB) stream from L2 into registers, issued before block 4's layer 3 so that their latency hides.

// ---- tile -> wave assignment ---------------------------------------------------------------
// 16-pixel tiles 0..32 (pixels 0..527; 528..531 is gap): wave w owns tiles w + 8*slot, slot < 4
// ("regular": one address register per wave, everything else immediates); tile 32 and, in layer 1,
// tile 31 are handed out as "extra" tiles.  Pair tiles 0..16: w + 8*slot, slot < 2; extra 16.
// Waves w and w+4 share a SIMD, and the barrier that ends a layer waits for the most loaded SIMD, so
// what is balanced is each LAYER's MFMA count per SIMD:
//   layer 1: remainder tiles (128 pixels) 0..4 go one each to waves 4,5,6 and two to wave 7, which gives up
//            main tile 31 (to wave 1; tile 32 to wave 0): 194, 194, 176, 190 per SIMD (18 per main tile,
//            32 per remainder tile).
//   layer 2: tile 32 is cut in four, M-tile x K-part, one piece on each of waves 0..3.
//   layer 3: pair tile 16 is cut in four along K on waves 0..3: 318.75 everywhere.
// The cut tiles are put back together through LDS scratch + tagged flag words (pairwise, no extra barrier).

struct Params {
  const float* x;       // [N, T, 129]
  float* y;             // [N, T, 129]    the mask (output of decode_final)
  const float* wpack;   // kWTotal (F32 form) / kGTotal (X6 form) floats
  const float* fin;     // kFinPack floats: decode_final's A fragments + its bin-128 weights (pack_v3); all-x6 form: kFinTFloats, its tap table
  float fin_bias;
  int N, T;
  int tiles_per_utt;    // ceil(T / kTF)
  int total_tiles;      // N * tiles_per_utt
  unsigned long long* stamps;  // diagnostic builds only (RCED_STAMPS): [wave][8] cycle sums of workgroup 0
  unsigned* err;        // sticky error word (host-visible): bit 1 / 2 = a layer-2 / layer-3 hand-off flag never came
};

__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma32(s16x8 a, s16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void pin() { __builtin_amdgcn_sched_barrier(0); }

// hipcc hoists loop-invariant per-lane address arithmetic out of the tile loop -- every layer's, for every wave role --
// and then keeps (or spills) dozens of VGPRs across all fifteen layers.  Each layer function therefore re-derives its
// lane coordinates from a copy of the lane id the optimiser cannot see through: a handful of VALU per layer.
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

// Stream one weight packet global -> LDS with LDS-DMA (no VGPR staging, no ds_write): wave w copies
// the 1-KiB chunks w, w+8, w+16; lane l of a chunk moves 16 bytes.  Completion: vmcnt(0) + barrier
// at the end of the layer during which it was issued.
template <int NFLOATS>
__device__ __forceinline__ void packet_dma(const float* __restrict__ src, float* dst, int wave, int lane) {
  lane = opaque(lane);   // (or hipcc keeps per-lane source addresses of every call site alive across the tile loop)
  constexpr int n4 = NFLOATS / 4;
  constexpr int chunks = (n4 + 63) / 64;
#pragma unroll
  for (int i = 0; i < (chunks + kWaves - 1) / kWaves; ++i) {
    const int c = wave + i * kWaves;
    if (c < chunks) {
      if (c * 64 + lane < n4)   // wave-uniform source in SGPRs + one 32-bit lane offset: no 64-bit VALU address per piece
        lds_dma16s(src + c * 256, (unsigned)lane * 16u, dst + c * 256);
    }
  }
}
__device__ __forceinline__ void layer_end_sync() {
  // This wave's LDS-DMA pieces (issued from asm: invisible to hipcc) and register-bound weight loads have landed.  The wait is
  // stated twice: as asm (never dropped) and as the builtin, which hipcc's wait-count pass sees -- without it the pass believes
  // the weight loads of the previous layer may still be in flight and puts s_waitcnt vmcnt(N) in front of their first uses,
  // where N counts only the loads it knows: in hardware that waits for the prefetches just issued for the NEXT layer.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched (gfx9 encoding)
#if defined(RCED_T_EXP) && (RCED_T_EXP & 16384)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // timing experiment (WRONG RESULTS): no workgroup barrier between the phases
#else
  __syncthreads();
#endif
}


// ReLU as ONE integer max per element: for IEEE floats max(bits, 0) == bits of relu(x) (negative
// floats are negative ints).  fmaxf() on an MFMA result makes hipcc add a canonicalising
// v_max_f32 v,v,v in front (2 VALU per element); an inline-asm v_max_f32 is NOT an option: hipcc pads
// no MFMA -> VALU wait states inside asm, and a ReLU issued right behind the last MFMA then reads the
// accumulator before it lands (seen as rare wrong values in each wave's first tile).
// (A NaN keeps its bits -- positive-signed NaNs, which is what the hardware generates, stay NaN.)
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
  tid = opaque(tid);   // (once per tile: nothing of this is worth keeping in registers across the tile loop)
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

// ---------------------------------------------------------------------------------------------
// Layer passes as SLOT STREAMS.
//
// What the passes are built around (measured, tools/micro/mfma_lds_rate.hip and the s_memtime stamps):
//   * v_mfma_f32_16x16x4_f32 shares the SIMD's vector ALU with ordinary VALU work: VALU instructions issued by either
//     wave of a SIMD are NOT hidden behind the other wave's MFMAs (4 extra VALU per 4-MFMA slot: 95 % -> 70 % of the
//     MFMA rate).  So the passes spend as few VALU instructions as they can: every LDS address is a per-lane base that
//     is computed once per kernel (struct Lane) plus a compile-time immediate; gap pixels are handled by exec masks from
//     precomputed validity bits behind wave-uniform branches (3 of 33 tiles), not by per-element selects; block-dependent
//     work (skip adds, skip saves, the last block's global stores) sits behind wave-uniform branches.
//   * LDS reads are cheap beside MFMAs (0.4 -> 1 read per MFMA costs ~2 %), LDS STORES are not (~85 B/clk per CU: a
//     layer's 64 KB of outputs is ~750 cycles): a wave therefore walks its tiles in small jobs ("slots" = one b64 K-step
//     of one job) in ONE software-pipelined stream -- operands of slot i+D are read while the MFMAs of slot i issue,
//     across job boundaries -- and job j's stores are issued between the MFMAs of job j+1.
//   * Split tiles (the odd tile of layers 2 and 3, shared by waves 0..3 so that every SIMD carries the same MFMA count)
//     are each wave's FIRST job: a helper publishes its partial sums through LDS scratch + a tagged flag word a few slots
//     into its next job and the reducer picks them up after its last job, when they have long been there.  LDS
//     operations of one wave execute in order, so "data store, then flag store" / "flag poll, then data load" need no
//     fence -- only the compiler must keep the order (volatile flag accesses + cbar()).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void cbar() { asm volatile("" ::: "memory"); }

// Compile-time slot loop: the streams are 40-90 slots long and every slot differs (job, step, which epilogue piece rides
// along), so they are instantiated slot by slot instead of being left to the loop unroller -- a `#pragma unroll` loop
// of this size exceeds hipcc's pragma-unroll threshold, stays a loop, and then indexes its operand ring and
// accumulators dynamically (s_set_gpr_idx: 2.6x slower).
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
template <int V>
using IC = std::integral_constant<int, V>;

// ---- LDS by byte address: base VGPR + immediate ------------------------------------------------------
typedef __attribute__((address_space(3))) char lds_char;
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}
template <class T>
__device__ __forceinline__ T lds_ld(unsigned base, int imm) {
  return *reinterpret_cast<const __attribute__((address_space(3))) T*>(reinterpret_cast<const lds_char*>(base) + imm);
}
template <class T>
__device__ __forceinline__ void lds_st(unsigned base, int imm, T v) {
  *reinterpret_cast<__attribute__((address_space(3))) T*>(reinterpret_cast<lds_char*>(base) + imm) = v;
}
__device__ __forceinline__ unsigned lds_peek_a(unsigned addr) {
  return *reinterpret_cast<volatile const __attribute__((address_space(3))) unsigned*>(reinterpret_cast<const lds_char*>(addr));
}
__device__ __forceinline__ void lds_poke_a(unsigned addr, unsigned v) {
  *reinterpret_cast<volatile __attribute__((address_space(3))) unsigned*>(reinterpret_cast<lds_char*>(addr)) = v;
}

// Per-lane LDS byte addresses and masks, computed ONCE per kernel (they depend on the lane and the wave only).
// Everything a layer touches is one of these plus a compile-time immediate (plus, for the odd tiles, a wave-uniform
// delta: one v_add per layer).  Members a form does not use are never computed.
struct Lane {
  unsigned a8, a4, kq16;   // lane*8, lane*4, (lane>>4)*16: offsets inside a weight packet (A fragments, tail, shifts)
  unsigned rd0, rd0b, rd0r;// block 0's layer 1: input-row window start of main tile `wave` (/ + 8) and of remainder tile xr0
  unsigned rd1, rd1b, wr1; // layer 1 main tile `wave` (/ the tile 8 further: a base of its own, see make_lane): B8 window start, B18 output
  unsigned rd1r, wr1r;     // layer 1 remainder tile xr0 (waves 4..7); X6: wr1r = the [h m] rows, wr1rl = the [l] rows
  unsigned wr1rl;
  unsigned rd2, rd2b, rd2t, rd2tb, wr2; // layer 2 tile `wave` (/ + 8): B18 window start (b64 steps / tail), B30 output (F32 form)
  unsigned rd2m, rd2r, rd2rl;           // X6 form: this lane's fragment of chunk 0 in the h plane; its remainder rows ([h m] / [l])
  unsigned rd1x, rd1xb, rd1xr, wr3p;    // fused form, layer 1 on the bf16 pipe: B8 plane row of main tile `role` (/ + 8) / remainder tile xr0, chunk 0; layers 2 + 3's output row
  unsigned rd2c, rd2cs;                 // fused form: this lane's four dwords of the last chunk (h part) and their byte stride from tile to tile
  unsigned rd3, rd3b, rd3t, rd3tb, wr3; // layer 3 pair tile `wave` (/ + 8): B30 window start (b64 steps / tail), B8 output
  unsigned wh0, wh1, whx;  // block 4's layer 3: where this lane's output pixel of pair tile 0 / 1 / 16 goes in the H image
  unsigned scr;            // lane*16: offset inside a hand-off scratch area
  unsigned vbits;          // validity bits, see kV*
};
// vbits: bit t (0..3) main tile wave+8t pixel is a real bin; 4,5 / 6,7: remainder tile 0 / 1, this lane's two pixels;
// 8,9: layer-3 pair tiles 0,1; 10: pair tile 16 (the split one); 11: lane < 48 (kq != 3: channels 28,29 vs padding 30,31);
// 12: lane >= 32 (the lanes whose slots of layer 2's last chunk are the remainder channels); X6 form: 13: this lane's second
// pair of layer-2 outputs is real (M-tile 0, or lane < 48); 16..23: pixel of layer-2 tile (wave & 3) + 4t is a real bin
// 24: lane & 15 == 0 (fused form: the one real pixel of a frame's last tile)
constexpr int kVMain = 0, kVRem = 4, kVL3 = 8, kVL3X = 10, kVLt48 = 11, kVUpper = 12, kVSt2 = 13, kVL2 = 16, kVN0 = 24;

template <class M>
__device__ __forceinline__ Lane make_lane(float* lds, int wave, int lane, int xr0, int w1) {   // w1: the wave's role in layer 1 (fused form: != wave)
  const int n = lane & 15, kq = lane >> 4;
  const unsigned B8 = lds_addr(lds + M::kB8Off + kB8Pad * kB8S), B30 = lds_addr(lds + M::kB30Off + kB30Pad * 30);
  const int px0 = 16 * w1 + n;
  Lane L;
  L.a8 = lane * 8;
  L.a4 = lane * 4;
  L.kq16 = kq * 16;
  L.scr = lane * 16;
  const unsigned X0 = lds_addr(lds + M::kX0Off);
  L.rd0 = X0 + 4 * (px0 + kq * kS);                         // lane kq <-> time taps 4*ih + kq
  // Remainder tiles: column n of tile xr = the kRemPx pixels from kRemPx * (16 xr + n) on.  Fused form: SEVEN of the eight pixel phases
  // are used (the rows of phase 7 are computed and dropped): a column stride of 7 rows puts the 16 columns of a ds_read_b128 lane
  // group on 16 different 16-byte bank slots, where a stride of 8 rows = 128 bytes puts them on two (8-way conflicts: 60 reads per
  // layer at 32 cycles each); five tiles still cover the 532 pixels (5 x 112), so the MFMA count is the same.
  // Which of the tile's 16 pixel columns a lane column n takes is permuted (nr): ds_read_b128 serves lanes {0-3, 12-15, 20-27} together,
  // i.e. columns {0-3, 12-15} at tap kq and columns {4-11} at tap kq + 1 -- with rows 7 nr(n) + kq they meet on one bank slot per
  // group instead of seven (nr maps {4..11} to rows 0..7 mod 16 and {0-3, 12-15} to 8..15).
  constexpr int kRemPx = M::kFused ? 7 : 8;
  const int nr = M::kFused ? (int)((0x92B41A3C5E70D6F8ull >> (4 * n)) & 15) : n;
  L.rd0r = X0 + 4 * (kRemPx * (16 * xr0 + nr) + kq * kS);
  L.rd1 = B8 + 4 * ((px0 - 4) * kB8S + 2 * kq);
  const int rpx = kRemPx * (16 * xr0 + nr);                 // first pixel of this lane's column in remainder tile xr0
  L.rd1r = B8 + 4 * ((rpx - 4) * kB8S + 2 * kq);
  if constexpr (M::kX6) {
    const unsigned PL = lds_addr(lds + M::kB18Off);         // the h plane; row r = pixel r - kB18Pad
    L.wr1 = PL + (px0 + kB18Pad) * 32 + kq * 8;             // channels 4kq..4kq+3 of pixel px0
    if constexpr (M::kAllX6) L.wr1 = PL + (kq >> 1) * M::kHalfBytes + (px0 + kB18Pad) * 16 + (kq & 1) * 8;   // (half-planes: Map<3>)
    L.wr1r = PL + M::kRemHMBytes + (rpx + 2 * kq + kB18Pad) * 8;   // channels 16,17 of pixels rpx+2kq, +1
    L.wr1rl = PL + M::kRemLBytes + (rpx + 2 * kq + kB18Pad) * 4;
    if constexpr (M::kFused) L.wr1r = PL + M::kRemOff + (rpx + 2 * kq + kB18Pad) * 4;   // [c16 c17] of the h part; m, l: + the part stride
    // layer 2: wave (g = M-tile, j) walks tiles j + 4t; px2 = this lane's pixel of tile j
    // chunk c: tap 2c + (kq >> 1) = pixel px2 - 2 + tap = row px2 + tap, channels 8 (kq & 1)..+7
    // (fused form: frame-aligned tiles -- waves 0..3: tiles 0..3 of frame `wave`, waves 4..7: tiles 4..8 of frame `wave - 4`)
    const int px2 = M::kFused ? kS * (wave & 3) + 64 * (wave >> 2) + n : RCED_L2_BOTH ? px0 : 16 * (wave & 3) + n;
    L.rd2m = PL + (px2 + (kq >> 1)) * 32 + (kq & 1) * 16;
    if constexpr (M::kAllX6) L.rd2m = PL + (kq & 1) * M::kHalfBytes + (px2 + (kq >> 1)) * 16;
    // the remainder channels' window of pixel px2 = rows px2 .. px2+4; lanes kq = 2 take rows px2..+3, kq = 3 rows px2+4..+7
    // (one real tap, three zero-weight slots); the lower lanes read their upper partners' rows (same addresses: broadcast)
    L.rd2r = PL + M::kRemHMBytes + (px2 + 4 * (kq & 1)) * 8;
    L.rd2rl = PL + M::kRemLBytes + (px2 + 4 * (kq & 1)) * 4;
    if constexpr (M::kFused) {   // lanes kq < 2: tap 4 (row px2 + 4), channels 8kq..; kq = 2: rows px2..px2+3 of [c16 c17]; kq = 3: rows px2+4..
      L.rd2c = kq < 2 ? L.rd2m + (M::kAllX6 ? 64 : 128) : PL + M::kRemOff + (px2 + 4 * (kq & 1)) * 4;
      L.rd2cs = kq < 2 ? (unsigned)M::kTileB18 : 64u;
      asm volatile("" : "+v"(L.rd2c), "+v"(L.rd2cs));
    }
    L.rd2 = L.rd2b = L.rd2t = L.rd2tb = 0u;
  } else {
    const unsigned B18 = lds_addr(lds + M::kB18Off + kB18Pad * 18);
    L.wr1 = B18 + 4 * (px0 * 18 + 4 * kq);
    L.wr1r = B18 + 4 * ((rpx + 2 * kq) * 18 + 16);          // channels 16,17 of pixels rpx+2kq, +1
    L.rd2 = B18 + 4 * ((px0 - 2) * 18 + 2 * kq);
    L.rd2t = L.rd2 + 4 * (8 * kL2Steps + (kq < 1 ? kq : 1) - 2 * kq);   // K = 90: tail k = 88 + kq is real for kq < 2
    L.wr1rl = L.rd2m = L.rd2r = L.rd2rl = 0u;
  }
  if constexpr (M::kX6 && !RCED_L2_BOTH) L.wr2 = B30 + 4 * ((16 * (wave & 3) + n) * 30 + 16 * (wave >> 2) + 4 * kq);   // channels 16 g + 4kq..
  else L.wr2 = B30 + 4 * (px0 * 30 + 4 * kq);
  L.rd3 = B30 + 4 * ((2 * px0 - 4) * 30 + 2 * kq);          // px0 doubles as the pixel-PAIR index of layer 3
  L.rd3t = L.rd3 + 4 * (8 * kL3Steps - kq);                 // K = 300: tail k = 296 + kq, all four real
  L.wr3 = B8 + 4 * ((2 * px0 + (kq >> 1)) * kB8S + 4 * (kq & 1));
  if constexpr (M::kFused) L.wr3 = B8 + 4 * ((kS * (wave & 3) + 64 * (wave >> 2) + n) * kB8S + 2 * kq);   // channels 2kq, 2kq+1 of this lane's pixel
  if constexpr (M::kFused) {   // the 8-channel tensor as bf16 planes, row = pixel + kB8Pad; chunk c of layer 1 = taps 4c + kq = rows pixel - 4 + 4c + kq
    const unsigned B8P = lds_addr(lds + M::kB8Off);
    L.rd1x = B8P + (px0 - 4 + kq + kB8Pad) * 16;
    L.rd1xb = L.rd1x + 128 * 16;
    L.rd1xr = B8P + (rpx - 4 + kq + kB8Pad) * 16;
    L.wr3p = B8P + (kS * (wave & 3) + 64 * (wave >> 2) + n + kB8Pad) * 16 + 4 * kq;
    asm volatile("" : "+v"(L.rd1x), "+v"(L.rd1xb), "+v"(L.rd1xr), "+v"(L.wr3p));
  }
  {
    const unsigned H = lds_addr(lds + M::kHOff);
    auto haddr = [&](int px) {   // px: tile-flat pixel of frame px / kS, bin px % kS
      const int fr = px / kS, f = px - fr * kS;
      return H + 4 * ((kHFrame * fr + 64 + f) * kHS + 4 * (kq & 1));
    };
    L.wh0 = haddr(2 * px0 + (kq >> 1));
    L.wh1 = haddr(2 * (px0 + 128) + (kq >> 1));
    L.whx = haddr(2 * (256 + n) + (kq >> 1));
    if constexpr (M::kFused) L.wh0 = H + 4 * ((kHFrame * (wave & 3) + 64 + 64 * (wave >> 2) + n) * kHS + 2 * kq);
    // all-x6 form: decode_final's image H': bin b = 64 (wave >> 2) + 16 t + n of frame wave & 3 at row 140 f + b + (b >> 4) = this + 17 t; channels 2kq, 2kq + 1
    if constexpr (M::kAllX6) L.wh0 = lds_addr(lds + M::kB8Off) + (kHFr * (wave & 3) + 68 * (wave >> 2) + n) * 16 + 4 * kq;
  }
  unsigned v = 0;
#pragma unroll
  for (int t = 0; t < 4; ++t) v |= (unsigned)px_valid(px0 + 128 * t) << (kVMain + t);
#pragma unroll
  for (int r = 0; r < 2; ++r) {   // remainder tiles xr0 and (wave 7 only) 4
    const int p = kRemPx * (16 * (r == 0 ? xr0 : 4) + nr) + 2 * kq;
    v |= (unsigned)(p >= 0 && px_valid(p)) << (kVRem + 2 * r);
    v |= (unsigned)(p >= 0 && px_valid(p + 1) && 2 * kq + 1 < kRemPx) << (kVRem + 2 * r + 1);   // (fused form: phase 7 belongs to the next column)
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) v |= (unsigned)px_valid(2 * (px0 + 128 * t) + (kq >> 1)) << (kVL3 + t);
  v |= (unsigned)px_valid(2 * (256 + n) + (kq >> 1)) << kVL3X;
  v |= (unsigned)(lane < 48) << kVLt48;
  v |= (unsigned)(lane >= 32) << kVUpper;
  v |= (unsigned)(n == 0) << kVN0;
  if constexpr (M::kX6) {
    v |= (unsigned)(wave < 4 || lane < 48) << kVSt2;
#pragma unroll
    for (int t = 0; t < 8; ++t) v |= (unsigned)px_valid(16 * (wave & 3) + n + 64 * t) << (kVL2 + t);
  }
  L.vbits = v;
  // The second tile of a pair job gets a base register of its own, hidden from the optimiser: with one base and two
  // immediates hipcc fuses the two reads of a slot into one ds_read2[st64]_b64, which needs a per-slot v_add for its
  // re-based address (VALU beside MFMAs is not free) and takes twice the LDS cycles of two ds_read_b64.
  L.rd0b = L.rd0 + 128 * 4;
  L.rd1b = L.rd1 + M::kT1R;
  L.rd2b = L.rd2 + M::kT2R;
  L.rd3b = L.rd3 + M::kT3R;
  L.rd2tb = L.rd2t + M::kT2R;
  L.rd3tb = L.rd3t + M::kT3R;
  asm volatile("" : "+v"(L.rd0b), "+v"(L.rd1b), "+v"(L.rd3b), "+v"(L.rd3tb));
  if constexpr (!M::kX6) asm volatile("" : "+v"(L.rd2b), "+v"(L.rd2tb));
#if RCED_LANE_OPAQUE
  // every other address too: left visible, hipcc keeps only the lane-dependent part in a VGPR and re-adds the (scalar)
  // buffer base at each use -- a v_add per job and per epilogue store
  asm volatile("" : "+v"(L.rd0), "+v"(L.rd0r), "+v"(L.rd1), "+v"(L.wr1), "+v"(L.rd1r), "+v"(L.wr1r));
  asm volatile("" : "+v"(L.wr2), "+v"(L.rd3), "+v"(L.rd3t), "+v"(L.wr3), "+v"(L.wh0), "+v"(L.wh1), "+v"(L.whx));
  if constexpr (M::kX6) asm volatile("" : "+v"(L.wr1rl), "+v"(L.rd2m), "+v"(L.rd2r), "+v"(L.rd2rl));
  else asm volatile("" : "+v"(L.rd2), "+v"(L.rd2t));
#endif
  return L;
}
__device__ __forceinline__ bool vbit(const Lane& L, int b) { return (L.vbits >> b) & 1u; }

// Bounded wait for a tagged LDS flag word.  All waves of the workgroup are resident, so the flag always comes; if it
// does not within ~4 M polls the wave records it in the sticky error word the host checks (RCED_ERR_STATE) and goes on
// rather than hang the GPU.
__device__ __forceinline__ void flag_wait(unsigned flag_addr, unsigned tag, unsigned* err, unsigned code) {
  bool ok = false;
  for (int spin = 0; spin < (1 << 22); ++spin) {
    if (__builtin_amdgcn_readfirstlane(lds_peek_a(flag_addr)) == tag) {
      ok = true;
      break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  if (!ok && err && (threadIdx.x & 63) == 0) atomicOr(err, code);
  cbar();
}

// wave-uniform: does 16-pixel tile T (pixels 16T..16T+15) contain a gap pixel?  (tiles 8, 16, 24)
__device__ __forceinline__ bool tile_has_gap(int T) { return span_has_gap(16 * T, 16); }

// ---- jobs -----------------------------------------------------------------------------------------------
// A job = one mini-stream: D slots of operands are read ahead, then every slot reads the operands of slot i+D and issues
// its MFMAs.  Jobs are self-contained (a wave's pipeline drains between two jobs: its SIMD partner's MFMAs fill the
// gap) so that ONE copy of each job's code serves every wave and every tile: what differs -- which tiles, the odd
// tiles, the shares of the split tiles -- is base registers and wave-uniform branches around the shared code.  With
// one unrolled stream per wave role (an earlier version) the kernel was 60 KB of code and ran 40 % slower than at
// 52 KB: the instruction cache (64 KB for two CUs) no longer held what 16 waves were executing.
// `pre` runs once the job's first operand reads are in flight: a layer's first job issues the next layer's weight
// transfers there (Once, below) -- ~30 mostly scalar instructions that used to sit between the barrier and the first LDS
// read of the layer, where nothing overlaps them; here they run in the shadow of the reads' latency.
template <int NSLOT, int D, class LoadF, class MathF, class Pre>
__device__ __forceinline__ void run_job(LoadF&& load, MathF&& math, Pre&& pre) {
  static_for<0, (D < NSLOT ? D : NSLOT)>(load);
  pin();
  pre();
  pin();
  static_for<0, NSLOT>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    if constexpr (i + D < NSLOT) load(IC<i + D>{});
    pin();
    math(ic);
    pin();
  });
}

template <class F>
struct Once {   // the wave's first job of a layer calls it; later jobs skip it (wave-uniform flag)
  F f;
  bool done = false;
  __device__ __forceinline__ void operator()() {
    if (!done) {
      f();
      done = true;
    }
  }
};
template <class F>
__device__ __forceinline__ Once<F> once(F f) { return Once<F>{f}; }

// ---- X6 form: weights that live in registers ------------------------------------------------------------
// Layer 1's A fragments (main pass: 18 floats per lane; remainder pass: 32, waves 4..7) and layer 2's (three chunks x two
// M-tiles x three bf16 parts x 16 bytes = 72 registers) come straight from global memory (the XCD's L2: 92 KB of packets
// shared by every workgroup), 16 bytes per lane and load, one layer ahead of their use.
// The main pass's fragments are fetched during the previous layer 3 (they are what layer 1 starts with); the remainder pass's,
// used by a wave's LAST job of layer 1, at the start of layer 1 itself: they are not carried across the block loop.
struct A1Regs {
  f32x4 m[5];   // main pass: floats 0..17 = the K-steps' fragments in order (blocks 1..4: two floats per b64 step)
  f32x4 sh;     // shift[4kq .. 4kq+3]
  f32x2 s2;     // shift[16], shift[17]
};
struct A1Rem {
  f32x4 r[8];   // remainder pass (waves 4..7)
};
// Layer 2: a wave computes ONE M-tile (waves 0..3: channels 0..15, waves 4..7: 16..29) -- 36 registers of A fragments.
constexpr int kL2MT = RCED_L2_BOTH ? 2 : 1;   // M-tiles a wave computes
struct A2Regs {
  s16x8 a[kL2MT][kL2Chunks][3];   // [M-tile slot][chunk][part h, m, l]
  f32x4 sh[kL2MT];                // shift[16 mt + 4kq ..]
};
// All of these loads are buffer loads: (the weight stream's buffer resource) + (a wave-uniform byte offset: the block's image +
// every constant, in an SGPR) + (one 32-bit per-lane byte offset, computed once per layer behind opaque()).  No address
// arithmetic on the vector ALU -- as global loads with 64-bit per-lane addresses hipcc spent a two-instruction VALU add on
// most of them, inside layer 1's fp32-MFMA stream where VALU issue is not hidden -- and nothing to hoist out of the tile loop.
typedef __amdgpu_buffer_rsrc_t wrsrc_t;
template <class T>
__device__ __forceinline__ T bld(wrsrc_t rs, unsigned voff, int sofs_floats) {
  if constexpr (sizeof(T) == 16) return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, sofs_floats * 4, 0));
  else return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, sofs_floats * 4, 0));
}
// piece I of the ten: 0..8 = (chunk I / 3, part I % 3), 9 = the shifts.  g2 = float offset of the block's layer-2 image in the
// stream; voff = lane * 16.  SLOT: which of the wave's M-tile slots receives M-tile mt
template <int I, int SLOT = 0>
__device__ __forceinline__ void a2_load_one(A2Regs& A, wrsrc_t rs, int g2, int mt, unsigned voff) {
  if constexpr (I < 9) {
    constexpr int c = I / 3, q = I % 3;
    A.a[SLOT][c][q] = bld<s16x8>(rs, voff, g2 + mt * (3 * 256) + ((c * 2) * 3 + q) * 256);
  } else {
    A.sh[SLOT] = bld<f32x4>(rs, (voff >> 4) & 0x30u, g2 + kG2Data + 16 * mt);   // shift[16 mt + 4 kq ..]: (lane >> 4) * 16 bytes
  }
}
template <int I>
__device__ __forceinline__ void a1_load_rem_one(A1Rem& A, wrsrc_t rs, int g1, unsigned voff) {
  A.r[I] = bld<f32x4>(rs, voff, g1 + kG1Main + I * 256);
}
template <int I>   // piece I of the seven: 0..4 = m[I], 5 = sh, 6 = s2
__device__ __forceinline__ void a1_load_one(A1Regs& A, wrsrc_t rs, int g1, unsigned voff) {
  if constexpr (I < 5) A.m[I] = bld<f32x4>(rs, voff, g1 + I * 256);
  else if constexpr (I == 5) A.sh = bld<f32x4>(rs, (voff >> 4) & 0x30u, g1 + kG1Main + kG1Rem);
  else A.s2 = bld<f32x2>(rs, 0u, g1 + kG1Main + kG1Rem + 16);
}

// Two fp32 values -> their three bf16 parts, packed (low half = the first value): x = h + m + l to 2^-24 (round to nearest
// at every step).  Written element by element: left to itself hipcc joins the two subtractions into one v_pk_add_f32, which
// beside MFMAs costs more than the two v_sub_f32 it replaces (MI355X guide, "packed f32 VALU ... an anti-lever").
// An Inf gives h = Inf and m = l = NaN (Inf - Inf), a NaN three NaNs: non-finite values stay non-finite downstream.
struct P3 {
  unsigned h, m, l;
};
__device__ __forceinline__ float unpk(float v) {   // keep the optimiser from re-pairing the two lanes of a pack
  asm volatile("" : "+v"(v));
  return v;
}
#ifndef RCED_X6_EXP
#define RCED_X6_EXP 0   // timing experiments only (wrong results): 1 = no split arithmetic, 2 = no weight loads in layer 1 (unreliable: the
                        // fragments become undefined values and hipcc deletes work that depends on them), 4 = none in layer 3,
                        // 8 = one plane store instead of three, 16 = no remainder-row reads / merge in layer 2
#endif
#ifndef RCED_SPLIT_DOT2
#define RCED_SPLIT_DOT2 0   // 1 = the residual x - bf16(x) of either element of a pack as ONE v_dot2c_f32_bf16 (pack . (-1, 0) + x) instead of shift / mask
                            // + v_sub_f32: 7 VALU per pair instead of 11, exact for finite values (tools/micro/dot2_split_test.hip: 4 M values,
                            // bit-identical; differs only where the OTHER element of the pack rounds to Inf and below 1e-32) -- and SLOWER: 6.62 against
                            // 6.52 ms (A/B on one box): the dot instruction does not issue at the rate of the two it replaces.  Not adopted.
#endif
__device__ __forceinline__ unsigned opaque_s(unsigned v) {   // a constant the compiler must keep in a register (not an inline operand)
  asm volatile("" : "+s"(v));
  return v;
}
__device__ __forceinline__ P3 split2(float x0, float x1) {
  P3 p;
  if (RCED_X6_EXP & 1) {
    p.h = __builtin_bit_cast(unsigned, x0);
    p.m = __builtin_bit_cast(unsigned, x1);
    p.l = p.h ^ p.m;
    return p;
  }
#if RCED_SPLIT_DOT2
  const bf16x2 k0 = __builtin_bit_cast(bf16x2, opaque_s(0x0000BF80u)), k1 = __builtin_bit_cast(bf16x2, opaque_s(0xBF800000u));   // (-1, 0), (0, -1)
  const bf16x2 bh = {(__bf16)x0, (__bf16)x1};
  const float r0 = __builtin_amdgcn_fdot2_f32_bf16(bh, k0, x0, false), r1 = __builtin_amdgcn_fdot2_f32_bf16(bh, k1, x1, false);
  const bf16x2 bm = {(__bf16)r0, (__bf16)r1};
  const float s0 = __builtin_amdgcn_fdot2_f32_bf16(bm, k0, r0, false), s1 = __builtin_amdgcn_fdot2_f32_bf16(bm, k1, r1, false);
  const bf16x2 bl = {(__bf16)s0, (__bf16)s1};
  p.h = __builtin_bit_cast(unsigned, bh);
  p.m = __builtin_bit_cast(unsigned, bm);
  p.l = __builtin_bit_cast(unsigned, bl);
  return p;
#else
  const bf16x2 bh = {(__bf16)x0, (__bf16)x1};
  p.h = __builtin_bit_cast(unsigned, bh);
  asm volatile("" : "+v"(p.h));   // (seen through, hipcc derives `h << 16` from a second, single-element conversion: +1 VALU per level)
  const float r0 = unpk(x0 - __builtin_bit_cast(float, p.h << 16)), r1 = unpk(x1 - __builtin_bit_cast(float, p.h & 0xffff0000u));
  const bf16x2 bm = {(__bf16)r0, (__bf16)r1};
  p.m = __builtin_bit_cast(unsigned, bm);
  asm volatile("" : "+v"(p.m));
  const float s0 = unpk(r0 - __builtin_bit_cast(float, p.m << 16)), s1 = unpk(r1 - __builtin_bit_cast(float, p.m & 0xffff0000u));
  const bf16x2 bl = {(__bf16)s0, (__bf16)s1};
  p.l = __builtin_bit_cast(unsigned, bl);
  return p;
#endif
}

// ---- layer 1: 8x9, 1 -> 18 on the input rows (block 0, FIRST) / 1x9, 8 -> 18 on B8 (blocks 1..4) -----------
// Blocks 1..4: a slot is a b64 K-step (two k-quads; k = tap*8 + ci): 9 slots per main tile (channels 0..15 of 16
// pixels), 16 per remainder tile (channels 16,17 of 128 pixels as 8 phases x 2 channels, K = 16 taps).  Block 0: a slot
// is a b32 K-step (one k-quad = time taps 4*ih + kq at one frequency tap; B operand straight out of the staged input
// rows X0): 18 / 32 slots.  Main tiles run in pairs (one A fragment for both); waves 0, 1 (tile 32 / 31) and wave 7
// (three regular tiles) have one single tile.
template <class M, bool FIRST>
struct L1Geo {
  static constexpr int SR = FIRST ? 32 : 16, SM = FIRST ? 18 : 9;      // slots of a remainder / main job
  static constexpr int kAStep = FIRST ? 64 : 128;                       // floats per K-step of A fragments
  static constexpr int kTR = FIRST ? 128 * 4 : M::kT1R;                 // read-side byte stride between a wave's regular tiles
  static constexpr int kTileR = FIRST ? 16 * 4 : 16 * kB8S * 4;         // ... between adjacent 16-pixel tiles
  static constexpr int D = FIRST ? 2 * RCED_D1 : RCED_D1;
  // byte offset of K-step st inside a lane's window: blocks 1..4: one pixel per b64 step; block 0: (ih, j)
  static constexpr int koff(int st, int per) { return FIRST ? ((st / per) * 4 * kS + st % per) * 4 : kB8S * 4 * st; }
};
// K-step I of the A fragment kept in registers (X6 form): FIRST: one float per step, else two
template <bool FIRST, bool REM, int I>
__device__ __forceinline__ f32x2 a1_frag(const A1Regs& A, const A1Rem& R) {
  if constexpr (FIRST) {
    if constexpr (REM) return f32x2{R.r[I / 4][I % 4], 0.f};
    else return f32x2{A.m[I / 4][I % 4], 0.f};
  } else {
    constexpr int i0 = 2 * I, i1 = 2 * I + 1;
    if constexpr (REM) return f32x2{R.r[i0 / 4][i0 % 4], R.r[i1 / 4][i1 % 4]};
    else return f32x2{A.m[i0 / 4][i0 % 4], A.m[i1 / 4][i1 % 4]};
  }
}

struct NoSpread {
  template <class T>
  __device__ __forceinline__ void operator()(T) const {}
};
// `sp(IC<k>)`, k < 18: the layer's spread hook (see layer1); a job with SM = 9 slots calls two per slot
template <class M, bool FIRST, int NT, bool REM, class Pre, class Sp = NoSpread>   // NT tiles in lockstep (1 or 2); REM: remainder tile (NT = 1)
__device__ __forceinline__ void l1_job(unsigned wa, const A1Regs& A, const A1Rem& AR, unsigned rdA, unsigned rdB, f32x4 init, f32x4 (&acc)[2],
                                       Pre& pre, Sp sp = Sp{}) {
  using G = L1Geo<M, FIRST>;
  constexpr int NS = REM ? G::SR : G::SM, D = G::D, RING = D + 1, PER = REM ? 16 : 9;
  f32x2 a[RING], b[RING][NT];
  acc[0] = init;
  if constexpr (NT > 1) acc[1] = init;
  run_job<NS, D>(
      [&](auto ic) {
        constexpr int i = decltype(ic)::value, r = i % RING, aoff = ((REM ? kW1Main : 0) + i * G::kAStep) * 4;
        if constexpr (FIRST) {
          if constexpr (!M::kX6) a[r].x = lds_ld<float>(wa, aoff);
          b[r][0].x = lds_ld<float>(rdA, G::koff(i, PER));
          if constexpr (NT > 1) b[r][1].x = lds_ld<float>(rdB, G::koff(i, PER));
        } else {
          if constexpr (!M::kX6) a[r] = lds_ld<f32x2>(wa, aoff);
          b[r][0] = lds_ld<f32x2>(rdA, G::koff(i, PER));
          if constexpr (NT > 1) b[r][1] = lds_ld<f32x2>(rdB, G::koff(i, PER));
        }
      },
      [&](auto ic) {
        constexpr int i = decltype(ic)::value, r = i % RING;
        f32x2 av;
        if constexpr (M::kX6) av = a1_frag<FIRST, REM, i>(A, AR);
        else av = a[r];
        acc[0] = mfma(av.x, b[r][0].x, acc[0]);
        if constexpr (NT > 1) acc[1] = mfma(av.x, b[r][1].x, acc[1]);
        if constexpr (!FIRST) {
          acc[0] = mfma(av.y, b[r][0].y, acc[0]);
          if constexpr (NT > 1) acc[1] = mfma(av.y, b[r][1].y, acc[1]);
        }
        if constexpr (NS == 9) {
          sp(IC<2 * i>{});
          sp(IC<2 * i + 1>{});
        } else if constexpr (i < 18) {
          sp(IC<i>{});
        }
      },
      pre);
}

// Store of a main tile's channels 4kq..4kq+3 (masked: the tile has gap pixels; wave-uniform).  F32 form: [pixel][18] fp32.
// X6 form: ReLU, then the three-part split -- ONCE per element, here where it is produced -- and one 8-byte store per plane.
template <class M>
__device__ __forceinline__ void l1_store(const Lane& L, f32x4 acc4, unsigned wr, int off, bool masked, int vb) {
#if defined(RCED_T_EXP) && (RCED_T_EXP & 4096)
  if (acc4.x == 12345.678f) lds_st<float>(wr, off, acc4.y);   // timing experiment (wrong results): layer 1 without its epilogues
  return;
#endif
  const f32x4 v = relu4(acc4);
  if constexpr (M::kX6) {
    const P3 p01 = split2(v.x, v.y), p23 = split2(v.z, v.w);
    if (!masked || vbit(L, vb)) {
      lds_st<u32x2>(wr, off, u32x2{p01.h, p23.h});
      if (!(RCED_X6_EXP & 8)) {
        lds_st<u32x2>(wr, off + M::kPlaneBytes, u32x2{p01.m, p23.m});
        lds_st<u32x2>(wr, off + 2 * M::kPlaneBytes, u32x2{p01.l, p23.l});
      }
    }
  } else {
    if (!masked || vbit(L, vb)) {   // predication (exec mask from SGPRs), not a branch around two copies of the stores
      lds_st<f32x2>(wr, off, f32x2{v.x, v.y});
      lds_st<f32x2>(wr, off + 8, f32x2{v.z, v.w});
    }
  }
}

// `dma` runs behind the first job's first operand reads; `sp(IC<k>)`, k = 0..17, is called once each from consecutive slots
// of the wave's pair job(s) (every wave's FIRST job; the X6 form issues one register-bound weight load per slot there: issued
// in one burst by eight waves they fill the vector-memory queue and block the waves' MFMAs behind them)
// `late()` runs between the pair job(s) and the wave's last jobs.
template <class M, bool FIRST, class Dma, class Sp, class Late>
__device__ __forceinline__ void layer1(const Lane& L, unsigned wbase, const A1Regs& A, const A1Rem& AR, int wave, Dma dma, Sp sp, Late late DET_ARG) {
  using G = L1Geo<M, FIRST>;
  DET_BEGIN();
  const unsigned wa = wbase + (FIRST ? L.a4 : L.a8);     // F32 form: A fragments in LDS: main [s][lane], remainder from kW1Main
  f32x4 sh;
  f32x2 s2;
  if constexpr (M::kX6) {
    sh = A.sh;
    s2 = A.s2;
  } else {
    sh = lds_ld<f32x4>(wbase + L.kq16, kW1Data * 4);
    s2 = lds_ld<f32x2>(wbase, (kW1Data + 16) * 4);
  }
  f32x4 acc[2];
  auto pre = once(dma);
  const unsigned rd = FIRST ? L.rd0 : L.rd1, rdb = FIRST ? L.rd0b : L.rd1b;
  constexpr int kTW = M::kT1W;
  // The wave's jobs, independent of each other.  Order (RCED_L1_ORDER, A/B on one box): pairs first, then the single tile,
  // then the remainder tiles is 0.6 % faster than the reverse -- the last job's epilogue is what sits exposed in front of
  // the layer's barrier, and a remainder tile's is two 8-byte stores against the four of a pair.
  auto do_rem = [&] {
  // ---- remainder tiles: waves 4, 5, 6 -> tiles 0, 1, 2; wave 7 -> tiles 3 and 4
  const int nrem = wave < 4 ? 0 : wave == 7 ? 2 : 1;
  unsigned rdr = FIRST ? L.rd0r : L.rd1r, wrr = L.wr1r, wrl = L.wr1rl;
  int xr = wave == 7 ? 3 : wave - 4, vb = kVRem;
#pragma unroll 1
  for (int r = 0; r < nrem; ++r) {
    l1_job<M, FIRST, 1, true>(wa, A, AR, rdr, 0u, f32x4{s2.x, s2.y, s2.x, s2.y}, acc, pre);
    const f32x4 v = relu4(acc[0]);   // rows 4kq+jj = (phase 2kq + (jj>>1), channel 16 + (jj&1)): two pixels x channels 16,17
    // every remainder tile but tile 0 contains gap pixels (tile 4 also runs past the tile): those are never written
    const bool va = xr == 0 || vbit(L, vb), vbb = (!M::kFused && xr == 0) || vbit(L, vb + 1);
    if constexpr (M::kFused) {
      const P3 pa = split2(v.x, v.y), pb = split2(v.z, v.w);
      if (va) {
        lds_st<unsigned>(wrr, 0, pa.h);
        lds_st<unsigned>(wrr, M::kPlaneBytes, pa.m);
        lds_st<unsigned>(wrr, 2 * M::kPlaneBytes, pa.l);
      }
      if (vbb) {
        lds_st<unsigned>(wrr, 4, pb.h);
        lds_st<unsigned>(wrr, M::kPlaneBytes + 4, pb.m);
        lds_st<unsigned>(wrr, 2 * M::kPlaneBytes + 4, pb.l);
      }
      wrr += 112 * 4;
    } else if constexpr (M::kX6) {
      const P3 pa = split2(v.x, v.y), pb = split2(v.z, v.w);
      if (va) {
        lds_st<u32x2>(wrr, 0, u32x2{pa.h, pa.m});
        lds_st<unsigned>(wrl, 0, pa.l);
      }
      if (vbb) {
        lds_st<u32x2>(wrr, 8, u32x2{pb.h, pb.m});
        lds_st<unsigned>(wrl, 4, pb.l);
      }
      wrr += 128 * 8;
      wrl += 128 * 4;
    } else {
      if (va) lds_st<f32x2>(wrr, 0, f32x2{v.x, v.y});
      if (vbb) lds_st<f32x2>(wrr, 18 * 4, f32x2{v.z, v.w});
      wrr += 8 * 16 * 18 * 4;
    }
    rdr += (M::kFused ? 7 : 8) * G::kTileR;   // wave 7's second tile: 4 = 3 + 1
    xr += 1;
    vb += 2;
  }
  };
  auto do_single = [&] {
  // ---- the single main tile: waves 0 / 1 -> tile 32 / 31 (no gap pixels); wave 7 -> its third regular tile (23)
  if (wave < 2 || wave == 7) {
    const int dt = wave == 0 ? 32 : wave == 1 ? 30 : 16;   // tiles away from regular tile `wave`
    l1_job<M, FIRST, 1, false>(wa, A, AR, rd + dt * G::kTileR, 0u, sh, acc, pre);
    l1_store<M>(L, acc[0], L.wr1 + dt * M::kTileB18, 0, false, 0);
  }
  };
  auto do_pairs = [&] {
  // ---- pairs of regular tiles: (wave, wave+8), (wave+16, wave+24) as one stream with pair 0's stores between pair 1's
  //      MFMAs; wave 7 has only the first pair
  if (wave == 7) {
    l1_job<M, FIRST, 2, false>(wa, A, AR, rd, rdb, sh, acc, pre, sp);
    l1_store<M>(L, acc[0], L.wr1, 0, false, 0);          // tiles 7, 15: no gap pixels
    l1_store<M>(L, acc[1], L.wr1, kTW, false, 0);
  } else {
    constexpr int SM = G::SM, NT = 2 * SM, D = G::D, RING = D + 1;
    f32x2 a[RING], b[RING][2];
    f32x4 acc2[2][2];   // [pair][tile]
    const bool g1 = tile_has_gap(wave + 8), g2 = tile_has_gap(wave + 16), g3 = tile_has_gap(wave + 24);   // tiles 8, 16, 24
    run_job<NT, D>(
        [&](auto ic) {
          constexpr int i = decltype(ic)::value, r = i % RING, p = i / SM, st = i % SM, aoff = st * G::kAStep * 4;
          if constexpr (FIRST) {
            if constexpr (!M::kX6) a[r].x = lds_ld<float>(wa, aoff);
            b[r][0].x = lds_ld<float>(rd, 2 * p * G::kTR + G::koff(st, 9));
            b[r][1].x = lds_ld<float>(rdb, 2 * p * G::kTR + G::koff(st, 9));
          } else {
            if constexpr (!M::kX6) a[r] = lds_ld<f32x2>(wa, aoff);
            b[r][0] = lds_ld<f32x2>(rd, 2 * p * G::kTR + G::koff(st, 9));
            b[r][1] = lds_ld<f32x2>(rdb, 2 * p * G::kTR + G::koff(st, 9));
          }
        },
        [&](auto ic) {
          constexpr int i = decltype(ic)::value, r = i % RING, p = i / SM, st = i % SM;
          if constexpr (st == 0) acc2[p][0] = acc2[p][1] = sh;
          f32x2 av;
          if constexpr (M::kX6) av = a1_frag<FIRST, false, st>(A, AR);
          else av = a[r];
          acc2[p][0] = mfma(av.x, b[r][0].x, acc2[p][0]);
          acc2[p][1] = mfma(av.x, b[r][1].x, acc2[p][1]);
          if constexpr (!FIRST) {
            acc2[p][0] = mfma(av.y, b[r][0].y, acc2[p][0]);
            acc2[p][1] = mfma(av.y, b[r][1].y, acc2[p][1]);
          }
          if constexpr (p == 1 && st == 1) l1_store<M>(L, acc2[0][0], L.wr1, 0, false, kVMain);
          if constexpr (p == 1 && st == 3) l1_store<M>(L, acc2[0][1], L.wr1, kTW, g1, kVMain + 1);
          if constexpr (i < 18) sp(IC<i>{});
        },
        pre);
    l1_store<M>(L, acc2[1][0], L.wr1, 2 * kTW, g2, kVMain + 2);
    l1_store<M>(L, acc2[1][1], L.wr1, 3 * kTW, g3, kVMain + 3);
  }
  };
#if RCED_L1_ORDER == 1
  do_pairs();
  DET(6);
  late();
  do_single();
  DET(5);
  do_rem();
  DET(4);
#else
  do_rem();
  DET(4);
  do_single();
  DET(5);
  do_pairs();
  DET(6);
  late();
#endif
}

// ---- layer 2: 1x5, 18 -> 30 (two M-tiles) ------------------------------------------------------------
// Tile 32 (pixels 512..527) is cut in four, one piece per SIMD, so that the layer's MFMA count is (about) the same on every
// SIMD: M-tile XM x K-part.  Waves 0 / 1 are the helpers (the first K-part of M-tile 0 / 1), waves 2 / 3 the reducers (the
// rest of M-tile 0 / 1; they add the helper's partial sums and own the epilogue).  The share is the wave's FIRST job: a helper
// publishes it through LDS scratch + a tagged flag word before its regular tiles, and the reducer picks it up after
// its last job, when it has long been there.  Scratch: in the B8 buffer, which is dead during layer 2 (layer 3
// rewrites every real pixel of it).
#ifndef RCED_L2_CUT
#define RCED_L2_CUT 6
#endif
constexpr int kL2Cut = RCED_L2_CUT;   // F32 form: the helper takes b64 slots [0, kL2Cut), the reducer [kL2Cut, 11) + the tail

// one M-tile of one tile: ReLU, [pixel][30] stores; lanes kq = 3 of M-tile 1 hold channels 28,29 and the padding 30,31
template <int MT>
__device__ __forceinline__ void l2_store(const Lane& L, f32x4 acc4, unsigned wr, int off, bool masked, int vb) {
  const f32x4 v = relu4(acc4);
  if (!masked || vbit(L, vb)) {
    lds_st<f32x2>(wr, off + 64 * MT, f32x2{v.x, v.y});
    if (MT == 0 || vbit(L, kVLt48)) lds_st<f32x2>(wr, off + 64 * MT + 8, f32x2{v.z, v.w});
  }
}
// reducers (waves 2, 3): add the helper's share, store tile 32 (pixels 512..527: no gap inside)
template <class M>
__device__ __forceinline__ void l2_reduce(const Lane& L, unsigned lds0, int wave, unsigned tag, unsigned* err, f32x4 accx, f32x4 part,
                                          unsigned pflag) {
  if (wave == 2 || wave == 3) {
    const int xm = wave - 2;
    if (!__builtin_amdgcn_readfirstlane(pflag == tag)) {   // not there yet when fetched (not seen in practice)
      flag_wait(lds0 + (M::kFlag2Off + xm) * 4, tag, err, 2u);
      part = lds_ld<f32x4>(lds0 + L.scr + xm * 1024, M::kScratch2Off * 4);
    }
    const f32x4 v = accx + part;
    const unsigned wrx = L.wr2 + (32 - wave) * (16 * 30 * 4);
    if (xm == 0) l2_store<0>(L, v, wrx, 0, false, 0);
    else l2_store<1>(L, v, wrx, 0, false, 0);
  }
}

// ---- F32 form: every wave has four regular tiles, walked as two pair jobs (11 b64 slots + the b32 tail; both tiles, both
// M-tiles: 8 / 4 MFMAs per slot on four accumulation chains, two A fragments for both tiles)
template <int XM, bool HELPER, class Pre>   // the share of tile 32: M-tile XM, slots [0, kL2Cut) (helper) or [kL2Cut, 11) + tail
__device__ __forceinline__ f32x4 l2_share(unsigned wa, unsigned wt, unsigned rdx, unsigned rdxt, f32x4 init, Pre& pre) {
  constexpr int S0 = HELPER ? 0 : kL2Cut, NS = HELPER ? kL2Cut : kL2Steps + 1 - kL2Cut, D = RCED_D2, RING = D + 1;
  f32x2 a[RING], b[RING];
  f32x4 acc = init;
  run_job<NS, D>(
      [&](auto ic) {
        constexpr int i = decltype(ic)::value, r = i % RING, st = S0 + i;
        if constexpr (st < kL2Steps) {
          a[r] = lds_ld<f32x2>(wa, (st * 2 + XM) * 128 * 4);
          b[r] = lds_ld<f32x2>(rdx, 32 * st);
        } else {
          a[r].x = lds_ld<float>(wt, (kL2Steps * 2 * 128 + XM * 64) * 4);
          b[r].x = lds_ld<float>(rdxt, 0);
        }
      },
      [&](auto ic) {
        constexpr int i = decltype(ic)::value, r = i % RING;
        acc = mfma(a[r].x, b[r].x, acc);
        if constexpr (S0 + i < kL2Steps) acc = mfma(a[r].y, b[r].y, acc);
      },
      pre);
  return acc;
}

template <class M, class Dma>
__device__ __forceinline__ void layer2_f32(const Lane& L, unsigned lds0, unsigned wbase, int wave, unsigned tag, unsigned* err, Dma dma DET_ARG) {
  constexpr int D = RCED_D2, RING = D + 1, NS = kL2Steps + 1;
  constexpr int kT2R = M::kT2R, kT2W = M::kT2W;
  DET_BEGIN();
  const unsigned wa = wbase + L.a8, wt = wbase + L.a4;
  f32x4 sh[2];
  sh[0] = lds_ld<f32x4>(wbase + L.kq16, kW2Data * 4);
  sh[1] = lds_ld<f32x4>(wbase + L.kq16, (kW2Data + 16) * 4);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // ---- the share of tile 32 (waves 0..3), first
  f32x4 accx = zero4, part = zero4;   // part / pflag: the reducers' copy of their helper's partial sums and flag word
  unsigned pflag = 0u;
  auto pre = once(dma);
  if (wave < 4) {
    const unsigned rdx = L.rd2 + (32 - wave) * (16 * 18 * 4), rdxt = L.rd2t + (32 - wave) * (16 * 18 * 4);
    if (wave == 0) accx = l2_share<0, true>(wa, wt, rdx, rdxt, zero4, pre);        // a helper's share starts from zero,
    else if (wave == 1) accx = l2_share<1, true>(wa, wt, rdx, rdxt, zero4, pre);
    else if (wave == 2) accx = l2_share<0, false>(wa, wt, rdx, rdxt, sh[0], pre);  // the reducer's from the shift
    else accx = l2_share<1, false>(wa, wt, rdx, rdxt, sh[1], pre);
    if (wave < 2) {   // publish (LDS operations of a wave execute in order: data, then flag)
      lds_st<f32x4>(lds0 + L.scr + wave * 1024, M::kScratch2Off * 4, accx);
      cbar();
      if (L.a4 == 0) lds_poke_a(lds0 + (M::kFlag2Off + wave) * 4, tag);
    }
  }
  DET(7);
  // ---- two pair jobs, tiles (wave, wave+8) and (wave+16, wave+24), as ONE stream: pair 0's stores ride between pair
  //      1's MFMAs (LDS stores are slow, ~85 B/clk per CU: issued in one burst they delay the next operand reads)
  {
    constexpr int NT = 2 * NS;
    f32x2 a[RING][2], b[RING][2];
    f32x4 acc[2][2][2];   // [pair][tile][M-tile]
    const bool g1 = tile_has_gap(wave + 8), g2 = tile_has_gap(wave + 16), g3 = tile_has_gap(wave + 24);   // tiles 8, 16, 24 (wave 0)
    run_job<NT, D>(
        [&](auto ic) {
          constexpr int i = decltype(ic)::value, r = i % RING, p = i / NS, st = i % NS;
          if constexpr (st < kL2Steps) {
            a[r][0] = lds_ld<f32x2>(wa, (st * 2 + 0) * 128 * 4);
            a[r][1] = lds_ld<f32x2>(wa, (st * 2 + 1) * 128 * 4);
            b[r][0] = lds_ld<f32x2>(L.rd2, 2 * p * kT2R + 32 * st);
            b[r][1] = lds_ld<f32x2>(L.rd2b, 2 * p * kT2R + 32 * st);
          } else {
            a[r][0].x = lds_ld<float>(wt, (kL2Steps * 2 * 128) * 4);
            a[r][1].x = lds_ld<float>(wt, (kL2Steps * 2 * 128 + 64) * 4);
            b[r][0].x = lds_ld<float>(L.rd2t, 2 * p * kT2R);
            b[r][1].x = lds_ld<float>(L.rd2tb, 2 * p * kT2R);
          }
        },
        [&](auto ic) {
          constexpr int i = decltype(ic)::value, r = i % RING, p = i / NS, st = i % NS;
          if constexpr (st == 0) {
            acc[p][0][0] = acc[p][1][0] = sh[0];
            acc[p][0][1] = acc[p][1][1] = sh[1];
          }
          acc[p][0][0] = mfma(a[r][0].x, b[r][0].x, acc[p][0][0]);
          acc[p][0][1] = mfma(a[r][1].x, b[r][0].x, acc[p][0][1]);
          acc[p][1][0] = mfma(a[r][0].x, b[r][1].x, acc[p][1][0]);
          acc[p][1][1] = mfma(a[r][1].x, b[r][1].x, acc[p][1][1]);
          if constexpr (st < kL2Steps) {
            acc[p][0][0] = mfma(a[r][0].y, b[r][0].y, acc[p][0][0]);
            acc[p][0][1] = mfma(a[r][1].y, b[r][0].y, acc[p][0][1]);
            acc[p][1][0] = mfma(a[r][0].y, b[r][1].y, acc[p][1][0]);
            acc[p][1][1] = mfma(a[r][1].y, b[r][1].y, acc[p][1][1]);
          }
          if constexpr (p == 1 && st == 1) l2_store<0>(L, acc[0][0][0], L.wr2, 0, false, kVMain);       // tile `wave`: no gap
          if constexpr (p == 1 && st == 3) l2_store<1>(L, acc[0][0][1], L.wr2, 0, false, kVMain);
          if constexpr (p == 1 && st == 5) l2_store<0>(L, acc[0][1][0], L.wr2, kT2W, g1, kVMain + 1);
          if constexpr (p == 1 && st == 7) l2_store<1>(L, acc[0][1][1], L.wr2, kT2W, g1, kVMain + 1);
          if constexpr (p == 1 && st == 9) {   // reducers: the helper's flag and partial sums, fetched inside the stream (see layer 3)
            if (wave == 2 || wave == 3) {
              pflag = lds_peek_a(lds0 + (M::kFlag2Off + wave - 2) * 4);
              cbar();
              part = lds_ld<f32x4>(lds0 + L.scr + (wave - 2) * 1024, M::kScratch2Off * 4);
            }
          }
        },
        pre);
    l2_store<0>(L, acc[1][0][0], L.wr2, 2 * kT2W, g2, kVMain + 2);
    l2_store<1>(L, acc[1][0][1], L.wr2, 2 * kT2W, g2, kVMain + 2);
    l2_store<0>(L, acc[1][1][0], L.wr2, 3 * kT2W, g3, kVMain + 3);
    l2_store<1>(L, acc[1][1][1], L.wr2, 3 * kT2W, g3, kVMain + 3);
  }
  l2_reduce<M>(L, lds0, wave, tag, err, accx, part, pflag);
}

// ---- the three-part operand of one K = 32 chunk and its six products (kernels_fused_v3_l23.h, kernels_fused_v3_allx6.h) ----
struct Parts {
  s16x8 h, m, l;
};
// the six products of one chunk, smallest first
__device__ __forceinline__ f32x4 l2x_mma(const s16x8 (&a)[3], const Parts& b, f32x4 acc) {
  acc = mfma32(a[1], b.m, acc);
  acc = mfma32(a[2], b.h, acc);
  acc = mfma32(a[0], b.l, acc);
  acc = mfma32(a[1], b.h, acc);
  acc = mfma32(a[0], b.m, acc);
  acc = mfma32(a[0], b.h, acc);
  return acc;
}
#if RCED_V3_LEGACY_FORMS
#include "kernels_fused_v3_legacy.h"
#endif
// ---- layer 3: 1x9, 30 -> 8 on pixel pairs ------------------------------------------------------------
// Rows = 2 pixel phases x 8 channels, K = 10 taps x 30 = 300 (37 b64 slots + the b32 tail).  Every wave has two regular
// pair tiles, run in lockstep (one A fragment per slot for both, one accumulation chain per tile).
// Pair tile 16 is split ALONG K in four, one part per SIMD (waves 0..3): the reducer (wave 0: slots [0,8), owns the
// epilogue and the skip registers) and three helpers (waves 1..3: [8,18), [18,28), [28,37) + tail).  The share is each
// wave's first job; partial sums go through 1-KiB scratch areas in the (dead during layer 3) B18 buffer + flag words.
#ifndef RCED_L3_CUTS
#define RCED_L3_CUTS 8, 18, 28   // the reducer (wave 0) takes the smallest share: it also collects and stores the tile (A/B: -0.25 %)
#endif
#ifndef RCED_L3_FETCH
#define RCED_L3_FETCH 24
#endif
constexpr int kL3CutsArr[3] = {RCED_L3_CUTS};
constexpr int kL3Cut1 = kL3CutsArr[0], kL3Cut2 = kL3CutsArr[1], kL3Cut3 = kL3CutsArr[2];
constexpr int kL3Fetch = RCED_L3_FETCH;   // slot of the reducer's regular job at which it fetches the helpers' flags and partial sums

template <int S0, int S1, class Pre>   // the share of pair tile 16: slots [S0, S1) (slot kL3Steps = the tail)
__device__ __forceinline__ f32x4 l3_share(unsigned wa, unsigned wt, unsigned rdx, unsigned rdxt, f32x4 init, Pre& pre) {
  constexpr int NS = S1 - S0, D = RCED_D3, RING = D + 1;
  f32x2 a[RING], b[RING];
  f32x4 acc = init;
  run_job<NS, D>(
      [&](auto ic) {
        constexpr int i = decltype(ic)::value, r = i % RING, st = S0 + i;
        if constexpr (st < kL3Steps) {
          a[r] = lds_ld<f32x2>(wa, st * 128 * 4);
          b[r] = lds_ld<f32x2>(rdx, 32 * st);
        } else {
          a[r].x = lds_ld<float>(wt, kL3Steps * 128 * 4);
          b[r].x = lds_ld<float>(rdxt, 0);
        }
      },
      [&](auto ic) {
        constexpr int i = decltype(ic)::value, r = i % RING;
        acc = mfma(a[r].x, b[r].x, acc);
        if constexpr (S0 + i < kL3Steps) acc = mfma(a[r].y, b[r].y, acc);
      },
      pre);
  return acc;
}

// Layer 3's epilogue for block BLK: ReLU, the block skips, stores.  Blocks 0 / 1 keep their outputs (CE1 / CE2) in the
// skip registers, blocks 3 / 4 add CE2 / CE1 after the ReLU, block 4 stores to the H image (decode_final's input)
// instead of B8.  Gap / past-the-tile pixels are never written (they stay zero): every store is predicated on the
// lane's validity bit -- the compare is loop-invariant, hipcc keeps it as an exec mask in SGPRs -- rather than put
// behind a wave-uniform "does this tile have a gap" branch.  Values of gap lanes in the skip registers are whatever was
// computed: they only ever meet gap pixels again.
template <class M, int BLK>
__device__ __forceinline__ void l3_epilogue(const Lane& L, int wave, f32x4 (&acc)[3], f32x4 (&skip_ce1)[3],
                                            f32x4 (&skip_ce2)[3]) {
  static_for<0, 3>([&](auto tc) {
    constexpr int t = decltype(tc)::value;
    if (t == 2 && wave != 0) return;    // pair tile 16 belongs to wave 0
    f32x4 v = relu4(acc[t]);
    if constexpr (BLK == 3) v += skip_ce2[t];
    if constexpr (BLK == 4) v += skip_ce1[t];
    if constexpr (BLK == 0) skip_ce1[t] = v;
    if constexpr (BLK == 1) skip_ce2[t] = v;
    constexpr int vb = t < 2 ? kVL3 + t : kVL3X;
    unsigned wr;
    int off = 0;
    if constexpr (BLK < 4) {
      wr = L.wr3;
      off = t < 2 ? t * M::kT3W : 16 * (32 * kB8S * 4);
    } else {
      wr = t == 0 ? L.wh0 : t == 1 ? L.wh1 : L.whx;
    }
    if (vbit(L, vb)) {
      lds_st<f32x2>(wr, off, f32x2{v.x, v.y});
      lds_st<f32x2>(wr, off + 8, f32x2{v.z, v.w});
    }
  });
}

// LAST: block 4's instance (peeled out of the block loop: its epilogue feeds decode_final).  `sp(IC<k>)`, k = 0..6, is called
// from every fourth slot of the regular job (X6 form: the next layer 1's main-pass A fragments, one load at a time).
template <class M, bool LAST, class Dma, class Sp>
__device__ __forceinline__ void layer3(const Params& P, const Lane& L, unsigned lds0, unsigned wbase, int blk, int wave,
                                       unsigned tag, f32x4 (&skip_ce1)[3], f32x4 (&skip_ce2)[3], Dma dma, Sp sp DET_ARG) {
  constexpr int D = RCED_D3, RING = D + 1, NS = kL3Steps + 1;
  DET_BEGIN();
  const unsigned wa = wbase + L.a8, wt = wbase + L.a4;
  const f32x4 sh = lds_ld<f32x4>(wbase + (L.kq16 & 16), kW3Data * 4);   // shift[4*(kq&1) ..]
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[3] = {sh, sh, zero4};   // [2]: pair tile 16 (wave 0)
  f32x4 part[3] = {zero4, zero4, zero4};   // wave 0: the helpers' partial sums of pair tile 16 and their flag words
  unsigned pflag[3];
  auto pre = once(dma);
  // ---- the share of pair tile 16 (waves 0..3), first
  if (wave < 4) {
    const unsigned rdx = L.rd3 + (16 - wave) * (16 * 60 * 4), rdxt = L.rd3t + (16 - wave) * (16 * 60 * 4);
    if (wave == 0) acc[2] = l3_share<0, kL3Cut1>(wa, wt, rdx, rdxt, sh, pre);
    else if (wave == 1) acc[2] = l3_share<kL3Cut1, kL3Cut2>(wa, wt, rdx, rdxt, zero4, pre);
    else if (wave == 2) acc[2] = l3_share<kL3Cut2, kL3Cut3>(wa, wt, rdx, rdxt, zero4, pre);
    else acc[2] = l3_share<kL3Cut3, kL3Steps + 1>(wa, wt, rdx, rdxt, zero4, pre);
    if (wave > 0) {   // publish the partial sums
      lds_st<f32x4>(lds0 + L.scr + (wave - 1) * 1024, M::kScratchOff * 4, acc[2]);
      cbar();
      if (L.a4 == 0) lds_poke_a(lds0 + (M::kFlagOff + wave - 1) * 4, tag);
    }
  }
  DET(0);
  // ---- the two regular pair tiles
  {
    f32x2 a[RING], b[RING][2];
    [[maybe_unused]] f32x4 accb[2] = {zero4, zero4};   // second chain of each tile (the slot's second k-quad)
    pflag[0] = pflag[1] = pflag[2] = 0u;
    run_job<NS, D>(
        [&](auto ic) {
          constexpr int i = decltype(ic)::value, r = i % RING;
          if constexpr (i < kL3Steps) {
            a[r] = lds_ld<f32x2>(wa, i * 128 * 4);
            b[r][0] = lds_ld<f32x2>(L.rd3, 32 * i);
            b[r][1] = lds_ld<f32x2>(L.rd3b, 32 * i);
          } else {
            a[r].x = lds_ld<float>(wt, kL3Steps * 128 * 4);
            b[r][0].x = lds_ld<float>(L.rd3t, 0);
            b[r][1].x = lds_ld<float>(L.rd3tb, 0);
          }
        },
        [&](auto ic) {
          constexpr int i = decltype(ic)::value, r = i % RING;
          acc[0] = mfma(a[r].x, b[r][0].x, acc[0]);
          acc[1] = mfma(a[r].x, b[r][1].x, acc[1]);
          if constexpr (i < kL3Steps) {
#if RCED_L3_CHAINS == 4
            accb[0] = mfma(a[r].y, b[r][0].y, accb[0]);
            accb[1] = mfma(a[r].y, b[r][1].y, accb[1]);
#else
            acc[0] = mfma(a[r].y, b[r][0].y, acc[0]);
            acc[1] = mfma(a[r].y, b[r][1].y, acc[1]);
#endif
          }
          // The helpers published their shares of pair tile 16 before their own regular tiles, i.e. long ago: the reducer
          // fetches flags and partial sums HERE, as three more loads in its operand stream, instead of in three serial
          // LDS round trips after its last MFMA, where every other wave of the workgroup waits for it at the barrier.
          if constexpr (i % 4 == 1 && i / 4 < 7) sp(IC<i / 4>{});
          if constexpr (i == kL3Fetch) {
            if (wave == 0) {
#pragma unroll
              for (int h = 0; h < 3; ++h) {
                pflag[h] = lds_peek_a(lds0 + (M::kFlagOff + h) * 4);
                cbar();
                part[h] = lds_ld<f32x4>(lds0 + L.scr + h * 1024, M::kScratchOff * 4);
              }
            }
          }
        },
        pre);
#if RCED_L3_CHAINS == 4
    acc[0] += accb[0];
    acc[1] += accb[1];
#endif
  }
  DET(1);
  if (wave == 0) {   // collect the helpers' shares
    const bool early = pflag[0] == tag && pflag[1] == tag && pflag[2] == tag;   // the same words in every lane
    if (!__builtin_amdgcn_readfirstlane(early)) {   // not there yet at slot kL3Fetch (not seen in practice): wait, re-read
#pragma unroll
      for (int h = 0; h < 3; ++h) flag_wait(lds0 + (M::kFlagOff + h) * 4, tag, P.err, 4u);
#pragma unroll
      for (int h = 0; h < 3; ++h) part[h] = lds_ld<f32x4>(lds0 + L.scr + h * 1024, M::kScratchOff * 4);
    }
    acc[2] += part[0];   // fixed order: the result does not depend on which path was taken
    acc[2] += part[1];
    acc[2] += part[2];
  }
  DET(2);
  // Block-dependent work (model.py:84-88: CE1 / CE2 outputs are kept, and added to CD2 / CD1 AFTER the ReLU): one
  // wave-uniform switch on blk, then straight-line code.  This epilogue is every wave's tail in front of the layer's
  // barrier -- nothing overlaps it -- so it is kept free of branches: as one generic body with `blk ==` tests inside
  // it was ~40 scalar branches and 24 v_cndmask per wave (the skip registers were merged at every join).
  if constexpr (LAST) l3_epilogue<M, 4>(L, wave, acc, skip_ce1, skip_ce2);
  else if (blk == 0) l3_epilogue<M, 0>(L, wave, acc, skip_ce1, skip_ce2);
  else if (blk == 1) l3_epilogue<M, 1>(L, wave, acc, skip_ce1, skip_ce2);
  else if (blk == 2) l3_epilogue<M, 2>(L, wave, acc, skip_ce1, skip_ce2);
  else l3_epilogue<M, 3>(L, wave, acc, skip_ce1, skip_ce2);
  DET(3);
}

#include "kernels_fused_v3_l23.h"

// ---- decode_final inside the kernel (layout and decomposition: see above) ---------------------------------
struct FinA {
  f32x2 a[kFinRun];   // this wave's 18 K-steps of A fragments
};
__device__ __forceinline__ void fin_prefetch(const Params& P, int wave, int lane, FinA& A) {
  lane = opaque(lane);
  const f32x2* src = reinterpret_cast<const f32x2*>(P.fin) + (size_t)(kFinRun * wave) * 64 + lane;
#pragma unroll
  for (int s = 0; s < kFinRun; ++s) A.a[s] = src[s * 64];
}
// zero the five 64-pixel pads of the H image (B18 holds layer-1 / layer-2 leftovers); block 4's layer 3 writes the bins
template <class M>
__device__ __forceinline__ void fin_zero_pads(unsigned lds0, int tid) {
  constexpr int kPadPairs = 64 * kHS / 2;   // b64 stores per pad
#pragma unroll
  for (int i = 0; i < (5 * kPadPairs + kThreads - 1) / kThreads; ++i) {
    const int q = tid + i * kThreads, p = q / kPadPairs, e = q - p * kPadPairs;
    if (q < 5 * kPadPairs) lds_st<f32x2>(lds0 + 4 * (M::kHOff + p * kHFrame * kHS + 2 * e), 0, f32x2{0.f, 0.f});
  }
}
typedef f32x4 __attribute__((aligned(4))) f32x4_u;   // a [frame][129] row is only 4-byte aligned
template <int CTRL>
__device__ __forceinline__ float row_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

template <class M>
__device__ __forceinline__ void final_phase(const Params& P, unsigned lds0, unsigned w128, int wave, int lane, int utt,
                                            int t0, const FinA& A, const XStage& xnext, float* x0) {
  lane = opaque(lane);   // once per tile: nothing derived from the lane id here is worth a register across the tile loop
  const int n = lane & 15, kq = lane >> 4;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  float* yt = P.y + ((size_t)utt * P.T + t0) * kF;   // the tile's first output row
  const int nfr = P.T - t0 < kTF ? P.T - t0 : kTF;   // frames of the tile inside the utterance
  if (wave == 2) {   // ---- bin 128 of the four frames (taps f' = 64..128, 8 channels each) on the VALU, FIRST: wave 6
                     //      keeps the SIMD's matrix pipe busy meanwhile
    // 16 lanes per frame; lane `sub` takes taps t = sub + 16 m, m < 5.  Taps 65..79 carry zero weights (pack_v3) and meet
    // the zero pad behind the frame, so there is no predicate and all loads of a batch are in flight together (as a
    // predicated loop hipcc serialised it into ten LDS round trips in front of this wave's MFMA run, which every other
    // wave then waited for at the barrier).
    const int fi = kq, sub = n;
    const unsigned hb = lds0 + 4 * (M::kHOff + (kHFrame * fi + 64 + 64 + sub) * kHS);
    const unsigned wb = w128 + sub * 32;   // bin-128 weights: LDS-DMA'd during block 4's layer 3 (X6) / behind the next tile's first packet (F32)
    f32x4 hv[5][2], wv[5][2];
#pragma unroll
    for (int m = 0; m < 5; ++m)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const f32x2 h0 = lds_ld<f32x2>(hb, (16 * m * kHS + 4 * c) * 4), h1 = lds_ld<f32x2>(hb, (16 * m * kHS + 4 * c + 2) * 4);
        hv[m][c] = f32x4{h0.x, h0.y, h1.x, h1.y};
        wv[m][c] = lds_ld<f32x4>(wb, (16 * m * 8 + 4 * c) * 4);
      }
    f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 5; ++m)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        s4.x = __builtin_fmaf(hv[m][c].x, wv[m][c].x, s4.x);
        s4.y = __builtin_fmaf(hv[m][c].y, wv[m][c].y, s4.y);
        s4.z = __builtin_fmaf(hv[m][c].z, wv[m][c].z, s4.z);
        s4.w = __builtin_fmaf(hv[m][c].w, wv[m][c].w, s4.w);
      }
    float sum = (s4.x + s4.y) + (s4.z + s4.w);
    // over the frame's 16 lanes (one DPP row), the same order in every lane: quads, then half rows, then the row
    sum += row_dpp<0xB1>(sum);    // quad_perm [1,0,3,2]
    sum += row_dpp<0x4E>(sum);    // quad_perm [2,3,0,1]
    sum += row_dpp<0x141>(sum);   // row_half_mirror
    sum += row_dpp<0x140>(sum);   // row_mirror
    if (sub == 0 && fi < nfr) yt[fi * kF + 128] = sum + P.fin_bias;
  }
  {  // ---- this wave's run of K-steps, both column tiles: column n of tile ct = (frame n >> 2, block 4*ct + (n & 3))
    const unsigned rd0 = lds0 + 4 * (M::kHOff + ((kHFrame * (n >> 2) + 16 * (n & 3)) * kHS + 2 * kq)) + wave * (kFinRun * kHS * 4);
    unsigned rd1 = rd0 + 64 * kHS * 4;
    asm volatile("" : "+v"(rd1));   // a base of its own (see make_lane)
    f32x4 acc[2][2] = {{zero4, zero4}, {zero4, zero4}};
    constexpr int D = 2, RING = D + 1;
    f32x2 b[RING][2];
    run_job<kFinRun, D>(
        [&](auto ic) {
          constexpr int i = decltype(ic)::value, r = i % RING;
          b[r][0] = lds_ld<f32x2>(rd0, i * kHS * 4);
          b[r][1] = lds_ld<f32x2>(rd1, i * kHS * 4);
        },
        [&](auto ic) {
          constexpr int i = decltype(ic)::value, r = i % RING;
          acc[0][0] = mfma(A.a[i].x, b[r][0].x, acc[0][0]);
          acc[1][0] = mfma(A.a[i].x, b[r][1].x, acc[1][0]);
          acc[0][1] = mfma(A.a[i].y, b[r][0].y, acc[0][1]);
          acc[1][1] = mfma(A.a[i].y, b[r][1].y, acc[1][1]);
        },
        [] {});
    const unsigned scr = lds0 + M::finscr0(wave) + lane * 16;
    lds_st<f32x4>(scr, 0, acc[0][0] + acc[0][1]);
    lds_st<f32x4>(scr, M::kFinScrCt, acc[1][0] + acc[1][1]);
  }
  // The next tile's input rows (in registers since block 4's layer 3) go to X0 here: X0 aliases the start of B30, dead
  // since the barrier that ended layer 3 and clear of the partial sums above.  The barrier below then also starts the
  // next tile: waves 2..7 go straight into its first layer while waves 0, 1 finish this tile's masks.
  xstage_store(xnext, x0, wave * 64 + lane);
  __syncthreads();
  if (wave < 2) {   // ---- finish column tile `wave`: partial sums of waves 0..7, in that order, + bias
    const unsigned scr = lds0 + wave * M::kFinScrCt + lane * 16;
    f32x4 v = lds_ld<f32x4>(scr, M::finscr0(0));
#pragma unroll
    for (int w = 1; w < kWaves; ++w) v += lds_ld<f32x4>(scr, M::finscr0(w));
    v += f32x4{P.fin_bias, P.fin_bias, P.fin_bias, P.fin_bias};
    const int fi = n >> 2, f0 = 16 * (4 * wave + (n & 3)) + 4 * kq;   // rows 4kq..4kq+3 = bins f0..f0+3 of frame fi
    if (fi < nfr) {
      *reinterpret_cast<f32x4_u*>(yt + fi * kF + f0) = v;
      store_wait_state();   // see lds_dma.h
    }
  } else if (wave == 3 && !M::kFused) {
    // The H image (and, X6, the bin-128 weights behind it) lay over B18, whose gap pixels every layer relies on being zero
    // and no layer ever writes: put the zeros back (every wave finished its H reads before the barrier above).
    // (Fused form: H has a place of its own.)
    if constexpr (M::kX6) {
      // per plane: the two leading pad rows and the four gap rows behind each frame = 18 rows of 32 bytes; 3 planes x 18 rows
      // x 2 halves = 108 sixteen-byte stores (the remainder rows lie behind the planes, untouched by H)
      constexpr int kRows = kB18Pad + 4 * kTF;
#pragma unroll
      for (int i = 0; i < (3 * kRows * 2 + 63) / 64; ++i) {
        const int q = lane + 64 * i, pl = q / (kRows * 2), e = q - pl * (kRows * 2), rr = e >> 1, half = e & 1;
        const int row = rr < kB18Pad ? rr : kB18Pad + kF + kS * ((rr - kB18Pad) >> 2) + ((rr - kB18Pad) & 3);
        if (q < 3 * kRows * 2) lds_st<f32x4>(lds0 + 4 * M::kB18Off + pl * M::kPlaneBytes + row * 32 + half * 16, 0, zero4);
      }
    } else {
      constexpr int kGapPairs = 4 * 18 / 2;   // b64 stores per gap
#pragma unroll
      for (int i = 0; i < (4 * kGapPairs + 63) / 64; ++i) {
        const int q = lane + 64 * i, g = q / kGapPairs, e = q - g * kGapPairs;
        if (q < 4 * kGapPairs)
          lds_st<f32x2>(lds0 + 4 * (M::kB18Off + (kB18Pad + kF + kS * g) * 18 + 2 * e), 0, f32x2{0.f, 0.f});
      }
    }
  }
}

#include "kernels_fused_v3_allx6.h"

// (no packed fp32 VALU: hipcc pairs the fused form's shift-adds of a lane's two couts into v_pk_add_f32 fed by two v_mov_b32_dpp --
// three instructions where two v_add_f32_dpp do, and the packed add is the slower instruction beside MFMAs)
template <class M>
__global__ __launch_bounds__(RCED_V3_LB) __attribute__((target("no-packed-fp32-ops"))) void fused_v3_kernel(Params P) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* const wbase = lds + M::kWOff;
#define WREG(i) (wbase + (i) * kWRegion)

  // zero all of LDS once: gap pixels and margins are never written afterwards
  for (int e = tid; e < M::kLdsFloats; e += kThreads) lds[e] = 0.f;
  __syncthreads();
  // F32 form: packet of the very first layer into region 0; X6 form: its A fragments into registers.  Input rows of the
  // first tile into registers.
  A1Regs A1;
  A2Regs A2;
  // piece k = 0..16 of a block's register-resident weights (X6 form; g = that block's images): layer 1's main pass (k < 7),
  // layer 2's M-tile of this wave (7..16).  (Layer 1's remainder pass, 8 more pieces for waves 4..7: inside layer 1.)
  const wrsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.wpack), 0, (M::kAllX6 ? kATotal : M::kFused ? kTTotal : M::kX6 ? kGTotal : kWTotal) * 4, 0x00020000);
  constexpr bool kL1X = M::kFused && RCED_T_L1X6;   // blocks 1..4's layer 1 on the bf16 pipe: their layer-1 images are kG1X floats, block 0's kG1
  auto wload = [&](auto kc, int g, unsigned voff, int g1size = kG1) {   // g: float offset of the block's images in the stream
    constexpr int k = decltype(kc)::value;
    if constexpr (k < 7) a1_load_one<k>(A1, wrs, g, voff);
    else if constexpr (k < 17) a2_load_one<k - 7>(A2, wrs, g + g1size, RCED_L2_BOTH ? 0 : wave >> 2, voff);
  };
  if constexpr (M::kAllX6) {   // no weight lives in registers: decode_final's tap table (once) and block 0's layer-1 image by LDS-DMA
    packet_dma<kFinTFloats>(P.fin, lds + M::kFinTOff, wave, lane);
    a1x_dma<M>(P.wpack, lds, wave, lane);
  } else if constexpr (M::kX6) {
    const unsigned voff = (unsigned)lane * 16u;
    static_for<0, 7>([&](auto kc) { wload(kc, 0, voff); });
    if constexpr (M::kFused) packet_dma<kFin128>(P.fin + kFinA, lds + M::kFin128Off, wave, lane);   // decode_final's bin-128 weights: once, a place of their own
  } else {
    packet_dma<kW1>(P.wpack, WREG(0), wave, lane);
  }
  int wcur = 0;
  unsigned epoch = 0;   // layer-3 instances so far (tags the split-tile hand-offs)
  // This workgroup's tiles: a CONTIGUOUS range (balanced: the first total % grid workgroups take one more).  Consecutive
  // tiles of an utterance share 7 of their 11 input rows; walked in order by one workgroup those re-reads hit the L2 (the
  // round-2 stride of gridDim.x sent 2.75 x the input to the memory fabric: FETCH_SIZE 187 MB for a 67.6 MB input once the
  // counter is calibrated -- tools/micro/fetch_cal.hip: it reports half the bytes for every access width on gfx950).
  const int tiles_base = P.total_tiles / (int)gridDim.x, tiles_rem = P.total_tiles % (int)gridDim.x;
  const int tile_begin = (int)blockIdx.x * tiles_base + ((int)blockIdx.x < tiles_rem ? (int)blockIdx.x : tiles_rem);
  const int tile_end = tile_begin + tiles_base + ((int)blockIdx.x < tiles_rem ? 1 : 0);
  XStage xst = xstage_load(P, tile_begin < tile_end ? tile_begin : P.total_tiles, tid);
#if RCED_STAMPS
  unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tfin = 0, tdet[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  layer_end_sync();

  // extra tiles of this wave (see the assignment comment above)
  // Fused form: layer 1's roles are swapped between the two waves of a SIMD (role = wave ^ 4): the issue arbiter prefers the OLDER
  // wave, so the heavier roles 4..7 (remainder tiles) go to waves 0..3 -- the lighter wave then fills the gaps and both end together
  // (6.52 -> 6.45 ms; s_setprio on top, or instead: nothing / worse.  In layers 2 + 3 it is the other way round: the five-tile halves on
  // the older waves WITHOUT s_setprio 6.55 ms, with it 6.46, the same as on the younger waves with it)
  const int w1 = M::kFused && RCED_T_L1SWAP ? wave ^ 4 : wave;
  const int xr0 = w1 == 7 ? 3 : w1 - 4;                         // layer 1 remainder tile of roles 4..7 (role 7 also 4)
#ifdef RCED_PRIO
  if ((wave >= 4) == (RCED_PRIO > 0)) __builtin_amdgcn_s_setprio(1);   // experiment: static priority for one half of the waves
#endif
  const Lane L = make_lane<M>(lds, wave, lane, xr0 < 0 ? 0 : xr0, w1);
  const unsigned lds0 = lds_addr(lds);
  xstage_store(xst, lds + M::kX0Off, tid);   // the first tile's input rows; every later tile's are stored by final_phase
  __syncthreads();
  if constexpr (M::kAllX6) {                 // ... and laid out as the first layer's planes
    convert_x0<M>(lds0, wave, lane);
    __syncthreads();
  }
  constexpr int kBlockFloats = M::kFused ? kTBlock : M::kX6 ? kGBlock : kWBlock;

  for (int tile = tile_begin; tile < tile_end; ++tile) {
    const int utt = tile / P.tiles_per_utt;
    const int t0 = (tile - utt * P.tiles_per_utt) * kTF;   // first frame of the tile
    f32x4 skip_ce1[3], skip_ce2[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) skip_ce1[t] = skip_ce2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x2 sk1[5], sk2[5];   // fused form: CE1 / CE2 outputs of this wave's (up to) five tiles
#pragma unroll
    for (int t = 0; t < 5; ++t) sk1[t] = sk2[t] = f32x2{0.f, 0.f};
    const float* wsrc = P.wpack;
    int gofs = 0;   // float offset of the block's images in the weight stream (= wsrc - P.wpack)

    // Layer 3 of block BLK < 4 inside the loop, block 4's behind it: decode_final's operands (36 registers of A fragments)
    // are fetched in front of THAT instance only; declared outside a five-iteration loop they were live through all of it.
    // The layer-1 section of block blk (with the other forms' layer 2 behind it).  WHICH: 1 = block 0's first layer, 0 = one of blocks
    // 1..4, 2 = decided at run time.  With layer 1 on the bf16 pipe (kL1X) block 0's instance runs in FRONT of the block loop: its fp32
    // fragments (A1, loaded at the end of the previous tile) would otherwise be live through all five iterations.
    auto l1_section = [&](auto whichc, int blk) __attribute__((always_inline)) {
      constexpr int which = decltype(whichc)::value;
      {  // ---- layer 1: (8x9, 1->18) for block 0, (1x9, 8->18) otherwise
        STAMP_BEGIN();
        const unsigned wb = lds_addr(WREG(wcur));
        // What is fetched for later once the layer's first operand reads are in flight.  F32 form: layer 2's packet.  X6 form:
        // layer 3's packet (its one LDS region is dead until then).
        auto dma = [&] {
          if constexpr (M::kFused) packet_dma<kG2 + kW3T>(wsrc + (M::kAllX6 || (kL1X && blk > 0) ? kG1X : kG1), WREG(0), wave, lane);   // layer 2's and layer 3's images, adjacent in the stream
          else if constexpr (M::kX6) packet_dma<kW3>(wsrc + kG1 + kG2, WREG(0), wave, lane);
          else packet_dma<kW2>(wsrc + kW1, WREG(wcur ^ 1), wave, lane);
        };
        // One load per slot of the pair job(s), k = 0..17: layer 2's fragments (k < 10) and this layer's remainder-pass fragments
        // (k = 10..17, waves 4..7; used by the wave's last job).  (Measured: the same loads inside layer 3's stream, which has
        // the room on the vector-memory path, cost 0.2 - 0.35 ms MORE -- its slots are the tightest in the kernel.)
        A1Rem A1r;
        const unsigned voff1 = (unsigned)opaque(lane) * 16u;
        auto sp1 = [&](auto kc) {
          constexpr int k = decltype(kc)::value;
          if constexpr (M::kX6 && !(RCED_X6_EXP & 2)) {
            // (every wave fetches the remainder pass's fragments although only waves 4..7 use them: behind a wave-uniform
            // branch each load cost two register copies of the "old" value, the MFMA -> VALU wait states in front of them
            // and the branch -- more than the 8 KiB of L2 traffic per wave it saves)
            if constexpr (k >= 10) a1_load_rem_one<k - 10>(A1r, wrs, gofs, voff1);
            else if constexpr (!M::kFused || ((RCED_T_A2REG == 1 || RCED_T_A2REG == 2) && k < 9)) wload(IC<k + 7>{}, gofs, voff1);   // (fused form: M-tile 0 only, without its shifts: the rest stays in LDS)
          }
        };
        // The second M-tile's fragments (RCED_L2_BOTH) are fetched behind the pair jobs, whose registers (the main pass's A
        // fragments, two pairs of accumulators) they take over; the waves' remaining jobs and the wait of the early finishers
        // at the layer's barrier cover them.
        auto late = [&] {
          if constexpr (M::kX6 && (!M::kFused || RCED_T_A2REG == 2) && RCED_L2_BOTH && !(RCED_X6_EXP & 2))
            static_for<0, 10>([&](auto kc) { a2_load_one<decltype(kc)::value, kL2MT - 1>(A2, wrs, gofs + kG1, 1, voff1); });
        };
        auto sp1x = [&](auto kc) {   // blocks 1..4 with layer 1 on the bf16 pipe: k = 12..20: layer 2's M-tile 0 (three loads per slot)
          constexpr int k = decltype(kc)::value;
          if constexpr (kL1X && k >= 12 && k < 21 && (RCED_T_A2REG == 1 || RCED_T_A2REG == 2)) wload(IC<k - 12 + 7>{}, gofs, voff1, kG1X);
        };
        if constexpr (kL1X) {
          if constexpr (which == 1) layer1<M, true>(L, wb, A1, A1r, w1, dma, sp1, late DET_PASS);
          else layer1_x6l<M>(L, lds0, w1, dma, sp1x DET_PASS);
        } else {
          if (blk == 0) layer1<M, true>(L, wb, A1, A1r, w1, dma, sp1, late DET_PASS);
          else layer1<M, false>(L, wb, A1, A1r, w1, dma, sp1, late DET_PASS);
        }

        if constexpr (!M::kX6) wcur ^= 1;
#if RCED_STAMPS
        const unsigned long long st_b_ = stamp();
        tsum[blk == 0 ? 6 : 0] += st_b_ - st_a_;
        layer_end_sync();
        tsum[blk == 0 ? 7 : 3] += stamp() - st_b_;
#else
        layer_end_sync();
#endif
        // ---- layer 2: (1x5, 18->30)
        if constexpr (!M::kFused) {
          STAMP_BEGIN();
          const unsigned tag2 = 0xC0000000u | (epoch + 1u);   // distinct from layer 3's tags (0x8.......)
          if constexpr (M::kX6) {
#if RCED_V3_LEGACY_FORMS
            if constexpr (RCED_L2_BOTH) layer2_x6_both<M>(L, lds0, A2, wave, tag2, P.err, [] {} DET_PASS);
            else layer2_x6<M>(L, lds0, A2, wave, tag2, P.err, [] {} DET_PASS);
#else
            static_assert(!M::kX6 || M::kFused, "form 1 is a legacy form (RCED_V3_LEGACY_FORMS)");
#endif
          } else {
            auto dma3 = [&] { packet_dma<kW3>(wsrc + kW1 + kW2, WREG(wcur ^ 1), wave, lane); };
            layer2_f32<M>(L, lds0, lds_addr(WREG(wcur)), wave, tag2, P.err, dma3 DET_PASS);
            wcur ^= 1;
          }
          STAMP_MATH(1);
          layer_end_sync();
          STAMP_WAIT(1);
        }
            }
    };
    if constexpr (M::kAllX6) l1_section(IC<0>{}, 0);   // (the first layer is layer1_x6l too)
    else if constexpr (kL1X) l1_section(IC<1>{}, 0);
#pragma unroll 1
    for (int blk = 0;; ++blk) {
      if constexpr (kL1X) {
        if (blk > 0) l1_section(IC<0>{}, blk);
      } else {
        l1_section(IC<2>{}, blk);
      }
      if (blk == 4) break;
      if constexpr (M::kFused) {   // ---- layers 2 + 3 as one stream (kernels_fused_v3_l23.h); the next layer 1's main-pass fragments ride in its last slots
        STAMP_BEGIN();
        ++epoch;
        const unsigned voff = (unsigned)opaque(lane) * 16u;
        auto sp = [&](auto jc) {
          constexpr int j = decltype(jc)::value;
          if constexpr (!kL1X) {   // (with layer 1 on the bf16 pipe its images travel by LDS-DMA: dma1 below)
            if constexpr (2 * j < 7) wload(IC<2 * j>{}, gofs + kBlockFloats, voff);
            if constexpr (2 * j + 1 < 7) wload(IC<2 * j + 1>{}, gofs + kBlockFloats, voff);
          }
        };
        auto dma1 = [&] {   // layer 1 on the bf16 pipe: the next block's layer-1 image into the (now dead) input-row area and the H image's bins
          if constexpr (kL1X) a1x_dma<M>(wsrc + (blk == 0 && !M::kAllX6 ? kTBlock : kTBlockX), lds, wave, lane);
        };
        layer23<M, false>(P, L, lds0, lds_addr(WREG(0)), A2, blk, wave, 0x80000000u | epoch, sk1, sk2, sp, dma1 DET_PASS);
        STAMP_MATH(2);
        layer_end_sync();
        STAMP_WAIT(2);
      } else {  // ---- layer 3: (1x9, 30->8) on pixel pairs; block skips; hand-off
        STAMP_BEGIN();
        const unsigned wb = lds_addr(WREG(M::kX6 ? 0 : wcur));
        ++epoch;
        const unsigned tag = 0x80000000u | epoch;   // sign bit set: never the bits of a ReLU output
        // next: layer 1 of the next block.  F32 form: its packet; X6 form: its main pass's A fragments into registers
        const float* wnext = wsrc + kBlockFloats;
        auto dma = [&] {
          if constexpr (!M::kX6) packet_dma<kW1>(wnext, WREG(wcur ^ 1), wave, lane);
        };
        const unsigned voff = (unsigned)opaque(lane) * 16u;
        auto sp = [&](auto kc) {
          if constexpr (M::kX6 && !(RCED_X6_EXP & 4) && decltype(kc)::value < 7) wload(kc, gofs + kBlockFloats, voff);   // the next layer 1's main pass (inside layer 2's stream instead: no difference;
                                                                                                   // layer 2's fragments here too, instead of inside layer 1: 8.31 against 8.13 ms -- A/B on one box)
        };
        layer3<M, false>(P, L, lds0, wb, blk, wave, tag, skip_ce1, skip_ce2, dma, sp DET_PASS);
        if constexpr (!M::kX6) wcur ^= 1;
        STAMP_MATH(2);
        layer_end_sync();
        STAMP_WAIT(2);
      }
      wsrc += M::kAllX6 || (kL1X && blk > 0) ? kTBlockX : kBlockFloats;
      gofs += M::kAllX6 || (kL1X && blk > 0) ? kTBlockX : kBlockFloats;
    }
    FinA finA;
    if constexpr (M::kAllX6) {   // ---- block 4's layers 2 + 3 (its output goes to decode_final's image H'), then decode_final and the next tile's planes
      STAMP_BEGIN();
      ++epoch;
      xst = xstage_load(P, tile + 1 < tile_end ? tile + 1 : P.total_tiles, tid);
      // the zero row out-of-range taps read: the image lies over the 8-channel planes (dead since block 4's layer 1), whose old contents are there
      if (wave == 0 && lane < 3) lds_st<u32x4>(lds0 + 4 * M::kB8Off + kHZeroRow * 16 + (unsigned)lane * kHPlaneBytes, 0, u32x4{0u, 0u, 0u, 0u});
      auto dma1 = [&] { a1x_dma<M>(P.wpack, lds, wave, lane); };   // the next tile's first layer: block 0's image (the stream wraps)
      layer23<M, true>(P, L, lds0, lds_addr(WREG(0)), A2, 4, wave, 0x80000000u | epoch, sk1, sk2, [](auto) {}, dma1 DET_PASS);
      STAMP_MATH(2);
      layer_end_sync();
      STAMP_WAIT(2);
    } else if constexpr (M::kFused) {   // ---- block 4's layers 2 + 3; decode_final's A fragments ride in its last slots, the next tile's input rows in front
      STAMP_BEGIN();
      ++epoch;
      xst = xstage_load(P, tile + 1 < tile_end ? tile + 1 : P.total_tiles, tid);
      const int lane_o = opaque(lane);
      auto sp = [&](auto jc) {
        constexpr int j = decltype(jc)::value, s0 = (kFinRun * j) / 5, s1 = (kFinRun * (j + 1)) / 5;
        const f32x2* src = reinterpret_cast<const f32x2*>(P.fin) + (size_t)(kFinRun * wave) * 64 + lane_o;
#pragma unroll
        for (int q = s0; q < s1; ++q) finA.a[q] = src[q * 64];
      };
      layer23<M, true>(P, L, lds0, lds_addr(WREG(0)), A2, 4, wave, 0x80000000u | epoch, sk1, sk2, sp, [] {} DET_PASS);
      STAMP_MATH(2);
      layer_end_sync();
      STAMP_WAIT(2);
      // the next tile's first layer: its A fragments (the stream wraps); in flight during decode_final
      const unsigned voff = (unsigned)opaque(lane) * 16u;
      static_for<0, 7>([&](auto kc) { wload(kc, 0, voff); });
    } else {  // ---- block 4's layer 3, and in front of it, once per tile: what decode_final and the next tile need
      STAMP_BEGIN();
      const unsigned wb = lds_addr(WREG(M::kX6 ? 0 : wcur));
      ++epoch;
      const unsigned tag = 0x80000000u | epoch;
      xst = xstage_load(P, tile + 1 < tile_end ? tile + 1 : P.total_tiles, tid);   // next tile's input rows (none: no loads)
      // decode_final's bin-128 weights ride along: X6 form: into B18 (dead from here on) behind the H image
      if constexpr (M::kX6) packet_dma<kFin128>(P.fin + kFinA, lds + M::kFin128Off, wave, lane);
      else packet_dma<kFin128>(P.fin + kFinA, WREG(wcur ^ 1) + kW1, wave, lane);
      fin_prefetch(P, wave, lane, finA);   // decode_final's A fragments: in flight during this layer
      fin_zero_pads<M>(lds0, tid);         // B18 is dead from here on (layer 3's own scratch sits below the H image)
      auto dma = [&] {   // layer 1 of block 0 of the next tile (the stream wraps)
        if constexpr (!M::kX6) packet_dma<kW1>(P.wpack, WREG(wcur ^ 1), wave, lane);
      };
      const unsigned voff = (unsigned)opaque(lane) * 16u;
      auto sp = [&](auto kc) {
        if constexpr (M::kX6 && !(RCED_X6_EXP & 4) && decltype(kc)::value < 7) wload(kc, 0, voff);
      };
      layer3<M, true>(P, L, lds0, wb, 4, wave, tag, skip_ce1, skip_ce2, dma, sp DET_PASS);
      if constexpr (!M::kX6) wcur ^= 1;
      STAMP_MATH(2);
      layer_end_sync();
      STAMP_WAIT(2);
    }
#if RCED_STAMPS
    const unsigned long long st_f_ = stamp();
#endif
    if constexpr (M::kAllX6) {
      final_phase_x6<M>(P, lds0, wave, lane, utt, t0, xst, lds + M::kX0Off DET_PASS);
    } else {
      const unsigned w128 = M::kX6 ? lds0 + 4 * M::kFin128Off : lds_addr(WREG(wcur) + kW1);   // (fused form: loaded once, at the kernel's start)
      final_phase<M>(P, lds0, w128, wave, lane, utt, t0, finA, xst, lds + M::kX0Off);   // no barrier at its end: layer 1's covers it
    }
#if RCED_STAMPS
    tfin += stamp() - st_f_;
#endif
  }
#undef WREG
#if RCED_STAMPS
  if (P.stamps && blockIdx.x == 0 && lane == 0)
    for (int i = 0; i < 8; ++i) P.stamps[wave * 8 + i] = tsum[i];
  if (P.stamps && blockIdx.x == 0 && lane == 0) P.stamps[64 + wave * 3] = tfin;
  if (P.stamps && blockIdx.x == 0 && lane == 0)
    for (int i = 0; i < 16; ++i) P.stamps[88 + wave * 16 + i] = tdet[i];
#endif
}


}  // namespace v3
}  // namespace rced
