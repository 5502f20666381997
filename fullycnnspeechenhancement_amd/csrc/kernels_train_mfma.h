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

#include <cstdio>
#include <type_traits>

#include "kernels_fused_chain.h"

namespace rced {
namespace tmm {

using chain::f32x2;
using chain::f32x4;
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 16 bytes at a dword-aligned address
using chain::mfma;
using chain::pin;

// Where a wgrad kernel's per-wave partial sums go.  pstride == 0: atomicAdd into dW / dbias (fp32 atomics: the order of
// the adds, hence the last bits of the result, changes from run to run).  pstride > 0 (the default, option
// RCED_TRAIN_DET=0 turns it off): every (workgroup, wave[, pixel parity]) writes its sums to a slice of its own,
// base[slice * pstride + idx], and wg_reduce adds the slices in a fixed order: the same bits every run.
__device__ __forceinline__ void wg_put(float* base, unsigned pstride, int slice, int idx, float v) {
  if (pstride) base[(size_t)slice * pstride + idx] = v;
  else atomicAdd(base + idx, v);
}
// out[e] = sum over slices, in a fixed order: a workgroup owns 16 elements x 64 slice-lanes (thread (e, p) adds slices
// p, p+64, ...), then the 64 partial sums in order.  n = nW + nB elements per slice (dbias behind dW).
// (Round 2 had 64 elements x 16 slice-lanes: 43 workgroups for the largest layer, 35 us per call on a 256-CU part, sixteen
// calls per step; with four times the workgroups and a quarter of the serial chain per thread ...)
constexpr int kWgrElems = 16;
static __global__ __launch_bounds__(1024) void wg_reduce(const float* __restrict__ part, int nslices, unsigned pstride, int nW,
                                                        int nB, float* __restrict__ dW, float* __restrict__ dbias) {
  __shared__ float red[64][kWgrElems];
  const int le = threadIdx.x & (kWgrElems - 1), p = threadIdx.x >> 4, e = blockIdx.x * kWgrElems + le, n = nW + nB;
  float s = 0.f;
  if (e < n)
    for (int sl = p; sl < nslices; sl += 64) s += part[(size_t)sl * pstride + e];
  red[p][le] = s;
  __syncthreads();
  if (p == 0 && e < n) {
    float t = red[0][le];
#pragma unroll
    for (int q = 1; q < 64; ++q) t += red[q][le];
    if (e < nW) dW[e] = t;
    else if (dbias) dbias[e - nW] = t;
  }
}

constexpr int kF = 129;
constexpr int kTF = 2;                 // frames per tile (frame-of-element splits are written as one compare: keep it 2)
constexpr int kWaves = 4, kThreads = 256;
#ifndef RCED_TM_STAMPS
#define RCED_TM_STAMPS 0   // diagnostic build: s_memtime phase sums of workgroup 0 of the 30->18 SUMS dgrad, printed at its end
#endif
#if RCED_TM_STAMPS
__device__ __forceinline__ unsigned long long tm_stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
__device__ unsigned long long g_tm2[4][4];  // [wave][coords, z reads + masked sums, stores, reduce] inside the epilogue
__device__ unsigned long long g_tm[4][4];   // [wave][gemm, barrier 2, epilogue, vmcnt wait] of conv_tile
#define TM_ST(i) do { if (stamps_on) { const unsigned long long n_ = tm_stamp(); ts[i] += n_ - tlast; tlast = n_; } } while (0)
#else
#define TM_ST(i)
#endif
#ifndef RCED_TM_EXP
#define RCED_TM_EXP 0   // timing experiments only (results wrong): bit0 = conv kernels fetch no tile after their first,
                        // bit1 = conv kernels store nothing, bit2 = conv kernels skip the MFMA pass, bit3 = they commit only their first tile,
                        // bit4 = no masked sums in the SUMS epilogue
#endif
#if RCED_TM_EXP != 0 && !defined(RCED_TIMING_ONLY)
#error "RCED_TM_EXP builds compute wrong results: timing experiments only (tools/mkexp.sh ... -DRCED_TIMING_ONLY -DRCED_TM_EXP=...)"
#endif
#ifndef RCED_TM_OPQ_MIN
#define RCED_TM_OPQ_MIN 1000   // prefetch size (VGPRs) from which the conv kernels' epilogue re-derives its lane coordinates
                               // (64 paid while the largest dgrad spilled; with the loop-invariant staging channels nothing
                               // spills and the recomputation only costs: 56.0 vs 56.9 ms per CR-CED step)
#endif
#ifndef RCED_TM_FIXCH
#define RCED_TM_FIXCH 1   // staging stride chosen so that a thread's pieces share their channels (Stage<C>::kStride)
#endif
#ifndef RCED_TM_OCC
#define RCED_TM_OCC 2   // workgroups per CU the register allocator must leave room for (conv / wgrad kernels): several
                        // of them sat at 260-300 VGPRs+AGPRs = ONE workgroup per CU; 2 costs a few spilled dwords in the
                        // largest, 3 spills hundreds of bytes and is slower (64.8 / 60.9 / 77.8 ms per CR-CED step)
#endif

// Remainder pass (RCED_TM_REM): a conv with 18 output channels fills two 16-row M-tiles 56 %.  As in the inference kernels
// (kernels_fused_v3.h's ->18 layers, kernels_fused_chain.h's 17..24-channel layers) channels 0..15 run as ONE M-tile (the
// main pass) and channels 16, 17 in a remainder pass whose 16 rows are (pixel phase p < 8, channel 16 + c): a column is a
// group of 8 adjacent pixels, K = (TAPS + 7) * cin (row (p, c) holds the kernel shifted by p taps), a remainder tile = 128
// pixels.  Per two-frame tile: 17 x K/4 + 3 x (K + 7 cin)/4 MFMAs instead of 34 x K/4 (-29 % for the 30 -> 18 dgrad, -36 %
// for the 8 -> 18 forward).  tm_rem(cout) = channels of the remainder pass (0: none); only the 18-channel form is built.
#ifndef RCED_TM_REM
#define RCED_TM_REM 1
#endif
__host__ __device__ constexpr int tm_rem(int cout) { return (RCED_TM_REM && cout == 18) ? 2 : 0; }
// Pixel phases per remainder column.  16 / R = 8 rows pairs fill the M-tile, but a column stride of 8 pixels puts the 16 columns of an
// operand read on the same banks whenever 8 * cin floats is a multiple of 256 bytes (cin = 8: 16-way conflicts -- 71 % of the LDS
// cycles of the 8 -> 18 forward convolution were conflict cycles, profiles/r04_rocprof_train_step.txt; cin = 30: 4-way).  SEVEN phases
// (the rows of phase 7 are zero and dropped) where the 8-pixel stride is a multiple of 256 bytes (cinp % 8 == 0: the 8 -> 18 forward
// convolutions, 0.66 -> 0.62 / 0.62 -> 0.57 ms): a stride of 224 bytes, 2-way; K shrinks by one tap and three tiles still cover the 266
// pixels of a two-frame tile (round 5; the inference kernel's layer 1 does the same, kernels_fused_v3.h).  NOT for cin = 30 (the
// 30 -> 18 dgrad inside bwd_fused_mfma<18,5,30>: 4-way at 8 phases, conflict-free at 7 -- and 12 % SLOWER, 2.24 -> 2.51 ms: that
// kernel's waves are balanced to the MFMA counts of the 8-phase form).
#ifndef RCED_TM_REM_P
#define RCED_TM_REM_P 7
#endif
__host__ __device__ constexpr int tm_rem_p(int r, int cinp) { return r == 2 && cinp % 8 == 0 ? RCED_TM_REM_P : (r ? 16 / r : 0); }

template <int CIN, int TAPS, int COUT>
struct Geo {
  static constexpr int kCinP = (CIN + 1) & ~1;
  static constexpr int kCoutP = (COUT + 1) & ~1;
  static constexpr int kG = (TAPS - 1) / 2;              // halo = gap between frames
  static constexpr int kS = kF + kG;
  static constexpr int kNPX = kTF * kS;
  static constexpr int kTiles = (kNPX + 15) / 16;
  static constexpr int kK = TAPS * kCinP;                // the layer's K
  static constexpr int kMT = (COUT + 15) / 16;
  // conv1xk_mfma with 8 output channels: an MFMA column is a PAIR of adjacent pixels and the 16 rows are
  // (pixel parity, cout), so no row is wasted: K grows by one tap ((TAPS+1)*CinP, the parity-1 rows are the kernel
  // shifted by one tap) and the number of column tiles halves.  kPH = parities per column.
  static constexpr int kPH = COUT == 8 ? 2 : 1;
  static constexpr int kCTiles = kPH == 2 ? (kNPX + 31) / 32 : kTiles;          // conv column tiles
  static constexpr int kRegular = kCTiles / kWaves, kExtra = kCTiles - kRegular * kWaves;
  static constexpr int kKP = (TAPS + kPH - 1) * kCinP;   // K of the conv packet
  static constexpr int kNB64 = kKP / 8, kNTail = (kKP % 8 + 3) / 4;
  // remainder pass: R channels past the first M-tile, P pixel phases per column, its K and its tiles per two-frame tile
  static constexpr int kR = tm_rem(COUT), kP = tm_rem_p(kR, kCinP);
  static constexpr int kKR = kR ? (TAPS + kP - 1) * kCinP : 0;
  static constexpr int kNRT = kR ? (kNPX + 16 * kP - 1) / (16 * kP) : 0;
  static constexpr int kMTm = kR ? 1 : kMT;              // M-tiles of the MAIN pass
  static constexpr int kDataMain = kNB64 * kMTm * 128 + kNTail * kMTm * 64;
  static constexpr int kDataRem = kR ? (kKR / 8) * 128 + ((kKR % 8 + 3) / 4) * 64 : 0;
  static constexpr int kData = kDataMain + kDataRem;
  static constexpr int kPacket = kData + 32;             // + shift[32]: channels 0..15 (0..31 without a remainder pass), then the 16 remainder rows'
  static_assert(!kR || kPH == 1, "remainder pass and pixel-pair columns do not combine");
  static constexpr int kInRows = kG + (kPH == 2 ? 32 * kCTiles + 1 : 16 * kTiles) + kG;
  static constexpr int kInFloats = ((kInRows * kCinP + 3) / 4) * 4;
  static constexpr int kLdsFloats = kInFloats + kPacket;
};

// Pack the layer's weights (TF layout [TAPS][CIN][COUT] in the variable blob) into the A-fragment packet.
// transpose = 0: forward  (rows = COUT_L outputs, k = tap*cinp + ci).
// transpose = 1: dgrad    (rows = CIN_L outputs; the packet is for a conv whose input has COUT_L channels:
//                          W_t[tap'][co_l][ci_l] = w[TAPS-1-tap'][ci_l][co_l]).
// cin / cout below are those of the conv being PACKED (dgrad: cin = COUT_L, cout = CIN_L).
// ph = 2 (cout = 8, Geo::kPH): rows are (parity p, co) = 8 p + co over K = (taps + 1) * cinp, row (p, co) holding the
// kernel of output co shifted by p taps; shift[8 p + co] = shift[co].
static __global__ void pack_packet(const float* __restrict__ w, const float* __restrict__ shift, int taps, int cin, int cout,
                            int transpose, int ph, float* __restrict__ packet) {
  const int R = ph == 1 ? tm_rem(cout) : 0, P = tm_rem_p(R, (cin + 1) & ~1);   // (as Geo::kP)
  const int cinp = (cin + 1) & ~1, K = (taps + ph - 1) * cinp, MT = R ? 1 : (cout + 15) / 16;
  const int NB = K / 8, NTL = (K % 8 + 3) / 4, dmain = NB * MT * 128 + NTL * MT * 64;
  const int KR = R ? (taps + P - 1) * cinp : 0, NBR = KR / 8, NTR = (KR % 8 + 3) / 4, drem = R ? NBR * 128 + NTR * 64 : 0;
  const int data = dmain + drem;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= data + 32) return;
  if (e >= data) {
    const int c = e - data;
    if (R) packet[e] = !shift ? 0.f : (c < 16 ? shift[c] : shift[16 + (c - 16) % R]);   // 16 main channels, 16 remainder rows
    else packet[e] = (shift && c < ph * cout) ? shift[ph == 2 ? (c & 7) : c] : 0.f;
    return;
  }
  int k, co, tap_shift = 0, klim = K;
  if (e < dmain) {
    if (e < NB * MT * 128) {
      const int s = e / (MT * 128), r = e - s * MT * 128, mt = r / 128, q = r - mt * 128, lane = q >> 1, ee = q & 1;
      k = 8 * s + 2 * (lane >> 4) + ee;
      co = 16 * mt + (lane & 15);
    } else {
      const int r = e - NB * MT * 128, j = r / (MT * 64), q = r - j * MT * 64, mt = q / 64, lane = q - mt * 64;
      k = 8 * NB + 4 * j + (lane >> 4);
      co = 16 * mt + (lane & 15);
    }
    if (R && co >= 16) co = cout;    // (never: MT = 1)
  } else {       // remainder pass: row i = (phase i / R, channel 16 + i % R) over K = (taps + P - 1) * cinp
    const int r = e - dmain;
    int lane;
    if (r < NBR * 128) {
      const int s = r / 128, q = r - s * 128;
      lane = q >> 1;
      k = 8 * s + 2 * (lane >> 4) + (q & 1);
    } else {
      const int q = r - NBR * 128, j = q / 64;
      lane = q - j * 64;
      k = 8 * NBR + 4 * j + (lane >> 4);
    }
    const int i = lane & 15;
    tap_shift = i / R;
    co = 16 + i % R;
    klim = tap_shift < P ? KR : 0;      // (rows of a dropped phase: zero)
  }
  float v = 0.f;
  int tap = k / cinp;
  const int ci = k - tap * cinp;
  tap -= tap_shift;
  if (ph == 2) {            // row = (parity, co): the parity-1 rows see the window one tap later
    tap -= co >> 3;
    co &= 7;
  }
  if (k < klim && co < cout && ci < cin && tap >= 0 && tap < taps)
    v = transpose ? w[((taps - 1 - tap) * cout + co) * cin + ci]    // w[tap_l][ci_l = co][co_l = ci], layer dims (cout, cin)
                  : w[(tap * cin + ci) * cout + co];
  packet[e] = v;
}


// ---- tile staging: kTF frames are contiguous in global ([frame][bin][C], C even => 16-byte aligned tile start and
// a whole number of float4).  A thread keeps PER float4 in flight (fetch), and writes them to LDS later (commit) as
// float2 pieces, which never straddle a frame or a pixel because C is even.  The fetch of the NEXT tile is issued
// before the MFMA work of the current one so HBM latency hides behind it.
// A thread's i-th float4 is piece tid + i * kStride, with kStride the largest thread count <= 256 for which
// 4 * kStride is a multiple of C: every piece of a thread then starts at the SAME channel, so whatever per-channel
// values the commit transforms need (BatchNorm folds) are loop-invariant for the thread -- read once per commit instead
// of once per piece (each read was an LDS round trip the compiler could not overlap: 4.4 k cycles of commit per tile).
// The few threads past kStride idle during staging.
template <int C, int NTHR = kThreads>
struct Stage {
  static constexpr int kFrame = kF * C, kElems = kTF * kFrame, kVec = kElems / 4;
  static constexpr int kMod = C % 4 == 0 ? C / 4 : C / 2;              // pieces per channel period
  static constexpr int kStride = RCED_TM_FIXCH ? NTHR - NTHR % kMod : NTHR;
  static constexpr int kPer = (kVec + kStride - 1) / kStride;
  static_assert(C % 2 == 0 && kElems % 4 == 0, "wide staging needs an even channel count");
  static_assert(!RCED_TM_FIXCH || (4 * kStride) % C == 0, "a thread's pieces all start at the same channel");
};
template <int C, int NTHR = kThreads>
__device__ __forceinline__ void tile_fetch(const float* __restrict__ base, int frame0, int frames, int tid,
                                           f32x4 (&pre)[Stage<C, NTHR>::kPer]) {
  using St = Stage<C, NTHR>;
  const float* src = base + (size_t)frame0 * St::kFrame;
  const int left = frames - frame0;
  if (left >= kTF) {   // whole tile (wave-uniform): straight-line 16-byte loads, no per-piece bounds logic
    const f32x4* s4 = reinterpret_cast<const f32x4*>(src) + tid;
    const bool active = St::kStride == NTHR || tid < St::kStride;
#if RCED_TM_NT
#define RCED_TM_LD(p) __builtin_nontemporal_load(p)
#else
#define RCED_TM_LD(p) (*(p))
#endif
#pragma unroll
    for (int i = 0; i < St::kPer; ++i) {
      if ((i + 1) * St::kStride <= St::kVec) pre[i] = active ? RCED_TM_LD(s4 + i * St::kStride) : f32x4{0.f, 0.f, 0.f, 0.f};
      else pre[i] = (active && tid + i * St::kStride < St::kVec) ? RCED_TM_LD(s4 + i * St::kStride) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    return;
  }
  const int nvalid = (left < kTF ? left : kTF) * St::kFrame;
#pragma unroll
  for (int i = 0; i < St::kPer; ++i) {
    const int q = tid + i * St::kStride;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (q < St::kVec && tid < St::kStride) {
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
// One 16-byte piece of a WHOLE tile (wave-uniform precondition: frames - frame0 >= kTF): what tile_fetch's full-tile branch
// does for piece I.  The kernels issue the next tile's pieces one at a time between the MFMAs of the current tile
// (gemm_pass's `each`, the wgrad loops): issued in one burst right behind the barrier they took 3-4 k cycles per tile in
// which nothing else ran (s_memtime stamps of bwd_fused_mfma: the memory pipeline's queues fill and vector-memory issue
// blocks), 14 % of the fused backward kernel.
template <int C, int NTHR, int I>
__device__ __forceinline__ void tile_fetch_piece(const float* __restrict__ base, int frame0, int tid,
                                                 f32x4 (&pre)[Stage<C, NTHR>::kPer]) {
  using St = Stage<C, NTHR>;
  static_assert(I < St::kPer, "piece index");
  // wave-uniform base (SGPR pair) + ONE unsigned 32-bit lane offset shared by every piece: the saddr form of
  // global_load_dwordx4 -- per-piece 64-bit lane addresses stayed live across the MFMA pass and spilled
  const char* sb = reinterpret_cast<const char*>(base + (size_t)frame0 * St::kFrame + (size_t)I * St::kStride * 4);
  const unsigned voff = (unsigned)tid * 16u;
  const bool active = (St::kStride == NTHR || tid < St::kStride) &&
                      ((I + 1) * St::kStride <= St::kVec || tid + I * St::kStride < St::kVec);
  if (active) pre[I] = *reinterpret_cast<const f32x4*>(sb + voff);   // idle lanes: the commit never looks at theirs
}
// The same piece by LDS-DMA into a raw copy of the tile in LDS (`stg`, float4 q of the tile at stg + 4 q): no VGPR holds
// the next tile while the current one is computed.  A wave's 64 lanes are 64 consecutive float4, i.e. one contiguous
// 1-KiB transfer (global_load_lds_dwordx4: lane i writes M0 + 16 i); wave-uniform source base + one lane offset.
// `tid` is the thread of the NTHR-thread mapping whose piece this is, not necessarily the issuing thread: the transfer
// touches no register, so any wave can issue any wave's share (bwd_fused_mfma: the wgrad half issues the whole tile).
template <int C, int NTHR, int I>
__device__ __forceinline__ void tile_dma_piece(const float* __restrict__ base, int frame0, int tid, float* stg) {
  using St = Stage<C, NTHR>;
  static_assert(I < St::kPer, "piece index");
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool active = (St::kStride == NTHR || tid < St::kStride) &&
                      ((I + 1) * St::kStride <= St::kVec || tid + I * St::kStride < St::kVec);
  // (the source base is wave-uniform by construction; taken through readfirstlane because hipcc otherwise hands the "s"
  // operand of the inline asm a VGPR pair in some instantiations -- an assembler error, not a silent one)
  const unsigned long long a = reinterpret_cast<unsigned long long>(base + (size_t)frame0 * St::kFrame + (size_t)(I * St::kStride + 64 * wave) * 4);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  const float* src = reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
  if (active) lds_dma16s(src, (unsigned)(tid & 63) * 16u, stg + (I * St::kStride + 64 * wave) * 4);
}
// staging copy -> the registers tile_commit* take (transient: only while a tile is committed)
template <int C, int NTHR>
__device__ __forceinline__ void stage_load(const float* stg, int tid, f32x4 (&pre)[Stage<C, NTHR>::kPer]) {
  using St = Stage<C, NTHR>;
  const f32x4* s4 = reinterpret_cast<const f32x4*>(stg) + tid;
#pragma unroll
  for (int i = 0; i < St::kPer; ++i)
    if ((St::kStride == NTHR || tid < St::kStride) && ((i + 1) * St::kStride <= St::kVec || tid + i * St::kStride < St::kVec))
      pre[i] = s4[i * St::kStride];
}
// a tile cut short by the end of the batch: tile_fetch's bounds-checked loads, then into the staging copy
template <int C, int NTHR>
__device__ __forceinline__ void stage_store(float* stg, int tid, const f32x4 (&pre)[Stage<C, NTHR>::kPer]) {
  using St = Stage<C, NTHR>;
  f32x4* s4 = reinterpret_cast<f32x4*>(stg) + tid;
#pragma unroll
  for (int i = 0; i < St::kPer; ++i)
    if ((St::kStride == NTHR || tid < St::kStride) && tid + i * St::kStride < St::kVec) s4[i * St::kStride] = pre[i];
}
template <int I, int N, class F>
__device__ __forceinline__ void tm_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    tm_static_for<I + 1, N>(f);
  }
}
#ifndef RCED_TM_BWD_DMA
#define RCED_TM_BWD_DMA 0     // 1: bwd_fused_mfma stages the next tile by LDS-DMA into a raw LDS copy, issued by the wgrad half
                              // (no VGPR holds the next tile: no scratch).  Measured slower than VGPR staging in every form
                              // (all waves in a burst 2.99 / 2.05 ms, spread over the MFMA phase 3.29 / 2.13, wgrad half alone
                              // 2.99 / 2.24, against 2.86 / 1.94 ms for the two CR-CED shapes): kept as a switch for the record
#endif
#ifndef RCED_TM_NT
#define RCED_TM_NT 1          // the tile fetches are nontemporal loads (every staged byte is used once): -0.6 ... -1 % per kernel (A/B, round 3)
#endif
#ifndef RCED_TM_BWD_STAGGER
#define RCED_TM_BWD_STAGGER 1 // VGPR staging: the wgrad half issues its share of the next tile's loads at the start of the MFMA
                              // phase, the dgrad half its share BEHIND its MFMA pass (in front of its epilogue): two half-size
                              // bursts, each beside the other half's MFMAs, instead of one that stalls all eight waves
#endif
#ifndef RCED_TM_BWD_DEPTH
#define RCED_TM_BWD_DEPTH (RCED_TM_BWD_DMA ? 2 : 1)   // operand prefetch depth of the fused backward kernel's dgrad pass
#endif
#ifndef RCED_TM_SPREAD
#define RCED_TM_SPREAD 0   // 1: the next tile's loads are issued piece by piece inside the MFMA phase; 0: in one burst.
                           // Measured (round 3, CR-CED step): spread 55.1 ms, burst 51.5 ms -- a load that finds the memory
                           // pipeline's queues full blocks its wave, and with it that wave's MFMAs, piece after piece
#endif
// dst index of element (frame fr, offset r inside the frame) = base_row(fr) * ROWSTRIDE-style mapping given by MAP
// Input transform applied while committing a tile: the producer layer's BatchNorm + ReLU (module.py:28-33),
//   act = relu(a * z + b),   a = gamma * rstd,  b = beta - a * mu   (per channel; bn_act_fwd2 / bwd_route2 use the
// same folded form, so the forward value and the backward ReLU mask always agree),
// so that a plain conv+BN+ReLU layer's activation never has to be written to / read from HBM: its consumers
// (the next layer's forward conv and wgrad) read z and rebuild it here.  table = [a | b][C] in LDS.
template <int C>
__device__ __forceinline__ void xform_table_fill(float* table, const float* mu, const float* rstd, const float* gamma,
                                                 const float* beta, int tid) {
  if (tid < C) {
    const float a = gamma[tid] * rstd[tid];
    table[tid] = a;
    table[C + tid] = beta[tid] - a * mu[tid];
  }
}

template <int C, int NTHR = kThreads, class MAP>
__device__ __forceinline__ void tile_commit(float* lds, int tid, const f32x4 (&pre)[Stage<C, NTHR>::kPer], MAP map) {
  using St = Stage<C, NTHR>;
#pragma unroll
  for (int i = 0; i < St::kPer; ++i) {
    const int q = tid + i * St::kStride;
    if (q < St::kVec && tid < St::kStride) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int e = 4 * q + 2 * h;
        const int fr = e >= St::kFrame ? 1 : 0, r = e - (fr ? St::kFrame : 0);   // kTF = 2: a compare, not a division
        *reinterpret_cast<f32x2*>(lds + map(fr, r)) = f32x2{pre[i][2 * h], pre[i][2 * h + 1]};
      }
    }
  }
}
// The same with the BatchNorm + ReLU transform.  All of a thread's float4 start at channel c = 4 tid mod C
// (Stage<C>::kStride), so the two channel pairs' (a, b) are read from the table once, in front of the pieces.
template <int C, int NTHR = kThreads, class MAP>
__device__ __forceinline__ void tile_commit_bnrelu(float* lds, int tid, const f32x4 (&pre)[Stage<C, NTHR>::kPer], MAP map,
                                                   const float* table, int frame0, int frames) {
  using St = Stage<C, NTHR>;
  static_assert(St::kFrame % C == 0 && C % 2 == 0, "frames start at channel 0; float2 pieces stay inside a pixel");
  constexpr int kStep = (4 * St::kStride) % C;   // 0 with RCED_TM_FIXCH
  int c = (4 * tid) % C;
  f32x2 ta[2], tb[2];
  auto load_tables = [&]() {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int ch = c + 2 * h;
      if (ch >= C) ch -= C;
      ta[h] = *reinterpret_cast<const f32x2*>(table + ch);
      tb[h] = *reinterpret_cast<const f32x2*>(table + C + ch);
    }
  };
  if (kStep == 0) load_tables();
#pragma unroll
  for (int i = 0; i < St::kPer; ++i) {
    const int q = tid + i * St::kStride;
    if (kStep != 0) load_tables();
    if (q < St::kVec && tid < St::kStride) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int e = 4 * q + 2 * h;
        const int fr = e >= St::kFrame ? 1 : 0, r = e - (fr ? St::kFrame : 0);   // kTF = 2: a compare, not a division
        const f32x2 a = ta[h], b = tb[h];
        f32x2 v = {fmaxf(fmaf(a.x, pre[i][2 * h], b.x), 0.f), fmaxf(fmaf(a.y, pre[i][2 * h + 1], b.y), 0.f)};
        if (frame0 + fr >= frames) v = f32x2{0.f, 0.f};      // frames past the batch stay zero
        *reinterpret_cast<f32x2*>(lds + map(fr, r)) = v;
      }
    }
    c += kStep;
    if (c >= C) c -= C;
  }
}

// BatchNorm backward applied while committing a tile of d_u (the gradient w.r.t. the BN output, after the ReLU
// mask): dz = gamma*rstd * (d_u - S1/P - zhat * S2/P), zhat = (z - mu)*rstd, S1 = sum d_u, S2 = sum d_u*zhat
// (bn_bwd_apply2), folded per channel into  dz = A*d_u + B*z + C.  table = [A | B | C][C] in LDS.  With it the
// dz tensor never exists in HBM: the two kernels that consume it (wgrad, dgrad) read d_u and z instead.
struct BnBwdArgs {
  const float* z;          // pre-BatchNorm output of the layer, same layout as d_u
  const float *mu, *rstd, *gamma;
  const double* sums;      // [C][2] = (S1, S2)
  double P;                // pixels in the batch
  const float* beta;       // non-null: the first input is g (gradient w.r.t. the layer's OUTPUT) and the ReLU mask
                           // [gamma*rstd*z + beta - ... > 0] is applied here too: d_u = g * [a*z + b > 0]
};
template <int C>
__device__ __forceinline__ void bnbwd_table_fill(float* table, const BnBwdArgs& a, int tid) {
  if (tid < C) {
    const float gr = a.gamma[tid] * a.rstd[tid];
    const float m1 = (float)(a.sums[2 * tid] / a.P), m2 = (float)(a.sums[2 * tid + 1] / a.P);
    const float B = -gr * m2 * a.rstd[tid];
    table[tid] = gr;
    table[C + tid] = B;
    table[2 * C + tid] = -gr * m1 - B * a.mu[tid];
    table[3 * C + tid] = a.beta ? a.beta[tid] - gr * a.mu[tid] : 0.f;   // b of the folded forward a*z + b (a = gr)
  }
}
struct NoExtra {
  __device__ __forceinline__ void operator()(int, int, f32x2) const {}
};
// extra(fr, r, v): called with every committed pair (frame of the tile, offset in the frame's [129][C] row, value) -- the fused
// backward kernel keeps a second copy of dz as bf16 planes for its dgrad half through it
template <int C, int NTHR = kThreads, class MAP, class EXTRA = NoExtra>
__device__ __forceinline__ void tile_commit_bnbwd(float* lds, int tid, const f32x4 (&pd)[Stage<C, NTHR>::kPer],
                                                  const f32x4 (&pz)[Stage<C, NTHR>::kPer], MAP map, const float* table,
                                                  int frame0, int frames, bool mask, EXTRA extra = EXTRA()) {
  using St = Stage<C, NTHR>;
  static_assert(St::kFrame % C == 0 && C % 2 == 0, "frames start at channel 0; float2 pieces stay inside a pixel");
  constexpr int kStep = (4 * St::kStride) % C;   // 0 with RCED_TM_FIXCH: the tables are read once per commit
  int c = (4 * tid) % C;
  f32x2 tA[2], tB[2], tK[2], tF[2];
  auto load_tables = [&]() {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int ch = c + 2 * h;
      if (ch >= C) ch -= C;
      tA[h] = *reinterpret_cast<const f32x2*>(table + ch);
      tB[h] = *reinterpret_cast<const f32x2*>(table + C + ch);
      tK[h] = *reinterpret_cast<const f32x2*>(table + 2 * C + ch);
      tF[h] = *reinterpret_cast<const f32x2*>(table + 3 * C + ch);
    }
  };
  if (kStep == 0) load_tables();
#pragma unroll
  for (int i = 0; i < St::kPer; ++i) {
    const int q = tid + i * St::kStride;
    if (kStep != 0) load_tables();
    if (q < St::kVec && tid < St::kStride) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int e = 4 * q + 2 * h;
        const int fr = e >= St::kFrame ? 1 : 0, r = e - (fr ? St::kFrame : 0);   // kTF = 2: a compare, not a division
        const f32x2 A = tA[h], B = tB[h], K = tK[h];
        f32x2 d = {pd[i][2 * h], pd[i][2 * h + 1]};
        if (mask) {   // wave-uniform: the input is g, not d_u
          const f32x2 fb = tF[h];
          d.x = fmaf(A.x, pz[i][2 * h], fb.x) > 0.f ? d.x : 0.f;
          d.y = fmaf(A.y, pz[i][2 * h + 1], fb.y) > 0.f ? d.y : 0.f;
        }
        f32x2 v = {fmaf(A.x, d.x, fmaf(B.x, pz[i][2 * h], K.x)), fmaf(A.y, d.y, fmaf(B.y, pz[i][2 * h + 1], K.y))};
        if (frame0 + fr >= frames) v = f32x2{0.f, 0.f};      // frames past the batch stay zero
        *reinterpret_cast<f32x2*>(lds + map(fr, r)) = v;
        extra(fr, r, v);
      }
    }
    c += kStep;
    if (c >= C) c -= C;
  }
}

// Sum over the 16 lanes of a DPP row (lanes 16r .. 16r+15); the total is valid in lane 15 of the row.
__device__ __forceinline__ float row_sum16(float v) {
  // row_shr:1 / 2 / 4 / 8 (0x111, 0x112, 0x114, 0x118); bound_ctrl: lanes shifted in from outside the row read 0
#define RCED_ROW_SHR(x, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xf, 0xf, true))
  v += RCED_ROW_SHR(v, 0x111);
  v += RCED_ROW_SHR(v, 0x112);
  v += RCED_ROW_SHR(v, 0x114);
  v += RCED_ROW_SHR(v, 0x118);
#undef RCED_ROW_SHR
  return v;
}

// BatchNorm-backward sums of the PRODUCER of the tensor a dgrad writes its gradient for (SUMS kernels).  When that
// tensor is the output of a plain conv+BN+ReLU layer, the gradient g this dgrad writes is only ever used masked:
// d_u = g * [a*z + b > 0], and the layer's BatchNorm backward needs S1 = sum d_u and S2 = sum d_u * zhat over the
// whole batch before anything else can run.  bwd_route2 got them from one more pass over g and z (HBM-bound, 8 ms of a
// CR-CED step); here the dgrad forms them in its epilogue, where g is still in the accumulators: the tile of z arrives
// by LDS-DMA (no VGPRs, in flight during the whole MFMA pass), and the kernel leaves per-workgroup records
// (sum d_u, sum d_u * z) per channel in `part`; sums_fix turns the second into S2 = rstd * (sum d_u z - mu * S1).
struct SumArgs {
  const float* z;                          // pre-BatchNorm output of the producer, [frames][129][COUT]
  const float *mu, *rstd, *gamma, *beta;   // its batch statistics and affine parameters
};

// SUMX (with SUMS, the fused backward kernel): the masked sums are formed from the TRANSFORMED tile x = relu(a z + b) the
// same workgroup has staged for its wgrad half (zt = that tile, pixel p at row kG + p): x > 0 is the mask and the second
// sum is sum d_u * x (sums_fix_x turns it into S2); no z tile, no extra barrier.
// A lane's fp32 shares of the per-channel sums (STATS: sum z, sum z^2; SUMS: sum d_u, sum d_u z), carried over a few tiles
// in registers and only then reduced over the lanes and added to the wave's record of doubles: the reduction (64 DPP adds,
// 16 conversions, 16 double adds per wave and two-M-tile tile) was a fifth of the forward convolutions' non-MFMA
// instructions when done per tile.  kSumFlush tiles x <= 5 values per lane stay far inside fp32 (the round-2 form summed 5).
#ifndef RCED_TM_SUMFLUSH
#define RCED_TM_SUMFLUSH 4
#endif
constexpr int kSumFlush = RCED_TM_SUMFLUSH;
#ifndef RCED_TM_BWD_CARRY
#define RCED_TM_BWD_CARRY 0     // 1: the fused backward kernel carries its masked sums too -- it sits at 256 VGPRs (10-12 dwords of
                                // scratch with them) and gains nothing (2.407 / 1.895 vs 2.405 / 1.889 ms): off
#endif
template <int MT>
struct SumState {
  float p1[MT][4], p2[MT][4], pr1[2], pr2[2];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < 4; ++j) p1[mt][j] = p2[mt][j] = 0.f;
    pr1[0] = pr1[1] = pr2[0] = pr2[1] = 0.f;
  }
};
// ---------------------------------------------------------------------------------------------
// fp32 quality on the bf16 matrix pipe for the forward convolutions (kernels_final_x6.h): the input tile lives
// in LDS as THREE bf16 planes [pixel][channel] (x = h + m + l, exact to 2^-24; channel stride rounded to 4 so that a lane's
// eight consecutive k of a pixel's im2col window start 8-byte aligned), the packet holds the weights the same way
// ([step][M-tile][part][lane] x 8 bf16, k = 32 S + 8 kq + e), and a product is six v_mfma_f32_16x16x32_bf16 (m m, l h, h l,
// m h, h m, h h) into the fp32 accumulator: 6 x 16 cycles per K = 32 where the fp32 pipe takes 8 x 32.  The epilogue
// (conv_tile) does not change: same accumulator layout.  Built for the shapes without a remainder pass.
// ---------------------------------------------------------------------------------------------
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// channel stride of a plane (bf16 elements): the even-padded channel count rounded to 4 (8-byte aligned windows), plus 4 where a
// column's stride (ph pixels) would be a multiple of 128 bytes -- all 16 columns of a B fragment on the same banks
// (30 -> 32 channels with pixel-pair columns: measured 1.50 ms against 0.85 ms for the fp32 kernel)
#ifndef RCED_X6_ALIGN
#define RCED_X6_ALIGN 2      // channel stride granularity of the planes: 4 (8-byte aligned windows, two ds_read_b64 per fragment) or
                             // 2 (the even-padded channel count itself: no K padding per tap, four ds_read_b32 per fragment).
                             // Measured on the 18 -> 30 forward: K 100 -> 90 (four steps -> three): 0.85 -> 0.78 ms
#endif
__host__ __device__ constexpr int x6_cs(int cin, int ph) {
  const int cs = ((((cin + 1) & ~1) + RCED_X6_ALIGN - 1) / RCED_X6_ALIGN) * RCED_X6_ALIGN;
  return (ph * cs * 2) % 128 == 0 ? cs + RCED_X6_ALIGN : cs;
}
template <int CIN, int TAPS, int COUT>
struct GeoX6 {
  using G = Geo<CIN, TAPS, COUT>;
  static constexpr int kCS = x6_cs(CIN, G::kPH);                         // channel stride of a plane, bf16 elements
  static constexpr int kKX = (TAPS + G::kPH - 1) * kCS;                  // K of the packet (window of TAPS (+1) pixels)
  static constexpr int kSteps = (kKX + 31) / 32;
  static constexpr int kPlane = ((G::kInRows * kCS + 32 + 7) / 8) * 8;   // bf16 per part; + 32: the last step reads past its window
  static constexpr int kInFloats = ((3 * kPlane / 2 + 3) / 4) * 4;       // the three parts, counted in floats
  // remainder pass (G::kR channels past the first M-tile; Geo): rows (pixel phase, channel) over K = (TAPS + P - 1) * kCS
  static constexpr int kKXR = G::kR ? (TAPS + G::kP - 1) * kCS : 0, kStepsR = (kKXR + 31) / 32;
  static constexpr int kDataMainFloats = kSteps * G::kMTm * 3 * 64 * 4;  // A fragments: 16 bytes per (step, M-tile, part, lane)
  static constexpr int kDataFloats = kDataMainFloats + kStepsR * 3 * 64 * 4;
  static constexpr int kPacket = kDataFloats + 32;                       // + shift[32]
  static constexpr int kShiftOff = kDataFloats;
  static constexpr int kLdsFloats = kInFloats + kPacket;
  // Two workgroups per CU are what overlaps one's staging with the other's MFMAs.  The 30 -> 8 layer does not fit that way
  // (three planes of a 30-channel tile + its packet = 102 KB): with one workgroup per CU it ran 1.44 ms, with its A fragments
  // streamed from L2 instead (every wave re-reads 37 KB per two-frame tile: ~9 TB/s of L2 traffic over the chip) 1.07 ms,
  // against 0.85 ms for the fp32 kernel -- so it stays on the fp32 MFMA, and this form is built where it fits.
  static constexpr bool kFits = kLdsFloats * 4 <= 76 * 1024;
};
__device__ __forceinline__ f32x4 mfma32(s16x8 a, s16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// two floats -> their three bf16 parts, each pair packed in 4 bytes
__device__ __forceinline__ void split3_pair(f32x2 v, s16x2& h, s16x2& m, s16x2& l) {
  const bf16x2 bh = {(__bf16)v.x, (__bf16)v.y};
  const f32x2 r1 = {v.x - (float)bh.x, v.y - (float)bh.y};
  const bf16x2 bm = {(__bf16)r1.x, (__bf16)r1.y};
  const bf16x2 bl = {(__bf16)(r1.x - (float)bm.x), (__bf16)(r1.y - (float)bm.y)};
  h = __builtin_bit_cast(s16x2, bh);
  m = __builtin_bit_cast(s16x2, bm);
  l = __builtin_bit_cast(s16x2, bl);
}
// eight floats (four pairs) -> their three bf16 parts as MFMA fragments, in registers (whole-vector casts only: bit-casting the
// MEMBERS of a __bf16 vector miscompiles, tools/micro/split_test.hip)
struct X6Parts {
  s16x8 h, m, l;
};
__device__ __forceinline__ X6Parts x6_split8(const f32x2 (&q)[4]) {
  s16x2 ph[4], pm[4], pl[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) split3_pair(q[j], ph[j], pm[j], pl[j]);
  X6Parts r;
  r.h = s16x8{ph[0].x, ph[0].y, ph[1].x, ph[1].y, ph[2].x, ph[2].y, ph[3].x, ph[3].y};
  r.m = s16x8{pm[0].x, pm[0].y, pm[1].x, pm[1].y, pm[2].x, pm[2].y, pm[3].x, pm[3].y};
  r.l = s16x8{pl[0].x, pl[0].y, pl[1].x, pl[1].y, pl[2].x, pl[2].y, pl[3].x, pl[3].y};
  return r;
}
#ifndef RCED_TM_WG_X6_B
#define RCED_TM_WG_X6_B 1    // bwd_fused_mfma<18,5,30>: whole runs of eight pixel groups of the wgrad half in the three-part bf16 form:
                             // 2.38 -> 2.28 ms.  (The 30 -> 8 kernel's wgrad has ONE N-tile: its 19 A fragments per 32 pixel pairs would each be
                             // split for six MFMAs -- 36 VALU for 96 cycles of MFMA: no gain, not built.)
#endif
__host__ __device__ constexpr bool bwd_wg_x6(int cin, int taps, int cout) { return RCED_TM_WG_X6_B && cin == 18 && taps == 5 && cout == 30; }
// in: part 0 of the tile at pixel 0 (bf16), the other parts PLANE elements further; off0 / offx: this lane's window start of
// its first regular / its extra column tile; TSTRIDE: elements between a wave's consecutive column tiles
template <int NR, int NX, int MT, int STEPS, int TSTRIDE, int PLANE>
__device__ __forceinline__ void gemm_pass_x6(const unsigned short* in, int off0, int offx, const s16x8* wpk, int lane,
                                             f32x4 (&acc)[NR + NX][MT]) {
  constexpr int NT = NR + NX;
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    s16x8 a[MT][3], b[NT][3];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int p = 0; p < 3; ++p) a[mt][p] = wpk[((s * MT + mt) * 3 + p) * 64 + lane];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const unsigned short* q = in + (t < NR ? off0 + t * TSTRIDE : offx) + 32 * s + p * PLANE;
        if constexpr (RCED_X6_ALIGN == 4) {
          const s16x4 lo = *reinterpret_cast<const s16x4*>(q), hi = *reinterpret_cast<const s16x4*>(q + 4);   // 8-byte aligned
          b[t][p] = s16x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        } else {
          const s16x2 q0 = *reinterpret_cast<const s16x2*>(q), q1 = *reinterpret_cast<const s16x2*>(q + 2),
                      q2 = *reinterpret_cast<const s16x2*>(q + 4), q3 = *reinterpret_cast<const s16x2*>(q + 6);   // 4-byte aligned
          b[t][p] = s16x8{q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, q3.x, q3.y};
        }
      }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        f32x4 v = acc[t][mt];
        v = mfma32(a[mt][1], b[t][1], v);
        v = mfma32(a[mt][2], b[t][0], v);
        v = mfma32(a[mt][0], b[t][2], v);
        v = mfma32(a[mt][1], b[t][0], v);
        v = mfma32(a[mt][0], b[t][1], v);
        v = mfma32(a[mt][0], b[t][0], v);
        acc[t][mt] = v;
      }
  }
}
// Commit a prefetched tile into the three planes.  XF: the producer's BatchNorm + ReLU first (tile_commit_bnrelu).
// G0 = pixel row of bin 0 of frame 0, GAP = rows between the frames (the halo).  All of a thread's float4 start at the same
// channel (Stage::kStride), so its pixel walks in constant steps: one division and one table read per commit, not per piece.
template <int C, int CS, int PLANE, bool XF, int G0, int GAP>
__device__ __forceinline__ void tile_commit_x6(unsigned short* planes, int tid, const f32x4 (&pre)[Stage<C>::kPer],
                                               const float* table, int frame0, int frames) {
  using St = Stage<C>;
  static_assert(C % 2 == 0 && (4 * St::kStride) % C == 0 && St::kFrame % C == 0, "float2 pieces inside a pixel; constant pixel step");
  constexpr int kPxStep = 4 * St::kStride / C;
  const int gp0 = (4 * tid) / C, c0 = 4 * tid - gp0 * C;          // pixel (over both frames) and channel of the thread's pieces
  const bool wrap = c0 + 2 >= C;                                     // the second pair lies in the next pixel
  const int c1 = wrap ? c0 + 2 - C : c0 + 2;
  f32x2 ta[2], tb[2];
  if constexpr (XF) {
    ta[0] = *reinterpret_cast<const f32x2*>(table + c0);
    tb[0] = *reinterpret_cast<const f32x2*>(table + C + c0);
    ta[1] = *reinterpret_cast<const f32x2*>(table + c1);
    tb[1] = *reinterpret_cast<const f32x2*>(table + C + c1);
  }
#pragma unroll
  for (int i = 0; i < St::kPer; ++i) {
    const int q = tid + i * St::kStride;
    if (q < St::kVec && tid < St::kStride) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int gp = gp0 + i * kPxStep + (h && wrap ? 1 : 0);
        const int fr = gp >= kF ? 1 : 0;
        f32x2 v = {pre[i][2 * h], pre[i][2 * h + 1]};
        if constexpr (XF) {
          v = f32x2{fmaxf(fmaf(ta[h].x, v.x, tb[h].x), 0.f), fmaxf(fmaf(ta[h].y, v.y, tb[h].y), 0.f)};
          if (frame0 + fr >= frames) v = f32x2{0.f, 0.f};      // frames past the batch stay zero
        }
        s16x2 ph, pm, pl;
        split3_pair(v, ph, pm, pl);
        unsigned short* d = planes + (G0 + gp + (fr ? GAP : 0)) * CS + (h ? c1 : c0);
        *reinterpret_cast<s16x2*>(d) = ph;
        *reinterpret_cast<s16x2*>(d + PLANE) = pm;
        *reinterpret_cast<s16x2*>(d + 2 * PLANE) = pl;
      }
    }
  }
}
// The weights (TF layout [TAPS][CIN][COUT]) as the three-part packet of a FORWARD convolution (pack_packet's transpose = 0).
// transpose = 1: the dgrad's packet (pack_packet's convention: cin / cout are those of the conv being packed).
static __global__ void pack_packet_x6(const float* __restrict__ w, const float* __restrict__ shift, int taps, int cin, int cout,
                                      int ph, float* __restrict__ packet, int transpose = 0) {
  const int R = ph == 1 ? tm_rem(cout) : 0, P = tm_rem_p(R, (cin + 1) & ~1);   // (as Geo::kP)
  const int cs = x6_cs(cin, ph), K = (taps + ph - 1) * cs, steps = (K + 31) / 32, MT = R ? 1 : (cout + 15) / 16;
  const int KR = R ? (taps + P - 1) * cs : 0, stepsR = (KR + 31) / 32;
  const int nmain = steps * MT * 64 * 8, nrem = stepsR * 64 * 8;   // one thread per (S, mt, lane, e); the three parts by the same thread
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= nmain + nrem + 32) return;
  const int data_floats = (steps * MT + stepsR) * 3 * 64 * 4;
  if (idx >= nmain + nrem) {
    const int c = idx - nmain - nrem;
    float v;
    if (R) v = !shift ? 0.f : (c < 16 ? shift[c] : shift[16 + (c - 16) % R]);   // 16 main channels, 16 remainder rows
    else v = (shift && c < ph * cout) ? shift[ph == 2 ? (c & 7) : c] : 0.f;
    packet[data_floats + c] = v;
    return;
  }
  int k, co, tap_shift = 0, klim = K;
  size_t base;
  if (idx < nmain) {
    const int e = idx & 7, lane = (idx >> 3) & 63, r = idx >> 9, mt = r % MT, S = r / MT;
    k = 32 * S + 8 * (lane >> 4) + e;
    co = 16 * mt + (lane & 15);
    base = ((size_t)(S * MT + mt) * 3) * 512 + lane * 8 + e;
  } else {     // remainder pass: row i = (phase i / R, channel 16 + i % R) over K = (taps + P - 1) * cs
    const int q = idx - nmain, e = q & 7, lane = (q >> 3) & 63, S = q >> 9, i = lane & 15;
    k = 32 * S + 8 * (lane >> 4) + e;
    tap_shift = i / R;
    co = 16 + i % R;
    klim = tap_shift < P ? KR : 0;      // (rows of a dropped phase: zero)
    base = ((size_t)steps * MT * 3 + (size_t)S * 3) * 512 + lane * 8 + e;
  }
  int tap = k / cs;
  const int ci = k - tap * cs;
  tap -= tap_shift;
  if (ph == 2) {            // row = (parity, co): the parity-1 rows see the window one tap later
    tap -= co >> 3;
    co &= 7;
  }
  float v = 0.f;
  if (k < klim && co < cout && ci < cin && tap >= 0 && tap < taps)
    v = transpose ? w[((taps - 1 - tap) * cout + co) * cin + ci] : w[(tap * cin + ci) * cout + co];
  const __bf16 h = (__bf16)v;
  const float r1 = v - (float)h;
  const __bf16 m = (__bf16)r1;
  const __bf16 l = (__bf16)(r1 - (float)m);
  unsigned short* p16 = reinterpret_cast<unsigned short*>(packet);
  p16[base] = __builtin_bit_cast(unsigned short, h);
  p16[base + 512] = __builtin_bit_cast(unsigned short, m);
  p16[base + 1024] = __builtin_bit_cast(unsigned short, l);
}

template <bool X6, int CIN, int TAPS, int COUT>
constexpr int conv_shift_off() {      // floats in front of the packet's shift[32]
  if constexpr (X6) return GeoX6<CIN, TAPS, COUT>::kShiftOff;
  else return Geo<CIN, TAPS, COUT>::kData;
}
template <int CIN, int TAPS, int COUT, bool ACCUM, bool STATS, int NX, bool SUMS = false, bool OPQ = false, bool SUMX = false,
          class Each = chain::NoEach, bool EXTACC = false, int DEPTH = 2, bool KS = false, int XMT = -1, bool X6 = false>
__device__ __forceinline__ void conv_tile(const float* lds_in, const float* lds_w, float* __restrict__ out, int frame0,
                                          int frames, int wave, int lane,
                                          double* red_wave, const float* zt = nullptr, const float* stab = nullptr,
                                          Each each = Each(), f32x4* acc_store = nullptr, const f32x4* ks_own = nullptr,
                                          const float* ks_lds = nullptr) {
  SumState<Geo<CIN, TAPS, COUT>::kMTm> local_sums;
  conv_tile<CIN, TAPS, COUT, ACCUM, STATS, NX, SUMS, OPQ, SUMX, Each, EXTACC, DEPTH, KS, XMT, X6>(
      lds_in, lds_w, out, frame0, frames, wave, lane, red_wave, zt, stab, each, acc_store, ks_own, ks_lds, local_sums, false, true);
}
template <int CIN, int TAPS, int COUT, bool ACCUM, bool STATS, int NX, bool SUMS = false, bool OPQ = false, bool SUMX = false,
          class Each = chain::NoEach, bool EXTACC = false, int DEPTH = 2, bool KS = false, int XMT = -1, bool X6 = false>
__device__ __forceinline__ void conv_tile(const float* lds_in, const float* lds_w, float* __restrict__ out, int frame0,
                                          int frames, int wave, int lane, double* red_wave, const float* zt, const float* stab,
                                          Each each, f32x4* acc_store, const f32x4* ks_own, const float* ks_lds,
                                          SumState<Geo<CIN, TAPS, COUT>::kMTm>& S, bool carried, bool flush,
                                          long acc_delta = 0) {
  // acc_delta (ACCUM): the tensor the result is added to sits acc_delta floats from `out` -- out = acc_from + conv instead of
  // out += conv: a gradient tensor whose other contribution is a plain copy (a post-ReLU skip) is never copied, the dgrad
  // that completes it reads the copy's source instead (train_api.hip, skip_alias)
  using G = Geo<CIN, TAPS, COUT>;
  // SUMS reads the COMPLETE gradient out of the accumulators: an overwriting dgrad, or (z form) the accumulating dgrad that
  // adds the last contribution (ACCUM: the operand is added in front of the sums; train_api.hip launches it only there)
  static_assert(!(SUMS && (STATS || (COUT & 1))), "SUMS: dgrads with an even cout");
  static_assert(!(SUMS && SUMX && (ACCUM || G::kPH != 1)), "SUMX: the fused backward kernel's overwriting, unpaired dgrad");
  constexpr int NR = G::kRegular, NT = NR + NX, MT = G::kMTm, PH = G::kPH;   // MT: M-tiles of the MAIN pass
  static_assert(!X6 || (!KS && XMT < 0 && (!SUMS || SUMX) && !ACCUM), "the three-part bf16 form: overwriting, no K split");
  constexpr int kShiftOff = conv_shift_off<X6, CIN, TAPS, COUT>();
  const float* in = lds_in + G::kG * G::kCinP;
  // XMT >= 0 (two-M-tile shapes, RCED_TM_MSPLIT): the odd column tile is cut by M-tile -- this wave's extra slot computes
  // and stores only M-tile XMT of column tile 4 NR; another wave has the other half
  const int xwave = XMT >= 0 ? 0 : wave;
  const int xtile = NR * kWaves + xwave;
  // acc_store: the caller's registers for the tile's accumulators (bwd_fused_mfma shares them with its wgrad half's
  // kernel-lifetime accumulators: a wave has one role, but two arrays are both live in every wave for the allocator)
  f32x4 acc_local[EXTACC ? 1 : NT][EXTACC ? 1 : MT];
  f32x4 (&acc)[NT][MT] = *reinterpret_cast<f32x4 (*)[NT][MT]>(EXTACC ? acc_store : &acc_local[0][0]);
#if RCED_TM_STAMPS
  const bool st_on = SUMS && (CIN == 30 || SUMX) && blockIdx.x == 0;
  unsigned long long c0 = 0;
#endif
  {
    const int n = lane & 15, kq = lane >> 4;
    const int px0 = 16 * wave + n, pxx = 16 * xtile + n;     // column index: a pixel, or a pixel pair when PH = 2
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const f32x4 sh = *reinterpret_cast<const f32x4*>(lds_w + kShiftOff + 16 * mt + 4 * kq);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t][mt] = sh;
    }
#if RCED_TM_STAMPS
    c0 = st_on ? tm_stamp() : 0;
#endif
    if constexpr (X6) {
      using GX = GeoX6<CIN, TAPS, COUT>;
      const unsigned short* inx = reinterpret_cast<const unsigned short*>(lds_in) + G::kG * GX::kCS;
      if (!(RCED_TM_EXP & 4))
        gemm_pass_x6<NR, NX, MT, GX::kSteps, PH * 64 * GX::kCS, GX::kPlane>(inx, (PH * px0 - G::kG) * GX::kCS + 8 * kq,
                                                                          (PH * pxx - G::kG) * GX::kCS + 8 * kq,
                                                                          reinterpret_cast<const s16x8*>(lds_w), lane, acc);
    } else if constexpr (KS) {
      // K-split odd tile (conv_ks_partial): this pass covers the regular slots only; the four waves' shares of the odd
      // tile meet in LDS behind a barrier every wave passes here, and wave 0 (NX = 1) takes the tile through the epilogue
      if (!(RCED_TM_EXP & 4))
        chain::gemm_pass<NR, 0, MT, G::kKP, PH * 64 * G::kCinP, DEPTH, -1, chain::NoPre, Each>(
            in, (PH * px0 - G::kG) * G::kCinP + 2 * kq, 0, lds_w, lane, *reinterpret_cast<f32x4 (*)[NR][MT]>(&acc[0][0]), chain::NoPre(), each);
      __syncthreads();
      if constexpr (NX == 1) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          f32x4 v = ks_own[mt];
#pragma unroll
          for (int w = 0; w < kWaves - 1; ++w) v += *reinterpret_cast<const f32x4*>(ks_lds + ((w * MT + mt) * 64 + lane) * 4);
          acc[NR][mt] = v;
        }
      }
    } else {
    if (!(RCED_TM_EXP & 4))
      chain::gemm_pass<NR, NX, MT, G::kKP, PH * 64 * G::kCinP, DEPTH, XMT, chain::NoPre, Each>(
          in, (PH * px0 - G::kG) * G::kCinP + 2 * kq, (PH * pxx - G::kG) * G::kCinP + 2 * kq, lds_w, lane, acc, chain::NoPre(), each);
    }
    each(-1);   // behind the pass, in front of the epilogue (bwd_fused_mfma's dgrad half issues its loads here)
  }
  // The epilogue's lane coordinates are re-derived behind an opaque barrier: left visible, hipcc hoists every per-lane
  // address of the epilogue (pixel -> frame / bin splits, LDS offsets of the z tile, 64-bit row offsets) out of the tile
  // loop, keeps them live across the MFMA pass and spills them; each reload in the epilogue is a scratch load +
  // s_waitcnt vmcnt(0), which also waits for the global stores issued just before it -- a full HBM write round trip per
  // reload, 15.8 k cycles of epilogue per tile against 10.5 k of MFMA pass (s_memtime stamps, RCED_TM_STAMPS).
  // Only where the prefetched next tile is large enough for that to happen (OPQ): the kernels that have the registers
  // lose 5-10 % to the recomputation.
  int le = lane;
  if constexpr (OPQ) asm volatile("" : "+v"(le));
  const int n = le & 15, kq = le >> 4;
  const int px0 = 16 * wave + n, pxx = 16 * xtile + n;
  // the lane's shares of sum z, sum z^2 (SUMS: sum d_u, sum d_u z), fp32: the caller's (carried over kSumFlush tiles) or local
  // S: the caller's (carried over kSumFlush tiles) or the wrapper's local one -- by reference, never through a pointer select
  float (&p1)[MT][4] = S.p1;
  float (&p2)[MT][4] = S.p2;
  float (&pr1)[2] = S.pr1;
  float (&pr2)[2] = S.pr2;
  f32x4 sa4[MT], sb4[MT];       // SUMS: folded BatchNorm a, b of this lane's four channels per M-tile
#if RCED_TM_STAMPS
  if (st_on) { const unsigned long long n = tm_stamp(); if (lane == 0) g_tm[wave][0] += n - c0; c0 = n; }
#endif
  if constexpr (SUMS && !SUMX) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the z tile have landed in LDS ...
#if RCED_TM_STAMPS
    if (st_on) { const unsigned long long n = tm_stamp(); if (lane == 0) g_tm[wave][3] += n - c0; c0 = n; }
#endif
    __syncthreads();                                    // ... and everybody else's
#if RCED_TM_STAMPS
    if (st_on) { const unsigned long long n = tm_stamp(); if (lane == 0) g_tm[wave][1] += n - c0; c0 = n; }
#endif
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int co0 = PH == 2 ? 4 * (kq & 1) : 16 * mt + 4 * kq;      // (PH = 2: lane rows 4 kq.. = parity kq >> 1, channels 4 (kq & 1)..)
      sa4[mt] = sb4[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (co0 + 1 < COUT) {
        const f32x2 a = *reinterpret_cast<const f32x2*>(stab + co0), b = *reinterpret_cast<const f32x2*>(stab + COUT + co0);
        sa4[mt].x = a.x; sa4[mt].y = a.y; sb4[mt].x = b.x; sb4[mt].y = b.y;
      }
      if (co0 + 3 < COUT) {
        const f32x2 a = *reinterpret_cast<const f32x2*>(stab + co0 + 2), b = *reinterpret_cast<const f32x2*>(stab + COUT + co0 + 2);
        sa4[mt].z = a.x; sa4[mt].w = a.y; sb4[mt].z = b.x; sb4[mt].w = b.y;
      }
    }
  }
  if constexpr (STATS || SUMS)
    if (!carried) S.zero();
#if RCED_TM_STAMPS
  unsigned long long e0 = st_on ? tm_stamp() : 0, ea[4] = {0, 0, 0, 0};
#define TM_E(i) do { if (st_on) { const unsigned long long n_ = tm_stamp(); ea[i] += n_ - e0; e0 = n_; } } while (0)
#else
#define TM_E(i)
#endif
  // Slot t of this wave is column tile k = wave + 4 t (the extra slot is t = NR): pixels [16 PH k, 16 PH (k + 1)), a
  // wave-uniform range.  15 of a tile's 17 column tiles lie inside ONE frame (no gap pixel, nothing past the tile): for those
  // the frame index is a scalar, every lane is valid, and a lane's global / LDS addresses are one per-lane offset (the
  // same for every slot) plus wave-uniform terms -- no per-lane pixel -> (frame, bin) split, no validity masks, no 64-bit
  // lane arithmetic (what used to be hoisted out of the tile loop for every slot and spilled).  The two mixed tiles keep
  // the per-lane path below.
#ifndef RCED_TM_CLEAN
#define RCED_TM_CLEAN 1
#endif
  constexpr bool kCleanPath = RCED_TM_CLEAN && (COUT % 2 == 0) && !(SUMS && !SUMX) && !(RCED_TM_EXP & 2);
  const int px_lane0 = PH == 2 ? 2 * px0 + (kq >> 1) : px0;                      // this lane's pixel in slot 0
  const unsigned lane_b = (unsigned)(px_lane0 * COUT + (PH == 2 ? 4 * (kq & 1) : 4 * kq)) * 4u;   // bytes
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if constexpr (kCleanPath) {
      const bool xslot = XMT >= 0 && t == NR;                        // the M-split odd tile: column tile 4 NR for every wave
      const int k = (xslot ? 0 : wave) + kWaves * t, pbeg = 16 * PH * k, pend = pbeg + 16 * PH - 1;
      const bool in0 = pend < kF, in1 = pbeg >= G::kS && pend < G::kS + kF;
      if (in0 || in1) {
        const int fr = in1 ? 1 : 0;
        if (frame0 + fr >= frames) continue;
        if constexpr (STATS) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              if (XMT >= 0 && t == NR && mt != XMT) continue;
              const float v = acc[t][mt][j];
              p1[mt][j] += v;
              p2[mt][j] = fmaf(v, v, p2[mt][j]);
            }
        }
        static_assert(XMT < 0 || !SUMS, "M-split odd tiles: forward shapes only");
        if constexpr (SUMS && SUMX) {
          const char* zb = reinterpret_cast<const char*>(zt) + lane_b;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            constexpr int dummy = 0; (void)dummy;
            const int imm = ((G::kG + 64 * t) * COUT + 16 * mt) * 4;
            const int co0 = 16 * mt + 4 * kq;
            f32x4 xv = {0.f, 0.f, 0.f, 0.f};
            if (16 * mt + 16 <= COUT || co0 + 1 < COUT) { const f32x2 q = *reinterpret_cast<const f32x2*>(zb + imm); xv.x = q.x; xv.y = q.y; }
            if (16 * mt + 16 <= COUT || co0 + 3 < COUT) { const f32x2 q = *reinterpret_cast<const f32x2*>(zb + imm + 8); xv.z = q.x; xv.w = q.y; }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              if (16 * mt + 16 > COUT && co0 + (j | 1) >= COUT) continue;
              const float du = xv[j] > 0.f ? acc[t][mt][j] : 0.f;
              p1[mt][j] += du;
              p2[mt][j] = fmaf(du, xv[j], p2[mt][j]);
            }
          }
        }
        // wave-uniform base of this slot (scalar registers) + the lane offset: the saddr form of the global accesses
        char* ob = reinterpret_cast<char*>(out + (size_t)(frame0 + fr) * (kF * COUT)) +
                   ((64 * PH * t - (xslot ? 16 * PH * wave : 0)) * COUT - (in1 ? G::kS * COUT : 0)) * 4;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          if (XMT >= 0 && t == NR && mt != XMT) continue;
          const f32x4 v = acc[t][mt];
          char* p = ob + 16 * mt * 4 + lane_b;
          if (PH == 2 || 16 * mt + 16 <= COUT) {
            f32x4 r = v;
            if (ACCUM) r += *reinterpret_cast<const f32x4u*>(p + 4 * acc_delta);
            *reinterpret_cast<f32x4u*>(p) = r;
          } else {
            const int co0 = 16 * mt + 4 * kq;
            if (co0 + 3 < COUT) {
              f32x4 r = v;
              if (ACCUM) r += *reinterpret_cast<const f32x4u*>(p + 4 * acc_delta);
              *reinterpret_cast<f32x4u*>(p) = r;
            } else if (co0 + 1 < COUT) {
              f32x2 r = {v.x, v.y};
              if (ACCUM) { const f32x2 o = *reinterpret_cast<const f32x2*>(p + 4 * acc_delta); r.x += o.x; r.y += o.y; }
              *reinterpret_cast<f32x2*>(p) = r;
            }
          }
        }
        continue;
      }
    }
    const int col = t < NR ? px0 + 64 * t : pxx;
    const int px = PH == 2 ? 2 * col + (kq >> 1) : col;   // PH = 2: lane rows 4kq.. = parity kq >> 1, couts 4 (kq & 1)..
    const int fr = px >= G::kS ? 1 : 0, f = px - (fr ? G::kS : 0);   // kTF = 2 (px < 2 kS + 16 is checked below)
    TM_E(0);
    if (px >= G::kNPX || f >= kF || frame0 + fr >= frames) continue;
    if constexpr (STATS) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (XMT >= 0 && t == NR && mt != XMT) continue;
          const float v = acc[t][mt][j];
          p1[mt][j] += v;
          p2[mt][j] = fmaf(v, v, p2[mt][j]);
        }
    }
    if constexpr (SUMS && SUMX) {
      const float* zp = zt + (G::kG + fr * G::kS + f) * COUT;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int co0 = 16 * mt + 4 * kq;
        f32x4 xv = {0.f, 0.f, 0.f, 0.f};
        if (co0 + 1 < COUT) { const f32x2 q = *reinterpret_cast<const f32x2*>(zp + co0); xv.x = q.x; xv.y = q.y; }
        if (co0 + 3 < COUT) { const f32x2 q = *reinterpret_cast<const f32x2*>(zp + co0 + 2); xv.z = q.x; xv.w = q.y; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (co0 + (j | 1) >= COUT) continue;
          const float du = xv[j] > 0.f ? acc[t][mt][j] : 0.f;
          p1[mt][j] += du;
          p2[mt][j] = fmaf(du, xv[j], p2[mt][j]);
        }
      }
    }
    if constexpr (SUMS && ACCUM) {       // complete the gradient first: the sums are taken of what is stored
      const float* ap = out + ((size_t)(frame0 + fr) * kF + f) * COUT + acc_delta;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int co0 = PH == 2 ? 4 * (kq & 1) : 16 * mt + 4 * kq;
        if (co0 + 3 < COUT) acc[t][mt] += *reinterpret_cast<const f32x4u*>(ap + co0);
        else if (co0 + 1 < COUT) { const f32x2 o = *reinterpret_cast<const f32x2*>(ap + co0); acc[t][mt].x += o.x; acc[t][mt].y += o.y; }
      }
    }
    if constexpr (SUMS && !SUMX && !(RCED_TM_EXP & 16)) {
      const float* zp = zt + (fr * kF + f) * COUT;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int co0 = PH == 2 ? 4 * (kq & 1) : 16 * mt + 4 * kq;
        f32x4 zv = {0.f, 0.f, 0.f, 0.f};
        if (co0 + 1 < COUT) { const f32x2 q = *reinterpret_cast<const f32x2*>(zp + co0); zv.x = q.x; zv.y = q.y; }
        if (co0 + 3 < COUT) { const f32x2 q = *reinterpret_cast<const f32x2*>(zp + co0 + 2); zv.z = q.x; zv.w = q.y; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (co0 + (j | 1) >= COUT) continue;   // pairs: (0,1) need co0+1 < COUT, (2,3) need co0+3 < COUT
          const float du = fmaf(sa4[mt][j], zv[j], sb4[mt][j]) > 0.f ? acc[t][mt][j] : 0.f;
          p1[mt][j] += du;
          p2[mt][j] = fmaf(du, zv[j], p2[mt][j]);
        }
      }
    }
    TM_E(1);
    float* op = out + ((size_t)(frame0 + fr) * kF + f) * COUT;
    if ((RCED_TM_EXP & 2) && acc[t][0][0] != 12345.678f) continue;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (XMT >= 0 && t == NR && mt != XMT) continue;
      const int co0 = PH == 2 ? 4 * (kq & 1) : 16 * mt + 4 * kq;
      f32x4 v = acc[t][mt];
      if (co0 + 1 < COUT || (co0 < COUT && (COUT & 1))) {
        // the lane's four channels are 16 contiguous bytes of the pixel's row: ONE 16-byte store (dword-aligned: the row
        // stride is a multiple of 8 bytes only; global_store_dwordx4 takes that) instead of two 8-byte ones -- half the
        // vector-memory instructions of the epilogue, which queue behind the next tile's loads; a last pair: 8 bytes
        if constexpr ((COUT & 1) == 0) {
          if (co0 + 3 < COUT) {
            f32x4u* p = reinterpret_cast<f32x4u*>(op + co0);
            f32x4 r = v;
            if (ACCUM && !SUMS) { const f32x4 o = *reinterpret_cast<const f32x4u*>(reinterpret_cast<const float*>(p) + acc_delta); r += o; }
            *p = r;
          } else if (co0 + 1 < COUT) {
            f32x2* p = reinterpret_cast<f32x2*>(op + co0);
            f32x2 r = {v.x, v.y};
            if (ACCUM && !SUMS) { const f32x2 o = *reinterpret_cast<const f32x2*>(reinterpret_cast<const float*>(p) + acc_delta); r.x += o.x; r.y += o.y; }
            *p = r;
          }
        } else {
          const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (co0 + j < COUT) op[co0 + j] = (ACCUM ? op[co0 + j + acc_delta] : 0.f) + vv[j];
        }
      }
    }
    TM_E(2);
  }
  // ---- remainder pass (Geo::kR = 2): channels 16, 17 of 8 adjacent pixels per column.  Remainder tile rt goes to wave
  // 3 - rt % 4 (wave 0 carries the odd main tile).  Lane (column n, kq) ends up with rows 4 kq + j = (phase 2 kq + j / 2,
  // channel 16 + j % 2): two pixels x two channels.  Three tiles per two-frame tile, so everything here is per lane.
  if constexpr (G::kR > 0) {
    static_assert(G::kR == 2 && (COUT & 1) == 0 && !KS, "the 18-channel form");
    constexpr int P = G::kP, KR = G::kKR;
    const float* wr = lds_w + G::kDataMain;
    const f32x4 rsh = *reinterpret_cast<const f32x4*>(lds_w + kShiftOff + 16 + 4 * kq);
#pragma unroll 1
    for (int rt = 3 - wave; rt < G::kNRT; rt += kWaves) {
      const int pb = P * (16 * rt + n);                    // first pixel of this lane's column
      f32x4 racc[1][1] = {{rsh}};
      if constexpr (X6) {
        using GX = GeoX6<CIN, TAPS, COUT>;
        const unsigned short* inx = reinterpret_cast<const unsigned short*>(lds_in) + G::kG * GX::kCS;
        gemm_pass_x6<1, 0, 1, GX::kStepsR, 0, GX::kPlane>(inx, (pb - G::kG) * GX::kCS + 8 * kq, 0,
                                                          reinterpret_cast<const s16x8*>(lds_w) + GX::kSteps * MT * 3 * 64, lane, racc);
      } else
      chain::gemm_pass<1, 0, 1, KR, 0, 1>(in, (pb - G::kG) * G::kCinP + 2 * kq, 0, wr, lane, racc);
      const float vv[4] = {racc[0][0].x, racc[0][0].y, racc[0][0].z, racc[0][0].w};
#pragma unroll
      for (int h = 0; h < 2; ++h) {                         // the lane's two pixels
        if (2 * kq + h >= P) continue;                       // (a dropped phase: tm_rem_p)
        const int px = pb + 2 * kq + h;
        const int fr = px >= G::kS ? 1 : 0, f = px - (fr ? G::kS : 0);
        if (px >= G::kNPX || f >= kF || frame0 + fr >= frames) continue;
        const f32x2 v = {vv[2 * h], vv[2 * h + 1]};
        if constexpr (STATS) {
          pr1[0] += v.x; pr2[0] = fmaf(v.x, v.x, pr2[0]);
          pr1[1] += v.y; pr2[1] = fmaf(v.y, v.y, pr2[1]);
        }
        if constexpr (SUMS && SUMX) {
          const f32x2 xv = *reinterpret_cast<const f32x2*>(zt + (G::kG + px) * COUT + 16);
          const float d0 = xv.x > 0.f ? v.x : 0.f, d1 = xv.y > 0.f ? v.y : 0.f;
          pr1[0] += d0; pr2[0] = fmaf(d0, xv.x, pr2[0]);
          pr1[1] += d1; pr2[1] = fmaf(d1, xv.y, pr2[1]);
        }
        if constexpr (SUMS && !SUMX) {
          const f32x2 zv = *reinterpret_cast<const f32x2*>(zt + (fr * kF + f) * COUT + 16);
          const f32x2 a2 = *reinterpret_cast<const f32x2*>(stab + 16), b2 = *reinterpret_cast<const f32x2*>(stab + COUT + 16);
          const float d0 = fmaf(a2.x, zv.x, b2.x) > 0.f ? v.x : 0.f, d1 = fmaf(a2.y, zv.y, b2.y) > 0.f ? v.y : 0.f;
          pr1[0] += d0; pr2[0] = fmaf(d0, zv.x, pr2[0]);
          pr1[1] += d1; pr2[1] = fmaf(d1, zv.y, pr2[1]);
        }
        f32x2* op2 = reinterpret_cast<f32x2*>(out + ((size_t)(frame0 + fr) * kF + f) * COUT + 16);
        f32x2 r = v;
        if (ACCUM) { const f32x2 o = *reinterpret_cast<const f32x2*>(reinterpret_cast<const float*>(op2) + acc_delta); r.x += o.x; r.y += o.y; }
        *op2 = r;
      }
    }
  }
#if RCED_TM_STAMPS
  if (st_on && lane == 0) for (int i = 0; i < 3; ++i) g_tm2[wave][i] += ea[i];
#endif
  if constexpr (STATS || SUMS) if (flush || !carried) {      // (sums that are not carried leave with every tile)
    // Running per-channel sums live in this wave's LDS record as doubles (in registers they cost 32 VGPRs for the whole
    // kernel, which is what decided the occupancy): add the tile's fp32 shares over the 16 pixel lanes of a row with
    // DPP shifts (lane 15 of the row ends up with the sum), then that lane adds them to red_wave[channel][2].
    const bool writer = (lane & 15) == 15 && (PH == 1 || lane < 32);
    float ra[MT][4], rb[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float a = row_sum16(p1[mt][j]), b = row_sum16(p2[mt][j]);
        if constexpr (PH == 2) {   // lanes kq and kq ^ 2 hold the two pixel parities of the same channels
          a += __shfl_xor(a, 32, 64);
          b += __shfl_xor(b, 32, 64);
        }
        ra[mt][j] = a;
        rb[mt][j] = b;
      }
#if RCED_TM_STAMPS
    if (st_on) { const unsigned long long n = tm_stamp(); if (lane == 0) g_tm[wave][2] += n - c0; c0 = n; }
#endif
    if constexpr (G::kR > 0) {   // channels 16, 17: every lane holds a share -- sum over the whole wave; lane 63 adds it to the record
      float q[4] = {row_sum16(pr1[0]), row_sum16(pr2[0]), row_sum16(pr1[1]), row_sum16(pr2[1])};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        q[u] += __shfl_xor(q[u], 16, 64);      // lanes 15 / 31 / 47 / 63 hold the row sums
        q[u] += __shfl_xor(q[u], 32, 64);
      }
      if (lane == 63) {
        typedef double f64x2r __attribute__((ext_vector_type(2)));
        f64x2r c16 = *reinterpret_cast<const f64x2r*>(red_wave + 2 * 16), c17 = *reinterpret_cast<const f64x2r*>(red_wave + 2 * 17);
        *reinterpret_cast<f64x2r*>(red_wave + 2 * 16) = f64x2r{c16.x + (double)q[0], c16.y + (double)q[1]};
        *reinterpret_cast<f64x2r*>(red_wave + 2 * 17) = f64x2r{c17.x + (double)q[2], c17.y + (double)q[3]};
      }
    }
    if (writer) {   // all reads, then all writes: one LDS round trip instead of one per value
      typedef double f64x2 __attribute__((ext_vector_type(2)));
      f64x2 cur[MT][4];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          cur[mt][j] = *reinterpret_cast<const f64x2*>(red_wave + 2 * ((PH == 2 ? 0 : 16 * mt) + 4 * kq + j));
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<f64x2*>(red_wave + 2 * ((PH == 2 ? 0 : 16 * mt) + 4 * kq + j)) =
              f64x2{cur[mt][j].x + (double)ra[mt][j], cur[mt][j].y + (double)rb[mt][j]};
    }
    if (carried) S.zero();
  }
}

// SUMS: the z tile of the frames a workgroup is about to write gradients for, global -> LDS by LDS-DMA (contiguous in
// both; 16 bytes per lane, 1 KiB per wave instruction).  A tile cut short by the end of the batch is copied with
// ordinary loads instead (a 16-byte piece could straddle the end of the tensor).
template <int COUT>
__device__ __forceinline__ void ztile_fetch(const float* __restrict__ z, float* zt, int frame0, int frames, int tid) {
  using St = Stage<COUT>;
  const float* src = z + (size_t)frame0 * St::kFrame;
  if (frames - frame0 >= kTF) {
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;   // the LDS destination goes through M0
    constexpr int chunks = (St::kVec + 63) / 64;
    for (int c = wave; c < chunks; c += kWaves) {
      const int idx = c * 64 + lane;
      if (idx < St::kVec) lds_dma16(src + (size_t)idx * 4, zt + c * 256);
    }
  } else {
    const int nvalid = (frames - frame0) * St::kFrame;
    for (int e = tid; e < nvalid; e += kThreads) zt[e] = src[e];
  }
}

// in [frames][129][CIN], out [frames][129][COUT].  Persistent over tiles of kTF frames.
// STATS: also emit this workgroup's per-channel (sum z, sum z^2) of what it wrote, as doubles, into
// part[blockIdx.x][COUT][2] -- the batch-norm statistics, without another pass over z.
// XF: `in` is the producer's pre-BatchNorm z and xa its statistics / affine parameters (see tile_commit_bnrelu).
struct XformArgs {
  const float *mu, *rstd, *gamma, *beta;
};
constexpr int kXfNone = 0, kXfBnRelu = 1, kXfBnBwd = 2;
// LDS floats in front of the SUMS areas (z tile, then [a | b][COUT]): the staging buffers and the XF tables, 16-byte aligned
template <int CIN, int TAPS, int COUT, int XF>
constexpr int conv_sums_off() {
  return (Geo<CIN, TAPS, COUT>::kLdsFloats + (XF == kXfBnRelu ? 2 * CIN : XF == kXfBnBwd ? 4 * CIN : 0) + 3) & ~3;
}
template <int CIN, int TAPS, int COUT, int XF, bool SUMS>
constexpr int conv_red_off() {   // even float offset: the records are doubles
  return conv_sums_off<CIN, TAPS, COUT, XF>() + (SUMS ? kTF * kF * COUT + ((2 * COUT + 3) & ~3) : 0);
}
constexpr int kConvRedFloats = kWaves * 64 * 2;
// K-split of the odd column tile (RCED_TM_KSPLIT).  A tile of two frames is 17 column tiles (9 with pixel-pair columns)
// for four waves: 5 / 4 / 4 / 4 (3 / 2 / 2 / 2), and with two workgroups per CU both heavy waves sit on SIMD 0 -- the
// end-of-tile barrier waited for 25 % (50 %) more MFMAs than the mean.  Now every wave runs a QUARTER of the odd tile's K
// steps FIRST (its own small pass; wave 0 starts from the shift, wave 3 also takes the b32 tail), waves 1..3 park their
// partial sums in LDS, and behind the barrier that follows the regular pass wave 0 adds them up (in wave order: the same
// bits every run) and takes the tile through the epilogue: 4.25 (2.25) tiles of MFMAs everywhere.
// Measured (round 3, CR-CED step, A/B on one box): the 9-tile kernels' forward (30 -> 8: 3 / 2 / 2 / 2) 1.012 -> 0.898 ms; the
// 17-tile shapes do not gain (18 -> 30 forward 1.049 -> 1.111 ms: 25 more VGPRs and the odd tile's A fragments read four
// times cost more than 5 / 4 / 4 / 4 does), so the split is used where a wave would otherwise carry >= 1.5 x the mean.
#ifndef RCED_TM_KSPLIT
#define RCED_TM_KSPLIT 1
#endif
#ifndef RCED_TM_MSPLIT
#define RCED_TM_MSPLIT 0     // 1: two-M-tile shapes cut the odd column tile by M-tile over two waves (see conv1xk_mfma).  Measured:
                             // 18 -> 30 forward 1.077 -> 1.10 ms -- the kernel is not bound by its fullest SIMD; off
#endif
template <int CIN, int TAPS, int COUT>
constexpr bool conv_ks_on() {
  using G = Geo<CIN, TAPS, COUT>;
  return RCED_TM_KSPLIT && G::kExtra == 1 && G::kRegular <= 2 && CIN % 2 == 0;
}
template <int CIN, int TAPS, int COUT, int XF, bool STATS, bool SUMS>
constexpr int conv_ks_off() {
  return ((conv_red_off<CIN, TAPS, COUT, XF, SUMS>() + ((STATS || SUMS) ? kConvRedFloats : 0)) + 3) & ~3;
}
template <int CIN, int TAPS, int COUT>
constexpr int conv_ks_floats() { return (kWaves - 1) * Geo<CIN, TAPS, COUT>::kMT * 64 * 4; }
template <int CIN, int TAPS, int COUT, int W>
__device__ __forceinline__ void conv_ks_part(const float* lds_in, const float* lds_w, int lane, f32x4 (&xacc)[Geo<CIN, TAPS, COUT>::kMT]) {
  using G = Geo<CIN, TAPS, COUT>;
  constexpr int MT = G::kMT, PH = G::kPH, NB = G::kKP / 8;
  constexpr int s0 = NB * W / kWaves, s1 = NB * (W + 1) / kWaves;
  constexpr int Kpart = W == kWaves - 1 ? G::kKP - 8 * s0 : 8 * (s1 - s0);   // the last share runs to the end of K, tail included
  const int n = lane & 15, kq = lane >> 4;
  const int pxx = 16 * (G::kRegular * kWaves) + n;                             // the odd tile's columns
  const float* in = lds_in + G::kG * G::kCinP;
  f32x4 a[1][MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
    a[0][mt] = W == 0 ? *reinterpret_cast<const f32x4*>(lds_w + G::kData + 16 * mt + 4 * kq) : f32x4{0.f, 0.f, 0.f, 0.f};
  chain::gemm_pass<1, 0, MT, Kpart, 0, 1>(in, (PH * pxx - G::kG) * G::kCinP + 2 * kq + 8 * s0, 0, lds_w + s0 * MT * 128, lane, a);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) xacc[mt] = a[0][mt];
}
template <int CIN, int TAPS, int COUT>
__device__ __forceinline__ void conv_ks_partial(const float* lds_in, const float* lds_w, float* ks_lds, int wave, int lane,
                                                f32x4 (&xacc)[Geo<CIN, TAPS, COUT>::kMT]) {
  constexpr int MT = Geo<CIN, TAPS, COUT>::kMT;
  if (wave == 0) conv_ks_part<CIN, TAPS, COUT, 0>(lds_in, lds_w, lane, xacc);
  else if (wave == 1) conv_ks_part<CIN, TAPS, COUT, 1>(lds_in, lds_w, lane, xacc);
  else if (wave == 2) conv_ks_part<CIN, TAPS, COUT, 2>(lds_in, lds_w, lane, xacc);
  else conv_ks_part<CIN, TAPS, COUT, 3>(lds_in, lds_w, lane, xacc);
  if (wave != 0) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) *reinterpret_cast<f32x4*>(ks_lds + (((wave - 1) * MT + mt) * 64 + lane) * 4) = xacc[mt];
  }
}
template <int CIN, int TAPS, int COUT, bool ACCUM, bool STATS, int XF, bool SUMS = false>
__global__ __launch_bounds__(kThreads, RCED_TM_OCC) void conv1xk_mfma(const float* __restrict__ in, const float* __restrict__ packet,
                                                          float* __restrict__ out, int frames, double* __restrict__ part,
                                                          XformArgs xa, BnBwdArgs ba, SumArgs sa, const float* acc_from) {
  using G = Geo<CIN, TAPS, COUT>;
  static_assert(!SUMS || CIN % 2 == 0, "SUMS lives in the wide-staging tile loop");
  const long acc_delta = ACCUM && acc_from ? (long)(acc_from - out) : 0;   // ACCUM: out = acc_from + conv (null: out += conv)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* lin = lds;
  float* lw = lds + G::kInFloats;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int e = tid; e < G::kLdsFloats; e += kThreads) lds[e] = e < G::kInFloats ? 0.f : packet[e - G::kInFloats];
  float* xt = lds + G::kLdsFloats;                      // [2 or 3][CIN], only with XF
  if constexpr (XF == kXfBnRelu) xform_table_fill<CIN>(xt, xa.mu, xa.rstd, xa.gamma, xa.beta, tid);
  if constexpr (XF == kXfBnBwd) bnbwd_table_fill<CIN>(xt, ba, tid);
  float* zt = lds + conv_sums_off<CIN, TAPS, COUT, XF>();   // [kTF][129][COUT], only with SUMS
  float* stab = zt + (SUMS ? Stage<SUMS ? COUT : 2>::kElems : 0);
  if constexpr (SUMS) xform_table_fill<COUT>(stab, sa.mu, sa.rstd, sa.gamma, sa.beta, tid);
  // [wave][32 channels][2] running sums (STATS / SUMS), doubles, behind everything else
  double* red = reinterpret_cast<double*>(lds + conv_red_off<CIN, TAPS, COUT, XF, SUMS>());
  if constexpr (STATS || SUMS)
    for (int e = tid; e < kWaves * 64; e += kThreads) red[e] = 0.0;
  double* red_wave = red + wave * 64;
  __syncthreads();
  const int ntiles = (frames + kTF - 1) / kTF;
  if constexpr (CIN % 2 == 0) {
    f32x4 pre[Stage<CIN>::kPer], pre2[XF == kXfBnBwd ? Stage<CIN>::kPer : 1];
    if ((int)blockIdx.x < ntiles) {
      tile_fetch<CIN>(in, blockIdx.x * kTF, frames, tid, pre);
      if constexpr (XF == kXfBnBwd) tile_fetch<CIN>(ba.z, blockIdx.x * kTF, frames, tid, pre2);
    }
#if RCED_TM_STAMPS
    const bool stamps_on = SUMS && CIN == 30 && blockIdx.x == 0;
    unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = stamps_on ? tm_stamp() : 0;
    if (stamps_on && lane == 0) for (int i = 0; i < 4; ++i) g_tm[wave][i] = g_tm2[wave][i] = 0;
#endif
    SumState<G::kMTm> sums;       // carried over kSumFlush tiles (conv_tile)
    sums.zero();
    int tile_no = 0;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      const int frame0 = tile * kTF;
      const bool flush = (++tile_no % kSumFlush) == 0 || tile + (int)gridDim.x >= ntiles;
      constexpr bool carried = (STATS || SUMS) && kSumFlush > 1;
      auto where = [](int fr, int r) { return (G::kG + fr * G::kS) * CIN + r; };
#if RCED_TM_STAMPS
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      TM_ST(0);   // wait for the prefetched tile
#endif
      if (!(RCED_TM_EXP & 8) || tile == (int)blockIdx.x) {
      if constexpr (XF == kXfBnRelu) tile_commit_bnrelu<CIN>(lin, tid, pre, where, xt, frame0, frames);
      else if constexpr (XF == kXfBnBwd) tile_commit_bnbwd<CIN>(lin, tid, pre, pre2, where, xt, frame0, frames, ba.beta != nullptr);
      else tile_commit<CIN>(lin, tid, pre, where);
      }
      TM_ST(1);   // commit
      __syncthreads();
      TM_ST(2);   // barrier 1
      if constexpr (SUMS)
        if (!(RCED_TM_EXP & 1) || tile == (int)blockIdx.x) ztile_fetch<COUT>(sa.z, zt, frame0, frames, tid);   // the previous tile's epilogue is behind a barrier
      // The next tile's loads: piece by piece between this tile's MFMAs (RCED_TM_SPREAD; not where the epilogue waits on
      // vmcnt(0) for its z tile (SUMS), which would wait for them too), or all at once here -- a tile cut short by the
      // end of the batch always takes the latter, with tile_fetch's per-element bounds.
      constexpr bool kSpread = RCED_TM_SPREAD && !SUMS;
      const int nframe0 = (tile + (int)gridDim.x) * kTF;
      const bool more = tile + (int)gridDim.x < ntiles && !(RCED_TM_EXP & 1);
      const bool spread = kSpread && more && frames - nframe0 >= kTF;
      if (more && !spread) {
        tile_fetch<CIN>(in, nframe0, frames, tid, pre);
        if constexpr (XF == kXfBnBwd) tile_fetch<CIN>(ba.z, nframe0, frames, tid, pre2);
      }
      pin();
      TM_ST(3);   // fetch issue
      constexpr int kP1 = Stage<CIN>::kPer, kNP = (XF == kXfBnBwd ? 2 : 1) * kP1, kNS = G::kKP / 8;
      auto each = [&](int s) {
        if constexpr (kSpread) {
          if (spread)
            tm_static_for<0, kNP>([&](auto ic) {
              constexpr int i = decltype(ic)::value;
              if (s == i * kNS / kNP) {
                if constexpr (i < kP1) tile_fetch_piece<CIN, kThreads, i>(in, nframe0, tid, pre);
                else if constexpr (XF == kXfBnBwd) tile_fetch_piece<CIN, kThreads, i - kP1>(ba.z, nframe0, tid, pre2);
              }
            });
        }
      };
      constexpr bool kOpq = (XF == kXfBnBwd ? 2 : 1) * Stage<CIN>::kPer * 4 >= RCED_TM_OPQ_MIN;   // VGPRs holding the next tile
      constexpr bool kKs = conv_ks_on<CIN, TAPS, COUT>() && !SUMS;
      if constexpr (kKs) {
        float* ks = lds + conv_ks_off<CIN, TAPS, COUT, XF, STATS, SUMS>();
        f32x4 xacc[G::kMT];
        conv_ks_partial<CIN, TAPS, COUT>(lin, lw, ks, wave, lane, xacc);
        if (wave == 0) conv_tile<CIN, TAPS, COUT, ACCUM, STATS, 1, SUMS, kOpq, false, decltype(each), false, 2, true>(lin, lw, out, frame0, frames, wave, lane, red_wave, zt, stab, each, nullptr, xacc, ks, sums, carried, flush, acc_delta);
        else conv_tile<CIN, TAPS, COUT, ACCUM, STATS, 0, SUMS, kOpq, false, decltype(each), false, 2, true>(lin, lw, out, frame0, frames, wave, lane, red_wave, zt, stab, each, nullptr, xacc, ks, sums, carried, flush, acc_delta);
      } else if constexpr (RCED_TM_MSPLIT && G::kExtra == 1 && G::kMTm == 2 && !SUMS && !ACCUM) {
        // The odd column tile by M-tile: two waves carry half of it each (4.5 / 4.5 / 4 / 4 tiles of MFMAs instead of
        // 5 / 4 / 4 / 4), and the second half of the grid -- with a resident grid of two workgroups per CU the likely partner
        // on the same CU -- gives its halves to waves 2, 3: 8.5 on every SIMD.
        const int xw = (int)blockIdx.x >= ((int)gridDim.x + 1) / 2 ? 2 : 0;
        if (wave == xw) conv_tile<CIN, TAPS, COUT, ACCUM, STATS, 1, SUMS, kOpq, false, decltype(each), false, 2, false, 0>(lin, lw, out, frame0, frames, wave, lane, red_wave, zt, stab, each, nullptr, nullptr, nullptr, sums, carried, flush, acc_delta);
        else if (wave == xw + 1) conv_tile<CIN, TAPS, COUT, ACCUM, STATS, 1, SUMS, kOpq, false, decltype(each), false, 2, false, 1>(lin, lw, out, frame0, frames, wave, lane, red_wave, zt, stab, each, nullptr, nullptr, nullptr, sums, carried, flush, acc_delta);
        else conv_tile<CIN, TAPS, COUT, ACCUM, STATS, 0, SUMS, kOpq, false>(lin, lw, out, frame0, frames, wave, lane, red_wave, zt, stab, each, nullptr, nullptr, nullptr, sums, carried, flush, acc_delta);
      } else {
      if (wave < G::kExtra) conv_tile<CIN, TAPS, COUT, ACCUM, STATS, 1, SUMS, kOpq, false>(lin, lw, out, frame0, frames, wave, lane, red_wave, zt, stab, each, nullptr, nullptr, nullptr, sums, carried, flush, acc_delta);
      else conv_tile<CIN, TAPS, COUT, ACCUM, STATS, 0, SUMS, kOpq, false>(lin, lw, out, frame0, frames, wave, lane, red_wave, zt, stab, each, nullptr, nullptr, nullptr, sums, carried, flush, acc_delta);
      }
      TM_ST(4);   // conv_tile
      __syncthreads();
      TM_ST(5);   // barrier 3
    }
#if RCED_TM_STAMPS
    if (stamps_on && lane == 0)
      printf("TMSE wave %d: coords %llu zsum %llu stores %llu\n", wave, g_tm2[wave][0], g_tm2[wave][1], g_tm2[wave][2]);
    if (stamps_on && lane == 0)
      printf("TMST wave %d: loadwait %llu commit %llu bar1 %llu fetch %llu tile %llu bar3 %llu | gemm %llu vmwait %llu bar2 %llu epi %llu\n", wave, ts[0], ts[1],
             ts[2], ts[3], ts[4], ts[5], g_tm[wave][0], g_tm[wave][3], g_tm[wave][1], g_tm[wave][2]);
#endif
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
      if (wave < G::kExtra) conv_tile<CIN, TAPS, COUT, ACCUM, STATS, 1>(lin, lw, out, frame0, frames, wave, lane, red_wave);
      else conv_tile<CIN, TAPS, COUT, ACCUM, STATS, 0>(lin, lw, out, frame0, frames, wave, lane, red_wave);
      __syncthreads();
    }
  }
  if constexpr (STATS || SUMS) {
    __syncthreads();   // every wave's record is complete (conv_tile adds to it tile by tile)
    if (tid < 2 * COUT) {
      const int c = tid >> 1, k = tid & 1;
      double t = 0.0;
      for (int w = 0; w < kWaves; ++w) t += red[(w * 32 + c) * 2 + k];
      part[((size_t)blockIdx.x * COUT + c) * 2 + k] = t;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Forward convolution in the three-part bf16 form (GeoX6): conv1xk_mfma's tile loop with the x6 commit and GEMM pass.
// in [frames][129][CIN] (XF: the producer's z), packet from pack_packet_x6, out = z [frames][129][COUT];
// STATS: per-workgroup (sum z, sum z^2) records in part, as conv1xk_mfma.
// ---------------------------------------------------------------------------------------------
template <int CIN, int TAPS, int COUT, int XF>
constexpr int conv_x6_red_off() {
  return (GeoX6<CIN, TAPS, COUT>::kLdsFloats + (XF == kXfBnRelu ? 2 * CIN : 0) + 3) & ~3;
}
template <int CIN, int TAPS, int COUT, bool STATS, int XF>
__global__ __launch_bounds__(kThreads, RCED_TM_OCC) void conv_x6_fwd(const float* __restrict__ in, const float* __restrict__ packet,
                                                         float* __restrict__ out, int frames, double* __restrict__ part,
                                                         XformArgs xa) {
  using G = Geo<CIN, TAPS, COUT>;
  using GX = GeoX6<CIN, TAPS, COUT>;
  static_assert(CIN % 2 == 0 && (XF == kXfNone || XF == kXfBnRelu), "wide staging; plain or BatchNorm + ReLU input");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  unsigned short* planes = reinterpret_cast<unsigned short*>(lds);
  float* lw = lds + GX::kInFloats;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  static_assert(GX::kFits, "two workgroups per CU");
  for (int e = tid; e < GX::kLdsFloats; e += kThreads) lds[e] = e < GX::kInFloats ? 0.f : packet[e - GX::kInFloats];
  float* xt = lds + GX::kLdsFloats;                      // [2][CIN], only with XF
  if constexpr (XF == kXfBnRelu) xform_table_fill<CIN>(xt, xa.mu, xa.rstd, xa.gamma, xa.beta, tid);
  double* red = reinterpret_cast<double*>(lds + conv_x6_red_off<CIN, TAPS, COUT, XF>());
  if constexpr (STATS)
    for (int e = tid; e < kWaves * 64; e += kThreads) red[e] = 0.0;
  double* red_wave = red + wave * 64;
  __syncthreads();
  const int ntiles = (frames + kTF - 1) / kTF;
  f32x4 pre[Stage<CIN>::kPer];
  if ((int)blockIdx.x < ntiles) tile_fetch<CIN>(in, blockIdx.x * kTF, frames, tid, pre);
  SumState<G::kMTm> sums;
  sums.zero();
  int tile_no = 0;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int frame0 = tile * kTF;
    const bool flush = (++tile_no % kSumFlush) == 0 || tile + (int)gridDim.x >= ntiles;
    constexpr bool carried = STATS && kSumFlush > 1;
    tile_commit_x6<CIN, GX::kCS, GX::kPlane, XF == kXfBnRelu, G::kG, G::kS - kF>(planes, tid, pre, xt, frame0, frames);
    __syncthreads();
    const int nframe0 = (tile + (int)gridDim.x) * kTF;
    if (tile + (int)gridDim.x < ntiles) tile_fetch<CIN>(in, nframe0, frames, tid, pre);
    pin();
    constexpr bool kOpq = Stage<CIN>::kPer * 4 >= RCED_TM_OPQ_MIN;
    if (wave < G::kExtra)
      conv_tile<CIN, TAPS, COUT, false, STATS, 1, false, kOpq, false, chain::NoEach, false, 2, false, -1, true>(
          lds, lw, out, frame0, frames, wave, lane, red_wave, nullptr, nullptr, chain::NoEach(), nullptr, nullptr, nullptr, sums, carried, flush);
    else
      conv_tile<CIN, TAPS, COUT, false, STATS, 0, false, kOpq, false, chain::NoEach, false, 2, false, -1, true>(
          lds, lw, out, frame0, frames, wave, lane, red_wave, nullptr, nullptr, chain::NoEach(), nullptr, nullptr, nullptr, sums, carried, flush);
    __syncthreads();
  }
  if constexpr (STATS) {
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
// XF: x is the producer's z (rebuilt to relu(bn(z)));  DZF: dz is d_u (rebuilt to the BatchNorm-backward dz).
// PH = 2 (COUT = 8 only): 8 output channels would fill half of the 16 MFMA columns, so a column is (pixel parity, co)
// and the pixel axis (MFMA K) walks pixel PAIRS q:  D[k'][(ph, co)] = sum_q x[2q + k'] * dz[2q + ph][co] with
// k' = tap + ph in 0..TAPS, and dW[tap][ci][co] collects D[(tap + ph, ci)][(ph, co)] of both parities: one more tap
// row (+1/TAPS MFMAs) for half the pixel steps.
template <int CIN, int TAPS, int COUT, bool XF, bool DZF, int PH = 1>
__global__ __launch_bounds__(kThreads, RCED_TM_OCC) void wgrad1xk_mfma(const float* __restrict__ x, const float* __restrict__ dz,
                                                           float* __restrict__ dW, float* __restrict__ dbias, int frames,
                                                           XformArgs xa, BnBwdArgs ba, unsigned pstride) {
  using G = Geo<CIN, TAPS, COUT>;
  static_assert(CIN % 2 == 0 && COUT % 2 == 0, "wgrad1xk_mfma stages float4 / float2 pieces");
  static_assert(PH == 1 || (PH == 2 && COUT == 8 && G::kNPX % 2 == 0), "two pixel parities x 8 channels = 16 columns");
  // one spare k row carries a constant 1, so its output row is sum_px dz = dbias
  constexpr int kRowsK = (TAPS + PH - 1) * G::kCinP;   // window rows (PH = 1: the layer's K)
  // NREM (18 output channels, tm_rem): the second N-tile would carry 2 channels in 16 columns.  Channels 16, 17 get a pass
  // of their own whose 16 columns are (pixel phase ph < 8, channel 16 + c) and whose K axis walks GROUPS of 8 pixels:
  //   D[k'][(ph, c)] = sum_q xwin[8 q][k'] * dz[8 q + ph][16 + c],   k' = (tap + ph) * cin + ci  in [0, (TAPS + 7) * cin),
  // (TAPS + 7) cin / 16 M-tiles per 32 pixels instead of the second N-tile's KT per 4 pixels: 330 + 72 MFMAs per two-frame
  // tile instead of 660 for the 8 -> 18 layers.  dW[tap][ci][16 + c] = sum_ph D[(tap + ph, ci)][(ph, c)] is gathered through
  // LDS at the end of the kernel; d bias of the two channels is the lanes' running sum of the B values they fed.
  constexpr bool NREM = PH == 1 && tm_rem(COUT) == 2;
  constexpr int KT = (kRowsK + 1 + 15) / 16, NTo = PH == 2 || NREM ? 1 : G::kMT;
  constexpr int kOneTile = kRowsK / 16, kOneRow = kRowsK % 16;
  constexpr int kRemRows = (TAPS + 7) * G::kCinP, KT8 = NREM ? kRemRows / 16 : 1;
  constexpr int kSteps8 = (G::kNPX + 31) / 32;                     // K steps of the remainder pass: 4 groups of 8 pixels each
  static_assert(!NREM || kRemRows % 16 == 0, "whole M-tiles");
  constexpr int kDzRows = NREM && 32 * kSteps8 + 8 > 16 * G::kTiles + 4 ? 32 * kSteps8 + 8 : 16 * G::kTiles + 4;
  static_assert(!NREM || (kDzRows * COUT <= (16 * G::kTiles + 4) * 32 && kWaves * kRemRows * 16 <= G::kInFloats + 64 + (16 * G::kTiles + 4) * 32),
                "the launcher's LDS (32 floats per dz row) holds the longer dz tile and the final gather's scratch");
  constexpr int kDzStride = COUT;                      // floats per pixel row of the dz tile = the tensor's own row: staging needs no
                                                       // per-piece division (columns co >= COUT of the second N-tile read the next
                                                       // pixel's first channels and are never written out); the launcher allocates 32
  constexpr int kGroups = PH == 2 ? (G::kNPX / 2 + 3) / 4 : G::kNPX / 4 + 1;   // 4 pixels (pixel pairs) per MFMA
  // the last group's padded rows read past the staged window: into the 64-float slack, which stays zero
  static_assert((4 * PH * (kGroups - 1) + PH * 3) * G::kCinP + 16 * (KT - 1) + 15 < G::kInFloats + 64, "A reads stay inside lin + slack");
  static_assert(4 * PH * (kGroups - 1) + PH * 3 + (PH - 1) < kDzRows, "B reads stay inside the dz tile");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* lin = lds;                                    // [kInRows + tail][CinP]
  float* ldz = lds + G::kInFloats + 64;                // [kDzRows][32]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kq = lane >> 4;
  for (int e = tid; e < G::kInFloats + 64 + kDzRows * kDzStride; e += kThreads) lds[e] = 0.f;
  float* xt = lds + G::kInFloats + 64 + kDzRows * kDzStride;   // [2][CIN] with XF, then [3][COUT] with DZF
  float* dt = xt + 2 * CIN;
  if constexpr (XF) xform_table_fill<CIN>(xt, xa.mu, xa.rstd, xa.gamma, xa.beta, tid);
  if constexpr (DZF) bnbwd_table_fill<COUT>(dt, ba, tid);
  f32x4 acc[KT][NTo];
#pragma unroll
  for (int a = 0; a < KT; ++a)
#pragma unroll
    for (int b = 0; b < NTo; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 racc[KT8];               // NREM: the remainder pass's accumulators, rows 16 kt + 4 kq + r, column i = (ph, c)
#pragma unroll
  for (int a = 0; a < KT8; ++a) racc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  const int ntiles = (frames + kTF - 1) / kTF;
  f32x4 prex[Stage<CIN>::kPer], prez[Stage<COUT>::kPer], prez2[DZF ? Stage<COUT>::kPer : 1];
  if ((int)blockIdx.x < ntiles) {
    tile_fetch<CIN>(x, blockIdx.x * kTF, frames, tid, prex);
    tile_fetch<COUT>(dz, blockIdx.x * kTF, frames, tid, prez);
    if constexpr (DZF) tile_fetch<COUT>(ba.z, blockIdx.x * kTF, frames, tid, prez2);
  }
  __syncthreads();
  // window start of pixel p is row p of lin; lane kq owns pixel px0 + kq (PH = 1) or the pair starting at px0 + 2 kq
  const float* ain = lin + PH * kq * G::kCinP + i;
  const float* bin = PH == 2 ? ldz + (2 * kq + (i >> 3)) * kDzStride + (i & 7) : ldz + kq * kDzStride + i;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    auto where = [](int fr, int r) { return (G::kG + fr * G::kS) * CIN + r; };
    if constexpr (XF) tile_commit_bnrelu<CIN>(lin, tid, prex, where, xt, tile * kTF, frames);
    else tile_commit<CIN>(lin, tid, prex, where);
    auto where_dz = [](int fr, int r) { return fr * G::kS * kDzStride + r; };
    if constexpr (DZF) tile_commit_bnbwd<COUT>(ldz, tid, prez, prez2, where_dz, dt, tile * kTF, frames, ba.beta != nullptr);
    else tile_commit<COUT>(ldz, tid, prez, where_dz);
    __syncthreads();
    const int nframe0 = (tile + (int)gridDim.x) * kTF;
    const bool more = tile + (int)gridDim.x < ntiles;
    const bool spread = RCED_TM_SPREAD && more && frames - nframe0 >= kTF;
    if (more && !spread) {
      tile_fetch<CIN>(x, nframe0, frames, tid, prex);
      tile_fetch<COUT>(dz, nframe0, frames, tid, prez);
      if constexpr (DZF) tile_fetch<COUT>(ba.z, nframe0, frames, tid, prez2);
    }
    pin();
    constexpr int kPX = Stage<CIN>::kPer, kPZ = Stage<COUT>::kPer, kNP = kPX + (DZF ? 2 : 1) * kPZ;
    constexpr int kIters = kGroups / kWaves;     // iterations every wave runs: the pieces are spread over them
    int it = 0;
    for (int g = wave; g < kGroups; g += kWaves, ++it) {
      if (spread)
        tm_static_for<0, kNP>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          if (it == i * kIters / kNP) {
            if constexpr (i < kPX) tile_fetch_piece<CIN, kThreads, i>(x, nframe0, tid, prex);
            else if constexpr (i < kPX + kPZ) tile_fetch_piece<COUT, kThreads, i - kPX>(dz, nframe0, tid, prez);
            else if constexpr (DZF) tile_fetch_piece<COUT, kThreads, i - kPX - kPZ>(ba.z, nframe0, tid, prez2);
          }
        });
      const int px0 = 4 * PH * g;
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
    if constexpr (NREM) {
      // lane (i, kq): A = row 16 kt + i of the window that starts at pixel 8 (q0 + kq); B = dz[8 (q0 + kq) + i / 2][16 + i % 2]
      // (rows past the tile's pixels are zero: never written)
      for (int s8 = kWaves - 1 - wave; s8 < kSteps8; s8 += kWaves) {   // from the other end: the waves with fewer main groups first
        const int p8 = 8 * (4 * s8 + kq);
        float a8[KT8];
#pragma unroll
        for (int kt = 0; kt < KT8; ++kt) a8[kt] = lin[p8 * G::kCinP + 16 * kt + i];
        const float b8 = ldz[(p8 + (i >> 1)) * kDzStride + 16 + (i & 1)];
        bsum += b8;
#pragma unroll
        for (int kt = 0; kt < KT8; ++kt) racc[kt] = mfma(a8[kt], b8, racc[kt]);
      }
    }
    __syncthreads();
  }
  // D row = k = 16*kt + 4*kq + r, column = co = 16*nt + i   (PH = 2: column = (parity i >> 3, co = i & 7), tap = k' - parity)
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int nt = 0; nt < NTo; ++nt) {
      const int co = PH == 2 ? (i & 7) : 16 * nt + i;
      const int ph = PH == 2 ? (i >> 3) : 0;
      const int slice = ((int)blockIdx.x * kWaves + wave) * PH + ph;   // (the two parities hold the same dW entries)
      const float vv[4] = {acc[kt][nt].x, acc[kt][nt].y, acc[kt][nt].z, acc[kt][nt].w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = 16 * kt + 4 * kq + r;
        const int tapk = k / G::kCinP, ci = k - tapk * G::kCinP, tap = tapk - ph;
        if (k < kRowsK && ci < CIN && co < COUT && tap >= 0 && tap < TAPS) wg_put(dW, pstride, slice, (tap * CIN + ci) * COUT + co, vv[r]);
        if (k == kRowsK && co < COUT && dbias) wg_put(dbias, pstride, slice, co, vv[r]);
      }
    }
  if constexpr (NREM) {
    // gather: this wave's D[k'][(ph, c)] through LDS (the tiles are dead: the loop ended on a barrier), then
    // dW[tap][ci][16 + c] = sum over ph of D[(tap + ph) cin + ci][(ph, c)], added in phase order
    float* sc = lds + wave * (kRemRows * 16);
#pragma unroll
    for (int kt = 0; kt < KT8; ++kt) {
      const float vv[4] = {racc[kt].x, racc[kt].y, racc[kt].z, racc[kt].w};
#pragma unroll
      for (int r = 0; r < 4; ++r) sc[(16 * kt + 4 * kq + r) * 16 + i] = vv[r];
    }
    __syncthreads();
    const int slice = (int)blockIdx.x * kWaves + wave;
    for (int e = lane; e < TAPS * CIN * 2; e += 64) {
      const int c = e & 1, ci = (e >> 1) % CIN, tap = (e >> 1) / CIN;
      float t = 0.f;
#pragma unroll
      for (int ph = 0; ph < 8; ++ph) t += sc[((tap + ph) * G::kCinP + ci) * 16 + 2 * ph + c];
      wg_put(dW, pstride, slice, (tap * CIN + ci) * COUT + 16 + c, t);
    }
    if (dbias) {      // lanes (i, kq) with the same c = i & 1 hold shares of sum dz[.][16 + c]
      float v = bsum;
      v += __shfl_xor(v, 2, 64);
      v += __shfl_xor(v, 4, 64);
      v += __shfl_xor(v, 8, 64);
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (lane < 2) wg_put(dbias, pstride, slice, 16 + lane, v);
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Backward of one 1xk layer in ONE kernel: wgrad and dgrad share the staged tiles.
// Both read the same (d_u or g, z) pair of the layer to rebuild dz (tile_commit_bnbwd), and once each of them ran at the
// 3.5-3.9 TB/s the memory system gives these kernels, reading that pair twice was the largest avoidable traffic of the step
// (4 GB per 30-channel layer).  One workgroup of EIGHT waves per CU: all 512 threads prefetch and commit the next tile
// (x with its halo, dz with its halo); then waves 0..3 run the dgrad pass over the dz tile (conv_tile, as conv1xk_mfma) while
// waves 4..7 run the wgrad MFMAs over (x, dz) (as wgrad1xk_mfma; its B operand comes straight out of the dgrad's dz tile,
// pixel stride COUT) -- the two halves of every SIMD's wave pair do different work, so one's epilogue / operand waits sit
// beside the other's MFMAs.  Two barriers per tile instead of five.  SUMS: the dgrad half also forms the producer's
// BatchNorm-backward sums, from the transformed x tile that is in LDS anyway (conv_tile SUMX): no z tile is fetched.
// CIN / COUT are the LAYER's (x has CIN channels, dz COUT); the packet is the dgrad packet (pack_packet transpose = 1).
// ---------------------------------------------------------------------------------------------
constexpr int kBwdThreads = 512;
#ifndef RCED_TM_BWD_X6_C
#define RCED_TM_BWD_X6_C 1   // bwd_fused_mfma<30,9,8>: its dgrad half (8 -> 30, no remainder pass) in the three-part bf16 form (GeoX6)
#endif
// fused backward shapes whose dgrad half runs on the bf16 pipe: the host packs their dgrad packet with pack_packet_x6
#ifndef RCED_TM_BWD_X6_B
#define RCED_TM_BWD_X6_B 0   // bwd_fused_mfma<18,5,30>: its dgrad half (30 -> 18: main pass + remainder pass) likewise.  Measured
                             // 3.2 ms against 2.35 ms: with ONE M-tile a B fragment (48 bytes per lane for the three parts) feeds only six
                             // MFMAs (96 cycles) -- four dgrad waves saturate the LDS port (24 cycles per fragment each); the
                             // two-M-tile shapes (8 -> 30 dgrad, 18 -> 30 forward) reuse every fragment twice and gain.  Off.
#endif
__host__ __device__ constexpr bool bwd_x6(int cin, int taps, int cout) {
  return (RCED_TM_BWD_X6_C && cin == 30 && taps == 9 && cout == 8) || (RCED_TM_BWD_X6_B && cin == 18 && taps == 5 && cout == 30);
}
template <int CIN, int TAPS, int COUT, bool X6D>
struct BwdX6Sizes {      // floats the dgrad's packet region and its planes take
  static constexpr int kPacket = Geo<COUT, TAPS, CIN>::kPacket, kPlanes = 0;
};
template <int CIN, int TAPS, int COUT>
struct BwdX6Sizes<CIN, TAPS, COUT, true> {
  using GX = GeoX6<COUT, TAPS, CIN>;
  static constexpr int kPacket = GX::kPacket > Geo<COUT, TAPS, CIN>::kPacket ? GX::kPacket : Geo<COUT, TAPS, CIN>::kPacket;
  static constexpr int kPlanes = GX::kInFloats;
};
template <int CIN, int TAPS, int COUT>
struct BwdGeo {
  using GD = Geo<COUT, TAPS, CIN>;     // the dgrad convolution: dz (COUT channels) -> dx (CIN channels)
  using GW = Geo<CIN, TAPS, COUT>;     // the wgrad's view: x tile with CIN channels
  static constexpr int r4(int v) { return (v + 3) & ~3; }
  static constexpr int kDzOff = 0;                                   // [GD::kInRows][COUT] (+ slack for the last group's reads)
  static constexpr int kPkOff = r4(GD::kInFloats + 64);
  static constexpr bool kX6D = bwd_x6(CIN, TAPS, COUT);
  using XS = BwdX6Sizes<CIN, TAPS, COUT, kX6D>;
  static constexpr int kXOff = r4(kPkOff + XS::kPacket);             // [GW::kInRows][CIN] (+ slack)
  static constexpr int kTabOff = r4(kXOff + GW::kInFloats + 64);     // [2][CIN] BatchNorm + ReLU of x, then [4][COUT] BatchNorm backward
  static constexpr int kRedOff = r4(kTabOff + 2 * CIN + 4 * COUT);   // [4 waves][32][2] doubles
  // raw copies of the NEXT tile (LDS-DMA targets, RCED_TM_BWD_DMA): x [2][129][CIN], d_u / g [2][129][COUT], z [2][129][COUT]
  static constexpr int kStgX = r4(kRedOff + kConvRedFloats);
  static constexpr int kStgD = kStgX + r4(kTF * kF * CIN);
  static constexpr int kStgZ = kStgD + r4(kTF * kF * COUT);
  static constexpr int kPlOff = r4(RCED_TM_BWD_DMA ? kStgZ + r4(kTF * kF * COUT) : kRedOff + kConvRedFloats);   // dz as three bf16 planes (kX6D)
  static constexpr int kLdsFloats = kPlOff + XS::kPlanes;
  static_assert(kLdsFloats * 4 <= 160 * 1024, "LDS budget");
  static_assert(GD::kG == GW::kG && GD::kS == GW::kS, "one pixel space");
};

// Split of the wgrad pixel groups over the four wgrad waves (see the tile loop).  d[w] = MFMAs of dgrad wave w per tile,
// g = MFMAs per wgrad group; wave w gets round((T - d[w]) / g) groups, T = the per-SIMD mean, the last wave the rest.
template <int CIN, int TAPS, int COUT>
struct BwdBalance {
  using GD = Geo<COUT, TAPS, CIN>;
  using GW = Geo<CIN, TAPS, COUT>;
  static constexpr int PH = COUT == 8 ? 2 : 1;
  static constexpr int kRowsK = (TAPS + PH - 1) * GW::kCinP;
  static constexpr int KT = (kRowsK + 1 + 15) / 16, NTo = PH == 2 ? 1 : GW::kMT;
  static constexpr int kGroups = PH == 2 ? (GW::kNPX / 2 + 3) / 4 : GW::kNPX / 4 + 1;
  static constexpr int kPerTile = (GD::kKP / 8 * 2 + (GD::kKP % 8 + 3) / 4) * GD::kMTm;   // MFMAs of one dgrad column tile (main pass)
  static constexpr int kPerRem = GD::kR ? GD::kKR / 8 * 2 + (GD::kKR % 8 + 3) / 4 : 0;      // ... of one remainder tile
  static constexpr int kPerGroup = KT * NTo;
  static constexpr int rem_tiles(int w) {                  // remainder tiles 3 - w, 7 - w, ... < kNRT (conv_tile)
    int c = 0;
    for (int rt = 3 - w; rt < GD::kNRT; rt += 4) ++c;
    return c;
  }
  static constexpr int dgrad(int w) { return (GD::kRegular + (w < GD::kExtra ? 1 : 0)) * kPerTile + rem_tiles(w) * kPerRem; }
  static constexpr int kTotal = dgrad(0) + dgrad(1) + dgrad(2) + dgrad(3) + kGroups * kPerGroup;
  static constexpr int count(int w) {       // groups of wgrad wave w (0..3)
    int used = 0;
    for (int v = 0; v < 4; ++v) {
      int n = (kTotal / 4 - dgrad(v) + kPerGroup / 2) / kPerGroup;
      if (n < 1) n = 1;
      if (v == 3 || used + n > kGroups - (3 - v)) n = v == 3 ? kGroups - used : kGroups - (3 - v) - used;
      if (v == w) return n;
      used += n;
    }
    return 0;
  }
  __host__ __device__ static constexpr int begin_c(int w) {
    int b = 0;
    for (int v = 0; v < w; ++v) b += count(v);
    return b;
  }
  __device__ __forceinline__ static int begin(int w) {      // w wave-uniform, 0..4
    return w == 0 ? 0 : w == 1 ? begin_c(1) : w == 2 ? begin_c(2) : w == 3 ? begin_c(3) : kGroups;
  }
  static constexpr int kMinIters = count(0) < count(1) ? (count(0) < count(2) ? (count(0) < count(3) ? count(0) : count(3))
                                                                               : (count(2) < count(3) ? count(2) : count(3)))
                                                        : (count(1) < count(2) ? (count(1) < count(3) ? count(1) : count(3))
                                                                               : (count(2) < count(3) ? count(2) : count(3)));
  static_assert(begin_c(4) == kGroups && count(0) > 0 && count(1) > 0 && count(2) > 0 && count(3) > 0, "every group has one owner");
};

#ifndef RCED_TM_OWN_B
#define RCED_TM_OWN_B 1     // bwd_fused_mfma<18,5,30>: tensors staged by the dgrad half alone (bit 0: x, bit 1: (d_u, z))
#endif
#ifndef RCED_TM_OWN_C
#define RCED_TM_OWN_C 0     // bwd_fused_mfma<30,9,8>
#endif
#ifndef RCED_TM_OWN_OTHER
#define RCED_TM_OWN_OTHER 0 // the R-CED shapes
#endif
template <int CIN, int TAPS, int COUT>
constexpr int bwd_own() {
  return (CIN == 18 && TAPS == 5 && COUT == 30) ? RCED_TM_OWN_B : (CIN == 30 && TAPS == 9 && COUT == 8) ? RCED_TM_OWN_C : RCED_TM_OWN_OTHER;
}
// Measured alternatives: tile i+1 committed into a second pair of LDS buffers by the wgrad half alone while
// tile i is computed (one barrier per tile), with the staging behind or in front of that half's MFMAs, with s_setprio on
// the staging half: 3.17 ms against 3.09 ms for this form -- the two waves of a SIMD do not overlap one's VALU / staging
// work with the other's MFMA chain to any useful degree; what counts is the instruction total per SIMD.
template <int CIN, int TAPS, int COUT, bool XF, bool SUMS>
__global__ __launch_bounds__(kBwdThreads) void bwd_fused_mfma(const float* __restrict__ x, const float* __restrict__ du,
                                                               const float* __restrict__ packet, float* __restrict__ dx,
                                                               float* __restrict__ dW, float* __restrict__ dbias, int frames,
                                                               double* __restrict__ part, XformArgs xa, BnBwdArgs ba, unsigned pstride) {
  using B = BwdGeo<CIN, TAPS, COUT>;
  using GD = typename B::GD;
  using GW = typename B::GW;
  static_assert(CIN % 2 == 0 && COUT % 2 == 0, "wide staging");
  constexpr int NTH = kBwdThreads;
  // X6D: the dgrad half in the three-part bf16 form -- dz is committed a second time as bf16 planes (tile_commit_bnbwd's
  // `extra`), the packet arrives from pack_packet_x6, conv_tile runs gemm_pass_x6; the wgrad half reads the fp32 dz tile as before
  constexpr bool X6D = B::kX6D;
  constexpr int kPkFloats = X6D ? BwdX6Sizes<CIN, TAPS, COUT, X6D>::kPacket : GD::kPacket;
  constexpr int PH = COUT == 8 ? 2 : 1;                       // wgrad column packing (see wgrad1xk_mfma)
  constexpr int kRowsK = (TAPS + PH - 1) * GW::kCinP;
  constexpr int KT = (kRowsK + 1 + 15) / 16, NTo = PH == 2 ? 1 : GW::kMT;
  constexpr int kOneTile = kRowsK / 16, kOneRow = kRowsK % 16;
  constexpr int kGroups = PH == 2 ? (GW::kNPX / 2 + 3) / 4 : GW::kNPX / 4 + 1;
  static_assert((4 * PH * (kGroups - 1) + PH * 3) * GW::kCinP + 16 * (KT - 1) + 15 < GW::kInFloats + 64, "A reads stay inside the x tile + slack");
  static_assert((GD::kG + 4 * PH * (kGroups - 1) + PH * 3 + (PH - 1)) * COUT + 31 < GD::kInFloats + 64, "B reads stay inside the dz tile + slack");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* ldz = lds + B::kDzOff;
  float* lw = lds + B::kPkOff;
  float* lx = lds + B::kXOff;
  float* xt = lds + B::kTabOff;
  float* dt = xt + 2 * CIN;
  double* red = reinterpret_cast<double*>(lds + B::kRedOff);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int role = wave8 >> 2, wave = wave8 & 3;              // role 0: dgrad, 1: wgrad
  const int i = lane & 15, kq = lane >> 4;
  for (int e = tid; e < B::kLdsFloats; e += NTH)
    lds[e] = (e >= B::kPkOff && e < B::kPkOff + kPkFloats) ? packet[e - B::kPkOff] : 0.f;
  __syncthreads();
  if constexpr (XF) xform_table_fill<CIN>(xt, xa.mu, xa.rstd, xa.gamma, xa.beta, tid);
  bnbwd_table_fill<COUT>(dt, ba, tid);
  // ONE register array for both roles: the wgrad half's [KT][NTo] accumulators live for the whole kernel, the dgrad
  // half's [column tiles][M-tiles] per tile -- a wave has one role, but as two arrays both are live in every wave as far
  // as the register allocator can tell (the kernel sat at 256 VGPRs + scratch)
  constexpr int kDgAcc = (GD::kRegular + 1) * GD::kMTm, kAccN = KT * NTo > kDgAcc ? KT * NTo : kDgAcc;
  f32x4 accs[kAccN];
  f32x4 (&acc)[KT][NTo] = *reinterpret_cast<f32x4 (*)[KT][NTo]>(&accs[0]);
#pragma unroll
  for (int a = 0; a < kAccN; ++a) accs[a] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int ntiles = (frames + kTF - 1) / kTF;
  constexpr bool DMA = RCED_TM_BWD_DMA != 0;
  // OWN: which of the tile's tensors are fetched and committed by the DGRAD half alone (bit 0: x, bit 1: (d_u, z)) instead
  // of by all 512 threads.  The wgrad half is the tile's critical path (stamps: commit, then ~7.5 k cycles blocked in the
  // issue of its share of the next tile's loads, then its MFMAs), the dgrad half waits 3-6 k cycles at the tile's end:
  // bytes moved from one half's burst to the other's shorten the first by what the second has to spare.
  constexpr int OWN = bwd_own<CIN, TAPS, COUT>();
  static_assert(OWN == 0 || (!DMA && !RCED_TM_SPREAD && RCED_TM_BWD_STAGGER), "ownership is built for the staggered VGPR staging");
  constexpr int XN = (OWN & 1) ? NTH / 2 : NTH, ZN = (OWN & 2) ? NTH / 2 : NTH;
  constexpr int kPX = Stage<CIN, XN>::kPer, kPZ = Stage<COUT, ZN>::kPer, kNP = kPX + 2 * kPZ;
  float* sx = lds + B::kStgX;
  float* sd = lds + B::kStgD;
  float* sz = lds + B::kStgZ;
  f32x4 prex[kPX], pred[kPZ], prez[kPZ];     // DMA: transient (stage_load -> commit); else the next tile for a whole tile time
  // the next tile into the staging copies: by LDS-DMA piece i (whole tiles), or all of a ragged last tile with bounds checks
  auto dma_piece = [&](auto ic, int f0, int vt) {     // vt: the thread of the piece map whose share is issued
    constexpr int i = decltype(ic)::value;
    if constexpr (i < kPX) tile_dma_piece<CIN, XN, i>(x, f0, vt, sx);
    else if constexpr (i < kPX + kPZ) tile_dma_piece<COUT, ZN, i - kPX>(du, f0, vt, sd);
    else tile_dma_piece<COUT, ZN, i - kPX - kPZ>(ba.z, f0, vt, sz);
  };
  auto stage_ragged = [&](int f0) {
    tile_fetch<CIN, XN>(x, f0, frames, tid, prex);
    tile_fetch<COUT, ZN>(du, f0, frames, tid, pred);
    tile_fetch<COUT, ZN>(ba.z, f0, frames, tid, prez);
    stage_store<CIN, XN>(sx, tid, prex);
    stage_store<COUT, ZN>(sd, tid, pred);
    stage_store<COUT, ZN>(sz, tid, prez);
  };
  // this thread's pieces of the tile at frame f0 (VGPR staging): the tensors its half owns, or shares
  auto fetch_mine = [&](int f0) {
    if (!(OWN & 1) || role == 0) tile_fetch<CIN, XN>(x, f0, frames, tid, prex);
    if (!(OWN & 2) || role == 0) {
      tile_fetch<COUT, ZN>(du, f0, frames, tid, pred);
      tile_fetch<COUT, ZN>(ba.z, f0, frames, tid, prez);
    }
  };
  __syncthreads();                            // the LDS image is zeroed / the packet is in place before anything lands in it
  if ((int)blockIdx.x < ntiles) {
    const int f0 = blockIdx.x * kTF;
    if constexpr (DMA) {
      if (frames - f0 >= kTF) tm_static_for<0, kNP>([&](auto ic) { dma_piece(ic, f0, tid); });
      else stage_ragged(f0);
    } else {
      fetch_mine(f0);
    }
  }
  __syncthreads();
  const float* ain = lx + PH * kq * GW::kCinP + i;
  const float* ldzp = ldz + GD::kG * COUT;
  const float* bin = PH == 2 ? ldzp + (2 * kq + (i >> 3)) * COUT + (i & 7) : ldzp + kq * COUT + i;
#if RCED_TM_STAMPS
  const bool stamps_on = SUMS && blockIdx.x == 0;
  unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = stamps_on ? tm_stamp() : 0;
  if (stamps_on && lane == 0 && role == 0) for (int q = 0; q < 4; ++q) g_tm[wave][q] = g_tm2[wave][q] = 0;
#endif
  SumState<GD::kMTm> sums;        // the dgrad half's masked sums, carried over kSumFlush tiles (conv_tile)
  sums.zero();
  int tile_no = 0;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int frame0 = tile * kTF;
    const bool flush = (++tile_no % kSumFlush) == 0 || tile + (int)gridDim.x >= ntiles;
    constexpr bool carried = SUMS && kSumFlush > 1 && RCED_TM_BWD_CARRY;
    auto where_x = [](int fr, int r) { return (GW::kG + fr * GW::kS) * CIN + r; };
    auto where_dz = [](int fr, int r) { return (GD::kG + fr * GD::kS) * COUT + r; };
    if constexpr (DMA) {
      // this wave's transfers of the tile have landed ... and everybody's (this is also the barrier behind the previous
      // tile's MFMA reads of the operand tiles the commit below overwrites)
      // (the dgrad half issued none -- except in the prologue -- and must not wait here: vmcnt(0) would also wait for
      // its epilogue's global stores, which are free to stay in flight across the barrier)
      if (role == 1 || tile == (int)blockIdx.x) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      TM_ST(0);   // wait for the staged tile
      __syncthreads();
      stage_load<CIN, XN>(sx, tid, prex);
    } else {
#if RCED_TM_STAMPS
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      TM_ST(0);   // wait for the prefetched tile
#endif
    }
    if (!(OWN & 1) || role == 0) {
      if constexpr (XF) tile_commit_bnrelu<CIN, XN>(lx, tid, prex, where_x, xt, frame0, frames);
      else tile_commit<CIN, XN>(lx, tid, prex, where_x);
    }
    if constexpr (DMA) {      // the x pieces are dead before the (d_u, z) pieces are read: 12-16 fewer registers at the peak
      pin();
      stage_load<COUT, ZN>(sd, tid, pred);
      stage_load<COUT, ZN>(sz, tid, prez);
    }
    if (!(OWN & 2) || role == 0) {
      if constexpr (X6D) {
        using GX = GeoX6<COUT, TAPS, CIN>;
        unsigned short* planes = reinterpret_cast<unsigned short*>(lds + B::kPlOff);
        auto to_planes = [&](int fr, int r, f32x2 v) {
          const int px = r / COUT, c = r - px * COUT;
          s16x2 ph, pm, pl;
          split3_pair(v, ph, pm, pl);
          unsigned short* d = planes + (GD::kG + fr * GD::kS + px) * GX::kCS + c;
          *reinterpret_cast<s16x2*>(d) = ph;
          *reinterpret_cast<s16x2*>(d + GX::kPlane) = pm;
          *reinterpret_cast<s16x2*>(d + 2 * GX::kPlane) = pl;
        };
        tile_commit_bnbwd<COUT, ZN>(ldz, tid, pred, prez, where_dz, dt, frame0, frames, ba.beta != nullptr, to_planes);
      } else {
        tile_commit_bnbwd<COUT, ZN>(ldz, tid, pred, prez, where_dz, dt, frame0, frames, ba.beta != nullptr);
      }
    }
    TM_ST(1);   // commit
    __syncthreads();
    TM_ST(2);   // barrier 1
    const int nframe0 = (tile + (int)gridDim.x) * kTF;
    const bool more = tile + (int)gridDim.x < ntiles;
    const bool spread = RCED_TM_SPREAD && !DMA && more && frames - nframe0 >= kTF;
    if (more && !spread) {
      if constexpr (DMA) {
        // Issued by the WGRAD half alone, each of its threads for two threads of the 512-thread piece map: the burst
        // finds the memory pipeline's queues full and blocks the issuing waves for 3-4 k cycles (stamps) -- now only the
        // wgrad wave of every SIMD, while its dgrad wave runs the MFMA pass; the wgrad wave's own MFMAs then run beside
        // the dgrad wave's epilogue.  (All eight waves issuing their own pieces stalled the whole CU: 14 % of the kernel.)
        if (frames - nframe0 >= kTF) {
          if (role == 1)
            tm_static_for<0, kNP>([&](auto ic) {
              dma_piece(ic, nframe0, tid - 256);
              dma_piece(ic, nframe0, tid);
            });
        } else {
          stage_ragged(nframe0);
        }
      } else if (!RCED_TM_BWD_STAGGER || role == 1) {
        fetch_mine(nframe0);
      }
    }
    pin();
    TM_ST(3);   // fetch issue
    auto piece = [&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if constexpr (DMA) dma_piece(ic, nframe0, tid);
      else if constexpr (i < kPX) tile_fetch_piece<CIN, XN, i>(x, nframe0, tid, prex);
      else if constexpr (i < kPX + kPZ) tile_fetch_piece<COUT, ZN, i - kPX>(du, nframe0, tid, pred);
      else tile_fetch_piece<COUT, ZN, i - kPX - kPZ>(ba.z, nframe0, tid, prez);
    };
    if (role == 0) {
      constexpr int kNS = GD::kKP / 8;
      auto each = [&](int s) {
        if (spread)
          tm_static_for<0, kNP>([&](auto ic) { if (s == decltype(ic)::value * kNS / kNP) piece(ic); });
        if (RCED_TM_BWD_STAGGER && !DMA && s == -1 && more && !spread) {
          fetch_mine(nframe0);
          pin();
        }
      };
      const float* dgin = X6D ? lds + B::kPlOff : ldz;      // the dgrad's input tile: bf16 planes, or the fp32 dz tile
      if (wave < GD::kExtra) conv_tile<COUT, TAPS, CIN, false, false, 1, SUMS, false, true, decltype(each), true, RCED_TM_BWD_DEPTH, false, -1, X6D>(dgin, lw, dx, frame0, frames, wave, lane, red + wave * 64, lx, nullptr, each, accs, nullptr, nullptr, sums, carried, flush);
      else conv_tile<COUT, TAPS, CIN, false, false, 0, SUMS, false, true, decltype(each), true, RCED_TM_BWD_DEPTH, false, -1, X6D>(dgin, lw, dx, frame0, frames, wave, lane, red + wave * 64, lx, nullptr, each, accs, nullptr, nullptr, sums, carried, flush);
    } else {
      // Groups of this wgrad wave: a contiguous range sized so that every SIMD (dgrad wave w + wgrad wave w) issues the
      // same number of MFMAs per tile -- the dgrad wave that carries the odd column tile gets fewer groups beside it
      // (the barrier that ends a tile waits for the fullest SIMD: 573 / 498 / 498 / 498 MFMAs before, for the 18 -> 30 layer).
      const int gbeg = BwdBalance<CIN, TAPS, COUT>::begin(wave), gend = BwdBalance<CIN, TAPS, COUT>::begin(wave + 1);
      constexpr int kIters = BwdBalance<CIN, TAPS, COUT>::kMinIters;
      int it = 0;
      int g = gbeg;
      if constexpr (bwd_wg_x6(CIN, TAPS, COUT) && PH == 1) {
        // Whole runs of eight groups (32 pixels) in the three-part bf16 form: the operands run along PIXELS here, so a lane's
        // fragment is eight strided scalars (as many LDS reads as the fp32 form issues for these pixels), split in registers
        // (split8); an A fragment feeds the NTo N-tiles, a B fragment the KT M-tiles: KT * NTo * 6 MFMAs of 16 cycles per 32
        // pixels where the fp32 form issues KT * NTo * 8 of 32.  The groups left over (< 8) take the fp32 loop below.
        const float* ain8 = lx + 8 * kq * GW::kCinP + i;
        const float* bin8 = ldzp + 8 * kq * COUT + i;
        for (; g + 8 <= gend; g += 8, it += 8) {
          const int pxc = 4 * g;
          X6Parts bp[NTo];
#pragma unroll
          for (int nt = 0; nt < NTo; ++nt) {
            f32x2 q[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) q[e] = f32x2{bin8[(pxc + 2 * e) * COUT + 16 * nt], bin8[(pxc + 2 * e + 1) * COUT + 16 * nt]};
            bp[nt] = x6_split8(q);
          }
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) {
            f32x2 q[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) q[e] = f32x2{ain8[(pxc + 2 * e) * GW::kCinP + 16 * kt], ain8[(pxc + 2 * e + 1) * GW::kCinP + 16 * kt]};
            if (kt == kOneTile && i == kOneRow) {
#pragma unroll
              for (int e = 0; e < 4; ++e) q[e] = f32x2{1.f, 1.f};
            }
            const X6Parts ap = x6_split8(q);
#pragma unroll
            for (int nt = 0; nt < NTo; ++nt) {
              f32x4 v = acc[kt][nt];
              v = mfma32(ap.m, bp[nt].m, v);
              v = mfma32(ap.l, bp[nt].h, v);
              v = mfma32(ap.h, bp[nt].l, v);
              v = mfma32(ap.m, bp[nt].h, v);
              v = mfma32(ap.h, bp[nt].m, v);
              v = mfma32(ap.h, bp[nt].h, v);
              acc[kt][nt] = v;
            }
          }
        }
      }
      for (; g < gend; ++g, ++it) {
        if (spread)
          tm_static_for<0, kNP>([&](auto ic) { if (it == decltype(ic)::value * kIters / kNP) piece(ic); });
        const int px0 = 4 * PH * g;
        float a[KT], b[NTo];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) a[kt] = ain[px0 * GW::kCinP + 16 * kt];
        if (i == kOneRow) a[kOneTile] = 1.f;
#pragma unroll
        for (int nt = 0; nt < NTo; ++nt) b[nt] = bin[px0 * COUT + 16 * nt];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int nt = 0; nt < NTo; ++nt) acc[kt][nt] = mfma(a[kt], b[nt], acc[kt][nt]);
      }
    }
    TM_ST(4);   // role work
    if constexpr (!DMA) __syncthreads();   // (DMA: the barrier at the top of the next tile, behind its vmcnt wait)
    TM_ST(5);   // barrier 2
  }
  if constexpr (DMA) __syncthreads();      // the dgrad half's LDS records are complete before they are summed below
#if RCED_TM_STAMPS
  if (stamps_on && lane == 0)
    printf("BWST<%d,%d,%d> wave8 %d: loadwait %llu commit %llu bar1 %llu fetch %llu work %llu bar2 %llu | gemm %llu epi %llu (coords %llu sums %llu stores %llu)\n",
           CIN, TAPS, COUT, wave8, ts[0], ts[1], ts[2], ts[3], ts[4], ts[5], role == 0 ? g_tm[wave][0] : 0ull,
           role == 0 ? g_tm[wave][2] : 0ull, role == 0 ? g_tm2[wave][0] : 0ull, role == 0 ? g_tm2[wave][1] : 0ull,
           role == 0 ? g_tm2[wave][2] : 0ull);
#endif
  if (role == 1) {
    // D row = k = 16*kt + 4*kq + r, column = co = 16*nt + i   (PH = 2: column = (parity i >> 3, co = i & 7), tap = k' - parity)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int nt = 0; nt < NTo; ++nt) {
        const int co = PH == 2 ? (i & 7) : 16 * nt + i;
        const int ph = PH == 2 ? (i >> 3) : 0;
        const int slice = ((int)blockIdx.x * 4 + (wave & 3)) * PH + ph;   // the four wgrad waves of the workgroup
        const float vv[4] = {acc[kt][nt].x, acc[kt][nt].y, acc[kt][nt].z, acc[kt][nt].w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = 16 * kt + 4 * kq + r;
          const int tapk = k / GW::kCinP, ci = k - tapk * GW::kCinP, tap = tapk - ph;
          if (k < kRowsK && ci < CIN && co < COUT && tap >= 0 && tap < TAPS) wg_put(dW, pstride, slice, (tap * CIN + ci) * COUT + co, vv[r]);
          if (k == kRowsK && co < COUT && dbias) wg_put(dbias, pstride, slice, co, vv[r]);
        }
      }
  }
  if constexpr (SUMS) {
    if (tid < 2 * CIN) {   // the dgrad half's records (complete: the tile loop ends on a barrier); channels of dx = CIN
      const int c = tid >> 1, k = tid & 1;
      double t = 0.0;
      for (int w = 0; w < kWaves; ++w) t += red[(w * 32 + c) * 2 + k];
      part[((size_t)blockIdx.x * CIN + c) * 2 + k] = t;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The 1x129, CH -> 1 output layer (decode_5 / decode_8 / decode_final; model.py:24,56,89).
// Forward: dense Toeplitz GEMM as in the inference kernels,
//   y[frame, f] = b + sum_{f', ci} h[frame, f', ci] * W[f' - f + 64, ci],  D[f (9 M-tiles), frame (N)], k = f'*CH + ci;
// the Toeplitz-expanded A fragments are rebuilt on the device every step (the weights move).
// ---------------------------------------------------------------------------------------------
template <int CH>
struct FinGeo {
  static constexpr int kK = kF * CH;
  static constexpr int kNB64 = kK / 8, kTail = kK - 8 * kNB64;    // tail: 0 (CH 8), 4 (CH 12), 2 (CH 10)
  static_assert(kTail <= 4 && CH % 2 == 0 && CH <= 14, "one b32 tail step at most; CH even, one spare N column");
  static constexpr int kMT = 9;
  static constexpr int kMain = kNB64 * kMT * 128;
  static constexpr int kPack = kMain + (kTail ? kMT * 64 : 0);
};
constexpr int kFinFrames = 64, kFinThreads = 192;

// w [129][CH] (TF [1,129,CH,1]) -> pack [s][mt][lane][e] (+ tail [mt][lane]); row f = 16*mt + i
static __global__ void pack_final_fwd(const float* __restrict__ w, int CH, float* __restrict__ pack) {
  const int K = kF * CH, NB = K / 8, main = NB * 9 * 128, total = main + ((K % 8) ? 9 * 64 : 0);
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  int k, f;
  if (e < main) {
    const int s = e / (9 * 128), r = e - s * 9 * 128, mt = r / 128, q = r - mt * 128, lane = q >> 1;
    k = 8 * s + 2 * (lane >> 4) + (q & 1);
    f = 16 * mt + (lane & 15);
  } else {
    const int r = e - main, mt = r / 64, lane = r - mt * 64;
    k = 8 * NB + (lane >> 4);
    f = 16 * mt + (lane & 15);
  }
  const int fp = k / CH, ci = k - fp * CH, tap = fp - f + 64;
  pack[e] = (k < K && f < kF && tap >= 0 && tap < kF) ? w[tap * CH + ci] : 0.f;
}

// (The GEMM itself is chain::final_gemm_lds_kernel<CH>, the inference path's kernel with its B operand staged through LDS:
// train_api.hip fin_forward.  Round 2's tmm::final_fwd read the h rows with strided global loads: 0.54 vs 0.41 ms.)

// Output-layer wgrad: dW[tap, ci] = sum_{frame, q} x[frame, q, ci] * dz[frame, q + 64 - tap]  (q = f + tap - 64).
// MFMA roles: M = tap (9 tiles), N = ci (one tile; column CH carries ones, so D[64][CH] = sum dz = dbias),
// K = q (4 bins per MFMA, 33 steps per frame).  A[i][kq] = dzpad[q0 + kq + 64 - (16 mt + i)] from a zero-padded
// per-wave LDS copy of the frame's dz row; B[kq][j] = x[frame, q0 + kq, j] straight from global (the 4 bins of a
// step are 4*CH contiguous floats).  A wave owns frames; the workgroup's waves add up through LDS, then atomics.
constexpr int kFwPad = 96, kFwRow = 304;   // dzpad index = 96 + bin; reads span [-79, 195] around it ([-94, 195] in the tap-pair form)
// CH = 8 (CR-CED), tap-pair form: eight input channels would fill half of the 16 columns.  A column is (shift s in {0, 1},
// ci) with B = x[q - s][ci] (the K axis q runs to 131, so q - 1 covers every bin), and the rows are the EVEN taps only:
// D[r][(s, ci)] = sum_q x[q - s][ci] dz[q + 64 - 2 r] = dW[2 r - s][ci] -- 5 M-tiles (65 rows) instead of 9 per K step,
// 165 instead of 297 MFMAs per frame.  The ones column
// that carried d bias is gone: the lanes add up the dz values they stage instead.
template <int CH>
__global__ __launch_bounds__(kThreads) void final_wgrad(const float* __restrict__ x, const float* __restrict__ dz,
                                                         float* __restrict__ dW, float* __restrict__ dbias, int frames,
                                                         unsigned pstride) {
  constexpr bool PAIR = CH == 8;
  constexpr int MTW = PAIR ? 5 : 9, kRedTaps = PAIR ? 160 : 144;
  __shared__ float rows[kWaves][kFwRow];
  __shared__ float red[kWaves][kRedTaps * 16];   // one copy per wave, added in wave order (no LDS atomics: deterministic)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kq = lane >> 4;
  float* row = rows[wave];
  for (int e = lane; e < kFwRow; e += 64) row[e] = 0.f;
  __syncthreads();
  f32x4 acc[MTW];
#pragma unroll
  for (int m = 0; m < MTW; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dsum = 0.f;                      // PAIR: this lane's share of sum dz = d bias
  constexpr int kSteps = (kF + 3) / 4;   // 33
  const float* ap = row + kFwPad + 64 + kq - (PAIR ? 2 * i : i);
  const int gw = blockIdx.x * kWaves + wave, nw = gridDim.x * kWaves;
  for (int fr = gw; fr < frames; fr += nw) {
    const float* xr = x + (size_t)fr * kF * CH;
    const float* dr = dz + (size_t)fr * kF;
    float b[kSteps];
#pragma unroll
    for (int s = 0; s < kSteps; ++s) {
      const int q = 4 * s + kq;
      if constexpr (PAIR) b[s] = (q - (i >> 3) >= 0 && q - (i >> 3) < kF) ? xr[(q - (i >> 3)) * CH + (i & 7)] : 0.f;
      else b[s] = (i < CH && q < kF) ? xr[q * CH + i] : (i == CH && q < kF ? 1.f : 0.f);
    }
    // the previous frame's MFMA reads of `row` are complete (same wave, in order): overwrite it
    const float d0 = dr[lane], d1 = dr[64 + lane], d2 = lane == 0 ? dr[128] : 0.f;
    row[kFwPad + lane] = d0;
    row[kFwPad + 64 + lane] = d1;
    if (lane == 0) row[kFwPad + 128] = d2;
    if constexpr (PAIR) dsum += (d0 + d1) + d2;
#pragma unroll
    for (int s = 0; s < kSteps; ++s) {
      float a[MTW];
#pragma unroll
      for (int m = 0; m < MTW; ++m) a[m] = ap[4 * s - (PAIR ? 32 : 16) * m];
#pragma unroll
      for (int m = 0; m < MTW; ++m) acc[m] = mfma(a[m], b[s], acc[m]);
    }
  }
  // D row = tap = 16*m + 4*kq + r, column = i (ci, or the ones column); PAIR: row = tap / 2, column = (tap % 2, ci)
#pragma unroll
  for (int m = 0; m < MTW; ++m) {
    const float vv[4] = {acc[m].x, acc[m].y, acc[m].z, acc[m].w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if constexpr (PAIR) {
        const int tap = 2 * (16 * m + 4 * kq + r) - (i >> 3);
        if (tap >= 0) red[wave][tap * 16 + (i & 7)] = vv[r];
      }
      else red[wave][(16 * m + 4 * kq + r) * 16 + i] = vv[r];
    }
  }
  if constexpr (PAIR) {   // d bias: whole-wave sum of the lanes' shares, parked in a column no tap uses
    float v = row_sum16(dsum);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (lane == 63) red[wave][15] = v;
  }
  __syncthreads();
  for (int e = tid; e < kRedTaps * 16; e += kThreads) {
    const int tap = e >> 4, c = e & 15;
    const float v = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
    if (tap < kF && c < CH) wg_put(dW, pstride, blockIdx.x, tap * CH + c, v);
    if (PAIR ? (e == 15 && dbias != nullptr) : (tap == 64 && c == CH && dbias != nullptr)) wg_put(dbias, pstride, blockIdx.x, 0, v);
  }
}


// ---------------------------------------------------------------------------------------------
// First layer (8 x KW kernel on the 1-channel input; model.py:10,36,81) weight gradient:
//   dW[i, j, co] = sum_{frame, f} x[t + i - 3, f + j - PL] * dz[frame, f, co]
// MFMA roles as wgrad1xk_mfma: M = k = i*KW + j (one spare row carries ones -> dbias), N = co, K = pixels.
// A tile is kTF consecutive frames; for each of them the 8 input rows t-3 .. t+4 (zero outside the utterance,
// SAME padding 3/4) are staged with their column halo, so A[k][pixel] = rows[(fl*8 + i)*RS + f + j]; dz as in
// wgrad1xk_mfma (optionally rebuilt from (d_u, z): DZF).
// ---------------------------------------------------------------------------------------------
// dz rows per frame of first_wgrad's LDS tile (shared with its launcher), and whether the 18-channel N-remainder form runs
constexpr bool first_wgrad_nrem(int kw, int cout) { return kw == 9 && tm_rem(cout) == 2; }
constexpr int first_wgrad_fs(int kw, int cout) { return first_wgrad_nrem(kw, cout) ? 160 : 132; }
template <int KW, int COUT, bool DZF>
__global__ __launch_bounds__(kThreads, 2) void first_wgrad(const float* __restrict__ x, const float* __restrict__ dz,
                                                         float* __restrict__ dW, float* __restrict__ dbias, int frames,
                                                         int T, BnBwdArgs ba, unsigned pstride) {
  constexpr int KH = 8, PT = 3, PL = (KW - 1) / 2, RS = kF + KW - 1, K1 = KH * KW;
  // NREM (18 channels; see wgrad1xk_mfma): channels 16, 17 as columns (bin phase ph < 8, c) over groups of 8 bins; the rows
  // are k' = ih * 16 + (j + ph) -- for KW = 9 exactly 16 shifted taps per time tap, 8 M-tiles -- and
  // dW[ih][j][16 + c] = sum_ph D[ih * 16 + j + ph][(ph, c)]
  constexpr bool NREM = first_wgrad_nrem(KW, COUT);
  constexpr int KT = (K1 + 1 + 15) / 16, NTo = NREM ? 1 : (COUT + 15) / 16;
  constexpr int kOneTile = K1 / 16, kOneRow = K1 % 16;
  constexpr int kFS = first_wgrad_fs(KW, COUT);         // dz rows per frame: 33 groups of 4 bins (NREM: 5 steps of 32 bins, zero past 129)
  constexpr int KT8 = NREM ? KH : 1, kSteps8 = NREM ? 5 : 0;
  constexpr int kDzStride = 32, kDzRows = kTF * kFS + 4;
  constexpr int kRowsFloats = ((kTF * KH * RS + 32 + 3) / 4) * 4;
  constexpr int kXElems = kTF * KH * kF, kPerX = (kXElems + kThreads - 1) / kThreads;
  static_assert(COUT % 2 == 0, "dz is staged in float2 pieces");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* rows = lds;                                    // [kTF*8][RS] (+ slack for the k >= K1 / f >= 129 reads)
  float* ldz = lds + kRowsFloats;                       // [kDzRows][32]
  float* dt = ldz + kDzRows * kDzStride;                // [3][COUT] with DZF
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kq = lane >> 4;
  for (int e = tid; e < kRowsFloats + kDzRows * kDzStride; e += kThreads) lds[e] = 0.f;
  if constexpr (DZF) bnbwd_table_fill<COUT>(dt, ba, tid);
  int offk[KT];                                         // this lane's window offset in every M-tile
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const int k = 16 * kt + i;
    offk[kt] = k < K1 ? (k / KW) * RS + (k % KW) : 0;
  }
  f32x4 acc[KT][NTo];
#pragma unroll
  for (int a = 0; a < KT; ++a)
#pragma unroll
    for (int b = 0; b < NTo; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 racc[KT8];
#pragma unroll
  for (int a = 0; a < KT8; ++a) racc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  const int ntiles = (frames + kTF - 1) / kTF;
  float prex[kPerX];
  f32x4 prez[Stage<COUT>::kPer], prez2[DZF ? Stage<COUT>::kPer : 1];
  // (utterance, time) of the tile's two frames from ONE wave-uniform division per tile: per element and tile the
  // runtime division frame / T was ~40 VALU instructions, 9 elements per thread -- three non-MFMA VALU per MFMA in the
  // first-layer kernels (rocprofv3, round 3); what depends on the element alone (fl, ih, f) is loop-invariant
  auto fetch_rows = [&](int tile) {
    static_assert(kTF == 2, "two frames per tile: the second is the first's successor");
    const int frame0 = tile * kTF;
    const int un0 = frame0 / T, ut0 = frame0 - un0 * T;
    const bool wrap = ut0 + 1 == T;
    const int un1 = wrap ? un0 + 1 : un0, ut1 = wrap ? 0 : ut0 + 1;
#pragma unroll
    for (int u = 0; u < kPerX; ++u) {
      const int e = tid + u * kThreads;
      const int fl = e / (KH * kF), r = e - fl * (KH * kF), ih = r / kF, f = r - ih * kF;
      const int un = fl ? un1 : un0, tt = (fl ? ut1 : ut0) + ih - PT;
      prex[u] = (e < kXElems && frame0 + fl < frames && tt >= 0 && tt < T) ? x[((size_t)un * T + tt) * kF + f] : 0.f;
    }
  };
  if ((int)blockIdx.x < ntiles) {
    fetch_rows(blockIdx.x);
    tile_fetch<COUT>(dz, blockIdx.x * kTF, frames, tid, prez);
    if constexpr (DZF) tile_fetch<COUT>(ba.z, blockIdx.x * kTF, frames, tid, prez2);
  }
  __syncthreads();
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
#pragma unroll
    for (int u = 0; u < kPerX; ++u) {
      const int e = tid + u * kThreads;
      if (e < kXElems) {
        const int row = e / kF, f = e - row * kF;        // row = fl*8 + ih
        rows[row * RS + PL + f] = prex[u];
      }
    }
    auto where_dz = [](int fr, int r) {
      const int f = r / COUT, co = r - f * COUT;
      return (fr * kFS + f) * kDzStride + co;
    };
    if constexpr (DZF) tile_commit_bnbwd<COUT>(ldz, tid, prez, prez2, where_dz, dt, tile * kTF, frames, ba.beta != nullptr);
    else tile_commit<COUT>(ldz, tid, prez, where_dz);
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) {
      fetch_rows(tile + gridDim.x);
      tile_fetch<COUT>(dz, (tile + gridDim.x) * kTF, frames, tid, prez);
      if constexpr (DZF) tile_fetch<COUT>(ba.z, (tile + gridDim.x) * kTF, frames, tid, prez2);
    }
    pin();
    for (int g = wave; g < kTF * 33; g += kWaves) {
      const int fl = g / 33, f0 = 4 * (g - fl * 33);
      const float* ap = rows + fl * KH * RS + f0 + kq;
      const float* bp = ldz + (fl * kFS + f0 + kq) * kDzStride + i;
      float a[KT], b[NTo];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) a[kt] = ap[offk[kt]];
      if (i == kOneRow) a[kOneTile] = 1.f;
#pragma unroll
      for (int nt = 0; nt < NTo; ++nt) b[nt] = bp[16 * nt];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int nt = 0; nt < NTo; ++nt) acc[kt][nt] = mfma(a[kt], b[nt], acc[kt][nt]);
    }
    if constexpr (NREM) {
      for (int g8 = kWaves - 1 - wave; g8 < kTF * kSteps8; g8 += kWaves) {
        const int fl = g8 / kSteps8, b8 = 8 * (4 * (g8 - fl * kSteps8) + kq);      // this lane's group starts at bin b8
        float a8[KT8];
#pragma unroll
        for (int kt = 0; kt < KT8; ++kt) a8[kt] = rows[(fl * KH + kt) * RS + b8 + i];
        const float v8 = ldz[(fl * kFS + b8 + (i >> 1)) * kDzStride + 16 + (i & 1)];
        bsum += v8;
#pragma unroll
        for (int kt = 0; kt < KT8; ++kt) racc[kt] = mfma(a8[kt], v8, racc[kt]);
      }
    }
    __syncthreads();
  }
  // D row = k = 16*kt + 4*kq + r (TF layout [8][KW][1][COUT] = k*COUT + co), column = co = 16*nt + i
#pragma unroll
  for (int kt = 0; kt < KT; ++kt)
#pragma unroll
    for (int nt = 0; nt < NTo; ++nt) {
      const int co = 16 * nt + i;
      const float vv[4] = {acc[kt][nt].x, acc[kt][nt].y, acc[kt][nt].z, acc[kt][nt].w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = 16 * kt + 4 * kq + r;
        if (k < K1 && co < COUT) wg_put(dW, pstride, (int)blockIdx.x * kWaves + wave, k * COUT + co, vv[r]);
        if (k == K1 && co < COUT && dbias) wg_put(dbias, pstride, (int)blockIdx.x * kWaves + wave, co, vv[r]);
      }
    }
  if constexpr (NREM) {
    float* sc = lds + wave * (KH * 16 * 16);      // the tiles are dead: the loop ended on a barrier
#pragma unroll
    for (int kt = 0; kt < KT8; ++kt) {
      const float vv[4] = {racc[kt].x, racc[kt].y, racc[kt].z, racc[kt].w};
#pragma unroll
      for (int r = 0; r < 4; ++r) sc[(16 * kt + 4 * kq + r) * 16 + i] = vv[r];
    }
    __syncthreads();
    const int slice = (int)blockIdx.x * kWaves + wave;
    for (int e = lane; e < K1 * 2; e += 64) {
      const int c = e & 1, k = e >> 1, ih = k / KW, j = k - ih * KW;
      float t = 0.f;
#pragma unroll
      for (int ph = 0; ph < 8; ++ph) t += sc[(ih * 16 + j + ph) * 16 + 2 * ph + c];
      wg_put(dW, pstride, slice, k * COUT + 16 + c, t);
    }
    if (dbias) {
      float v = bsum;
      v += __shfl_xor(v, 2, 64);
      v += __shfl_xor(v, 4, 64);
      v += __shfl_xor(v, 8, 64);
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (lane < 2) wg_put(dbias, pstride, slice, 16 + lane, v);
    }
  }
}


// ---------------------------------------------------------------------------------------------
// First layer forward: z[frame, f, co] = bias[co] + sum_{i<8, j<KW} W[i, j, co] * x[t + i - 3, f + j - PL].
// Same staging as first_wgrad (8 input rows per frame with their column halo).  MFMA roles: M = co, N = 16 bins of
// one frame, K = 8*KW in b32 steps s = ih*KW + j with lane kq <-> time tap 4*ih + kq (so a step's four k values
// are four rows at the same column).  Packet [step][mt][lane] = W[4*ih + kq][j][16*mt + i], then 32 shifts.
// STATS as conv1xk_mfma: per-workgroup (sum z, sum z^2) records for the BatchNorm statistics.
// ---------------------------------------------------------------------------------------------
static __global__ void pack_first(const float* __restrict__ w, const float* __restrict__ bias, int kw, int cout,
                           float* __restrict__ packet) {
  // 18 channels (tm_rem): the main section holds ONE M-tile (channels 0..15); a second section holds the remainder pass's
  // A fragments -- rows (bin phase ph < 8, channel 16 + c), K = 8 time rows x 16 window columns in b32 steps
  // s' = ih*16 + j' with lane kq <-> time tap 4 ih + kq:  A[(ph, c)][(i, j')] = W[i][j' - ph][16 + c]  (0 outside the kw taps)
  const int rem = tm_rem(cout);
  const int MT = rem ? 1 : (cout + 15) / 16, steps = 2 * kw, data = steps * MT * 64, rdata = rem ? 32 * 64 : 0;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= data + rdata + 32) return;
  if (e >= data + rdata) {
    const int c = e - data - rdata;
    packet[e] = c < cout ? bias[c] : 0.f;
    return;
  }
  if (e >= data) {
    const int q = e - data, sr = q / 64, lane = q - sr * 64;
    const int ih = sr / 16, jw = sr - ih * 16, i = 4 * ih + (lane >> 4), ph = (lane & 15) >> 1, c = lane & 1, j = jw - ph;
    packet[e] = (j >= 0 && j < kw) ? w[(i * kw + j) * cout + 16 + c] : 0.f;
    return;
  }
  const int s = e / (MT * 64), r = e - s * MT * 64, mt = r / 64, lane = r - mt * 64;
  const int ih = s / kw, j = s - ih * kw, i = 4 * ih + (lane >> 4), co = 16 * mt + (lane & 15);
  packet[e] = co < cout ? w[(i * kw + j) * cout + co] : 0.f;      // TF layout [8][kw][1][cout]
}

template <int KW, int COUT, bool STATS>
__global__ __launch_bounds__(kThreads, 2) void first_fwd(const float* __restrict__ x, const float* __restrict__ packet,
                                                       float* __restrict__ z, int frames, int T, double* __restrict__ part) {
  constexpr int KH = 8, PT = 3, PL = (KW - 1) / 2, RS = kF + KW - 1, STEPS = 2 * KW;
  // NREM (18 channels, tm_rem; the other kernels' remainder form): channels 0..15 are ONE M-tile; channels 16, 17 run as rows
  // (bin phase ph < 8, channel) over a window of 16 columns x 8 time rows (32 b32 steps), one column per group of 8 bins:
  // 17 columns per frame = 3 remainder tiles per two-frame tile.  324 + 96 MFMAs per tile instead of 720, shared as
  // 4 main tiles + 1 remainder tile (104) on waves 0..2 and 6 main tiles (108) on wave 3 -- before: 5 slots x 2 M-tiles = 180
  // on every wave (two of them idle slots).
  constexpr bool NREM = tm_rem(COUT) == 2;
  constexpr int MT = NREM ? 1 : (COUT + 15) / 16, kMain = STEPS * MT * 64, kData = kMain + (NREM ? 32 * 64 : 0);
  constexpr int kRowsFloats = ((kTF * KH * RS + 32 + 3) / 4) * 4;
  constexpr int kXElems = kTF * KH * kF, kPerX = (kXElems + kThreads - 1) / kThreads;
  constexpr int kTilesPerFrame = (kF + 15) / 16, kTiles = kTF * kTilesPerFrame;       // 9 per frame
  constexpr int NTW = NREM ? kTiles / kWaves : (kTiles + kWaves - 1) / kWaves;         // slots every wave runs
  constexpr int NXT = NREM ? kTiles - NTW * kWaves : 0;                                 // NREM: the tiles left over, all on the last wave
  constexpr int kRemCols = (kF + 7) / 8;                                                // 17 groups of 8 bins per frame
  static_assert(COUT % 2 == 0, "z is stored in float2 pieces");
  static_assert(!NREM || (NXT == 2 && kTF * kRemCols <= 16 * (kWaves - 1) && KW + 7 <= 16), "the 18-channel split of the work");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* rows = lds;                       // [kTF*8][RS] + slack
  float* lw = lds + kRowsFloats;           // packet
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, kq = lane >> 4;
  for (int e = tid; e < kRowsFloats + kData + 32; e += kThreads) lds[e] = e < kRowsFloats ? 0.f : packet[e - kRowsFloats];
  double st1[MT][4], st2[MT][4], sr1[2] = {0.0, 0.0}, sr2[2] = {0.0, 0.0};
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < 4; ++j) st1[mt][j] = st2[mt][j] = 0.0;
  const int ntiles = (frames + kTF - 1) / kTF;
  float prex[kPerX];
  // (utterance, time) of the tile's two frames from ONE wave-uniform division per tile: per element and tile the
  // runtime division frame / T was ~40 VALU instructions, 9 elements per thread -- three non-MFMA VALU per MFMA in the
  // first-layer kernels (rocprofv3, round 3); what depends on the element alone (fl, ih, f) is loop-invariant
  auto fetch_rows = [&](int tile) {
    static_assert(kTF == 2, "two frames per tile: the second is the first's successor");
    const int frame0 = tile * kTF;
    const int un0 = frame0 / T, ut0 = frame0 - un0 * T;
    const bool wrap = ut0 + 1 == T;
    const int un1 = wrap ? un0 + 1 : un0, ut1 = wrap ? 0 : ut0 + 1;
#pragma unroll
    for (int u = 0; u < kPerX; ++u) {
      const int e = tid + u * kThreads;
      const int fl = e / (KH * kF), r = e - fl * (KH * kF), ih = r / kF, f = r - ih * kF;
      const int un = fl ? un1 : un0, tt = (fl ? ut1 : ut0) + ih - PT;
      prex[u] = (e < kXElems && frame0 + fl < frames && tt >= 0 && tt < T) ? x[((size_t)un * T + tt) * kF + f] : 0.f;
    }
  };
  if ((int)blockIdx.x < ntiles) fetch_rows(blockIdx.x);
  __syncthreads();
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
#pragma unroll
    for (int u = 0; u < kPerX; ++u) {
      const int e = tid + u * kThreads;
      if (e < kXElems) {
        const int row = e / kF, f = e - row * kF;
        rows[row * RS + PL + f] = prex[u];
      }
    }
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) fetch_rows(tile + gridDim.x);
    pin();
    float p1[MT][4], p2[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < 4; ++j) p1[mt][j] = p2[mt][j] = 0.f;
    // one pass over NS column tiles q(t): MFMAs, then the stores and the lanes' shares of the sums
    auto main_pass = [&](auto ns_tag, auto qof) {
      constexpr int NS = decltype(ns_tag)::value;
      f32x4 acc[NS][MT];
      int boff[NS];
#pragma unroll
      for (int t = 0; t < NS; ++t) {
        const int q = qof(t), fl = q / kTilesPerFrame, f0 = 16 * (q - fl * kTilesPerFrame);
        boff[t] = q < kTiles ? (fl * KH + kq) * RS + f0 + n : kq * RS + n;     // idle slots re-read tile 0 (not stored)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[t][mt] = *reinterpret_cast<const f32x4*>(lw + kData + 16 * mt + 4 * kq);
      }
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        const int ih = s / KW, j = s - ih * KW;
        float a[MT], b[NS];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = lw[(s * MT + mt) * 64 + lane];
#pragma unroll
        for (int t = 0; t < NS; ++t) b[t] = rows[boff[t] + 4 * ih * RS + j];
#pragma unroll
        for (int t = 0; t < NS; ++t)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[t][mt] = mfma(a[mt], b[t], acc[t][mt]);
      }
      // D row = co = 16*mt + 4*kq + r, column = bin f0 + n
#pragma unroll
      for (int t = 0; t < NS; ++t) {
        const int q = qof(t), fl = q / kTilesPerFrame, f = 16 * (q - fl * kTilesPerFrame) + n;
        const int frame = tile * kTF + fl;
        if (q >= kTiles || f >= kF || frame >= frames) continue;
        float* op = z + ((size_t)frame * kF + f) * COUT;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int co0 = 16 * mt + 4 * kq;
          const f32x4 v = acc[t][mt];
          if (co0 + 1 < COUT) *reinterpret_cast<f32x2*>(op + co0) = f32x2{v.x, v.y};
          if (co0 + 3 < COUT) *reinterpret_cast<f32x2*>(op + co0 + 2) = f32x2{v.z, v.w};
          if constexpr (STATS) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              p1[mt][j] += v[j];
              p2[mt][j] = fmaf(v[j], v[j], p2[mt][j]);
            }
          }
        }
      }
    };
    // tile index q = wave + 4*slot: frame fl = q / 9, bins 16*(q % 9) .. +15
    main_pass(std::integral_constant<int, NTW>{}, [&](int t) { return wave + kWaves * t; });
    if constexpr (NREM) {
      if (wave == kWaves - 1) {
        main_pass(std::integral_constant<int, NXT>{}, [&](int t) { return NTW * kWaves + t; });
      } else {
        // remainder tile rt = wave: column n is group b8 of 8 bins of frame fl; rows 4 kq + r = (ph 2 kq + r / 2, channel 16 + r % 2)
        const int col = 16 * wave + n, fl = col >= kRemCols ? 1 : 0, b8 = col - fl * kRemCols;
        const bool live = col < kTF * kRemCols;
        const int boffr = live ? (fl * KH + kq) * RS + 8 * b8 : kq * RS;
        const f32x2 rb = *reinterpret_cast<const f32x2*>(lw + kData + 16);
        f32x4 racc = {rb.x, rb.y, rb.x, rb.y};
        const float* wr = lw + kMain;
#pragma unroll
        for (int sr = 0; sr < 32; ++sr) {
          const int ih = sr / 16, jw = sr - ih * 16;
          racc = mfma(wr[sr * 64 + lane], rows[boffr + 4 * ih * RS + jw], racc);
        }
        const int frame = tile * kTF + fl;
        float q1[2] = {0.f, 0.f}, q2[2] = {0.f, 0.f};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int f = 8 * b8 + 2 * kq + h;
          if (!live || f >= kF || frame >= frames) continue;
          const f32x2 v = {racc[2 * h], racc[2 * h + 1]};
          *reinterpret_cast<f32x2*>(z + ((size_t)frame * kF + f) * COUT + 16) = v;
          if constexpr (STATS) {
            q1[0] += v.x; q2[0] = fmaf(v.x, v.x, q2[0]);
            q1[1] += v.y; q2[1] = fmaf(v.y, v.y, q2[1]);
          }
        }
        if constexpr (STATS) {
          sr1[0] += (double)q1[0]; sr2[0] += (double)q2[0];
          sr1[1] += (double)q1[1]; sr2[1] += (double)q2[1];
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
    __syncthreads();
  }
  if constexpr (STATS) {
    double* red = reinterpret_cast<double*>(lds);             // [wave][32 channels][2]; the tile loop is over
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
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
    if constexpr (NREM) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        double a = sr1[c], b = sr2[c];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          a += __shfl_xor(a, o, 64);
          b += __shfl_xor(b, o, 64);
        }
        if (lane == 0) {
          red[(wave * 32 + 16 + c) * 2 + 0] = a;
          red[(wave * 32 + 16 + c) * 2 + 1] = b;
        }
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
// Output-layer dgrad (1x129, CH -> 1): dx[frame, f', ci] = sum_f dz[frame, f] * W[f' - f + 64, ci]
// as a dense Toeplitz GEMM: D[m = f'*CH + ci (129*CH rows), frame (N)] = sum_{k = f} A[m, f] * dz[frame, f],
// A[m, f] = W[f' - f + 64, ci] (zero outside the 129 taps), K = 129 padded to 132 (33 b32 steps).
// A (A-fragment order, rebuilt on the device every step) streams from L2; B = the frames' dz rows staged in LDS.
// One workgroup = 4 waves = 64 frames; wave w owns M-tiles w, w+4, ... in chunks of kDgMc accumulators.
// ---------------------------------------------------------------------------------------------
constexpr int kDgSteps = 33, kDgFrames = 64, kDgMc = 3;
// pack [mtile][step][lane] = A[16*mtile + (lane & 15), 4*step + (lane >> 4)]
static __global__ void pack_final_dgrad(const float* __restrict__ w, int CH, float* __restrict__ pack) {
  const int M = kF * CH, MT = (M + 15) / 16, total = MT * kDgSteps * 64;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int mt = e / (kDgSteps * 64), r = e - mt * kDgSteps * 64, s = r / 64, lane = r - s * 64;
  const int m = 16 * mt + (lane & 15), f = 4 * s + (lane >> 4);
  const int fp = m / CH, ci = m - fp * CH, tap = fp - f + 64;
  pack[e] = (m < M && f < kF && tap >= 0 && tap < kF) ? w[tap * CH + ci] : 0.f;
}

template <int CH>
__global__ __launch_bounds__(kThreads) void final_dgrad(const float* __restrict__ dz, const float* __restrict__ apack,
                                                         float* __restrict__ dx, int frames) {
  constexpr int M = kF * CH, MT = (M + 15) / 16;            // 1032 rows -> 65 M-tiles (CH 8)
  constexpr int kRow = 136;                                  // floats per staged dz row (132 used; 8 mod 32: no bank conflicts)
  __shared__ float rows[kDgFrames * kRow];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, kq = lane >> 4;
  const int f0 = blockIdx.x * kDgFrames;
  for (int e = tid; e < kDgFrames * kRow; e += kThreads) {
    const int fr = e / kRow, f = e - fr * kRow;
    rows[e] = (f < kF && f0 + fr < frames) ? dz[(size_t)(f0 + fr) * kF + f] : 0.f;
  }
  __syncthreads();
  // B[k = 4*s + kq][frame = 16*t + n]
  const float* bp = rows + n * kRow + kq;
  for (int m0 = wave * kDgMc; m0 < MT; m0 += kWaves * kDgMc) {
    f32x4 acc[kDgMc][4];
#pragma unroll
    for (int c = 0; c < kDgMc; ++c)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* ap = apack + (size_t)m0 * kDgSteps * 64 + lane;
#pragma unroll 3
    for (int s = 0; s < kDgSteps; ++s) {
      float a[kDgMc], b[4];
#pragma unroll
      for (int c = 0; c < kDgMc; ++c) a[c] = (m0 + c < MT) ? ap[(c * kDgSteps + s) * 64] : 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) b[t] = bp[16 * t * kRow + 4 * s];
#pragma unroll
      for (int c = 0; c < kDgMc; ++c)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[c][t] = mfma(a[c], b[t], acc[c][t]);
    }
    // D row = m = 16*(m0 + c) + 4*kq + r (four consecutive floats of the frame's [129*CH] row), column = frame
#pragma unroll
    for (int c = 0; c < kDgMc; ++c) {
      const int mrow = 16 * (m0 + c) + 4 * kq;
      if (m0 + c >= MT) continue;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int fr = f0 + 16 * t + n;
        if (fr >= frames) continue;
        float* op = dx + (size_t)fr * M + mrow;
        const f32x4 v = acc[c][t];
        if (mrow + 3 < M) {
          *reinterpret_cast<f32x2*>(op) = f32x2{v.x, v.y};          // M even (CH even): 8-byte aligned
          *reinterpret_cast<f32x2*>(op + 2) = f32x2{v.z, v.w};
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (mrow + r < M) op[r] = v[r];
        }
      }
    }
  }
}

}  // namespace tmm
}  // namespace rced
