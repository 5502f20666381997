// CR-CED (V3) fused forward, TWO-TEAM variant of kernels_fused_v3.h (read that file first).
//
// Why: in the one-team kernel the two waves that share a SIMD reach every layer's epilogue + barrier
// + first-operand fetch together, so the SIMD's MFMA pipe idles ~2-3 k cycles per layer (~15-18 % of
// the kernel: tools/stamps.py).  Here a 512-thread workgroup is split into two independent TEAMS of
// four waves (one wave of each team per SIMD).  Each team owns a 2-frame tile, its own LDS
// activation buffers and its own barrier (an LDS counter: s_barrier is workgroup-wide), and the teams
// drift out of phase, so one team's sync bubble is filled by the other team's MFMAs.
//
// What had to change to make two teams fit:
//   * LDS: 2 x (B8 + B18 + B30 for 266 pixels) = 125 KB leaves 38.7 KB: exactly one ping-pong of the largest
//     weight packet (2 x 19.3 KB), SHARED by the two teams.  Both teams run the same layer sequence, so layer
//     number L (counted per team over all its tiles) lives in slot L & 1.  Protocol (two LDS words per slot):
//       fin[slot]  every wave adds 1 after its last read of the slot's packet; the wave that makes it a
//                  multiple of 8 knows all eight are done with layer L and streams layer L+2 into the slot
//                  by LDS-DMA (<= 19 instructions), waits for it to land, and sets
//       rdy[slot]  = L + 3 ("layer L+2 is here"); a wave entering layer L spins until rdy[L & 1] >= L + 1.
//     So a team can run at most one layer ahead of the other, and the refill stall is taken by the wave (and
//     team) that is behind, while the team that is ahead keeps the MFMA pipes busy.
//   * tiles: 17 sixteen-pixel tiles, 9 pair tiles and 3 remainder tiles per team, dealt over 4 waves;
//     team 1 plays the roles rotated by two waves so that the heavier roles of the two teams land on
//     different SIMDs.
//   * both teams of a workgroup run the same number of tile iterations (a team without a tile computes on
//     zeros and stores nothing), so the slot protocol never waits for a team that has left.
// Packet layout, pass structure, epilogues and the K-split hand-off are those of kernels_fused_v3.h.
#pragma once
#include <hip/hip_runtime.h>

#include "lds_dma.h"

#include "kernels_fused_v3.h"

namespace rced {
namespace v3t {

using v3::f32x2;
using v3::f32x4;
using v3::kF;
using v3::kHCh;
using v3::kS;
using v3::lds_peek;
using v3::lds_poke;
using v3::mfma;
using v3::pin;
using v3::relu4;

constexpr int kTF = 2;
constexpr int kNPX = kTF * kS;            // 266
constexpr int kTeams = 2, kTeamWaves = 4, kTeamThreads = 256;
constexpr int kThreads = kTeams * kTeamThreads;
constexpr int kB8S = 10;
constexpr int kB8Pad = 4, kB18Pad = 2, kB30Pad = 4;
constexpr int kB8Rows = kB8Pad + kNPX, kB18Rows = kB18Pad + kNPX, kB30Rows = kB30Pad + kNPX;
constexpr int kB8Off = 0;
constexpr int kB18Off = kB8Off + kB8Rows * kB8S;
constexpr int kB30Off = kB18Off + kB18Rows * 18;
constexpr int kTeamFloats = ((kB30Off + kB30Rows * 30 + 3) / 4) * 4;    // 15,624
constexpr int kWOff = kTeams * kTeamFloats;                             // shared weight ping-pong
constexpr int kWRegion = v3::kWRegion;
constexpr int kSyncOff = kWOff + 2 * kWRegion;                          // unsigned words: see sync_word()
constexpr int kLdsFloats = kSyncOff + 16;
constexpr int kLdsBytes = kLdsFloats * 4;
static_assert(kLdsBytes <= 160 * 1024, "LDS budget");
static_assert((kWOff * 4) % 16 == 0 && (kWRegion * 4) % 16 == 0, "LDS-DMA destinations are 16-byte aligned");
enum { kSyncCtr0 = 0, kSyncCtr1 = 4, kSyncFin = 8, kSyncRdy = 12 };     // ctr[team], fin[slot], rdy[slot]
constexpr int kX0Rows = kTF + 7;
constexpr int kX0Floats = ((kX0Rows * kS + 24 + 3) / 4) * 4;            // 1224
constexpr int kX0Off = kB30Off + kB30Pad * 30;
static_assert(kX0Floats <= 5 * kTeamThreads && kX0Floats <= 44 * 30, "X0 staging");
constexpr int kScratchOff = kB18Off + kB18Pad * 18 + 8 * 18;            // K-split hand-off, in the team's B18
constexpr int kFlagOff = kScratchOff + 256;
constexpr int kDepthA = 2;                                              // A-fragment (LDS) prefetch depth, steps

using v3::kL2Steps;
using v3::kL3Steps;
using v3::kW1;
using v3::kW1Data;
using v3::kW1Main;
using v3::kW2;
using v3::kW2Data;
using v3::kW3;
using v3::kW3Data;
using v3::kWBlock;

struct Params {
  const float* x;
  float* h;
  const float* wpack;
  int N, T;
  int tiles_per_utt;   // ceil(T / 2)
  int total_tiles;
  unsigned long long* stamps;   // diagnostic builds only (RCED_STAMPS): [wave][8] cycle sums of workgroup 0
};

__device__ __forceinline__ bool px_valid(int px) {
  const int fr = px / kS;
  return px < kNPX && (px - fr * kS) < kF;
}
__device__ __forceinline__ bool span_has_gap(int p0, int len) {
  const int fr = p0 / kS;
  return p0 + len > kNPX || (p0 - fr * kS) + len > kF;
}

// Team barrier: every wave of the team has finished its LDS writes and reads of this phase.
__device__ __forceinline__ void team_barrier(unsigned* ctr, unsigned& phase, int lane) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  phase += kTeamWaves;
  if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  for (int spin = 0; spin < (1 << 24); ++spin) {
    if ((int)(__builtin_amdgcn_readfirstlane(lds_peek(ctr)) - phase) >= 0) break;
    __builtin_amdgcn_s_sleep(1);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Implicit-GEMM pass as v3::gemm_pass: A operand (packet in the shared LDS slot) kDepthA steps ahead, B operand
// (the team's activations) DEPTH steps ahead.
template <int NR, int NX, int MT, int XMT, int NB64, int XS0, int XS1, bool XTAIL, int STRIDE, int DEPTH>
__device__ __forceinline__ void gemm_pass(const float* act, int off0, int offx, int tailoff, const float* w,
                                          int lane, f32x4 (&acc)[NR + NX][MT]) {
  constexpr int NT = NR + NX, RING = DEPTH + 1, RA = kDepthA + 1;
  const f32x2* wp = reinterpret_cast<const f32x2*>(w) + lane;
  const float* wt = w + NB64 * MT * 128 + lane;
  f32x2 a[RA][MT], b[RING][NT];
  float at[MT], bt[NT];
  auto xlive = [](int s) { return NX > 0 && s >= XS0 && s < XS1; };
  auto xmt = [](int mt) { return XMT < 0 || mt == XMT; };
  auto loadA = [&](int s) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a[s % RA][mt] = wp[(s * MT + mt) * 64];
  };
  auto loadB = [&](int s) {
#pragma unroll
    for (int t = 0; t < NR; ++t) b[s % RING][t] = *reinterpret_cast<const f32x2*>(act + off0 + t * STRIDE + 8 * s);
    if constexpr (NX > 0)
      if (xlive(s)) b[s % RING][NR] = *reinterpret_cast<const f32x2*>(act + offx + 8 * s);
  };
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) at[mt] = wt[mt * 64];
#pragma unroll
  for (int s = 0; s < kDepthA && s < NB64; ++s) loadA(s);
#pragma unroll
  for (int t = 0; t < NR; ++t) bt[t] = act[off0 + t * STRIDE + 8 * NB64 + tailoff];
  if constexpr (NX > 0 && XTAIL) bt[NR] = act[offx + 8 * NB64 + tailoff];
#pragma unroll
  for (int s = 0; s < DEPTH && s < NB64; ++s) loadB(s);
#pragma unroll
  for (int s = 0; s < NB64; ++s) {
    if (s + kDepthA < NB64) loadA(s + kDepthA);
    if (s + DEPTH < NB64) loadB(s + DEPTH);
    pin();
#pragma unroll
    for (int e = 0; e < 2; ++e) {
#pragma unroll
      for (int t = 0; t < NR; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[t][mt] = mfma(a[s % RA][mt][e], b[s % RING][t][e], acc[t][mt]);
      if constexpr (NX > 0)
        if (xlive(s)) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            if (xmt(mt)) acc[NR][mt] = mfma(a[s % RA][mt][e], b[s % RING][NR][e], acc[NR][mt]);
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

// Layer 1 main (NMR regular + NMX extra tiles, 9 b64 steps) and remainder (NR tiles, 16 b64 steps)
// interleaved.  FIRST: block 0 (8x9 kernel on the input rows, b32 steps: 18 / 32).
template <int NMR, int NMX, int NR, bool FIRST>
__device__ __forceinline__ void l1_pass(const float* in, int offm0, int offmx, int offr, const float* w,
                                        int lane, f32x4 (&accm)[NMR + NMX][1], f32x4 (&accr)[2]) {
  constexpr int NM = NMR + NMX;
  constexpr int MAIN = FIRST ? 18 : 9, REM = FIRST ? 32 : 16;
  constexpr int SLOTS = NR > 0 ? REM : MAIN;
  constexpr int DB = 2, RB = DB + 1, RA = kDepthA + 1;
  // operands: b64 (f32x2) for blocks 1..4, b32 for block 0 (kept in .x)
  f32x2 am[RA], ar[RA], bm[RB][NM], br[RB];
  auto main_step = [](int i) { return NR > 0 ? ((i * MAIN) / REM != ((i + 1) * MAIN) / REM ? (i * MAIN) / REM : -1) : i; };
  auto loadA = [&](int i) {
    if constexpr (NR > 0) {
      if constexpr (FIRST) ar[i % RA].x = w[kW1Main + i * 64 + lane];
      else ar[i % RA] = reinterpret_cast<const f32x2*>(w + kW1Main)[i * 64 + lane];
    }
    const int m = main_step(i);
    if (m >= 0) {
      if constexpr (FIRST) am[i % RA].x = w[m * 64 + lane];
      else am[i % RA] = reinterpret_cast<const f32x2*>(w)[m * 64 + lane];
    }
  };
  auto loadB = [&](int i) {
    if constexpr (NR > 0) {
      if constexpr (FIRST) br[i % RB].x = in[offr + (i / 16) * 4 * kS + (i % 16)];
      else br[i % RB] = *reinterpret_cast<const f32x2*>(in + offr + kB8S * i);
    }
    const int m = main_step(i);
    if (m >= 0) {
      const int d = FIRST ? (m / 9) * 4 * kS + (m % 9) : kB8S * m;
#pragma unroll
      for (int t = 0; t < NM; ++t) {
        const int o = (t < NMR ? offm0 + t * (FIRST ? 64 : 64 * kB8S) : offmx) + d;
        if constexpr (FIRST) bm[i % RB][t].x = in[o];
        else bm[i % RB][t] = *reinterpret_cast<const f32x2*>(in + o);
      }
    }
  };
#pragma unroll
  for (int i = 0; i < kDepthA && i < SLOTS; ++i) loadA(i);
#pragma unroll
  for (int i = 0; i < DB && i < SLOTS; ++i) loadB(i);
#pragma unroll
  for (int i = 0; i < SLOTS; ++i) {
    if (i + kDepthA < SLOTS) loadA(i + kDepthA);
    if (i + DB < SLOTS) loadB(i + DB);
    pin();
#pragma unroll
    for (int e = 0; e < (FIRST ? 1 : 2); ++e) {
      if constexpr (NR > 0) accr[i & 1] = mfma(ar[i % RA][e], br[i % RB][e], accr[i & 1]);
      if (main_step(i) >= 0) {
#pragma unroll
        for (int t = 0; t < NM; ++t) accm[t][0] = mfma(am[i % RA][e], bm[i % RB][t][e], accm[t][0]);
      }
    }
    pin();
  }
}

template <int MT, int COUT>
__device__ __forceinline__ void store_p1(float* out, const f32x4 (&acc)[MT], int px, int kq, bool gap) {
  const bool ok = gap ? px_valid(px) : true;
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

struct XStage {
  float v[5];
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

// ---- layers, by ROLE (0..3) within the team ---------------------------------------------------
// 16-pixel tiles 0..16: role r owns r + 4*slot (slot < 4); tile 16 extra.  Pair tiles 0..8: r + 4*slot
// (slot < 2); tile 8 extra.  Remainder tiles 0..2.
//   layer 1: role 0 = 5 main tiles; roles 1,2,3 = 4 main + remainder tile role-1   (90 / 104 MFMAs)
//   layer 2: roles 0,1 = 4 tiles + one M-tile of tile 16                           (207 / 184)
//   layer 3: roles 2,3 = 2 pair tiles + half of pair tile 8 along K                (188 / 150)
template <int NMX, int NR>
__device__ __forceinline__ void layer1(float* tb, const float* w, bool first, int role, int lane) {
  const float* lds_shift = w + kW1Data;
  constexpr int NM = 4 + NMX;
  const int n = lane & 15, kq = lane >> 4;
  float* b8 = tb + kB8Off + kB8Pad * kB8S;
  float* b18 = tb + kB18Off + kB18Pad * 18;
  const float* x0 = tb + kX0Off;
  const int px0 = 16 * role + n, pxx = 16 * 16 + n;
  const int rt = role - 1;                       // remainder tile of roles 1..3
  const int pxr = 8 * (16 * rt + n);
  f32x4 accm[NM][1], accr[2];
  const f32x4 sh = *reinterpret_cast<const f32x4*>(lds_shift + 4 * kq);
  const f32x2 s2 = *reinterpret_cast<const f32x2*>(lds_shift + 16);
#pragma unroll
  for (int t = 0; t < NM; ++t) accm[t][0] = sh;
  accr[0] = f32x4{s2.x, s2.y, s2.x, s2.y};
  accr[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (first)
    l1_pass<4, NMX, NR, true>(x0, px0 + kq * kS, pxx + kq * kS, pxr + kq * kS, w, lane, accm, accr);
  else
    l1_pass<4, NMX, NR, false>(b8, (px0 - 4) * kB8S + 2 * kq, (pxx - 4) * kB8S + 2 * kq, (pxr - 4) * kB8S + 2 * kq, w,
                               lane, accm, accr);
#pragma unroll
  for (int t = 0; t < 4; ++t) store_p1<1, 18>(b18, accm[t], px0 + 64 * t, kq, span_has_gap(16 * (role + 4 * t), 16));
  if constexpr (NMX > 0) store_p1<1, 18>(b18, accm[4], pxx, kq, true);
  if constexpr (NR > 0) {
    const f32x4 v = relu4(accr[0] + accr[1]);
    const int pa = pxr + 2 * kq;
    if (px_valid(pa)) *reinterpret_cast<f32x2*>(b18 + pa * 18 + 16) = f32x2{v.x, v.y};
    if (px_valid(pa + 1)) *reinterpret_cast<f32x2*>(b18 + (pa + 1) * 18 + 16) = f32x2{v.z, v.w};
  }
}

template <int XMT>
__device__ __forceinline__ void layer2(float* tb, const float* w, int role, int lane) {
  const float* lds_shift = w + kW2Data;
  constexpr int NX = XMT >= 0 ? 1 : 0, NT = 4 + NX;
  const int n = lane & 15, kq = lane >> 4;
  const float* b18 = tb + kB18Off + kB18Pad * 18;
  float* b30 = tb + kB30Off + kB30Pad * 30;
  const int px0 = 16 * role + n, pxx = 16 * 16 + n;
  f32x4 acc[NT][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const f32x4 sh = *reinterpret_cast<const f32x4*>(lds_shift + 16 * mt + 4 * kq);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t][mt] = sh;
  }
  const int tailoff = (kq < 1 ? kq : 1) - 2 * kq;
  gemm_pass<4, NX, 2, XMT, kL2Steps, 0, kL2Steps, true, 64 * 18, 1>(b18, (px0 - 2) * 18 + 2 * kq, (pxx - 2) * 18 + 2 * kq,
                                                                   tailoff, w, lane, acc);
#pragma unroll
  for (int t = 0; t < 4; ++t) store_p1<2, 30>(b30, acc[t], px0 + 64 * t, kq, span_has_gap(16 * (role + 4 * t), 16));
  if constexpr (NX > 0) {   // one M-tile of tile 16 (pixels 256..271: 262.. is gap / past the tile)
    const bool ok = px_valid(pxx);
    const int co0 = 16 * XMT + 4 * kq;
    f32x4 v = relu4(acc[4][XMT]);
    if (!ok) v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (pxx < kNPX) {
      float* p = b30 + pxx * 30 + co0;
      if (co0 + 1 < 30) *reinterpret_cast<f32x2*>(p) = f32x2{v.x, v.y};
      if (co0 + 3 < 30) *reinterpret_cast<f32x2*>(p + 2) = f32x2{v.z, v.w};
    }
  }
}

constexpr int kRolePlain = 0, kRoleReducer = 1, kRoleHelper = 2;
constexpr int kL3Split = 19;

template <int ROLE>
__device__ __forceinline__ void layer3(const Params& P, float* tb, const float* w, int blk, int role, int lane,
                                       unsigned tag, bool live, int utt, int t0, f32x4 (&skip_ce1)[3],
                                       f32x4 (&skip_ce2)[3]) {
  const float* lds_shift = w + kW3Data;
  constexpr int NX = ROLE == kRolePlain ? 0 : 1, NT = 2 + NX;
  constexpr int NEPI = ROLE == kRoleHelper ? 2 : NT;
  const int n = lane & 15, kq = lane >> 4;
  const float* b30 = tb + kB30Off + kB30Pad * 30;
  float* b8 = tb + kB8Off + kB8Pad * kB8S;
  const int q0 = 16 * role + n, qx = 16 * 8 + n;
  f32x4 acc[NT][1];
  const f32x4 sh = *reinterpret_cast<const f32x4*>(lds_shift + 4 * (kq & 1));
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t][0] = (t == 2 && ROLE == kRoleHelper) ? f32x4{0.f, 0.f, 0.f, 0.f} : sh;
  constexpr int XS0 = ROLE == kRoleHelper ? kL3Split : 0;
  constexpr int XS1 = ROLE == kRoleReducer ? kL3Split : kL3Steps;
  gemm_pass<2, NX, 1, -1, kL3Steps, XS0, XS1, ROLE == kRoleHelper, 64 * 60, 2>(
      b30, (2 * q0 - 4) * 30 + 2 * kq, (2 * qx - 4) * 30 + 2 * kq, -kq, w, lane, acc);
  if constexpr (ROLE == kRoleHelper) {
    *reinterpret_cast<f32x4*>(tb + kScratchOff + 4 * lane) = acc[2][0];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) lds_poke(tb + kFlagOff, tag);
  }
  if constexpr (ROLE == kRoleReducer) {
    for (int spin = 0; spin < (1 << 22); ++spin) {
      if (__builtin_amdgcn_readfirstlane(lds_peek(tb + kFlagOff)) == tag) break;
      __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    acc[2][0] += *reinterpret_cast<const f32x4*>(tb + kScratchOff + 4 * lane);
  }
#pragma unroll
  for (int t = 0; t < NEPI; ++t) {
    const int q = (t < 2) ? q0 + 64 * t : qx;
    const int px = 2 * q + (kq >> 1);
    f32x4 v = relu4(acc[t][0]);
    if (blk == 3) v += skip_ce2[t];
    if (blk == 4) v += skip_ce1[t];
    const bool gap = span_has_gap(32 * (t < 2 ? role + 4 * t : 8), 32);
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
    } else if (live && ok && t0 + fr < P.T) {
      float* hp = P.h + (((size_t)utt * P.T + t0 + fr) * kF + f) * kHCh + 4 * (kq & 1);
      *reinterpret_cast<f32x4*>(hp) = v;
    }
  }
}

// One wave streams a whole packet global -> LDS (LDS-DMA, 1 KiB per instruction).
template <int NFLOATS>
__device__ __forceinline__ void wave_dma(const float* __restrict__ src, float* dst, int lane) {
  constexpr int n4 = NFLOATS / 4, chunks = (n4 + 63) / 64;
#pragma unroll
  for (int c = 0; c < chunks; ++c) {
    const int idx = c * 64 + lane;
    if (idx < n4)
      lds_dma16(src + (size_t)idx * 4, dst + c * 256);
  }
}
// Packet of layer number l (any team's count): layer l % 15 of the net.
__device__ __forceinline__ void refill(const float* __restrict__ wpack, float* dst, unsigned l, int lane) {
  const unsigned li = l % 15u, blk = li / 3u, j = li - 3u * blk;
  const float* src = wpack + blk * kWBlock;
  if (j == 0) wave_dma<kW1>(src, dst, lane);
  else if (j == 1) wave_dma<kW2>(src + kW1, dst, lane);
  else wave_dma<kW3>(src + kW1 + kW2, dst, lane);
}

__global__ __launch_bounds__(kThreads) void fused_v3t_kernel(Params P) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int team = wave >> 2;
  const int role = (wave + 2 * team) & 3;   // team 1 plays the roles rotated by two waves
  const int ttid = tid & (kTeamThreads - 1);
  for (int e = tid; e < kLdsFloats; e += kThreads) lds[e] = 0.f;
  __syncthreads();
  unsigned* sync = reinterpret_cast<unsigned*>(lds + kSyncOff);
  // layers 0 and 1 of the first tile into the two slots
  if (wave == 0) refill(P.wpack, lds + kWOff, 0, lane);
  if (wave == 1) refill(P.wpack, lds + kWOff + kWRegion, 1, lane);
  if (tid == 0) { sync[kSyncRdy + 0] = 1; sync[kSyncRdy + 1] = 2; }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();   // last workgroup-wide barrier: from here on the two teams only meet through fin / rdy

  float* tb = lds + team * kTeamFloats;
  unsigned* ctr = sync + (team ? kSyncCtr1 : kSyncCtr0);
  unsigned phase = 0, epoch = 0;
  unsigned L = 0;   // this team's layer count (identical sequence in both teams)
#if RCED_STAMPS
  unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // enter wait, math L1/L2/L3, leave, team barrier
  unsigned long long tq = v3::stamp();
#define TSTAMP(i) { const unsigned long long tn_ = v3::stamp(); ts[i] += tn_ - tq; tq = tn_; }
#else
#define TSTAMP(i)
#endif

  // enter layer L: its packet must have landed in slot L & 1
  auto enter = [&]() -> const float* {
    for (int spin = 0; spin < (1 << 24); ++spin) {
      if ((int)(__builtin_amdgcn_readfirstlane(lds_peek(sync + kSyncRdy + (L & 1u))) - (L + 1u)) >= 0) break;
      __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return lds + kWOff + (L & 1u) * kWRegion;
  };
  // leave layer L: this wave no longer reads the slot; the last of the 8 waves refills it with layer L + 2
  auto leave = [&]() {
    unsigned old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(sync + kSyncFin + (L & 1u), 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
    old = __builtin_amdgcn_readfirstlane(old);
    if ((old & 7u) == 7u) {
      refill(P.wpack, lds + kWOff + (L & 1u) * kWRegion, L + 2u, lane);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) lds_poke(sync + kSyncRdy + (L & 1u), L + 3u);
    }
    ++L;
  };

  XStage xst;
  auto xload = [&](int tl) {
    const bool live = tl < P.total_tiles;
    const int u = live ? tl / P.tiles_per_utt : 0;
    const int t0 = live ? (tl - u * P.tiles_per_utt) * kTF : 0;
    const float* xu = P.x + (size_t)u * P.T * kF;
#pragma unroll
    for (int i = 0; i < 5; ++i) xst.v[i] = xstage_one(P, live, xu, t0, ttid + i * kTeamThreads);
  };
  const int tstride = gridDim.x * kTeams;
  int tile = blockIdx.x * kTeams + team;
  xload(tile);

  // both teams iterate while the workgroup's FIRST tile of the round exists (team 1's may be missing: dummy)
  for (; tile - team < P.total_tiles; tile += tstride) {
    const bool live = tile < P.total_tiles;
    const int utt = live ? tile / P.tiles_per_utt : 0;
    const int t0 = live ? (tile - utt * P.tiles_per_utt) * kTF : 0;
#pragma unroll
    for (int i = 0; i < 5; ++i)
      if (ttid + i * kTeamThreads < kX0Floats) tb[kX0Off + ttid + i * kTeamThreads] = xst.v[i];
    team_barrier(ctr, phase, lane);

    f32x4 skip_ce1[3], skip_ce2[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) skip_ce1[t] = skip_ce2[t] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int blk = 0; blk < 5; ++blk) {
      {
        TSTAMP(7);
        const float* w = enter();
        TSTAMP(0);
        if (role == 0) layer1<1, 0>(tb, w, blk == 0, role, lane);
        else layer1<0, 1>(tb, w, blk == 0, role, lane);
        TSTAMP(1);
        leave();
        TSTAMP(4);
        team_barrier(ctr, phase, lane);
        TSTAMP(5);
      }
      {
        const float* w = enter();
        TSTAMP(0);
        if (role == 0) layer2<0>(tb, w, role, lane);
        else if (role == 1) layer2<1>(tb, w, role, lane);
        else layer2<-1>(tb, w, role, lane);
        TSTAMP(2);
        leave();
        TSTAMP(4);
        team_barrier(ctr, phase, lane);
        TSTAMP(5);
      }
      {
        if (blk == 4) xload(tile + tstride);   // next tile's input rows, one layer ahead
        ++epoch;
        const unsigned tag = 0x80000000u | epoch;
        TSTAMP(7);
        const float* w = enter();
        TSTAMP(0);
        if (role == 2) layer3<kRoleReducer>(P, tb, w, blk, role, lane, tag, live, utt, t0, skip_ce1, skip_ce2);
        else if (role == 3) layer3<kRoleHelper>(P, tb, w, blk, role, lane, tag, live, utt, t0, skip_ce1, skip_ce2);
        else layer3<kRolePlain>(P, tb, w, blk, role, lane, tag, live, utt, t0, skip_ce1, skip_ce2);
        TSTAMP(3);
        leave();
        TSTAMP(4);
        team_barrier(ctr, phase, lane);
        TSTAMP(5);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // a refill issued for layers nobody will run
#if RCED_STAMPS
  if (P.stamps && blockIdx.x == 0 && lane == 0)
    for (int i = 0; i < 8; ++i) P.stamps[wave * 8 + i] = ts[i];
#endif
#undef TSTAMP
}

}  // namespace v3t
}  // namespace rced
