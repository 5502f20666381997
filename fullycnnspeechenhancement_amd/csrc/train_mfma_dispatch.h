// Host-side launchers of the MFMA training kernels (kernels_train_mfma.h), shared by the translation units that
// instantiate them: train_api.hip (CR-CED V3 and R-CED V1 shapes) and train_mfma_v2.hip (R-CED V2's even-padded
// shapes) -- split so the two sets compile side by side.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <mutex>

#include "kernels_train_mfma.h"

namespace rced {
namespace tmd {

constexpr int kPairGrid = 2048;     // workgroups of the channel-aligned elementwise kernels (and their partial sums)

// Deterministic weight gradients (default): the wgrad kernels write per-wave slices into this buffer instead of adding to
// dW with fp32 atomics, and tmm::wg_reduce sums the slices in a fixed order.  The training step points it at the
// trainer's buffer for the duration of its backward pass (one caller thread per trainer); part == nullptr selects atomics.
struct WgDet {
  float** part = nullptr;         // the trainer's slice buffer, grown on demand; nullptr selects atomics (RCED_TRAIN_DET=0)
  size_t* cap_floats = nullptr;
  int* error = nullptr;           // raised when the buffer cannot be grown: the step then fails (RCED_ERR_ALLOC) -- a
                                  // deterministic trainer never falls back to atomics silently
};
inline thread_local WgDet g_wgdet;
// Launch helper: `launch(dW_arg, dbias_arg, pstride)` launches the wgrad kernel; slices = partial-sum slices it writes
// (grid x waves x parities: the grid comes from the runtime's occupancy answer, so the size is only known here).
template <class F>
inline int wg_launch(F&& launch, int slices, int nW, int nB, float* dW, float* dbias, hipStream_t st) {
  const WgDet& d = g_wgdet;
  const unsigned pstride = (unsigned)((nW + nB + 3) & ~3);
  if (!d.part) {
    launch(dW, dbias, 0u);          // atomics (not reproducible bit for bit): asked for with RCED_TRAIN_DET=0
    return 0;
  }
  const size_t need = (size_t)slices * pstride;
  if (need > *d.cap_floats) {       // first step, or a launcher whose grid grew: earlier launches still read the old buffer
    (void)hipStreamSynchronize(st);
    if (*d.part) (void)hipFree(*d.part);
    *d.part = nullptr;
    *d.cap_floats = 0;
    const size_t want = need + need / 4;
    if (hipMalloc(reinterpret_cast<void**>(d.part), want * sizeof(float)) != hipSuccess) {
      *d.part = nullptr;
      if (d.error) *d.error = 1;
      return -1;                    // nothing launched: the caller's step reports the failure before Adam runs
    }
    *d.cap_floats = want;
  }
  launch(*d.part, *d.part + nW, pstride);
  hipLaunchKernelGGL(tmm::wg_reduce, dim3((nW + nB + tmm::kWgrElems - 1) / tmm::kWgrElems), dim3(1024), 0, st, (const float*)*d.part, slices, pstride, nW, nB,
                     dW, dbias);
  return 1;
}

// Kernels that want more than the default dynamic LDS need the attribute once per (kernel, device): `done` is that
// kernel's bit mask over device ordinals (a process may hold trainers on several devices).
inline std::mutex g_launch_cache_mu;   // the per-kernel caches below are function-local statics shared by every trainer / thread
inline void allow_lds(const void* kernel, size_t lds, unsigned long long& done) {
  if (lds <= 48 * 1024) return;
  std::lock_guard<std::mutex> lock(g_launch_cache_mu);
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (done & bit) return;
  (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  done |= bit;
}

// Persistent grids are sized from the occupancy the runtime reports for the kernel (registers + LDS): cus x resident
// workgroups per CU, so every workgroup is resident from the start and walks the same number of tiles.
inline int resident_grid(const void* kernel, size_t lds, int cus, int& occ_cache, int threads = tmm::kThreads) {
  std::lock_guard<std::mutex> lock(g_launch_cache_mu);
  if (occ_cache <= 0) {
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, threads, lds) != hipSuccess || occ < 1) occ = 1;
    occ_cache = occ;
  }
  return cus * occ_cache;
}

constexpr int tm_packet_parities(int cout) { return cout == 8 ? 2 : 1; }   // Geo::kPH
constexpr size_t tm_packet_floats(int cin, int taps, int cout) {     // Geo::kPacket (static_asserts below)
  const int cinp = (cin + 1) & ~1;
  const int ph = tm_packet_parities(cout), R = ph == 1 ? tmm::tm_rem(cout) : 0, P = tmm::tm_rem_p(R, cinp);   // as Geo::kP
  const int K = (taps + ph - 1) * cinp, MT = R ? 1 : (cout + 15) / 16;
  const int KR = R ? (taps + P - 1) * cinp : 0;
  return (size_t)(K / 8) * MT * 128 + (size_t)((K % 8 + 3) / 4) * MT * 64 + (R ? (size_t)(KR / 8) * 128 + (size_t)((KR % 8 + 3) / 4) * 64 : 0) + 32;
}

// ---- the forward convolutions in the three-part bf16 form (tmm::conv_x6_fwd) ----
constexpr size_t tm_packet_x6_floats(int cin, int taps, int cout) {   // tmm::GeoX6::kPacket (static_assert below)
  const int ph = tm_packet_parities(cout), cs = tmm::x6_cs(cin, ph), K = (taps + ph - 1) * cs;
  const int R = ph == 1 ? tmm::tm_rem(cout) : 0, P = tmm::tm_rem_p(R, (cin + 1) & ~1), KR = R ? (taps + P - 1) * cs : 0;   // as GeoX6 / pack_packet_x6
  return (size_t)(((K + 31) / 32) * (R ? 1 : (cout + 15) / 16) + (KR + 31) / 32) * 3 * 64 * 4 + 32;
}
// the host-side sizes are the kernels' own for every shape the three nets' training steps launch
static_assert(tm_packet_floats(8, 9, 18) == (size_t)tmm::Geo<8, 9, 18>::kPacket && tm_packet_floats(18, 5, 30) == (size_t)tmm::Geo<18, 5, 30>::kPacket &&
              tm_packet_floats(30, 9, 8) == (size_t)tmm::Geo<30, 9, 8>::kPacket && tm_packet_floats(18, 9, 8) == (size_t)tmm::Geo<18, 9, 8>::kPacket &&
              tm_packet_floats(30, 5, 18) == (size_t)tmm::Geo<30, 5, 18>::kPacket && tm_packet_floats(8, 9, 30) == (size_t)tmm::Geo<8, 9, 30>::kPacket,
              "tm_packet_floats == Geo::kPacket");
static_assert(tm_packet_x6_floats(18, 5, 30) == (size_t)tmm::GeoX6<18, 5, 30>::kPacket && tm_packet_x6_floats(8, 9, 30) == (size_t)tmm::GeoX6<8, 9, 30>::kPacket,
              "tm_packet_x6_floats == GeoX6::kPacket");
inline int tm_packet_x6_threads(int cin, int taps, int cout) {     // pack_packet_x6: one thread per (step, M-tile, lane, element) + 32 shifts
  return (int)((tm_packet_x6_floats(cin, taps, cout) - 32) / (3 * 4) * 8) + 32;
}
template <int CIN, int TAPS, int COUT, bool STATS, int XF>
int tm_conv_x6_launch1(const float* in, const float* packet, float* out, int frames, int cus, double* part, tmm::XformArgs xa,
                       hipStream_t st) {
  const int ntiles = (frames + tmm::kTF - 1) / tmm::kTF;
  const size_t lds = (size_t)(tmm::conv_x6_red_off<CIN, TAPS, COUT, XF>() + (STATS ? tmm::kConvRedFloats : 0)) * sizeof(float);
  static unsigned long long attr = 0;
  static int occ = 0;
  const void* kfn = reinterpret_cast<const void*>(tmm::conv_x6_fwd<CIN, TAPS, COUT, STATS, XF>);
  allow_lds(kfn, lds, attr);
  const int grid = std::min(ntiles, std::min(resident_grid(kfn, lds, cus, occ), kPairGrid));
  hipLaunchKernelGGL((tmm::conv_x6_fwd<CIN, TAPS, COUT, STATS, XF>), dim3(grid), dim3(tmm::kThreads), lds, st, in, packet, out,
                     frames, part, xa);
  return grid;
}
// forward shapes built in this form: CR-CED's 18 -> 30 layers (no remainder pass; the 30 -> 8 layers' three planes + packet do not
// leave room for two workgroups per CU: tmm::GeoX6::kFits).  Returns the grid, 0 if not built.
#ifndef RCED_TM_X6_FWD_818
#define RCED_TM_X6_FWD_818 0   // 1: the 8 -> 18 forward convolutions (main pass + remainder pass) in the three-part bf16 form too.  Measured (round 6,
                               // A/B in one call, parity tests green): the step 40.30 -> 40.60 ms -- these layers (K = 72, 1.8 GB per call) wait for
                               // their tiles, not for the fp32 matrix pipe; not adopted
#endif
#if RCED_TM_X6_FWD_818
#define RCED_TM_X6_FWD(X) X(18, 5, 30) X(8, 9, 18)
#else
#define RCED_TM_X6_FWD(X) X(18, 5, 30)
#endif
inline bool tm_x6_has(int cin, int taps, int cout) {
#define X(CI, TP, CO) if (cin == CI && taps == TP && cout == CO) return true;
  RCED_TM_X6_FWD(X)
#undef X
  return false;
}
inline int tm_conv_x6(int cin, int taps, int cout, bool stats, const float* in, const float* packet, float* out, int frames, int cus,
                      double* part, const tmm::XformArgs* xa, hipStream_t st) {
  const tmm::XformArgs nx{nullptr, nullptr, nullptr, nullptr};
#define X(CI, TP, CO)                                                                                                              \
  if (cin == CI && taps == TP && cout == CO) {                                                                                     \
    if (xa) return stats ? tm_conv_x6_launch1<CI, TP, CO, true, tmm::kXfBnRelu>(in, packet, out, frames, cus, part, *xa, st)       \
                         : tm_conv_x6_launch1<CI, TP, CO, false, tmm::kXfBnRelu>(in, packet, out, frames, cus, nullptr, *xa, st);  \
    return stats ? tm_conv_x6_launch1<CI, TP, CO, true, tmm::kXfNone>(in, packet, out, frames, cus, part, nx, st)                  \
                 : tm_conv_x6_launch1<CI, TP, CO, false, tmm::kXfNone>(in, packet, out, frames, cus, nullptr, nx, st);             \
  }
  RCED_TM_X6_FWD(X)
#undef X
  return 0;
}

constexpr tmm::SumArgs kNoSums{nullptr, nullptr, nullptr, nullptr, nullptr};
template <int CIN, int TAPS, int COUT, bool ACCUM, bool STATS, int XF, bool SUMS = false>
int tm_conv_launch1(const float* in, const float* packet, float* out, int frames, int cus, double* part,
                    tmm::XformArgs xa, tmm::BnBwdArgs ba, hipStream_t st, tmm::SumArgs sa = kNoSums,
                    const float* acc_from = nullptr) {
  using G = tmm::Geo<CIN, TAPS, COUT>;
  const int ntiles = (frames + tmm::kTF - 1) / tmm::kTF;
  size_t lds = (G::kLdsFloats + (XF == tmm::kXfBnRelu ? 2 * CIN : XF == tmm::kXfBnBwd ? 4 * CIN : 0)) * sizeof(float);
  if constexpr (STATS || SUMS)   // (SUMS: + the z tile being written and the producer's folded BatchNorm) + the running sums
    lds = (size_t)(tmm::conv_red_off<CIN, TAPS, COUT, XF, SUMS>() + tmm::kConvRedFloats) * sizeof(float);
  if constexpr (tmm::conv_ks_on<CIN, TAPS, COUT>() && !SUMS)   // + the three parked partial sums of the K-split odd tile
    lds = (size_t)(tmm::conv_ks_off<CIN, TAPS, COUT, XF, STATS, SUMS>() + tmm::conv_ks_floats<CIN, TAPS, COUT>()) * sizeof(float);
  static unsigned long long attr = 0;
  static int occ = 0;
  const void* kfn = reinterpret_cast<const void*>(tmm::conv1xk_mfma<CIN, TAPS, COUT, ACCUM, STATS, XF, SUMS>);
  allow_lds(kfn, lds, attr);
  const int grid = std::min(ntiles, std::min(resident_grid(kfn, lds, cus, occ), kPairGrid));
  hipLaunchKernelGGL((tmm::conv1xk_mfma<CIN, TAPS, COUT, ACCUM, STATS, XF, SUMS>), dim3(grid), dim3(tmm::kThreads), lds, st, in,
                     packet, out, frames, part, xa, ba, sa, acc_from);
  return grid;
}
// One 1xk convolution on the MFMA kernels.  The shape decides the role: a layer's forward shape gets
//   out = conv(in) + shift, optionally with the per-workgroup (sum, sum of squares) records in `part` (stats) and
//   optionally with in = relu(bn(z)) rebuilt from the producer's z (xa);
// a dgrad shape gets  out (=|+=) conv(in)  with in = dz, optionally rebuilt from (d_u, z) (ba).
// Returns the grid size (= number of partial-sum records when stats), 0 if no kernel was built for the request.
// sa (dgrad shapes; overwrite mode, or -- 8-channel outputs -- the accumulating dgrad that adds the LAST contribution): also
// leave the producer's BatchNorm-backward records in `part` (tmm::SumArgs);
// 0 is returned when no such kernel exists for the shape and the caller launches again without sa.
// acc_from (accum only): the tensor the result is added to, when that is not `out` itself (out = acc_from + conv).
template <int CIN, int TAPS, int COUT, bool FWD>
int tm_conv_launch(bool accum, bool stats, const float* in, const float* packet, float* out, int frames, int cus,
                   double* part, const tmm::XformArgs* xa, const tmm::BnBwdArgs* ba, hipStream_t st,
                   const tmm::SumArgs* sa = nullptr, const float* acc_from = nullptr) {
  const tmm::XformArgs nx{nullptr, nullptr, nullptr, nullptr};
  const tmm::BnBwdArgs nb{nullptr, nullptr, nullptr, nullptr, nullptr, 1.0, nullptr};
  if (sa) {
    if constexpr (!FWD && CIN % 2 == 0 && COUT % 2 == 0) {
      if (stats || xa) return 0;
      if (accum) {
        // the accumulating form exists for the 8-channel tensors (CR-CED's skip sources) with the rebuilt dz only
        if constexpr (COUT == 8) {
          if (ba) return tm_conv_launch1<CIN, TAPS, COUT, true, false, tmm::kXfBnBwd, true>(in, packet, out, frames, cus, part, nx, *ba, st, *sa, acc_from);
        }
        return 0;
      }
      if (ba) return tm_conv_launch1<CIN, TAPS, COUT, false, false, tmm::kXfBnBwd, true>(in, packet, out, frames, cus, part, nx, *ba, st, *sa);
      if constexpr (COUT != 8)
        return tm_conv_launch1<CIN, TAPS, COUT, false, false, tmm::kXfNone, true>(in, packet, out, frames, cus, part, nx, nb, st, *sa);
    }
    return 0;
  }
  if constexpr (FWD) {
    if (accum || ba) return 0;
    if (xa) {
      if constexpr (CIN % 2 == 0) {
        if (stats) return tm_conv_launch1<CIN, TAPS, COUT, false, true, tmm::kXfBnRelu>(in, packet, out, frames, cus, part, *xa, nb, st);
        return tm_conv_launch1<CIN, TAPS, COUT, false, false, tmm::kXfBnRelu>(in, packet, out, frames, cus, nullptr, *xa, nb, st);
      }
      return 0;
    }
    if (stats) return tm_conv_launch1<CIN, TAPS, COUT, false, true, tmm::kXfNone>(in, packet, out, frames, cus, part, nx, nb, st);
    return tm_conv_launch1<CIN, TAPS, COUT, false, false, tmm::kXfNone>(in, packet, out, frames, cus, nullptr, nx, nb, st);
  } else {
    if (stats || xa) return 0;
    if (ba) {
      if constexpr (CIN % 2 == 0) {
        if (accum) return tm_conv_launch1<CIN, TAPS, COUT, true, false, tmm::kXfBnBwd>(in, packet, out, frames, cus, nullptr, nx, *ba, st, kNoSums, acc_from);
        return tm_conv_launch1<CIN, TAPS, COUT, false, false, tmm::kXfBnBwd>(in, packet, out, frames, cus, nullptr, nx, *ba, st);
      }
      return 0;
    }
    if (accum) return tm_conv_launch1<CIN, TAPS, COUT, true, false, tmm::kXfNone>(in, packet, out, frames, cus, nullptr, nx, nb, st, kNoSums, acc_from);
    return tm_conv_launch1<CIN, TAPS, COUT, false, false, tmm::kXfNone>(in, packet, out, frames, cus, nullptr, nx, nb, st);
  }
}


template <int CIN, int TAPS, int COUT, bool XF, bool DZF>
int tm_wgrad_launch1(const float* x, const float* dz, float* dW, float* dbias, int frames, int cus, tmm::XformArgs xa,
                     tmm::BnBwdArgs ba, hipStream_t st) {
  using G = tmm::Geo<CIN, TAPS, COUT>;
  const int ntiles = (frames + tmm::kTF - 1) / tmm::kTF;
  constexpr int PH = COUT == 8 ? 2 : 1;   // 8 output channels: two pixel parities share the 16 MFMA columns
  const size_t lds = (G::kInFloats + 64 + (size_t)(16 * G::kTiles + 4) * (PH == 2 ? 8 : 32) + 2 * CIN + 4 * COUT) * sizeof(float);
  static unsigned long long attr = 0;
  static int occ = 0;
  const void* kfn = reinterpret_cast<const void*>(tmm::wgrad1xk_mfma<CIN, TAPS, COUT, XF, DZF, PH>);
  allow_lds(kfn, lds, attr);
  const int grid = std::min(ntiles, resident_grid(kfn, lds, cus, occ));
  wg_launch([&](float* dw, float* db, unsigned ps) {
    hipLaunchKernelGGL((tmm::wgrad1xk_mfma<CIN, TAPS, COUT, XF, DZF, PH>), dim3(grid), dim3(tmm::kThreads), lds, st, x, dz, dw, db,
                       frames, xa, ba, ps);
  }, grid * tmm::kWaves * PH, TAPS * CIN * COUT, COUT, dW, dbias, st);
  return 1;
}
// xa: x is the producer's z (see tm_conv); ba: dz is d_u, rebuilt through BatchNorm backward from (d_u, z)
template <int CIN, int TAPS, int COUT>
int tm_wgrad_launch(const float* x, const float* dz, float* dW, float* dbias, int frames, int cus, const tmm::XformArgs* xa,
                    const tmm::BnBwdArgs* ba, hipStream_t st) {
  const tmm::XformArgs nx{nullptr, nullptr, nullptr, nullptr};
  const tmm::BnBwdArgs nb{nullptr, nullptr, nullptr, nullptr, nullptr, 1.0, nullptr};
  if (xa && ba) return tm_wgrad_launch1<CIN, TAPS, COUT, true, true>(x, dz, dW, dbias, frames, cus, *xa, *ba, st);
  if (xa) return tm_wgrad_launch1<CIN, TAPS, COUT, true, false>(x, dz, dW, dbias, frames, cus, *xa, nb, st);
  if (ba) return tm_wgrad_launch1<CIN, TAPS, COUT, false, true>(x, dz, dW, dbias, frames, cus, nx, *ba, st);
  return tm_wgrad_launch1<CIN, TAPS, COUT, false, false>(x, dz, dW, dbias, frames, cus, nx, nb, st);
}

// wgrad + dgrad of one layer in one kernel (tmm::bwd_fused_mfma).  Returns the grid size (= partial-sum records when sums).
template <int CIN, int TAPS, int COUT, bool XF, bool SUMS>
int tm_bwd_fused_launch(const float* x, const float* du, const float* packet, float* dx, float* dW, float* dbias, int frames,
                        int cus, double* part, tmm::XformArgs xa, tmm::BnBwdArgs ba, hipStream_t st) {
  using B = tmm::BwdGeo<CIN, TAPS, COUT>;
  const int ntiles = (frames + tmm::kTF - 1) / tmm::kTF;
  const size_t lds = (size_t)B::kLdsFloats * sizeof(float);
  static unsigned long long attr = 0;
  static int occ = 0;
  const void* kfn = reinterpret_cast<const void*>(tmm::bwd_fused_mfma<CIN, TAPS, COUT, XF, SUMS>);
  allow_lds(kfn, lds, attr);
  const int grid = std::min(ntiles, std::min(resident_grid(kfn, lds, cus, occ, tmm::kBwdThreads), kPairGrid));
  constexpr int PH = COUT == 8 ? 2 : 1;
  wg_launch([&](float* dw, float* db, unsigned ps) {
    hipLaunchKernelGGL((tmm::bwd_fused_mfma<CIN, TAPS, COUT, XF, SUMS>), dim3(grid), dim3(tmm::kBwdThreads), lds, st, x, du, packet,
                       dx, dw, db, frames, part, xa, ba, ps);
  }, grid * 4 * PH, TAPS * CIN * COUT, COUT, dW, dbias, st);
  return grid;
}

// One list-driven dispatcher set per translation unit: TM_FWD / TM_BWD are X-macro lists of (cin, taps, cout).
#define RCED_TM_DEFINE_DISPATCH(SUFFIX, TM_FWD, TM_BWD)                                                                   \
  int tm_conv##SUFFIX(bool fwd, int cin, int taps, int cout, bool accum, bool stats, const float* in, const float* packet, \
                      float* out, int frames, int cus, double* part, const tmm::XformArgs* xa, const tmm::BnBwdArgs* ba,    \
                      hipStream_t st, const tmm::SumArgs* sa = nullptr, const float* acc_from = nullptr) {                  \
    TM_FWD(RCED_TM_CONV_FWD_CASE)                                                                                           \
    TM_BWD(RCED_TM_CONV_BWD_CASE)                                                                                           \
    return 0;                                                                                                               \
  }                                                                                                                         \
  bool tm_has##SUFFIX(bool fwd, int cin, int taps, int cout) {                                                              \
    if (fwd) { TM_FWD(RCED_TM_HAS_CASE) } else { TM_BWD(RCED_TM_HAS_CASE) }                                                 \
    return false;                                                                                                           \
  }                                                                                                                         \
  int tm_wgrad##SUFFIX(int cin, int taps, int cout, const float* x, const float* dz, float* dW, float* dbias, int frames,   \
                       int cus, const tmm::XformArgs* xa, const tmm::BnBwdArgs* ba, hipStream_t st) {                       \
    TM_FWD(RCED_TM_WGRAD_CASE) /* the list tm_has(true, ...) answers from: fuse_dz / virt rely on the two agreeing */       \
    return 0;                                                                                                               \
  }
#define RCED_TM_CONV_FWD_CASE(CI, TP, CO)                \
  if (fwd && cin == CI && taps == TP && cout == CO)      \
    return rced::tmd::tm_conv_launch<CI, TP, CO, true>(accum, stats, in, packet, out, frames, cus, part, xa, ba, st, sa, acc_from);
#define RCED_TM_CONV_BWD_CASE(CI, TP, CO)                \
  if (!fwd && cin == CI && taps == TP && cout == CO)     \
    return rced::tmd::tm_conv_launch<CI, TP, CO, false>(accum, stats, in, packet, out, frames, cus, part, xa, ba, st, sa, acc_from);
#define RCED_TM_HAS_CASE(CI, TP, CO) \
  if (cin == CI && taps == TP && cout == CO) return true;
#define RCED_TM_WGRAD_CASE(CI, TP, CO)              \
  if (cin == CI && taps == TP && cout == CO)        \
    return rced::tmd::tm_wgrad_launch<CI, TP, CO>(x, dz, dW, dbias, frames, cus, xa, ba, st);

}  // namespace tmd
}  // namespace rced
