// Fused-kernel runtime: packs BN-folded weights into MFMA A-fragment streams, owns the hand-off
// workspace, launches the fused multi-layer kernel + the final Toeplitz GEMM.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/rced.h"
#include "kernels_fused_chain.h"
#include "kernels_fused_v3.h"
#include "kernels_frame16.h"
#include "kernels_final_x6.h"
#include "rced_internal.h"

using namespace rced;

inline unsigned short bf16_rne(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0;   // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

inline int env_default(const char* name, int dflt) {   // an environment variable as the DEFAULT of a per-handle option
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
struct rced_fused {
  float* scratch = nullptr;   // V1/V2: skip fragments, per workgroup
  size_t scratch_bytes = 0;
  float* wpack = nullptr;     // packed A-fragment stream (CR-CED: of the F32 form, built when that form is first selected)
  float* wpack_x6 = nullptr;  // CR-CED, X6 form (v3::kGTotal floats; built when that form is first selected)
  float* wpack_t = nullptr;   // CR-CED, fused form (v3::kTTotal floats)
  float* wpack_a = nullptr;   // CR-CED, all-x6 form (v3::kATotal floats)
  float* fin_tab = nullptr;   // ... its decode_final tap table (v3::kFinTFloats)
  // Per-handle options of the R-CED output layer's kernel; the environment variables of the same meaning only supply the
  // defaults, read when the handle is created (rced_create), never afterwards.
  int final_x6 = 1;           // option "final_x6" (default: RCED_FINAL_X6): 1 = x6::final_gemm_x6_kernel (three-part bf16 products), 0 = fp32 MFMA
  int final_lds = 1;          // option "final_lds" (default: RCED_FINAL_LDS): the fp32 kernel with (1) / without (0) LDS staging of its B operand
  int v3_l2x6 = 3;            // option "v3_l2x6": 3 = every layer at fp32 quality on the bf16 matrix pipe (the product); 0 = every layer on the fp32
                              // MFMA (the bit-exact comparator): kernels_fused_v3.h.  (1, 2: rounds 3 / 4's forms, RCED_V3_LEGACY_FORMS builds only)
  float* fin_apack = nullptr; // v3::kFinPack
  float fin_bias = 0.f;
  float* h = nullptr;         // [frames, 129, 8] hand-off to the final layer
  size_t h_bytes = 0;
  unsigned long long* stamps = nullptr;  // diagnostic builds (RCED_STAMPS) only
  int bf16 = 0;               // option "bf16" (V1/V2): bf16 activations + weights, one launch (kernels_frame16.h)
  unsigned* wpack16 = nullptr;   // its packet stream (built when the option is first set)
  unsigned short* fin_apack_x6 = nullptr;  // ... as three bf16 parts per value: fp32 quality on the bf16 pipe (x6::final_gemm_x6_kernel)
  unsigned* scratch16 = nullptr; // its skip fragments, per wave (2 workgroups per CU x 4 waves)
  int bf16_wgs_per_cu = 1;
  int bf16_frames = 0;        // option "bf16_frames": frames per workgroup of the bf16 kernel (0 = chosen per call, 4, 8)
  int grid_limit = 0;         // option "fused_grid": workgroups of the persistent kernel (0 = #CUs)
  int latency_form = env_default("RCED_LATENCY_FORM", 1);   // option "latency_form" (V1/V2, fp32): one-frame tiles for calls with fewer
                                                                  // 3-frame tiles than CUs (chain_forward)
  unsigned* err_host = nullptr;   // sticky error word of the CR-CED kernel's wave-to-wave hand-offs: pinned host memory the
  unsigned* err_dev = nullptr;    // kernel reaches through its device alias, so the host reads it without a device sync
};

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return rced_fail(e_ == hipErrorOutOfMemory ? RCED_ERR_ALLOC : RCED_ERR_HIP, "%s: %s", #expr, \
                       hipGetErrorString(e_));                                                 \
  } while (0)

namespace {

// BN-folded weight of layer L: [kh,kw,cin,cout4] host copy
inline float wq(const rced_layer_dev& d, int tap, int ci, int co, int cin) {
  return d.host_w[((size_t)tap * cin + ci) * d.cout4 + co];
}

// three bf16 parts of a weight (round to nearest at every step): v = h + m + l to 2^-24
inline void split3(float v, unsigned short* h, unsigned short* mm, unsigned short* l) {
  auto b2f = [](unsigned short b) { const unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; };
  *h = bf16_rne(v);
  const float r1 = v - b2f(*h);
  *mm = bf16_rne(r1);
  *l = bf16_rne(r1 - b2f(*mm));
}

// CR-CED weight streams (kernels_fused_v3.h, "packed weight streams").  x6 = false: one LDS packet per layer (F32 form);
// x6 = true: layer 1 / layer 2 as register images for 16-byte-per-lane global loads, layer 3's packet unchanged.
void pack_v3(const rced_model* m, int form, std::vector<float>* wpack) {
  const bool x6 = form != 0, fusedf = form >= 2, allx6 = form == 3;
  wpack->assign(allx6 ? v3::kATotal : fusedf ? v3::kTTotal : x6 ? v3::kGTotal : v3::kWTotal, 0.f);
  auto put_shift = [&](float* at, int layer) {  // the packet's last 32 floats: shift[co]
    const int cout = m->net->layer[layer].cout;
    for (int c = 0; c < cout; ++c) at[c] = m->layers[layer].host_shift[c];
  };
  float* dst = wpack->data();
  for (int blk = 0; blk < 5; ++blk) {
    const rced_layer_dev& l1 = m->layers[3 * blk + 0];
    const rced_layer_dev& l2 = m->layers[3 * blk + 1];
    const rced_layer_dev& l3 = m->layers[3 * blk + 2];
    // ---- layer 1 = main pass (channels 0..15) + remainder pass (channels 16,17 x 8 pixel phases)
    // A[row][K-step s][e]: block 0 (8x9x1): one k per lane and step: main s = ih*9 + j, remainder s = ih*16 + u, time tap
    // 4*ih + kq; blocks 1..4 (1x9, 8 -> 18): two k per lane and step, k = 8s + 2kq + e = tap*8 + ci.
    // Remainder rows i = (phase r = i>>1, channel 16 + (i&1)), frequency tap = u - r over a window of 16 taps.
    auto w1main = [&](int lane, int s, int e) {
      const int i = lane & 15, kq = lane >> 4;
      if (blk == 0) return wq(l1, (4 * (s / 9) + kq) * 9 + s % 9, 0, i, 1);
      const int k = 8 * s + 2 * kq + e;
      return wq(l1, k / 8, k % 8, i, 8);
    };
    auto w1rem = [&](int lane, int s, int e) {
      const int i = lane & 15, kq = lane >> 4, r = i >> 1, co = 16 + (i & 1);
      if (blk == 0) {
        const int tap = s % 16 - r;
        return (tap >= 0 && tap < 9) ? wq(l1, (4 * (s / 16) + kq) * 9 + tap, 0, co, 1) : 0.f;
      }
      const int k = 8 * s + 2 * kq + e, tap = k / 8 - r;
      return (tap >= 0 && tap < 9) ? wq(l1, tap, k % 8, co, 8) : 0.f;
    };
    if (fusedf && RCED_T_L1X6 && (blk > 0 || allx6)) {
      // layer 1 on the bf16 pipe (kernels_fused_v3_l23.h, layer1_x6): main pass [chunk c][part][lane] x 8 bf16, row = channel lane & 15,
      // k-slot 8kq + e = (tap 4c + kq, channel e); remainder pass [chunk c][part][lane] x 8 bf16, row i = (phase r = i >> 1, channel
      // 16 + (i & 1)), k-slot = (window tap u = 4c + kq, channel e), frequency tap = u - r.
      // All-x6 form, block 0 (8x9, 1 -> 18): "channel" e = time row e of the kernel (the planes hold x[t + e - 3][f] at entry e).
      auto w1x = [&](int tap, int e, int co) { return blk == 0 ? wq(l1, e * 9 + tap, 0, co, 1) : wq(l1, tap, e, co, 8); };
      unsigned short* d16 = reinterpret_cast<unsigned short*>(dst);
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 8; ++e) {
          const int i = lane & 15, kq = lane >> 4;
          for (int c = 0; c < 3; ++c) {
            const int tap = 4 * c + kq;
            const size_t base = (size_t)(c * 3) * 512 + lane * 8 + e;
            split3(tap < 9 ? w1x(tap, e, i) : 0.f, &d16[base], &d16[base + 512], &d16[base + 1024]);
          }
          for (int c = 0; c < 4; ++c) {
            const int r = i >> 1, co = 16 + (i & 1), tap = 4 * c + kq - r;
            const size_t base = (size_t)(9 + c * 3) * 512 + lane * 8 + e;
            split3((tap >= 0 && tap < 9) ? w1x(tap, e, co) : 0.f, &d16[base], &d16[base + 512], &d16[base + 1024]);
          }
        }
      put_shift(dst + v3::kG1XMain + v3::kG1XRem, 3 * blk + 0);
      dst += v3::kG1X;
    } else if (x6) {
      // register images [j][lane][4 floats]: float 4j + q of a lane = its fragment of K-step 4j + q (block 0) or element
      // (4j + q) & 1 of K-step (4j + q) >> 1 (blocks 1..4)
      for (int lane = 0; lane < 64; ++lane) {
        for (int i = 0; i < 18; ++i)
          dst[(i / 4) * 256 + lane * 4 + i % 4] = blk == 0 ? w1main(lane, i, 0) : w1main(lane, i / 2, i % 2);
        for (int i = 0; i < 32; ++i)
          dst[v3::kG1Main + (i / 4) * 256 + lane * 4 + i % 4] = blk == 0 ? w1rem(lane, i, 0) : w1rem(lane, i / 2, i % 2);
      }
      put_shift(dst + v3::kG1Main + v3::kG1Rem, 3 * blk + 0);
      dst += v3::kG1;
    } else {
      if (blk == 0) {
        for (int s = 0; s < 18; ++s)
          for (int lane = 0; lane < 64; ++lane) dst[s * 64 + lane] = w1main(lane, s, 0);
        for (int s = 0; s < 32; ++s)
          for (int lane = 0; lane < 64; ++lane) dst[v3::kW1Main + s * 64 + lane] = w1rem(lane, s, 0);
      } else {
        for (int s = 0; s < 9; ++s)
          for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 2; ++e) dst[s * 128 + lane * 2 + e] = w1main(lane, s, e);
        for (int s = 0; s < 16; ++s)
          for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 2; ++e) dst[v3::kW1Main + s * 128 + lane * 2 + e] = w1rem(lane, s, e);
      }
      put_shift(dst + v3::kW1Data, 3 * blk + 0);
      dst += v3::kW1;
    }
    // ---- layer 2: 1x5, 18 -> 30
    if (x6) {
      // [chunk][M-tile][part][lane] x 8 bf16; slot k = 32c + 8kq + e of the K axis (kernels_fused_v3.h, "X6 form"):
      //   c < 2: tap 2c + (kq >> 1), channel 8 (kq & 1) + e;  c = 2: kq < 2: tap 4, channel 8kq + e; kq = 2: tap e >> 1,
      //   channel 16 + (e & 1); kq = 3: tap 4, channel 16 + e (e < 2), else zero
      auto w2x = [&](int i, int mt, int c, int kq, int e) {
        const int co = 16 * mt + i;
        int tap, ci;
        if (c < 2) { tap = 2 * c + (kq >> 1); ci = 8 * (kq & 1) + e; }
        else if (kq < 2) { tap = 4; ci = 8 * kq + e; }
        else if (kq == 2) { tap = e >> 1; ci = 16 + (e & 1); }
        else { tap = 4; ci = 16 + e; if (e >= 2) return 0.f; }
        return co < 30 ? wq(l2, tap, ci, co, 18) : 0.f;
      };
      unsigned short* d16 = reinterpret_cast<unsigned short*>(dst);
      for (int c = 0; c < v3::kL2Chunks; ++c)
        for (int mt = 0; mt < 2; ++mt)
          for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 8; ++e) {
              const size_t base = ((size_t)(c * 2 + mt) * 3) * 512 + lane * 8 + e;
              split3(w2x(lane & 15, mt, c, lane >> 4, e), &d16[base], &d16[base + 512], &d16[base + 1024]);
            }
      put_shift(dst + v3::kG2Data, 3 * blk + 1);
      dst += v3::kG2;
    } else {
      // K = 90: 11 b64 steps (k = 8s + 2kq + e = tap*18 + ci) + b32 tail (k = 88 + kq)
      auto w2 = [&](int i, int mt, int k) {
        const int co = 16 * mt + i;
        return (k < 90 && co < 30) ? wq(l2, k / 18, k % 18, co, 18) : 0.f;
      };
      for (int s = 0; s < v3::kL2Steps; ++s)
        for (int mt = 0; mt < 2; ++mt)
          for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 2; ++e)
              dst[(s * 2 + mt) * 128 + lane * 2 + e] = w2(lane & 15, mt, 8 * s + 2 * (lane >> 4) + e);
      for (int mt = 0; mt < 2; ++mt)
        for (int lane = 0; lane < 64; ++lane)
          dst[v3::kL2Steps * 2 * 128 + mt * 64 + lane] = w2(lane & 15, mt, 8 * v3::kL2Steps + (lane >> 4));
      put_shift(dst + v3::kW2Data, 3 * blk + 1);
      dst += v3::kW2;
    }
    // ---- layer 3, fused form: [M-tile j][part][lane] x 8 bf16.  Row i of M-tile j = (cout 2 (i >> 2) + ((i >> 1) & 1), tap 2j + (i & 1));
    //      k-slot 8kq + e = channel (e < 4 ? 4kq + e : 16 + 4kq + e - 4): the registers layer 2's two M-tiles leave in a lane
    if (fusedf) {
      unsigned short* d16 = reinterpret_cast<unsigned short*>(dst);
      for (int j = 0; j < v3::kL3MT; ++j)
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 8; ++e) {
            const int i = lane & 15, kq = lane >> 4, co = 2 * (i >> 2) + ((i >> 1) & 1), tap = 2 * j + (i & 1);
            const int ci = e < 4 ? 4 * kq + e : 16 + 4 * kq + (e - 4);
            const float w = (tap < 9 && ci < 30) ? wq(l3, tap, ci, co, 30) : 0.f;
            const size_t base = ((size_t)j * 3) * 512 + lane * 8 + e;
            split3(w, &d16[base], &d16[base + 512], &d16[base + 1024]);
          }
      put_shift(dst + v3::kW3TData, 3 * blk + 2);
      dst += v3::kW3T;
      continue;
    }
    // ---- layer 3: 1x9, 30 -> 8 on pixel pairs: row i = (phase r, co), k = u*30 + ci, tap = u - r;
    //      37 b64 steps + b32 tail (k = 296 + kq)
    {
      auto w3 = [&](int i, int k) {
        const int r = i >> 3, co = i & 7, u = k / 30, ci = k % 30, tap = u - r;
        return (k < 300 && tap >= 0 && tap < 9) ? wq(l3, tap, ci, co, 30) : 0.f;
      };
      for (int s = 0; s < v3::kL3Steps; ++s)
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 2; ++e) dst[s * 128 + lane * 2 + e] = w3(lane & 15, 8 * s + 2 * (lane >> 4) + e);
      for (int lane = 0; lane < 64; ++lane)
        dst[v3::kL3Steps * 128 + lane] = w3(lane & 15, 8 * v3::kL3Steps + (lane >> 4));
    }
    put_shift(dst + v3::kW3Data, 3 * blk + 2);
    dst += v3::kW3;
  }
}

// decode_final (1x129, 8 -> 1): A fragments of the in-kernel GEMM, [u][lane][e]: row r = lane & 15 (pixel phase),
// k = (window tap u, channel 2*kq + e), value W[u - r][c] (zero outside taps 0..128); then the bin-128 weights
// W[t][c], t = 0..64 (kernels_fused_v3.h, "decode_final inside the kernel")
void pack_v3_final(const rced_model* m, std::vector<float>* fin, float* fin_bias) {
  const rced_layer_dev& lf = m->layers[15];
  fin->assign(v3::kFinPack, 0.f);
  for (int u = 0; u < v3::kFinU; ++u)
    for (int lane = 0; lane < 64; ++lane)
      for (int e = 0; e < 2; ++e) {
        const int r = lane & 15, c = 2 * (lane >> 4) + e, tap = u - r;
        (*fin)[(size_t)u * 128 + lane * 2 + e] = (tap >= 0 && tap < 129) ? wq(lf, tap, c, 0, 8) : 0.f;
      }
  for (int t = 0; t <= 64; ++t)
    for (int c = 0; c < 8; ++c) (*fin)[v3::kFinA + t * 8 + c] = wq(lf, t, c, 0, 8);
  *fin_bias = lf.host_shift[0];
}

// All-x6 form: decode_final's tap table, [part h, m, l][row][8 channels] bf16: row t + 15 = the three-part split of W[tap t][c], zero
// rows for t < 0 and t > 128 (kernels_fused_v3.h, Map<3>: the A fragment of lane (kq, m) in chunk q is row 4q + kq - m + 15)
void pack_v3_fintab(const rced_model* m, std::vector<float>* tab) {
  const rced_layer_dev& lf = m->layers[15];
  tab->assign(v3::kFinTFloats, 0.f);
  unsigned short* d16 = reinterpret_cast<unsigned short*>(tab->data());
  constexpr int kPart16 = v3::kFinTPart / 2;
  for (int t = 0; t < 129; ++t)
    for (int c = 0; c < 8; ++c) {
      const size_t at = (size_t)(t + 15) * 8 + c;
      split3(wq(lf, t, c, 0, 8), &d16[at], &d16[at + kPart16], &d16[at + 2 * kPart16]);
    }
}

int upload(float** dev, const std::vector<float>& host);

// ---- R-CED V1 / V2 (kernels_fused_chain.h) ---------------------------------------------------
template <class N>
void pack_chain(const rced_model* m, std::vector<float>* wpack, std::vector<float>* fin, float* fin_bias) {
  using G = chain::Geo<N>;
  wpack->assign(G::kWTotal, 0.f);
  float* dst = wpack->data();
  for (int l = 0; l < N::kLayers; ++l) {
    const rced_layer_dev& L = m->layers[l];
    const chain::LayerDesc d = N::layer[l];
    const int MT = G::MT(l);
    if (l == 0) {  // 8 x taps x 1: [s = ih*taps + j][lane]; lane = (row i = co, kq), time tap = 4*ih + kq
      for (int s = 0; s < 2 * d.taps; ++s)
        for (int lane = 0; lane < 64; ++lane) {
          const int i = lane & 15, kq = lane >> 4, ih = s / d.taps, j = s % d.taps, ti = 4 * ih + kq;
          dst[s * 64 + lane] = i < d.cout ? wq(L, ti * d.taps + j, 0, i, 1) : 0.f;
        }
    } else {
      const int K = G::K(l), NB = G::NB64(l), NTL = G::NTAIL(l);
      auto wv = [&](int i, int mt, int k) {
        const int co = 16 * mt + i, tap = k / d.cinp, ci = k % d.cinp;
        return (k < K && co < d.cout && ci < d.cin) ? wq(L, tap, ci, co, d.cin) : 0.f;
      };
      for (int s = 0; s < NB; ++s)
        for (int mt = 0; mt < MT; ++mt)
          for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 2; ++e)
              dst[(s * MT + mt) * 128 + lane * 2 + e] = wv(lane & 15, mt, 8 * s + 2 * (lane >> 4) + e);
      for (int j = 0; j < NTL; ++j)
        for (int mt = 0; mt < MT; ++mt)
          for (int lane = 0; lane < 64; ++lane)
            dst[NB * MT * 128 + (j * MT + mt) * 64 + lane] = wv(lane & 15, mt, 8 * NB + 4 * j + (lane >> 4));
      if (G::R(l) > 0) {   // remainder pass: row i = (pixel phase i / R, channel 16 + i % R), k = (window tap u, ci), tap = u - phase
        const int R = G::R(l), P = G::PH(l), KR = G::KR(l), NBR = KR / 8, NTR = (KR % 8 + 3) / 4;
        float* rem = dst + G::main_data(l);
        auto rv = [&](int i, int k) {
          const int u = k / d.cinp, ci = k % d.cinp, tap = u - i / R;
          return (i < P * R && k < KR && ci < d.cin && tap >= 0 && tap < d.taps) ? wq(L, tap, ci, 16 + i % R, d.cin) : 0.f;
        };
        for (int s = 0; s < NBR; ++s)
          for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 2; ++e) rem[s * 128 + lane * 2 + e] = rv(lane & 15, 8 * s + 2 * (lane >> 4) + e);
        for (int j = 0; j < NTR; ++j)
          for (int lane = 0; lane < 64; ++lane) rem[NBR * 128 + j * 64 + lane] = rv(lane & 15, 8 * NBR + 4 * j + (lane >> 4));
        for (int i = 0; i < P * R; ++i) dst[G::data(l) + 32 + i] = L.host_shift[16 + i % R];
      }
    }
    for (int c = 0; c < d.cout; ++c) dst[G::data(l) + c] = L.host_shift[c];
    dst += G::packet(l);
  }
  // final 1x129 layer, CH -> 1, Toeplitz A-fragments: [s][m][lane][e] then tail [m][lane]
  constexpr int CH = N::kFinalCh;
  using FG = chain::FinalGeo<CH>;
  const rced_layer_dev& lf = m->layers[N::kLayers];
  fin->assign(FG::kPack, 0.f);
  auto fv = [&](int i, int mt, int k) {
    const int f = 16 * mt + i, fp = k / CH, ci = k % CH, tap = fp - f + 64;
    return (k < FG::kK && f < 129 && tap >= 0 && tap < 129) ? wq(lf, tap, ci, 0, CH) : 0.f;
  };
  for (int s = 0; s < FG::kNB64; ++s)
    for (int mt = 0; mt < FG::kMT; ++mt)
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 2; ++e)
          (*fin)[((size_t)s * FG::kMT + mt) * 128 + lane * 2 + e] = fv(lane & 15, mt, 8 * s + 2 * (lane >> 4) + e);
  for (int mt = 0; mt < FG::kMT; ++mt)
    for (int lane = 0; lane < 64; ++lane)
      (*fin)[(size_t)FG::kNB64 * FG::kMT * 128 + mt * 64 + lane] = fv(lane & 15, mt, 8 * FG::kNB64 + (lane >> 4));
  *fin_bias = lf.host_shift[0];
}

template <int CH>
void chain_final_layer(const rced_fused* f, float* y, int frames, hipStream_t st) {
  const dim3 grid((frames + chain::kFinFrames - 1) / chain::kFinFrames);
  if (f->final_x6 && f->fin_apack_x6)
    hipLaunchKernelGGL(x6::final_gemm_x6_kernel<CH>, grid, dim3(chain::kFinThreads), 0, st, (const float*)f->h,
                       (const unsigned short*)f->fin_apack_x6, f->fin_bias, y, frames, (const float*)nullptr);
  else if (f->final_lds)
    hipLaunchKernelGGL(chain::final_gemm_lds_kernel<CH>, grid, dim3(chain::kFinThreads), 0, st, (const float*)f->h,
                       (const float*)f->fin_apack, f->fin_bias, y, frames);
  else
    hipLaunchKernelGGL(chain::final_gemm_kernel<CH>, grid, dim3(chain::kFinThreads), 0, st, (const float*)f->h,
                       (const float*)f->fin_apack, f->fin_bias, y, frames);
}

// N0: the net as built (3-frame tiles); N = N0, or its latency form (below)
template <class N0, class N>
int chain_forward_tf(rced_model* m, rced_fused* f, const float* x, float* y, int Nb, int T, hipStream_t st) {
  using G = chain::Geo<N>;
  static_assert(G::kWTotal == chain::Geo<N0>::kWTotal && G::kScratchFloatsPerWg <= chain::Geo<N0>::kScratchFloatsPerWg,
                "the forms of one net share its packets and its skip scratch");
  chain::Params P;
  P.x = x;
  P.h = f->h;
  P.wpack = f->wpack;
  P.scratch = f->scratch;
  P.N = Nb;
  P.T = T;
  P.tiles_per_utt = (T + N::kTF - 1) / N::kTF;
  P.total_tiles = Nb * P.tiles_per_utt;
  const int cus = f->grid_limit > 0 ? std::min(f->grid_limit, m->num_cus) : m->num_cus;   // scratch is sized for #CUs
  const int grid = std::min(P.total_tiles, cus);
  m->prof_begin(RCED_K_FUSED, st);
  hipLaunchKernelGGL(chain::fused_chain_kernel<N>, dim3(grid), dim3(chain::kThreads), G::kLdsBytes, st, P);
  m->prof_end(RCED_K_FUSED, st);
  HIP_TRY(hipGetLastError());
  const int frames = Nb * T;
  m->prof_begin(RCED_K_FINAL, st);
  chain_final_layer<N::kFinalCh>(f, y, frames, st);
  m->prof_end(RCED_K_FINAL, st);
  HIP_TRY(hipGetLastError());
  return RCED_OK;
}
// Latency form: a call with fewer 3-frame tiles than the part has CUs (BASELINE config 1: one utterance of 256 frames = 86 tiles)
// leaves CUs idle while every busy one walks a tile's layers at a third of its rate; with ONE-frame tiles the same call is 3 x
// the workgroups of a third of the work each.  Same packets, same arithmetic per pixel (the remainder pass's columns are frame-aligned:
// chain::Geo::CPF): the results are bit-identical (test).
constexpr int kLatencyTF = 1;
template <class N>
int chain_forward(rced_model* m, rced_fused* f, const float* x, float* y, int Nb, int T, hipStream_t st) {
  const int cus = f->grid_limit > 0 ? std::min(f->grid_limit, m->num_cus) : m->num_cus;
  if (f->latency_form && Nb * ((T + N::kTF - 1) / N::kTF) < cus)
    return chain_forward_tf<N, chain::WithTF<N, kLatencyTF>>(m, f, x, y, Nb, T, st);
  return chain_forward_tf<N, N>(m, f, x, y, Nb, T, st);
}

template <class N>
int chain_create(rced_model* m, rced_fused* f) {
  using G = chain::Geo<N>;
  std::vector<float> wpack, fin;
  pack_chain<N>(m, &wpack, &fin, &f->fin_bias);
  int rc = upload(&f->wpack, wpack);
  if (!rc) rc = upload(&f->fin_apack, fin);
  if (rc) return rc;
  {  // the output layer's A[f, k] once more, every value as three bf16 parts in K-32 fragment order (kernels_final_x6.h)
    constexpr int CH = N::kFinalCh;
    using X = x6::FinalX6<CH>;
    const rced_layer_dev& lf = m->layers[N::kLayers];
    std::vector<unsigned short> p(X::kPackShorts, 0);
    auto b2f = [](unsigned short b) { const unsigned u = (unsigned)b << 16; float v; memcpy(&v, &u, 4); return v; };
    for (int S = 0; S < X::kSteps; ++S)
      for (int mt = 0; mt < X::kMT; ++mt)
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 8; ++e) {
            const int k = 32 * S + 8 * (lane >> 4) + e, fo = 16 * mt + (lane & 15), fp = k / CH, ci = k % CH, tap = fp - fo + 64;
            const float v = (k < X::kK && fo < 129 && tap >= 0 && tap < 129) ? wq(lf, tap, ci, 0, CH) : 0.f;
            const unsigned short h = bf16_rne(v);
            const float r1 = v - b2f(h);
            const unsigned short mm = bf16_rne(r1);
            const unsigned short l = bf16_rne(r1 - b2f(mm));
            const size_t base = ((size_t)(S * X::kMT + mt) * 3) * 512 + lane * 8 + e;
            p[base] = h;
            p[base + 512] = mm;
            p[base + 1024] = l;
          }
    HIP_TRY(hipMalloc(&f->fin_apack_x6, p.size() * sizeof(unsigned short)));
    HIP_TRY(hipMemcpy(f->fin_apack_x6, p.data(), p.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
  }
  f->scratch_bytes = (size_t)m->num_cus * G::kScratchFloatsPerWg * sizeof(float);
  HIP_TRY(hipMalloc(&f->scratch, f->scratch_bytes));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chain::fused_chain_kernel<N>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, G::kLdsBytes));
  using NL = chain::WithTF<N, kLatencyTF>;
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chain::fused_chain_kernel<NL>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, chain::Geo<NL>::kLdsBytes));
  return RCED_OK;
}

// ---- bf16 variant (kernels_frame16.h) ----------------------------------------------------------
// Packets: [K-step s][M-tile mt][lane] x 8 bf16 -- lane (kq, m): row 16 mt + m (cout), K slot j = 4 s + kq = (tap j / OCT, octet
// j % OCT), element e = channel 8 (j % OCT) + e (first layer: OCT = 1, e = the kernel's time row); behind all packets 32 fp32
// shifts per layer.
template <class N>
void pack_frame16(const rced_model* m, std::vector<unsigned>* wpack) {
  using G = frame16::Geo<N>;
  wpack->assign(G::kWBytes / 4, 0u);
  for (int l = 0; l < N::kLayers; ++l) {
    const rced_layer_dev& L = m->layers[l];
    const chain::LayerDesc d = N::layer[l];
    const int MT = G::MT(l), OCT = G::oct_in(l);
    unsigned char* pk = reinterpret_cast<unsigned char*>(wpack->data()) + G::packet_off(l);
    unsigned short* d16 = reinterpret_cast<unsigned short*>(pk);
    for (int s = 0; s < G::steps(l); ++s)
      for (int mt = 0; mt < MT; ++mt)
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 8; ++e) {
            const int j = 4 * s + (lane >> 4), tap = j / OCT, ch = 8 * (j % OCT) + e, co = 16 * mt + (lane & 15);
            float v = 0.f;
            if (tap < d.taps && co < d.cout) {
              if (l == 0) v = wq(L, e * d.taps + tap, 0, co, 1);            // [kh = 8 time rows][kw = taps][1][cout]
              else if (ch < d.cin) v = wq(L, tap, ch, co, d.cin);
            }
            d16[((size_t)(s * MT + mt) * 64 + lane) * 8 + e] = bf16_rne(v);
          }
    float* sh = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(wpack->data()) + G::kShiftOff) + 32 * l;
    for (int c = 0; c < d.cout; ++c) sh[c] = L.host_shift[c];
  }
  // the output layer's tap tables (frame16::Geo's packet comment): TA[kq][r] = tap r + kq - 15 of channels 0 .. 7;
  // TB[c][k][j] = tap RB k + c + j - 15 of channels 8 .. 8 + CB - 1; zero outside taps 0 .. 128
  constexpr int CH = N::kFinalCh, RB = G::kFinRB, CB = G::kFinCB;
  const rced_layer_dev& lf = m->layers[N::kLayers];
  unsigned char* tt = reinterpret_cast<unsigned char*>(wpack->data()) + G::packet_off(N::kLayers);
  auto tapw = [&](int tap, int c) { return bf16_rne((tap >= 0 && tap < 129 && c < CH) ? wq(lf, tap, c, 0, CH) : 0.f); };
  for (int kq = 0; kq < 4; ++kq)
    for (int r = 0; r < G::kTARows; ++r)
      for (int c = 0; c < 8; ++c)
        reinterpret_cast<unsigned short*>(tt + kq * G::kTACopy)[r * 8 + c] = tapw(r + kq - 15, c);
  for (int c = 0; c < RB; ++c)
    for (int k = 0; k < G::kTBRows; ++k)
      for (int j = 0; j < RB; ++j)
        for (int e = 0; e < CB; ++e)
          reinterpret_cast<unsigned short*>(tt + G::kTBOff + c * G::kTBCopy)[(k * RB + j) * CB + e] = tapw(RB * k + c + j - 15, 8 + e);
}
template <class N, int W>
int frame16_prepare(rced_model* m, int* per_cu) {
  using G = frame16::Geo<N, W>;
  const void* k = reinterpret_cast<const void*>(frame16::frame16_kernel<N, W>);
  HIP_TRY(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, G::kLdsBytes));
  int n = 1;   // resident workgroups per CU with this LDS footprint and the kernel's register count
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, G::kThreads, G::kLdsBytes) != hipSuccess || n < 1) n = 1;
  *per_cu = std::min(n, W == 4 ? 2 : 1);
  return RCED_OK;
}
template <class N>
int frame16_enable(rced_model* m, rced_fused* f) {
  using G = frame16::Geo<N>;   // packets and the skip scratch do not depend on the frames per workgroup
  if (f->wpack16) return RCED_OK;
  std::vector<unsigned> wpack;
  pack_frame16<N>(m, &wpack);
  unsigned* wdev = nullptr;
  HIP_TRY(hipMalloc(&wdev, wpack.size() * sizeof(unsigned)));
  if (hipMemcpy(wdev, wpack.data(), wpack.size() * sizeof(unsigned), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(wdev);
    return rced_fail(RCED_ERR_HIP, "hipMemcpy(bf16 packets)");
  }
  if (int rc = frame16_prepare<N, 4>(m, &f->bf16_wgs_per_cu)) { (void)hipFree(wdev); return rc; }
  int one = 1;
  if (int rc = frame16_prepare<N, 8>(m, &one)) { (void)hipFree(wdev); return rc; }
  static_assert(G::kScratchBytesPerWave == frame16::Geo<N, 8>::kScratchBytesPerWave, "one scratch serves both forms");
  if (!f->scratch16 && hipMalloc(&f->scratch16, (size_t)8 * m->num_cus * G::kScratchBytesPerWave) != hipSuccess) {   // 8 waves per CU either way
    (void)hipFree(wdev);
    return rced_fail(RCED_ERR_HIP, "hipMalloc(bf16 skip scratch)");
  }
  f->wpack16 = wdev;   // published last: the mode counts as built only when everything above succeeded
  return RCED_OK;
}
template <class N, int W>
int frame16_launch(rced_model* m, rced_fused* f, const float* x, float* y, int Nb, int T, hipStream_t st) {
  using G = frame16::Geo<N, W>;
  frame16::Params P;
  P.x = x;
  P.y = y;
  P.fin_bias = f->fin_bias;
  P.wpack = f->wpack16;
  P.scratch = f->scratch16;
  P.N = Nb;
  P.T = T;
  P.tiles_per_utt = (T + W - 1) / W;
  P.total_tiles = Nb * P.tiles_per_utt;
  P.stamps = nullptr;
#if RCED_F16_STAMPS
  if (!f->stamps && hipMalloc(&f->stamps, 216 * sizeof(unsigned long long)) != hipSuccess) f->stamps = nullptr;
  P.stamps = f->stamps;
#endif
  const int wgs = (W == 4 ? f->bf16_wgs_per_cu : 1) * m->num_cus;   // four waves: LDS (<= 80 KB) allows two per CU, VGPRs decide (frame16_enable)
  const int grid = std::min(P.total_tiles, f->grid_limit > 0 ? std::min(f->grid_limit, wgs) : wgs);
  m->prof_begin(RCED_K_FUSED, st);   // all 16 / 10 layers: the output layer is the kernel's last phase
  hipLaunchKernelGGL((frame16::frame16_kernel<N, W>), dim3(grid), dim3(G::kThreads), G::kLdsBytes, st, P);
  m->prof_end(RCED_K_FUSED, st);
  HIP_TRY(hipGetLastError());
  return RCED_OK;
}
// Frames per workgroup, per call (option "bf16_frames": 0 = this rule, 4 / 8 = always that form; results are bit-identical either way --
// a frame's arithmetic does not depend on it).  Eight (one workgroup per CU, every packet feeds eight frames, one skip in LDS) is 2 - 5 %
// faster once every CU has two full tiles; four (two workgroups per CU) starts twice as many workgroups on a small call and leaves a
// lone tile one wave per SIMD: [1, 256] 23 us against 32 us.
template <class N>
int frame16_forward(rced_model* m, rced_fused* f, const float* x, float* y, int Nb, int T, hipStream_t st) {
  const long long frames = (long long)Nb * T;
  const int w = f->bf16_frames ? f->bf16_frames : frames >= 16LL * m->num_cus ? 8 : 4;
  return w == 8 ? frame16_launch<N, 8>(m, f, x, y, Nb, T, st) : frame16_launch<N, 4>(m, f, x, y, Nb, T, st);
}

int upload(float** dev, const std::vector<float>& host) {   // *dev is set only when the copy succeeded
  float* p = nullptr;
  HIP_TRY(hipMalloc(&p, host.size() * sizeof(float)));
  const hipError_t e = hipMemcpy(p, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(p);
    return rced_fail(RCED_ERR_HIP, "hipMemcpy(weights): %s", hipGetErrorString(e));
  }
  *dev = p;
  return RCED_OK;
}

}  // namespace

// tools/summarize_prof.py prints these next to a profile (the kernel-trace CSV reports 0 for dynamic LDS)
static_assert(v3::MapA::kLdsBytes == 159024 && v3::MapF32::kLdsBytes == 163024, "update DYNAMIC_LDS in tools/summarize_prof.py");
#if RCED_V3_LEGACY_FORMS
static_assert(v3::MapT::kLdsBytes == 163792 && v3::MapX6::kLdsBytes == 162976, "update DYNAMIC_LDS in tools/summarize_prof.py");
#endif
// forms of the CR-CED kernel this build holds: 3 (the product) and 0 (the bit-exact fp32-MFMA comparator); 1 and 2 in legacy builds only
inline bool v3_form_built(int form) { return form == 0 || form == 3 || (RCED_V3_LEGACY_FORMS && (form == 1 || form == 2)); }
template <class M>
int v3_set_lds() {
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(v3::fused_v3_kernel<M>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, M::kLdsBytes);
  if (e != hipSuccess) return rced_fail(RCED_ERR_HIP, "hipFuncSetAttribute(LDS %d): %s", M::kLdsBytes, hipGetErrorString(e));
  return RCED_OK;
}
// option "v3_l2x6": a form's weight stream is built when it is first selected
// (a form counts as built only when its upload AND its LDS attribute succeeded -- the device pointer is published last)
int v3_enable_form(rced_model* m, rced_fused* f, int form) {
  if (!v3_form_built(form)) return rced_fail(RCED_ERR_ARG, "v3_l2x6 takes 3 (the product: every layer at fp32 quality on the bf16 matrix pipe) or 0 (every "
                                                         "layer on the fp32 MFMA, the bit-exact comparator), got %d", form);
  float** dev = form == 0 ? &f->wpack : form == 1 ? &f->wpack_x6 : form == 2 ? &f->wpack_t : &f->wpack_a;
  if (*dev) return RCED_OK;
  std::vector<float> wpack;
  pack_v3(m, form, &wpack);
  float* fresh = nullptr;
  int rc = upload(&fresh, wpack);
  if (!rc && form == 3 && !f->fin_tab) {
    std::vector<float> tab;
    pack_v3_fintab(m, &tab);
    rc = upload(&f->fin_tab, tab);
    if (rc && f->fin_tab) {
      (void)hipFree(f->fin_tab);
      f->fin_tab = nullptr;
    }
  }
#if RCED_V3_LEGACY_FORMS
  if (!rc && (form == 1 || form == 2)) rc = form == 1 ? v3_set_lds<v3::MapX6>() : v3_set_lds<v3::MapT>();
#endif
  if (!rc && (form == 0 || form == 3)) rc = form == 0 ? v3_set_lds<v3::MapF32>() : v3_set_lds<v3::MapA>();
  if (rc) {
    if (fresh) (void)hipFree(fresh);
    return rc;
  }
  *dev = fresh;
  return RCED_OK;
}

int fused_create(rced_model* m) {
  m->fused = nullptr;
  rced_fused* f = new rced_fused();
  f->final_x6 = env_default("RCED_FINAL_X6", 1) != 0;
  f->final_lds = env_default("RCED_FINAL_LDS", 1) != 0;
  if (m->variant != RCED_V3) {
    m->fused = f;
    const int rc = m->variant == RCED_V1 ? chain_create<chain::NetV1>(m, f) : chain_create<chain::NetV2>(m, f);
    if (rc) fused_destroy(m);
    return rc;
  }
  std::vector<float> fin;
  pack_v3_final(m, &fin, &f->fin_bias);
  int rc = upload(&f->fin_apack, fin);
  if (!rc) {   // the environment only supplies the DEFAULT of the per-handle option
    const int form = env_default("RCED_V3_L2X6", 3);
    f->v3_l2x6 = v3_form_built(form) ? form : 3;
    rc = v3_enable_form(m, f, f->v3_l2x6);
  }
  if (!rc) {
    hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&f->err_host), 64, hipHostMallocMapped);
    if (e == hipSuccess) {
      *f->err_host = 0u;
      e = hipHostGetDevicePointer(reinterpret_cast<void**>(&f->err_dev), f->err_host, 0);
    }
    if (e != hipSuccess) rc = rced_fail(RCED_ERR_HIP, "error word (pinned host memory): %s", hipGetErrorString(e));
  }
#if RCED_STAMPS
  if (!rc && hipMalloc(&f->stamps, (64 + 24 + 128) * sizeof(unsigned long long)) != hipSuccess) f->stamps = nullptr;
#endif
  m->fused = f;
  if (rc) {
    fused_destroy(m);
    return rc;
  }
  return RCED_OK;
}

void fused_destroy(rced_model* m) {
  rced_fused* f = m->fused;
  if (!f) return;
  if (f->wpack) (void)hipFree(f->wpack);
  if (f->wpack_x6) (void)hipFree(f->wpack_x6);
  if (f->wpack_t) (void)hipFree(f->wpack_t);
  if (f->wpack_a) (void)hipFree(f->wpack_a);
  if (f->fin_tab) (void)hipFree(f->fin_tab);
  if (f->wpack16) (void)hipFree(f->wpack16);
  if (f->fin_apack_x6) (void)hipFree(f->fin_apack_x6);
  if (f->scratch16) (void)hipFree(f->scratch16);
  if (f->fin_apack) (void)hipFree(f->fin_apack);
  if (f->h) (void)hipFree(f->h);
  if (f->scratch) (void)hipFree(f->scratch);
  if (f->err_host) (void)hipHostFree(f->err_host);
  delete f;
  m->fused = nullptr;
}

int fused_reserve(rced_model* m, int N, int T) {
  rced_fused* f = m->fused;
  if (m->variant == RCED_V3) return RCED_OK;   // decode_final runs inside the fused kernel: no hand-off tensor
  const int ch = m->variant == RCED_V1 ? chain::NetV1::kFinalCh : chain::NetV2::kFinalCh;
  const size_t need = (size_t)N * T * v3::kF * ch * sizeof(float);
  if (need <= f->h_bytes) return RCED_OK;
  if (f->h) {
    HIP_TRY(hipDeviceSynchronize());
    (void)hipFree(f->h);
    f->h = nullptr;
    f->h_bytes = 0;
  }
  HIP_TRY(hipMalloc(&f->h, need));
  f->h_bytes = need;
  return RCED_OK;
}

// The CR-CED kernel's split-tile hand-offs wait on LDS flag words with a bounded spin; a flag that never arrives is
// recorded in a sticky word instead of hanging the GPU (kernels_fused_v3.h flag_wait).  The results of that launch are
// wrong, so the model refuses further work: checked before every launch and after every synchronising entry point.
int fused_check(rced_model* m) {
  rced_fused* f = m->fused;
  if (!f || !f->err_host) return RCED_OK;
  const unsigned e = *reinterpret_cast<volatile unsigned*>(f->err_host);
  if (e == 0u) return RCED_OK;
  return rced_fail(RCED_ERR_STATE, "fused CR-CED kernel: a wave-to-wave hand-off timed out (code %u); results of that "
                                   "launch are invalid -- destroy and recreate the model", e);
}

int fused_forward(rced_model* m, const float* x, float* y, int N, int T, hipStream_t st) {
  rced_fused* f = m->fused;
  if (int rc = fused_reserve(m, N, T)) return rc;
  if (m->variant == RCED_V1)
    return f->bf16 ? frame16_forward<chain::NetV1>(m, f, x, y, N, T, st)
                   : chain_forward<chain::NetV1>(m, f, x, y, N, T, st);
  if (m->variant == RCED_V2)
    return f->bf16 ? frame16_forward<chain::NetV2>(m, f, x, y, N, T, st)
                   : chain_forward<chain::NetV2>(m, f, x, y, N, T, st);
  if (int rc = fused_check(m)) return rc;   // a hand-off flag that never came in an EARLIER launch: refuse to go on
  v3::Params P;
  P.err = f->err_dev;
  P.x = x;
  P.y = y;
  P.wpack = f->v3_l2x6 == 3 ? f->wpack_a : f->v3_l2x6 == 2 ? f->wpack_t : f->v3_l2x6 ? f->wpack_x6 : f->wpack;
  P.fin = f->v3_l2x6 == 3 ? f->fin_tab : f->fin_apack;
  P.fin_bias = f->fin_bias;
  P.N = N;
  P.T = T;
  P.tiles_per_utt = (T + v3::kTF - 1) / v3::kTF;
  P.total_tiles = N * P.tiles_per_utt;
  P.stamps = f->stamps;
  const int cus = f->grid_limit > 0 ? f->grid_limit : m->num_cus;
  const int grid = std::min(P.total_tiles, cus);
  m->prof_begin(RCED_K_FUSED, st);   // all 16 layers: decode_final is the kernel's last phase
  if (f->v3_l2x6 == 3) hipLaunchKernelGGL(v3::fused_v3_kernel<v3::MapA>, dim3(grid), dim3(v3::kThreads), v3::MapA::kLdsBytes, st, P);
#if RCED_V3_LEGACY_FORMS
  else if (f->v3_l2x6 == 2) hipLaunchKernelGGL(v3::fused_v3_kernel<v3::MapT>, dim3(grid), dim3(v3::kThreads), v3::MapT::kLdsBytes, st, P);
  else if (f->v3_l2x6 == 1) hipLaunchKernelGGL(v3::fused_v3_kernel<v3::MapX6>, dim3(grid), dim3(v3::kThreads), v3::MapX6::kLdsBytes, st, P);
#endif
  else hipLaunchKernelGGL(v3::fused_v3_kernel<v3::MapF32>, dim3(grid), dim3(v3::kThreads), v3::MapF32::kLdsBytes, st, P);
  m->prof_end(RCED_K_FUSED, st);
  HIP_TRY(hipGetLastError());
  return RCED_OK;
}

// Returns RCED_OPT_UNKNOWN (rced_internal.h) for a key that is not an option of the fused runtime; for a key that is, RCED_OK or a failure
// whose message rced_fail() has set -- rced_set_option overwrites the message only in the first case (the specific refusals,
// "v3_l2x6 selects the form of the CR-CED kernel only" ..., used to be replaced by the generic "unknown option" text).
int fused_set_option(rced_model* m, const char* key, int value) {
  if (!m->fused) return RCED_OPT_UNKNOWN;
  if (!strcmp(key, "fused_grid")) {
    if (value < 0) return rced_fail(RCED_ERR_ARG, "fused_grid must be >= 0 (0 = one workgroup per CU), got %d", value);
    m->fused->grid_limit = value;
    return RCED_OK;
  }
  for (int i = 0; i < 2; ++i) {   // the R-CED output layer's kernel selection (fp32 mode; the bf16 mode's runs inside its kernel)
    static const char* const names[2] = {"final_x6", "final_lds"};
    if (strcmp(key, names[i])) continue;
    if (m->variant == RCED_V3) return rced_fail(RCED_ERR_ARG, "%s selects the R-CED V1 / V2 output-layer kernel; CR-CED's runs inside its fused kernel", key);
    if (value != 0 && value != 1) return rced_fail(RCED_ERR_ARG, "%s takes 0 or 1, got %d", key, value);
    (i == 0 ? m->fused->final_x6 : m->fused->final_lds) = value;
    return RCED_OK;
  }
  if (!strcmp(key, "latency_form")) {
    if (m->variant == RCED_V3) return rced_fail(RCED_ERR_ARG, "latency_form selects the R-CED V1 / V2 kernel's one-frame tiles; CR-CED has one form of tile");
    if (value != 0 && value != 1) return rced_fail(RCED_ERR_ARG, "latency_form takes 0 or 1, got %d", value);
    m->fused->latency_form = value;
    return RCED_OK;
  }
  if (!strcmp(key, "v3_l2x6")) {
    if (m->variant != RCED_V3) return rced_fail(RCED_ERR_ARG, "v3_l2x6 selects the form of the CR-CED kernel only");
    if (int rc = v3_enable_form(m, m->fused, value)) return rc;
    m->fused->v3_l2x6 = value;
    return RCED_OK;
  }
  if (!strcmp(key, "bf16")) {
    if (m->variant == RCED_V3) return value ? rced_fail(RCED_ERR_ARG, "bf16 is built for R-CED V1 / V2 only") : RCED_OK;
    if (value) {
      if (int rc = m->variant == RCED_V1 ? frame16_enable<chain::NetV1>(m, m->fused) : frame16_enable<chain::NetV2>(m, m->fused))
        return rc;
    }
    m->fused->bf16 = value != 0;
    return RCED_OK;
  }
  if (!strcmp(key, "bf16_frames")) {
    if (m->variant == RCED_V3) return rced_fail(RCED_ERR_ARG, "bf16_frames selects the form of the bf16 R-CED V1 / V2 kernel only");
    if (value != 0 && value != 4 && value != 8) return rced_fail(RCED_ERR_ARG, "bf16_frames takes 0 (chosen per call), 4 or 8, got %d", value);
    m->fused->bf16_frames = value;
    return RCED_OK;
  }
  if (!strcmp(key, "inject_handoff_error")) {   // test hook: write `value` into the sticky error word as the kernel would
    if (!m->fused->err_host) return rced_fail(RCED_ERR_ARG, "this variant's fused kernel has no hand-off error word");
    *reinterpret_cast<volatile unsigned*>(m->fused->err_host) = (unsigned)value;
    return RCED_OK;
  }
  return RCED_OPT_UNKNOWN;
}

int fused_get_option(rced_model* m, const char* key, int* value) {
  if (!m->fused) return RCED_ERR_ARG;
#if RCED_F16_STAMPS
  if (!strncmp(key, "f16stamp", 8) && m->fused->stamps) {  // "f16stampNN": cycles since stamp 0 (kernels_frame16.h)
    unsigned long long h[216];
    if (hipMemcpy(h, m->fused->stamps, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return RCED_ERR_HIP;
    const int i = atoi(key + 8);
    if (i < 0 || i >= 216) return RCED_ERR_ARG;
    *value = (int)(h[i] - h[0]);
    return RCED_OK;
  }
#endif
#if RCED_STAMPS
  if (!strncmp(key, "stamp", 5) && m->fused->stamps) {  // "stampNN": kilo-cycles, NN = wave*8 + slot
    unsigned long long h[216];
    if (hipMemcpy(h, m->fused->stamps, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return RCED_ERR_HIP;
    const int i = atoi(key + 5);
    if (i < 0 || i >= 216) return RCED_ERR_ARG;
    *value = (int)(h[i] / 1000);
    return RCED_OK;
  }
#endif
  if (!strcmp(key, "fused_grid")) {
    *value = m->fused->grid_limit;
    return RCED_OK;
  }
  if (!strcmp(key, "bf16_frames")) {
    *value = m->fused->bf16_frames;
    return RCED_OK;
  }
  if (!strcmp(key, "fused_final")) {   // 1: the 1x129 output layer runs inside the fused kernel (no hand-off tensor in HBM)
    *value = m->variant == RCED_V3 || m->fused->bf16;
    return RCED_OK;
  }
  if (!strcmp(key, "v3_l2x6")) {
    *value = m->variant == RCED_V3 ? m->fused->v3_l2x6 : 0;
    return RCED_OK;
  }
  if (!strcmp(key, "final_x6") || !strcmp(key, "final_lds")) {
    *value = !strcmp(key, "final_x6") ? m->fused->final_x6 : m->fused->final_lds;
    return RCED_OK;
  }
  if (!strcmp(key, "bf16")) {
    *value = m->fused->bf16;
    return RCED_OK;
  }
  if (!strcmp(key, "latency_form")) {
    *value = m->variant == RCED_V3 ? 0 : m->fused->latency_form;
    return RCED_OK;
  }
  return RCED_ERR_ARG;
}
