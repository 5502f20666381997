// C ABI of the training step (include/rced.h, "training" section): host-side orchestration.
// Replaces FullyCNNTrainer.creat_graph + train_step (model_utils/trainer.py:156-192): forward with
// train-mode BatchNorm, loss = sum((target - pred)^2) / batch_size, backward, TF-form Adam, moving
// statistics update.  Layer by layer; see kernels_train.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/rced.h"
#include "kernels_generic.h"
#include "kernels_train.h"
#include "kernels_train_mfma.h"
#include "kernels_final_x6.h"
#include "train_mfma_dispatch.h"
#include "rced_internal.h"
#include "rced_spec.h"

using namespace rced;

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return rced_fail(e_ == hipErrorOutOfMemory ? RCED_ERR_ALLOC : RCED_ERR_HIP, "%s: %s", #expr, \
                       hipGetErrorString(e_));                                                 \
  } while (0)

namespace {
struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) { prev = -1; return; }
    ok = (prev == dev) || (hipSetDevice(dev) == hipSuccess);
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
constexpr int kReduceGrid = 512;
using rced::tmd::kPairGrid;
using rced::tmd::allow_lds;
using rced::tmd::tm_packet_floats;
constexpr float kAdamB1 = 0.9f, kAdamB2 = 0.999f, kAdamEps = 1e-8f;   // tf.train.AdamOptimizer defaults
constexpr float kBnMomentum = 0.99f;                                  // tf.layers.batch_normalization default

struct LayerOff {   // float offsets into the variable blob
  size_t kernel, bias, gamma, beta, mmean, mvar;
  int cin, cout, cout4, cin4, K;
};
}  // namespace

struct rced_trainer {
  int variant = 0, device = 0, batch_size = 1, num_cus = 256;
  // The MFMA kernels stage channel pairs, so odd channel counts (R-CED V2: 15, 19, 21, 23, 25) are trained in an
  // even-padded internal layout: `inet` = the net with every hidden cout rounded up to even.  A phantom channel has
  // zero kernel / bias / gamma / beta, so it is 0 after conv, BatchNorm and ReLU, contributes nothing downstream,
  // receives zero gradients and is not trainable; variables cross the ABI in the reference's own (unpadded) layout.
  const NetSpec* net = nullptr;   // = &inet
  NetSpec inet{};
  size_t nvars = 0;               // floats in the internal blob
  size_t xvars = 0;               // floats in the external (reference) blob
  std::vector<int> x2i;           // external float index -> internal float index
  std::vector<LayerOff> off;
  float *params = nullptr, *grads = nullptr, *m = nullptr, *v = nullptr;
  unsigned char* trainable = nullptr;
  std::vector<float*> wf, wt, bias4, mu, rstd;   // per layer
  std::vector<float*> pk_fwd, pk_bwd;            // per layer: MFMA A-fragment packets (1xk layers with an MFMA kernel)
  int use_mfma = 1;
  bool fuse_dz = true;         // RCED_TRAIN_FUSE_DZ=0: always materialise dz with bn_bwd_apply
  bool fuse_bwd = true;        // RCED_TRAIN_FUSE_BWD=0: separate wgrad and dgrad kernels everywhere
  bool fuse_sums = true;       // RCED_TRAIN_FUSE_SUMS=0: BatchNorm-backward sums of plain layers from bwd_route2, not from the dgrad
  bool det = true;             // RCED_TRAIN_DET=0: weight gradients by fp32 atomics (the round-1 behaviour; not reproducible bit for bit)
  float* wpart = nullptr;      // per-wave slices of the wgrad kernels' partial sums (deterministic mode; tmd::WgDet)
  size_t wpart_floats = 0;
  int wg_error = 0;            // tmd::wg_launch could not grow wpart: the step fails instead of using atomics
  std::vector<char> virt;      // virt[id]: tensor id (= relu(bn(z[id-1]))) is never materialised; its consumer rebuilds it
  float* pk_first = nullptr;   // A fragments of the 8xk first layer (rebuilt every step)
  std::vector<float*> pk_fwd_x6;   // the forward packets of the layers that run in the three-part bf16 form (tmm::conv_x6_fwd)
  std::vector<float*> pk_bwd_x6;   // the dgrad packets of the fused backward kernels whose dgrad half runs in that form (tmm::bwd_x6)
  bool use_x6 = true;          // RCED_TRAIN_X6=0: every convolution on the fp32 MFMA (forward 18 -> 30 layers, the output layer's forward and dgrad)
  float* pk_fin = nullptr, *pk_fin_bwd = nullptr;     // Toeplitz A fragments of the 1x129 output layer (rebuilt every step)
  float* zero32 = nullptr;
  double *part = nullptr, *sums = nullptr;
  int* redo = nullptr;         // device flag behind `sums` (same allocation): sums_fix_x asks for the exact recomputation
  int* tiny_host = nullptr;    // [layers] pinned, host-mapped: 1 = some |gamma| of the layer is below kTinyGamma (written by
  int* tiny_dev = nullptr;     // tiny_gamma_scan behind every Adam step, read by the host at the start of the next step)
  // activations for P pixels
  size_t cap_px = 0;
  std::vector<float*> out, z, G;   // out/G indexed by tensor id (0 unused), z by layer
  float* D = nullptr;
  long long global_step = 0;
  ~rced_trainer() {
    DeviceGuard g(device);
    auto fr = [](void* p) { if (p) (void)hipFree(p); };
    fr(params); fr(grads); fr(m); fr(v); fr(trainable); fr(zero32); fr(part); fr(sums); fr(D); fr(wpart);
    if (tiny_host) (void)hipHostFree(tiny_host);
    for (auto* p : wf) fr(p);
    for (auto* p : wt) fr(p);
    for (auto* p : bias4) fr(p);
    for (auto* p : mu) fr(p);
    for (auto* p : rstd) fr(p);
    for (auto* p : pk_fwd) fr(p);
    for (auto* p : pk_bwd) fr(p);
    for (auto* p : pk_fwd_x6) fr(p);
    for (auto* p : pk_bwd_x6) fr(p);
    fr(pk_fin);
    fr(pk_fin_bwd);
    fr(pk_first);
    free_acts();
  }
  std::vector<void*> act_bases;   // the allocations behind out / z / G (those pointers may sit at a skew inside them)
  void free_acts() {
    for (void* p : act_bases) (void)hipFree(p);
    act_bases.clear();
    out.clear(); z.clear(); G.clear();
    cap_px = 0;
  }
};

namespace {

int ensure_acts(rced_trainer* t, size_t P) {
  if (P <= t->cap_px) return RCED_OK;
  HIP_TRY(hipDeviceSynchronize());
  t->free_acts();
  if (t->D) { (void)hipFree(t->D); t->D = nullptr; }
  const NetSpec& net = *t->net;
  const int L = net.n_layers;
  t->out.assign(L + 1, nullptr);
  t->G.assign(L + 1, nullptr);
  t->z.assign(L, nullptr);
  int maxc = 1;
  // (Round 3 measured a different start offset inside its allocation for every tensor -- so that the tensors a kernel streams
  // in lockstep do not walk the same HBM channel sequence -- as run-to-run noise; the switch is gone.)
  auto alloc = [&](float** out_ptr, size_t bytes) -> int {
    void* base = nullptr;
    HIP_TRY(hipMalloc(&base, bytes));
    t->act_bases.push_back(base);
    *out_ptr = reinterpret_cast<float*>(base);
    return RCED_OK;
  };
  for (int l = 0; l < L; ++l) {
    const int c = net.layer[l].cout;
    maxc = std::max(maxc, c);
    if (int rc = alloc(&t->z[l], P * c * sizeof(float))) return rc;
    if (!t->virt.empty() && t->virt[l + 1])
      t->out[l + 1] = nullptr;   // rebuilt from z by its consumer (BnReluXform)
    else if (net.layer[l].use_norm || net.layer[l].use_act || net.layer[l].skip_pre >= 0 || net.layer[l].skip_post >= 0) {
      if (int rc = alloc(&t->out[l + 1], P * c * sizeof(float))) return rc;
    } else
      t->out[l + 1] = t->z[l];   // plain conv layer (decode_final): its output IS z
    if (int rc = alloc(&t->G[l + 1], P * c * sizeof(float))) return rc;
  }
  HIP_TRY(hipMalloc(&t->D, P * maxc * sizeof(float)));
  t->cap_px = P;
  return RCED_OK;
}

int launch_conv(const float* x, float* y, const float* w, const float* shift, const float* skip, int frames, int T, int F,
                int cin, int cout, int cout4, int kh, int kw, int pt, int pl, hipStream_t st) {
  const size_t row = (size_t)kh * (F + kw - 1) * cin * sizeof(float);
  if (row > 64 * 1024) return rced_fail(RCED_ERR_ARG, "layer needs %zu B of LDS", row);
  const int fpw = generic_frames_per_wg(F, cout4, kh, row);
  hipLaunchKernelGGL(conv_layer_generic, dim3((frames + fpw - 1) / fpw), dim3(kGenericThreads), row * fpw, st, x, y, w,
                     shift, skip, (const float*)nullptr, T, F, cin, cout, cout4, kh, kw, 0, pt, pl, fpw, frames);
  HIP_TRY(hipGetLastError());
  return RCED_OK;
}

int reduce_channels(rced_trainer* t, const float* a, const float* b, const float* mu, const float* rstd, size_t P, int C,
                    hipStream_t st) {
  hipLaunchKernelGGL(train::chan_reduce, dim3(kReduceGrid), dim3(train::kThreads), 0, st, a, b, mu, rstd, P, C, t->part);
  hipLaunchKernelGGL(train::reduce_finish, dim3(2 * C), dim3(train::kThreads), 0, st, (const double*)t->part, kReduceGrid, C, t->sums);
  HIP_TRY(hipGetLastError());
  return RCED_OK;
}

// ---- MFMA paths for the 1xk layers (kernels_train_mfma.h, launchers in train_mfma_dispatch.h) ----
// forward shapes (cin, taps, cout) of the 1xk layers and the shapes of their dgrad convolutions (cout, taps, cin):
// CR-CED V3, then R-CED V1.  R-CED V2 (even-padded internal layout) lives in train_mfma_v2.hip.
#define RCED_TM_FWD(X)                        \
  X(8, 9, 18) X(18, 5, 30) X(30, 9, 8)        \
  X(12, 11, 16) X(16, 9, 20) X(20, 7, 24) X(24, 7, 32) X(32, 7, 24) X(24, 9, 20) X(20, 11, 16) X(16, 13, 12)
#define RCED_TM_BWD(X)                                   \
  X(18, 9, 8) X(30, 5, 18) X(8, 9, 30) X(1, 129, 8)      \
  X(16, 11, 12) X(20, 9, 16) X(24, 7, 20) X(32, 7, 24) X(24, 7, 32) X(20, 9, 24) X(16, 11, 20) X(12, 13, 16)
RCED_TM_DEFINE_DISPATCH(_main, RCED_TM_FWD, RCED_TM_BWD)
}  // namespace
// train_mfma_v2.hip
int rced_tm_conv_v2(bool fwd, int cin, int taps, int cout, bool accum, bool stats, const float* in, const float* packet,
                    float* out, int frames, int cus, double* part, const rced::tmm::XformArgs* xa,
                    const rced::tmm::BnBwdArgs* ba, hipStream_t st, const rced::tmm::SumArgs* sa, const float* acc_from);
bool rced_tm_has_v2(bool fwd, int cin, int taps, int cout);
int rced_tm_wgrad_v2(int cin, int taps, int cout, const float* x, const float* dz, float* dW, float* dbias, int frames, int cus,
                     const rced::tmm::XformArgs* xa, const rced::tmm::BnBwdArgs* ba, hipStream_t st);
namespace {
// Returns the grid size (= number of partial-sum records when stats), 0 if no kernel was built for the request.
int tm_conv(bool fwd, int cin, int taps, int cout, bool accum, bool stats, const float* in, const float* packet, float* out,
            int frames, int cus, double* part, const tmm::XformArgs* xa, const tmm::BnBwdArgs* ba, hipStream_t st,
            const tmm::SumArgs* sa = nullptr, const float* acc_from = nullptr) {
  if (tm_has_main(fwd, cin, taps, cout))
    return tm_conv_main(fwd, cin, taps, cout, accum, stats, in, packet, out, frames, cus, part, xa, ba, st, sa, acc_from);
  return rced_tm_conv_v2(fwd, cin, taps, cout, accum, stats, in, packet, out, frames, cus, part, xa, ba, st, sa, acc_from);
}
bool tm_has(bool fwd, int cin, int taps, int cout) { return tm_has_main(fwd, cin, taps, cout) || rced_tm_has_v2(fwd, cin, taps, cout); }
int tm_wgrad(int cin, int taps, int cout, const float* x, const float* dz, float* dW, float* dbias, int frames, int cus,
             const tmm::XformArgs* xa, const tmm::BnBwdArgs* ba, hipStream_t st) {
  if (tm_has_main(true, cin, taps, cout)) return tm_wgrad_main(cin, taps, cout, x, dz, dW, dbias, frames, cus, xa, ba, st);
  return rced_tm_wgrad_v2(cin, taps, cout, x, dz, dW, dbias, frames, cus, xa, ba, st);
}

// ---- wgrad + dgrad of a layer in one kernel (tmm::bwd_fused_mfma): the CR-CED shapes whose input tensor has one consumer ----
#define RCED_TM_FUSED(X) X(18, 5, 30) X(30, 9, 8)
// (x virtual, sums) must be (1, 1) or (0, 0).  Returns the grid size, 0 if no kernel was built for the request.
int tm_bwd_fused(int cin, int taps, int cout, bool xf_sums, const float* x, const float* du, const float* packet, float* dx,
                 float* dW, float* dbias, int frames, int cus, double* part, const tmm::XformArgs* xa, const tmm::BnBwdArgs* ba,
                 hipStream_t st) {
  const tmm::XformArgs nx{nullptr, nullptr, nullptr, nullptr};
  if (!ba || (xf_sums && !xa)) return 0;
#define X(CI, TP, CO)                                                                                                          \
  if (cin == CI && taps == TP && cout == CO)                                                                                   \
    return xf_sums ? rced::tmd::tm_bwd_fused_launch<CI, TP, CO, true, true>(x, du, packet, dx, dW, dbias, frames, cus, part, *xa, *ba, st) \
                   : rced::tmd::tm_bwd_fused_launch<CI, TP, CO, false, false>(x, du, packet, dx, dW, dbias, frames, cus, part, nx, *ba, st);
  RCED_TM_FUSED(X)
#undef X
  return 0;
}

// ---- first layer (8 x kw on the 1-channel input): MFMA wgrad (kernels_train_mfma.h) ----
#define RCED_FIRST(X) X(9, 18) X(13, 12) X(11, 10)
bool first_has(const LayerSpec& s, int cin) {
#define X(KW, CO) if (s.kh == 8 && cin == 1 && s.src == 0 && s.kw == KW && s.cout == CO) return true;
  RCED_FIRST(X)
#undef X
  return false;
}
template <int KW, int COUT>
int first_wgrad_launch(const float* x, const float* dz, float* dW, float* dbias, int frames, int T, int cus,
                       const tmm::BnBwdArgs* ba, hipStream_t st) {
  constexpr int RS = 129 + KW - 1;
  const size_t lds = (((size_t)(tmm::kTF * 8 * RS + 32 + 3) / 4) * 4 + (size_t)(tmm::kTF * tmm::first_wgrad_fs(KW, COUT) + 4) * 32 + 4 * COUT) * sizeof(float);
  const int ntiles = (frames + tmm::kTF - 1) / tmm::kTF;
  const tmm::BnBwdArgs nb{nullptr, nullptr, nullptr, nullptr, nullptr, 1.0, nullptr};
  static unsigned long long attr_t = 0, attr_f = 0;
  static int occ_t = 0, occ_f = 0;
  allow_lds(reinterpret_cast<const void*>(tmm::first_wgrad<KW, COUT, true>), lds, attr_t);
  allow_lds(reinterpret_cast<const void*>(tmm::first_wgrad<KW, COUT, false>), lds, attr_f);
  // persistent grid = what is resident (it was cus * 3 with two workgroups per CU resident: half of the second round idle)
  const dim3 grid(std::min(ntiles, ba ? rced::tmd::resident_grid(reinterpret_cast<const void*>(tmm::first_wgrad<KW, COUT, true>), lds, cus, occ_t)
                                      : rced::tmd::resident_grid(reinterpret_cast<const void*>(tmm::first_wgrad<KW, COUT, false>), lds, cus, occ_f)));
  rced::tmd::wg_launch([&](float* dw, float* db, unsigned ps) {
    if (ba) hipLaunchKernelGGL((tmm::first_wgrad<KW, COUT, true>), grid, dim3(tmm::kThreads), lds, st, x, dz, dw, db, frames, T, *ba, ps);
    else hipLaunchKernelGGL((tmm::first_wgrad<KW, COUT, false>), grid, dim3(tmm::kThreads), lds, st, x, dz, dw, db, frames, T, nb, ps);
  }, (int)grid.x * tmm::kWaves, 8 * KW * COUT, COUT, dW, dbias, st);
  return 1;
}
int first_wgrad(const LayerSpec& s, const float* x, const float* dz, float* dW, float* dbias, int frames, int T, int cus,
                const tmm::BnBwdArgs* ba, hipStream_t st) {
#define X(KW, CO) if (s.kw == KW && s.cout == CO) return first_wgrad_launch<KW, CO>(x, dz, dW, dbias, frames, T, cus, ba, st);
  RCED_FIRST(X)
#undef X
  return 0;
}

template <int KW, int COUT>
int first_fwd_launch(const float* x, const float* w, const float* bias, float* packet, float* z, int frames, int T, int cus,
                     bool stats, double* part, hipStream_t st) {
  constexpr int MT = tmm::tm_rem(COUT) ? 1 : (COUT + 15) / 16, data = 2 * KW * MT * 64 + (tmm::tm_rem(COUT) ? 32 * 64 : 0), RS = 129 + KW - 1;
  hipLaunchKernelGGL(tmm::pack_first, dim3((data + 32 + 255) / 256), dim3(256), 0, st, w, bias, KW, COUT, packet);
  const size_t lds = (((size_t)(tmm::kTF * 8 * RS + 32 + 3) / 4) * 4 + data + 32) * sizeof(float);
  const int ntiles = (frames + tmm::kTF - 1) / tmm::kTF;
  static int occ_s = 0, occ_n = 0;
  const int res = stats ? rced::tmd::resident_grid(reinterpret_cast<const void*>(tmm::first_fwd<KW, COUT, true>), lds, cus, occ_s)
                        : rced::tmd::resident_grid(reinterpret_cast<const void*>(tmm::first_fwd<KW, COUT, false>), lds, cus, occ_n);
  const int grid = std::min(ntiles, std::min(res, kPairGrid));
  if (stats) hipLaunchKernelGGL((tmm::first_fwd<KW, COUT, true>), dim3(grid), dim3(tmm::kThreads), lds, st, x, (const float*)packet, z, frames, T, part);
  else hipLaunchKernelGGL((tmm::first_fwd<KW, COUT, false>), dim3(grid), dim3(tmm::kThreads), lds, st, x, (const float*)packet, z, frames, T, part);
  return grid;
}
// returns the grid size (= partial-sum records when stats)
int first_fwd(const LayerSpec& s, const float* x, const float* w, const float* bias, float* packet, float* z, int frames, int T,
              int cus, bool stats, double* part, hipStream_t st) {
#define X(KW, CO) if (s.kw == KW && s.cout == CO) return first_fwd_launch<KW, CO>(x, w, bias, packet, z, frames, T, cus, stats, part, st);
  RCED_FIRST(X)
#undef X
  return 0;
}
size_t first_packet_floats(const LayerSpec& s) {   // pack_first: main section, the 18-channel form's remainder section, 32 shifts
  return (size_t)2 * s.kw * (tmm::tm_rem(s.cout) ? 1 : (s.cout + 15) / 16) * 64 + (tmm::tm_rem(s.cout) ? 32 * 64 : 0) + 32;
}

// ---- output layer (1x129, CH -> 1): Toeplitz forward + MFMA wgrad (kernels_train_mfma.h) ----
#define RCED_FIN_CH(X) X(8) X(10) X(12)
size_t fin_pack_floats(int ch) {
#define X(CH) if (ch == CH) return tmm::FinGeo<CH>::kPack;
  RCED_FIN_CH(X)
#undef X
  return 0;
}
bool is_output_layer(const LayerSpec& s, int cin) {
  return s.kh == 1 && s.kw == kFeatureDim && s.cout == 1 && !s.use_norm && !s.use_act && s.skip_pre < 0 && s.skip_post < 0 &&
         fin_pack_floats(cin) > 0;
}
// floats of the buffer that holds the output layer's packed A: the fp32 form, or the three-part bf16 form (kernels_final_x6.h)
size_t fin_pack_alloc_floats(int ch) {
  const size_t x6 = ((size_t)((129 * ch + 31) / 32) * 9 * 3 * 64 * 8 * sizeof(unsigned short) + 3) / 4;
  return std::max(fin_pack_floats(ch), x6);
}
int fin_forward(int ch, const float* h, const float* w, const float* bias, float* pack, float* y, int frames, hipStream_t st,
                bool use_x6) {
  if (use_x6) {
    // fp32 quality on the bf16 matrix pipe: every operand as three bf16 parts, six MFMAs of K = 32 per product
    const int total6 = ((129 * ch + 31) / 32) * 9 * 64 * 8;
    hipLaunchKernelGGL(x6::pack_final_x6_dev, dim3((total6 + 255) / 256), dim3(256), 0, st, w, ch, reinterpret_cast<unsigned short*>(pack));
    const dim3 grid6((frames + chain::kFinFrames - 1) / chain::kFinFrames);
#define X(CH)                                                                                                                      \
  if (ch == CH)                                                                                                                    \
    hipLaunchKernelGGL((x6::final_gemm_x6_kernel<CH>), grid6, dim3(chain::kFinThreads), 0, st, h,                                  \
                       (const unsigned short*)reinterpret_cast<unsigned short*>(pack), 0.f, y, frames, bias);
    RCED_FIN_CH(X)
#undef X
    return 1;
  }
  const int total = (int)fin_pack_floats(ch);
  hipLaunchKernelGGL(tmm::pack_final_fwd, dim3((total + 255) / 256), dim3(256), 0, st, w, ch, pack);
  const dim3 grid((frames + tmm::kFinFrames - 1) / tmm::kFinFrames);
  // the inference path's output-layer GEMM with its B operand staged through LDS (kernels_fused_chain.h: 0.54 -> 0.41 ms at
  // config 5's size against tmm::final_fwd's strided global reads); same packed A fragments, the bias read on the device
#define X(CH)                                                                                                               \
  if (ch == CH) {                                                                                                           \
    static_assert(tmm::FinGeo<CH>::kPack == chain::FinalGeo<CH>::kPack && tmm::kFinFrames == chain::kFinFrames, "one pack");   \
    hipLaunchKernelGGL((chain::final_gemm_lds_kernel<CH>), grid, dim3(chain::kFinThreads), 0, st, h, (const float*)pack, 0.f,  \
                       y, frames, bias);                                                                                    \
  }
  RCED_FIN_CH(X)
#undef X
  return 1;
}
size_t fin_dgrad_pack_floats(int ch) {   // the fp32 pack, or the three-part bf16 pack of x6::final_dgrad_x6_kernel
  const size_t mt = (size_t)((129 * ch + 15) / 16);
  return std::max(mt * tmm::kDgSteps * 64, (mt * x6::kDgX6Steps * 3 * 64 * 8 * sizeof(unsigned short) + 3) / 4);
}
int fin_dgrad(int ch, const float* dz, const float* w, float* pack, float* dx, int frames, hipStream_t st, bool use_x6) {
  if (use_x6) {
    const int total6 = ((129 * ch + 15) / 16) * x6::kDgX6Steps * 64 * 8;
    hipLaunchKernelGGL(x6::pack_dgrad_x6_dev, dim3((total6 + 255) / 256), dim3(256), 0, st, w, ch, reinterpret_cast<unsigned short*>(pack));
    const dim3 grid6((frames + x6::kDgX6Frames - 1) / x6::kDgX6Frames);
#define X(CH) \
  if (ch == CH) hipLaunchKernelGGL((x6::final_dgrad_x6_kernel<CH>), grid6, dim3(x6::kDgX6Threads), 0, st, dz, (const unsigned short*)reinterpret_cast<unsigned short*>(pack), dx, frames);
    RCED_FIN_CH(X)
#undef X
    return 1;
  }
  const int total = (int)(((129 * ch + 15) / 16) * tmm::kDgSteps * 64);
  hipLaunchKernelGGL(tmm::pack_final_dgrad, dim3((total + 255) / 256), dim3(256), 0, st, w, ch, pack);
  const dim3 grid((frames + tmm::kDgFrames - 1) / tmm::kDgFrames);
#define X(CH) \
  if (ch == CH) hipLaunchKernelGGL((tmm::final_dgrad<CH>), grid, dim3(tmm::kThreads), 0, st, dz, (const float*)pack, dx, frames);
  RCED_FIN_CH(X)
#undef X
  return 1;
}
int fin_wgrad(int ch, const float* x, const float* dz, float* dW, float* dbias, int frames, int cus, hipStream_t st) {
  const dim3 grid(std::min((frames + tmm::kWaves - 1) / tmm::kWaves, cus * 2));
#define X(CH)                                                                                                              \
  if (ch == CH)                                                                                                            \
    rced::tmd::wg_launch([&](float* dw, float* db, unsigned ps) {                                                          \
      hipLaunchKernelGGL((tmm::final_wgrad<CH>), grid, dim3(tmm::kThreads), 0, st, x, dz, dw, db, frames, ps);             \
    }, (int)grid.x, 129 * CH, 1, dW, dbias, st);
  RCED_FIN_CH(X)
#undef X
  return 1;
}

// (sum d_u, sum d_u * z) from a SUMS dgrad -> (S1, S2 = sum d_u * zhat), zhat = (z - mu) * rstd
__global__ void sums_fix(double* __restrict__ sums, const float* __restrict__ mu, const float* __restrict__ rstd, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) sums[2 * c + 1] = (double)rstd[c] * (sums[2 * c + 1] - (double)mu[c] * sums[2 * c]);
}

// (sum d_u, sum d_u * x), x = relu(a z + b), from the fused backward kernel -> (S1, S2): where d_u != 0, z = (x - b) / a.
// That division is exact enough only while |gamma| is not tiny (x was rounded to fp32: zhat comes back with an error of
// eps |x| / |gamma|), and meaningless for gamma = 0 -- where S2 is still needed: it is gamma's own gradient, and TF's
// d gamma = sum dy * zhat does not vanish with gamma.  So the kernel also raises *redo when any channel of the layer has
// |gamma| < kTinyGamma, and the step then recomputes the layer's sums exactly from (g, z) with bwd_route2 -- launched
// every step with this flag as its only_if, i.e. as a no-op in every step a real training run ever takes.
constexpr float kTinyGamma = 1e-3f;
__global__ void sums_fix_x(double* __restrict__ sums, const float* __restrict__ mu, const float* __restrict__ rstd,
                           const float* __restrict__ gamma, const float* __restrict__ beta, int C, int* __restrict__ redo) {
  const int c = threadIdx.x;     // one workgroup of >= C threads
  bool tiny = false;
  if (c < C) {
    const float a = gamma[c] * rstd[c];                 // the folded forward, exactly as xform_table_fill forms it
    const float b = beta[c] - a * mu[c];
    const double s1 = sums[2 * c], sx = sums[2 * c + 1];
    tiny = !(fabsf(gamma[c]) >= kTinyGamma);
    sums[2 * c + 1] = a != 0.f ? (double)rstd[c] * ((sx - (double)b * s1) / (double)a - (double)mu[c] * s1) : 0.0;
  }
  const int any = __syncthreads_or(tiny);
  if (threadIdx.x == 0) *redo = any;
}

__global__ void sums_to_float(const double* __restrict__ sums, int C, int which, float* __restrict__ dst) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) dst[c] = (float)sums[2 * c + which];
}

// ONE launch for what used to be three to five (reduce_finish over 2 C workgroups, then sums_fix / sums_fix_x or
// bn_stats_finish, then two sums_to_float): a CR-CED step issued ~150 of these 5-us kernels, all on the critical path.
// One workgroup of 1024 threads: thread (i, p) adds records p, p + 16, ... of value i < 2 C, the 16 partial sums are added
// in order (the same bits every run), then threads c < C finish the layer:
//   kFinPlain  sums = (S1, S2) as they are                      (bwd_route2's records)
//   kFinZ      S2 = rstd (sum d_u z - mu S1)                    (a SUMS dgrad's records: sums_fix)
//   kFinX      S2 from sum d_u x, *redo on tiny gamma           (the fused backward kernel's records: sums_fix_x)
//   kFinStats  (sum z, sum z^2) -> mu, rstd, moving statistics  (forward: bn_stats_finish)
// kFinPlain / Z / X also write d beta = S1 and d gamma = S2 (g_beta / g_gamma).  only_if: a device flag, no-op when zero.
enum { kFinPlain = 0, kFinZ = 1, kFinX = 2, kFinStats = 3 };
struct FinishArgs {
  const float *mu, *rstd, *gamma, *beta;      // kFinZ / kFinX
  float *g_beta, *g_gamma;                    // backward modes (may be null)
  int* redo;                                  // kFinX
  double P; float eps, momentum;              // kFinStats
  float *mu_out, *rstd_out, *moving_mean, *moving_var;
};
__global__ __launch_bounds__(1024) void bn_finish(const double* __restrict__ part, int nparts, int C, double* __restrict__ sums,
                                                  int mode, FinishArgs a, const int* __restrict__ only_if) {
  if (only_if && *only_if == 0) return;
  __shared__ double red[16][64];
  const int i = threadIdx.x & 63, p = threadIdx.x >> 6, n = 2 * C;
  double t = 0.0;
  if (i < n) {
    // four independent chains (records p, p+16, p+32, p+48 of every 64), added in a fixed order: the loads of one
    // chain were a serial ~100 ns each, 128 of them for bwd_route2's 2048 records (14 us per launch, 40 launches)
    double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
    int k = p;
    for (; k + 48 < nparts; k += 64) {
      const double a0 = part[(size_t)k * n + i], a1 = part[(size_t)(k + 16) * n + i];
      const double a2 = part[(size_t)(k + 32) * n + i], a3 = part[(size_t)(k + 48) * n + i];
      t0 += a0; t1 += a1; t2 += a2; t3 += a3;
    }
    for (; k < nparts; k += 16) t0 += part[(size_t)k * n + i];
    t = (t0 + t1) + (t2 + t3);
  }
  red[p][i] = t;
  __syncthreads();
  if (p == 0 && i < n) {
    double v = red[0][i];
#pragma unroll
    for (int q = 1; q < 16; ++q) v += red[q][i];
    red[0][i] = v;
  }
  __syncthreads();
  const int c = threadIdx.x;
  bool tiny = false;
  if (c < C) {
    const double s1 = red[0][2 * c];
    double s2 = red[0][2 * c + 1];
    if (mode == kFinStats) {
      const double m = s1 / a.P;
      double var = s2 / a.P - m * m;
      if (var < 0.0) var = 0.0;
      a.mu_out[c] = (float)m;
      a.rstd_out[c] = (float)(1.0 / sqrt(var + (double)a.eps));
      const double unbiased = a.P > 1.0 ? var * a.P / (a.P - 1.0) : var;
      a.moving_mean[c] = (float)((double)a.momentum * (double)a.moving_mean[c] + (1.0 - (double)a.momentum) * m);
      a.moving_var[c] = (float)((double)a.momentum * (double)a.moving_var[c] + (1.0 - (double)a.momentum) * unbiased);
    } else {
      if (mode == kFinZ) {
        s2 = (double)a.rstd[c] * (s2 - (double)a.mu[c] * s1);
      } else if (mode == kFinX) {
        const float ga = a.gamma[c] * a.rstd[c];                 // the folded forward, exactly as xform_table_fill forms it
        const float gb = a.beta[c] - ga * a.mu[c];
        tiny = !(fabsf(a.gamma[c]) >= kTinyGamma);
        s2 = ga != 0.f ? (double)a.rstd[c] * ((s2 - (double)gb * s1) / (double)ga - (double)a.mu[c] * s1) : 0.0;
      }
      if (a.g_beta) a.g_beta[c] = (float)s1;
      if (a.g_gamma) a.g_gamma[c] = (float)s2;
    }
    sums[2 * c] = s1;
    sums[2 * c + 1] = s2;
  }
  if (mode == kFinX) {
    const int any = __syncthreads_or(tiny);
    if (threadIdx.x == 0) *a.redo = any;
  }
}

// One workgroup per layer: does any channel of the layer have |gamma| < kTinyGamma?  (the fused backward kernel's sums
// cannot be trusted there: see sums_fix_x.)  Launched behind every Adam step into host-mapped memory, so that the NEXT
// step knows on the host which layers need the exact recomputation -- instead of launching a conditional bwd_route2 +
// finish pair for every plain layer in every step (20 no-op launches per CR-CED step).
struct TinyScanArgs { int gamma_off[kMaxLayers]; int cout[kMaxLayers]; };
__global__ void tiny_gamma_scan(const float* __restrict__ params, TinyScanArgs a, int* __restrict__ flags) {
  const int l = blockIdx.x, c = threadIdx.x;
  const bool tiny = a.gamma_off[l] >= 0 && c < a.cout[l] && !(fabsf(params[a.gamma_off[l] + c]) >= kTinyGamma);
  const int any = __syncthreads_or(tiny);
  if (c == 0) flags[l] = any;
}
}  // namespace

extern "C" {

int rced_train_create(int variant, const float* blob, size_t n_floats, int batch_size, int device, rced_trainer** out) {
  if (!out) return rced_fail(RCED_ERR_ARG, "out is NULL");
  *out = nullptr;
  const NetSpec* net = net_spec(variant);
  if (!net) return rced_fail(RCED_ERR_ARG, "unknown variant %d", variant);
  if (!blob || n_floats != net_num_weights(*net)) return rced_fail(RCED_ERR_ARG, "blob must hold %zu floats", net_num_weights(*net));
  if (batch_size <= 0) return rced_fail(RCED_ERR_ARG, "batch_size must be positive");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return rced_fail(RCED_ERR_HIP, "no HIP device visible (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return rced_fail(RCED_ERR_ARG, "device %d out of range", device);
  DeviceGuard g(device);
  if (!g.ok) return rced_fail(RCED_ERR_HIP, "hipSetDevice(%d) failed", device);
  rced_trainer* t = new rced_trainer();
  t->variant = variant;
  t->device = device;
  t->batch_size = batch_size;
  t->xvars = n_floats;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) t->num_cus = prop.multiProcessorCount;
    if (const char* e = getenv("RCED_TRAIN_MFMA")) t->use_mfma = atoi(e);
    if (const char* e = getenv("RCED_TRAIN_FUSE_DZ")) t->fuse_dz = atoi(e) != 0;
    if (const char* e = getenv("RCED_TRAIN_FUSE_SUMS")) t->fuse_sums = atoi(e) != 0;
    if (const char* e = getenv("RCED_TRAIN_DET")) t->det = atoi(e) != 0;
    if (const char* e = getenv("RCED_TRAIN_FUSE_BWD")) t->fuse_bwd = atoi(e) != 0;
    if (const char* e = getenv("RCED_TRAIN_X6")) t->use_x6 = atoi(e) != 0;
  }
  const NetSpec* xnet = net;     // the reference's layout (what crosses the ABI)
  t->inet = *xnet;
  const int L = xnet->n_layers;
  if (t->use_mfma)
    for (int l = 0; l < L; ++l)
      if (xnet->layer[l].use_norm) t->inet.layer[l].cout = (xnet->layer[l].cout + 1) & ~1;
  t->net = net = &t->inet;
  size_t o = 0;
  t->off.resize(L);
  for (int l = 0; l < L; ++l) {
    const LayerSpec& s = net->layer[l];
    LayerOff& f = t->off[l];
    f.cin = layer_cin(*net, l);
    f.cout = s.cout;
    f.cout4 = (s.cout + 3) & ~3;
    f.cin4 = (f.cin + 3) & ~3;
    f.K = s.kh * s.kw * f.cin;
    f.kernel = o; o += (size_t)f.K * s.cout;
    f.bias = o; o += s.cout;
    if (s.use_norm) {
      f.gamma = o; o += s.cout;
      f.beta = o; o += s.cout;
      f.mmean = o; o += s.cout;
      f.mvar = o; o += s.cout;
    } else {
      f.gamma = f.beta = f.mmean = f.mvar = 0;
    }
  }
  t->nvars = o;
  // external -> internal index map (graph order: kernel [kh][kw][cin][cout], bias, gamma, beta, moving_mean, moving_variance),
  // the trainable mask and the internal start values (phantom entries: 0, moving_variance 1)
  std::vector<unsigned char> mask(t->nvars, 0);
  std::vector<float> start(t->nvars, 0.f);
  t->x2i.reserve(n_floats);
  for (int l = 0; l < L; ++l) {
    const LayerSpec& xs = xnet->layer[l];
    const LayerOff& f = t->off[l];
    const int xcin = layer_cin(*xnet, l), taps = xs.kh * xs.kw;
    for (int r = 0; r < taps; ++r)
      for (int ci = 0; ci < xcin; ++ci)
        for (int co = 0; co < xs.cout; ++co) {
          const size_t i = f.kernel + ((size_t)r * f.cin + ci) * f.cout + co;
          t->x2i.push_back((int)i);
          mask[i] = 1;
        }
    for (int co = 0; co < xs.cout; ++co) { t->x2i.push_back((int)(f.bias + co)); mask[f.bias + co] = 1; }
    if (xs.use_norm) {
      for (int co = 0; co < xs.cout; ++co) { t->x2i.push_back((int)(f.gamma + co)); mask[f.gamma + co] = 1; }
      for (int co = 0; co < xs.cout; ++co) { t->x2i.push_back((int)(f.beta + co)); mask[f.beta + co] = 1; }
      for (int co = 0; co < xs.cout; ++co) t->x2i.push_back((int)(f.mmean + co));
      for (int co = 0; co < xs.cout; ++co) t->x2i.push_back((int)(f.mvar + co));
      for (int co = 0; co < f.cout; ++co) start[f.mvar + co] = 1.f;
    }
  }
  if (t->x2i.size() != n_floats) { delete t; return rced_fail(RCED_ERR_ARG, "blob layout: %zu floats mapped, %zu given", t->x2i.size(), n_floats); }
  for (size_t e = 0; e < n_floats; ++e) start[t->x2i[e]] = blob[e];
  n_floats = t->nvars;   // from here on: the internal blob
  blob = start.data();
  auto fail_free = [&](int rc) { delete t; return rc; };
#define TRY_OR_FREE(expr)                                                      \
  do {                                                                         \
    hipError_t e_ = (expr);                                                    \
    if (e_ != hipSuccess) return fail_free(rced_fail(RCED_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_))); \
  } while (0)
  const size_t bytes = n_floats * sizeof(float);
  TRY_OR_FREE(hipMalloc(&t->params, bytes));
  TRY_OR_FREE(hipMalloc(&t->grads, bytes));
  TRY_OR_FREE(hipMalloc(&t->m, bytes));
  TRY_OR_FREE(hipMalloc(&t->v, bytes));
  TRY_OR_FREE(hipMalloc(&t->trainable, n_floats));
  TRY_OR_FREE(hipMemcpy(t->params, blob, bytes, hipMemcpyHostToDevice));
  TRY_OR_FREE(hipMemset(t->m, 0, bytes));
  TRY_OR_FREE(hipMemset(t->v, 0, bytes));
  TRY_OR_FREE(hipMemcpy(t->trainable, mask.data(), n_floats, hipMemcpyHostToDevice));
  TRY_OR_FREE(hipMalloc(&t->zero32, 64 * sizeof(float)));
  TRY_OR_FREE(hipMemset(t->zero32, 0, 64 * sizeof(float)));
  TRY_OR_FREE(hipMalloc(&t->part, (size_t)std::max(kReduceGrid, kPairGrid) * train::kMaxC * 2 * sizeof(double)));
  TRY_OR_FREE(hipMalloc(&t->sums, (train::kMaxC * 2 + 2) * sizeof(double)));
  t->redo = reinterpret_cast<int*>(t->sums + train::kMaxC * 2);
  TRY_OR_FREE(hipMemset(t->redo, 0, 2 * sizeof(double)));
  TRY_OR_FREE(hipHostMalloc(reinterpret_cast<void**>(&t->tiny_host), kMaxLayers * sizeof(int), hipHostMallocMapped));
  TRY_OR_FREE(hipHostGetDevicePointer(reinterpret_cast<void**>(&t->tiny_dev), t->tiny_host, 0));
  for (int l = 0; l < kMaxLayers; ++l) {
    t->tiny_host[l] = 0;
    if (l < L && net->layer[l].use_norm)
      for (int c = 0; c < net->layer[l].cout; ++c)
        if (!(std::fabs(blob[t->off[l].gamma + c]) >= kTinyGamma)) t->tiny_host[l] = 1;    // (phantom channels: gamma 0)
  }
  t->wf.assign(L, nullptr); t->wt.assign(L, nullptr); t->bias4.assign(L, nullptr);
  t->mu.assign(L, nullptr); t->rstd.assign(L, nullptr);
  t->pk_fwd.assign(L, nullptr); t->pk_bwd.assign(L, nullptr); t->pk_fwd_x6.assign(L, nullptr); t->pk_bwd_x6.assign(L, nullptr);
  for (int l = 0; l < L; ++l) {
    const LayerSpec& s = net->layer[l];
    const LayerOff& f = t->off[l];
    TRY_OR_FREE(hipMalloc(&t->wf[l], (size_t)f.K * f.cout4 * sizeof(float)));
    TRY_OR_FREE(hipMalloc(&t->wt[l], (size_t)s.kh * s.kw * s.cout * f.cin4 * sizeof(float)));
    TRY_OR_FREE(hipMalloc(&t->bias4[l], 64 * sizeof(float)));
    TRY_OR_FREE(hipMalloc(&t->mu[l], 64 * sizeof(float)));
    TRY_OR_FREE(hipMalloc(&t->rstd[l], 64 * sizeof(float)));
    if (first_has(s, f.cin) && !t->pk_first) TRY_OR_FREE(hipMalloc(&t->pk_first, first_packet_floats(s) * sizeof(float)));
    if (is_output_layer(s, f.cin) && !t->pk_fin_bwd) TRY_OR_FREE(hipMalloc(&t->pk_fin_bwd, fin_dgrad_pack_floats(f.cin) * sizeof(float)));
    if (is_output_layer(s, f.cin) && !t->pk_fin) TRY_OR_FREE(hipMalloc(&t->pk_fin, fin_pack_alloc_floats(f.cin) * sizeof(float)));
    if (s.kh == 1 && tm_has(true, f.cin, s.kw, s.cout)) TRY_OR_FREE(hipMalloc(&t->pk_fwd[l], tm_packet_floats(f.cin, s.kw, s.cout) * sizeof(float)));
    if (t->use_x6 && s.kh == 1 && t->pk_fwd[l] && rced::tmd::tm_x6_has(f.cin, s.kw, s.cout))
      TRY_OR_FREE(hipMalloc(&t->pk_fwd_x6[l], rced::tmd::tm_packet_x6_floats(f.cin, s.kw, s.cout) * sizeof(float)));
    if (s.kh == 1 && tm_has(false, s.cout, s.kw, f.cin)) TRY_OR_FREE(hipMalloc(&t->pk_bwd[l], tm_packet_floats(s.cout, s.kw, f.cin) * sizeof(float)));
    if (t->pk_bwd[l] && tmm::bwd_x6(f.cin, s.kw, s.cout))   // (compiled into the fused kernel of this shape: not a per-trainer switch)
      TRY_OR_FREE(hipMalloc(&t->pk_bwd_x6[l], rced::tmd::tm_packet_x6_floats(s.cout, s.kw, f.cin) * sizeof(float)));
  }
  // Tensors that need not exist in HBM: output of a plain conv+BN+ReLU layer (no skip in or out) whose only
  // consumer is a 1xk layer with MFMA forward and wgrad kernels.  RCED_TRAIN_FUSE_ACT=0 turns this off.
  t->virt.assign(L + 1, 0);
  {
    const char* e = getenv("RCED_TRAIN_FUSE_ACT");
    const bool fuse = t->use_mfma && !(e && atoi(e) == 0);
    std::vector<int> uses(L + 1, 0), conv_user(L + 1, -1), post_uses(L + 1, 0);
    for (int l = 0; l < L; ++l) {
      const LayerSpec& s = t->net->layer[l];
      if (s.src > 0) { ++uses[s.src]; conv_user[s.src] = l; }
      if (s.skip_pre > 0) uses[s.skip_pre] += 2;     // a skip added before the ReLU disqualifies (its backward reads the tensor)
      // a skip added AFTER a ReLU (CR-CED's block skips) is read once, by bn_act_fwd2, which rebuilds it from the
      // producer's z just as well (same bytes, same arithmetic); its backward needs no values
      if (s.skip_post > 0) { if (s.cout % 2 == 0 && s.skip_pre < 0) ++post_uses[s.skip_post]; else uses[s.skip_post] += 2; }
    }
    for (int id = 1; fuse && id < L; ++id) {
      const LayerSpec& p = t->net->layer[id - 1];
      const int c = conv_user[id];
      if (uses[id] != 1 || c < 0 || post_uses[id] > 1) continue;
      const LayerSpec& q = t->net->layer[c];
      t->virt[id] = p.use_norm && p.use_act && p.skip_pre < 0 && p.skip_post < 0 && p.cout % 2 == 0 && q.kh == 1 &&
                    q.cout % 2 == 0 && t->pk_fwd[c] != nullptr && tm_has(true, p.cout, q.kw, q.cout);
    }
  }
#undef TRY_OR_FREE
  *out = t;
  return RCED_OK;
}

void rced_train_destroy(rced_trainer* t) { delete t; }

long long rced_train_global_step(rced_trainer* t) { return t ? t->global_step : -1; }

namespace {
// internal (even-padded) device array -> the reference's layout on the host, and back (phantom entries become 0)
int download_external(rced_trainer* t, const float* dev, float* host) {
  std::vector<float> tmp(t->nvars);
  HIP_TRY(hipMemcpy(tmp.data(), dev, t->nvars * sizeof(float), hipMemcpyDeviceToHost));
  for (size_t e = 0; e < t->xvars; ++e) host[e] = tmp[t->x2i[e]];
  return RCED_OK;
}
int upload_external(rced_trainer* t, const float* host, float* dev) {
  std::vector<float> tmp(t->nvars, 0.f);
  for (size_t e = 0; e < t->xvars; ++e) tmp[t->x2i[e]] = host[e];
  HIP_TRY(hipMemcpy(dev, tmp.data(), t->nvars * sizeof(float), hipMemcpyHostToDevice));
  return RCED_OK;
}
}  // namespace

int rced_train_get_variables(rced_trainer* t, float* blob_host, size_t n_floats) {
  if (!t || !blob_host || n_floats != t->xvars) return rced_fail(RCED_ERR_ARG, "bad arguments");
  DeviceGuard g(t->device);
  return download_external(t, t->params, blob_host);
}

int rced_train_get_gradients(rced_trainer* t, float* blob_host, size_t n_floats) {
  if (!t || !blob_host || n_floats != t->xvars) return rced_fail(RCED_ERR_ARG, "bad arguments");
  DeviceGuard g(t->device);
  return download_external(t, t->grads, blob_host);
}

int rced_train_get_state(rced_trainer* t, float* m_blob_host, float* v_blob_host, size_t n_floats, long long* global_step) {
  if (!t || !m_blob_host || !v_blob_host || n_floats != t->xvars) return rced_fail(RCED_ERR_ARG, "bad arguments");
  DeviceGuard g(t->device);
  if (int rc = download_external(t, t->m, m_blob_host)) return rc;
  if (int rc = download_external(t, t->v, v_blob_host)) return rc;
  if (global_step) *global_step = t->global_step;
  return RCED_OK;
}

int rced_train_set_state(rced_trainer* t, const float* m_blob_host, const float* v_blob_host, size_t n_floats,
                         long long global_step) {
  if (!t || !m_blob_host || !v_blob_host || n_floats != t->xvars || global_step < 0)
    return rced_fail(RCED_ERR_ARG, "bad arguments");
  DeviceGuard g(t->device);
  if (int rc = upload_external(t, m_blob_host, t->m)) return rc;
  if (int rc = upload_external(t, v_blob_host, t->v)) return rc;
  t->global_step = global_step;
  return RCED_OK;
}

}  // extern "C"

namespace {
// pred_dev != nullptr: forward only (rced_train_forward) -- batch-statistics BatchNorm, nothing is updated, the
// prediction is copied to pred_dev.  Otherwise the full step.
int train_run(rced_trainer* t, const float* x_dev, const float* y_dev, float* pred_dev, int N, int T, float lr,
              double* loss_out, void* stream) {
  if (!t) return rced_fail(RCED_ERR_ARG, "trainer is NULL");
  if (N <= 0 || T <= 0 || !x_dev || (!y_dev && !pred_dev)) return rced_fail(RCED_ERR_ARG, "bad batch");
  const bool forward_only = pred_dev != nullptr;
  DeviceGuard g(t->device);
  if (!g.ok) return rced_fail(RCED_ERR_HIP, "hipSetDevice(%d) failed", t->device);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const NetSpec& net = *t->net;
  const int L = net.n_layers, F = kFeatureDim, frames = N * T;
  const size_t P = (size_t)frames * F;
  if (int rc = ensure_acts(t, P)) return rc;
  // deterministic weight gradients: the wgrad launchers of this thread write slices into t->wpart (tmd::WgDet)
  struct WgScope {
    explicit WgScope(rced_trainer* tr) {
      tr->wg_error = 0;
      rced::tmd::g_wgdet = tr->det ? rced::tmd::WgDet{&tr->wpart, &tr->wpart_floats, &tr->wg_error} : rced::tmd::WgDet{};
    }
    ~WgScope() { rced::tmd::g_wgdet = rced::tmd::WgDet{}; }
  };
  if (t->det && !forward_only && !t->wpart) {
    // Sized BEFORE the forward for the most any wgrad launcher can ask for: slices = workgroups (CUs x at most 4 resident per
    // CU) x 8 waves x 2 pixel parities, slice stride = the net's largest kernel + bias.  A failure to allocate is therefore
    // reported here, before this step has touched anything; growing inside tmd::wg_launch remains as a last resort (it
    // synchronises, and if it fails the step returns RCED_ERR_ALLOC below with the moving statistics already advanced).
    size_t maxps = 0;
    for (int l = 0; l < L; ++l) {
      const LayerSpec& s = net.layer[l];
      const int cin = s.src == 0 ? 1 : (net.layer[s.src - 1].cout + 1) & ~1, cout = (s.cout + 1) & ~1;   // (R-CED V2: even-padded)
      maxps = std::max(maxps, (size_t)s.kh * s.kw * cin * cout + cout + 4);
    }
    t->wpart_floats = (size_t)t->num_cus * 4 * 8 * 2 * maxps;
    HIP_TRY(hipMalloc(&t->wpart, t->wpart_floats * sizeof(float)));
  }
  const WgScope wg_scope(t);
  auto blocks = [](size_t n) { return dim3((unsigned)std::min<size_t>((n + train::kThreads - 1) / train::kThreads, 65535)); };
  auto pair_grid = [&](int C) {   // channel-aligned kernels: rows = 256 / (C/2) pixels per workgroup pass
    const size_t rows = train::kThreads / (C / 2);
    return dim3((unsigned)std::min<size_t>((P + rows - 1) / rows, kPairGrid));
  };
  // a virtual tensor is read as its producer's z plus that layer's BatchNorm parameters (BnReluXform)
  tmm::XformArgs xa_tmp{nullptr, nullptr, nullptr, nullptr};
  auto conv_in = [&](int id) -> const float* { return id == 0 ? x_dev : (t->virt[id] ? t->z[id - 1] : t->out[id]); };
  auto xform_of = [&](int id, tmm::XformArgs* xa) -> const tmm::XformArgs* {
    if (id <= 0 || !t->virt[id]) return nullptr;
    const LayerOff& pf = t->off[id - 1];
    *xa = tmm::XformArgs{t->mu[id - 1], t->rstd[id - 1], t->params + pf.gamma, t->params + pf.beta};
    return xa;
  };
  auto tensor = [&](int id) -> const float* { return id < 0 ? nullptr : (id == 0 ? x_dev : t->out[id]); };

  // consumers[id]: how many layers read tensor id (as conv input or as a skip)
  std::vector<int> consumers(L + 1, 0);
  for (int l = 0; l < L; ++l) {
    if (net.layer[l].src > 0) ++consumers[net.layer[l].src];
    if (net.layer[l].skip_pre > 0) ++consumers[net.layer[l].skip_pre];
    if (net.layer[l].skip_post > 0) ++consumers[net.layer[l].skip_post];
  }
  // ---- weights in the layouts the direct-convolution kernels want (only for layers that fall back to them)
  for (int l = 0; l < L; ++l) {
    const LayerSpec& s = net.layer[l];
    const LayerOff& f = t->off[l];
    const bool mfma_fwd = t->use_mfma && (t->pk_fwd[l] || (t->pk_first && first_has(s, f.cin)) || (t->pk_fin && is_output_layer(s, f.cin)));
    const bool mfma_bwd = t->use_mfma && (s.src == 0 || t->pk_bwd[l] ||
                                          (t->pk_fin_bwd && is_output_layer(s, f.cin) && consumers[s.src] == 1));
    if (!mfma_fwd) {
      hipLaunchKernelGGL(train::repack_fwd, dim3((f.K * f.cout4 + 255) / 256), dim3(256), 0, st, t->params + f.kernel, f.K,
                         s.cout, f.cout4, t->wf[l]);
      hipLaunchKernelGGL(train::repack_fwd, dim3(1), dim3(64), 0, st, t->params + f.bias, 1, s.cout, f.cout4, t->bias4[l]);
    }
    if (!mfma_bwd) {
      const int nt = s.kh * s.kw * s.cout * f.cin4;
      hipLaunchKernelGGL(train::repack_dgrad, dim3((nt + 255) / 256), dim3(256), 0, st, t->params + f.kernel, s.kh, s.kw,
                         f.cin, s.cout, f.cin4, t->wt[l]);
    }
  }
  if (t->use_mfma)
    for (int l = 0; l < L; ++l) {
      const LayerSpec& s = net.layer[l];
      const LayerOff& f = t->off[l];
      if (t->pk_fwd_x6[l]) {
        const int n = rced::tmd::tm_packet_x6_threads(f.cin, s.kw, s.cout);
        hipLaunchKernelGGL(tmm::pack_packet_x6, dim3((n + 255) / 256), dim3(256), 0, st, (const float*)(t->params + f.kernel),
                           (const float*)(t->params + f.bias), s.kw, f.cin, s.cout, rced::tmd::tm_packet_parities(s.cout), t->pk_fwd_x6[l]);
      } else if (t->pk_fwd[l]) {
        const int n = (int)tm_packet_floats(f.cin, s.kw, s.cout);
        hipLaunchKernelGGL(tmm::pack_packet, dim3((n + 255) / 256), dim3(256), 0, st, (const float*)(t->params + f.kernel),
                           (const float*)(t->params + f.bias), s.kw, f.cin, s.cout, 0, rced::tmd::tm_packet_parities(s.cout),
                           t->pk_fwd[l]);
      }
      if (t->pk_bwd_x6[l]) {
        // the fused backward kernel of this shape runs its dgrad half in the three-part bf16 form: its packet in that form too
        // (the fp32 packet below stays: the separate dgrad kernel takes it when the fused one is not used)
        const int n = rced::tmd::tm_packet_x6_threads(s.cout, s.kw, f.cin);
        hipLaunchKernelGGL(tmm::pack_packet_x6, dim3((n + 255) / 256), dim3(256), 0, st, (const float*)(t->params + f.kernel),
                           (const float*)nullptr, s.kw, s.cout, f.cin, rced::tmd::tm_packet_parities(f.cin), t->pk_bwd_x6[l], 1);
      }
      if (t->pk_bwd[l]) {
        const int n = (int)tm_packet_floats(s.cout, s.kw, f.cin);
        hipLaunchKernelGGL(tmm::pack_packet, dim3((n + 255) / 256), dim3(256), 0, st, (const float*)(t->params + f.kernel),
                           (const float*)nullptr, s.kw, s.cout, f.cin, 1, rced::tmd::tm_packet_parities(f.cin), t->pk_bwd[l]);
      }
    }
  // ---- forward (is_training=True)
  for (int l = 0; l < L; ++l) {
    const LayerSpec& s = net.layer[l];
    const LayerOff& f = t->off[l];
    int stat_parts = 0;   // > 0: the conv kernel already left that many (sum z, sum z^2) records in t->part
    if (t->use_mfma && t->pk_fwd_x6[l] &&
        (stat_parts = rced::tmd::tm_conv_x6(f.cin, s.kw, s.cout, s.use_norm != 0, conv_in(s.src), t->pk_fwd_x6[l], t->z[l], frames,
                                            t->num_cus, t->part, xform_of(s.src, &xa_tmp), st)) > 0) {
      if (!s.use_norm) stat_parts = 0;
    } else if (t->use_mfma && t->pk_fwd[l] &&
        (stat_parts = tm_conv(true, f.cin, s.kw, s.cout, false, s.use_norm != 0, conv_in(s.src), t->pk_fwd[l], t->z[l],
                              frames, t->num_cus, t->part, xform_of(s.src, &xa_tmp), nullptr, st)) > 0) {
      if (!s.use_norm) stat_parts = 0;
    } else if (t->use_mfma && t->pk_first && first_has(s, f.cin) &&
               (stat_parts = first_fwd(s, x_dev, t->params + f.kernel, t->params + f.bias, t->pk_first, t->z[l], frames, T,
                                       t->num_cus, s.use_norm != 0, t->part, st)) > 0) {
      if (!s.use_norm) stat_parts = 0;
    } else if (t->use_mfma && t->pk_fin && is_output_layer(s, f.cin)) {
      fin_forward(f.cin, tensor(s.src), t->params + f.kernel, t->params + f.bias, t->pk_fin, t->z[l], frames, st, t->use_x6);
    } else if (s.src > 0 && t->virt[s.src]) {
      // a virtual input exists only inside the MFMA kernels' staging: never hand its (null) pointer to the direct kernel
      return rced_fail(RCED_ERR_STATE, "layer %d: no MFMA forward kernel for a layer whose input is not materialised", l);
    } else if (int rc = launch_conv(tensor(s.src), t->z[l], t->wf[l], t->bias4[l], nullptr, frames, T, F, f.cin, s.cout,
                                    f.cout4, s.kh, s.kw, (s.kh - 1) / 2, (s.kw - 1) / 2, st)) {
      return rc;
    }
    const size_t n = P * s.cout;
    if (s.use_norm) {
      if (stat_parts > 0) {
        FinishArgs fa{};
        fa.P = (double)P; fa.eps = kBnEps; fa.momentum = forward_only ? 1.f : kBnMomentum;   // momentum 1: the moving statistics stay as they are
        fa.mu_out = t->mu[l]; fa.rstd_out = t->rstd[l]; fa.moving_mean = t->params + f.mmean; fa.moving_var = t->params + f.mvar;
        hipLaunchKernelGGL(bn_finish, dim3(1), dim3(1024), 0, st, (const double*)t->part, stat_parts, s.cout, t->sums, (int)kFinStats, fa,
                           (const int*)nullptr);
      } else {
        if (int rc = reduce_channels(t, t->z[l], t->z[l], nullptr, nullptr, P, s.cout, st)) return rc;
        hipLaunchKernelGGL(train::bn_stats_finish, dim3(1), dim3(64), 0, st, (const double*)t->sums, (double)P, s.cout,
                           kBnEps, forward_only ? 1.f : kBnMomentum, t->mu[l], t->rstd[l], t->params + f.mmean,
                           t->params + f.mvar);
      }
    }
    if (t->out[l + 1] != t->z[l] && !t->virt[l + 1]) {
      // a virtual post-ReLU skip source: rebuilt from its producer's z inside bn_act_fwd2
      const bool vskip = s.skip_post > 0 && t->virt[s.skip_post] != 0;
      tmm::XformArgs vs{nullptr, nullptr, nullptr, nullptr};
      if (vskip) xform_of(s.skip_post, &vs);
      if (vskip && s.cout % 2 != 0) return rced_fail(RCED_ERR_STATE, "layer %d: a virtual skip source needs the pair kernel", l);
      if (s.cout % 2 == 0)
        hipLaunchKernelGGL(train::bn_act_fwd2, pair_grid(s.cout), dim3(train::kThreads), 0, st, (const float2*)t->z[l],
                           s.use_norm ? (const float*)t->mu[l] : nullptr, (const float*)t->rstd[l],
                           (const float*)(t->params + f.gamma), (const float*)(t->params + f.beta),
                           (const float2*)tensor(s.skip_pre), (const float2*)(vskip ? t->z[s.skip_post - 1] : tensor(s.skip_post)),
                           s.use_act, P, s.cout, (float2*)t->out[l + 1], vskip ? vs.mu : nullptr, vs.rstd, vs.gamma, vs.beta);
      else
        hipLaunchKernelGGL(train::bn_act_fwd, blocks(n), dim3(train::kThreads), 0, st, (const float*)t->z[l],
                           s.use_norm ? (const float*)t->mu[l] : nullptr, (const float*)t->rstd[l],
                           (const float*)(t->params + f.gamma), (const float*)(t->params + f.beta), tensor(s.skip_pre),
                           tensor(s.skip_post), s.use_act, n, s.cout, t->out[l + 1]);
    }
  }
  HIP_TRY(hipGetLastError());
  if (forward_only) {
    HIP_TRY(hipMemcpyAsync(pred_dev, t->out[L], P * sizeof(float), hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    return RCED_OK;
  }
  // ---- loss and its gradient (trainer.py:146-147,153)
  hipLaunchKernelGGL(train::loss_fwd_bwd, dim3(kReduceGrid), dim3(train::kThreads), 0, st, (const float*)t->out[L], y_dev,
                     P, 1.f / (float)t->batch_size, t->G[L], t->part);
  std::vector<double> hp(kReduceGrid);
  HIP_TRY(hipMemcpyAsync(hp.data(), t->part, kReduceGrid * sizeof(double), hipMemcpyDeviceToHost, st));
  // ---- backward
  HIP_TRY(hipMemsetAsync(t->grads, 0, t->nvars * sizeof(float), st));
  // G[id] collects d loss / d tensor id from every consumer (the conv reading it, skip adds).  A tensor with one
  // consumer is written (=) by that consumer's dgrad; the others are zeroed here and accumulated into (+=).
  auto overwrite = [&](int l) {   // layer l's dgrad may overwrite G[src] (MFMA path only; the generic kernel always +=)
    const LayerSpec& s = net.layer[l];
    const bool mfma_dgrad = t->pk_bwd[l] != nullptr || (t->pk_fin_bwd != nullptr && is_output_layer(s, t->off[l].cin));
    return t->use_mfma && mfma_dgrad && s.src > 0 && consumers[s.src] == 1;
  };
  std::vector<char> lazy_zero(L + 1, 0), written(L + 1, 0);
  {
    std::vector<char> plain(L + 1, 0);
    for (int l = 0; l < L; ++l) if (overwrite(l)) plain[net.layer[l].src] = 1;
    // A tensor that some layer adds as a skip gets its first gradient contribution from that layer's bwd_route2 (the skip
    // consumer comes later in the net than the convolution that reads the tensor, so earlier in this loop), which can
    // STORE it: no memset, no read-modify-write there.  written[id] tracks it; whoever must add to a tensor nobody has
    // written yet zeroes it first (ensure_zero: not reached in the three nets).
    for (int l = 0; l < L; ++l) {
      const LayerSpec& s = net.layer[l];
      if (t->use_mfma && s.cout % 2 == 0) {
        if (s.skip_pre > 0) lazy_zero[s.skip_pre] = 1;
        if (s.skip_post > 0) lazy_zero[s.skip_post] = 1;
      }
    }
    for (int id = 1; id < L; ++id) {
      if (plain[id] || !lazy_zero[id]) written[id] = 1;
      if (!plain[id] && !lazy_zero[id]) HIP_TRY(hipMemsetAsync(t->G[id], 0, P * net.layer[id - 1].cout * sizeof(float), st));
    }
  }
  auto ensure_zero = [&](int id) -> int {
    if (id > 0 && !written[id]) {
      HIP_TRY(hipMemsetAsync(t->G[id], 0, P * net.layer[id - 1].cout * sizeof(float), st));
      written[id] = 1;
    }
    return RCED_OK;
  };
  // fused_sums[l] > 0: the dgrad that wrote G[l + 1] (layer l's only consumer) has left that many (sum d_u, sum d_u z)
  // records of layer l's BatchNorm backward in t->part (tmm::SumArgs): no bwd_route2 pass for layer l.
  int tiny[kMaxLayers];
  for (int l = 0; l < kMaxLayers; ++l) tiny[l] = t->tiny_host[l];     // as of the end of the previous step (it synchronised)
  std::vector<int> fused_sums(L, 0);
  std::vector<char> sums_from_x(L, 0);   // those records hold (sum d_u, sum d_u * x) (fused backward kernel) rather than (.., sum d_u * z)
  const bool fuse_sums_on = t->fuse_sums;
  auto fuse_dz_of = [&](int l) {
    const LayerSpec& s = net.layer[l];
    const LayerOff& f = t->off[l];
    return t->fuse_dz && t->use_mfma && s.use_norm && s.cout % 2 == 0 &&
           ((t->use_mfma && first_has(s, f.cin)) || (s.kh == 1 && f.cin % 2 == 0 && tm_has(true, f.cin, s.kw, s.cout) &&
                                                    (s.src == 0 || t->pk_bwd[l] != nullptr)));
  };
  auto lazy_mask_of = [&](int l) {
    const LayerSpec& s = net.layer[l];
    // (a skip added AFTER the ReLU -- CR-CED's block outputs, model.py:75-76 -- does not enter the mask: d_u = g [bn(z) > 0]
    // there too, so those layers' d_u need not be written either; bwd_route2 still routes g to the skip's source)
    return fuse_dz_of(l) && s.use_act && s.skip_pre < 0;
  };
  // alias_src[id]: G[id]'s first contribution would be a plain copy of another gradient tensor (layer ls adds tensor id
  // AFTER its ReLU: d tensor id += G[ls + 1] unchanged).  The copy is not made: the one dgrad that completes G[id] reads
  // its accumulate operand from G[ls + 1] instead (out = acc_from + conv).  G[ls + 1] is final by then -- its writers are
  // the consumers of tensor ls + 1, all later layers -- and every G tensor is its own allocation.
  std::vector<const float*> alias_src(L + 1, nullptr);
  auto conv_consumer_of = [&](int id) {   // the layer reading tensor id as its convolution input, when there is exactly one
    int lc = -1;
    for (int k = 0; k < L; ++k)
      if (net.layer[k].src == id) lc = lc < 0 ? k : -2;
    return lc;
  };
  auto alias_ok = [&](int l) {
    const LayerSpec& s = net.layer[l];
    if (!t->use_mfma || !t->fuse_dz || s.skip_post <= 0 || s.skip_pre > 0 || written[s.skip_post] || consumers[s.skip_post] != 2) return false;
    const int lc = conv_consumer_of(s.skip_post);
    if (lc < 0 || lc >= l || !t->pk_bwd[lc]) return false;
    const LayerSpec& c = net.layer[lc];
    return c.kh == 1 && c.cout % 2 == 0 && tm_has(false, c.cout, c.kw, t->off[lc].cin);
  };
  // a producer whose BatchNorm-backward sums may come out of its consumer's dgrad: masked lazily, and nothing left to
  // route -- no skip, or a post-ReLU skip whose gradient is not copied (alias_ok; evaluated here, at the consumer, and again at
  // the producer itself, with nothing writing the skip's gradient tensor in between)
  auto sums_in_dgrad_ok = [&](int pl) { return lazy_mask_of(pl) && (net.layer[pl].skip_post < 0 || alias_ok(pl)); };
  for (int l = L - 1; l >= 0; --l) {
    const LayerSpec& s = net.layer[l];
    const LayerOff& f = t->off[l];
    const size_t n = P * s.cout;
    const float* mu = s.use_norm ? t->mu[l] : nullptr;
    const bool pairs = s.cout % 2 == 0;
    // BatchNorm backward: either applied in place on D (bn_bwd_apply*), or -- when both consumers of dz are MFMA
    // kernels -- folded into their staging, which reads (d_u, z) and never materialises dz (tile_commit_bnbwd).
    // For a plain conv+BN+ReLU layer (no skip in or out) d_u is not materialised either: the consumers read the
    // incoming gradient g and apply the ReLU mask themselves; bwd_route2 then only produces the two sums.
    const bool first_mfma = t->use_mfma && first_has(s, f.cin);
    const bool fuse_dz = fuse_dz_of(l);
    const bool lazy_mask = lazy_mask_of(l);
    // a linear layer without BatchNorm or skips (decode_final): d_u IS the incoming gradient -- no routing pass, no copy
    const bool passthrough = t->use_mfma && !s.use_act && !s.use_norm && s.skip_pre < 0 && s.skip_post < 0 && is_output_layer(s, f.cin) && t->pk_fin &&
                             t->pk_fin_bwd && s.src > 0 && consumers[s.src] == 1;   // (both of its consumers below take dsrc)
    const float* dsrc = lazy_mask || passthrough ? t->G[l + 1] : t->D;      // what wgrad / dgrad read as their "dz" input
    bool grads_out = false;      // d beta / d gamma already written by bn_finish
    FinishArgs fb{};
    fb.mu = mu; fb.rstd = t->rstd[l]; fb.gamma = t->params + f.gamma; fb.beta = t->params + f.beta;
    fb.g_beta = t->grads + f.beta; fb.g_gamma = t->grads + f.gamma; fb.redo = t->redo;
    // a post-ReLU skip whose gradient is not copied (alias_src above): recorded before the branches -- the layer's sums may
    // already be there (fused_sums), in which case nothing of this layer is routed at all
    const bool alias = alias_ok(l);
    if (alias) {
      alias_src[s.skip_post] = t->G[l + 1];
      written[s.skip_post] = 1;
    }
    if (passthrough) {
      // nothing to route
    } else if (lazy_mask && fused_sums[l] > 0) {
      if (s.skip_post > 0 && !alias)
        return rced_fail(RCED_ERR_STATE, "layer %d: sums came out of the dgrad but its skip gradient still needs routing", l);
      hipLaunchKernelGGL(bn_finish, dim3(1), dim3(1024), 0, st, (const double*)t->part, fused_sums[l], s.cout, t->sums,
                         (int)(sums_from_x[l] ? kFinX : kFinZ), fb, (const int*)nullptr);
      grads_out = true;
      if (sums_from_x[l] && tiny[l]) {
        // |gamma| tiny somewhere in this layer (known on the host since the previous step's tiny_gamma_scan): the sums
        // again, exactly, from (g, z); see sums_fix_x.  Never taken in a real training run.
        const dim3 grid = pair_grid(s.cout);
        hipLaunchKernelGGL(train::bwd_route2, grid, dim3(train::kThreads), 0, st, (const float2*)t->G[l + 1],
                           (const float2*)t->z[l], mu, (const float*)t->rstd[l], (const float*)(t->params + f.gamma),
                           (const float*)(t->params + f.beta), (const float2*)nullptr, s.use_act, P, s.cout, (float2*)nullptr,
                           (float2*)nullptr, (float2*)nullptr, t->part, (const int*)nullptr);
        hipLaunchKernelGGL(bn_finish, dim3(1), dim3(1024), 0, st, (const double*)t->part, (int)grid.x, s.cout, t->sums,
                           (int)kFinPlain, fb, (const int*)nullptr);
      }
    } else if (pairs) {
      const dim3 grid = pair_grid(s.cout);
      // (a layer has at most one skip; first writer of its gradient tensor: store, see `written`)
      const int skip_id = s.skip_pre > 0 ? s.skip_pre : s.skip_post;
      const int skip_first = skip_id > 0 && !written[skip_id] ? 1 : 0;
      if (skip_id > 0) written[skip_id] = 1;
      hipLaunchKernelGGL(train::bwd_route2, grid, dim3(train::kThreads), 0, st, (const float2*)t->G[l + 1],
                         (const float2*)t->z[l], mu, (const float*)t->rstd[l], (const float*)(t->params + f.gamma),
                         (const float*)(t->params + f.beta), (const float2*)tensor(s.skip_pre), s.use_act, P, s.cout,
                         (float2*)(s.skip_pre > 0 ? t->G[s.skip_pre] : nullptr),
                         (float2*)(s.skip_post > 0 && !alias ? t->G[s.skip_post] : nullptr), (float2*)(lazy_mask ? nullptr : t->D),
                         s.use_norm ? t->part : (double*)nullptr, (const int*)nullptr, skip_first);
      if (s.use_norm) {
        hipLaunchKernelGGL(bn_finish, dim3(1), dim3(1024), 0, st, (const double*)t->part, (int)grid.x, s.cout, t->sums,
                           (int)kFinPlain, fb, (const int*)nullptr);
        grads_out = true;
      }
    } else {
      if (int rc = ensure_zero(s.skip_pre)) return rc;
      if (int rc = ensure_zero(s.skip_post)) return rc;
      hipLaunchKernelGGL(train::bwd_route, blocks(n), dim3(train::kThreads), 0, st, (const float*)t->G[l + 1],
                         (const float*)t->z[l], mu, (const float*)t->rstd[l], (const float*)(t->params + f.gamma),
                         (const float*)(t->params + f.beta), tensor(s.skip_pre), s.use_act, n, s.cout,
                         s.skip_pre > 0 ? t->G[s.skip_pre] : nullptr, s.skip_post > 0 ? t->G[s.skip_post] : nullptr, t->D);
      if (s.use_norm)
        if (int rc = reduce_channels(t, t->D, t->z[l], mu, t->rstd[l], P, s.cout, st)) return rc;
    }
    const tmm::BnBwdArgs ba_l{t->z[l], mu, t->rstd[l], t->params + f.gamma, t->sums, (double)P,
                              lazy_mask ? t->params + f.beta : nullptr};
    const tmm::BnBwdArgs* ba = fuse_dz ? &ba_l : nullptr;
    if (s.use_norm) {
      if (!grads_out) {
        hipLaunchKernelGGL(sums_to_float, dim3(1), dim3(64), 0, st, (const double*)t->sums, s.cout, 0, t->grads + f.beta);
        hipLaunchKernelGGL(sums_to_float, dim3(1), dim3(64), 0, st, (const double*)t->sums, s.cout, 1, t->grads + f.gamma);
      }
      if (fuse_dz) {
        // nothing: wgrad and dgrad below rebuild dz from D = d_u and z
      } else if (pairs)
        hipLaunchKernelGGL(train::bn_bwd_apply2, pair_grid(s.cout), dim3(train::kThreads), 0, st, (float2*)t->D,
                           (const float2*)t->z[l], mu, (const float*)t->rstd[l], (const float*)(t->params + f.gamma),
                           (const double*)t->sums, (double)P, P, s.cout);
      else
        hipLaunchKernelGGL(train::bn_bwd_apply, blocks(n), dim3(train::kThreads), 0, st, t->D, (const float*)t->z[l], mu,
                           (const float*)t->rstd[l], (const float*)(t->params + f.gamma), (const double*)t->sums, (double)P, n,
                           s.cout);
    }
    // wgrad and dgrad in one kernel where the layer's input tensor has this layer as its only consumer
    bool fused_done = false;
    if (t->fuse_bwd && t->use_mfma && s.kh == 1 && s.src > 0 && fuse_dz && t->pk_bwd[l] && overwrite(l)) {
      const int pl = s.src - 1;
      const bool xvirt = t->virt[s.src] != 0, want_sums = fuse_sums_on && sums_in_dgrad_ok(pl);
      if (xvirt == want_sums) {
        const int g = tm_bwd_fused(f.cin, s.kw, s.cout, xvirt, conv_in(s.src), dsrc, t->pk_bwd_x6[l] ? t->pk_bwd_x6[l] : t->pk_bwd[l], t->G[s.src], t->grads + f.kernel,
                                   t->grads + f.bias, frames, t->num_cus, t->part, xform_of(s.src, &xa_tmp), ba, st);
        if (g > 0) {
          fused_done = true;
          if (want_sums) { fused_sums[pl] = g; sums_from_x[pl] = 1; }
        }
      }
    }
    // dW and dbias = sum dz (the MFMA wgrad kernel produces both)
    if (fused_done) {
      // both gradients are out
    } else if (t->use_mfma && s.kh == 1 && tm_wgrad(f.cin, s.kw, s.cout, conv_in(s.src), dsrc, t->grads + f.kernel,
                                             t->grads + f.bias, frames, t->num_cus, xform_of(s.src, &xa_tmp), ba, st)) {
      // MFMA path
    } else if (first_mfma && first_wgrad(s, x_dev, dsrc, t->grads + f.kernel, t->grads + f.bias, frames, T, t->num_cus, ba, st)) {
      // MFMA path, first layer
    } else if (t->use_mfma && t->pk_fin && is_output_layer(s, f.cin)) {
      fin_wgrad(f.cin, tensor(s.src), dsrc, t->grads + f.kernel, t->grads + f.bias, frames, t->num_cus, st);
    } else {
      if ((s.src > 0 && t->virt[s.src]) || fuse_dz)   // the direct kernel needs the activation and dz in HBM
        return rced_fail(RCED_ERR_STATE, "layer %d: no MFMA wgrad kernel for a layer with fused activation / dz", l);
      if (int rc = reduce_channels(t, t->D, t->D, nullptr, nullptr, P, s.cout, st)) return rc;
      hipLaunchKernelGGL(sums_to_float, dim3(1), dim3(64), 0, st, (const double*)t->sums, s.cout, 0, t->grads + f.bias);
      const int fpw = 16;
      const size_t lds = ((size_t)s.kh * (F + s.kw - 1) * f.cin + (size_t)F * s.cout) * sizeof(float);
      if (f.K * s.cout > train::kWgradMaxOut * train::kThreads || lds > 64 * 1024)
        return rced_fail(RCED_ERR_ARG, "layer %d too large for conv_wgrad", l);
      hipLaunchKernelGGL(train::conv_wgrad, dim3((frames + fpw - 1) / fpw), dim3(train::kThreads), lds, st, tensor(s.src),
                         (const float*)t->D, T, F, f.cin, s.cout, s.kh, s.kw, frames, fpw, t->grads + f.kernel);
    }
    // dx into G[src] (+=), as a forward conv of dz with the flipped / transposed kernel and the other SAME half
    if (s.src > 0 && !fused_done) {
      if (t->use_mfma && t->pk_fin_bwd && is_output_layer(s, f.cin) && consumers[s.src] == 1) {
        fin_dgrad(f.cin, dsrc, t->params + f.kernel, t->pk_fin_bwd, t->G[s.src], frames, st, t->use_x6);   // overwrites G[src]
      } else if (const int pl = s.src - 1;   // the layer that produced this dgrad's output tensor
                 fuse_sums_on && t->use_mfma && t->pk_bwd[l] && overwrite(l) && sums_in_dgrad_ok(pl) && [&] {
                   const LayerOff& pf = t->off[pl];
                   const tmm::SumArgs sa{t->z[pl], t->mu[pl], t->rstd[pl], t->params + pf.gamma, t->params + pf.beta};
                   fused_sums[pl] = tm_conv(false, s.cout, s.kw, f.cin, false, false, dsrc, t->pk_bwd[l], t->G[s.src], frames,
                                            t->num_cus, t->part, nullptr, ba, st, &sa);
                   return fused_sums[pl] > 0;
                 }()) {
        // MFMA path; layer pl's BatchNorm-backward sums come out of the same kernel
      } else if (const int pl = s.src - 1;   // an accumulating dgrad that adds the LAST contribution to G[src] -- the conv
                 // consumer of a tensor comes before its skip consumers in the net, so after them here -- sees the complete
                 // gradient in its epilogue: layer pl's sums come out of it as well (8-channel tensors: CR-CED's skip sources)
                 fuse_sums_on && t->use_mfma && t->pk_bwd[l] && !overwrite(l) && ba && written[s.src] &&
                 conv_consumer_of(s.src) == l && sums_in_dgrad_ok(pl) && [&] {
                   const LayerOff& pf = t->off[pl];
                   const tmm::SumArgs sa{t->z[pl], t->mu[pl], t->rstd[pl], t->params + pf.gamma, t->params + pf.beta};
                   fused_sums[pl] = tm_conv(false, s.cout, s.kw, f.cin, true, false, dsrc, t->pk_bwd[l], t->G[s.src], frames,
                                            t->num_cus, t->part, nullptr, ba, st, &sa, alias_src[s.src]);
                   return fused_sums[pl] > 0;
                 }()) {
        // MFMA path; sums of layer pl included
      } else if (t->use_mfma && t->pk_bwd[l] && (overwrite(l) || ensure_zero(s.src) == RCED_OK) &&
          tm_conv(false, s.cout, s.kw, f.cin, !overwrite(l), false, dsrc, t->pk_bwd[l], t->G[s.src], frames, t->num_cus,
                  nullptr, nullptr, ba, st, nullptr, overwrite(l) ? nullptr : alias_src[s.src])) {
        // MFMA path
      } else if (alias_src[s.src]) {
        return rced_fail(RCED_ERR_STATE, "layer %d: no accumulating MFMA dgrad kernel for an aliased skip gradient", l);
      } else if (fuse_dz) {
        return rced_fail(RCED_ERR_STATE, "layer %d: no MFMA dgrad kernel for a layer with fused dz", l);
      } else if (int rc0 = ensure_zero(s.src)) {
        return rc0;
      } else if (int rc = launch_conv(t->D, t->G[s.src], t->wt[l], t->zero32, t->G[s.src], frames, T, F, s.cout, f.cin,
                                      f.cin4, s.kh, s.kw, (s.kh - 1) - (s.kh - 1) / 2, (s.kw - 1) - (s.kw - 1) / 2, st)) {
        return rc;
      }
    }
  }
  if (t->wg_error) {   // (cannot happen with the up-front size above unless the runtime reports > 4 workgroups per CU)
    (void)hipStreamSynchronize(st);   // nothing of this step is left in flight behind the error
    return rced_fail(RCED_ERR_ALLOC, "deterministic weight gradients: the per-wave slice buffer could not be grown; the step was "
                                     "abandoned before the optimizer ran, but the moving statistics have advanced by this batch");
  }
  // ---- Adam (TF form), trainer.py:175-179
  // global_step: the reference fetches it in the same sess.run as train_op (trainer.py:186-191); TF1 orders a read and an
  // assign_add in one run only through control dependencies, and slim.learning.create_train_op makes train_op =
  // (increment global_step, then return the loss), so the fetched value is unordered against the increment in principle.
  // This library returns the POST-increment value, deterministically: after the first step train_step returns 1 and
  // the loop's next learning rate is noam(1) (trainer.py:215, step = global_step + 1 = 2).  See DESIGN.md 3.5.
  t->global_step += 1;
  const double tt = (double)t->global_step;
  const float lr_t = (float)((double)lr * std::sqrt(1.0 - std::pow((double)kAdamB2, tt)) / (1.0 - std::pow((double)kAdamB1, tt)));
  hipLaunchKernelGGL(train::adam_step, dim3((unsigned)((t->nvars + train::kThreads - 1) / train::kThreads)),
                     dim3(train::kThreads), 0, st, t->params, (const float*)t->grads, t->m, t->v,
                     (const unsigned char*)t->trainable, t->nvars, lr_t, kAdamB1, kAdamB2, kAdamEps);
  {
    TinyScanArgs ta;
    for (int l = 0; l < kMaxLayers; ++l) {
      ta.gamma_off[l] = l < L && net.layer[l].use_norm ? (int)t->off[l].gamma : -1;
      ta.cout[l] = l < L ? net.layer[l].cout : 0;
    }
    hipLaunchKernelGGL(tiny_gamma_scan, dim3(kMaxLayers), dim3(64), 0, st, (const float*)t->params, ta, t->tiny_dev);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(st));
  double loss = 0.0;
  for (double v : hp) loss += v;
  if (loss_out) *loss_out = loss / (double)t->batch_size;
  return RCED_OK;
}
}  // namespace

extern "C" {

int rced_train_step(rced_trainer* t, const float* x_dev, const float* y_dev, int N, int T, float lr, double* loss_out,
                    void* stream) {
  if (!y_dev) return rced_fail(RCED_ERR_ARG, "bad batch");
  return train_run(t, x_dev, y_dev, nullptr, N, T, lr, loss_out, stream);
}

int rced_train_forward(rced_trainer* t, const float* x_dev, float* pred_dev, int N, int T, void* stream) {
  if (!pred_dev) return rced_fail(RCED_ERR_ARG, "bad batch");
  return train_run(t, x_dev, nullptr, pred_dev, N, T, 0.f, nullptr, stream);
}

// module.py:11-34 with is_training=True, as ONE op on device pointers: z = conv(x) + bias; BatchNorm with the statistics of
// this batch (mean and biased variance over N*T*F, eps 1e-3 -- tf.layers.batch_normalization(training=True)); + skip; ReLU.
// The layerwise kernels of the training step (direct convolution, fp64 channel sums), not the MFMA path: a single op has
// no packed-weight state to keep between calls.
int rced_conv_bn_relu_train(const float* x, float* y, const float* kernel, const float* bias, const float* gamma_beta,
                            const float* skip_input, int use_act, int N, int T, int F, int cin, int cout, int kh, int kw,
                            float* batch_mean_var_out, int device, void* stream) {
  if (N < 0 || T < 0 || F <= 0 || cin <= 0 || cout <= 0 || cout > train::kThreads || kh <= 0 || kw <= 0)
    return rced_fail(RCED_ERR_ARG, "bad shape");
  if (N == 0 || T == 0) return RCED_OK;
  if (!x || !y || !kernel || !bias || !gamma_beta) return rced_fail(RCED_ERR_ARG, "null pointer");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return rced_fail(RCED_ERR_HIP, "no HIP device visible (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return rced_fail(RCED_ERR_ARG, "device %d out of range", device);
  DeviceGuard g(device);
  if (!g.ok) return rced_fail(RCED_ERR_HIP, "hipSetDevice(%d) failed", device);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int frames = N * T, K = kh * kw * cin, cout4 = (cout + 3) & ~3;
  const size_t P = (size_t)frames * F, n = P * cout;
  // one scratch allocation: z | repacked kernel | bias4 | mu | rstd | two dummies for the moving statistics | partial sums | sums
  const size_t fl = n + (size_t)K * cout4 + cout4 + 4 * (size_t)cout;
  const size_t dbl = (size_t)kReduceGrid * cout * 2 + 2 * (size_t)cout;
  float* buf = nullptr;
  HIP_TRY(hipMalloc(&buf, ((fl + 1) & ~(size_t)1) * sizeof(float) + dbl * sizeof(double)));
  float *z = buf, *wf = z + n, *b4 = wf + (size_t)K * cout4, *mu = b4 + cout4, *rstd = mu + cout, *mm = rstd + cout, *mv = mm + cout;
  double* part = reinterpret_cast<double*>(buf + ((fl + 1) & ~(size_t)1));
  double* sums = part + (size_t)kReduceGrid * cout * 2;
  int rc = RCED_OK;
  hipLaunchKernelGGL(train::repack_fwd, dim3((K * cout4 + 255) / 256), dim3(256), 0, st, kernel, K, cout, cout4, wf);
  hipLaunchKernelGGL(train::repack_fwd, dim3(1), dim3(256), 0, st, bias, 1, cout, cout4, b4);
  rc = launch_conv(x, z, wf, b4, nullptr, frames, T, F, cin, cout, cout4, kh, kw, (kh - 1) / 2, (kw - 1) / 2, st);
  if (rc == RCED_OK) {
    hipLaunchKernelGGL(train::chan_reduce, dim3(kReduceGrid), dim3(train::kThreads), 0, st, (const float*)z, (const float*)z,
                       (const float*)nullptr, (const float*)nullptr, P, cout, part);
    hipLaunchKernelGGL(train::reduce_finish, dim3(2 * cout), dim3(train::kThreads), 0, st, (const double*)part, kReduceGrid, cout, sums);
    hipLaunchKernelGGL(train::bn_stats_finish, dim3((cout + 63) / 64), dim3(64), 0, st, (const double*)sums, (double)P, cout, kBnEps,
                       1.f, mu, rstd, mm, mv);    // momentum 1: the two dummies are left alone
    if (batch_mean_var_out)
      hipLaunchKernelGGL(train::batch_mean_var, dim3((cout + 63) / 64), dim3(64), 0, st, (const double*)sums, (double)P, cout,
                         batch_mean_var_out);
    auto blocks = dim3((unsigned)std::min<size_t>((n + train::kThreads - 1) / train::kThreads, 65535));
    hipLaunchKernelGGL(train::bn_act_fwd, blocks, dim3(train::kThreads), 0, st, (const float*)z, (const float*)mu, (const float*)rstd,
                       gamma_beta, gamma_beta + cout, skip_input, (const float*)nullptr, use_act, n, cout, y);
    if (hipGetLastError() != hipSuccess) rc = rced_fail(RCED_ERR_HIP, "single-op training launch failed");
  }
  (void)hipStreamSynchronize(st);
  (void)hipFree(buf);
  return rc;
}

}  // extern "C"
