// R-CED V2's shapes on the MFMA training kernels.  V2's channel counts (10 12 14 15 19 21 23 25 ...) are trained in the
// even-padded internal layout of train_api.hip (10 12 14 16 20 22 24 26 24 22 20 16 14 12 10), so every shape here
// is even.  A translation unit of its own: these 28 shapes are half of the training kernels' compile time.
#include <hip/hip_runtime.h>

#include "train_mfma_dispatch.h"

using namespace rced;

#define RCED_TM_FWD_V2(X)                                                                                      \
  X(10, 7, 12) X(12, 5, 14) X(14, 5, 16) X(16, 5, 20) X(20, 5, 22) X(22, 7, 24) X(24, 11, 26) X(26, 7, 24)      \
  X(24, 5, 22) X(22, 5, 20) X(20, 5, 16) X(16, 5, 14) X(14, 7, 12) X(12, 11, 10)
#define RCED_TM_BWD_V2(X)                                                                                      \
  X(12, 7, 10) X(14, 5, 12) X(16, 5, 14) X(20, 5, 16) X(22, 5, 20) X(24, 7, 22) X(26, 11, 24) X(24, 7, 26)      \
  X(22, 5, 24) X(20, 5, 22) X(16, 5, 20) X(14, 5, 16) X(12, 7, 14) X(10, 11, 12)

namespace {
RCED_TM_DEFINE_DISPATCH(_v2, RCED_TM_FWD_V2, RCED_TM_BWD_V2)
}  // namespace

int rced_tm_conv_v2(bool fwd, int cin, int taps, int cout, bool accum, bool stats, const float* in, const float* packet,
                    float* out, int frames, int cus, double* part, const tmm::XformArgs* xa, const tmm::BnBwdArgs* ba,
                    hipStream_t st, const tmm::SumArgs* sa, const float* acc_from) {
  return tm_conv_v2(fwd, cin, taps, cout, accum, stats, in, packet, out, frames, cus, part, xa, ba, st, sa, acc_from);
}
bool rced_tm_has_v2(bool fwd, int cin, int taps, int cout) { return tm_has_v2(fwd, cin, taps, cout); }
int rced_tm_wgrad_v2(int cin, int taps, int cout, const float* x, const float* dz, float* dW, float* dbias, int frames, int cus,
                     const tmm::XformArgs* xa, const tmm::BnBwdArgs* ba, hipStream_t st) {
  return tm_wgrad_v2(cin, taps, cout, x, dz, dW, dbias, frames, cus, xa, ba, st);
}
