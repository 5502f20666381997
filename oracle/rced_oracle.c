/*
 * ORACLE (test infrastructure, NOT product code): plain-C restatement of the
 * R-CED / CR-CED forward pass.
 *
 * PARITY UNPINNED.  The reference path is TensorFlow 1.14 graph code
 * (/root/reference/model_utils/module.py:11-34, model_utils/model.py:6-96); TF is
 * not installable in the build container and the reference has no tests, so this
 * file follows the reference source plus TF's documented defaults:
 *   conv2d: NHWC x HWIO, stride 1, padding SAME (k-1 total, floor half first:
 *           k=8 -> 3 before / 4 after), use_bias=True          (module.py:27)
 *   batch_normalization(training=False): (y-mean)*gamma*rsqrt(var+1e-3)+beta
 *                                                              (module.py:29)
 *   + skip_input, then ReLU                                    (module.py:30-33)
 *   V3 block skip added AFTER the ReLU                         (model.py:75-76)
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * the library built from this file.  Build: make -C oracle  ->  oracle/librced_oracle.so
 *
 * Two instantiations: REAL=double (the checker) and REAL=float (the timed CPU
 * baseline, "port").  OpenMP over frames; the thread count used is reported by
 * oracle_num_threads().
 */
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_MAX_LAYERS 32
#define ORACLE_MAX_COUT 64

/* One layer descriptor = 9 ints: cout kh kw use_norm use_act src skip_pre skip_post cin */
#define DESC_STRIDE 9

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void oracle_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

#define DEFINE_LAYER(NAME, REAL)                                                                   \
  /* x:[N,T,F,cin] kernel:[kh,kw,cin,cout] bias:[cout] bn: gamma,beta,mean,var (4*cout) or NULL */ \
  void NAME(const REAL* x, int N, int T, int F, int cin, const float* kernel, const float* bias,   \
            const float* bn, const REAL* skip_pre, int use_act, const REAL* skip_post, int kh,     \
            int kw, int cout, REAL* y) {                                                           \
    const int pt = (kh - 1) / 2, pl = (kw - 1) / 2; /* SAME: floor half before */                  \
    REAL scale[ORACLE_MAX_COUT], shift_mean[ORACLE_MAX_COUT], beta[ORACLE_MAX_COUT];               \
    for (int c = 0; c < cout; ++c) {                                                               \
      if (bn) {                                                                                    \
        REAL g = bn[c], b = bn[cout + c], m = bn[2 * cout + c], v = bn[3 * cout + c];              \
        REAL inv = (REAL)1 / (sizeof(REAL) == 8 ? (REAL)__builtin_sqrt((double)(v + (REAL)1e-3))   \
                                                : (REAL)__builtin_sqrtf((float)(v + (REAL)1e-3))); \
        scale[c] = g * inv;                                                                        \
        shift_mean[c] = m;                                                                         \
        beta[c] = b;                                                                               \
      } else {                                                                                     \
        scale[c] = 1;                                                                              \
        shift_mean[c] = 0;                                                                         \
        beta[c] = 0;                                                                               \
      }                                                                                            \
    }                                                                                              \
    _Pragma("omp parallel for collapse(2) schedule(static)") for (int n = 0; n < N; ++n) {         \
      for (int t = 0; t < T; ++t) {                                                                \
        REAL acc[ORACLE_MAX_COUT];                                                                 \
        for (int f = 0; f < F; ++f) {                                                              \
          for (int c = 0; c < cout; ++c) acc[c] = 0;                                               \
          for (int i = 0; i < kh; ++i) {                                                           \
            const int tt = t + i - pt;                                                             \
            if (tt < 0 || tt >= T) continue; /* zero padding */                                    \
            for (int j = 0; j < kw; ++j) {                                                         \
              const int ff = f + j - pl;                                                           \
              if (ff < 0 || ff >= F) continue;                                                     \
              const REAL* xp = x + (((size_t)n * T + tt) * F + ff) * cin;                          \
              const float* wp = kernel + ((size_t)(i * kw + j) * cin) * cout;                      \
              for (int ci = 0; ci < cin; ++ci) {                                                   \
                const REAL xv = xp[ci];                                                            \
                const float* w = wp + (size_t)ci * cout;                                           \
                for (int c = 0; c < cout; ++c) acc[c] += xv * (REAL)w[c];                          \
              }                                                                                    \
            }                                                                                      \
          }                                                                                        \
          const size_t o = (((size_t)n * T + t) * F + f) * cout;                                   \
          for (int c = 0; c < cout; ++c) {                                                         \
            REAL v = acc[c] + (REAL)bias[c];                                                       \
            if (bn) v = (v - shift_mean[c]) * scale[c] + beta[c];                                  \
            if (skip_pre) v += skip_pre[o + c];                                                    \
            if (use_act) v = v > 0 ? v : 0;                                                        \
            if (skip_post) v += skip_post[o + c];                                                  \
            y[o + c] = v;                                                                          \
          }                                                                                        \
        }                                                                                          \
      }                                                                                            \
    }                                                                                              \
  }

DEFINE_LAYER(oracle_conv_bn_relu_f64, double)
DEFINE_LAYER(oracle_conv_bn_relu_f32, float)

/* blob: per layer kernel, bias, then (if use_norm) gamma, beta, moving_mean, moving_variance.
 * desc: n_layers * DESC_STRIDE ints.  x:[N,T,F,1] float32 in, y:[N,T,F,1] out.
 * Returns 0, or -1 on a malformed descriptor / allocation failure. */
#define DEFINE_FORWARD(NAME, LAYER, REAL)                                                        \
  int NAME(const int* desc, int n_layers, const float* blob, const float* x, int N, int T,       \
           int F, REAL* y) {                                                                     \
    if (n_layers <= 0 || n_layers > ORACLE_MAX_LAYERS) return -1;                                \
    REAL* tens[ORACLE_MAX_LAYERS + 1];                                                           \
    int chans[ORACLE_MAX_LAYERS + 1];                                                            \
    int last_use[ORACLE_MAX_LAYERS + 1];                                                         \
    memset(tens, 0, sizeof(tens));                                                               \
    const size_t px = (size_t)N * T * F;                                                         \
    for (int i = 0; i <= n_layers; ++i) last_use[i] = -1;                                        \
    for (int l = 0; l < n_layers; ++l) {                                                         \
      const int* d = desc + l * DESC_STRIDE;                                                     \
      if (d[5] < 0 || d[5] > l || d[6] > l || d[7] > l || d[0] > ORACLE_MAX_COUT) return -1;     \
      last_use[d[5]] = l;                                                                        \
      if (d[6] >= 0) last_use[d[6]] = l;                                                         \
      if (d[7] >= 0) last_use[d[7]] = l;                                                         \
    }                                                                                            \
    tens[0] = (REAL*)malloc(px * sizeof(REAL));                                                  \
    if (!tens[0]) return -1;                                                                     \
    for (size_t i = 0; i < px; ++i) tens[0][i] = (REAL)x[i];                                     \
    chans[0] = 1;                                                                                \
    const float* w = blob;                                                                       \
    int rc = 0;                                                                                  \
    for (int l = 0; l < n_layers && rc == 0; ++l) {                                              \
      const int* d = desc + l * DESC_STRIDE;                                                     \
      const int cout = d[0], kh = d[1], kw = d[2], use_norm = d[3], use_act = d[4];              \
      const int src = d[5], sp = d[6], so = d[7], cin = d[8];                                    \
      if (cin != chans[src]) { rc = -1; break; }                                                 \
      const float* kernel = w;  w += (size_t)kh * kw * cin * cout;                               \
      const float* bias = w;    w += cout;                                                       \
      const float* bn = NULL;                                                                    \
      if (use_norm) { bn = w; w += 4 * cout; }                                                   \
      REAL* out = (l == n_layers - 1) ? y : (REAL*)malloc(px * cout * sizeof(REAL));            \
      if (!out) { rc = -1; break; }                                                              \
      LAYER(tens[src], N, T, F, cin, kernel, bias, bn, sp >= 0 ? tens[sp] : NULL, use_act,       \
            so >= 0 ? tens[so] : NULL, kh, kw, cout, out);                                       \
      tens[l + 1] = out;                                                                         \
      chans[l + 1] = cout;                                                                       \
      for (int i = 0; i <= l; ++i)                                                               \
        if (tens[i] && last_use[i] <= l) { free(tens[i]); tens[i] = NULL; }                      \
    }                                                                                            \
    for (int i = 0; i <= n_layers; ++i)                                                          \
      if (tens[i] && tens[i] != y) free(tens[i]);                                                \
    return rc;                                                                                   \
  }

DEFINE_FORWARD(oracle_forward_f64, oracle_conv_bn_relu_f64, double)
DEFINE_FORWARD(oracle_forward_f32, oracle_conv_bn_relu_f32, float)
