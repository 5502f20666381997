// C ABI of the STFT front-end / ISTFT rebuild (include/rced.h, "audio" section): host side.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <mutex>
#include <vector>

#include "../../include/rced.h"
#include <cstdlib>
#include <cstring>

#include "kernels_audio.h"
#include "kernels_audio_x6.h"
#include "rced_internal.h"

using namespace rced;

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return rced_fail(e_ == hipErrorOutOfMemory ? RCED_ERR_ALLOC : RCED_ERR_HIP, "%s: %s", #expr, \
                       hipGetErrorString(e_));                                                 \
  } while (0)

namespace {

constexpr int kMaxDevices = 16;
struct AudioTables {   // per device, built once
  float* stft = nullptr;
  float* istft512 = nullptr;
  float* istft256 = nullptr;
  // the three-part bf16 form (kernels_audio_x6.h): A fragments, the rank-1 column of im(bin 128), the head table
  unsigned short* stft_x6 = nullptr;
  unsigned short* istft_x6[2] = {nullptr, nullptr};   // [nfft == 512]
  float* cim[2] = {nullptr, nullptr};
  float* chead[2] = {nullptr, nullptr};
};
AudioTables g_tab[kMaxDevices];
std::mutex g_mu;

double hamming(int k) { return 0.54 - 0.46 * std::cos(2.0 * M_PI * k / (audio::kFrame - 1)); }   // np.hamming(256)

std::vector<float> pack_stft() {   // [st][mt][lane][e]; row m = 2b + {0: re, 1: im}; window folded in
  std::vector<float> p(audio::kStftPack, 0.f);
  for (int st = 0; st < audio::kStftSteps; ++st)
    for (int mt = 0; mt < audio::kStftMT; ++mt)
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 2; ++e) {
          const int m = 16 * mt + (lane & 15), k = 8 * st + 2 * (lane >> 4) + e, b = m >> 1;
          if (b >= audio::kBins) continue;
          const double th = 2.0 * M_PI * ((b * k) % audio::kFrame) / audio::kFrame;
          p[((size_t)st * audio::kStftMT + mt) * 128 + lane * 2 + e] =
              (float)(hamming(k) * ((m & 1) ? -std::sin(th) : std::cos(th)));
        }
  return p;
}

std::vector<float> pack_istft(int nfft) {   // [st][mt][lane][e]; row = sample n; k = 2b + {re, im}
  std::vector<float> p(audio::kIstftPack, 0.f);
  for (int st = 0; st < audio::kIstftSteps; ++st)
    for (int mt = 0; mt < audio::kIstftMT; ++mt)
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 2; ++e) {
          const int n = 16 * mt + (lane & 15), k = 8 * st + 2 * (lane >> 4) + e, b = k >> 1, c = k & 1;
          if (b >= audio::kBins) continue;
          // numpy.fft.irfft: bin 0 (and the Nyquist bin nfft/2) count once and lose their imaginary part
          const bool edge = b == 0 || 2 * b == nfft;
          if (edge && c) continue;
          const double th = 2.0 * M_PI * ((long long)b * n % nfft) / nfft;
          const double coef = (edge ? 1.0 : 2.0) / nfft / hamming(n);   // de_window folded in
          p[((size_t)st * audio::kIstftMT + mt) * 128 + lane * 2 + e] = (float)(coef * (c ? -std::sin(th) : std::cos(th)));
        }
  return p;
}

inline unsigned short bf16_rne(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0;
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
inline void split3(float v, unsigned short* h, unsigned short* mm, unsigned short* l) {   // v = h + m + l to 2^-24
  auto b2f = [](unsigned short b) { const unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; };
  *h = bf16_rne(v);
  const float r1 = v - b2f(*h);
  *mm = bf16_rne(r1);
  *l = bf16_rne(r1 - b2f(*mm));
}
// [mt][chunk][part][lane][8] bf16 from a coefficient function coef(row, k), k = 32 chunk + 8 (lane >> 4) + e
template <class F>
std::vector<unsigned short> pack_x6(int mtiles, F coef) {
  std::vector<unsigned short> p((size_t)mtiles * audio::x6::kPackPerMT, 0);
  for (int mt = 0; mt < mtiles; ++mt)
    for (int c = 0; c < audio::x6::kChunks; ++c)
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 8; ++e) {
          const int row = 16 * mt + (lane & 15), k = 32 * c + 8 * (lane >> 4) + e;
          const size_t at = (size_t)mt * audio::x6::kPackPerMT + ((size_t)(c * 3) * 64 + lane) * 8 + e;
          split3((float)coef(row, k), &p[at], &p[at + 512], &p[at + 1024]);
        }
  return p;
}
// STFT rows (kernels_audio_x6.h): 0 = re(0), 1 = re(128), 2b / 2b + 1 = re / im of bin b
double stft_coef(int row, int k) {
  const int b = row == 1 ? 128 : row >> 1;
  const bool im = row >= 2 && (row & 1);
  const double th = 2.0 * M_PI * ((b * k) % audio::kFrame) / audio::kFrame;
  return hamming(k) * (im ? -std::sin(th) : std::cos(th));
}
// ISTFT coefficient of spectrum term (bin b, c = 0 re / 1 im) for output sample n (numpy.fft.irfft: bin 0 and the Nyquist bin nfft / 2
// count once and lose their imaginary part; 1 / nfft and the 1 / hamming de-window folded in)
double istft_coef(int nfft, int n, int b, int c) {
  const bool edge = b == 0 || 2 * b == nfft;
  if (edge && c) return 0.0;
  const double th = 2.0 * M_PI * ((long long)b * n % nfft) / nfft;
  return (edge ? 1.0 : 2.0) / nfft / hamming(n) * (c ? -std::sin(th) : std::cos(th));
}

template <class T>
int upload_raw(T** dev, const std::vector<T>& host) {
  T* p = nullptr;
  HIP_TRY(hipMalloc(&p, host.size() * sizeof(T)));
  const hipError_t e = hipMemcpy(p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(p);
    return rced_fail(RCED_ERR_HIP, "hipMemcpy(audio tables): %s", hipGetErrorString(e));
  }
  *dev = p;
  return RCED_OK;
}

int build_x6(AudioTables& t) {
  if (int rc = upload_raw(&t.stft_x6, pack_x6(audio::x6::kStftMTx, stft_coef))) return rc;
  for (int v = 0; v < 2; ++v) {
    const int nfft = v ? 512 : 256;
    // rows = samples 128..255; slot k = 2b + c, slot 1 = re of bin 128
    auto coef = [&](int row, int k) { return k == 1 ? istft_coef(nfft, 128 + row, 128, 0) : istft_coef(nfft, 128 + row, k >> 1, k & 1); };
    if (int rc = upload_raw(&t.istft_x6[v], pack_x6(audio::x6::kIstftMTx, coef))) return rc;
    std::vector<float> cim(128), head((size_t)2 * audio::kBins * 128);
    for (int r = 0; r < 128; ++r) cim[r] = (float)istft_coef(nfft, 128 + r, 128, 1);
    for (int k = 0; k < 2 * audio::kBins; ++k)
      for (int n = 0; n < 128; ++n) head[(size_t)k * 128 + n] = (float)istft_coef(nfft, n, k >> 1, k & 1);
    if (int rc = upload_raw(&t.cim[v], cim)) return rc;
    if (int rc = upload_raw(&t.chead[v], head)) return rc;
  }
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(audio::x6::istft_x6_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     audio::x6::kIstftLdsBytes);
  if (e == hipSuccess)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(audio::x6::istft_x6_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            audio::x6::kIstftLdsBytes);
  if (e != hipSuccess) return rced_fail(RCED_ERR_HIP, "hipFuncSetAttribute(istft LDS): %s", hipGetErrorString(e));
  return RCED_OK;
}

// frame-range split of an utterance over workgroups: enough workgroups for every CU when the batch is small
int range_split(int N, int T) {
  const int nblk = (T + audio::kFramesPerWg - 1) / audio::kFramesPerWg;
  int s = 1;
  while (s < nblk && N * s < 256) s *= 2;
  return s < nblk ? s : nblk;
}

int upload(float** dev, const std::vector<float>& host) {
  HIP_TRY(hipMalloc(dev, host.size() * sizeof(float)));
  HIP_TRY(hipMemcpy(*dev, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
  return RCED_OK;
}

int tables(int device, AudioTables** out) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return rced_fail(RCED_ERR_HIP, "no HIP device visible (this library has no CPU fallback)");
  if (device < 0 || device >= n || device >= kMaxDevices) return rced_fail(RCED_ERR_ARG, "device %d out of range", device);
  std::lock_guard<std::mutex> lk(g_mu);
  AudioTables& t = g_tab[device];
  if (!t.stft) {
    if (int rc = upload(&t.stft, pack_stft())) return rc;
    if (int rc = upload(&t.istft512, pack_istft(512))) return rc;
    if (int rc = upload(&t.istft256, pack_istft(256))) return rc;
    if (int rc = build_x6(t)) return rc;
  }
  *out = &t;
  return RCED_OK;
}

struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) { prev = -1; return; }
    ok = (prev == dev) || (hipSetDevice(dev) == hipSuccess);
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace

extern "C" {

int rced_stft_num_frames(int length) { return length > 0 ? audio::num_frames(length) : 0; }

int rced_stft(const float* pcm_dev, const int* lengths_dev, int N, int L, int T, float* mag_dev, float* phase_dev,
              int device, void* stream) {
  return rced_stft_ex(pcm_dev, lengths_dev, N, L, T, mag_dev, phase_dev, device, stream, RCED_AUDIO_X6);
}

int rced_stft_ex(const float* pcm_dev, const int* lengths_dev, int N, int L, int T, float* mag_dev, float* phase_dev,
                 int device, void* stream, int kernels) {
  if (kernels != RCED_AUDIO_X6 && kernels != RCED_AUDIO_F32) return rced_fail(RCED_ERR_ARG, "kernels must be RCED_AUDIO_X6 (1) or RCED_AUDIO_F32 (0), got %d", kernels);
  if (N < 0 || L < 0 || T < 0) return rced_fail(RCED_ERR_ARG, "negative shape");
  if (N == 0 || T == 0) return RCED_OK;
  if (L == 0) return rced_fail(RCED_ERR_ARG, "empty signals (L = 0) with T > 0");
  if (!pcm_dev || !mag_dev) return rced_fail(RCED_ERR_ARG, "null pointer");
  if (N > 65535) return rced_fail(RCED_ERR_ARG, "N > 65535 utterances per call");
  DeviceGuard g(device);
  AudioTables* t = nullptr;
  if (int rc = tables(device, &t)) return rc;
  if (!g.ok) return rced_fail(RCED_ERR_HIP, "hipSetDevice(%d) failed", device);
  if (kernels == RCED_AUDIO_X6) {
    hipLaunchKernelGGL(audio::x6::stft_x6_kernel, dim3(N, 2, range_split(N, T)), dim3(audio::x6::kThreadsX), 0, static_cast<hipStream_t>(stream),
                       pcm_dev, lengths_dev, (const unsigned short*)t->stft_x6, L, T, mag_dev, phase_dev);
  } else {
    const dim3 grid((T + audio::kFramesPerWg - 1) / audio::kFramesPerWg, N);
    hipLaunchKernelGGL(audio::stft_kernel, grid, dim3(audio::kThreads), 0, static_cast<hipStream_t>(stream), pcm_dev,
                       lengths_dev, (const float*)t->stft, L, T, mag_dev, phase_dev);
  }
  HIP_TRY(hipGetLastError());
  return RCED_OK;
}

int rced_istft(const float* mag_dev, const float* phase_dev, int N, int T, int nfft, float* audio_dev, int device,
               void* stream) {
  return rced_istft_ex(mag_dev, phase_dev, N, T, nfft, audio_dev, device, stream, RCED_AUDIO_X6);
}

int rced_istft_ex(const float* mag_dev, const float* phase_dev, int N, int T, int nfft, float* audio_dev, int device,
                  void* stream, int kernels) {
  if (kernels != RCED_AUDIO_X6 && kernels != RCED_AUDIO_F32) return rced_fail(RCED_ERR_ARG, "kernels must be RCED_AUDIO_X6 (1) or RCED_AUDIO_F32 (0), got %d", kernels);
  if (N < 0 || T < 0) return rced_fail(RCED_ERR_ARG, "negative shape");
  if (nfft != 512 && nfft != 256) return rced_fail(RCED_ERR_ARG, "nfft must be 512 (reference default) or 256");
  if (N == 0 || T == 0) return RCED_OK;
  if (!mag_dev || !phase_dev || !audio_dev) return rced_fail(RCED_ERR_ARG, "null pointer");
  if (N > 65535) return rced_fail(RCED_ERR_ARG, "N > 65535 utterances per call");
  DeviceGuard g(device);
  AudioTables* t = nullptr;
  if (int rc = tables(device, &t)) return rc;
  if (!g.ok) return rced_fail(RCED_ERR_HIP, "hipSetDevice(%d) failed", device);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (kernels == RCED_AUDIO_X6) {
    const int v = nfft == 512, split = range_split(N, T);
    if (split == 1) {   // one workgroup per utterance: de_frame and de_emphasis inside the kernel
      hipLaunchKernelGGL(audio::x6::istft_x6_kernel<true>, dim3(N, 1, 1), dim3(audio::x6::kThreadsX), audio::x6::kIstftLdsBytes, st, mag_dev, phase_dev,
                         (const unsigned short*)t->istft_x6[v], (const float*)t->cim[v], (const float*)t->chead[v], T, audio_dev);
      HIP_TRY(hipGetLastError());
      return RCED_OK;
    }
    hipLaunchKernelGGL(audio::x6::istft_x6_kernel<false>, dim3(N, 1, split), dim3(audio::x6::kThreadsX), audio::x6::kIstftLdsBytes, st, mag_dev,
                       phase_dev, (const unsigned short*)t->istft_x6[v], (const float*)t->cim[v], (const float*)t->chead[v], T, audio_dev);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(audio::x6::istft_head_kernel, dim3(N), dim3(audio::kStep), 0, st, mag_dev, phase_dev, (const float*)t->chead[v], T, audio_dev);
  } else {
    const dim3 grid((T + audio::kFramesPerWg - 1) / audio::kFramesPerWg, N);
    hipLaunchKernelGGL(audio::istft_frames_kernel, grid, dim3(audio::kThreads), 0, st, mag_dev, phase_dev,
                       (const float*)(nfft == 512 ? t->istft512 : t->istft256), T, audio_dev);
  }
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(audio::deemphasis_kernel, dim3(N), dim3(audio::kThreads), 0, st, audio_dev,
                     (T + 1) * audio::kStep);
  HIP_TRY(hipGetLastError());
  return RCED_OK;
}

}  // extern "C"
