// C ABI of the STFT front-end / ISTFT rebuild (include/rced.h, "audio" section): host side.
#include <hip/hip_runtime.h>

#include <cmath>
#include <mutex>
#include <vector>

#include "../../include/rced.h"
#include "kernels_audio.h"
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
  if (N < 0 || L < 0 || T < 0) return rced_fail(RCED_ERR_ARG, "negative shape");
  if (N == 0 || T == 0) return RCED_OK;
  if (L == 0) return rced_fail(RCED_ERR_ARG, "empty signals (L = 0) with T > 0");
  if (!pcm_dev || !mag_dev) return rced_fail(RCED_ERR_ARG, "null pointer");
  if (N > 65535) return rced_fail(RCED_ERR_ARG, "N > 65535 utterances per call");
  DeviceGuard g(device);
  AudioTables* t = nullptr;
  if (int rc = tables(device, &t)) return rc;
  if (!g.ok) return rced_fail(RCED_ERR_HIP, "hipSetDevice(%d) failed", device);
  const dim3 grid((T + audio::kFramesPerWg - 1) / audio::kFramesPerWg, N);
  hipLaunchKernelGGL(audio::stft_kernel, grid, dim3(audio::kThreads), 0, static_cast<hipStream_t>(stream), pcm_dev,
                     lengths_dev, (const float*)t->stft, L, T, mag_dev, phase_dev);
  HIP_TRY(hipGetLastError());
  return RCED_OK;
}

int rced_istft(const float* mag_dev, const float* phase_dev, int N, int T, int nfft, float* audio_dev, int device,
               void* stream) {
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
  const dim3 grid((T + audio::kFramesPerWg - 1) / audio::kFramesPerWg, N);
  hipLaunchKernelGGL(audio::istft_frames_kernel, grid, dim3(audio::kThreads), 0, st, mag_dev, phase_dev,
                     (const float*)(nfft == 512 ? t->istft512 : t->istft256), T, audio_dev);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(audio::deemphasis_kernel, dim3(N), dim3(audio::kThreads), 0, st, audio_dev,
                     (T + 1) * audio::kStep);
  HIP_TRY(hipGetLastError());
  return RCED_OK;
}

}  // extern "C"
