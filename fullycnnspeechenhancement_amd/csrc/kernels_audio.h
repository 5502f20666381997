// STFT front-end and ISTFT rebuild around the CNN, on device.
//   N1  AudioFeature.compute_spectrogram + power_spectrum + divide_phase
//       data_utils/audio_feature.py:22-44, 47-55, 58-77, 79-88, 91-99, 102-115
//   N2  AudioReBuild.rebuild_audio
//       model_utils/utils.py:171-183 (merge 119-126, irfft 115-117, de_window 128-137, de_frame 139-147,
//       de_emphasis 104-113)
// Fixed to the configuration every reference cfg uses: 8 kHz, 32 ms window (256 samples), 16 ms stride
// (128 samples), hamming, rfft(256) -> 129 bins.
//
// Both transforms are dense GEMMs on v_mfma_f32_16x16x4_f32 (frames on N, outputs on M):
//   STFT : D[2b + {re,im}][frame] = sum_k A[2b+..][k] * e[128*frame + k],  K = 256; the hamming window
//          is folded into A; e = pre-emphasised signal, staged once per 64 frames in LDS (frames overlap,
//          so it is one contiguous segment); re/im of a bin land in the same lane -> |.| and unit phase.
//   ISTFT: D[sample n][frame] = sum_k C[n][k] * X[k][frame],  K = 258 (re/im interleaved); 1/nfft, the
//          factor 2 of irfft and the 1/hamming[n] de-window are folded into C.  nfft = 512 reproduces the
//          reference as shipped (AudioReBuild() default); nfft = 256 is the true inverse.
//   de-emphasis y[i] = x[i] + 0.97 y[i-1] is a first-order linear recurrence -> blocked affine scan.
#pragma once
#include <hip/hip_runtime.h>

namespace rced {
namespace audio {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kFrame = 256, kStep = 128, kBins = 129;
constexpr float kPre = 0.97f;
constexpr int kThreads = 256;        // 4 waves
constexpr int kFramesPerWg = 64;     // 4 N-tiles of 16 frames

__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// STFT.  A-fragment pack: [s (32 b64-steps)][mt (17)][lane][2], row m = 16*mt + i:
//   m = 2b   -> w[k] * cos(2 pi b k / 256),   m = 2b+1 -> -w[k] * sin(2 pi b k / 256),   b < 129.
// ---------------------------------------------------------------------------------------------
constexpr int kStftMT = 17;                                  // 272 rows >= 258
constexpr int kStftSteps = kFrame / 8;                       // 32
constexpr int kStftPack = kStftSteps * kStftMT * 128;        // floats
constexpr int kSegSkew = 132;                                // LDS floats per 128 samples (conflict-free b64)
constexpr int kSegFloats = (kFramesPerWg + 1) * kSegSkew;    // 65 half-frames

// num_frames(L) = ceil(|L - 256| / 128 + 1)   (audio_feature.py:70)
__host__ __device__ inline int num_frames(int len) {
  const int d = len >= kFrame ? len - kFrame : kFrame - len;
  return (d + kStep - 1) / kStep + 1;
}

// pcm [N, L]; lengths [N] or nullptr (= L); mag [N, T, 129]; phase [N, T, 129, 2] or nullptr.
__global__ __launch_bounds__(kThreads) void stft_kernel(const float* __restrict__ pcm, const int* __restrict__ lengths,
                                                         const float* __restrict__ apack, int L, int T,
                                                         float* __restrict__ mag, float* __restrict__ phase) {
  __shared__ __attribute__((aligned(16))) float seg[kSegFloats];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, kq = lane >> 4;
  const int utt = blockIdx.y, f0 = blockIdx.x * kFramesPerWg;
  const int len = lengths ? min(lengths[utt], L) : L;
  const int nf = len > 0 ? num_frames(len) : 0;
  const float* s = pcm + (size_t)utt * L;
  // stage the pre-emphasised, zero-padded segment [128*f0, 128*(f0+65)): e[0] = s[0],
  // e[g] = s[g] - 0.97*s[g-1] as ONE float32 multiply and ONE float32 subtract (the reference
  // pre-emphasises in float32, audio_feature.py:54; an fma would differ in the last bit)
  for (int i = tid; i < (kFramesPerWg + 1) * kStep; i += kThreads) {
    const int g = f0 * kStep + i;
    float v = 0.f;
    if (g < len) v = g == 0 ? s[0] : __fsub_rn(s[g], __fmul_rn(kPre, s[g - 1]));
    seg[(i >> 7) * kSegSkew + (i & 127)] = v;
  }
  __syncthreads();

  // wave w owns M-tiles w, w+4, w+8, w+12 (+16 for wave 0)
  f32x4 acc[5][4];
#pragma unroll
  for (int j = 0; j < 5; ++j)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[j][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x2* ap = reinterpret_cast<const f32x2*>(apack) + lane;
  const bool five = wave == 0;
#pragma unroll 4
  for (int st = 0; st < kStftSteps; ++st) {
    f32x2 a[5], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = ap[(st * kStftMT + wave + 4 * j) * 64];
    a[4] = five ? ap[(st * kStftMT + 16) * 64] : f32x2{0.f, 0.f};
    // sample k = 8*st + 2*kq + e of frame (16*t + n): segment index 128*(16t+n) + k, skewed per 128
    const int k = 8 * st + 2 * kq;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int half = 16 * t + n + (k >> 7);
      b[t] = *reinterpret_cast<const f32x2*>(seg + half * kSegSkew + (k & 127));
    }
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j][t] = mfma(a[j][e], b[t][e], acc[j][t]);
        if (five) acc[4][t] = mfma(a[4][e], b[t][e], acc[4][t]);
      }
  }
  // epilogue: rows 4*kq.. of M-tile mt = (re, im) of bins 8*mt + 2*kq, +1
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    if (j == 4 && !five) break;
    const int mt = j < 4 ? wave + 4 * j : 16;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int fr = f0 + 16 * t + n;
      if (fr >= T) continue;
      const bool live = fr < nf;   // frames past the utterance: zero spectrum (padding_batch), phase 1+0j
      const f32x4 v = acc[j][t];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int b = 8 * mt + 2 * kq + h;
        if (b >= kBins) continue;
        const float re = live ? (h ? v.z : v.x) : 0.f, im = live ? (h ? v.w : v.y) : 0.f;
        const float m = sqrtf(re * re + im * im);
        const size_t o = ((size_t)utt * T + fr) * kBins + b;
        mag[o] = m;
        if (phase) {
          const float inv = m > 0.f ? 1.f / m : 0.f;
          *reinterpret_cast<f32x2*>(phase + 2 * o) = m > 0.f ? f32x2{re * inv, im * inv} : f32x2{1.f, 0.f};
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// ISTFT frames.  C-fragment pack: [s (33 b64-steps, K = 258 -> 264)][mt (16)][lane][2], row = sample n.
// ---------------------------------------------------------------------------------------------
constexpr int kIstftMT = 16;
constexpr int kIstftK = 2 * kBins;                            // 258
constexpr int kIstftSteps = (kIstftK + 7) / 8;                // 33
constexpr int kIstftPack = kIstftSteps * kIstftMT * 128;
constexpr int kXStride = 268;                                 // LDS floats per frame (264 + 4: conflict-free b64)

// mag [N,T,129], phase [N,T,129,2] -> x [N, (T+1)*128]: first half of frame 0, second half of every frame
__global__ __launch_bounds__(kThreads) void istft_frames_kernel(const float* __restrict__ mag,
                                                                 const float* __restrict__ phase,
                                                                 const float* __restrict__ cpack, int T,
                                                                 float* __restrict__ x) {
  __shared__ __attribute__((aligned(16))) float xs[kFramesPerWg * kXStride];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, kq = lane >> 4;
  const int utt = blockIdx.y, f0 = blockIdx.x * kFramesPerWg;
  // stage X[frame][2b] = mag*re, [2b+1] = mag*im (merge_magphase), zero past bin 128 and past T
  for (int i = tid; i < kFramesPerWg * (kXStride / 2); i += kThreads) {
    const int fr = i / (kXStride / 2), b = i - fr * (kXStride / 2);
    f32x2 v = {0.f, 0.f};
    if (b < kBins && f0 + fr < T) {
      const size_t o = ((size_t)utt * T + f0 + fr) * kBins + b;
      const float m = mag[o];
      const f32x2 p = *reinterpret_cast<const f32x2*>(phase + 2 * o);
      v = f32x2{m * p.x, m * p.y};
    }
    *reinterpret_cast<f32x2*>(xs + fr * kXStride + 2 * b) = v;
  }
  __syncthreads();
  // wave w owns M-tiles w, w+4, w+8, w+12 (samples 16*mt .. 16*mt+15)
  f32x4 acc[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[j][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x2* cp = reinterpret_cast<const f32x2*>(cpack) + lane;
#pragma unroll 3
  for (int st = 0; st < kIstftSteps; ++st) {
    f32x2 a[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = cp[(st * kIstftMT + wave + 4 * j) * 64];
#pragma unroll
    for (int t = 0; t < 4; ++t) b[t] = *reinterpret_cast<const f32x2*>(xs + (16 * t + n) * kXStride + 8 * st + 2 * kq);
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j][t] = mfma(a[j][e], b[t][e], acc[j][t]);
  }
  // de_frame (utils.py:139-147): out[128*t + n] for n >= 128, plus n < 128 of frame 0
  float* xo = x + (size_t)utt * (T + 1) * kStep;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n0 = 16 * (wave + 4 * j) + 4 * kq;   // first of this lane's 4 samples
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int fr = f0 + 16 * t + n;
      if (fr >= T) continue;
      if (n0 >= kStep || fr == 0) *reinterpret_cast<f32x4*>(xo + (size_t)fr * kStep + n0) = acc[j][t];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// De-emphasis (utils.py:104-113): y[0] = x[0], y[i] = x[i] + 0.97*y[i-1], in place on [N, len].
// One workgroup per utterance; chunks of 256 threads x 16 samples; thread-local scan, then an
// inclusive scan of the affine maps (decay, offset) across threads, then the carry is applied.
// ---------------------------------------------------------------------------------------------
constexpr int kDeBlock = 16;
__global__ __launch_bounds__(kThreads) void deemphasis_kernel(float* __restrict__ x, int len) {
  __shared__ float sa[kThreads], sb[kThreads];
  const int tid = threadIdx.x;
  float* xu = x + (size_t)blockIdx.x * len;
  float pw[kDeBlock + 1];
  pw[0] = 1.f;
#pragma unroll
  for (int j = 1; j <= kDeBlock; ++j) pw[j] = pw[j - 1] * kPre;
  float carry = 0.f;   // y just before this chunk
  for (int c0 = 0; c0 < len; c0 += kThreads * kDeBlock) {
    const int i0 = c0 + tid * kDeBlock;
    float y[kDeBlock];
    float run = 0.f;
#pragma unroll
    for (int j = 0; j < kDeBlock; ++j) {
      const float v = (i0 + j < len) ? xu[i0 + j] : 0.f;
      run = fmaf(kPre, run, v);
      y[j] = run;
    }
    // affine map of this block: y_out = A * y_in + B, A = 0.97^16, B = run
    float A = pw[kDeBlock], B = run;
    sa[tid] = A;
    sb[tid] = B;
    __syncthreads();
    for (int d = 1; d < kThreads; d <<= 1) {
      float a2 = 1.f, b2 = 0.f;
      if (tid >= d) { a2 = sa[tid - d]; b2 = sb[tid - d]; }
      __syncthreads();
      if (tid >= d) {   // compose: (earlier map) then (mine)
        B = fmaf(A, b2, B);
        A = A * a2;
        sa[tid] = A;
        sb[tid] = B;
      }
      __syncthreads();
    }
    // value entering this thread's block = (scan of the previous thread)(carry)
    const float yin = tid == 0 ? carry : fmaf(sa[tid - 1], carry, sb[tid - 1]);
#pragma unroll
    for (int j = 0; j < kDeBlock; ++j)
      if (i0 + j < len) xu[i0 + j] = fmaf(pw[j + 1], yin, y[j]);
    const float next = fmaf(sa[kThreads - 1], carry, sb[kThreads - 1]);
    __syncthreads();
    carry = next;
  }
}

}  // namespace audio
}  // namespace rced
