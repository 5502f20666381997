// STFT front-end / ISTFT rebuild at fp32 quality on the bf16 matrix pipe (three-part operands, six products per MFMA-sized term: the
// form of the CR-CED kernel, kernels_fused_v3.h).  Same reference rows as kernels_audio.h (N1: data_utils/audio_feature.py:22-44,
// N2: model_utils/utils.py:171-183); the fp32-MFMA kernels there stay in the library as the comparator (rced_stft_ex / rced_istft_ex with RCED_AUDIO_F32).
//
// Both transforms are dense DFT GEMMs with K = 256 exactly once the two identically-zero terms are dropped:
//   STFT : 258 real rows (re, im of 129 bins) -> 256: im of bin 0 and of bin 128 are zero for a real signal, so row 1 carries re of bin
//          128 instead: row 0 = re(0), row 1 = re(128), rows 2b, 2b + 1 = re(b), im(b), b = 1..127  -> 16 M-tiles.
//   ISTFT: K slots 2b + c (c = 0 re, 1 im), b = 0..127; slot 1 (im of bin 0: irfft ignores it) carries re of bin 128.  im of bin 128
//          (ignored at nfft = 256, used at the reference's shipped nfft = 512) is one rank-1 update on the VALU.  Only samples 128..255
//          of a frame survive de_frame (utils.py:139-147) -- 8 M-tiles, not 16; samples 0..127 of frame 0 are a 33-k-MAC side kernel.
// Decomposition: ONE M-TILE PER WAVE with its A fragments resident in registers (8 chunks x 3 parts x 4 = 96 VGPRs, read once per
// workgroup), workgroups of 8 waves, each walking the 64-frame blocks of one utterance: the operand that is re-read per block is the
// small one (the signal).  The fp32 kernels re-fetched 278 KB of A fragments from the L2 per 64 frames: 570 MB per call at config 3.
// The signal block is split into its three bf16 parts ONCE, when it is staged (rows of 128 samples + 16 bytes of pad: a lane group's
// sixteen frames sit on sixteen different bank slots).
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_audio.h"

namespace rced {
namespace audio {
namespace x6 {

typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kWaves = 8, kThreadsX = kWaves * 64;
constexpr int kChunks = kFrame / 32;                 // K = 256 in eight K = 32 chunks
constexpr int kPackPerMT = kChunks * 3 * 64 * 8;     // bf16 per M-tile: [chunk][part][lane][8]
constexpr int kStftMTx = 16, kIstftMTx = 8;
constexpr int kStftPackX = kStftMTx * kPackPerMT;    // bf16
constexpr int kIstftPackX = kIstftMTx * kPackPerMT;  // bf16 (+ the rank-1 column and the head table, fp32: audio_api.hip)

__device__ __forceinline__ f32x4 mfma32(s16x8 a, s16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
struct Parts {
  s16x8 h, m, l;
};
// the six products of one chunk for two independent chains, smallest first (kernels_fused_v3_l23.h mma2)
__device__ __forceinline__ void mma2(const s16x8 (&a)[3], const Parts& b0, f32x4& c0, const Parts& b1, f32x4& c1) {
  c0 = mfma32(a[1], b0.m, c0);
  c1 = mfma32(a[1], b1.m, c1);
  c0 = mfma32(a[2], b0.h, c0);
  c1 = mfma32(a[2], b1.h, c1);
  c0 = mfma32(a[0], b0.l, c0);
  c1 = mfma32(a[0], b1.l, c1);
  c0 = mfma32(a[1], b0.h, c0);
  c1 = mfma32(a[1], b1.h, c1);
  c0 = mfma32(a[0], b0.m, c0);
  c1 = mfma32(a[0], b1.m, c1);
  c0 = mfma32(a[0], b0.h, c0);
  c1 = mfma32(a[0], b1.h, c1);
}
// two fp32 values -> three packed bf16 pairs, x = h + m + l to 2^-24 (round to nearest at every step)
struct P3 {
  unsigned h, m, l;
};
__device__ __forceinline__ P3 split2(float x0, float x1) {
  P3 p;
  const bf16x2 bh = {(__bf16)x0, (__bf16)x1};
  p.h = __builtin_bit_cast(unsigned, bh);
  const float r0 = x0 - __builtin_bit_cast(float, p.h << 16), r1 = x1 - __builtin_bit_cast(float, p.h & 0xffff0000u);
  const bf16x2 bm = {(__bf16)r0, (__bf16)r1};
  p.m = __builtin_bit_cast(unsigned, bm);
  const float s0 = r0 - __builtin_bit_cast(float, p.m << 16), s1 = r1 - __builtin_bit_cast(float, p.m & 0xffff0000u);
  const bf16x2 bl = {(__bf16)s0, (__bf16)s1};
  p.l = __builtin_bit_cast(unsigned, bl);
  return p;
}

// this wave's A fragments: [chunk][part], lane's 16 bytes each
struct AFrag {
  s16x8 a[kChunks][3];
};
__device__ __forceinline__ void load_a(AFrag& A, const unsigned short* pack, int mt, int lane) {
  const u32x4* src = reinterpret_cast<const u32x4*>(pack + (size_t)mt * kPackPerMT) + lane;
#pragma unroll
  for (int c = 0; c < kChunks; ++c)
#pragma unroll
    for (int q = 0; q < 3; ++q) A.a[c][q] = __builtin_bit_cast(s16x8, src[(c * 3 + q) * 64]);
}

// D[16 rows of this wave's M-tile][64 frames] += A x B over K = 256.  B: three bf16 images in LDS; `lane_off` = this lane's byte offset
// (frame n of an N-tile, k-quad kq), `tile_stride` = bytes between N-tiles, chunk_off(c) = byte offset of chunk c inside a frame's row.
template <class ChunkOff>
__device__ __forceinline__ void gemm_block(const AFrag& A, const char* img, int part_bytes, int lane_off, int tile_stride, ChunkOff chunk_off,
                                           f32x4 (&acc)[4]) {
  Parts b[2][4];
  auto ld = [&](int c, int r) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const char* p = img + lane_off + t * tile_stride + chunk_off(c);
      b[r][t].h = *reinterpret_cast<const s16x8*>(p);
      b[r][t].m = *reinterpret_cast<const s16x8*>(p + part_bytes);
      b[r][t].l = *reinterpret_cast<const s16x8*>(p + 2 * part_bytes);
    }
  };
  ld(0, 0);
#pragma unroll
  for (int c = 0; c < kChunks; ++c) {
    if (c + 1 < kChunks) ld(c + 1, (c + 1) & 1);
    mma2(A.a[c], b[c & 1][0], acc[0], b[c & 1][1], acc[1]);
    mma2(A.a[c], b[c & 1][2], acc[2], b[c & 1][3], acc[3]);
  }
}

// ---- STFT -------------------------------------------------------------------------------------------------------------------------
constexpr int kSegRowB = 2 * kStep + 16;                       // bytes per 128 samples of one part: 272
constexpr int kSegPartB = (kFramesPerWg + 1) * kSegRowB;       // 65 rows: 17,680
constexpr int kSegPairs = (kFramesPerWg + 1) * (kStep / 2);    // sample pairs of a block's segment: 4,160
constexpr int kSegIter = (kSegPairs + kThreadsX - 1) / kThreadsX;   // per thread: 9
// The raw samples a thread needs for its pairs of one block: s[g-1], s[g], s[g+1], g = 128 f0 + 2 (tid + 512 it).  Fetched one block AHEAD
// (27 registers), so that the loads' latency -- nine dependent trips to the L2 / HBM per block when they sat in the staging loop -- lies
// under the previous block's MFMAs.
struct SegRaw {
  float m[kSegIter], z[kSegIter], p[kSegIter];
};
__device__ __forceinline__ void seg_fetch(SegRaw& R, const float* __restrict__ s, int len, int f0, int tid) {
#pragma unroll
  for (int it = 0; it < kSegIter; ++it) {
    const int g = f0 * kStep + 2 * (tid + it * kThreadsX);
    const bool in = tid + it * kThreadsX < kSegPairs && g < len;
    R.z[it] = in ? s[g] : 0.f;
    R.m[it] = in && g > 0 ? s[g - 1] : 0.f;
    R.p[it] = in && g + 1 < len ? s[g + 1] : 0.f;
  }
}
// grid (N, 2 M-groups, S frame ranges); pack: kStftPackX bf16
__global__ __launch_bounds__(kThreadsX) void stft_x6_kernel(const float* __restrict__ pcm, const int* __restrict__ lengths,
                                                             const unsigned short* __restrict__ apack, int L, int T, float* __restrict__ mag,
                                                             float* __restrict__ phase) {
  __shared__ __attribute__((aligned(16))) char seg[3 * kSegPartB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, kq = lane >> 4;
  const int utt = blockIdx.x, mt = blockIdx.y * kWaves + wave;
  const int len = lengths ? min(lengths[utt], L) : L;
  const int nf = len > 0 ? num_frames(len) : 0;
  const float* s = pcm + (size_t)utt * L;
  AFrag A;
  load_a(A, apack, mt, lane);
  const int nblk = (T + kFramesPerWg - 1) / kFramesPerWg, per = (nblk + (int)gridDim.z - 1) / (int)gridDim.z;
  const int blk0 = blockIdx.z * per, blk1 = min(nblk, blk0 + per);
  SegRaw R;
  if (blk0 < blk1) seg_fetch(R, s, len, blk0 * kFramesPerWg, tid);
  for (int blk = blk0; blk < blk1; ++blk) {
    const int f0 = blk * kFramesPerWg;
    __syncthreads();   // the previous block's reads of seg are done
    // stage the pre-emphasised, zero-padded segment [128 f0, 128 (f0 + 65)) as three bf16 parts, two samples per store.
    // e[0] = s[0], e[g] = s[g] - 0.97 s[g-1] as ONE float32 multiply and ONE float32 subtract (audio_feature.py:54 works in float32)
#pragma unroll
    for (int it = 0; it < kSegIter; ++it) {
      const int p = tid + it * kThreadsX;
      if (p < kSegPairs) {
        const int g = f0 * kStep + 2 * p;
        float e0 = 0.f, e1 = 0.f;
        if (g < len) {
          e0 = g == 0 ? R.z[it] : __fsub_rn(R.z[it], __fmul_rn(kPre, R.m[it]));
          if (g + 1 < len) e1 = __fsub_rn(R.p[it], __fmul_rn(kPre, R.z[it]));
        }
        const P3 q = split2(e0, e1);
        char* d = seg + (p >> 6) * kSegRowB + (p & 63) * 4;
        *reinterpret_cast<unsigned*>(d) = q.h;
        *reinterpret_cast<unsigned*>(d + kSegPartB) = q.m;
        *reinterpret_cast<unsigned*>(d + 2 * kSegPartB) = q.l;
      }
    }
    __syncthreads();
    if (blk + 1 < blk1) seg_fetch(R, s, len, f0 + kFramesPerWg, tid);   // in flight during this block's MFMAs
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // sample k = 32c + 8kq + e of frame 16t + n: row (16t + n) + (c >> 2), byte 64 (c & 3) + 16 kq
    gemm_block(A, seg, kSegPartB, n * kSegRowB + 16 * kq, 16 * kSegRowB, [](int c) { return (c >> 2) * kSegRowB + 64 * (c & 3); }, acc);
    // epilogue: rows 4kq + {0,1} / {2,3} = (re, im) of bins 8 mt + 2kq + {0, 1}; M-tile 0, kq 0: rows 0, 1 = re of bin 0, re of bin 128.
    // (Staging the results in LDS and writing whole rows -- 256 contiguous bytes per frame instead of one lane per frame -- was measured:
    // 0.212 against 0.202 ms; the stores are not what this kernel waits for.)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int fr = f0 + 16 * t + n;
      if (fr >= T) continue;
      const bool live = fr < nf;   // frames past the utterance: zero spectrum (padding_batch), phase 1 + 0j
      const f32x4 v = acc[t];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float re = live ? (h ? v.z : v.x) : 0.f, im = live ? (h ? v.w : v.y) : 0.f;
        const int b = 8 * mt + 2 * kq + h;
        const size_t row = ((size_t)utt * T + fr) * kBins;
        if (b == 0) {   // (re(0), re(128)): two bins with zero imaginary part
          const float re128 = im;
          const float m128 = fabsf(re128);
          mag[row + kBins - 1] = m128;
          if (phase) *reinterpret_cast<f32x2*>(phase + 2 * (row + kBins - 1)) = m128 > 0.f ? f32x2{re128 / m128, 0.f} : f32x2{1.f, 0.f};
          im = 0.f;
        }
        const float m = sqrtf(re * re + im * im);
        mag[row + b] = m;
        if (phase) {
          const float inv = m > 0.f ? 1.f / m : 0.f;
          *reinterpret_cast<f32x2*>(phase + 2 * (row + b)) = m > 0.f ? f32x2{re * inv, im * inv} : f32x2{1.f, 0.f};
        }
      }
    }
  }
}

// ---- ISTFT ------------------------------------------------------------------------------------------------------------------------
constexpr int kXRowB = 2 * kFrame + 16;                        // bytes per frame of one part: 256 slots + pad = 528
constexpr int kXPartB = kFramesPerWg * kXRowB;                 // 33,792
constexpr int kORow = kStep + 4;                               // floats per frame of the output staging (conflict-free 16-byte stores)
constexpr int kIstftLdsBytes = 3 * kXPartB + (kFramesPerWg + 2 * kBins + kStep) * 4;   // images + xim + xk (frame 0's spectrum, fp32) + the head
static_assert(kFramesPerWg * kORow * 4 + 2 * kThreadsX * 4 <= 3 * kXPartB, "output staging + scan arrays alias the (dead) images");
// mag [N,T,129], phase [N,T,129,2] -> x [N, (T+1)*128].  grid (N, 1, S).  cpack: kIstftPackX bf16; cim: 128 floats, the coefficients of
// im(bin 128) for samples 128..255 (zero at nfft = 256); chead: [k = 2b + c (258)][n (128)] fp32 for samples 0..127 of frame 0.
// FUSED (S = 1: one workgroup walks the whole utterance): de_frame AND de_emphasis (utils.py:139-147, 104-113) here too -- the block's
// 8,192 samples go through LDS, a blocked affine scan (16 samples per thread, 512 threads, the carry across blocks in a register) turns
// them into y[i] = x[i] + 0.97 y[i-1], and they leave with coalesced 16-byte stores: no second pass over the signal in HBM, no
// scattered 16-byte stores from the MFMA layout.  !FUSED: the frames' second halves only (istft_head_kernel + deemphasis_kernel follow).
template <bool FUSED>
__global__ __launch_bounds__(kThreadsX) void istft_x6_kernel(const float* __restrict__ mag, const float* __restrict__ phase,
                                                              const unsigned short* __restrict__ cpack, const float* __restrict__ cim,
                                                              const float* __restrict__ chead, int T, float* __restrict__ x) {
  extern __shared__ __attribute__((aligned(16))) char xs[];
  float* xim = reinterpret_cast<float*>(xs + 3 * kXPartB);
  float* xk = xim + kFramesPerWg;        // frame 0's spectrum (fp32, k = 2b + c)
  float* hbuf = xk + 2 * kBins;          // samples 0..127 of frame 0
  float* obuf = reinterpret_cast<float*>(xs);                       // [frame][kORow]: aliases the images once the GEMM has read them
  float* sa = obuf + kFramesPerWg * kORow;                          // the scan's affine maps
  float* sb = sa + kThreadsX;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, kq = lane >> 4;
  const int utt = blockIdx.x;
  AFrag A;
  load_a(A, cpack, wave, lane);
  const f32x4 ci = *reinterpret_cast<const f32x4*>(cim + 16 * wave + 4 * kq);   // rows 4kq .. 4kq+3 of this wave's M-tile
  float* xo = x + (size_t)utt * (T + 1) * kStep;
  const int nblk = (T + kFramesPerWg - 1) / kFramesPerWg, per = (nblk + (int)gridDim.z - 1) / (int)gridDim.z;
  const int blk0 = blockIdx.z * per, blk1 = min(nblk, blk0 + per);
  float pw[kDeBlock + 1];
  pw[0] = 1.f;
#pragma unroll
  for (int j = 1; j <= kDeBlock; ++j) pw[j] = pw[j - 1] * kPre;
  float carry = 0.f;   // y just before the block (FUSED)
  for (int blk = blk0; blk < blk1; ++blk) {
    const int f0 = blk * kFramesPerWg;
    __syncthreads();
    // stage X[frame][slot 2b + c] = mag * (re, im) of bin b (merge_magphase, utils.py:119-126) as three bf16 parts; slot 1 = re of bin 128;
    // im of bin 128 to xim (fp32); zero past T
    for (int i = tid; i < kFramesPerWg * 128; i += kThreadsX) {
      const int fr = i >> 7, b = i & 127;
      float v0 = 0.f, v1 = 0.f;
      if (f0 + fr < T) {
        const size_t o = ((size_t)utt * T + f0 + fr) * kBins + b;
        const float m = mag[o];
        const f32x2 p = *reinterpret_cast<const f32x2*>(phase + 2 * o);
        v0 = m * p.x;
        v1 = m * p.y;
        if (FUSED && blk == 0 && fr == 0) {
          xk[2 * b] = v0;
          xk[2 * b + 1] = v1;
        }
        if (b == 0) {
          const size_t o8 = o + kBins - 1;
          const float m8 = mag[o8];
          const f32x2 p8 = *reinterpret_cast<const f32x2*>(phase + 2 * o8);
          v1 = m8 * p8.x;
          xim[fr] = m8 * p8.y;
          if (FUSED && blk == 0 && fr == 0) {
            xk[2 * (kBins - 1)] = v1;
            xk[2 * (kBins - 1) + 1] = m8 * p8.y;
          }
        }
      } else if (b == 0) {
        xim[fr] = 0.f;
      }
      const P3 q = split2(v0, v1);
      char* d = xs + fr * kXRowB + b * 4;
      *reinterpret_cast<unsigned*>(d) = q.h;
      *reinterpret_cast<unsigned*>(d + kXPartB) = q.m;
      *reinterpret_cast<unsigned*>(d + 2 * kXPartB) = q.l;
    }
    __syncthreads();
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    gemm_block(A, xs, kXPartB, n * kXRowB + 16 * kq, 16 * kXRowB, [](int c) { return 64 * c; }, acc);
    // + the rank-1 term of im(bin 128); rows = samples 128 + 16 wave + 4kq + j of frame 16 t + n
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float xi = xim[16 * t + n];
      acc[t].x = fmaf(ci.x, xi, acc[t].x);
      acc[t].y = fmaf(ci.y, xi, acc[t].y);
      acc[t].z = fmaf(ci.z, xi, acc[t].z);
      acc[t].w = fmaf(ci.w, xi, acc[t].w);
    }
    if constexpr (!FUSED) {   // de_frame (utils.py:139-147): sample 128 + s of frame fr goes to out[128 fr + 128 + s]
      const int n0 = kStep + 16 * wave + 4 * kq;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int fr = f0 + 16 * t + n;
        if (fr < T) *reinterpret_cast<f32x4*>(xo + (size_t)fr * kStep + n0) = acc[t];
      }
    } else {
      float head_y = 0.f;
      if (blk == 0) {   // samples 0..127 of frame 0 (the only frame whose first half survives de_frame): K = 258 on the VALU of waves 0, 1
        if (tid < kStep) {
          float s = 0.f;
#pragma unroll 6
          for (int k = 0; k < 2 * kBins; ++k) s = fmaf(chead[k * kStep + tid], xk[k], s);
          hbuf[tid] = s;
        }
      }
      __syncthreads();   // every wave has read its last fragment: the images are dead
#pragma unroll
      for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(obuf + (16 * t + n) * kORow + 16 * wave + 4 * kq) = acc[t];
      if (blk == 0 && tid == 0) {   // de-emphasis of the head, serially (128 steps, once per utterance): y[0] = x[0]
        float run = 0.f;
        for (int j = 0; j < kStep; ++j) {
          run = fmaf(kPre, run, hbuf[j]);
          hbuf[j] = run;
        }
      }
      __syncthreads();
      if (blk == 0) {
        if (tid < kStep) xo[tid] = hbuf[tid];
        head_y = hbuf[kStep - 1];
        carry = head_y;
      }
      // blocked affine scan over the block's 8,192 samples in time order: thread i owns samples 16 i .. 16 i + 15 (frame i >> 3)
      const int fl = tid >> 3, s0 = kDeBlock * (tid & 7);
      const bool live = f0 + fl < T;
      float y[kDeBlock];
      float run = 0.f;
#pragma unroll
      for (int j4 = 0; j4 < kDeBlock / 4; ++j4) {
        const f32x4 v = live ? *reinterpret_cast<const f32x4*>(obuf + fl * kORow + s0 + 4 * j4) : f32x4{0.f, 0.f, 0.f, 0.f};
        run = fmaf(kPre, run, v.x);
        y[4 * j4] = run;
        run = fmaf(kPre, run, v.y);
        y[4 * j4 + 1] = run;
        run = fmaf(kPre, run, v.z);
        y[4 * j4 + 2] = run;
        run = fmaf(kPre, run, v.w);
        y[4 * j4 + 3] = run;
      }
      float Am = pw[kDeBlock], Bm = run;   // this thread's map: y_out = Am * y_in + Bm
      sa[tid] = Am;
      sb[tid] = Bm;
      __syncthreads();
      for (int d = 1; d < kThreadsX; d <<= 1) {
        float a2 = 1.f, b2 = 0.f;
        if (tid >= d) {
          a2 = sa[tid - d];
          b2 = sb[tid - d];
        }
        __syncthreads();
        if (tid >= d) {   // compose: (earlier map) then (mine)
          Bm = fmaf(Am, b2, Bm);
          Am = Am * a2;
          sa[tid] = Am;
          sb[tid] = Bm;
        }
        __syncthreads();
      }
      const float yin = tid == 0 ? carry : fmaf(sa[tid - 1], carry, sb[tid - 1]);
      if (live) {
        float* dst = xo + (size_t)(f0 + fl + 1) * kStep + s0;
#pragma unroll
        for (int j4 = 0; j4 < kDeBlock / 4; ++j4)
          *reinterpret_cast<f32x4*>(dst + 4 * j4) = f32x4{fmaf(pw[4 * j4 + 1], yin, y[4 * j4]), fmaf(pw[4 * j4 + 2], yin, y[4 * j4 + 1]),
                                                          fmaf(pw[4 * j4 + 3], yin, y[4 * j4 + 2]), fmaf(pw[4 * j4 + 4], yin, y[4 * j4 + 3])};
      }
      carry = fmaf(sa[kThreadsX - 1], carry, sb[kThreadsX - 1]);   // (frames past T contribute zeros: unused)
    }
  }
}

// samples 0..127 of frame 0 of every utterance (the only frame whose first half survives de_frame): thread n, K = 258 on the VALU.
// chead: [k = 2b + c (258)][n (128)] fp32, the fp32 kernels' coefficients (1 / nfft, irfft's factor 2, 1 / hamming folded in)
__global__ __launch_bounds__(kStep) void istft_head_kernel(const float* __restrict__ mag, const float* __restrict__ phase,
                                                            const float* __restrict__ chead, int T, float* __restrict__ x) {
  __shared__ float xk[2 * kBins];
  const int n = threadIdx.x, utt = blockIdx.x;
  for (int b = n; b < kBins; b += kStep) {
    const size_t o = (size_t)utt * T * kBins + b;
    const float m = mag[o];
    xk[2 * b] = m * phase[2 * o];
    xk[2 * b + 1] = m * phase[2 * o + 1];
  }
  __syncthreads();
  float s = 0.f;
#pragma unroll 6
  for (int k = 0; k < 2 * kBins; ++k) s = fmaf(chead[k * kStep + n], xk[k], s);
  x[(size_t)utt * (T + 1) * kStep + n] = s;
}

}  // namespace x6
}  // namespace audio
}  // namespace rced
