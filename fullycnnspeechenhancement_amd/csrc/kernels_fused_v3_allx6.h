// All-x6 form of the CR-CED kernel (Map<3>): the two phases that exist in this form only.  Included by kernels_fused_v3.h inside
// namespace rced::v3; the design note sits there, at Map<3>.
//   convert_x0      the input rows of a tile -> the bf16 planes the first layer (layer1_x6l) reads: an im2col along time
//   final_phase_x6  decode_final (model_utils/model.py:89-90: 1x129, 8 -> 1, no BN, no ReLU) as a three-part bf16 GEMM
#pragma once

// ---- input rows -> planes --------------------------------------------------------------------------------------------------------
// X0 holds the tile's 11 input rows as fp32: float 4 + 133 r + f = x[t0 + r - 3][f] (zero outside the utterance and in the gap columns;
// xstage_load / xstage_store).  Pixel (frame i, bin f) of the planes gets the eight values x[t0 + i + rr - 3][f], rr = 0..7: the first
// layer's "8 input channels".  One pixel per thread (bins 0..127 of the four frames), bin 128 of the four frames by four lanes of wave 7;
// waves 5, 6 put the zeros back into the planes' pad and gap rows (decode_final's image lay over them).
template <class M>
__device__ __forceinline__ void convert_px(unsigned lds0, int i, int f) {
  const unsigned src = lds0 + 4 * (M::kX0Off + 4 + i * kS + f);
  float v[8];
#pragma unroll
  for (int rr = 0; rr < 8; ++rr) v[rr] = lds_ld<float>(src, rr * kS * 4);
  u32x4 h, m, l;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const P3 p = split2(v[2 * q], v[2 * q + 1]);
    h[q] = p.h;
    m[q] = p.m;
    l[q] = p.l;
  }
  const unsigned dst = lds0 + 4 * M::kB8Off + (kB8Pad + i * kS + f) * 16;
  lds_st<u32x4>(dst, 0, h);
  lds_st<u32x4>(dst, kB8PlaneBytes, m);
  lds_st<u32x4>(dst, 2 * kB8PlaneBytes, l);
}
template <class M>
__device__ __forceinline__ void convert_x0(unsigned lds0, int wave, int lane) {
  lane = opaque(lane);
  const int tid = wave * 64 + lane;
  convert_px<M>(lds0, tid >> 7, tid & 127);
  if (wave == 7 && lane < kTF) convert_px<M>(lds0, lane, kF - 1);
  // pad rows 0..3, the four gap rows behind each frame, the four rows behind the tile: 24 rows x 3 parts = 72 sixteen-byte stores
  const int z = tid - 5 * 64;
  if (z >= 0 && z < 72) {
    const int part = z / 24, rr = z - part * 24;
    const int row = rr < 4 ? rr : rr < 20 ? kB8Pad + kS * ((rr - 4) >> 2) + kF + ((rr - 4) & 3) : kB8Pad + kNPX + (rr - 20);
    lds_st<u32x4>(lds0 + 4 * M::kB8Off + part * kB8PlaneBytes + row * 16, 0, u32x4{0u, 0u, 0u, 0u});
  }
}
static_assert(kB8Rows == kB8Pad + kNPX + 4, "convert_x0 zeroes the planes' pad, gap and trailing rows by this map");

// ---- decode_final ------------------------------------------------------------------------------------------------------------------
// Chunk q of the K axis = window taps u = 4q + kq (kq = lane >> 4), 8 channels each; column n of column tile ct = (frame n >> 2, block
// j = 4 ct + (n & 3)), output bins 16 j + m.  The tap of lane (kq, n) in chunk q is input bin b = 16 j - 64 + 4q + kq, at row
// 140 f + b + (b >> 4) of the image, b >> 4 = j - 4 + (q >> 2); outside 0..128 the lane reads the zero row.  Column tile 2 is block 8 (bin
// 128): columns with n & 3 != 0 are idle (always the zero row), chunks 0..16 only.
// Runs of chunks: wave w < 4: chunks 4w .. 4w + 3, three column tiles; wave w >= 4: run 11 - w of five chunks from 16 + 5 (run - 4), two
// column tiles -- and the third for chunk 16 (wave 7).  Sets of six MFMAs per SIMD (waves w, w + 4): 12 + 10, 12 + 10, 12 + 10, 12 + 11.
template <class M, int NCH, bool CT2ALL>
__device__ __forceinline__ void fin_run(unsigned ra, unsigned ta, int b0, int b2, int q0, bool ct2first, unsigned zaddr, f32x4 (&acc)[3]) {
  s16x8 a[2][3];
  Parts b[2][3];
  auto ld = [&](auto ic) {
    constexpr int i = decltype(ic)::value, r = i & 1;
    const int q = q0 + i;   // wave-uniform
    const unsigned ta_i = ta + (unsigned)(64 * q);
#pragma unroll
    for (int p = 0; p < 3; ++p) a[r][p] = lds_ld<s16x8>(ta_i, p * kFinTPart);
    const unsigned ra_i = ra + (unsigned)(16 * (4 * q + (q >> 2)));
    const int bq = b0 + 4 * q;
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) {
      if (ct == 2 && !(CT2ALL || i == 0)) continue;
      const bool ok = ct < 2 ? (unsigned)(bq + 64 * ct) <= 128u : (unsigned)(b2 + 4 * q) <= 128u;
      const unsigned ad = ok ? ra_i + 1088u * ct : zaddr;
      b[r][ct].h = lds_ld<s16x8>(ad, 0);
      b[r][ct].m = lds_ld<s16x8>(ad, kHPlaneBytes);
      b[r][ct].l = lds_ld<s16x8>(ad, 2 * kHPlaneBytes);
    }
  };
  ld(IC<0>{});
  pin();
  static_for<0, NCH>([&](auto ic) {
    constexpr int i = decltype(ic)::value, r = i & 1;
    if constexpr (i + 1 < NCH) ld(IC<i + 1>{});
    pin();
    mma2(a[r], b[r][0], acc[0], a[r], b[r][1], acc[1]);
    if constexpr (CT2ALL) {
      acc[2] = l2x_mma(a[r], b[r][2], acc[2]);
    } else if constexpr (i == 0) {
      if (ct2first) acc[2] = l2x_mma(a[r], b[r][2], acc[2]);
    }
    pin();
  });
}

template <class M>
__device__ __forceinline__ void final_phase_x6(const Params& P, unsigned lds0, int wave, int lane, int utt, int t0, const XStage& xnext, float* x0 DET_ARG) {
  DET_BEGIN();
  lane = opaque(lane);
  const int n = lane & 15, kq = lane >> 4;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  float* yt = P.y + ((size_t)utt * P.T + t0) * kF;   // the tile's first output row
  const int nfr = P.T - t0 < kTF ? P.T - t0 : kTF;   // frames of the tile inside the utterance
  xstage_store(xnext, x0, wave * 64 + lane);         // the next tile's input rows: read by convert_x0, behind the barrier
  {
    const int fi = n >> 2, j0 = n & 3;
    const int b0 = 16 * j0 - 64 + kq;                              // this lane's tap of column tile 0 in chunk 0
    const int b2 = j0 == 0 ? b0 + 128 : 0x40000000;                // ... of column tile 2 (block 8), idle columns: never in range
    const unsigned ra = lds0 + 4 * M::kB8Off + (unsigned)((kHFr * fi + 17 * j0 - 68 + kq) * 16);   // its row address, chunk 0, column tile 0
    const unsigned ta = lds0 + 4 * M::kFinTOff + (unsigned)((kq - n + 15) * 16);                   // the tap table's row for (kq, m = n), chunk 0
    const unsigned zaddr = lds0 + 4 * M::kB8Off + kHZeroRow * 16;
    f32x4 acc[3] = {zero4, zero4, zero4};
    if (wave < 4) {
      fin_run<M, 4, true>(ra, ta, b0, b2, 4 * wave, false, zaddr, acc);
    } else {
      const int run = 11 - wave;                                   // 7, 6, 5, 4
      fin_run<M, 5, false>(ra, ta, b0, b2, 16 + 5 * (run - 4), run == 4, zaddr, acc);
    }
    const unsigned scr = lds0 + (unsigned)M::finscr(wave, 0) + (unsigned)lane * 16u;
#pragma unroll
    for (int ct = 0; ct < 3; ++ct) lds_st<f32x4>(scr, ct * M::kPlaneBytes, acc[ct]);
  }
  DET(8);
  __syncthreads();
  DET(9);
  if (wave < 3) {   // ---- finish column tile `wave`: partial sums of waves 0..7, in that order, + bias
    const unsigned scr = lds0 + (unsigned)(M::finscr(0, 0) + wave * M::kPlaneBytes) + (unsigned)lane * 16u;
    f32x4 v = lds_ld<f32x4>(scr, 0);
#pragma unroll
    for (int w = 1; w < kWaves; ++w) v += lds_ld<f32x4>(scr, M::finscr(w, 0) - M::finscr(0, 0));
    v += f32x4{P.fin_bias, P.fin_bias, P.fin_bias, P.fin_bias};
    const int fi = n >> 2;
    if (wave < 2) {
      const int f0 = 16 * (4 * wave + (n & 3)) + 4 * kq;   // rows 4kq..4kq+3 = bins f0..f0+3 of frame fi
      if (fi < nfr) {
        *reinterpret_cast<f32x4_u*>(yt + fi * kF + f0) = v;
        store_wait_state();   // see lds_dma.h
      }
    } else if (kq == 0 && (n & 3) == 0 && fi < nfr) {
      yt[fi * kF + kF - 1] = v.x;                          // block 8, row 0 = bin 128
    }
  }
  DET(10);
  convert_x0<M>(lds0, wave, lane);
  DET(11);
  __syncthreads();   // the planes are complete (and the partial sums read) before the next tile's first layer
  DET(12);
}
