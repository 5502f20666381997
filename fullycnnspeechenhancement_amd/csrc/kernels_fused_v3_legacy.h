// LEGACY form 1 of the CR-CED kernel (RCED_V3_LEGACY_FORMS builds only; NOT in the product library): the 18 -> 30 layers alone on the bf16
// pipe, their B fragments split in the consumer.  Included by kernels_fused_v3.h inside namespace rced::v3.
#pragma once
// ---- X6 form ------------------------------------------------------------------------------------------------
// K = 96 slots in three K = 32 chunks; slot k = 32 c + 8 kq + e of the MFMA's K axis is
//   c = 0, 1:        tap 2c + (kq >> 1), channel 8 (kq & 1) + e              (one 16-byte row half of a plane)
//   c = 2, kq < 2:   tap 4, channel 8 kq + e
//   c = 2, kq = 2:   tap e >> 1 (0..3), channel 16 + (e & 1)                 (the remainder channels' window, four rows)
//   c = 2, kq = 3:   tap 4, channel 16 + e for e < 2; zero weights for e >= 2 (they meet the next three rows: finite values)
// A wave computes ONE M-tile: waves 0..3 channels 0..15, waves 4..7 channels 16..29 (g = wave >> 2), each for the tiles
// j + 4t (j = wave & 3, t < 8) -- half the A fragments per wave (36 registers, half the global loads) for twice the B reads,
// which the LDS has room for: a slot of the stream = one chunk of one tile = three conflict-free ds_read_b128 (the fragment's
// h, m, l parts; in the last chunk the remainder rows besides, selected into the upper lanes by 12 v_cndmask) for six MFMAs of
// 16 cycles, 8 waves: half of the LDS's cycles.  No other VALU (beside bf16 MFMAs, which hold the SIMD's issue port for half of
// their 16 cycles, up to two VALU per MFMA are nearly free -- MI355X guide -- but layer 1's epilogue already did the split).
// Tile 32 is cut by chunk inside each M-tile group: members j = 0, 1 are the helpers (chunks 0, 1), j = 2 the reducer (chunk 2,
// the shift, the epilogue): 150 / 150 / 150 / 144 MFMAs per wave.
struct RemRaw {          // the remainder channels' rows of one window half, as loaded: [h16 h17 | m16 m17] x 4, [l16 l17] x 4
  u32x2 hm[4];
  unsigned lq[4];
};
template <int C>
__device__ __forceinline__ void l2x_load(unsigned rdm, unsigned rdr, unsigned rdrl, int om, int orr, int orl, Parts& b, RemRaw& rr) {
  b.h = lds_ld<s16x8>(rdm, om + 64 * C);
  if (RCED_X6_EXP & 32) {
    b.m = b.l = b.h;
  } else {
    b.m = lds_ld<s16x8>(rdm, om + 64 * C + MapX6::kPlaneBytes);
    b.l = lds_ld<s16x8>(rdm, om + 64 * C + 2 * MapX6::kPlaneBytes);
  }
  if constexpr (C == 2 && !(RCED_X6_EXP & 16)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      rr.hm[j] = lds_ld<u32x2>(rdr, orr + 8 * j);
      rr.lq[j] = lds_ld<unsigned>(rdrl, orl + 4 * j);
    }
  }
}
// last chunk: the upper lanes' slots are the remainder channels
__device__ __forceinline__ void l2x_merge(Parts& b, const RemRaw& rr, bool upper) {
  if (RCED_X6_EXP & 16) return;
  u32x4 h = __builtin_bit_cast(u32x4, b.h), m = __builtin_bit_cast(u32x4, b.m), l = __builtin_bit_cast(u32x4, b.l);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    h[j] = upper ? rr.hm[j].x : h[j];
    m[j] = upper ? rr.hm[j].y : m[j];
    l[j] = upper ? rr.lq[j] : l[j];
  }
  b.h = __builtin_bit_cast(s16x8, h);
  b.m = __builtin_bit_cast(s16x8, m);
  b.l = __builtin_bit_cast(s16x8, l);
}
// ReLU, [pixel][30] store of this wave's M-tile: lanes kq = 3 of M-tile 1 hold channels 28,29 and the padding 30,31
__device__ __forceinline__ void l2x_store(const Lane& L, f32x4 acc4, unsigned wr, int off, bool masked, int vb) {
  if (RCED_X6_EXP & 64) {
    if (acc4.x == 12345.678f) lds_st<float>(wr, off, acc4.y);
    return;
  }
  const f32x4 v = relu4(acc4);
  if (!masked || vbit(L, vb)) {
    lds_st<f32x2>(wr, off, f32x2{v.x, v.y});
    if (vbit(L, kVSt2)) lds_st<f32x2>(wr, off + 8, f32x2{v.z, v.w});
  }
}
// chunk C of tile 32 (this wave's M-tile)
template <int C, class Pre>
__device__ __forceinline__ f32x4 l2x_share(const Lane& L, const A2Regs& A, unsigned rdm, unsigned rdr, unsigned rdrl, f32x4 init, Pre& pre) {
  Parts b;
  RemRaw rr;
  l2x_load<C>(rdm, rdr, rdrl, 0, 0, 0, b, rr);
  pin();
  pre();
  pin();
  if constexpr (C == 2) l2x_merge(b, rr, vbit(L, kVUpper));
  return l2x_mma(A.a[0][C], b, init);
}

template <class M, class Dma>
__device__ __forceinline__ void layer2_x6(const Lane& L, unsigned lds0, const A2Regs& A, int wave, unsigned tag, unsigned* err, Dma dma DET_ARG) {
  constexpr int kT2R = M::kT2R, kT2W = M::kT2W;
  DET_BEGIN();
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const bool upper = vbit(L, kVUpper);
  const int g = wave >> 2, j = wave & 3;
  // ---- the share of tile 32 (members 0..2 of either group), first
  f32x4 accx = zero4, part0 = zero4, part1 = zero4;
  unsigned pflag0 = 0u, pflag1 = 0u;
  auto pre = once(dma);
  if (j < 3) {
    const int dt = 32 - j;   // tiles from this lane's tile j to tile 32
    const unsigned rdm = L.rd2m + dt * 512, rdr = L.rd2r + dt * 128, rdrl = L.rd2rl + dt * 64;
    if (j == 0) accx = l2x_share<0>(L, A, rdm, rdr, rdrl, zero4, pre);        // a helper's share starts from zero,
    else if (j == 1) accx = l2x_share<1>(L, A, rdm, rdr, rdrl, zero4, pre);
    else accx = l2x_share<2>(L, A, rdm, rdr, rdrl, A.sh[0], pre);             // the reducer's from the shift
    if (j < 2) {   // publish (LDS operations of a wave execute in order: data, then flag)
      lds_st<f32x4>(lds0 + L.scr + (2 * g + j) * 1024, M::kScratch2Off * 4, accx);
      cbar();
      if (L.a4 == 0) lds_poke_a(lds0 + (M::kFlag2Off + 2 * g + j) * 4, tag);
    }
  }
  DET(7);
  // ---- the eight regular tiles j + 4t as ONE stream of 24 chunk slots; tile t's stores ride behind tile t+1's first MFMAs
  {
    constexpr int NS = 8 * kL2Chunks, D = RCED_D2X, RING = D + 1;
    static_assert(D < kL2Chunks, "one last-chunk fragment in flight at a time (RemRaw is not ringed)");
    Parts b[RING];
    RemRaw rr;
    f32x4 acc[2];   // [tile & 1]
    const bool gj = j == 0;   // tiles 8, 16, 24 (member 0's t = 2, 4, 6) contain gap pixels
    run_job<NS, D>(
        [&](auto ic) {
          constexpr int i = decltype(ic)::value, t = i / kL2Chunks, c = i % kL2Chunks;
          l2x_load<c>(L.rd2m, L.rd2r, L.rd2rl, t * kT2R, t * 64 * 8, t * 64 * 4, b[i % RING], rr);
        },
        [&](auto ic) {
          constexpr int i = decltype(ic)::value, t = i / kL2Chunks, c = i % kL2Chunks, r = i % RING, u = t & 1;
          if constexpr (c == 2) l2x_merge(b[r], rr, upper);
          acc[u] = l2x_mma(A.a[0][c], b[r], c == 0 ? A.sh[0] : acc[u]);
          if constexpr (t > 0 && c == 0) {   // the previous tile's results
            constexpr bool gt = t - 1 == 2 || t - 1 == 4 || t - 1 == 6;
            l2x_store(L, acc[u ^ 1], L.wr2, (t - 1) * kT2W, gt && gj, kVL2 + t - 1);
          }
          if constexpr (t == 7 && c == 2) {   // reducers: the helpers' flags and partial sums, fetched inside the stream
            if (j == 2) {
              pflag0 = lds_peek_a(lds0 + (M::kFlag2Off + 2 * g) * 4);
              pflag1 = lds_peek_a(lds0 + (M::kFlag2Off + 2 * g + 1) * 4);
              cbar();
              part0 = lds_ld<f32x4>(lds0 + L.scr + (2 * g) * 1024, M::kScratch2Off * 4);
              part1 = lds_ld<f32x4>(lds0 + L.scr + (2 * g + 1) * 1024, M::kScratch2Off * 4);
            }
          }
        },
        pre);
    l2x_store(L, acc[1], L.wr2, 7 * kT2W, false, kVL2 + 7);   // tile j + 28: no gap
  }
  // ---- reducers: add the helpers' shares (fixed order), store tile 32 (pixels 512..527: no gap inside)
  if (j == 2) {
    if (!__builtin_amdgcn_readfirstlane(pflag0 == tag && pflag1 == tag)) {   // not there yet when fetched (not seen in practice)
      flag_wait(lds0 + (M::kFlag2Off + 2 * g) * 4, tag, err, 2u);
      flag_wait(lds0 + (M::kFlag2Off + 2 * g + 1) * 4, tag, err, 2u);
      part0 = lds_ld<f32x4>(lds0 + L.scr + (2 * g) * 1024, M::kScratch2Off * 4);
      part1 = lds_ld<f32x4>(lds0 + L.scr + (2 * g + 1) * 1024, M::kScratch2Off * 4);
    }
    f32x4 v = accx + part0;
    v += part1;
    l2x_store(L, v, L.wr2 + (32 - j) * (16 * 30 * 4), 0, false, 0);
  }
}

// ---- the same layer with BOTH M-tiles per wave (RCED_L2_BOTH): tiles w + 8t, a slot = one chunk of one tile = 12 MFMAs on two
// accumulation chains; tile 32 cut M-tile x K-part over waves 0..3 (helpers: chunk 0; reducers: chunks 1, 2)
template <int XM, bool HELPER, class Pre>
__device__ __forceinline__ f32x4 l2b_share(const Lane& L, const A2Regs& A, unsigned rdm, unsigned rdr, unsigned rdrl, f32x4 init, Pre& pre) {
  constexpr int C0 = HELPER ? 0 : 1, NC = HELPER ? 1 : 2, XS = XM < kL2MT ? XM : 0;
  Parts b[2];
  RemRaw rr;
  f32x4 acc = init;
  const bool upper = vbit(L, kVUpper);
  run_job<NC, 1>(
      [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        l2x_load<C0 + i>(rdm, rdr, rdrl, 0, 0, 0, b[i % 2], rr);
      },
      [&](auto ic) {
        constexpr int i = decltype(ic)::value, c = C0 + i;
        if constexpr (c == 2) l2x_merge(b[i % 2], rr, upper);
        acc = l2x_mma(A.a[XS][c], b[i % 2], acc);
      },
      pre);
  return acc;
}
template <class M, class Dma>
__device__ __forceinline__ void layer2_x6_both(const Lane& L, unsigned lds0, const A2Regs& A, int wave, unsigned tag, unsigned* err, Dma dma DET_ARG) {
  constexpr int kT2R = M::kT2R, kT2W = M::kT2W, S1 = kL2MT - 1;
  DET_BEGIN();
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const bool upper = vbit(L, kVUpper);
  f32x4 accx = zero4, part = zero4;
  unsigned pflag = 0u;
  auto pre = once(dma);
  if (wave < 4) {
    const int dt = 32 - wave;
    const unsigned rdm = L.rd2m + dt * 512, rdr = L.rd2r + dt * 128, rdrl = L.rd2rl + dt * 64;
    if (wave == 0) accx = l2b_share<0, true>(L, A, rdm, rdr, rdrl, zero4, pre);
    else if (wave == 1) accx = l2b_share<1, true>(L, A, rdm, rdr, rdrl, zero4, pre);
    else if (wave == 2) accx = l2b_share<0, false>(L, A, rdm, rdr, rdrl, A.sh[0], pre);
    else accx = l2b_share<1, false>(L, A, rdm, rdr, rdrl, A.sh[S1], pre);
    if (wave < 2) {
      lds_st<f32x4>(lds0 + L.scr + wave * 1024, M::kScratch2Off * 4, accx);
      cbar();
      if (L.a4 == 0) lds_poke_a(lds0 + (M::kFlag2Off + wave) * 4, tag);
    }
  }
  DET(7);
  {
    constexpr int NS = 4 * kL2Chunks, D = RCED_D2X, RING = D + 1;
    Parts b[RING];
    RemRaw rr;
    f32x4 acc[2][2];   // [tile & 1][M-tile]
    const bool g1 = tile_has_gap(wave + 8), g2 = tile_has_gap(wave + 16), g3 = tile_has_gap(wave + 24);
    run_job<NS, D>(
        [&](auto ic) {
          constexpr int i = decltype(ic)::value, t = i / kL2Chunks, c = i % kL2Chunks;
          l2x_load<c>(L.rd2m, L.rd2r, L.rd2rl, t * kT2R, t * 128 * 8, t * 128 * 4, b[i % RING], rr);
        },
        [&](auto ic) {
          constexpr int i = decltype(ic)::value, t = i / kL2Chunks, c = i % kL2Chunks, r = i % RING, u = t & 1;
          if constexpr (c == 2) l2x_merge(b[r], rr, upper);
          acc[u][0] = l2x_mma(A.a[0][c], b[r], c == 0 ? A.sh[0] : acc[u][0]);
          acc[u][1] = l2x_mma(A.a[S1][c], b[r], c == 0 ? A.sh[S1] : acc[u][1]);
          if constexpr (t > 0 && c < 2) {   // the previous tile's results: M-tile 0 behind this tile's first slot, M-tile 1 behind its second
            const bool g = t == 2 ? g1 : t == 3 ? g2 : false;
            if constexpr (c == 0) l2_store<0>(L, acc[u ^ 1][0], L.wr2, (t - 1) * kT2W, g, kVMain + t - 1);
            else l2_store<1>(L, acc[u ^ 1][1], L.wr2, (t - 1) * kT2W, g, kVMain + t - 1);
          }
          if constexpr (t == 3 && c == 2) {
            if (wave == 2 || wave == 3) {
              pflag = lds_peek_a(lds0 + (M::kFlag2Off + wave - 2) * 4);
              cbar();
              part = lds_ld<f32x4>(lds0 + L.scr + (wave - 2) * 1024, M::kScratch2Off * 4);
            }
          }
        },
        pre);
    l2_store<0>(L, acc[1][0], L.wr2, 3 * kT2W, g3, kVMain + 3);
    l2_store<1>(L, acc[1][1], L.wr2, 3 * kT2W, g3, kVMain + 3);
  }
  l2_reduce<M>(L, lds0, wave, tag, err, accx, part, pflag);
}

