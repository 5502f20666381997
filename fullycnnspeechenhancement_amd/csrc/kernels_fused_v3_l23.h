// Fused form of the CR-CED kernel (Map<2>): layers 2 (1x5, 18 -> 30) and 3 (1x9, 30 -> 8) as ONE stream on the bf16 matrix pipe.
// Included by kernels_fused_v3.h inside namespace rced::v3 (the design note sits there, at Map<2>).
//
// Per 16-pixel tile a wave issues
//   X: layer 2, three K = 32 chunks x two M-tiles x six three-part products   = 36 MFMAs  -> acc2[M-tile]  (lane (kq, n): 4 channels each)
//      ReLU + split2 of the eight accumulators (44 + 8 VALU)                              -> b3 = layer 3's B fragment, in registers
//   Y: layer 3, five M-tiles of (cout, tap) rows x six products                = 30 MFMAs  -> P[j] (lane (kq, n): couts 2kq, 2kq+1 x taps 2j, 2j+1)
//      shift-add: out[p] += P_t[p + t - 4], 34 v_add_f32_dpp
// Layer 3 runs for TWO tiles at a time (YY): one set of A fragments from LDS per M-tile and pair -- read per tile they were a fifth of
// the forward's time -- and two independent accumulation chains per slot.  Order: X0 X1 X2 YY01 X3 [X4] YY23 [Y4]; tile t's split
// runs between the MFMAs of X(t+1), a pair's shift-adds between its own MFMAs, one M-tile behind.  Output accumulators
// W / X / Y / Z = the pixels of the tile in front of the pair, of its two tiles and of the tile behind it; after a pair, W and X are
// complete (epilogue: shift, ReLU, block skips, one 8-byte store per lane) and Y, Z become the next pair's W, X.
#pragma once
#ifndef RCED_T_A2REG
#define RCED_T_A2REG 3   // layer 2's A fragments: 0 = all from LDS, one slot ahead; 1 = M-tile 0 in registers (36, loaded from global memory during
                         // layer 1), M-tile 1 from LDS; 2 = all in registers; 3 = as 1, but the registers are filled from LDS (the block's image is
                         // there anyway) when the phase starts: eight waves x 9 KB less on the vector-memory path during layer 1
#endif
#ifndef RCED_T_PRIO
#define RCED_T_PRIO 1    // the waves with five tiles (4..7) run this phase at raised priority: their SIMD partners (four tiles) are the older
                         // waves, which the issue arbiter prefers -- left alone they finish at two thirds of the phase and the rest runs single
#endif
#ifndef RCED_T_A3D
#define RCED_T_A3D 1     // (2 measured: no gain -- it is not their latency) layer 3's A fragments are read this many slots ahead of their use (ring of RCED_T_A3D + 1)
#endif
#ifndef RCED_T_SGB
#define RCED_T_SGB 1     // sched_group_barrier pattern inside the slots (see interleave())
#endif
#ifndef RCED_T_L1PF
#define RCED_T_L1PF 1    // all-x6 form, layer 1: a job's first operands are read in front of the previous job's epilogue (layer1_x6l)
#endif
#ifndef RCED_T_EXP
#define RCED_T_EXP 0   // timing experiments only (WRONG RESULTS): 1 = no shift-adds, 2 = layer 3's A fragments read once (round 5: this build's
                       // 10 % are not the reads -- with equal fragments hipcc merges the MFMAs of M-tiles 2..4 into those of 0, 1: 252 of the
                       // kernel's 1,218 static MFMAs disappear), 4 = no split arithmetic, 8 = layer 2's B fragments read once per tile,
                       // 256 = a third of layer 1's B reads, 512 = no weight transfers, 4096 = layer 1 without its main tiles' epilogues, 8192 = layer 1's pair jobs only
#endif
#if (RCED_T_EXP != 0 || RCED_X6_EXP != 0) && !defined(RCED_TIMING_ONLY)
#error "RCED_T_EXP / RCED_X6_EXP builds compute wrong results: timing experiments only (tools/mkexp.sh ... -DRCED_TIMING_ONLY -DRCED_T_EXP=...)"
#endif

// a copy of v shifted along the 16-lane row, zero where the source lane lies outside the row
template <int CTRL>
__device__ __forceinline__ float dpp0(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// One tap's partial output p (at this lane's INPUT pixel) goes to output pixel (input pixel + S), S = 4 - tap: into this tile's
// accumulator and, for the |S| lanes that leave the row, into the neighbour's.  row_shr:k = lane n reads lane n - k (0x110 + k),
// row_shl:k = lane n reads lane n + k (0x100 + k).
template <int S>
__device__ __forceinline__ void tap_add(float p, float& oP, float& oC, float& oN) {
  if constexpr (S == 0) {
    oC += p;
  } else if constexpr (S > 0) {
    oC += dpp0<0x110 + S>(p);
    oN += dpp0<0x100 + 16 - S>(p);
  } else {
    oC += dpp0<0x100 - S>(p);
    oP += dpp0<0x110 + 16 + S>(p);
  }
}
// two independent accumulation chains, their six products in lockstep (one chain's MFMAs back to back wait for each other:
// left to itself hipcc sometimes emits the chains one after the other)
__device__ __forceinline__ void mma2(const s16x8 (&a0)[3], const Parts& b0, f32x4& c0, const s16x8 (&a1)[3], const Parts& b1, f32x4& c1) {
  c0 = mfma32(a0[1], b0.m, c0);
  c1 = mfma32(a1[1], b1.m, c1);
  c0 = mfma32(a0[2], b0.h, c0);
  c1 = mfma32(a1[2], b1.h, c1);
  c0 = mfma32(a0[0], b0.l, c0);
  c1 = mfma32(a1[0], b1.l, c1);
  c0 = mfma32(a0[1], b0.h, c0);
  c1 = mfma32(a1[1], b1.h, c1);
  c0 = mfma32(a0[0], b0.m, c0);
  c1 = mfma32(a1[0], b1.m, c1);
  c0 = mfma32(a0[0], b0.h, c0);
  c1 = mfma32(a1[0], b1.h, c1);
}
// the order of a slot's instructions: NM times {one MFMA, up to NV VALU} (the VALU of a slot are the previous tile's split / the
// previous M-tile's shift-adds: hipcc otherwise puts them in one block behind the slot's last MFMA, where nothing covers them)
// The next slot's LDS reads ride between the MFMAs too (ND per MFMA): issued in one block at the slot's start they are a stretch of
// the wave's in-order issue in which none of its MFMAs can go (layer 3's three reads per slot cost 0.6 ms of the forward that way).
template <int NM, int NV, int ND = 0>
__device__ __forceinline__ void interleave() {
#if RCED_T_SGB
#pragma unroll
  for (int i = 0; i < NM; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    if (ND > 0) __builtin_amdgcn_sched_group_barrier(0x100, ND, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
  }
#endif
}
template <int J>
__device__ __forceinline__ void shift_add(f32x4 pj, float& p0, float& p1, float& c0, float& c1, float& n0, float& n1) {
  if (RCED_T_EXP & 1) {
    c0 += pj.x;
    c1 += pj.z;
    return;
  }
  tap_add<4 - 2 * J>(pj.x, p0, c0, n0);
  if constexpr (2 * J + 1 < 9) tap_add<3 - 2 * J>(pj.y, p0, c0, n0);   // (tap 9 is padding: zero weights)
  tap_add<4 - 2 * J>(pj.z, p1, c1, n1);
  if constexpr (2 * J + 1 < 9) tap_add<3 - 2 * J>(pj.w, p1, c1, n1);
}

// `sp(IC<j>)`, j < 5: called once from each layer-3 slot of the wave's LAST job (by then layer 2's operand registers are free):
// the register-bound loads of what comes next.
// `dma1()`: called once, behind the first operand reads (RCED_T_L1X6: the next layer 1's images, LDS-DMA'd into areas that are dead now)
template <class M, bool LAST, class Sp, class Dma1>
__device__ __forceinline__ void layer23(const Params& P, const Lane& L, unsigned lds0, unsigned wbase, const A2Regs& A, int blk, int wave,
                                        unsigned tag, f32x2 (&sk1)[5], f32x2 (&sk2)[5], Sp sp, Dma1 dma1 DET_ARG) {
  DET_BEGIN();
  const bool right = wave >= 4;
  const int fr = wave & 3;
  // A fragments of both layers come from LDS (the block's images, LDS-DMA'd during layer 1), one slot ahead of their use: held in
  // registers (72 for layer 2) the kernel spilled; beside bf16 MFMAs six more conflict-free ds_read_b128 per slot cost next to nothing
  // (tools/micro/bf16_stream_rate.hip: 3 -> 9 reads per 12-MFMA slot: +0 .. 5 %)
  const unsigned a2 = wbase + L.scr, a3 = a2 + kG2 * 4;                           // this lane's 16 bytes of every A fragment
  const f32x4 sh2[2] = {lds_ld<f32x4>(wbase + L.kq16, kG2Data * 4), lds_ld<f32x4>(wbase + L.kq16, (kG2Data + 16) * 4)};
  const f32x2 sh3 = lds_ld<f32x2>(wbase + (L.kq16 >> 1), (kG2 + kW3TData) * 4);   // shift[2kq], shift[2kq + 1]
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const f32x2 zero2 = {0.f, 0.f};
  Parts b2[2];
  s16x8 a2m0[3][3];            // RCED_T_A2REG = 3: M-tile 0's fragments [chunk][part], read from LDS once per phase
  if constexpr (RCED_T_A2REG == 3) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int q = 0; q < 3; ++q) a2m0[c][q] = lds_ld<s16x8>(a2, ((2 * c) * 3 + q) * 1024);
  }
  s16x8 a2r[2][2][3];          // [ring][M-tile][part]
  unsigned rdc = L.rd2c;       // the last chunk's four dwords of the tile whose layer 2 comes next
  f32x4 acc2[2][2];            // [tile & 1][M-tile]
  u32x4 b3h[2], b3m[2], b3l[2];   // [tile & 1]: layer 3's B fragment (three parts)
  s16x8 a3r[RCED_T_A3D + 1][3];
  f32x4 pj[2][2], p4[2] = {zero4, zero4};   // [tile & 1]: P of M-tiles 0..3 (ring of two) / of M-tile 4, whose shift-adds run from inside the NEXT slot
  // (fresh accumulators hold -0.0, the identity of the IEEE addition: `-0.0 + x` folds to x, `0.0 + x` does not -- hipcc emitted a
  // v_mov_b32_dpp + v_add_f32 0 pair for every first contribution)
  float w0 = -0.f, w1 = -0.f, x0 = -0.f, x1 = -0.f, y0 = -0.f, y1 = -0.f, z0 = -0.f, z1 = -0.f;
  f32x2 hold = zero2;

  auto ldX = [&](auto tc, auto cc) {
    constexpr int t = decltype(tc)::value, c = decltype(cc)::value;
    if ((RCED_T_EXP & 8) && (c > 0 || t > 1)) return;
    Parts& b = b2[(3 * t + c) & 1];
#pragma unroll
    for (int mt = (RCED_T_A2REG == 3 ? 1 : RCED_T_A2REG); mt < 2; ++mt)
#pragma unroll
      for (int q = 0; q < 3; ++q) a2r[(3 * t + c) & 1][mt][q] = lds_ld<s16x8>(a2, ((2 * c + mt) * 3 + q) * 1024);
    if constexpr (c < 2) {
      constexpr int off = t * M::kTileB18 + (M::kAllX6 ? 32 : 64) * c;   // 16 rows per tile, two rows (taps) per chunk
      b.h = lds_ld<s16x8>(L.rd2m, off);
      b.m = lds_ld<s16x8>(L.rd2m, off + M::kPlaneBytes);
      b.l = lds_ld<s16x8>(L.rd2m, off + 2 * M::kPlaneBytes);
    } else {   // the same four dwords per part for every lane (see Map<2>): plane rows for the lower lanes, the remainder channels' for the upper
      u32x4 h, m, l;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        h[i] = lds_ld<unsigned>(rdc, 4 * i);
        m[i] = lds_ld<unsigned>(rdc, 4 * i + M::kPlaneBytes);
        l[i] = lds_ld<unsigned>(rdc, 4 * i + 2 * M::kPlaneBytes);
      }
      b.h = __builtin_bit_cast(s16x8, h);
      b.m = __builtin_bit_cast(s16x8, m);
      b.l = __builtin_bit_cast(s16x8, l);
      rdc += L.rd2cs;
    }
  };
  auto doX = [&](auto tc, auto cc) {
    constexpr int t = decltype(tc)::value, c = decltype(cc)::value, r = (3 * t + c) & 1, u = t & 1;
    if constexpr (c == 0) {
      acc2[u][0] = sh2[0];
      acc2[u][1] = sh2[1];
    }
    mma2(RCED_T_A2REG == 3 ? a2m0[c] : RCED_T_A2REG >= 1 ? A.a[0][c] : a2r[r][0], b2[r], acc2[u][0], RCED_T_A2REG == 2 ? A.a[1][c] : a2r[r][1], b2[r], acc2[u][1]);
  };
  // ReLU + split of one pair of tile t's layer-2 outputs: piece q = 2 * M-tile + half -> k-slots 2q, 2q + 1 of the B fragment
  auto split_piece = [&](auto tc, auto qc) {
    constexpr int t = decltype(tc)::value, q = decltype(qc)::value, u = t & 1, mt = q >> 1;
    float v0 = relu1((q & 1) ? acc2[u][mt].z : acc2[u][mt].x), v1 = relu1((q & 1) ? acc2[u][mt].w : acc2[u][mt].y);
    if constexpr (t == 4) {   // a frame's last tile (waves 4..7 only): one real pixel, the rest is gap / the next frame -> zero
      const bool ok = vbit(L, kVN0);
      v0 = ok ? v0 : 0.f;
      v1 = ok ? v1 : 0.f;
    }
    P3 p;
    if (RCED_T_EXP & 4) {
      p.h = __builtin_bit_cast(unsigned, v0);
      p.m = __builtin_bit_cast(unsigned, v1);
      p.l = p.h;
    } else {
      p = split2(v0, v1);
    }
    b3h[u][q] = p.h;
    b3m[u][q] = p.m;
    b3l[u][q] = p.l;
  };
  // layer 3's A fragments come from LDS RCED_T_A3D slots ahead, through a ring; five M-tiles per job: when one layer-3 job follows
  // another directly, its ring positions are shifted (PAR) -- its first M-tiles are read while the last ones of this job are in use
  auto ldY = [&](auto jc, auto parc) {
    constexpr int j = decltype(jc)::value, r = (j + decltype(parc)::value) % (RCED_T_A3D + 1);
    if ((RCED_T_EXP & 2) && j > 1) return;
#pragma unroll
    for (int q = 0; q < 3; ++q) a3r[r][q] = lds_ld<s16x8>(a3, (((RCED_T_EXP & 32) ? 0 : j) * 3 + ((RCED_T_EXP & 64) ? 0 : q)) * 1024);
  };
  auto doY = [&](auto uc, auto jc, auto parc) {   // M-tile j of the tile in buffer u
    constexpr int u = decltype(uc)::value, j = decltype(jc)::value, r = (j + decltype(parc)::value) % (RCED_T_A3D + 1);
    Parts b;
    b.h = __builtin_bit_cast(s16x8, b3h[u]);
    b.m = __builtin_bit_cast(s16x8, b3m[u]);
    b.l = __builtin_bit_cast(s16x8, b3l[u]);
    if constexpr (j == kL3MT - 1) p4[u] = l2x_mma(a3r[r], b, zero4);
    else pj[u][j & 1] = l2x_mma(a3r[r], b, zero4);
  };
  auto doYY = [&](auto jc, auto parc) {   // M-tile j of both tiles of a pair (buffers 0 and 1), the two chains in lockstep
    constexpr int j = decltype(jc)::value, r = (j + decltype(parc)::value) % (RCED_T_A3D + 1);
    Parts ba, bb;
    ba.h = __builtin_bit_cast(s16x8, b3h[0]);
    ba.m = __builtin_bit_cast(s16x8, b3m[0]);
    ba.l = __builtin_bit_cast(s16x8, b3l[0]);
    bb.h = __builtin_bit_cast(s16x8, b3h[1]);
    bb.m = __builtin_bit_cast(s16x8, b3m[1]);
    bb.l = __builtin_bit_cast(s16x8, b3l[1]);
    f32x4 ca = zero4, cb = zero4;
    mma2(a3r[r], ba, ca, a3r[r], bb, cb);
    if constexpr (j == kL3MT - 1) {
      p4[0] = ca;
      p4[1] = cb;
    } else {
      pj[0][j & 1] = ca;
      pj[1][j & 1] = cb;
    }
  };
  // epilogue of local tile t (its sums are complete): shift, ReLU, block skips (model.py:84-88: CE1 / CE2 outputs are kept and
  // added to CD2 / CD1 AFTER the ReLU), store -- to B8, block 4 to decode_final's H image
  auto finish = [&](auto tc, float s0, float s1) {
    constexpr int t = decltype(tc)::value;
    f32x2 v = {relu1(s0 + sh3.x), relu1(s1 + sh3.y)};
    if constexpr (LAST) {
      v += sk1[t];
    } else {
      v += blk == 3 ? sk2[t] : zero2;
      sk1[t] = blk == 0 ? v : sk1[t];
      sk2[t] = blk == 1 ? v : sk2[t];
    }
    if constexpr (M::kAllX6 && LAST) {      // decode_final's image H': three bf16 planes, bin b at row b + (b >> 4) of its frame (Map<3>)
      const P3 p = split2(v.x, v.y);
      if (t < 4 || vbit(L, kVN0)) {
        lds_st<unsigned>(L.wh0, t * 272, p.h);
        lds_st<unsigned>(L.wh0, t * 272 + kHPlaneBytes, p.m);
        lds_st<unsigned>(L.wh0, t * 272 + 2 * kHPlaneBytes, p.l);
      }
    } else if constexpr (RCED_T_L1X6 && !LAST) {   // the next layer 1 runs on the bf16 pipe: its input as three bf16 planes, 16-byte rows
      const P3 p = split2(v.x, v.y);
      if (t < 4 || vbit(L, kVN0)) {
        lds_st<unsigned>(L.wr3p, t * 256, p.h);
        lds_st<unsigned>(L.wr3p, t * 256 + kB8PlaneBytes, p.m);
        lds_st<unsigned>(L.wr3p, t * 256 + 2 * kB8PlaneBytes, p.l);
      }
    } else {
      if (t < 4 || vbit(L, kVN0)) lds_st<f32x2>(LAST ? L.wh0 : L.wr3, t * (16 * kB8S * 4), v);
    }
  };
  static_assert(kHS == kB8S, "one tile stride for both destinations");
  const unsigned edge = lds0 + 4 * M::kEdgeOff + fr * 1024 + L.a8, eflag = lds0 + 4 * M::kEdgeFlagOff + fr * 8;
  auto publish = [&](int dir, float v0, float v1) {   // LDS operations of a wave execute in order: data, then flag
    lds_st<f32x2>(edge, dir * 512, f32x2{v0, v1});
    cbar();
    if (L.a4 == 0) lds_poke_a(eflag + dir * 4, tag);
  };
  auto fetch = [&](int dir) {
    flag_wait(eflag + dir * 4, tag, P.err, 8u);
    return lds_ld<f32x2>(edge, dir * 512);
  };
  // what follows a pair's last MFMAs, run from inside the NEXT slot: its last shift-adds; then the tile in front of the pair
  // and the pair's first tile are complete
  auto tail_pair = [&](auto ac) {
    constexpr int a = decltype(ac)::value;   // the pair's first local tile (0 or 2)
    shift_add<4>(p4[0], w0, w1, x0, x1, y0, y1);
    shift_add<4>(p4[1], x0, x1, y0, y1, z0, z1);
    if constexpr (a == 0) {
      if (right) {
        publish(1, w0, w1);                // tile 4's share of tile 3's last pixels (the left wave's last tile)
        hold = f32x2{x0, x1};              // tile 4 itself waits for tile 3's share
      } else {
        finish(IC<0>{}, x0, x1);
      }
    } else {
      finish(IC<a - 1>{}, w0, w1);
      finish(IC<a>{}, x0, x1);
    }
    w0 = y0;
    w1 = y1;
    x0 = z0;
    x1 = z1;
    y0 = y1 = z0 = z1 = -0.f;
  };
  // X(t) with tile t - 1's split between its MFMAs.  NEXT: what the slot behind it is: 0 = X(t + 1), 1 = a layer-3 job
  auto runX = [&](auto tc, auto nc, auto prev_tail) {
    constexpr int t = decltype(tc)::value, next = decltype(nc)::value;
    static_for<0, 3>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      if constexpr (c < 2) ldX(tc, IC<c + 1>{});
      else if constexpr (next == 0) ldX(IC<t + 1>{}, IC<0>{});
      if constexpr (next == 1 && c + RCED_T_A3D >= 3) ldY(IC<c + RCED_T_A3D - 3>{}, IC<0>{});   // the layer-3 job behind this one: its first M-tile(s)
      if (RCED_T_SGB < 2) pin();
      doX(tc, cc);
      if constexpr (c == 0) prev_tail();
      if constexpr (t > 0) {
        if constexpr (c == 0) {
          split_piece(IC<t - 1>{}, IC<0>{});
          split_piece(IC<t - 1>{}, IC<1>{});
        } else if constexpr (c == 1) {
          split_piece(IC<t - 1>{}, IC<2>{});
          split_piece(IC<t - 1>{}, IC<3>{});
        }
      }
      interleave<12, (c == 0 ? 5 : 3), (RCED_T_SGB >= 2 ? 2 : 0)>();
      pin();
    });
  };
  // layer 3 of the pair of tiles (a, a + 1) (PAIR) or of the single tile a.  NEXT: 0 = X(NX) follows, 1 = another layer-3 job, 2 = nothing
  auto runY = [&](auto ac, auto pairc, auto nc, auto nxc, auto parc, auto prev_tail) {
    constexpr int a = decltype(ac)::value, next = decltype(nc)::value, u0 = a & 1;
    constexpr bool pair = decltype(pairc)::value != 0;
    static_for<0, kL3MT>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      constexpr int D = RCED_T_A3D, par = decltype(parc)::value, npar = (kL3MT + par) % (D + 1);   // (npar: the ring phase of a layer-3 job that follows directly)
      if constexpr (j + D < kL3MT) ldY(IC<j + D>{}, parc);
      else if constexpr (next == 1) ldY(IC<j + D - kL3MT>{}, IC<npar>{});
      if constexpr (next == 0 && j == kL3MT - 1) ldX(nxc, IC<0>{});
      if (RCED_T_SGB < 2) pin();
      if constexpr (pair) doYY(jc, parc);
      else doY(IC<u0>{}, jc, parc);
      if constexpr (j == 0) {
        prev_tail();
      } else {
        shift_add<j - 1>(pj[u0][(j - 1) & 1], w0, w1, x0, x1, y0, y1);
        if constexpr (pair) shift_add<j - 1>(pj[u0 ^ 1][(j - 1) & 1], x0, x1, y0, y1, z0, z1);
      }
      if constexpr (next == 2) sp(jc);
      interleave<(pair ? 12 : 6), (j == 0 ? 4 : 2), (RCED_T_SGB >= 2 ? 1 : 0)>();
      pin();
    });
  };
  auto none = [] {};
  const IC<0> i0;
  const IC<1> i1;
  const IC<2> i2;
  const IC<3> i3;
  const IC<4> i4;

  if (RCED_T_PRIO && right) __builtin_amdgcn_s_setprio(RCED_T_PRIO);
  // ---- the stream.  Waves 0..3: tiles 0..3 of frame fr (X0 X1 X2 YY01 X3 YY23); waves 4..7: tiles 4..8 (X0 X1 X2 YY01 X3 X4 YY23 Y4)
  ldX(i0, i0);
  pin();
  if (!(RCED_T_EXP & 512)) dma1();   // (512: timing experiment, wrong results: no weight transfers at all)
  pin();
  runX(i0, i0, none);
  runX(i1, i0, none);
  runX(i2, i1, none);
  DET(0);
  runY(i0, i1, i0, i3, i0, none);                          // YY01; X3 follows
  DET(1);
  if (!right) {
    runX(i3, i1, [&] { tail_pair(i0); });
    DET(2);
    static_for<0, 4>([&](auto qc) { split_piece(i3, qc); });
    runY(i2, i1, i2, i0, i0, none);                        // YY23
    tail_pair(i2);                                         // finishes tiles 1, 2; w = tile 3 so far, x = its share of tile 4's first pixels
    publish(0, x0, x1);
    const f32x2 e = fetch(1);                              // published behind the right wave's FIRST pair: long there
    finish(i3, w0 + e.x, w1 + e.y);
  } else {
    runX(i3, i0, [&] { tail_pair(i0); });
    runX(i4, i1, none);
    DET(2);
    runY(i2, i1, i1, i0, i0, none);                        // YY23; Y4 follows
    static_for<0, 4>([&](auto qc) { split_piece(i4, qc); });
    runY(i4, i0, i2, i0, IC<kL3MT % (RCED_T_A3D + 1)>{}, [&] { tail_pair(i2); });   // Y4: w = local tile 3, x = local tile 4 (nothing lies behind it: the gap)
    shift_add<4>(p4[0], w0, w1, x0, x1, y0, y1);
    finish(i3, w0, w1);
    finish(i4, x0, x1);
    const f32x2 e = fetch(0);                              // the left wave has four tiles, this one five: long there
    finish(i0, hold.x + e.x, hold.y + e.y);
    if (RCED_T_PRIO) __builtin_amdgcn_s_setprio(0);
  }
  DET(3);
}


// ---- layer 1 of blocks 1..4 (1x9, 8 -> 18) on the bf16 pipe (RCED_T_L1X6) --------------------------------------------------
// Input: the 8-channel tensor as three bf16 planes [pixel][8] (16-byte rows: one pixel's channels = one octet of the K axis),
// written by layers 2 + 3's epilogue.  K = 72 in three K = 32 chunks: k-slot 8kq + e = tap 4c + kq, channel e (taps 9..11: zero
// weights): a lane's B fragment of a chunk is ONE aligned ds_read_b128 per part.  Channels 0..15 = one M-tile, 18 MFMAs per
// 16-pixel tile; channels 16, 17 by the remainder pass (rows = 8 pixel phases x 2 channels, K = 16 window taps x 8 channels = four
// chunks, 24 MFMAs per tile of 16 columns x SEVEN pixels: phase 7's rows are dropped -- a column stride of 7 rows = 112 bytes keeps
// its reads free of bank conflicts (make_lane), 8 rows = 128 bytes made every one of them 8-way conflicted).
__device__ __forceinline__ Parts b8_load(unsigned rd, int off) {
  Parts b;
  b.h = lds_ld<s16x8>(rd, off);
  if (RCED_T_EXP & 256) {   // timing experiment (wrong results): a third of layer 1's B reads
    b.m = b.l = b.h;
    return b;
  }
  b.m = lds_ld<s16x8>(rd, off + kB8PlaneBytes);
  b.l = lds_ld<s16x8>(rd, off + 2 * kB8PlaneBytes);
  return b;
}

// ---- its A fragments live in LDS ---------------------------------------------------------------------------------------------
// The 21 one-KiB pieces of a block's layer-1 image (main pass 9, remainder pass 12) + its shifts are LDS-DMA'd during the PREVIOUS
// block's layers 2 + 3 into areas that are dead from then until this layer 1 has run: the input rows of the first layer (pieces 0..4 +
// the shifts) and the bins of decode_final's image (four pieces per frame; its zero pads are not touched).  No weight registers at
// all: a version with the fragments in registers (42 + 48, loaded from global memory like the first layer's) was parity-exact and
// spilled (192 .. 336 B, the skip arrays reloaded inside the fused phase's epilogues: 7.7 ms against 6.15 for this one).
// (All-x6 form: a region of its own, the 21 pieces back to back and the shifts behind them.)
template <class M>
constexpr int a1x_off(int i) {   // byte offset of piece i from the start of LDS
  if constexpr (M::kAllX6) return M::kW1Off * 4 + i * 1024;
  else return i < 5 ? M::kX0Off * 4 + i * 1024
                    : M::kHOff * 4 + (kHFrame * ((i - 5) / 4) + 64) * kHS * 4 + 8 * (((i - 5) / 4) & 1) + ((i - 5) % 4) * 1024;
}
template <class M>
constexpr int a1x_shift_off() {
  if constexpr (M::kAllX6) return M::kW1Off * 4 + 21 * 1024;
  else return M::kX0Off * 4 + 5 * 1024;
}
#if RCED_V3_LEGACY_FORMS
static_assert(a1x_shift_off<MapT>() + 128 <= (MapT::kX0Off + kX0Floats) * 4 && a1x_off<MapT>(20) + 1024 <= (MapT::kHOff + (kHFrame * 3 + 64 + kF) * kHS) * 4 &&
              a1x_off<MapT>(8) % 16 == 0 && a1x_off<MapT>(9) % 16 == 0 && a1x_off<MapT>(13) % 16 == 0 && a1x_off<MapT>(17) % 16 == 0,
              "layer 1's LDS-resident images: inside the input-row area / the real bins of the H image, 16-byte aligned");
#endif
// the DMA of one block's image (src = its first float in the weight stream): 22 chunks over the 8 waves
template <class M>
__device__ __forceinline__ void a1x_dma(const float* src, float* lds, int wave, int lane) {
  lane = opaque(lane);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int ci = wave + 8 * k;
    if (ci < 21) {
      lds_dma16s(src + ci * 256, (unsigned)lane * 16u, lds + a1x_off<M>(ci) / 4);
    } else if (ci == 21) {
      if (lane < 8) lds_dma16s(src + 21 * 256, (unsigned)lane * 16u, lds + a1x_shift_off<M>() / 4);
    }
  }
}
template <class M, int I>
__device__ __forceinline__ s16x8 a1x_ld(unsigned aX, unsigned aH) {   // aX / aH: this lane's 16 bytes in the input-row area / the H image (all-x6 form: aX = its own region)
  if constexpr (I < 5 || M::kAllX6) return lds_ld<s16x8>(aX, I * 1024);
  else return lds_ld<s16x8>(aH, a1x_off<M>(I) - M::kHOff * 4);
}
template <class M, class Dma, class Sp>
__device__ __forceinline__ void layer1_x6l(const Lane& L, unsigned lds0, int role, Dma dma, Sp sp DET_ARG) {
  DET_BEGIN();
  constexpr int kTW = M::kT1W, kTR = 128 * 16;
  const unsigned aX = lds0 + (M::kAllX6 ? a1x_off<M>(0) : M::kX0Off * 4) + L.scr, aH = lds0 + M::kHOff * 4 + L.scr;
  const f32x4 sh = lds_ld<f32x4>(lds0 + a1x_shift_off<M>() + L.kq16, 0);
  const f32x2 s2 = lds_ld<f32x2>(lds0 + a1x_shift_off<M>(), 64);
  auto pre = once(dma);
  auto lda = [&](auto pc, s16x8 (&a)[3]) {   // the three parts of piece group G = pieces 3G..3G+2
    constexpr int g = decltype(pc)::value;
    a[0] = a1x_ld<M, 3 * g>(aX, aH);
    a[1] = a1x_ld<M, 3 * g + 1>(aX, aH);
    a[2] = a1x_ld<M, 3 * g + 2>(aX, aH);
  };
  // The first operands of a wave's NEXT job are read in front of the epilogue of the current one (all-x6 form): a job that starts by
  // issuing its reads waits a full LDS round trip with nothing of its own to issue (the single-tile and remainder jobs ran at a third
  // of their MFMA rate); behind 26 .. 52 VALU of split + stores the reads have landed when the job begins.
  constexpr bool kPF = M::kAllX6 && RCED_T_L1PF;
  const bool has_single = role < 2 || role == 7, has_rem = role >= 4;
  const unsigned rd_single = L.rd1x + (role == 0 ? 32 : role == 1 ? 30 : 16) * 256;
  Parts nb;
  s16x8 na[3];
  auto prefetch = [&](unsigned rd, int aoff) {   // aoff: byte offset of the job's first piece group (single tile: pieces 0..2, remainder: 9..11)
    nb = b8_load(rd, 0);
#pragma unroll
    for (int q = 0; q < 3; ++q) na[q] = lds_ld<s16x8>(aX + aoff, q * 1024);
  };
  {
    Parts b[2][2];
    s16x8 a[3][3];     // the main pass's nine fragments, read once per job
    f32x4 acc[2][2];   // [pair][tile]
    const bool two = role != 7;
    const bool g1 = tile_has_gap(role + 8), g2 = tile_has_gap(role + 16), g3 = tile_has_gap(role + 24);
    auto ld = [&](auto ic) {
      constexpr int i = decltype(ic)::value, p = i / 3, c = i % 3;
      if constexpr (p == 0) lda(IC<c>{}, a[c]);
      b[i & 1][0] = b8_load(L.rd1x, 2 * p * kTR + 64 * c);
      b[i & 1][1] = b8_load(L.rd1xb, 2 * p * kTR + 64 * c);
    };
    auto slot = [&](auto ic) {
      constexpr int i = decltype(ic)::value, p = i / 3, c = i % 3;
      if constexpr (c == 0) acc[p][0] = acc[p][1] = sh;
      mma2(a[c], b[i & 1][0], acc[p][0], a[c], b[i & 1][1], acc[p][1]);
      if constexpr (p == 0) static_for<0, 3>([&](auto qc) { sp(IC<12 + 3 * i + decltype(qc)::value>{}); });
      if constexpr (p == 1 && c == 0) l1_store<M>(L, acc[0][0], L.wr1, 0, false, kVMain);
      if constexpr (p == 1 && c == 1) l1_store<M>(L, acc[0][1], L.wr1, kTW, g1, kVMain + 1);
      interleave<12, 3>();
    };
    ld(IC<0>{});
    pin();
    if (!(RCED_T_EXP & 512)) pre();
    pin();
    static_for<0, 3>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if constexpr (i < 2) ld(IC<i + 1>{});
      else if (two) ld(IC<3>{});
      pin();
      slot(ic);
      pin();
      DETX(i);
    });
    if (two) {
      static_for<3, 6>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (i < 5) ld(IC<i + 1>{});
        pin();
        slot(ic);
        pin();
        DETX(i);
      });
      if constexpr (kPF) {
        if (has_single) prefetch(rd_single, 0);
        else if (has_rem) prefetch(L.rd1xr, 9 * 1024);
        pin();
      }
      l1_store<M>(L, acc[1][0], L.wr1, 2 * kTW, g2, kVMain + 2);
      l1_store<M>(L, acc[1][1], L.wr1, 3 * kTW, g3, kVMain + 3);
      DETX(6);
    } else {
      if constexpr (kPF) {
        prefetch(rd_single, 0);   // role 7: its single tile is next
        pin();
      }
      l1_store<M>(L, acc[0][0], L.wr1, 0, false, 0);
      l1_store<M>(L, acc[0][1], L.wr1, kTW, false, 0);
    }
  }
  DET(6);
  if (RCED_T_EXP & 8192) return;   // timing experiment (wrong results): the pair jobs only
  if (role < 2 || role == 7) {   // the single main tile
    const int dt = role == 0 ? 32 : role == 1 ? 30 : 16;
    const unsigned rd = L.rd1x + dt * 256;
    Parts b[2];
    s16x8 a[2][3];
    f32x4 acc = sh;
    if constexpr (kPF) {
      b[0] = nb;
      a[0][0] = na[0];
      a[0][1] = na[1];
      a[0][2] = na[2];
    } else {
      b[0] = b8_load(rd, 0);
      lda(IC<0>{}, a[0]);
    }
    pin();
    static_for<0, 3>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      if constexpr (c < 2) {
        b[(c + 1) & 1] = b8_load(rd, 64 * (c + 1));
        lda(IC<c + 1>{}, a[(c + 1) & 1]);
      }
      pin();
      acc = l2x_mma(a[c & 1], b[c & 1], acc);
      pin();
    });
    if constexpr (kPF) {
      if (has_rem) prefetch(L.rd1xr, 9 * 1024);   // role 7: its first remainder tile is next
      pin();
    }
    l1_store<M>(L, acc, L.wr1 + dt * M::kTileB18, 0, false, 0);
  }
  DET(5);
  {   // remainder tiles
    const int nrem = role < 4 ? 0 : role == 7 ? 2 : 1;
    unsigned rdr = L.rd1xr, wrr = L.wr1r;
    int xr = role == 7 ? 3 : role - 4, vb = kVRem;
    const f32x4 init = {s2.x, s2.y, s2.x, s2.y};
#pragma unroll 1
    for (int r = 0; r < nrem; ++r) {
      Parts b[2];
      s16x8 a[2][3];
      f32x4 acc = init;
      if (kPF && r == 0) {
        b[0] = nb;
        a[0][0] = na[0];
        a[0][1] = na[1];
        a[0][2] = na[2];
      } else {
        b[0] = b8_load(rdr, 0);
        lda(IC<3>{}, a[0]);
      }
      pin();
      static_for<0, 4>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        if constexpr (c < 3) {
          b[(c + 1) & 1] = b8_load(rdr, 64 * (c + 1));
          lda(IC<3 + c + 1>{}, a[(c + 1) & 1]);
        }
        pin();
        acc = l2x_mma(a[c & 1], b[c & 1], acc);
        pin();
      });
      const f32x4 v = relu4(acc);
      const bool va = xr == 0 || vbit(L, vb), vbb = vbit(L, vb + 1);   // (vb + 1: also "not phase 7")
      const P3 pa = split2(v.x, v.y), pb = split2(v.z, v.w);
      if (va) {
        lds_st<unsigned>(wrr, 0, pa.h);
        lds_st<unsigned>(wrr, M::kPlaneBytes, pa.m);
        lds_st<unsigned>(wrr, 2 * M::kPlaneBytes, pa.l);
      }
      if (vbb) {
        lds_st<unsigned>(wrr, 4, pb.h);
        lds_st<unsigned>(wrr, M::kPlaneBytes + 4, pb.m);
        lds_st<unsigned>(wrr, 2 * M::kPlaneBytes + 4, pb.l);
      }
      wrr += 112 * 4;
      rdr += 112 * 16;
      xr += 1;
      vb += 2;
    }
  }
  DET(4);
}
