// LEGACY forms of the CR-CED kernel (RCED_V3_LEGACY_FORMS builds only: tools/ab.sh history runs; NOT in the product library).
// Included by kernels_fused_v3.h inside namespace rced::v3.  Form 2 = round 4's product: layers 2 + 3 fused on the bf16 pipe,
// the first layer and decode_final on the fp32 MFMA.
#pragma once
template <>
struct Map<2> {
  static constexpr bool X6 = true, kX6 = true, kFused = true, kAllX6 = false;
  static constexpr int kB8Off = 0;
  static constexpr bool kL1X6 = RCED_T_L1X6 != 0;
  static constexpr int kB18Off = kB8Off + (kL1X6 ? 3 * kB8PlaneBytes / 4 : kB8Rows * kB8S);
  // B18 here: three blocks (h, m, l), each = the plane [pixel][16] bf16 (32-byte rows) followed by the remainder channels' rows
  // [c16 c17] (4 bytes per pixel).  With the same stride between the parts of both, the last K = 32 chunk of layer 2 is, for EVERY
  // lane, four consecutive dwords from one per-lane address (lower lanes: tap 4 of the plane; upper lanes: the remainder channels'
  // window) + the part's stride: four ds_read_b32 per part straight into the fragment, no select (the other X6 form reads 16 + 8 + 4
  // bytes and merges them with 12 v_cndmask).
  static constexpr int kRemOff = kB18Rows * 32;                                   // the remainder rows inside a block
  static constexpr int kRemRows = kB18Rows + 6;
  static constexpr int kPlaneBytes = ((kRemOff + kRemRows * 4 + 15) / 16) * 16;   // stride between the parts: 19,248
  static constexpr int kRemHMBytes = 0, kRemLBytes = 0;                           // (the other X6 form's layout)
  static constexpr int kB18Bytes = 3 * kPlaneBytes;
  // one weight region: layer 2's A fragments + shifts, then layer 3's (the block's stream images of both, one LDS-DMA during layer 1).
  // (Reads that run past B18 land here: masked pixels only.)
  static constexpr int kWOff = kB18Off + kB18Bytes / 4;
  static constexpr int kW3TOff = kWOff + kG2;
  static constexpr int kWRegions = 1;
  static constexpr int kB30Off = kWOff;                     // (no such buffer: make_lane's unused layer-2/3 addresses of the other forms)
  static constexpr int kX0Off = kW3TOff + kW3T;             // the buffers below have places of their own: nothing aliases B18, nothing is re-zeroed
  static constexpr int kHOff = kX0Off + kX0Floats;
  static constexpr int kFin128Off = kHOff + kHPix * kHS;
  static constexpr int kEdgeOff = kFin128Off + kFin128;     // [frame 4][direction 2][lane 64] x 8 bytes: the partial sums that cross the middle of a frame
  static constexpr int kEdgeFlagOff = kEdgeOff + 4 * 2 * 128;
  static constexpr int kLdsFloats = kEdgeFlagOff + 8;
  static constexpr int kLdsBytes = kLdsFloats * 4;
  static_assert(kLdsBytes <= 160 * 1024, "LDS budget");
  static_assert((kWOff * 4) % 16 == 0 && (kW3TOff * 4) % 16 == 0 && (kB18Off * 4) % 16 == 0 && (kHOff % 2) == 0, "aligned buffers");
  static_assert((kFin128Off * 4) % 16 == 0 && (kEdgeOff * 4) % 16 == 0, "aligned buffers");
  // decode_final's partial sums (8 waves x 2 column tiles x 1 KiB) lie in B8, dead from block 4's layer 1 to the next tile's block 0
  // (whose layers 2 + 3 rewrite every real pixel): 4 KiB in the rows of each frame's first 103 bins, never a gap row
  // (planes: 2 KiB in the first 128 rows of frame w % 4 of plane w / 4)
  static constexpr int finscr0(int w) {
    return kL1X6 ? kB8Off * 4 + (w >> 2) * kB8PlaneBytes + (kB8Pad + kS * (w & 3)) * 16
                 : (kB8Off + (kB8Pad + kS * (w >> 1)) * kB8S) * 4 + 8 * ((w >> 1) & 1) + (w & 1) * 2048;
  }
  static constexpr int kFinScrCt = 1024;
  static constexpr int kT1R = 128 * kB8S * 4, kT1W = 128 * 32;
  static constexpr int kT2R = 0, kT2W = 0, kT3R = 0, kT3W = 0;   // (other forms' layers)
  static constexpr int kTileB18 = 16 * 32;
};
typedef Map<2> MapT;
static_assert(MapT::finscr0(7) % 16 == 0 && MapT::finscr0(2) % 16 == 0 && (MapT::kL1X6 || MapT::finscr0(1) + 2048 <= (kB8Pad + kF) * kB8S * 4), "decode_final's partial sums: 16-byte aligned, inside real rows of B8");


