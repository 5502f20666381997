// LDS-DMA (global -> LDS without VGPR staging) issued from inline assembly.
//
// Why not __builtin_amdgcn_global_load_lds: hipcc's waitcnt insertion models the builtin as a FLAT-family access that
// may return through LGKM_CNT, so while one is in flight it treats that counter as out of order and turns EVERY
// s_waitcnt for an LDS read into lgkmcnt(0).  The fused kernels keep the next layer's weight packet in flight during
// the whole current layer, so every "wait for the operands of step s" also drained the prefetches of steps s+1..s+D
// that had just been issued -- the software pipeline of the MFMA passes was silently serialised (one full LDS
// round trip per group of steps).  Issued from asm the transfer is invisible to that pass: LDS waits get exact
// counts again, and completion is awaited explicitly (s_waitcnt vmcnt(0) in layer_end_sync) as it already was.
// global_load_lds_dwordx4: lane i moves 16 bytes from its address to LDS byte address M0 + 16*i.
#pragma once
#include <hip/hip_runtime.h>

namespace rced {

__device__ __forceinline__ void lds_dma16(const float* gsrc_lane, float* lds_dst_wave) {
  const unsigned m0v = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds_dst_wave;
  unsigned saved;   // m0 is saved and restored inside the statement, so the compiler's view of it stays valid
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(saved)
      : "v"(gsrc_lane), "s"(m0v)
      : "memory");
}

// The same with a wave-uniform base address in SGPRs and ONE 32-bit per-lane byte offset (lane * 16 for every chunk of
// every packet): a 64-bit per-lane address per chunk cost a v_lshl_add_u64 per transfer and VGPR pairs that stayed live
// across the tile loop (the bf16 R-CED kernel, capped at 128 VGPRs for two workgroups per CU, spilled them).
// HAZARD: hipcc pads no wait states for instructions inside inline asm.  The SGPR base may have been written by a VALU
// instruction right in front of the statement (v_readlane of a spilled SGPR, v_readfirstlane), and "VALU writes SGPR ->
// VMEM reads that SGPR" needs 5 wait states on this part: the two s_mov + s_nop 3 provide 6 (and cover the 1 wait state
// "SALU writes M0 -> LDS-DMA").  Without them the transfer could read a stale base -- seen as run-to-run differences of
// the bf16 R-CED V2 kernel in some builds, never as a fault.
__device__ __forceinline__ void lds_dma16s(const float* gsrc_wave, unsigned lane_byte_off, float* lds_dst_wave) {
  const unsigned m0v = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds_dst_wave;
  unsigned saved;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 3\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(saved)
      : "v"(lane_byte_off), "s"(gsrc_wave), "s"(m0v)
      : "memory");
}

// One wait state behind a 16-byte global store issued from an epilogue.  Measured on this part (bf16 R-CED V2 kernel, two
// workgroups per CU; tests/tools/rerun_grid.py): the sequence
//     v_cmp_* vcc, ...   /   buffer_store_dwordx4 ...   /   s_and_saveexec_b64 sN, vcc
// -- an EXEC write from a freshly written VCC in the instruction slot right behind the store -- lost lanes of the store
// (run-to-run differences at bf16-rounding size, one skip fragment element at a time); with a single s_nop between the
// store and the EXEC write it never did, and hipcc knows no such hazard.  The shipped kernels did not contain the
// sequence (their masks come from SGPR pairs), but whether they do is the register allocator's choice, so the stores
// carry the wait state themselves.  Cost: one cycle per store.
__device__ __forceinline__ void store_wait_state() { asm volatile("s_nop 0"); }

}  // namespace rced
