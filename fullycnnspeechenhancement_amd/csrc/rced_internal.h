// Internal model object shared by the C ABI (rced_api.hip) and the fused-kernel runtime.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <vector>

#include "rced_spec.h"

// kernel kinds for the built-in HIP-event profiler ("profile" option)
enum {
  RCED_K_GENERIC = 0,   // conv_layer_generic (layerwise path)
  RCED_K_FUSED = 1,     // fused multi-layer MFMA kernel (the dominant kernel)
  RCED_K_FINAL = 2,     // final 1x129 layer as a Toeplitz GEMM
  RCED_K_COUNT = 3
};

struct rced_layer_dev {
  int cin = 0, cout4 = 0;
  float* w = nullptr;      // [kh,kw,cin,cout4] BN-folded, device
  float* shift = nullptr;  // [cout4] device
  std::vector<float> host_w, host_shift;  // the same, host copies (fused packers read these)
};

struct rced_fused;  // opaque: kernels_fused.hip

struct rced_model {
  int variant = 0, device = 0, num_cus = 0;
  const rced::NetSpec* net = nullptr;
  std::vector<float> host_blob;
  std::vector<rced_layer_dev> layers;
  // layerwise path
  std::vector<int> slot_of_tensor, slot_ch_offset;
  void* workspace = nullptr;
  size_t workspace_bytes = 0;
  // rced_forward_host staging
  void *stage_x = nullptr, *stage_y = nullptr;
  size_t stage_bytes = 0;
  hipStream_t host_streams[3] = {nullptr, nullptr, nullptr};   // rced_forward_host: upload, compute, download
  std::vector<hipEvent_t> host_events;                         // its per-chunk events (upload done / compute done), reused
  int host_chunks = 0;                                         // option "host_chunks" (0 = default 8)
  // options
  int path = 0;
  bool profile = false;
  rced_fused* fused = nullptr;
  // profiler
  struct ProfEvent { int kind; hipEvent_t start, stop; };
  std::vector<ProfEvent> prof_events;
  void prof_begin(int kind, hipStream_t st);
  void prof_end(int kind, hipStream_t st);
  void prof_reset();
  float prof_dominant_ms(int* kind_out, int* launches_out);
  ~rced_model();
};

// fused-kernel runtime (kernels_fused.hip).  fused_create leaves m->fused null when the
// variant has no fused implementation; all return RCED_* codes and set the thread error.
int fused_create(rced_model* m);
void fused_destroy(rced_model* m);
int fused_reserve(rced_model* m, int N, int T);
int fused_forward(rced_model* m, const float* x, float* y, int N, int T, hipStream_t st);
int fused_check(rced_model* m);   // RCED_ERR_STATE if an earlier launch recorded a hand-off time-out
#define RCED_OPT_UNKNOWN (-1)   // fused_set_option: "not a key of mine" (internal; never crosses the C ABI)
int fused_set_option(rced_model* m, const char* key, int value);
int fused_get_option(rced_model* m, const char* key, int* value);
int rced_fail(int code, const char* fmt, ...);
