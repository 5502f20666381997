// Fused-kernel runtime: placeholder until the MFMA kernels land (no variant is fused yet).
#include "../../include/rced.h"
#include "rced_internal.h"

int fused_create(rced_model* m) { m->fused = nullptr; return RCED_OK; }
void fused_destroy(rced_model*) {}
int fused_reserve(rced_model*, int, int) { return RCED_OK; }
int fused_forward(rced_model*, const float*, float*, int, int, hipStream_t) {
  return rced_fail(RCED_ERR_STATE, "no fused path");
}
int fused_set_option(rced_model*, const char*, int) { return RCED_ERR_ARG; }
int fused_get_option(rced_model*, const char*, int*) { return RCED_ERR_ARG; }
