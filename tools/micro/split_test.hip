#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
struct Parts { s16x8 h, m, l; };
__device__ __forceinline__ Parts split8(const f32x2 (&q)[4]) {
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  s16x2 ph[4], pm[4], pl[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const bf16x2 bh = {(__bf16)q[j].x, (__bf16)q[j].y};
    const f32x2 r1 = {q[j].x - (float)bh.x, q[j].y - (float)bh.y};
    const bf16x2 bm = {(__bf16)r1.x, (__bf16)r1.y};
    const bf16x2 bl = {(__bf16)(r1.x - (float)bm.x), (__bf16)(r1.y - (float)bm.y)};
    ph[j] = __builtin_bit_cast(s16x2, bh);      // (whole-vector casts: element-wise bit_cast of a bf16 vector's members miscompiled)
    pm[j] = __builtin_bit_cast(s16x2, bm);
    pl[j] = __builtin_bit_cast(s16x2, bl);
  }
  Parts r;
  r.h = s16x8{ph[0].x, ph[0].y, ph[1].x, ph[1].y, ph[2].x, ph[2].y, ph[3].x, ph[3].y};
  r.m = s16x8{pm[0].x, pm[0].y, pm[1].x, pm[1].y, pm[2].x, pm[2].y, pm[3].x, pm[3].y};
  r.l = s16x8{pl[0].x, pl[0].y, pl[1].x, pl[1].y, pl[2].x, pl[2].y, pl[3].x, pl[3].y};
  return r;
}
__global__ void k(const float* in, short* out) {
  f32x2 q[4];
  for (int j = 0; j < 4; ++j) q[j] = *reinterpret_cast<const f32x2*>(in + threadIdx.x * 8 + 2 * j);
  Parts p = split8(q);
  for (int e = 0; e < 8; ++e) { out[threadIdx.x * 24 + e] = p.h[e]; out[threadIdx.x * 24 + 8 + e] = p.m[e]; out[threadIdx.x * 24 + 16 + e] = p.l[e]; }
}
int main() {
  float h[64 * 8]; for (int i = 0; i < 512; ++i) h[i] = 0.1f * i + 0.001234f * (i % 7);
  float* d; short* o; hipMalloc(&d, sizeof h); hipMalloc(&o, 64 * 24 * 2);
  hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, o);
  short ho[64 * 24]; hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
  auto b2f = [](short s) { union { unsigned u; float f; } c; c.u = ((unsigned)(unsigned short)s) << 16; return c.f; };
  double worst = 0;
  for (int t = 0; t < 64; ++t) for (int e = 0; e < 8; ++e) {
    const float x = h[t * 8 + e], r = b2f(ho[t * 24 + e]) + b2f(ho[t * 24 + 8 + e]) + b2f(ho[t * 24 + 16 + e]);
    const double er = fabs((double)x - (double)r) / (fabs(x) + 1e-30); if (er > worst) worst = er;
  }
  printf("worst relative reconstruction error %.3e\n", worst);
  return 0;
}
