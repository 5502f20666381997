// Is  r = x - bf16(x)  computed as ONE v_dot2_f32_bf16 (h_pack . (-1, 0) + x) bit-identical to the shift + v_sub_f32 form?
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o dot2_split_test dot2_split_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <random>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, unsigned* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  const float x0 = x[2 * i], x1 = x[2 * i + 1];
  const bf16x2 bh = {(__bf16)x0, (__bf16)x1};
  unsigned h = __builtin_bit_cast(unsigned, bh);
  asm volatile("" : "+v"(h));
  const float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
  float d0, d1;
  asm volatile("v_dot2_f32_bf16 %0, %2, %3, %4\n\tv_dot2_f32_bf16 %1, %2, %5, %6" : "=&v"(d0), "=&v"(d1) : "v"(h), "v"(0x0000BF80u), "v"(x0), "v"(0xBF800000u), "v"(x1));
  out[4 * i] = __builtin_bit_cast(unsigned, r0);
  out[4 * i + 1] = __builtin_bit_cast(unsigned, d0);
  out[4 * i + 2] = __builtin_bit_cast(unsigned, r1);
  out[4 * i + 3] = __builtin_bit_cast(unsigned, d1);
}
int main() {
  const int n = 1 << 22;
  std::vector<float> x(n);
  std::mt19937 g(1);
  for (int i = 0; i < n; ++i) {
    unsigned u = g();
    if (i % 4 == 0) u &= 0x7fffffffu;                      // positive, any exponent (denormals, Inf, NaN included)
    else if (i % 4 == 1) u = (u & 0x007fffffu) | ((100 + g() % 60) << 23);   // ordinary magnitudes
    else if (i % 4 == 2) u = (u & 0x807fffffu) | ((120 + g() % 16) << 23);
    float f; memcpy(&f, &u, 4);
    x[i] = f;
  }
  x[0] = 0.f; x[1] = 1e30f; x[2] = 3.0e38f; x[3] = 1e-38f; x[4] = 1e-40f; x[5] = 1.f;
  float* dx; unsigned* dout;
  hipMalloc(&dx, n * 4); hipMalloc(&dout, n * 8);
  hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
  k<<<n / 2 / 256, 256>>>(dx, dout, n);
  std::vector<unsigned> o(2 * n);
  hipMemcpy(o.data(), dout, n * 8, hipMemcpyDeviceToHost);
  long bad = 0, bad_normal = 0, shown = 0;
  for (int i = 0; i < n; ++i) {
    const unsigned a = o[2 * i], b = o[2 * i + 1];
    unsigned u; memcpy(&u, &x[i], 4);
    const unsigned ex = (u >> 23) & 0xff;
    const bool nan_both = ((a & 0x7fffffffu) > 0x7f800000u) && ((b & 0x7fffffffu) > 0x7f800000u);
    if (a != b && !nan_both) {
      ++bad;
      unsigned un; memcpy(&un, &x[i ^ 1], 4);
      const unsigned exn = (un >> 23) & 0xff;
      if (ex > 20 && ex < 255 && exn < 255) { ++bad_normal; if (shown++ < 10) printf("x=%08x neighbour=%08x sub=%08x dot2=%08x\n", u, un, a, b); }
    }
  }
  printf("values %d, mismatches %ld, of which at ordinary magnitudes (exponent > 20, finite pair) %ld\n", n, bad, bad_normal);
  return 0;
}
