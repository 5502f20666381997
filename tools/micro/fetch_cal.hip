// FETCH_SIZE calibration: streams of KNOWN size read with a known access width, for rocprofv3 --pmc FETCH_SIZE.
// MI355X_MICROARCH.md (HBM section) corrects FETCH_SIZE by x2 for 16-byte-per-lane streaming loads on gfx950 and calls the
// 4-byte-per-lane width "uncalibrated"; the fused CR-CED kernel reads its input (67.6 MB at BASELINE config 3) with 4-byte
// lane loads, so its roofline.traffic needs this factor.  Three kernels, each reading exactly 258 * 512 * 129 * 4 bytes
// = 68,161,536 B... (67,633,152 B = 256 x 512 x 129 floats) once: read4 (one dword per lane, coalesced), read8, read16.
//   hipcc --offload-arch=gfx950 -O3 -o fetch_cal tools/micro/fetch_cal.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- ./fetch_cal
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <class T>
__global__ void read_stream(const T* __restrict__ x, size_t n, float* __restrict__ out) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const T v = x[i];
    if constexpr (sizeof(T) == 4) acc += v;
    else if constexpr (sizeof(T) == 8) acc += v.x + v.y;
    else acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 123.456f) out[blockIdx.x] = acc;      // never: keeps the loads
}
int main() {
  const size_t floats = (size_t)256 * 512 * 129;   // 67,633,152 B: the CR-CED input at BASELINE config 3
  float *x, *out, *flush;
  hipMalloc(&x, floats * 4);
  hipMalloc(&out, 4096 * 4);
  hipMalloc(&flush, (size_t)1 << 30);               // 1 GiB written between the reads: nothing of x stays in L2 / MALL
  hipMemset(x, 0, floats * 4);
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(flush, rep, (size_t)1 << 30);
    hipLaunchKernelGGL(read_stream<float>, dim3(2048), dim3(256), 0, 0, (const float*)x, floats, out);
    hipMemset(flush, rep + 8, (size_t)1 << 30);
    hipLaunchKernelGGL(read_stream<f32x2>, dim3(2048), dim3(256), 0, 0, (const f32x2*)x, floats / 2, out);
    hipMemset(flush, rep + 16, (size_t)1 << 30);
    hipLaunchKernelGGL(read_stream<f32x4>, dim3(2048), dim3(256), 0, 0, (const f32x4*)x, floats / 4, out);
  }
  hipDeviceSynchronize();
  printf("bytes per kernel: %zu\n", floats * 4);
  return 0;
}
