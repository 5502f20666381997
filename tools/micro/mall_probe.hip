// Does a consumer kernel that walks a tensor in the REVERSE of its producer's order find the tensor's tail in the memory-side
// cache (MI355X: 256 MB)?  Producer writes N bytes front to back; consumer reads them front to back or back to front.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mall_probe tools/micro/mall_probe.hip && /tmp/mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void produce(f4* __restrict__ out, size_t n4, float v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) out[i] = f4{v, v, v, v};
}
// chunked walk: block b handles chunks b, b + grid, ... (or mirrored), so that "order" is the same notion as a persistent tile loop
__global__ void consume(const f4* __restrict__ in, size_t n4, float* __restrict__ sink, int reverse, int nt) {
  const size_t chunk = 4096;   // float4 per chunk = 64 KB
  const size_t nchunks = n4 / chunk;
  f4 acc = {0, 0, 0, 0};
  for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const size_t cc = reverse ? nchunks - 1 - c : c;
    for (size_t i = threadIdx.x; i < chunk; i += blockDim.x) {
      const f4 v = nt ? __builtin_nontemporal_load(in + cc * chunk + i) : in[cc * chunk + i];
      acc += v;
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.f) *sink = 1.f;
}
int main() {
  const size_t sizes[] = {128ull << 20, 512ull << 20, 2048ull << 20};
  float* sink; hipMalloc(&sink, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (size_t bytes : sizes) {
    f4* buf; hipMalloc(&buf, bytes);
    const size_t n4 = bytes / 16;
    for (int nt = 0; nt < 2; ++nt)
      for (int rev = 0; rev < 2; ++rev) {
        float best = 1e9f, bestp = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
          hipEventRecord(e0);
          produce<<<2048, 256>>>(buf, n4, (float)rep);
          hipEventRecord(e1); hipEventSynchronize(e1);
          float msp; hipEventElapsedTime(&msp, e0, e1);
          hipEventRecord(e0);
          consume<<<1024, 256>>>(buf, n4, sink, rev, nt);
          hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          if (ms < best) best = ms;
          if (msp < bestp) bestp = msp;
        }
        printf("%5zu MB  nt=%d  %s: write %.3f ms (%.2f TB/s)  read %.3f ms (%.2f TB/s)\n", bytes >> 20, nt, rev ? "reverse" : "forward",
               bestp, bytes / bestp / 1e9, best, bytes / best / 1e9);
      }
    hipFree(buf);
  }
  return 0;
}
