// Micro-benchmark: cost of the atomics a wavefront path tracer needs on gfx950.
//   A: one returning atomicAdd per wave on ONE counter (queue append)
//   B: the same on S counters in separate 128-byte lines (segmented queue)
//   C: three float atomicAdd per lane at pixel*3+c (interleaved RGB sums)
//   D: plain 12-byte stores per lane to consecutive slots (per-sample result slots)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

__global__ void kA(unsigned* ctr, unsigned* out, int shards) {
  unsigned lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  unsigned base = 0;
  if (lane == 0) base = atomicAdd(&ctr[(wave % shards) * 32], 64u);
  base = __builtin_amdgcn_readfirstlane(base);
  out[blockIdx.x * blockDim.x + threadIdx.x] = base + lane;
}
__global__ void kC(float* acc, const unsigned* pix, unsigned n) {
  unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned p = pix[i];
  atomicAdd(&acc[3 * p], 1.0f); atomicAdd(&acc[3 * p + 1], 2.0f); atomicAdd(&acc[3 * p + 2], 3.0f);
}
__global__ void kCplanar(float* acc, const unsigned* pix, unsigned n, unsigned npix) {
  unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned p = pix[i];
  atomicAdd(&acc[p], 1.0f); atomicAdd(&acc[npix + p], 2.0f); atomicAdd(&acc[2 * npix + p], 3.0f);
}
struct __attribute__((packed)) f3p { float x, y, z; };
__global__ void kD(float* res, const unsigned* slot, unsigned n) {
  unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned s = slot[i];
  res[3 * (size_t)s] = 1.0f; res[3 * (size_t)s + 1] = 2.0f; res[3 * (size_t)s + 2] = 3.0f;
}
int main() {
  const unsigned n = 1u << 22, npix = 1u << 20;
  unsigned *ctr, *out, *pix; float *acc, *res;
  CK(hipMalloc(&ctr, 4096 * 128)); CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&pix, n * 4));
  CK(hipMalloc(&acc, npix * 3 * 4)); CK(hipMalloc(&res, (size_t)n * 12));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto f) {
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < 10; r++) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); printf("%-44s %8.1f us per launch\n", name, ms * 100);
  };
  for (int shards : {1, 8, 64, 512}) {
    char nm[64]; snprintf(nm, 64, "A/B wave append, %d counters, 65536 waves", shards);
    timeit(nm, [&] { hipMemsetAsync(ctr, 0, 4096 * 128); kA<<<n / 256, 256>>>(ctr, out, shards); });
  }
  std::vector<unsigned> h(n);
  for (int mode = 0; mode < 3; mode++) {
    // 0: consecutive pixels; 1: every 4th lane active pattern (strided pixels); 2: random pixels
    for (unsigned i = 0; i < n; i++) h[i] = mode == 0 ? i % npix : mode == 1 ? (i * 7u) % npix : (i * 2654435761u) % npix;
    CK(hipMemcpy(pix, h.data(), n * 4, hipMemcpyHostToDevice));
    const char* mn[] = {"consecutive", "stride-7", "random"};
    char nm[64];
    snprintf(nm, 64, "C float atomics RGB interleaved, %s", mn[mode]);
    timeit(nm, [&] { kC<<<n / 256, 256>>>(acc, pix, n); });
    snprintf(nm, 64, "C float atomics planar, %s", mn[mode]);
    timeit(nm, [&] { kCplanar<<<n / 256, 256>>>(acc, pix, n, npix); });
    for (unsigned i = 0; i < n; i++) h[i] = mode == 0 ? i : mode == 1 ? (unsigned)(((unsigned long long)i * 7u) % n) : (unsigned)(((unsigned long long)i * 2654435761u) % n);
    CK(hipMemcpy(pix, h.data(), n * 4, hipMemcpyHostToDevice));
    snprintf(nm, 64, "D plain 12-B stores, %s", mn[mode]);
    timeit(nm, [&] { kD<<<n / 256, 256>>>(res, pix, n); });
  }
  return 0;
}
