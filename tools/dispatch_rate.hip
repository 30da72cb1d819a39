// How fast the chip starts waves: launches of 16.8 M threads whose waves leave at once, after one dependent
// global load, or after spinning for a given number of clock ticks, in blocks of 64 to 512 threads.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/dispatch_rate tools/dispatch_rate.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_empty(const unsigned* p, unsigned* out) {}
__global__ void k_one_load(const unsigned* p, unsigned* out) {
  if (p[blockIdx.x & 1023u] == 0xdeadbeefu) out[0] = 1;
}
__global__ void k_spin(const unsigned* p, unsigned* out, unsigned ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < ticks) {}
  if (p[blockIdx.x & 1023u] == 0xdeadbeefu) out[0] = 1;
}

int main() {
  unsigned *p, *out;
  hipMalloc(&p, 4096);
  hipMalloc(&out, 4);
  hipMemset(p, 0, 4096);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const unsigned threads = 1u << 24;
  for (unsigned B : {64u, 128u, 256u, 512u}) {
    for (int kind = 0; kind < 5; kind++) {
      const unsigned ticks = kind == 2 ? 2000u : kind == 3 ? 8000u : 18000u;
      float best = 1e9f;
      for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(a);
        if (kind == 0) k_empty<<<threads / B, B>>>(p, out);
        else if (kind == 1) k_one_load<<<threads / B, B>>>(p, out);
        else k_spin<<<threads / B, B>>>(p, out, ticks);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
      }
      const double waves = threads / 64.0;
      const char* names[] = {"empty", "one load", "spin 2000", "spin 8000", "spin 18000"};
      // ideal for a spin of T ticks: waves / (1024 SIMDs x 8 slots) x T / 2.4 GHz (if the tick is a shader cycle)
      printf("block %3u %-10s %8.3f ms  %7.2f G waves/s", B, names[kind], best, waves / best / 1e6);
      if (kind >= 2) printf("  (slots always full: %.3f ms)", waves / 8192.0 * ticks / 2.4e6);
      printf("\n");
    }
  }
  return 0;
}
