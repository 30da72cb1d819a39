// Micro-benchmark: 64-byte records fetched at random by every lane, the access shape of
// a BVH walk.  (a) each lane reads its own record with four 16-byte loads (64 distinct
// lines per load instruction); (b) the four lanes of a quad read each other's records
// together, 16 contiguous bytes per lane (16 distinct lines per load instruction), with
// and without handing the parts back to their owners.
// Build: hipcc --offload-arch=gfx950 -O3 -o gather_bench gather_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t next_index(uint32_t& s, uint32_t mask) {
  s = s * 747796405u + 2891336453u;
  uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
  return ((w >> 22u) ^ w) & mask;
}

__global__ __launch_bounds__(128) void per_lane(const float4* tab, uint32_t mask, int steps, float* out) {
  uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 1u;
  float acc = 0.f;
  uint32_t idx = next_index(s, mask);
  for (int i = 0; i < steps; i++) {
    const float4* r = tab + 4 * (size_t)idx;
    float4 a = r[0], b = r[1], c = r[2], d = r[3];
    acc += a.x + b.y + c.z + d.w;
    // the next index depends on the data, as a tree walk's does
    idx = (next_index(s, mask) + (uint32_t)(acc != 12345.f ? 0 : 1)) & mask;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// (c) every second fetch takes the other 64-byte half of the 128-byte line the previous one took:
// what a walk sees when sibling records are laid out in pairs.
__global__ __launch_bounds__(128) void per_lane_pairs(const float4* tab, uint32_t mask, int steps, float* out) {
  uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 1u;
  float acc = 0.f;
  uint32_t idx = next_index(s, mask);
  for (int i = 0; i < steps; i++) {
    const float4* r = tab + 4 * (size_t)idx;
    float4 a = r[0], b = r[1], c = r[2], d = r[3];
    acc += a.x + b.y + c.z + d.w;
    const uint32_t bump = (uint32_t)(acc != 12345.f ? 0 : 1);
    idx = (i & 1) ? (next_index(s, mask) + bump) & mask : (idx ^ 1u) + bump;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <bool HAND_BACK>
__global__ __launch_bounds__(128) void per_quad(const float4* tab, uint32_t mask, int steps, float* out) {
  uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 1u;
  const int lane = threadIdx.x & 63, part = lane & 3, quad0 = lane & ~3;
  float acc = 0.f;
  uint32_t idx = next_index(s, mask);
  for (int i = 0; i < steps; i++) {
    float4 v[4];
#pragma unroll
    for (int m = 0; m < 4; m++) {
      uint32_t im = (uint32_t)__shfl((int)idx, quad0 + m);
      v[m] = tab[4 * (size_t)im + part];
    }
    if (HAND_BACK) {
      // lane (quad0 + j) wants part p of its record: held by lane quad0 + p in v[j]
      float4 q[4];
#pragma unroll
      for (int p = 0; p < 4; p++) {
        float4 pick = part == 0 ? v[0] : part == 1 ? v[1] : part == 2 ? v[2] : v[3];  // placeholder select
        (void)pick;
        float4 r;
        // every lane reads from lane quad0+p the register indexed by its own part
        float x0 = __shfl(v[0].x, quad0 + p), x1 = __shfl(v[1].x, quad0 + p), x2 = __shfl(v[2].x, quad0 + p), x3 = __shfl(v[3].x, quad0 + p);
        float y0 = __shfl(v[0].y, quad0 + p), y1 = __shfl(v[1].y, quad0 + p), y2 = __shfl(v[2].y, quad0 + p), y3 = __shfl(v[3].y, quad0 + p);
        float z0 = __shfl(v[0].z, quad0 + p), z1 = __shfl(v[1].z, quad0 + p), z2 = __shfl(v[2].z, quad0 + p), z3 = __shfl(v[3].z, quad0 + p);
        float w0 = __shfl(v[0].w, quad0 + p), w1 = __shfl(v[1].w, quad0 + p), w2 = __shfl(v[2].w, quad0 + p), w3 = __shfl(v[3].w, quad0 + p);
        r.x = part == 0 ? x0 : part == 1 ? x1 : part == 2 ? x2 : x3;
        r.y = part == 0 ? y0 : part == 1 ? y1 : part == 2 ? y2 : y3;
        r.z = part == 0 ? z0 : part == 1 ? z1 : part == 2 ? z2 : z3;
        r.w = part == 0 ? w0 : part == 1 ? w1 : part == 2 ? w2 : w3;
        q[p] = r;
      }
      acc += q[0].x + q[1].y + q[2].z + q[3].w;
    } else {
      acc += v[0].x + v[1].y + v[2].z + v[3].w;
    }
    idx = (next_index(s, mask) + (uint32_t)(acc != 12345.f ? 0 : 1)) & mask;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
  const int steps = 64;
  const int threads = 1 << 22;
  float* out;
  CHECK(hipMalloc(&out, threads * sizeof(float)));
  for (uint32_t log_records : {14u, 18u, 20u, 22u}) {  // 1 MB, 16 MB, 64 MB, 256 MB of records
    const size_t n = (size_t)1 << log_records;
    float4* tab;
    CHECK(hipMalloc(&tab, n * 64));
    CHECK(hipMemset(tab, 0, n * 64));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int variant = 0; variant < 4; variant++) {
      float best = 1e30f;
      for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0));
        if (variant == 0) per_lane<<<threads / 128, 128>>>(tab, (uint32_t)n - 1, steps, out);
        if (variant == 1) per_quad<false><<<threads / 128, 128>>>(tab, (uint32_t)n - 1, steps, out);
        if (variant == 2) per_quad<true><<<threads / 128, 128>>>(tab, (uint32_t)n - 1, steps, out);
        if (variant == 3) per_lane_pairs<<<threads / 128, 128>>>(tab, (uint32_t)n - 1, steps, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
      }
      const double fetches = (double)threads * steps;
      printf("table %4zu MB  %-22s %8.3f ms  %7.2f G records/s  %7.2f TB/s\n", n * 64 >> 20,
             variant == 0 ? "per-lane 4x16B" : variant == 1 ? "per-quad (no hand-back)" : variant == 2 ? "per-quad + hand-back" : "per-lane, line pairs",
             best, fetches / best / 1e6, fetches * 64 / best / 1e9);
    }
    CHECK(hipFree(tab));
  }
  return 0;
}
