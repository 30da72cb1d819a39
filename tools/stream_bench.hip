// Micro-benchmark: what HBM rate does a kernel with k_shade's access shape reach when it
// does nothing else?  Per element: reads of 16+8+8+8 (+16+16) bytes from separate arrays,
// writes of 16+8+16+16+8 bytes to separate arrays for a fraction of the elements
// (compacted by wave, appended with one atomic per wave) and 12 bytes to a result array
// for the others.  Compare with a plain float4 copy.
// Build: hipcc --offload-arch=gfx950 -O3 -o stream_bench stream_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void copy4(const float4* a, float4* b, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) b[i] = a[i];
}

struct Q { float4* r0; float2* r1; float4* s0; float4* s1; uint2* s2; };

// MODE bits: 1 compact by wave, 2 read all state, 4 no result writes, 8 result planes, 16 compact by block,
// 32 every lane writes a result (no holes in the result lines)
template <int MODE>
__global__ __launch_bounds__(256) void shade_like(Q in, Q out, const uint2* hits, float* res, unsigned long long* counters,
                                                  uint32_t cap, uint32_t keep_mod, size_t n_total) {
  constexpr bool COMPACT = MODE & 1, READ_STATE = MODE & 2, NO_RES = MODE & 4, PLANES = MODE & 8, BLOCKC = MODE & 16, ALLRES = MODE & 32;
  __shared__ uint32_t wave_cnt[4];
  __shared__ uint32_t block_base;
  const uint32_t s = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  const uint32_t i = s * cap + j;
  float4 r0 = in.r0[i];
  float2 r1 = in.r1[i];
  uint2 s2 = in.s2[i];
  uint2 h = hits[i];
  float4 s0 = make_float4(1, 1, 1, -1), s1 = make_float4(0, 0, 0, 0);
  if (READ_STATE || (s2.y & 1)) { s0 = in.s0[i]; s1 = in.s1[i]; }
  const bool keep = ((s2.y >> 3) % 100) < keep_mod;
  r0.x += h.x * 1e-30f; s1.x += r1.x * s0.x;
  uint32_t pos = j;
  if (BLOCKC) {
    const unsigned long long m = __ballot(keep);
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) wave_cnt[w] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0)
      block_base = (uint32_t)atomicAdd(&counters[16 * s], (unsigned long long)(wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3]));
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t k = 0; k < w; k++) before += wave_cnt[k];
    pos = block_base + before + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
  } else if (COMPACT) {
    const unsigned long long m = __ballot(keep);
    uint32_t base = 0;
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if (lane == 0) base = (uint32_t)atomicAdd(&counters[16 * s], (unsigned long long)__popcll(m));
    base = __builtin_amdgcn_readfirstlane(base);
    pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
  }
  if (keep) {
    const uint32_t o = s * cap + pos;
    out.r0[o] = r0; out.r1[o] = r1; out.s0[o] = s0; out.s1[o] = s1; out.s2[o] = s2;
  }
  if ((ALLRES || !keep) && !NO_RES) {
    if (PLANES) {
      res[s2.x] = s1.x; res[n_total + s2.x] = s1.y; res[2 * n_total + s2.x] = s1.z;
    } else {
      float* d = res + 3 * (size_t)s2.x;
      d[0] = s1.x; d[1] = s1.y; d[2] = s1.z;
    }
  }
}

// The same job with the path state in two arrays of 32-byte records instead of five arrays:
// R = (o, d, ordinal, rng), S = (attenuation, mis, colour, nee).
struct Q2 { float4* R; float4* S; };
template <bool BLOCKC, bool NO_RES>
__global__ __launch_bounds__(256) void shade_like_rec(Q2 in, Q2 out, const uint2* hits, float* res,
                                                      unsigned long long* counters, uint32_t cap, uint32_t keep_mod) {
  __shared__ uint32_t wave_cnt[4];
  __shared__ uint32_t block_base;
  const uint32_t s = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  const uint32_t i = s * cap + j;
  float4 ra = in.R[2 * (size_t)i], rb = in.R[2 * (size_t)i + 1];
  uint2 h = hits[i];
  float4 sa = in.S[2 * (size_t)i], sb = in.S[2 * (size_t)i + 1];
  const uint32_t rngbits = __float_as_uint(rb.w), ordinal = __float_as_uint(rb.z);
  const bool keep = ((rngbits >> 3) % 100) < keep_mod;
  ra.x += h.x * 1e-30f; sb.x += rb.x * sa.x;
  const unsigned long long m = __ballot(keep);
  uint32_t pos;
  if (BLOCKC) {
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) wave_cnt[w] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0)
      block_base = (uint32_t)atomicAdd(&counters[16 * s], (unsigned long long)(wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3]));
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t k = 0; k < w; k++) before += wave_cnt[k];
    pos = block_base + before + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
  } else {
    uint32_t base = 0;
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if (lane == 0) base = (uint32_t)atomicAdd(&counters[16 * s], (unsigned long long)__popcll(m));
    base = __builtin_amdgcn_readfirstlane(base);
    pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
  }
  if (keep) {
    const size_t o = (size_t)s * cap + pos;
    out.R[2 * o] = ra; out.R[2 * o + 1] = rb; out.S[2 * o] = sa; out.S[2 * o + 1] = sb;
  } else if (!NO_RES) {
    float* d = res + 3 * (size_t)ordinal;
    d[0] = sb.x; d[1] = sb.y; d[2] = sb.z;
  }
}

int main() {
  const uint32_t nseg = 512, cap = 32768;
  const size_t n = (size_t)nseg * cap;  // 16 Mi
  Q q[2];
  for (int k = 0; k < 2; k++) {
    CHECK(hipMalloc(&q[k].r0, n * 16)); CHECK(hipMalloc(&q[k].r1, n * 8)); CHECK(hipMalloc(&q[k].s0, n * 16));
    CHECK(hipMalloc(&q[k].s1, n * 16)); CHECK(hipMalloc(&q[k].s2, n * 8));
    CHECK(hipMemset(q[k].r0, 0, n * 16)); CHECK(hipMemset(q[k].s0, 0, n * 16)); CHECK(hipMemset(q[k].s1, 0, n * 16));
    CHECK(hipMemset(q[k].r1, 0, n * 8));
  }
  uint2* hits; float* res; unsigned long long* ctr;
  CHECK(hipMalloc(&hits, n * 8)); CHECK(hipMemset(hits, 0, n * 8));
  CHECK(hipMalloc(&res, n * 12 * 4)); CHECK(hipMalloc(&ctr, nseg * 128));
  // s2: x = ordinal (consecutive), y = pseudo-random bits
  uint2* h2 = (uint2*)malloc(n * 8);
  uint32_t st = 1;
  for (size_t i = 0; i < n; i++) { st = st * 747796405u + 2891336453u; h2[i] = make_uint2((uint32_t)i, st >> 7); }
  CHECK(hipMemcpy(q[0].s2, h2, n * 8, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, double bytes, auto launch) {
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
      CHECK(hipMemset(ctr, 0, nseg * 128));
      CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
    }
    printf("%-46s %7.3f ms  %6.2f TB/s\n", name, best, bytes / best / 1e9);
  };
  timeit("float4 copy, 16 Mi elements x 3 arrays' worth", 2.0 * n * 16, [&] { copy4<<<n / 256, 256>>>(q[0].r0, q[1].r0, n); });
  dim3 grid(cap / 256, nseg);
  struct V { int mode; uint32_t keep; const char* name; };
  const V vs[] = {
      {0, 42, "in place (holes), results AoS"},
      {1, 42, "wave-compacted, results AoS"},
      {1 | 2, 42, "wave-compacted, all state read, results AoS"},
      {1 | 4, 42, "wave-compacted, no result writes"},
      {1 | 8, 42, "wave-compacted, result planes"},
      {16, 42, "block-compacted, results AoS"},
      {16 | 8, 42, "block-compacted, result planes"},
      {1 | 32, 42, "wave-compacted, every lane writes a result"},
      {1 | 4, 100, "keep all, wave atomic, no results"},
      {4, 100, "keep all, in place, no results"},
      {1, 78, "wave-compacted keep 78%, results AoS"},
      {16 | 8, 78, "block-compacted keep 78%, result planes"},
  };
  for (const V& v : vs) {
    const double f = v.keep / 100.0;
    const double rd = (v.mode & 2) ? 72 : 40 + 32 * 0.5;
    const double wr = f * 64 + ((v.mode & 4) ? 0 : (v.mode & 32) ? 12 : (1 - f) * 12);
    auto go = [&](auto tag) {
      timeit(v.name, n * (rd + wr), [&] { shade_like<decltype(tag)::value><<<grid, 256>>>(q[0], q[1], hits, res, ctr, cap, v.keep, n); });
    };
    switch (v.mode) {
      case 0: go(std::integral_constant<int, 0>{}); break;
      case 1: go(std::integral_constant<int, 1>{}); break;
      case 3: go(std::integral_constant<int, 3>{}); break;
      case 5: go(std::integral_constant<int, 5>{}); break;
      case 9: go(std::integral_constant<int, 9>{}); break;
      case 16: go(std::integral_constant<int, 16>{}); break;
      case 24: go(std::integral_constant<int, 24>{}); break;
      case 4: go(std::integral_constant<int, 4>{}); break;
      case 33: go(std::integral_constant<int, 33>{}); break;
    }
  }
  {
    Q2 a, b;
    CHECK(hipMalloc(&a.R, n * 32)); CHECK(hipMalloc(&a.S, n * 32)); CHECK(hipMalloc(&b.R, n * 32)); CHECK(hipMalloc(&b.S, n * 32));
    CHECK(hipMemset(a.S, 0, n * 32));
    float4* hr = (float4*)malloc(n * 32);
    for (size_t i = 0; i < n; i++) {
      hr[2 * i] = make_float4(0, 0, 0, 0);
      float fx, fy; memcpy(&fx, &h2[i].x, 4); memcpy(&fy, &h2[i].y, 4);
      hr[2 * i + 1] = make_float4(0, 0, fx, fy);
    }
    CHECK(hipMemcpy(a.R, hr, n * 32, hipMemcpyHostToDevice));
    for (uint32_t keep : {42u, 78u, 100u}) {
      const double f = keep / 100.0;
      char name[128];
      snprintf(name, 128, "32-byte records, wave-compacted keep %u%%, results", keep);
      timeit(name, n * (72 + f * 64 + (1 - f) * 12), [&] { shade_like_rec<false, false><<<grid, 256>>>(a, b, hits, res, ctr, cap, keep); });
      snprintf(name, 128, "32-byte records, block-compacted keep %u%%, results", keep);
      timeit(name, n * (72 + f * 64 + (1 - f) * 12), [&] { shade_like_rec<true, false><<<grid, 256>>>(a, b, hits, res, ctr, cap, keep); });
      snprintf(name, 128, "32-byte records, wave-compacted keep %u%%, no results", keep);
      timeit(name, n * (72 + f * 64), [&] { shade_like_rec<false, true><<<grid, 256>>>(a, b, hits, res, ctr, cap, keep); });
    }
  }
  return 0;
}
