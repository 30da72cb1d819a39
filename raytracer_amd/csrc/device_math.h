// device_math.h -- float3 arithmetic for the HIP kernels with the reference's
// Vec3 conventions (reference Vec3.cpp:15-114).
//
// The translation unit is compiled with -ffp-contract=off, so each operator
// below is exactly one IEEE-754 binary32 operation, in the order written:
// the same arithmetic an x86-64 SSE build of the reference performs.
//   a - b   is  a + (-b)                       (Vec3.cpp:87-89)
//   v / f   is  v * (1.0f / f)                 (Vec3.cpp:78-80)
//   normalized() divides each component by the norm (Vec3.cpp:27-30)
//   dot is (x*x' + y*y') + z*z'                (Vec3.cpp:15-17)
#pragma once
#include <hip/hip_runtime.h>

namespace rtx {

struct f3 {
  float x, y, z;
};

__device__ __forceinline__ f3 mk3(float x, float y, float z) {
  f3 r;
  r.x = x;
  r.y = y;
  r.z = z;
  return r;
}
__device__ __forceinline__ f3 splat3(float s) { return mk3(s, s, s); }
__device__ __forceinline__ f3 vadd(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 vneg(f3 a) { return mk3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ f3 vsub(f3 a, f3 b) { return vadd(a, vneg(b)); }
__device__ __forceinline__ f3 vscale(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 vmul(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ f3 vdivs(f3 a, float s) { return vscale(a, 1.0f / s); }
__device__ __forceinline__ float vdot(f3 a, f3 o) { return a.x * o.x + a.y * o.y + a.z * o.z; }
__device__ __forceinline__ f3 vcross(f3 a, f3 o) {
  return mk3(a.y * o.z - a.z * o.y, a.z * o.x - a.x * o.z, a.x * o.y - a.y * o.x);
}
__device__ __forceinline__ float vsqrnorm(f3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
__device__ __forceinline__ float vnorm(f3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
__device__ __forceinline__ f3 vnormalized(f3 a) {
  float n = vnorm(a);
  return mk3(a.x / n, a.y / n, a.z / n);
}
// std::max(x, std::max(y, z)) with std::max(a,b) = (a < b) ? b : a  (Vec3.cpp:35-37)
__device__ __forceinline__ float std_max(float a, float b) { return (a < b) ? b : a; }
__device__ __forceinline__ float std_min(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float vmax3(f3 a) { return std_max(a.x, std_max(a.y, a.z)); }

// Pcg, reference Sampler.h:24-41: state and the increment of its step.  The reference's increment is the constant
// kPcgIncrement for every generator; rtx_config.rng_streams gives every sample an odd increment of its own, which
// puts it on a cycle of its own (stream_start).
constexpr uint32_t kPcgIncrement = 2891336453u;
struct Rng {
  uint32_t s, inc;
};
__device__ __forceinline__ uint32_t pcg_next(Rng& g) {
  uint32_t old = g.s;
  g.s = g.s * 747796405u + g.inc;
  uint32_t w = ((old >> ((old >> 28u) + 4u)) ^ old) * 277803737u;
  return (w >> 22u) ^ w;
}
__device__ __forceinline__ float pcg_uniform(Rng& g) {
  float u = (float)pcg_next(g) * 0x1p-32f;
  return std_min(0x1.fffffep-1f, u);
}
// rtx_config.rng_streams: the generator of sample `sample` of pixel `pixel`.  The 64-bit key (sample, pixel) goes
// through a bijection of the 64-bit integers (xor with the seed, then the finaliser of MurmurHash3: xor-shifts and
// odd multipliers); the low word is the state, the high word with its lowest bit set the increment.  Two samples
// get the same (state, increment) only if their images differ in bit 32 alone: with n samples about n^2 / 2^65
// such pairs are expected, 31 for the 3.4e10 samples of a 3840x2160 frame at 4096 per pixel.
__host__ __device__ __forceinline__ unsigned long long stream_mix(uint32_t pixel, uint32_t sample, unsigned long long seed) {
  unsigned long long k = (((unsigned long long)sample << 32) | pixel) ^ seed;
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return k;
}
__host__ __device__ __forceinline__ uint32_t pcg_hash(uint32_t in) {
  uint32_t st = in * 747796405u + 2891336453u;
  uint32_t w = ((st >> ((st >> 28u) + 4u)) ^ st) * 277803737u;
  return (w >> 22u) ^ w;
}

}  // namespace rtx
