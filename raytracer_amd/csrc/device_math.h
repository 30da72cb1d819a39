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

// Pcg, reference Sampler.h:24-41.
__device__ __forceinline__ uint32_t pcg_next(uint32_t& s) {
  uint32_t old = s;
  s = s * 747796405u + 2891336453u;
  uint32_t w = ((old >> ((old >> 28u) + 4u)) ^ old) * 277803737u;
  return (w >> 22u) ^ w;
}
__device__ __forceinline__ float pcg_uniform(uint32_t& s) {
  float u = (float)pcg_next(s) * 0x1p-32f;
  return std_min(0x1.fffffep-1f, u);
}
__host__ __device__ __forceinline__ uint32_t pcg_hash(uint32_t in) {
  uint32_t st = in * 747796405u + 2891336453u;
  uint32_t w = ((st >> ((st >> 28u) + 4u)) ^ st) * 277803737u;
  return (w >> 22u) ^ w;
}

}  // namespace rtx
