// kernels_debug.h -- fixed-ray-set kernels behind the parity entry points (rtx_trace_closest, rtx_trace_occluded) and the
// known-answer hooks of the generator (rtx_debug_pcg, rtx_debug_streams).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "queues.h"

namespace rtx {

// ----------------------------------------------------------------------------
// Fixed-ray-set kernels (parity entry points).

template <int BLOCK, int MODE>
__global__ __launch_bounds__(BLOCK) void k_trace_closest(SceneDev sc, const float* rays, uint32_t n,
                                                         uint32_t* kind, uint32_t* prim, float* t_out,
                                                         float* normal, TraceParams tp) {
  RTX_TRACE_PROLOGUE();
  uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  f3 o = mk3(rays[6 * (size_t)i], rays[6 * (size_t)i + 1], rays[6 * (size_t)i + 2]);
  f3 d = mk3(rays[6 * (size_t)i + 3], rays[6 * (size_t)i + 4], rays[6 * (size_t)i + 5]);
  float t;
  uint32_t id;
  cast_ray<MODE>(sc, geom, o, d, stk, t, id);
  f3 n3 = mk3(0, 0, 0);
  if (id == kMiss) {
    kind[i] = 0;
    prim[i] = kMiss;
  } else {
    f3 albedo;
    float emis;
    int surface;
    hit_surface(sc, geom, id, o, d, t, n3, albedo, emis, surface);
    if (id & kSphereBit) {
      uint32_t k = id & ~kSphereBit;
      if (k < sc.num_spheres) {
        kind[i] = 2;
      } else if (k < sc.num_spheres + sc.num_boxes) {
        kind[i] = 3;
        k -= sc.num_spheres;
      } else {
        kind[i] = 4;
        k -= sc.num_spheres + sc.num_boxes;
      }
      prim[i] = k;
    } else {
      kind[i] = 1;
      prim[i] = sc.slot_prim[id];
    }
  }
  t_out[i] = t;
  if (normal) {
    normal[3 * (size_t)i] = n3.x;
    normal[3 * (size_t)i + 1] = n3.y;
    normal[3 * (size_t)i + 2] = n3.z;
  }
}

template <int BLOCK, int MODE>
__global__ __launch_bounds__(BLOCK) void k_trace_occluded(SceneDev sc, const float* rays,
                                                          const float* dist, uint32_t n, uint8_t* out,
                                                          TraceParams tp) {
  RTX_TRACE_PROLOGUE();
  uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  f3 o = mk3(rays[6 * (size_t)i], rays[6 * (size_t)i + 1], rays[6 * (size_t)i + 2]);
  f3 d = mk3(rays[6 * (size_t)i + 3], rays[6 * (size_t)i + 4], rays[6 * (size_t)i + 5]);
  float distance = dist[i];
  bool occ = false;
  if (!(distance < 2.0f * sc.eps))  // Scene.h:192-194
    occ = occluded<MODE>(sc, geom, o, d, distance - sc.eps, stk);
  out[i] = occ ? 1 : 0;
}

// Device-side known-answer check of the generator (reference Sampler.h:24-41): thread 0 draws
// n values of rand_pcg() from Pcg(seed), thread 1 n values of Uniform() from another Pcg(seed),
// and thread 2 writes pcg_hash(seed + i).
__global__ void k_debug_pcg(uint32_t seed, uint32_t n, uint32_t* rand_out, float* uniform_out, uint32_t* hash_out) {
  const uint32_t which = blockIdx.x * blockDim.x + threadIdx.x;
  Rng state = {seed, kPcgIncrement};
  if (which == 0 && rand_out)
    for (uint32_t i = 0; i < n; i++) rand_out[i] = pcg_next(state);
  if (which == 1 && uniform_out)
    for (uint32_t i = 0; i < n; i++) uniform_out[i] = pcg_uniform(state);
  if (which == 2 && hash_out)
    for (uint32_t i = 0; i < n; i++) hash_out[i] = pcg_hash(seed + i);
}

// The generator every sample of a render call starts with, as primary_ray sets it: out[k] = state | increment << 32
// for the k-th sample of the call (sample rp.sample_base + k / band_pixels of pixel k mod band_pixels of the band).
__global__ void k_debug_streams(CameraDev cam, RenderParams rp, unsigned long long n, unsigned long long* out) {
  const unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const uint32_t in_band = (uint32_t)(k % rp.band_pixels);
  const uint32_t sample = rp.sample_base + (uint32_t)(k / rp.band_pixels);
  const uint32_t pixel = rp.row0 * cam.width + in_band;
  Rng g;
  g.s = pcg_hash(pixel ^ pcg_hash(sample ^ pcg_hash(rp.seed32)));
  g.inc = kPcgIncrement;
  if (rp.rng_streams) {
    const unsigned long long m = stream_mix(pixel, sample, rp.seed64);
    g.s = (uint32_t)m;
    g.inc = (uint32_t)(m >> 32) | 1u;
  }
  out[k] = (unsigned long long)g.s | ((unsigned long long)g.inc << 32);
}

}  // namespace rtx
