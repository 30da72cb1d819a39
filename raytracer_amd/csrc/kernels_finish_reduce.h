// kernels_finish_reduce.h -- the end of a frame: k_finish (the last paths run to their ends, one per lane), the ordered
// per-pixel sum of the per-sample results (k_reduce; k_mark_unfinished, k_reduce_head, k_reduce_tail where it runs
// beside k_finish) and k_resolve.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "queues.h"

namespace rtx {

// Runs every path still in queue `qi` to its end, one path per lane: used once
// the samples are exhausted and so few paths remain that a round of small
// kernels is all launch latency.  Every lane leaves after at most kFinishCap
// bounces (a path that long has probability below 0.99^65536).
constexpr uint32_t kFinishCap = 65536;
template <int BLOCK, int MODE, int PLAIN = 0>
__global__ __launch_bounds__(BLOCK) void k_finish(SceneDev sc, Queue q, const SegCounter* seg, int qi,
                                                  uint32_t cap, float* res, Summary* sum,
                                                  TraceParams tp, const Primary* __restrict__ prp) {
  if constexpr (PLAIN >= 1) {  // (as in k_trace_shade: what the host has seen the scene and the call not to be)
    sc.num_boxes = 0;
    sc.num_planes = 0;
    sc.integrator = 0;
    sc.max_bounces = 0;
    sc.rng_streams = 0;
  }
  if constexpr (PLAIN == 2) {
    sc.sky = 1;
    sc.carries_colour = 0;
    sc.few_surfaces = 1;
  }
  RTX_TRACE_PROLOGUE();
  const uint32_t s = blockIdx.y;
  const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
  const bool active = j < seg_paths(seg[s].c[qi]);
  uint32_t nb = 0, nn = 0, ns = 0;
  bool bad = false, truncated = false;
  if (active) {
    PathState p;
    load_path(sc, q, s * cap + j, j >= seg[s].gen_slot0, *prp,
              (uint32_t)seg[s].gen_ordinal0 + (j - seg[s].gen_slot0), p);
    LastSphere last;
    for (uint32_t it = 0; it < kFinishCap; it++) {
      float t;
      uint32_t id;
      cast_ray<MODE>(sc, geom, p.o, p.d, stk, t, id, &last);
      nb++;
      bool want_shadow, did_nee;
      ShadowReq sr;
      bool alive = shade_bounce(sc, geom, p, t, id, want_shadow, did_nee, sr, &last);
      nn += did_nee ? 1u : 0u;
      if (want_shadow) {
        ns++;
        if (!occluded<MODE>(sc, geom, p.o, sr.d, sr.tfar, stk)) p.col = vadd(p.col, sr.contrib);
      }
      if (!alive) break;
      truncated = it + 1 == kFinishCap;
    }
    store_result(res, p.ordinal, p.col);
    bad = colour_not_finite(p.col);
  }
  wave_count(bad, &sum->stat_nan);
  wave_count(truncated, &sum->stat_truncated);
  // Statistics: one atomic per wave and counter (this kernel runs once per render).
  for (int off = 32; off > 0; off >>= 1) {
    nb += __shfl_down((int)nb, off, 64);
    nn += __shfl_down((int)nn, off, 64);
    ns += __shfl_down((int)ns, off, 64);
  }
  if ((threadIdx.x & 63u) == 0 && nb) {
    atomicAdd(&sum->stat_bounces, (unsigned long long)nb);
    if (nn) atomicAdd(&sum->stat_nee, (unsigned long long)nn);
    if (ns) atomicAdd(&sum->stat_shadow, (unsigned long long)ns);
  }
}

// accum[f] += sum over the chunk's samples of res[sample][f], in sample order
// (the order of a sequential loop over the samples, Raytracer.cpp:248-253).
// f runs over the 3*band_pixels floats of the band; accum points at the band.
__global__ __launch_bounds__(256) void k_reduce(const float* res, float* accum, uint32_t nfloats,
                                                uint32_t spp_chunk) {
  // Four floats per lane (16-byte loads) when the band allows it; the additions
  // of one float stay in sample order either way.
  if ((nfloats & 3u) == 0) {
    // (a grid smaller than the band: each lane takes several groups of four floats, one after the other)
    const uint32_t n4 = nfloats >> 2;
    for (uint32_t f = blockIdx.x * blockDim.x + threadIdx.x; f < n4; f += gridDim.x * blockDim.x) {
      float4 acc = reinterpret_cast<const float4*>(accum)[f];
      const float4* src = reinterpret_cast<const float4*>(res) + f;
#pragma unroll 8
      for (uint32_t s = 0; s < spp_chunk; s++) {
        float4 v = src[(size_t)s * n4];
        acc.x += v.x;
        acc.y += v.y;
        acc.z += v.z;
        acc.w += v.w;
      }
      reinterpret_cast<float4*>(accum)[f] = acc;
    }
    return;
  }
  uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= nfloats) return;
  float acc = accum[f];
  const float* src = res + f;
  for (uint32_t s = 0; s < spp_chunk; s++) acc += src[(size_t)s * nfloats];
  accum[f] = acc;
}

// The ordered sum split in two so that most of it can run beside k_finish, which keeps a few
// waves busy for a long time and leaves the memory system idle.
//  k_mark_unfinished  writes a marker (a NaN with a payload no arithmetic produces) into the
//                     result slots of the paths k_finish is about to run;
//  k_reduce_head      (other stream, beside k_finish) adds each float's samples in order up to its
//                     first marker and notes where it stopped; a slot k_finish fills meanwhile
//                     reads as marker or as result, and both are handled;
//  k_reduce_tail      (after both) goes on from there for the floats that were stopped.
// Every float's additions stay in sample order, so the sums keep the bits of k_reduce.
constexpr uint32_t kUnfinishedBits = 0x7fc0deadu;
__global__ __launch_bounds__(64) void k_mark_unfinished(Queue q, const SegCounter* seg, int qi, uint32_t cap,
                                                        float* res, const Primary* __restrict__ prp) {
  const uint32_t s = blockIdx.y;
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= seg_paths(seg[s].c[qi])) return;
  const uint32_t g0 = seg[s].gen_slot0;
  const uint32_t ordinal =
      (j >= g0 && prp->recompute) ? (uint32_t)seg[s].gen_ordinal0 + (j - g0) : q.s2[(size_t)s * cap + j].x;
  float* dst = res + 3 * (size_t)ordinal;
  const float marker = __uint_as_float(kUnfinishedBits);
  dst[0] = marker;
  dst[1] = marker;
  dst[2] = marker;
}
__device__ __forceinline__ void add_until_marker(float v, uint32_t s, float& acc, uint32_t& stop, uint32_t none) {
  if (stop == none) {
    if (__float_as_uint(v) == kUnfinishedBits)
      stop = s;
    else
      acc += v;
  }
}
// nfloats is a multiple of four; stop4[f] receives, per float, the sample it stopped at (spp_chunk = none).
__global__ __launch_bounds__(256) void k_reduce_head(const float* res, float* accum, uint4* stop4, uint32_t nfloats,
                                                     uint32_t spp_chunk) {
  const uint32_t n4 = nfloats >> 2;
  const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n4) return;
  float4 acc = reinterpret_cast<const float4*>(accum)[f];
  uint4 stop = make_uint4(spp_chunk, spp_chunk, spp_chunk, spp_chunk);
  const float4* src = reinterpret_cast<const float4*>(res) + f;
#pragma unroll 8
  for (uint32_t s = 0; s < spp_chunk; s++) {
    float4 v = src[(size_t)s * n4];
    add_until_marker(v.x, s, acc.x, stop.x, spp_chunk);
    add_until_marker(v.y, s, acc.y, stop.y, spp_chunk);
    add_until_marker(v.z, s, acc.z, stop.z, spp_chunk);
    add_until_marker(v.w, s, acc.w, stop.w, spp_chunk);
  }
  reinterpret_cast<float4*>(accum)[f] = acc;
  stop4[f] = stop;
}
__global__ __launch_bounds__(256) void k_reduce_tail(const float* res, float* accum, const uint4* stop4,
                                                     uint32_t nfloats, uint32_t spp_chunk) {
  const uint32_t n4 = nfloats >> 2;
  const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n4) return;
  const uint4 stop = stop4[f];
  uint32_t first = stop.x < stop.y ? stop.x : stop.y;
  first = first < stop.z ? first : stop.z;
  first = first < stop.w ? first : stop.w;
  if (first >= spp_chunk) return;
  float4 acc = reinterpret_cast<const float4*>(accum)[f];
  const float4* src = reinterpret_cast<const float4*>(res) + f;
#pragma unroll 8
  for (uint32_t s = first; s < spp_chunk; s++) {  // (few lanes get here; their loads should overlap)
    float4 v = src[(size_t)s * n4];
    if (s >= stop.x) acc.x += v.x;
    if (s >= stop.y) acc.y += v.y;
    if (s >= stop.z) acc.z += v.z;
    if (s >= stop.w) acc.w += v.w;
  }
  reinterpret_cast<float4*>(accum)[f] = acc;
}

__global__ void k_resolve(const float* accum, float* out, size_t n, float spp) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = accum[i] / spp;
}

}  // namespace rtx
