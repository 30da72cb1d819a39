// kernels_wide_tree.h -- the kernels of large meshes (four-child quantised tree in global memory): k_intersect_wide and
// k_shadow_wide (single-phase walk, lanes kept supplied from per-segment counters) and the shading kernels that follow
// a separate search kernel, k_shade_wave_append and k_shade_block_append (also used with rtx_config.fused_trace = 1).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "queues.h"

namespace rtx {

constexpr uint32_t kNoRay = 0xffffffffu;

#ifdef RTX_WALK_STATS
// Diagnostic build only (make EXTRA=-DRTX_WALK_STATS): what the waves of k_intersect_wide do, summed over the
// launches since the last read.  [0] wave iterations, [1] lanes at a node, [2] lanes at a triangle, summed over the
// iterations, [3] iterations with both kinds, [4] with nodes only, [5] with triangles only, [6] refill stops, [7] rays.
__device__ unsigned long long g_walk_stats[8];
#endif

// k_intersect for the wide tree in global memory.  In the single-phase walk a lane whose ray has
// ended would only idle until the longest ray of the wave ends (4 of 5 lanes on the 1 M-triangle
// mesh), so a wave keeps its lanes supplied instead: the waves of a segment take its rays 64 at a
// time from seg[s].grab until none are left, and once `refill_lanes` lanes are idle the wave
// stores their hits and hands them the next rays of the chunk it holds.  The chunk in hand sits in
// registers (lane L holds ray cb+L) and is handed out by lane shuffle; the base of the chunk after
// it has already been taken, and its rays are requested as soon as it becomes the chunk in hand.
// Against fixed runs of 256 rays per wave (which end in a tail in which the wave waits for its
// longest ray) the 1 M-triangle mesh's launch takes 12 % less.  A wave that finds the counter at or
// past the fill is done; the counter overshoots by up to two chunks per wave, which nothing reads.
// Supplying lanes costs about three steps of the walk (the set-up of the new rays runs on the idle
// lanes only, and the step after it waits for the stores and loads it issued): the threshold is
// the host's (measured: 32 of 64 lanes for the mesh, 16 for scenes of a few thousand triangles).
#ifndef RTX_WIDE_OCC
#define RTX_WIDE_OCC 7
#endif
template <int BLOCK>
__global__ __launch_bounds__(BLOCK, RTX_WIDE_OCC) void k_intersect_wide(SceneDev sc, Queue q, uint2* hits, SegCounter* seg,
                                                          int qi, uint32_t cap, TraceParams tp, uint32_t refill_lanes) {
  constexpr int MODE = kGlobalBvh4;
  const SegBlock sb = seg_block();
  const uint32_t s = sb.s;
  const uint32_t n = seg_paths(seg[s].c[qi]);
  if (sb.x * BLOCK >= n) return;  // more blocks than 64-ray chunks: nothing left for this one
  RTX_TRACE_PROLOGUE();
  const uint32_t lane = threadIdx.x & 63u;
  const size_t base = (size_t)s * cap;
  uint32_t* const grab = &seg[s].grab;
  // (the counter's answer stays in lane 0's register until the chunk is needed: reading it into a
  // scalar at once would wait for the round trip)
  auto take = [&]() -> uint32_t {
    uint32_t v = 0;
    if (lane == 0) v = atomicAdd(grab, 64u);
    return v;
  };
  uint32_t cb = (uint32_t)__builtin_amdgcn_readfirstlane((int)take());  // chunk in hand: rays [cb + used, ce)
  uint32_t ncb_v = take();                                              // the one after it (lane 0)
  cb = cb < n ? cb : n;
  uint32_t ce = cb + 64u < n ? cb + 64u : n;
  uint32_t used = 0;
  float4 pa = make_float4(0, 0, 0, 0);
  float2 pb = make_float2(0, 0);
  if (cb + lane < ce) {
    pa = q.r0[base + cb + lane];
    pb = q.r1[base + cb + lane];
  }
  Walk w;
  w.o = w.d = w.inv = mk3(0, 0, 0);
  w.cur = kTraversalDone;
  w.sp = 0;
  ClosestHit best = {INFINITY, kMiss, kMiss};
  uint32_t ri = kNoRay, object_id = kMiss;
#ifdef RTX_WALK_STATS
  unsigned long long ws_it = 0, ws_node = 0, ws_leaf = 0, ws_both = 0, ws_nonly = 0, ws_lonly = 0, ws_refill = 0;
#endif
  for (;;) {
    const bool walking = w.cur != kTraversalDone;
    const unsigned long long wm = __ballot(walking);
    const uint32_t idle = 64u - (uint32_t)__popcll(wm);
    if (wm == 0 || (idle >= refill_lanes && cb + used < ce)) {  // wave-uniform
#ifdef RTX_WALK_STATS
      ws_refill++;
#endif
      const unsigned long long im = ~wm;
      const uint32_t rank =
          __builtin_amdgcn_mbcnt_hi((uint32_t)(im >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)im, 0u));
      const uint32_t src = used + rank;
      const float ax = __shfl(pa.x, (int)src), ay = __shfl(pa.y, (int)src), az = __shfl(pa.z, (int)src);
      const float aw = __shfl(pa.w, (int)src), bx = __shfl(pb.x, (int)src), by = __shfl(pb.y, (int)src);
      if (!walking) {
        if (ri != kNoRay)  // as cast_ray: the mesh hit if there is one at or before the object's distance
          hits[base + ri] = make_uint2(__float_as_uint(best.t), best.slot != kMiss ? best.slot : object_id);
        ri = kNoRay;
        if (cb + src < ce) {
          ri = cb + src;
          walk_begin<MODE>(sc, mk3(ax, ay, az), mk3(aw, bx, by), w);
          best.t = INFINITY;
          object_id = kMiss;
          analytic_closest(sc, w.o, w.d, best.t, object_id);
          best.prim = kMiss;
          best.slot = kMiss;
        }
      }
      used += idle;
      if (cb + used >= ce && cb < n) {  // chunk used up: the next one becomes the chunk in hand
        const uint32_t ncb = (uint32_t)__builtin_amdgcn_readfirstlane((int)ncb_v);
        cb = ncb < n ? ncb : n;
        ce = cb + 64u < n ? cb + 64u : n;
        used = 0;
        if (cb < n) ncb_v = take();
        if (cb + lane < ce) {
          pa = q.r0[base + cb + lane];
          pb = q.r1[base + cb + lane];
        }
      }
      if (cb >= n && __ballot(ri != kNoRay) == 0) break;
    }
#ifdef RTX_WALK_STATS
    {
      const unsigned long long nm = __ballot(w.cur != kTraversalDone && w.cur >= 0);
      const unsigned long long lm = __ballot(w.cur != kTraversalDone && w.cur < 0);
      ws_it++;
      ws_node += __popcll(nm);
      ws_leaf += __popcll(lm);
      ws_both += (nm != 0 && lm != 0);
      ws_nonly += (nm != 0 && lm == 0);
      ws_lonly += (nm == 0 && lm != 0);
    }
#endif
    if (w.cur != kTraversalDone) closest_step(geom, stk, sc.eps, w, best);
  }
#ifdef RTX_WALK_STATS
  if (lane == 0) {
    atomicAdd(&g_walk_stats[0], ws_it);
    atomicAdd(&g_walk_stats[1], ws_node);
    atomicAdd(&g_walk_stats[2], ws_leaf);
    atomicAdd(&g_walk_stats[3], ws_both);
    atomicAdd(&g_walk_stats[4], ws_nonly);
    atomicAdd(&g_walk_stats[5], ws_lonly);
    atomicAdd(&g_walk_stats[6], ws_refill);
  }
#endif
}

// One bounce for every path of queue `qi`; survivors are appended to the same
// segment of `out`, finished paths store their colour to their sample's slot.
__global__ __launch_bounds__(256) void k_shade_wave_append(SceneDev sc, Queue in, Queue out, ShadowQueue sq,
                                               const uint2* hits, SegCounter* seg, int qi, uint32_t cap,
                                               float* res, const Primary* __restrict__ prp, Summary* sum) {
  const uint32_t s = blockIdx.y;
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t n = seg_paths(seg[s].c[qi]);
  if ((j & ~63u) >= n) return;  // whole waves beyond the fill leave; partial waves stay for the ballots
  const bool active = j < n;
  const uint32_t i = s * cap + j;

  PathState p;
  ShadowReq sr;
  bool alive = false, want_shadow = false, did_nee = false, bad = false;
  if (active) {
    load_path(sc, in, i, j >= seg[s].gen_slot0, *prp, (uint32_t)seg[s].gen_ordinal0 + (j - seg[s].gen_slot0), p);
    uint2 h = hits[i];
    alive = shade_bounce(sc, global_geom(sc), p, __uint_as_float(h.x), h.y, want_shadow, did_nee, sr);
    if (!alive) {
      store_result(res, p.ordinal, p.col);
      bad = colour_not_finite(p.col);
    }
  }
  wave_count(bad, &sum->stat_nan);
  uint32_t pos, spos;
  wave_append(alive, want_shadow, did_nee, &seg[s].c[qi ^ 1], pos, spos);
  if (alive) store_path(sc, out, s * cap + pos, p);
  if (want_shadow) {
    const uint32_t k = s * cap + spos;
    sq.q0[k] = make_float4(p.o.x, p.o.y, p.o.z, sr.tfar);
    sq.q1[k] = make_float4(sr.d.x, sr.d.y, sr.d.z, __uint_as_float(s * cap + pos));
    sq.q2[k] = make_float4(sr.contrib.x, sr.contrib.y, sr.contrib.z, 0.0f);
  }
}

// The same with one atomic per block.  Measured on MI355X: 7 % faster where most paths
// go on and light samples are queued as well (closed room with lights: two appended
// streams, long runs), 2 % slower where most paths end at once (open scene, sky).
__global__ __launch_bounds__(256) void k_shade_block_append(SceneDev sc, Queue in, Queue out, ShadowQueue sq,
                                                            const uint2* hits, SegCounter* seg, int qi,
                                                            uint32_t cap, float* res, const Primary* __restrict__ prp,
                                                            Summary* sum) {
  const uint32_t s = blockIdx.y;
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t n = seg_paths(seg[s].c[qi]);
  if (blockIdx.x * blockDim.x >= n) return;  // whole blocks beyond the fill leave; the rest stay for the barriers
  const bool active = j < n;
  const uint32_t i = s * cap + j;

  PathState p;
  ShadowReq sr;
  bool alive = false, want_shadow = false, did_nee = false, bad = false;
  if (active) {
    load_path(sc, in, i, j >= seg[s].gen_slot0, *prp, (uint32_t)seg[s].gen_ordinal0 + (j - seg[s].gen_slot0), p);
    uint2 h = hits[i];
    alive = shade_bounce(sc, global_geom(sc), p, __uint_as_float(h.x), h.y, want_shadow, did_nee, sr);
    if (!alive) {
      store_result(res, p.ordinal, p.col);
      bad = colour_not_finite(p.col);
    }
  }
  wave_count(bad, &sum->stat_nan);
  uint32_t pos, spos;
  block_append(alive, want_shadow, did_nee, &seg[s].c[qi ^ 1], pos, spos);
  if (alive) store_path(sc, out, s * cap + pos, p);
  if (want_shadow) {
    const uint32_t k = s * cap + spos;
    sq.q0[k] = make_float4(p.o.x, p.o.y, p.o.z, sr.tfar);
    sq.q1[k] = make_float4(sr.d.x, sr.d.y, sr.d.z, __uint_as_float(s * cap + pos));
    sq.q2[k] = make_float4(sr.contrib.x, sr.contrib.y, sr.contrib.z, 0.0f);
  }
}

// k_shadow for the wide tree in global memory, with the lanes of a wave kept supplied as in
// k_intersect_wide (chunks of 64 light samples taken from seg[s].grab_shadow).
template <int BLOCK>
__global__ __launch_bounds__(BLOCK, 8) void k_shadow_wide(SceneDev sc, Queue paths, ShadowQueue sq, SegCounter* seg,
                                                       int q_out, uint32_t cap, TraceParams tp,
                                                       uint32_t refill_lanes) {
  constexpr int MODE = kGlobalBvh4;
  const SegBlock sb = seg_block();
  const uint32_t s = sb.s;
  const uint32_t n = seg_shadow(seg[s].c[q_out]);
  if (sb.x * BLOCK >= n) return;
  RTX_TRACE_PROLOGUE();
  const uint32_t lane = threadIdx.x & 63u;
  const size_t base = (size_t)s * cap;
  uint32_t* const grab = &seg[s].grab_shadow;
  auto take = [&]() -> uint32_t {
    uint32_t v = 0;
    if (lane == 0) v = atomicAdd(grab, 64u);
    return v;
  };
  uint32_t cb = (uint32_t)__builtin_amdgcn_readfirstlane((int)take());
  uint32_t ncb_v = take();
  cb = cb < n ? cb : n;
  uint32_t ce = cb + 64u < n ? cb + 64u : n;
  uint32_t used = 0;
  float4 pa = make_float4(0, 0, 0, 0), pb = make_float4(0, 0, 0, 0);
  if (cb + lane < ce) {
    pa = sq.q0[base + cb + lane];
    pb = sq.q1[base + cb + lane];
  }
  Walk w;
  w.o = w.d = w.inv = mk3(0, 0, 0);
  w.cur = kTraversalDone;
  w.sp = 0;
  uint32_t ri = kNoRay, slot = 0;
  float tfar = 0.0f;
  bool hit = false;
  for (;;) {
    const bool walking = w.cur != kTraversalDone;
    const unsigned long long wm = __ballot(walking);
    const uint32_t idle = 64u - (uint32_t)__popcll(wm);
    if (wm == 0 || (idle >= refill_lanes && cb + used < ce)) {  // wave-uniform
      const unsigned long long im = ~wm;
      const uint32_t rank =
          __builtin_amdgcn_mbcnt_hi((uint32_t)(im >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)im, 0u));
      const uint32_t src = used + rank;
      const float ax = __shfl(pa.x, (int)src), ay = __shfl(pa.y, (int)src), az = __shfl(pa.z, (int)src);
      const float aw = __shfl(pa.w, (int)src), bx = __shfl(pb.x, (int)src), by = __shfl(pb.y, (int)src);
      const float bz = __shfl(pb.z, (int)src), bw = __shfl(pb.w, (int)src);
      if (!walking) {
        if (ri != kNoRay && !hit) add_contribution(paths, sq, (uint32_t)(base + ri), slot);
        ri = kNoRay;
        if (cb + src < ce) {
          ri = cb + src;
          tfar = aw;
          slot = __float_as_uint(bw);
          const f3 o = mk3(ax, ay, az), d = mk3(bx, by, bz);
          hit = analytic_any(sc, o, d, tfar);
          if (!hit) walk_begin<MODE>(sc, o, d, w);
        }
      }
      used += idle;
      if (cb + used >= ce && cb < n) {
        const uint32_t ncb = (uint32_t)__builtin_amdgcn_readfirstlane((int)ncb_v);
        cb = ncb < n ? ncb : n;
        ce = cb + 64u < n ? cb + 64u : n;
        used = 0;
        if (cb < n) ncb_v = take();
        if (cb + lane < ce) {
          pa = sq.q0[base + cb + lane];
          pb = sq.q1[base + cb + lane];
        }
      }
      if (cb >= n && __ballot(ri != kNoRay) == 0) break;
    }
    if (w.cur != kTraversalDone) any_step(geom, stk, sc.eps, tfar, w, hit);
  }
}

}  // namespace rtx
