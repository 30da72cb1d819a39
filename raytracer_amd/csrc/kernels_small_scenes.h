// kernels_small_scenes.h -- the kernels of scenes whose tree fits LDS, or that have no tree at all (the reference's
// own scenes): k_trace_shade (search and shading of a bounce in one launch, two bounces per launch for the sphere
// scenes, triangle tests and light samples as tasks of the block for rooms of a few surfaces), k_intersect and k_shadow
// (the per-lane walks, used with rtx_config.fused_trace = 1 and for queued light samples), k_shadow_leaf_tasks.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "queues.h"

namespace rtx {

template <int BLOCK, int MODE>
__global__ __launch_bounds__(BLOCK, MODE == kGlobalBvh2 ? 4 : 8) void k_intersect(SceneDev sc, Queue q, uint2* hits,
                                                                             const SegCounter* seg, int qi,
                                                                             uint32_t cap, TraceParams tp,
                                                                             const Primary* __restrict__ prp) {
  const SegBlock sb = MODE == kGlobalBvh4 ? seg_block() : SegBlock{blockIdx.y, blockIdx.x};
  const uint32_t s = sb.s;
  const uint32_t n = seg_paths(seg[s].c[qi]);
  if (sb.x * BLOCK >= n) return;  // the whole block is beyond the fill: skip the staging too
  RTX_TRACE_PROLOGUE();
  // Rays from this slot on are recomputed, not loaded.
  const uint32_t computed0 = prp->recompute ? seg[s].gen_slot0 : 0xffffffffu;
  const uint32_t k0 = (uint32_t)seg[s].gen_ordinal0;
  // Block-stride loop over the segment: a block that staged the scene in LDS
  // amortises that over several batches of rays.
  // The next batch's ray is loaded before the current one is traced, so the HBM
  // latency of the load hides behind the traversal.
  const uint32_t stride = gridDim.x * BLOCK;
  uint32_t j = sb.x * BLOCK + threadIdx.x;
  float4 a = make_float4(0, 0, 0, 0);
  float2 b = make_float2(0, 0);
  if (j < n && j < computed0) {
    a = q.r0[s * cap + j];
    b = q.r1[s * cap + j];
  }
  while (j < n) {
    const uint32_t i = s * cap + j;
    const uint32_t jn = j + stride;
    float4 an = a;
    float2 bn = b;
    if (jn < n && jn < computed0) {
      an = q.r0[s * cap + jn];
      bn = q.r1[s * cap + jn];
    }
    f3 o = mk3(a.x, a.y, a.z), d = mk3(a.w, b.x, b.y);
    if (j >= computed0) {
      Rng rng;
      primary_ray(prp->cam, prp->rp, k0 + (j - computed0), o, d, rng, sc.rng_streams != 0);
    }
    float t;
    uint32_t id;
    cast_ray<MODE>(sc, geom, o, d, stk, t, id);
    hits[i] = make_uint2(__float_as_uint(t), id);
    a = an;
    b = bn;
    j = jn;
  }
}

// Closest mesh hit for the rays of a whole block when the tree has a handful of leaves (Cornell's seventeen).
// In a walk a ray needs about eight node and 1.2 leaf visits, but its wave runs twenty node rounds and seven
// leaf rounds, the latter with a sixth of the lanes at work, because the lanes' walks differ.  Here every lane
// tests the boxes of all leaves (read from LDS as broadcasts, no stack, no divergence; what an ordered walk would have
// culled behind a hit is a twentieth of the leaf visits in such a scene), the (ray, leaf) pairs go to a queue
// in LDS, and the block's waves test them in dense batches of 64; a hit is merged into its ray's record with a
// 64-bit minimum of (t, primitive id, slot): the smallest t and, among equal t, the lowest primitive id, which is
// the rule of Mesh::intersect (reference Obj.h:184-195).  Every thread of the block calls this (barriers);
// `active` lanes bring a ray, and get the closest hit with t <= t_limit (best_slot kMiss, best_t = t_limit:
// none).  The queue holds sc.leaf_task_cap pairs per thread on average (a ray meets 1.3 leaf boxes); a lane whose
// pairs do not fit tests its leaves itself.  LDS at `ws`: leaf_task_lds_bytes; the caller has set ws[0] to
// (0, queue capacity, ...) as uint32 before its last barrier.
#ifdef RTX_PHASE_STATS
// Diagnostic build only (make EXTRA=-DRTX_PHASE_STATS): shader cycles the waves of the bounce kernel k_trace_shade
// spend between the stamps below, summed over the waves of every 64th block; [15] counts those waves.  tools/phase_stats.py reads them.
__device__ unsigned long long g_phase_stats[16];
#define RTX_STAMP(clock, k)                                                                   \
  do {                                                                                        \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();                             \
    if ((threadIdx.x & 63u) == 0 && (blockIdx.x & 63u) == 0) atomicAdd(&g_phase_stats[k], now_ - (clock)); \
    (clock) = now_;                                                                           \
  } while (0)
#else
#define RTX_STAMP(clock, k) \
  do {                      \
  } while (0)
#endif
struct LeafTaskHead {
  uint32_t count, valid_end;    // pairs reserved so far; pairs [0, valid_end) are written
  uint32_t count2, valid_end2;  // the same for a second round of tasks in the same launch (no barrier to reset the first)
};
// head | leaf words | ray origins | ray directions | closest-hit keys (flags of a first any-hit round) | pairs | flags
// of a second round.  The second round's flags have words of their own: a lane that enters the second round writes
// its flag while another wave may still be reading its ray's key of the first.
__host__ __device__ constexpr size_t leaf_task_lds_bytes(uint32_t block, uint32_t cap_per_thread) {
  return sizeof(LeafTaskHead) + 4 * kLeafTaskMaxLeaves + (size_t)block * (32 + 8 + 2 * cap_per_thread + 4);
}
// mask = 2 mask + (the ray of walk `w` meets the box (a, b) between w.tmin and tmax).  The comparison's result
// enters the addition as its carry: one instruction where a select, a shift and an or stand otherwise.
__device__ __forceinline__ void shift_in_box_met(uint32_t& mask, float4 a, float4 b, const Walk& w, float tmax) {
  float tn, tf;
  slab_interval(a.x, a.y, a.z, a.w, b.x, b.y, w.inv, w.add_lo, w.add_hi, w.tmin, tmax, tn, tf);
  asm("v_cmp_le_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(mask) : "v"(tn), "v"(tf) : "vcc");
}
// Bit l set: the ray of walk `w` meets the box of leaf l between w.tmin and tmax.  `lb`: the box list in LDS
// (stage_leaf_task_geom_lds); every lane reads the same address.  Four boxes a turn, written out (the compiler
// does not unroll a loop around the asm statement), so that the reads of a turn are in flight together.
__device__ __forceinline__ uint32_t leaf_boxes_met(const float4* lb, uint32_t nl, const Walk& w, float tmax) {
  uint32_t mask = 0, l = 0;
  for (; l + 4 <= nl; l += 4) {
    const float4 a0 = lb[2 * l], b0 = lb[2 * l + 1], a1 = lb[2 * l + 2], b1 = lb[2 * l + 3];
    const float4 a2 = lb[2 * l + 4], b2 = lb[2 * l + 5], a3 = lb[2 * l + 6], b3 = lb[2 * l + 7];
    shift_in_box_met(mask, a0, b0, w, tmax);
    shift_in_box_met(mask, a1, b1, w, tmax);
    shift_in_box_met(mask, a2, b2, w, tmax);
    shift_in_box_met(mask, a3, b3, w, tmax);
  }
  for (; l < nl; l++) shift_in_box_met(mask, lb[2 * l], lb[2 * l + 1], w, tmax);
  return nl != 0 ? __brev(mask) >> (32u - nl) : 0u;  // (the bits arrived in reverse)
}
template <int BLOCK>
__device__ __forceinline__ void leaf_task_begin(const SceneDev& sc, float4* ws) {
  LeafTaskHead* const head = reinterpret_cast<LeafTaskHead*>(ws);
  uint32_t* const leaf_word = reinterpret_cast<uint32_t*>(ws + 1);
  if (threadIdx.x == 0) {
    head->count = 0;
    head->valid_end = sc.leaf_task_cap * BLOCK;
    head->count2 = 0;
    head->valid_end2 = sc.leaf_task_cap * BLOCK;
  }
  if (threadIdx.x < sc.num_leaf_boxes) leaf_word[threadIdx.x] = __float_as_uint(sc.leaf_box[2 * threadIdx.x + 1].z);
}
template <int BLOCK>
__device__ __forceinline__ void block_closest_by_leaf_tasks(const SceneDev& sc, const Geom& g, float4* ws, bool active,
                                                            f3 o, f3 d, float t_limit, float& best_t,
                                                            uint32_t& best_slot, unsigned long long& clk) {
  LeafTaskHead* const head = reinterpret_cast<LeafTaskHead*>(ws);
  uint32_t* const leaf_word = reinterpret_cast<uint32_t*>(ws + 1);
  float4* const ray_o = ws + 1 + kLeafTaskMaxLeaves / 4;
  float4* const ray_d = ray_o + BLOCK;
  unsigned long long* const best = reinterpret_cast<unsigned long long*>(ray_d + BLOCK);
  uint16_t* const tasks = reinterpret_cast<uint16_t*>(best + BLOCK);
  const uint32_t tid = threadIdx.x, cap = sc.leaf_task_cap * BLOCK;
  uint32_t own = 0;  // leaves this lane tests itself (its pairs did not fit)
  if (active) {
    Walk w;
    walk_begin<kLdsBvh2>(sc, o, d, w);
    RTX_STAMP(clk, 12);
    uint32_t mask = leaf_boxes_met(g.nodes, sc.num_leaf_boxes, w, t_limit);
    RTX_STAMP(clk, 13);
    ray_o[tid] = make_float4(o.x, o.y, o.z, 0.0f);
    ray_d[tid] = make_float4(d.x, d.y, d.z, 0.0f);
    best[tid] = ((unsigned long long)__float_as_uint(t_limit) << 32) | 0xffffffffull;  // any hit at t_limit wins over none
    if (mask != 0) {
      const uint32_t cnt = (uint32_t)__popc(mask);
      uint32_t at = atomicAdd(&head->count, cnt);
      if (at + cnt <= cap) {
        while (mask != 0) {
          const uint32_t l = (uint32_t)__builtin_ctz(mask);
          mask &= mask - 1u;
          tasks[at++] = (uint16_t)((tid << 5) | l);
        }
      } else {
        // (every reservation after this one starts beyond the capacity as well, so the written pairs are the
        // ones below the lowest start of a lane that did not fit)
        atomicMin(&head->valid_end, at);
        own = mask;
      }
    }
  }
  RTX_STAMP(clk, 2);
  __syncthreads();
  RTX_STAMP(clk, 3);
  const uint32_t ntasks = min(head->count, head->valid_end);
  for (uint32_t k = tid; k < ntasks; k += BLOCK) {
    const uint32_t task = tasks[k], r = task >> 5;
    const float4 O = ray_o[r], D = ray_d[r];
    float t = INFINITY;
    uint32_t prim = kMiss, slot = kMiss;
    leaf_closest<kLdsBvh2>(g, leaf_word[task & 31u], mk3(O.x, O.y, O.z), mk3(D.x, D.y, D.z), sc.eps, t, prim, slot);
    if (slot != kMiss)
      atomicMin(&best[r], ((unsigned long long)__float_as_uint(t) << 32) | ((unsigned long long)prim << 16) | slot);
  }
  while (own != 0) {
    const uint32_t l = (uint32_t)__builtin_ctz(own);
    own &= own - 1u;
    float t = INFINITY;
    uint32_t prim = kMiss, slot = kMiss;
    leaf_closest<kLdsBvh2>(g, leaf_word[l], o, d, sc.eps, t, prim, slot);
    if (slot != kMiss)
      atomicMin(&best[tid], ((unsigned long long)__float_as_uint(t) << 32) | ((unsigned long long)prim << 16) | slot);
  }
  RTX_STAMP(clk, 4);
  __syncthreads();
  RTX_STAMP(clk, 5);
  best_t = t_limit;
  best_slot = kMiss;
  if (active) {
    const unsigned long long key = best[tid];
    best_t = __uint_as_float((uint32_t)(key >> 32));
    best_slot = (uint32_t)key == 0xffffffffu ? kMiss : ((uint32_t)key & 0xffffu);
  }
}

// The same for occlusion: is any triangle of the mesh hit with eps < t < tfar (mesh_any)?  A pair that finds a
// hit sets its ray's flag; the other pairs of that ray are tested all the same (a walk would have stopped), which
// costs little where most segments are free.  The caller has run leaf_task_begin and a barrier since the last call.
template <int BLOCK, bool SECOND = false>
__device__ __forceinline__ bool block_any_by_leaf_tasks(const SceneDev& sc, const Geom& g, float4* ws, bool active,
                                                        f3 o, f3 d, float tfar, unsigned long long& clk) {
  LeafTaskHead* const head = reinterpret_cast<LeafTaskHead*>(ws);
  uint32_t* const leaf_word = reinterpret_cast<uint32_t*>(ws + 1);
  float4* const ray_o = ws + 1 + kLeafTaskMaxLeaves / 4;
  float4* const ray_d = ray_o + BLOCK;
  uint16_t* const tasks = reinterpret_cast<uint16_t*>(reinterpret_cast<unsigned long long*>(ray_d + BLOCK) + BLOCK);
  const uint32_t tid = threadIdx.x, cap = sc.leaf_task_cap * BLOCK;
  uint32_t* const flag = SECOND ? reinterpret_cast<uint32_t*>(tasks + cap) : reinterpret_cast<uint32_t*>(ray_d + BLOCK);
  uint32_t* const count = SECOND ? &head->count2 : &head->count;
  uint32_t* const valid_end = SECOND ? &head->valid_end2 : &head->valid_end;
  uint32_t own = 0;
  if (active) {
    Walk w;
    walk_begin<kLdsBvh2>(sc, o, d, w);
    uint32_t mask = leaf_boxes_met(g.nodes, sc.num_leaf_boxes, w, tfar);
    ray_o[tid] = make_float4(o.x, o.y, o.z, tfar);
    ray_d[tid] = make_float4(d.x, d.y, d.z, 0.0f);
    flag[tid] = 0;
    if (mask != 0) {
      const uint32_t cnt = (uint32_t)__popc(mask);
      uint32_t at = atomicAdd(count, cnt);
      if (at + cnt <= cap) {
        while (mask != 0) {
          const uint32_t l = (uint32_t)__builtin_ctz(mask);
          mask &= mask - 1u;
          tasks[at++] = (uint16_t)((tid << 5) | l);
        }
      } else {
        atomicMin(valid_end, at);
        own = mask;
      }
    }
  }
  RTX_STAMP(clk, 7);
  __syncthreads();
  RTX_STAMP(clk, 8);
  const uint32_t ntasks = min(*count, *valid_end);
  for (uint32_t k = tid; k < ntasks; k += BLOCK) {
    const uint32_t task = tasks[k], r = task >> 5;
    const float4 O = ray_o[r], D = ray_d[r];
    if (leaf_any<kLdsBvh2>(g, leaf_word[task & 31u], mk3(O.x, O.y, O.z), mk3(D.x, D.y, D.z), sc.eps, O.w)) flag[r] = 1;
  }
  bool hit = false;
  while (own != 0 && !hit) {
    const uint32_t l = (uint32_t)__builtin_ctz(own);
    own &= own - 1u;
    hit = leaf_any<kLdsBvh2>(g, leaf_word[l], o, d, sc.eps, tfar);
  }
  RTX_STAMP(clk, 9);
  __syncthreads();
  RTX_STAMP(clk, 10);
  return active && (hit || flag[tid] != 0);
}

// k_intersect and k_shade_wave_append in one launch, for scenes whose closest-hit search is
// a handful of tests (tree staged in LDS, a few nodes): the search then costs less than
// carrying the ray to a kernel of its own and the hit back, so each path is read once per
// bounce and the hit stays in registers.
// (eight waves per SIMD: the compiler then keeps to 96 scalar registers as well, which it passed by five
// otherwise, and a wave of this kernel mostly waits: sample2 8.8 -> 8.3 ms, Cornell's bounces 51.7 -> 49.4)
// (kLdsLeafTasks: 66 vector registers, so seven waves per SIMD; built for eight it spills 51 and Cornell's frame
// goes from 47 to 55 ms)
// BOUNCES = 2 (light-free scenes without a tree): a path that goes on after the launch's first bounce makes its
// second in the same lane.  Such a kernel moves bytes more than it computes (a path is 48 B read and 48 B written per
// bounce, and four in ten go on), and every second write and read back of a path falls away; the lanes of the
// paths that ended idle through the second bounce, which costs less than the bytes (sample2 8.0 -> 7.0 ms frame by
// frame; three bounces per launch: 9.0).
template <int BLOCK, int MODE, bool LIGHTS, bool INLINE_LIGHT_SAMPLES = false, int BOUNCES = 1, int PLAIN = 0>
__global__ __launch_bounds__(BLOCK, MODE == kLdsLeafTasks ? 7 : 8) void k_trace_shade(SceneDev sc, Queue in, Queue out, ShadowQueue sq,
                                                       SegCounter* seg, int qi, uint32_t cap, float* res,
                                                       TraceParams tp, const Primary* __restrict__ prp, Summary* sum) {
  // PLAIN >= 1: the host has seen that the scene has no boxes and no planes and that the call is the path tracer
  // without a bounce limit, with the reference's one generator cycle (what the reference's scenes are): with these known when the kernel is built, their
  // branches, and the scalar registers that held their operands, are gone.  PLAIN == 2: under the sky as well, with
  // nothing that emits and a handful of surfaces (the reference's sphere scenes).
  if constexpr (PLAIN >= 1) {
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
  const uint32_t s = blockIdx.y;
  const uint32_t n = seg_paths(seg[s].c[qi]);
  if (blockIdx.x * BLOCK >= n) return;  // whole blocks beyond the fill leave; the rest stay for the barrier and the ballots
  [[maybe_unused]] unsigned long long clk = 0;
#ifdef RTX_PHASE_STATS
  clk = __builtin_amdgcn_s_memtime();
  if constexpr (MODE == kLdsLeafTasks)
    if ((threadIdx.x & 63u) == 0 && (blockIdx.x & 63u) == 0) atomicAdd(&g_phase_stats[15], 1ull);
#endif
  constexpr int WALK = MODE == kLdsLeafTasks ? kLdsBvh2 : MODE;   // (how a lane of this kernel would walk the tree)
  extern __shared__ float4 lds_dyn[];
  Stack stk;
  stk.lds = (lds_int_ptr)(reinterpret_cast<int*>(lds_dyn) + threadIdx.x);
  stk.stride = BLOCK;
  stk.lds_entries = tp.lds_entries;
  stk.all_in_lds = true;
  stk.spill_stride = tp.spill_stride;
  stk.spill = (global_int_ptr)(tp.spill + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * BLOCK + threadIdx.x));
  // LDS: [per-lane stacks | scene] for the walks, [scene | leaf-task workspace] for kLdsLeafTasks
  const uint32_t scene_at = MODE == kLdsLeafTasks ? 0u : (tp.lds_entries * BLOCK) / 4;
  float4* const task_ws = lds_dyn + kLdsRecordStride * (sc.num_nodes + sc.num_tris);
  const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
  const bool active = j < n;
  const uint32_t i = s * cap + j;
  PathState p;
  if constexpr (MODE == kLdsLeafTasks) leaf_task_begin<BLOCK>(sc, task_ws);  // (ahead of the staging's barrier)
  const Geom geom = MODE == kLdsLeafTasks ? stage_leaf_task_geom_lds(sc, lds_dyn, threadIdx.x, BLOCK)
                    : WALK == kLdsBvh2   ? stage_geom_lds(sc, lds_dyn + scene_at, threadIdx.x, BLOCK)
                                         : global_geom(sc);
  if constexpr (MODE == kLdsLeafTasks) RTX_STAMP(clk, 0);

  ShadowReq sr;
  SurfaceHit sh;
  bool alive = false, want_shadow = false, did_nee = false, bad = false;
  float t = INFINITY;
  uint32_t id = kMiss;
  if constexpr (MODE == kLdsLeafTasks) {
    // cast_ray with the mesh part done by the whole block
    float ts = INFINITY;
    uint32_t sid = kMiss;
    // (requesting the path ahead of the staging, as k_shadow_leaf_tasks does with its samples: no faster here)
    p.o = p.d = mk3(0.0f, 0.0f, 0.0f);
    if (active) {
      load_path(sc, in, i, j >= seg[s].gen_slot0, *prp, (uint32_t)seg[s].gen_ordinal0 + (j - seg[s].gen_slot0), p);
      analytic_closest(sc, p.o, p.d, ts, sid);
    }
    RTX_STAMP(clk, 1);
    uint32_t slot;
    block_closest_by_leaf_tasks<BLOCK>(sc, geom, task_ws, active, p.o, p.d, ts, t, slot, clk);
    id = slot != kMiss ? slot : sid;
  }
  // Without lights a bounce queues no light sample, so the append counts paths only, and which paths go on is
  // known once the roulette is through: the slots are reserved there and the atomic's round trip runs beside
  // the sampling of the next direction.
  constexpr bool reserve_early = !LIGHTS;  // (LIGHTS: the scene has any, sc.num_lights != 0)
  bool live = active;  // this lane carries a path
  if constexpr (MODE != kLdsLeafTasks) {
#ifdef RTX_PHASE_STATS
    if ((threadIdx.x & 63u) == 0 && (blockIdx.x & 63u) == 0) atomicAdd(&g_phase_stats[15], 1ull);
#endif
    RTX_STAMP(clk, 0);
    if (active)
      load_path(sc, in, i, j >= seg[s].gen_slot0, *prp, (uint32_t)seg[s].gen_ordinal0 + (j - seg[s].gen_slot0), p);
    RTX_STAMP(clk, 1);
    if constexpr (BOUNCES > 1) {
      // The first BOUNCES - 1 bounces of a launch: a path that goes on stays in its lane (its state is neither
      // written nor read back), the lanes of the paths that ended idle.  Counted in seg[s].more_bounces.
      static_assert(!LIGHTS, "a bounce that samples a light queues the sample against the path's slot");
#pragma unroll 1
      for (int b = 1; b < BOUNCES; b++) {
        bool goes_on = false;
        if (live) {
          cast_ray<WALK>(sc, geom, p.o, p.d, stk, t, id);
          RTX_STAMP(clk, 2);
          goes_on = shade_bounce(sc, global_geom(sc), p, t, id, want_shadow, did_nee, sr);
          if (!goes_on) {
            store_result(res, p.ordinal, p.col);
            bad = bad || colour_not_finite(p.col);
          }
          RTX_STAMP(clk, 3);
        }
        live = goes_on;
        wave_count(live, &seg[s].more_bounces);
        RTX_STAMP(clk, 4);
      }
    }
  }
  if (live) {
    if constexpr (MODE != kLdsLeafTasks) {
      cast_ray<WALK>(sc, geom, p.o, p.d, stk, t, id);
      RTX_STAMP(clk, 5);
    }
    alive = reserve_early ? shade_arrive<true>(sc, global_geom(sc), p, t, id, sh)
                          : shade_bounce(sc, global_geom(sc), p, t, id, want_shadow, did_nee, sr);
    if (!alive) {
      store_result(res, p.ordinal, p.col);
      bad = bad || colour_not_finite(p.col);
    }
  }
  wave_count(bad, &sum->stat_nan);
  RTX_STAMP(clk, 6);
  bool queue_sample = want_shadow;
  if constexpr (INLINE_LIGHT_SAMPLES) {
    // The light sample is traced here, as a second round of tasks of the block: it is counted, not queued.
    static_assert(MODE == kLdsLeafTasks && LIGHTS, "the block's task queue is what traces the sample");
    if (!want_shadow) {  // (lanes without a sample bring defined values to the block's call)
      sr.d = mk3(0.0f, 0.0f, 0.0f);
      sr.tfar = 0.0f;
    }
    const bool try_light = want_shadow && !analytic_any(sc, p.o, sr.d, sr.tfar);
    const bool occ = block_any_by_leaf_tasks<BLOCK, true>(sc, geom, task_ws, try_light, p.o, sr.d, sr.tfar, clk);
    if (try_light && !occ) p.col = vadd(p.col, sr.contrib);
    queue_sample = false;
  }
  uint32_t pos, spos = 0;  // (the per-block append of k_shade_block_append measured slower here)
  if (reserve_early) {
    const PathTicket tk = wave_reserve_paths(alive, &seg[s].c[qi ^ 1]);
    if (alive) shade_leave(sc, p, sh, want_shadow, did_nee, sr);
    pos = wave_path_slot(tk);
    RTX_STAMP(clk, 7);
  } else {
    wave_append(alive, want_shadow, did_nee, &seg[s].c[qi ^ 1], pos, spos);
  }
  if (alive) store_path(sc, out, s * cap + pos, p);
  if (queue_sample) {
    const uint32_t k = s * cap + spos;
    sq.q0[k] = make_float4(p.o.x, p.o.y, p.o.z, sr.tfar);
    sq.q1[k] = make_float4(sr.d.x, sr.d.y, sr.d.z, __uint_as_float(s * cap + pos));
    sq.q2[k] = make_float4(sr.contrib.x, sr.contrib.y, sr.contrib.z, 0.0f);
  }
  RTX_STAMP(clk, 11);
}

// Traces the NEE segments queued by k_shade; an unoccluded one adds its
// contribution to the colour of the path it belongs to (Integrator.h:217-223).
template <int BLOCK, int MODE>
__global__ __launch_bounds__(BLOCK, 8) void k_shadow(SceneDev sc, Queue paths, ShadowQueue sq,
                                                  const SegCounter* seg, int q_out, uint32_t cap,
                                                  TraceParams tp) {
  const SegBlock sb = MODE == kGlobalBvh4 ? seg_block() : SegBlock{blockIdx.y, blockIdx.x};
  const uint32_t s = sb.s;
  const uint32_t n = seg_shadow(seg[s].c[q_out]);
  if (sb.x * BLOCK >= n) return;
  RTX_TRACE_PROLOGUE();
  for (uint32_t j = sb.x * BLOCK + threadIdx.x; j < n; j += gridDim.x * BLOCK) {
    const uint32_t k = s * cap + j;
    float4 a = sq.q0[k];
    float4 b = sq.q1[k];
    bool occ = occluded<MODE>(sc, geom, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), a.w, stk);
    if (!occ) add_contribution(paths, sq, k, __float_as_uint(b.w));
  }
}

// k_shadow for trees of a handful of leaves (block_any_by_leaf_tasks), one batch of light samples per block.
// Dynamic LDS: [scene | leaf-task workspace].
template <int BLOCK>
__global__ __launch_bounds__(BLOCK, 7) void k_shadow_leaf_tasks(SceneDev sc, Queue paths, ShadowQueue sq,
                                                             const SegCounter* seg, int q_out, uint32_t cap) {
  const uint32_t s = blockIdx.y;
  const uint32_t n = seg_shadow(seg[s].c[q_out]);
  if (blockIdx.x * BLOCK >= n) return;
  extern __shared__ float4 lds_dyn[];
  float4* const task_ws = lds_dyn + kLdsRecordStride * (sc.num_nodes + sc.num_tris);
  const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
  const bool active = j < n;
  const uint32_t k = s * cap + (active ? j : 0u);
  float4 a = make_float4(0, 0, 0, 0), b = a;
  if (active) {  // (requested ahead of the staging, whose copy and barrier then run beside the round trip)
    a = sq.q0[k];
    b = sq.q1[k];
  }
  leaf_task_begin<BLOCK>(sc, task_ws);
  const Geom geom = stage_leaf_task_geom_lds(sc, lds_dyn, threadIdx.x, BLOCK);  // (ends with a barrier)
  const bool free_so_far = active && !analytic_any(sc, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), a.w);
  [[maybe_unused]] unsigned long long clk = 0;
  const bool occ = block_any_by_leaf_tasks<BLOCK>(sc, geom, task_ws, free_so_far, mk3(a.x, a.y, a.z),
                                                  mk3(b.x, b.y, b.z), a.w, clk);
  if (free_so_far && !occ) add_contribution(paths, sq, k, __float_as_uint(b.w));
}

}  // namespace rtx
