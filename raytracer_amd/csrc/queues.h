// queues.h -- queue bookkeeping: appends by wave and by block (ballot + mbcnt prefix, one atomic per wave or block),
// the per-round plan (k_begin, k_plan), the kernel that writes primary rays where they are not recomputed
// (k_generate), the traversal kernels' stack set-up and the segment -> XCD mapping.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "shading.h"

namespace rtx {

// ----------------------------------------------------------------------------
// Wave-level append to a segment: every lane of the wave must call this
// together.  One 64-bit atomic per wave reserves queue slots for the lanes with
// `take_path`, shadow-queue slots for the lanes with `take_shadow`, and counts
// the lanes that sampled a light.
__device__ __forceinline__ void wave_append(bool take_path, bool take_shadow, bool did_nee,
                                            unsigned long long* counter, uint32_t& path_pos,
                                            uint32_t& shadow_pos) {
  const unsigned long long mp = __ballot(take_path), ms = __ballot(take_shadow), mn = __ballot(did_nee);
  const unsigned long long add = (unsigned long long)__popcll(mp) |
                                 ((unsigned long long)__popcll(ms) << kPathBits) |
                                 ((unsigned long long)__popcll(mn) << (kPathBits + kShadowBits));
  uint32_t lo = 0, hi = 0;
  if (add) {
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if (lane == 0) {
      unsigned long long old = atomicAdd(counter, add);
      lo = (uint32_t)old;
      hi = (uint32_t)(old >> 32);
    }
    lo = __builtin_amdgcn_readfirstlane(lo);
    hi = __builtin_amdgcn_readfirstlane(hi);
  }
  const unsigned long long old = ((unsigned long long)hi << 32) | lo;
  path_pos = seg_paths(old) +
             __builtin_amdgcn_mbcnt_hi((uint32_t)(mp >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mp, 0u));
  shadow_pos = seg_shadow(old) +
               __builtin_amdgcn_mbcnt_hi((uint32_t)(ms >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ms, 0u));
}

// wave_append in two halves, for a caller that only appends paths (no light samples) and has work to do between
// reserving the slots and filling them: the atomic's round trip then runs beside that work.  The counter's old
// value stays in lane 0's register until wave_path_slot asks for it.
struct PathTicket {
  unsigned long long mask;  // lanes that take a slot
  uint32_t old_lo;          // low word of the counter before the reservation (lane 0)
};
__device__ __forceinline__ PathTicket wave_reserve_paths(bool take_path, unsigned long long* counter) {
  PathTicket tk;
  tk.mask = __ballot(take_path);
  tk.old_lo = 0;
  if (tk.mask != 0) {
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    // (the path count is the counter's low field, so a 32-bit add to its low word is the same addition; its
    // answer is one register, where the unused high word of a 64-bit answer would be handed to the code that
    // follows as a free register, and the first write to it would wait for the atomic)
    if (lane == 0) tk.old_lo = atomicAdd(reinterpret_cast<uint32_t*>(counter), (uint32_t)__popcll(tk.mask));
  }
  return tk;
}
__device__ __forceinline__ uint32_t wave_path_slot(const PathTicket& tk) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)tk.old_lo);
  return (lo & (uint32_t)kFieldMask) +
         __builtin_amdgcn_mbcnt_hi((uint32_t)(tk.mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)tk.mask, 0u));
}

// Block-level append: one atomic per block reserves the slots of all its waves, so the
// survivors of a block land in one run (fewer runs, fewer shared lines).  Every thread of
// the block must call it (two barriers).  Same outputs as wave_append.
__device__ __forceinline__ void block_append(bool alive, bool want_shadow, bool did_nee, unsigned long long* counter,
                                             uint32_t& pos, uint32_t& spos) {
  __shared__ uint32_t wave_counts[16][3];
  __shared__ unsigned long long block_old;
  const unsigned long long mp = __ballot(alive), ms = __ballot(want_shadow), mn = __ballot(did_nee);
  const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  if (lane == 0) {
    wave_counts[w][0] = (uint32_t)__popcll(mp);
    wave_counts[w][1] = (uint32_t)__popcll(ms);
    wave_counts[w][2] = (uint32_t)__popcll(mn);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t a = 0, b = 0, c = 0;
    for (uint32_t k = 0; k < (blockDim.x >> 6); k++) {
      a += wave_counts[k][0];
      b += wave_counts[k][1];
      c += wave_counts[k][2];
    }
    const unsigned long long add = (unsigned long long)a | ((unsigned long long)b << kPathBits) |
                                   ((unsigned long long)c << (kPathBits + kShadowBits));
    block_old = add ? atomicAdd(counter, add) : 0ull;
  }
  __syncthreads();
  const unsigned long long old = block_old;
  pos = seg_paths(old);
  spos = seg_shadow(old);
  for (uint32_t k = 0; k < w; k++) {
    pos += wave_counts[k][0];
    spos += wave_counts[k][1];
  }
  pos += __builtin_amdgcn_mbcnt_hi((uint32_t)(mp >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mp, 0u));
  spos += __builtin_amdgcn_mbcnt_hi((uint32_t)(ms >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ms, 0u));
}

// ----------------------------------------------------------------------------
// Kernels.

__global__ void k_begin(SegCounter* seg, Summary* sum, uint32_t nseg, unsigned long long total_samples,
                        Primary pr, Primary* pr_out) {
  uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < nseg) {
    seg[s].c[0] = 0;
    seg[s].c[1] = 0;
    seg[s].gen_slot0 = 0;
    seg[s].gen_count = 0;
    seg[s].gen_ordinal0 = 0;
    seg[s].grab = 0;
    seg[s].grab_shadow = 0;
    seg[s].more_bounces = 0;
  }
  if (s == 0) {
    *pr_out = pr;  // camera and sample numbering of this call, for the kernels that recompute primary rays
    sum->next_sample = 0;
    sum->total_samples = total_samples;
    sum->stat_paths = sum->stat_bounces = sum->stat_nee = sum->stat_shadow = 0;
    sum->stat_stored = 0;
    sum->active_total = 0;
    sum->active_max = 0;
    sum->stat_nan = 0;
    sum->stat_truncated = 0;
  }
}

// One block, one thread per segment (nseg <= 1024).  Runs after k_shade/k_shadow
// have filled queue `qf` from queue `qd`: decides how many new samples each
// segment of qf takes (segments are topped up in order until the samples run
// out), publishes the new fills, clears qd's counters and updates the summary.
__global__ __launch_bounds__(1024) void k_plan(SegCounter* seg, Summary* sum, int qf, int qd,
                                               uint32_t nseg, uint32_t cap) {
  // blockDim.x is a multiple of 64 that covers nseg.  Scans and sums go through the lanes of a
  // wave with shuffles and across the (at most 16) waves through LDS: three barriers in all.
  __shared__ unsigned long long wave_room[16];
  __shared__ uint32_t wave_part[16][6];
  const uint32_t s = threadIdx.x, lane = s & 63u, wave = s >> 6, nwaves = blockDim.x >> 6;
  uint32_t n = 0, sh = 0, ne = 0, drained = 0, room = 0;
  // Everything this block reads is requested here, ahead of the first barrier, so that the loads share one
  // round trip (the kernel is a chain of dependent steps between two rounds with the whole chip waiting on it).
  const unsigned long long next = sum->next_sample, total = sum->total_samples;
  unsigned long long stat_paths = 0, stat_bounces = 0, stat_shadow = 0, stat_nee = 0, stat_stored = 0;
  if (s == 0) {
    stat_stored = sum->stat_stored;
    stat_paths = sum->stat_paths;
    stat_bounces = sum->stat_bounces;
    stat_shadow = sum->stat_shadow;
    stat_nee = sum->stat_nee;
  }
  if (s < nseg) {
    unsigned long long c = seg[s].c[qf];
    n = seg_paths(c);
    sh = seg_shadow(c);
    ne = seg_nee(c);
    room = cap - n;
    if (qd >= 0) drained = seg_paths(seg[s].c[qd]) + seg[s].more_bounces;
  }
  // inclusive scan of the room within the wave
  unsigned long long incl = room;
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned long long up = __shfl_up(incl, off, 64);
    if ((int)lane >= off) incl += up;
  }
  if (lane == 63) wave_room[wave] = incl;
  __syncthreads();
  unsigned long long before_wave = 0, room_total = 0;
  for (uint32_t w = 0; w < nwaves; w++) {
    if (w < wave) before_wave += wave_room[w];
    room_total += wave_room[w];
  }
  const unsigned long long left = total - next;
  const unsigned long long before_room = before_wave + incl - room;
  const unsigned long long before = before_room < left ? before_room : left;
  const unsigned long long can = left - before;
  const uint32_t g = (uint32_t)(can < room ? can : room);
  if (s < nseg) {
    seg[s].gen_slot0 = n;
    seg[s].gen_count = g;
    seg[s].gen_ordinal0 = next + before;
    seg[s].c[qf] = (unsigned long long)(n + g);
    seg[s].grab = 0;
    seg[s].grab_shadow = 0;
    seg[s].more_bounces = 0;
    if (qd >= 0) seg[s].c[qd] = 0;
  }
  // sums of g, shadow rays, light samples, drained paths and the paths the round stored; maximum of n + g
  uint32_t v0 = g, v1 = n + g, v2 = sh, v3 = ne, v4 = drained, v5 = n;
  for (int off = 32; off > 0; off >>= 1) {
    v0 += __shfl_down(v0, off, 64);
    const uint32_t o1 = __shfl_down(v1, off, 64);
    v1 = v1 > o1 ? v1 : o1;
    v2 += __shfl_down(v2, off, 64);
    v3 += __shfl_down(v3, off, 64);
    v4 += __shfl_down(v4, off, 64);
    v5 += __shfl_down(v5, off, 64);
  }
  if (lane == 0) {
    wave_part[wave][5] = v5;
    wave_part[wave][0] = v0;
    wave_part[wave][1] = v1;
    wave_part[wave][2] = v2;
    wave_part[wave][3] = v3;
    wave_part[wave][4] = v4;
  }
  __syncthreads();
  if (s == 0) {
    unsigned long long gen_total = 0, shadow_total = 0, nee_total = 0, drained_total = 0, stored_total = 0;
    uint32_t fill_max = 0;
    for (uint32_t w = 0; w < nwaves; w++) {
      gen_total += wave_part[w][0];
      fill_max = fill_max > wave_part[w][1] ? fill_max : wave_part[w][1];
      shadow_total += wave_part[w][2];
      nee_total += wave_part[w][3];
      drained_total += wave_part[w][4];
      stored_total += wave_part[w][5];
    }
    sum->next_sample = next + gen_total;
    sum->stat_paths = stat_paths + gen_total;
    sum->stat_bounces = stat_bounces + drained_total;
    sum->stat_shadow = stat_shadow + shadow_total;
    sum->stat_nee = stat_nee + nee_total;
    sum->stat_stored = stat_stored + stored_total;
    sum->active_total = (uint32_t)((unsigned long long)nseg * cap - room_total + gen_total);
    sum->active_max = fill_max;
  }
}

// Writes the primary rays of the samples k_plan assigned to each segment (used when
// the kernels do not recompute them, see Primary).
__global__ __launch_bounds__(256) void k_generate(Queue q, const SegCounter* seg, CameraDev cam,
                                                  RenderParams rp, int use_depth) {
  const uint32_t s = blockIdx.y;
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= seg[s].gen_count) return;
  const uint32_t ordinal = (uint32_t)(seg[s].gen_ordinal0 + j);
  f3 o, d;
  Rng rng;
  primary_ray(cam, rp, ordinal, o, d, rng, rp.rng_streams != 0);
  // A fresh path: only the ray, the sample ordinal and the generator are stored
  // (see load_path).
  const uint32_t slot = s * rp.cap + seg[s].gen_slot0 + j;
  q.r0[slot] = make_float4(o.x, o.y, o.z, d.x);
  q.r1[slot] = make_float2(d.y, d.z);
  q.s2[slot] = make_uint2(ordinal, rng.s);
  if (use_depth) q.depth[slot] = 0;
  if (rp.rng_streams) q.inc[slot] = rng.inc;
}

// Dynamic LDS of the traversal kernels: the per-lane stacks (lds_entries x BLOCK
// ints), then, in kLdsBvh2 mode, the staged BVH and triangles.  `tp` carries the
// stack geometry; the spill column of a thread is indexed by its position in the launch.
struct TraceParams {
  int lds_entries;        // stack entries kept in LDS
  int* spill;             // deeper entries: spill[(entry - lds_entries) * spill_stride + thread]
  uint32_t spill_stride;  // >= threads per launch
};
#define RTX_TRACE_PROLOGUE()                                                                  \
  extern __shared__ float4 lds_dyn[];                                                         \
  Stack stk;                                                                                  \
  stk.lds = (lds_int_ptr)(reinterpret_cast<int*>(lds_dyn) + threadIdx.x);                                    \
  stk.stride = BLOCK;                                                                         \
  stk.lds_entries = tp.lds_entries;                                                           \
  stk.all_in_lds = MODE != kGlobalBvh4;                                                       \
  stk.spill_stride = tp.spill_stride;                                                         \
  stk.spill = (global_int_ptr)(tp.spill + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * BLOCK + threadIdx.x)); \
  const Geom geom = MODE == kLdsBvh2 ? stage_geom_lds(sc, lds_dyn + (tp.lds_entries * BLOCK) / 4, \
                                                      threadIdx.x, BLOCK)                     \
                                     : global_geom(sc)

// Which (segment, block within the segment) this workgroup serves.  Workgroups are
// handed to the eight XCDs round-robin by linear id, and each XCD has its own L2:
// the mapping gives every block of segment s to XCD s mod 8.  A segment holds the
// paths of a compact run of pixels, bounce after bounce, so its rays meet the same
// part of the tree and one L2 keeps that part, instead of eight L2s each keeping
// the same mixture.
struct SegBlock {
  uint32_t s, x;
};
__device__ __forceinline__ SegBlock seg_block() {
  const uint32_t nseg = gridDim.y, gx = gridDim.x;
  if (nseg & 7u) return {blockIdx.y, blockIdx.x};
  const uint32_t id = blockIdx.y * gx + blockIdx.x;
  const uint32_t xcd = id & 7u, k = id >> 3, per = nseg >> 3;
  return {(k % per) * 8u + xcd, k / per};
}

}  // namespace rtx
