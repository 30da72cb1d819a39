// kernels.h -- the wavefront path tracer's device code (gfx950, wave64).
//
// One kernel per stage over SoA queues in HBM, as the reference's own TODO asks
// ("small kernels over a singular megakernel", "all the attributes of a certain
// type are arrays", reference README.md:40-42):
//
//   k_plan       queue bookkeeping: how many new samples each segment takes
//   k_generate   primary rays for new samples         (Raytracer.cpp:246-250, Camera.h)
//   k_intersect  closest hit: tree + analytic objects (Scene.h:144-189); two-child tree
//                staged in LDS for small scenes, k_intersect_wide on the four-child
//                quantised tree in global memory for large ones
//   k_shade_*    emission/MIS, Russian roulette, BRDF sampling, NEE set-up,
//                compaction of the survivors           (Integrator.h:98-291);
//                one append per wave or per block
//   k_shadow     any-hit for the NEE segments          (Scene.h:191-225)
//   k_finish     runs the last few paths to their end, one path per lane
//   k_reduce     adds the per-sample results to the per-pixel sums, in sample order
//
// Path state lives in five SoA arrays (64 bytes per path) that every kernel
// reads and writes with 8- or 16-byte coalesced accesses:
//   r0 float4 (o.x, o.y, o.z, d.x)     r1 float2 (d.y, d.z)
//   s0 float4 (attenuation.rgb, mis_brdf_pdf)
//   s1 float4 (colour.rgb, nee_pdf)    s2 uint2  (sample ordinal, rng state)
// There are two such queues; k_shade reads one and appends the surviving paths
// to the other (ballot + mbcnt prefix, one atomic per wave or block).
//
// Each queue is cut into `nseg` segments of `cap` slots with a counter of their
// own (one per 128-byte line): a wave appends to the segment it read from, so a
// launch spreads its appends over nseg counters instead of serialising ~10 ns
// apiece on one address (measured: 65536 wave appends take 750 us on one
// counter, 24 us on 64, 12 us on 512).  A segment can never overflow: it
// receives at most the paths it held.
//
// A finished path stores its colour to the slot of its sample ordinal in a
// per-sample result buffer (plain 12-byte store, no atomics; float atomics at
// scattered pixels cost ~10x a plain store here).  k_reduce then adds the
// samples of every pixel in sample order, which makes the image bit-reproducible
// and gives it the summation order of a sequential loop over the samples.
//
// Where to find what (one header per stage):
//   scene_types.h            SceneDev, CameraDev, Queue, SegCounter, Summary, Stack, the LDS copy of a small tree
//   traversal.h              intersection tests, tree walks, cast_ray / occluded / hit_surface
//   shading.h                primary_ray, path state, shade_arrive / shade_leave (one bounce of pathTrace)
//   queues.h                 wave_append / block_append, k_begin, k_plan, k_generate
//   kernels_small_scenes.h   k_trace_shade, k_intersect, k_shadow, k_shadow_leaf_tasks
//   kernels_wide_tree.h      k_intersect_wide, k_shadow_wide, k_shade_wave_append, k_shade_block_append
//   kernels_finish_reduce.h  k_finish, k_reduce*, k_mark_unfinished, k_resolve
//   kernels_debug.h          k_trace_closest, k_trace_occluded, k_debug_pcg, k_debug_streams
//
// Which kernels a frame runs (rtx_hip.hip, render_chunk):
//   mesh of up to 8 triangles, no lights (sample1, sample2)   k_plan, k_trace_shade<64 or 128, kFewTriangles, .., BOUNCES = 2>
//   tree in LDS, a room of a few surfaces (MIS Cornell)        k_plan, k_trace_shade<256, kLdsLeafTasks, true, true>
//   tree in LDS otherwise (up to about 200 triangles)          k_plan, k_trace_shade<.., kLdsBvh2>, k_shadow where lit
//   large mesh (four-child tree in global memory)              k_plan, k_generate, k_intersect_wide,
//                                                              k_shade_block_append, k_shadow_wide
//   every frame ends with k_finish and k_reduce (k_mark_unfinished, k_reduce_head, k_reduce_tail for large chunks).
#pragma once
#include "scene_types.h"
#include "traversal.h"
#include "shading.h"
#include "queues.h"
#include "kernels_small_scenes.h"
#include "kernels_wide_tree.h"
#include "kernels_finish_reduce.h"
#include "kernels_debug.h"
