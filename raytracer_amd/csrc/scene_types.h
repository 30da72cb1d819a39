// scene_types.h -- what the host hands the kernels: the scene's arrays, the camera, the path queues and their
// per-segment counters, the traversal stacks, and the copy of a small tree into LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_math.h"

namespace rtx {

constexpr uint32_t kMiss = 0xffffffffu;
constexpr uint32_t kSphereBit = 0x80000000u;
// Traversal cursor value for "nothing left".  It cannot collide with a leaf
// reference: ~0x80000000 would be a leaf of 8 triangles at slot 2^28-1, and the
// builder makes leaves of at most 2.
constexpr int kTraversalDone = (int)0x80000000u;

struct SceneDev {
  const float4* nodes;   // 4 per BVH node (bvh_build.h)
  const float4* tris;    // per triangle slot, one 64-byte record: (a.xyz, primitive id bits),
                         // (b-a, material index bits), (c-a, flags), ((b-a)x(c-a), 0).  Wide tree: leaf records
                         // instead, densely in leaf order (a leaf word counts in records): the same record with the
                         // triangle's slot in row 2's last word, or one record for the two halves of a quad (pair_intersect)
  const uint32_t* slot_prim;  // per triangle slot: the primitive id (the index the caller gave the triangle)
  const float4* tri_nrm; // per triangle slot: (normalize((b-a)x(c-a)), material index bits): what shading needs of
                         // a hit, computed once on the host with the float operations hit_surface would use
  const float4* mat0;    // per material: (albedo.rgb, emission)
  const int* mat1;       //               surface
  const float4* tri_mat; // small scenes only (else null): the material copied out per triangle slot, (albedo.rgb,
  const int* tri_surf;   // emission) and surface, so that a hit's loads do not wait for its material index
  int few_surfaces;      // 1: a handful of spheres and triangles and nothing else (hit_surface reads them all)
  const float4* leaf_box;  // trees of up to kLeafTaskMaxLeaves leaves (else null): per leaf (lo.x, hi.x, lo.y, hi.y),
  uint32_t num_leaf_boxes; //   (lo.z, hi.z, leaf word bits, 0): the leaves' boxes as the two-child nodes hold them
  uint32_t leaf_task_cap;  // (ray, leaf) pairs the block's queue holds per thread (block_closest_by_leaf_tasks)
  // Analytic objects, in the order spheres, boxes, planes; object i of that
  // combined list has material (sph1[i], sph2[i]).
  const float4* sph0;    // per sphere: (c.xyz, r)
  const float4* sph1;    // per analytic object: (albedo.rgb, emission)
  const int* sph2;       //                      surface
  const int* sph3;       // per sphere: 1 = no triangle comes near its ball (LastSphere::clear)
  const float4* box0;    // per box: (min.xyz, 0)
  const float4* box1;    //          (max.xyz, 0)
  const float4* pln0;    // per plane: (point.xyz, 0)
  const float4* pln1;    //            (normal.xyz, 0)
  const float4* light0;  // per light: (a.xyz, rcp_area)
  const float4* light1;  //            (b.xyz, emission)
  const float4* light2;  //            (c.xyz, 0)
  const float4* light3;  //            (n.xyz, 0)
  uint32_t num_nodes, num_tris, num_spheres, num_boxes, num_planes, num_lights;
  float eps, nee_threshold, rr_cap;
  int sky;
  uint32_t max_bounces;
  int integrator;
  int rng_streams;     // rtx_config.rng_streams: every sample on a generator cycle of its own
  int carries_colour;  // 0: nothing in the scene emits and there are no lights, so a path's colour and
                       // nee_pdf stay exactly 0 until it ends; the queues then skip them (s1)
};

struct CameraDev {
  uint32_t width, height;
  float tan_x, tan_y;  // tanf of the half angles, computed on the host (Camera.h:19-20)
  float ox, oy, oz;    // camera origin; the caller's offset `pos` is added separately
  float offx, offy, offz;
  int type;
  float focal_length, lens_radius, stretch_x, stretch_y;
};

struct Queue {
  float4* r0;
  float2* r1;
  float4* s0;
  float4* s1;
  uint2* s2;
  uint32_t* depth;  // bounces so far; touched only when max_bounces != 0
  uint32_t* inc;    // the generator's increment; touched only with rng_streams
};

struct ShadowQueue {
  float4* q0;  // (o.xyz, tfar = distance - eps)
  float4* q1;  // (d.xyz, destination slot bits)
  float4* q2;  // (contribution.rgb, 0): read only for segments found unoccluded
};

// Per-segment counters, one 128-byte line per segment.  c[q] packs the fill of
// queue q's segment: bits 0-20 paths, 21-41 shadow rays, 42-63 light samples
// taken this round, so that one 64-bit atomic per wave serves all three.
struct SegCounter {
  unsigned long long c[2];
  uint32_t gen_slot0, gen_count;      // k_plan's refill plan for k_generate
  unsigned long long gen_ordinal0;
  uint32_t grab;                      // rays of the segment handed out so far (k_intersect_wide); k_plan clears it
  uint32_t grab_shadow;               // the same for its light samples (k_shadow_wide)
  uint32_t more_bounces;              // bounces a launch made beyond the first of each of its paths (k_trace_shade, BOUNCES > 1)
  uint32_t pad32;
  unsigned long long pad[10];
};
static_assert(sizeof(SegCounter) == 128, "one line per segment");
constexpr int kPathBits = 21, kShadowBits = 21;
constexpr unsigned long long kFieldMask = (1ull << 21) - 1;
__host__ __device__ __forceinline__ uint32_t seg_paths(unsigned long long c) { return (uint32_t)(c & kFieldMask); }
__host__ __device__ __forceinline__ uint32_t seg_shadow(unsigned long long c) { return (uint32_t)((c >> kPathBits) & kFieldMask); }
__host__ __device__ __forceinline__ uint32_t seg_nee(unsigned long long c) { return (uint32_t)(c >> (kPathBits + kShadowBits)); }

// Whole-queue state, written by k_plan (one block) and snapshotted by the host.
struct Summary {
  unsigned long long next_sample;    // ordinal of the next sample to generate
  unsigned long long total_samples;  // band_pixels * spp of this chunk
  unsigned long long stat_paths, stat_bounces, stat_nee, stat_shadow;
  unsigned long long stat_stored;  // paths a bounce wrote to a queue for the next launch to read back
  uint32_t active_total;  // paths in the queue about to be traced
  uint32_t active_max;    // the fullest segment
  uint32_t stat_nan;        // finished paths whose colour has a NaN or an infinity in it
  uint32_t stat_truncated;  // paths k_finish gave up on at kFinishCap bounces
};

struct RenderParams {
  uint32_t nseg, cap;    // segments per queue, slots per segment
  uint32_t row0;         // first row of the band
  uint32_t band_pixels;  // (row1-row0)*width
  uint32_t sample_base;  // sample index of ordinal 0
  uint32_t seed32;       // seed folded to 32 bits
  unsigned long long seed64;  // the seed as given (rng_streams)
  int rng_streams;
};

#ifndef RTX_FULL_RECORD_READS
#define RTX_FULL_RECORD_READS 1
#endif
// Where the traversal kernels read the BVH and the triangles from: the global
// arrays of SceneDev, or a copy of them in LDS (scenes of a few hundred
// triangles: every lane's node and triangle fetches then cost an LDS read
// instead of an L1/L2 round trip, and stop competing for the L1's 64 B/clk).
struct Geom {
  const float4* nodes;
  const float4* tris;  // four float4 per triangle slot
  uint32_t stride;     // float4 between consecutive records: 4 in global memory; kLdsRecordStride in the LDS copy
};
// Records in the LDS copy are 80 bytes apart, not 64: a 16-byte read of record i then starts at
// bank 20 i mod 64, sixteen different places for sixteen consecutive records, where 64-byte
// records offer four (lanes that read different records collide on the banks otherwise).
#ifndef RTX_LDS_RECORD_STRIDE
#define RTX_LDS_RECORD_STRIDE 5
#endif
constexpr uint32_t kLdsRecordStride = RTX_LDS_RECORD_STRIDE;

// How a traversal kernel reads the tree.
enum TraceMode {
  kGlobalBvh2 = 0,  // two-child nodes in global memory
  kLdsBvh2 = 1,     // two-child nodes and triangles staged in LDS (small scenes)
  kGlobalBvh4 = 2,  // four-child quantised nodes in global memory (large scenes)
  kFewTriangles = 3,  // no tree: every ray tests every triangle (meshes of up to kFewTrianglesMax)
  kLdsLeafTasks = 4,  // bounce kernel of trees of up to kLeafTaskMaxLeaves leaves: block_closest_by_leaf_tasks
};
constexpr uint32_t kLeafTaskMaxLeaves = 32;
// A mesh this small (the ground quad of the reference's sphere scenes) is tested like the analytic
// objects: all lanes read the same records, which the compiler turns into scalar loads; no tree,
// no stack, and no copy of the scene into LDS with the barrier behind it.
constexpr uint32_t kFewTrianglesMax = 8;

// Per-lane traversal stack: the first lds_entries entries live in LDS (one column
// per lane, `stride` ints apart); deeper entries, which a well-built tree needs
// only for rare rays, go to a global array with one column per launched thread.
typedef __attribute__((address_space(3))) int* lds_int_ptr;
typedef __attribute__((address_space(1))) int* global_int_ptr;
struct Stack {
  lds_int_ptr lds;        // explicit address spaces: no flat-aperture tests in the walk
  global_int_ptr spill;
  int stride;          // LDS column stride (threads per block)
  int lds_entries;
  uint32_t spill_stride;  // threads per launch
  bool all_in_lds;        // two-child trees: the host sized the LDS part for the deepest path
  __device__ __forceinline__ void push(int& sp, int v) const {
    if (all_in_lds || sp < lds_entries)
      lds[sp * stride] = v;
    else
      spill[(size_t)(sp - lds_entries) * spill_stride] = v;
    sp++;
  }
  __device__ __forceinline__ int pop(int& sp) const {
    sp--;
    return (all_in_lds || sp < lds_entries) ? lds[sp * stride] : spill[(size_t)(sp - lds_entries) * spill_stride];
  }
};
__device__ __forceinline__ Geom global_geom(const SceneDev& sc) {
  Geom g;
  g.nodes = sc.nodes;
  g.tris = sc.tris;
  g.stride = 4;
  return g;
}
// Copies [nodes | triangle records] into `dst` (LDS) with the whole block and
// returns the view; ends with a barrier, so every thread of the block must call it.
__device__ __forceinline__ Geom stage_geom_lds(const SceneDev& sc, float4* dst, int tid, int nthreads) {
  const uint32_t nn = 4 * sc.num_nodes, nt = 4 * sc.num_tris;
  float4* const tdst = dst + kLdsRecordStride * sc.num_nodes;
  for (uint32_t k = tid; k < nn; k += nthreads) dst[(k >> 2) * kLdsRecordStride + (k & 3u)] = sc.nodes[k];
  for (uint32_t k = tid; k < nt; k += nthreads) tdst[(k >> 2) * kLdsRecordStride + (k & 3u)] = sc.tris[k];
  __syncthreads();
  Geom g;
  g.nodes = dst;
  g.tris = tdst;
  g.stride = kLdsRecordStride;
  return g;
}
// The same for the kernels that test every leaf's box and never walk (kLdsLeafTasks): the list of leaf boxes, two
// float4 each, takes the place of the node records (2 n float4 against the 5 (n - 1) of a tree of n leaves; the
// host checks that it fits), so the triangle records start where they start in the other layout.  The lanes then
// read a box with two LDS reads of one address (a broadcast), which a wave can have several of in flight; read
// through the scalar cache, one box at a time for want of scalar registers, the box loop ran at a third of the
// rate its arithmetic allows.  Geom::nodes is the box list.
__device__ __forceinline__ Geom stage_leaf_task_geom_lds(const SceneDev& sc, float4* dst, int tid, int nthreads) {
  const uint32_t nb = 2 * sc.num_leaf_boxes, nt = 4 * sc.num_tris;
  float4* const tdst = dst + kLdsRecordStride * sc.num_nodes;
  for (uint32_t k = tid; k < nb; k += nthreads) dst[k] = sc.leaf_box[k];
  for (uint32_t k = tid; k < nt; k += nthreads) tdst[(k >> 2) * kLdsRecordStride + (k & 3u)] = sc.tris[k];
  __syncthreads();
  Geom g;
  g.nodes = dst;
  g.tris = tdst;
  g.stride = kLdsRecordStride;
  return g;
}

}  // namespace rtx
