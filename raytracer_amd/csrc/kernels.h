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
                         // (b-a, material index bits), (c-a, 0), ((b-a)x(c-a), 0)
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

// ----------------------------------------------------------------------------
// Geometry tests.

// Triangle::intersect, reference Obj.h:110-121, on a triangle stored as
// (a, b-a, c-a).  The comparisons keep the reference's form so that infinities
// and NaNs from a ray parallel to the plane fall through identically.
__device__ __forceinline__ float tri_intersect_n(f3 a, f3 ba, f3 ca, f3 n, f3 o, f3 d) {
  f3 roa = vsub(o, a);
  f3 q = vcross(roa, d);
  float inv = 1.0f / vdot(d, n);
  float u = inv * -vdot(q, ca);
  float v = inv * vdot(q, ba);
  if (u < 0.0f || u > 1.0f || v < 0.0f || (u + v) > 1.0f) return INFINITY;
  return inv * -vdot(n, roa);
}
__device__ __forceinline__ float tri_intersect(f3 a, f3 ba, f3 ca, f3 o, f3 d) {
  return tri_intersect_n(a, ba, ca, vcross(ba, ca), o, d);
}
// What consecutive triangles of a leaf can share.  The two triangles of a quad (a, b, c), (a, c, d) have the
// same first vertex, so o - a and (o - a) x d are the same floats for both; where their stored normals are
// equal as well (any axis-aligned rectangle), so are 1/(d.n) and t.  The host compares the records and sets
// the flags in a record's third row (kSameVertex, kSamePlane: relative to the record before it); the lanes
// then skip the operations whose results they still hold, and every hit keeps the bits a full test gives.
struct TriCarry {
  f3 roa, q;
  float inv, t;
};
constexpr uint32_t kSameVertex = 1u;  // first vertex equals the previous record's
constexpr uint32_t kSamePlane = 2u;   // and so does the normal (zero components may differ in sign)
// Triangle `slot` against a ray: Triangle::intersect, reference Obj.h:110-121, as tri_intersect_n.  The record
// carries the normal (b-a)x(c-a), computed on the host with the same float operations (no contraction on
// either side), so the lanes do not recompute it; it shares the record's 64-byte line.  `reuse`: the lane
// tested the record before this one with the same ray and `s` still holds what that test left.
template <bool LDS_COPY = true>
__device__ __forceinline__ float tri_test(const Geom& g, uint32_t slot, bool reuse, f3 o, f3 d, TriCarry& s,
                                          uint32_t& prim) {
  const float4* r = g.tris + g.stride * (size_t)slot;
  float4 A = r[0], B = r[1], C = r[2], N = r[3];
#if RTX_FULL_RECORD_READS
  if constexpr (LDS_COPY)
  // keeps the reads 16 bytes wide: the unused fourth components would otherwise turn them into
  // 12-byte LDS reads, which take twice the LDS cycles of a 16-byte one
  asm volatile("" ::"v"(B.w), "v"(N.w));
#endif
  prim = __float_as_uint(A.w);
  const uint32_t shared = reuse ? __float_as_uint(C.w) : 0u;
  const f3 ba = mk3(B.x, B.y, B.z), ca = mk3(C.x, C.y, C.z), n = mk3(N.x, N.y, N.z);
  if (!(shared & kSameVertex)) {
    s.roa = vsub(o, mk3(A.x, A.y, A.z));
    s.q = vcross(s.roa, d);
  }
  if (!(shared & kSamePlane)) {
    s.inv = 1.0f / vdot(d, n);
    s.t = s.inv * -vdot(n, s.roa);
  }
  float u = s.inv * -vdot(s.q, ca);
  float v = s.inv * vdot(s.q, ba);
  if (u < 0.0f || u > 1.0f || v < 0.0f || (u + v) > 1.0f) return INFINITY;
  return s.t;
}

// Sphere::intersect, reference Obj.h:257-268.
__device__ __forceinline__ float sphere_intersect(f3 c, float r, f3 o, f3 d, float eps) {
  f3 oc = vsub(o, c);
  float B = 2.0f * vdot(d, oc);
  float C = vsqrnorm(oc) - r * r;
  float disc2 = B * B - 4.0f * C;
  if (disc2 < 0.0f) return INFINITY;
  float disc = sqrtf(disc2);
  float t1 = (-B - disc) / 2.0f;
  return t1 > eps ? t1 : (-B + disc) / 2.0f;
}

// Box::intersect, the axis-aligned box of the reference's commented-out class
// (Obj.h:43-68): per-axis divisions, std::min/std::max folds,
// tmax >= tmin ? (tmin > EPS ? tmin : tmax) : INFINITY.
__device__ __forceinline__ float box_intersect(f3 mn, f3 mx, f3 o, f3 d, float eps) {
  float t1 = (mn.x - o.x) / d.x;
  float t2 = (mx.x - o.x) / d.x;
  float tmin = std_min(t1, t2);
  float tmax = std_max(t1, t2);
  t1 = (mn.y - o.y) / d.y;
  t2 = (mx.y - o.y) / d.y;
  tmin = std_max(tmin, std_min(t1, t2));
  tmax = std_min(tmax, std_max(t1, t2));
  t1 = (mn.z - o.z) / d.z;
  t2 = (mx.z - o.z) / d.z;
  tmin = std_max(tmin, std_min(t1, t2));
  tmax = std_min(tmax, std_max(t1, t2));
  return tmax >= tmin ? (tmin > eps ? tmin : tmax) : INFINITY;
}
// Box::normal (Obj.h:70-87): the face whose axis has the largest |p - centre|
// relative to the box's extent on that axis; sgn() of Obj.h:21-23.
__device__ __forceinline__ float sgnf(float x) { return (float)((x > 0) - (x < 0)); }
__device__ __forceinline__ f3 box_normal(f3 mn, f3 mx, f3 p) {
  f3 c = vdivs(vadd(mx, mn), 2.0f);
  f3 v = vsub(p, c);
  const float cv[3] = {v.x, v.y, v.z};
  const float ext[3] = {mx.x - mn.x, mx.y - mn.y, mx.z - mn.z};
  int am = 0;
  for (int i = 1; i < 3; i++)
    if (fabsf(cv[i]) * ext[am] > fabsf(cv[am]) * ext[i]) am = i;
  return mk3((float)(am == 0) * sgnf(v.x), (float)(am == 1) * sgnf(v.y), (float)(am == 2) * sgnf(v.z));
}
// Plane(point, normal): the reference only has call sites (Raytracer.cpp:184-221), no
// class; this is the build's definition: t = ((point - o) . n) / (d . n), normal = n as given.
__device__ __forceinline__ float plane_intersect(f3 pt, f3 n, f3 o, f3 d) {
  return vdot(vsub(pt, o), n) / vdot(d, n);
}

// Slab test of one child box against [0, tmax] in the form t = plane*inv + add,
// one fma per plane.  `add` is -(o*inv), whose rounding error acts like a shift of
// the plane by up to |o| 2^-24; walk_begin moves the two addends of an axis apart
// by four times that, towards "box larger", so the test stays conservative for any
// origin.  The comparison's slack covers the relative errors (fma rounding and the
// reciprocal's ulp).  The inverse is finite (clamped), so no term is NaN.
__device__ __forceinline__ bool slab(float lox, float hix, float loy, float hiy, float loz, float hiz,
                                     f3 inv, f3 add_lo, f3 add_hi, float tmin, float tmax, float& tnear) {
  float ax = __builtin_fmaf(lox, inv.x, add_lo.x), bx = __builtin_fmaf(hix, inv.x, add_hi.x);
  float ay = __builtin_fmaf(loy, inv.y, add_lo.y), by = __builtin_fmaf(hiy, inv.y, add_hi.y);
  float az = __builtin_fmaf(loz, inv.z, add_lo.z), bz = __builtin_fmaf(hiz, inv.z, add_hi.z);
  float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), tmin));
  float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tmax));
  tnear = tn;
  return tn <= tf * 1.000001f;
}

// One triangle of a leaf against the running closest hit, with the rules of
// Mesh::intersect (reference Obj.h:184-195): a hit needs eps < t; the smallest t
// wins and, among equal t, the lowest primitive id (the linear scan keeps the first).
template <int MODE>
__device__ __forceinline__ void leaf_closest(const Geom& g, uint32_t leaf, f3 o, f3 d, float eps,
                                             float& best_t, uint32_t& best_prim, uint32_t& best_slot) {
  uint32_t first = leaf >> 3, cnt = (leaf & 7u) + 1u;
  TriCarry carry;
  for (uint32_t k = 0; k < cnt; k++) {
    uint32_t slot = first + k;
    uint32_t prim;
    float t = tri_test<MODE != kFewTriangles>(g, slot, k != 0, o, d, carry, prim);
    if (eps < t && t < INFINITY && (t < best_t || (t == best_t && prim < best_prim))) {
      best_t = t;
      best_prim = prim;
      best_slot = slot;
    }
  }
}
// Any triangle of a leaf with eps < t < tfar (the interval Scene::isOccluded gives
// Embree, reference Scene.h:216-217, ends taken open as in Mesh::intersect).
template <int MODE>
__device__ __forceinline__ bool leaf_any(const Geom& g, uint32_t leaf, f3 o, f3 d, float eps, float tfar) {
  uint32_t first = leaf >> 3, cnt = (leaf & 7u) + 1u;
  bool hit = false;
  TriCarry carry;
  for (uint32_t k = 0; k < cnt; k++) {
    uint32_t slot = first + k;
    uint32_t prim;
    float t = tri_test<MODE != kFewTriangles>(g, slot, k != 0, o, d, carry, prim);
    if (eps < t && t < tfar) hit = true;
  }
  return hit;
}

// The four children of a quantised node against [0, tmax].  a_k = scale_k/d_k and
// b_k = (origin_k - o_k)/d_k turn a plane byte q into t = q*a_k + b_k (one fma; the
// boxes are padded and quantised outwards by far more than its rounding).  The near
// and far plane of each axis are picked by the direction's sign, so an inverted box
// (an absent child: lo = 255, hi = 0) always misses.  |d_k| is clamped away from zero
// for this test only, which keeps every term finite.
struct Node4Test {
  float tn[4];
  bool hit[4];
};
__device__ __forceinline__ float ubyte_f(uint32_t w, int i) { return (float)((w >> (8 * i)) & 0xffu); }
__device__ __forceinline__ Node4Test test_node4(float4 q0, float4 q1f, float4 q2f, float4 q3f, f3 o, f3 inv,
                                                float tmin, float tmax, int4& refs) {
  const uint32_t ew = __float_as_uint(q0.w);
  const float ax = __uint_as_float((ew & 0xffu) << 23) * inv.x;
  const float ay = __uint_as_float(((ew >> 8) & 0xffu) << 23) * inv.y;
  const float az = __uint_as_float(((ew >> 16) & 0xffu) << 23) * inv.z;
  const float bx = (q0.x - o.x) * inv.x, by = (q0.y - o.y) * inv.y, bz = (q0.z - o.z) * inv.z;
  const uint32_t lox = __float_as_uint(q1f.x), loy = __float_as_uint(q1f.y), loz = __float_as_uint(q1f.z);
  const uint32_t hix = __float_as_uint(q1f.w), hiy = __float_as_uint(q2f.x), hiz = __float_as_uint(q2f.y);
  const uint32_t nx = inv.x < 0.0f ? hix : lox, fx = inv.x < 0.0f ? lox : hix;
  const uint32_t ny = inv.y < 0.0f ? hiy : loy, fy = inv.y < 0.0f ? loy : hiy;
  const uint32_t nz = inv.z < 0.0f ? hiz : loz, fz = inv.z < 0.0f ? loz : hiz;
  refs = make_int4(__float_as_int(q3f.x), __float_as_int(q3f.y), __float_as_int(q3f.z), __float_as_int(q3f.w));
  const int r[4] = {refs.x, refs.y, refs.z, refs.w};
  Node4Test out;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    float tnx = __builtin_fmaf(ubyte_f(nx, i), ax, bx), tfx = __builtin_fmaf(ubyte_f(fx, i), ax, bx);
    float tny = __builtin_fmaf(ubyte_f(ny, i), ay, by), tfy = __builtin_fmaf(ubyte_f(fy, i), ay, by);
    float tnz = __builtin_fmaf(ubyte_f(nz, i), az, bz), tfz = __builtin_fmaf(ubyte_f(fz, i), az, bz);
    float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, tmin));
    float tf = fminf(fminf(tfx, tfy), fminf(tfz, tmax));
    out.tn[i] = tn;
    out.hit[i] = tn <= tf * 1.0000004f && r[i] != kTraversalDone;
  }
  return out;
}
__device__ __forceinline__ f3 safe_inverse(f3 d) {
  const float tiny = 1.0e-20f;
  float x = fabsf(d.x) < tiny ? copysignf(tiny, d.x) : d.x;
  float y = fabsf(d.y) < tiny ? copysignf(tiny, d.y) : d.y;
  float z = fabsf(d.z) < tiny ? copysignf(tiny, d.z) : d.z;
  return mk3(1.0f / x, 1.0f / y, 1.0f / z);
}
// The same with the hardware reciprocal (one ulp): only for box tests whose
// comparison allows for it.
__device__ __forceinline__ f3 fast_safe_inverse(f3 d) {
  const float tiny = 1.0e-20f;
  float x = fabsf(d.x) < tiny ? copysignf(tiny, d.x) : d.x;
  float y = fabsf(d.y) < tiny ? copysignf(tiny, d.y) : d.y;
  float z = fabsf(d.z) < tiny ? copysignf(tiny, d.z) : d.z;
  return mk3(__builtin_amdgcn_rcpf(x), __builtin_amdgcn_rcpf(y), __builtin_amdgcn_rcpf(z));
}
__device__ __forceinline__ void order2(float& ta, int& ra, float& tb, int& rb) {
  // selects, not a branch around four moves: one comparison and four conditional moves, no mask bookkeeping
  const bool s = tb < ta;
  const float lo = s ? tb : ta, hi = s ? ta : tb;
  const int rl = s ? rb : ra, rh = s ? ra : rb;
  ta = lo;
  tb = hi;
  ra = rl;
  rb = rh;
}

// Traversal state of one ray.  `cur` is the node (>= 0) or leaf (< 0, ~leaf word)
// the lane stands on, kTraversalDone when the walk is over; `sp` its stack depth.
struct Walk {
  f3 o, d, inv;
  f3 add_lo, add_hi;  // two-child nodes only: the slab test's addends, see slab()
  float tmin;         // boxes the ray has left before this distance are skipped, see walk_begin
  int cur, sp;
};
template <int MODE>
__device__ __forceinline__ void walk_begin(const SceneDev& sc, f3 o, f3 d, Walk& w) {
  w.o = o;
  w.d = d;
  if (MODE == kGlobalBvh4) {
    w.inv = safe_inverse(d);
  } else {
    w.inv = fast_safe_inverse(d);
    const f3 nb = mk3(-(o.x * w.inv.x), -(o.y * w.inv.y), -(o.z * w.inv.z));
    const float k = 2.3841858e-7f;  // 2^-22
    const f3 e = mk3(copysignf(fabsf(nb.x) * k, w.inv.x), copysignf(fabsf(nb.y) * k, w.inv.y),
                     copysignf(fabsf(nb.z) * k, w.inv.z));
    w.add_lo = mk3(nb.x - e.x, nb.y - e.y, nb.z - e.z);
    w.add_hi = mk3(nb.x + e.x, nb.y + e.y, nb.z + e.z);
  }
  // A hit needs eps < t (Obj.h:184-195, Scene.h:216-217), so a box the ray leaves before eps holds none: above
  // all the box of the surface the ray starts on, whose two triangles every bounce and every light sample would
  // otherwise test for nothing.  Half of eps leaves room for the noise of a computed t: the boxes are padded by
  // 1e-5 of their diagonal, fifty times the rounding of n.(o-a) for a vertex that far away, so a triangle whose
  // computed t passes eps has its true crossing, and with it its box, beyond eps/2.
  w.tmin = 0.5f * sc.eps;
  w.cur = sc.num_nodes == 0 ? (int)kTraversalDone : 0;
  w.sp = 0;
}
struct ClosestHit {
  float t;
  uint32_t prim, slot;
};

// "while-while" traversal: all lanes of the wave first walk inner nodes until each
// holds a leaf (or is done), then all test triangles; the two phases do not mix
// within an iteration, which keeps more lanes busy in each.

// One two-child node of a closest-hit walk: the nearer child next, the other pushed.
__device__ __forceinline__ void closest_node_step(const Geom& g, const Stack& stk, Walk& w, const ClosestHit& best) {
  const float4* nd = g.nodes + g.stride * (size_t)w.cur;
  float4 q0 = nd[0], q1 = nd[1], q2 = nd[2], q3 = nd[3];
  float tn0, tn1;
  bool h0 = slab(q0.x, q0.y, q0.z, q0.w, q2.x, q2.y, w.inv, w.add_lo, w.add_hi, w.tmin, best.t, tn0);
  bool h1 = slab(q1.x, q1.y, q1.z, q1.w, q2.z, q2.w, w.inv, w.add_lo, w.add_hi, w.tmin, best.t, tn1);
  int c0 = __float_as_int(q3.x), c1 = __float_as_int(q3.y);
  if (h0 && h1) {
    bool swap = tn1 < tn0;
    stk.push(w.sp, swap ? c0 : c1);
    w.cur = swap ? c1 : c0;
  } else if (h0) {
    w.cur = c0;
  } else if (h1) {
    w.cur = c1;
  } else {
    w.cur = w.sp == 0 ? kTraversalDone : stk.pop(w.sp);
  }
}
// Inner nodes of a closest-hit walk: front to back, until the lane reaches a leaf.
template <int MODE>
__device__ __forceinline__ void closest_nodes(const Geom& g, const Stack& stk, Walk& w, const ClosestHit& best) {
  if (MODE == kGlobalBvh4) {
    while (w.cur >= 0) {
      int4 refs;
      const float4* nd = g.nodes + g.stride * (size_t)w.cur;
      Node4Test nt = test_node4(nd[0], nd[1], nd[2], nd[3], w.o, w.inv, w.tmin, best.t, refs);
      float t0 = nt.hit[0] ? nt.tn[0] : INFINITY, t1 = nt.hit[1] ? nt.tn[1] : INFINITY;
      float t2 = nt.hit[2] ? nt.tn[2] : INFINITY, t3 = nt.hit[3] ? nt.tn[3] : INFINITY;
      int r0 = refs.x, r1 = refs.y, r2 = refs.z, r3 = refs.w;
      // sort the four by entry distance (misses at +inf sink to the end)
      order2(t0, r0, t1, r1);
      order2(t2, r2, t3, r3);
      order2(t0, r0, t2, r2);
      order2(t1, r1, t3, r3);
      order2(t1, r1, t2, r2);
      if (t0 < INFINITY) {
        if (t3 < INFINITY) stk.push(w.sp, r3);
        if (t2 < INFINITY) stk.push(w.sp, r2);
        if (t1 < INFINITY) stk.push(w.sp, r1);
        w.cur = r0;
      } else {
        w.cur = w.sp == 0 ? kTraversalDone : stk.pop(w.sp);
      }
    }
  } else {
    while (w.cur >= 0) closest_node_step(g, stk, w, best);
  }
}
// Leaves of a closest-hit walk: until the lane pops an inner node or runs out.
template <int MODE>
__device__ __forceinline__ void closest_leaves(const Geom& g, const Stack& stk, float eps, Walk& w,
                                               ClosestHit& best) {
  while (w.cur < 0 && w.cur != kTraversalDone) {
    leaf_closest<MODE>(g, ~(uint32_t)w.cur, w.o, w.d, eps, best.t, best.prim, best.slot);
    w.cur = w.sp == 0 ? kTraversalDone : stk.pop(w.sp);
  }
}

// Inner nodes of an any-hit walk (no ordering: every child the segment touches is visited).
template <int MODE>
__device__ __forceinline__ void any_nodes(const Geom& g, const Stack& stk, Walk& w, float tfar) {
  if (MODE == kGlobalBvh4) {
    while (w.cur >= 0) {
      int4 refs;
      const float4* nd = g.nodes + g.stride * (size_t)w.cur;
      Node4Test nt = test_node4(nd[0], nd[1], nd[2], nd[3], w.o, w.inv, w.tmin, tfar, refs);
      int next = kTraversalDone;
      if (nt.hit[3]) next = refs.w;
      if (nt.hit[2]) {
        if (next != kTraversalDone) stk.push(w.sp, next);
        next = refs.z;
      }
      if (nt.hit[1]) {
        if (next != kTraversalDone) stk.push(w.sp, next);
        next = refs.y;
      }
      if (nt.hit[0]) {
        if (next != kTraversalDone) stk.push(w.sp, next);
        next = refs.x;
      }
      w.cur = next != kTraversalDone ? next : (w.sp == 0 ? kTraversalDone : stk.pop(w.sp));
    }
  } else {
    while (w.cur >= 0) {
      const float4* nd = g.nodes + g.stride * (size_t)w.cur;
      float4 q0 = nd[0], q1 = nd[1], q2 = nd[2], q3 = nd[3];
      float tn0, tn1;
      bool h0 = slab(q0.x, q0.y, q0.z, q0.w, q2.x, q2.y, w.inv, w.add_lo, w.add_hi, w.tmin, tfar, tn0);
      bool h1 = slab(q1.x, q1.y, q1.z, q1.w, q2.z, q2.w, w.inv, w.add_lo, w.add_hi, w.tmin, tfar, tn1);
      int c0 = __float_as_int(q3.x), c1 = __float_as_int(q3.y);
      if (h0 && h1) {
        stk.push(w.sp, c1);
        w.cur = c0;
      } else if (h0) {
        w.cur = c0;
      } else if (h1) {
        w.cur = c1;
      } else {
        w.cur = w.sp == 0 ? kTraversalDone : stk.pop(w.sp);
      }
    }
  }
}
// Leaves of an any-hit walk; the first hit ends the walk.
template <int MODE>
__device__ __forceinline__ void any_leaves(const Geom& g, const Stack& stk, float eps, float tfar, Walk& w,
                                           bool& hit) {
  while (w.cur < 0 && w.cur != kTraversalDone) {
    hit = hit || leaf_any<MODE>(g, ~(uint32_t)w.cur, w.o, w.d, eps, tfar);
    w.cur = (hit || w.sp == 0) ? kTraversalDone : stk.pop(w.sp);
  }
}

// Leaf pairs of the wide tree.  The reference fan-tessellates polygons (Scene.h:55-115), so the two halves of a quad
// arrive as consecutive triangles (a, b, c), (a, c, d), and a builder with leaves of two puts most such halves into
// one leaf (98 % of the leaves of a 1 M-triangle height field).  Such a leaf is one 64-byte record of its four
// vertices and two primitive ids instead of two records of a triangle each (kLeafPairFlag in the leaf word): the walk
// on the wide tree is bound by the record fetches that miss the L2, and half of those are triangles.  The edges and
// normals the single record stores are computed here instead, with the float operations the host uses for it
// (b + (-a), the cross product without contraction): a hit has the same bits either way.
//   row 0 (a.xyz, primitive id of (a, b, c))   row 1 (b.xyz, primitive id of (a, c, d))   row 2 (c.xyz, 0)   row 3 (d.xyz, 0)
constexpr uint32_t kLeafPairFlag = 4u;  // in the low three bits of a leaf word, beside count - 1 = 1
struct PairHit {
  float t0, t1;  // Triangle::intersect of each half (Obj.h:110-121); INFINITY: rejected
};
__device__ __forceinline__ PairHit pair_intersect(float4 q0, float4 q1, float4 q2, float4 q3, f3 o, f3 d) {
  const f3 a = mk3(q0.x, q0.y, q0.z);
  const f3 ba = vsub(mk3(q1.x, q1.y, q1.z), a), ca = vsub(mk3(q2.x, q2.y, q2.z), a), da = vsub(mk3(q3.x, q3.y, q3.z), a);
  const f3 roa = vsub(o, a);
  const f3 q = vcross(roa, d);
  PairHit h;
  {
    const f3 n = vcross(ba, ca);
    const float inv = 1.0f / vdot(d, n);
    const float u = inv * -vdot(q, ca);
    const float v = inv * vdot(q, ba);
    h.t0 = (u < 0.0f || u > 1.0f || v < 0.0f || (u + v) > 1.0f) ? INFINITY : inv * -vdot(n, roa);
  }
  {
    const f3 n = vcross(ca, da);
    const float inv = 1.0f / vdot(d, n);
    const float u = inv * -vdot(q, da);
    const float v = inv * vdot(q, ca);
    h.t1 = (u < 0.0f || u > 1.0f || v < 0.0f || (u + v) > 1.0f) ? INFINITY : inv * -vdot(n, roa);
  }
  return h;
}

// Single-phase walk for the wide tree in global memory, where every step waits on a
// fetch: nodes and triangles are both 64-byte records, so each iteration fetches
// the one record its lane stands on, whichever kind, and all lanes wait once.  A
// lane at an inner node then tests its four children, a lane at a leaf tests one
// triangle (a leaf of two takes two iterations).  Lanes no longer wait through
// whole node or leaf phases of the others.
// (The address is formed as an integer and the loads go through a global-address-space
// pointer: both arrays are device allocations in this mode.)
typedef float fvec4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) fvec4* global_rec;
__device__ __forceinline__ global_rec record_of(const Geom& g, int cur) {
  const uint32_t leaf = ~(uint32_t)cur;
  const uintptr_t a = cur >= 0 ? (uintptr_t)g.nodes + 64u * (uintptr_t)(uint32_t)cur
                               : (uintptr_t)g.tris + 64u * (uintptr_t)(leaf >> 3);
  return (global_rec)a;
}
__device__ __forceinline__ float4 as_float4(fvec4 v) { return make_float4(v.x, v.y, v.z, v.w); }
// After one triangle of leaf word `leaf`: the rest of the leaf, or the next stack entry.
__device__ __forceinline__ int after_triangle(uint32_t leaf, const Stack& stk, int& sp) {
  const uint32_t more = leaf & 7u;
  if (more != 0) return (int)~((((leaf >> 3) + 1u) << 3) | (more - 1u));
  return sp == 0 ? (int)kTraversalDone : stk.pop(sp);
}
__device__ __forceinline__ void closest_step(const Geom& g, const Stack& stk, float eps, Walk& w,
                                             ClosestHit& best) {
  {
    global_rec rec = record_of(g, w.cur);
    const float4 q0 = as_float4(rec[0]), q1 = as_float4(rec[1]), q2 = as_float4(rec[2]), q3 = as_float4(rec[3]);
    if (w.cur >= 0) {
      int4 refs;
      Node4Test nt = test_node4(q0, q1, q2, q3, w.o, w.inv, w.tmin, best.t, refs);
      float t0 = nt.hit[0] ? nt.tn[0] : INFINITY, t1 = nt.hit[1] ? nt.tn[1] : INFINITY;
      float t2 = nt.hit[2] ? nt.tn[2] : INFINITY, t3 = nt.hit[3] ? nt.tn[3] : INFINITY;
      int r0 = refs.x, r1 = refs.y, r2 = refs.z, r3 = refs.w;
      order2(t0, r0, t1, r1);
      order2(t2, r2, t3, r3);
      order2(t0, r0, t2, r2);
      order2(t1, r1, t3, r3);
      order2(t1, r1, t2, r2);
      if (t0 < INFINITY) {
        if (t3 < INFINITY) stk.push(w.sp, r3);
        if (t2 < INFINITY) stk.push(w.sp, r2);
        if (t1 < INFINITY) stk.push(w.sp, r1);
        w.cur = r0;
      } else {
        w.cur = w.sp == 0 ? kTraversalDone : stk.pop(w.sp);
      }
    } else {
      const uint32_t leaf = ~(uint32_t)w.cur;
      if (leaf & kLeafPairFlag) {  // both halves of a quad from one record, in slot order
        const PairHit h = pair_intersect(q0, q1, q2, q3, w.o, w.d);
        const uint32_t p0 = __float_as_uint(q0.w), p1 = __float_as_uint(q1.w);
        if (eps < h.t0 && h.t0 < INFINITY && (h.t0 < best.t || (h.t0 == best.t && p0 < best.prim))) {
          best.t = h.t0;
          best.prim = p0;
          best.slot = leaf >> 3;
        }
        if (eps < h.t1 && h.t1 < INFINITY && (h.t1 < best.t || (h.t1 == best.t && p1 < best.prim))) {
          best.t = h.t1;
          best.prim = p1;
          best.slot = (leaf >> 3) + 1u;
        }
        w.cur = w.sp == 0 ? (int)kTraversalDone : stk.pop(w.sp);
      } else {
        const float t = tri_intersect_n(mk3(q0.x, q0.y, q0.z), mk3(q1.x, q1.y, q1.z), mk3(q2.x, q2.y, q2.z),
                                        mk3(q3.x, q3.y, q3.z), w.o, w.d);
        const uint32_t prim = __float_as_uint(q0.w);
        if (eps < t && t < INFINITY && (t < best.t || (t == best.t && prim < best.prim))) {
          best.t = t;
          best.prim = prim;
          best.slot = leaf >> 3;
        }
        w.cur = after_triangle(leaf, stk, w.sp);
      }
    }
  }
}
__device__ __forceinline__ void closest_single_phase(const Geom& g, const Stack& stk, float eps, Walk& w,
                                                     ClosestHit& best) {
  while (w.cur != kTraversalDone) closest_step(g, stk, eps, w, best);
}
__device__ __forceinline__ void any_step(const Geom& g, const Stack& stk, float eps, float tfar, Walk& w,
                                         bool& hit) {
  {
    global_rec rec = record_of(g, w.cur);
    const float4 q0 = as_float4(rec[0]), q1 = as_float4(rec[1]), q2 = as_float4(rec[2]), q3 = as_float4(rec[3]);
    if (w.cur >= 0) {
      int4 refs;
      Node4Test nt = test_node4(q0, q1, q2, q3, w.o, w.inv, w.tmin, tfar, refs);
      int next = kTraversalDone;
      if (nt.hit[3]) next = refs.w;
      if (nt.hit[2]) {
        if (next != kTraversalDone) stk.push(w.sp, next);
        next = refs.z;
      }
      if (nt.hit[1]) {
        if (next != kTraversalDone) stk.push(w.sp, next);
        next = refs.y;
      }
      if (nt.hit[0]) {
        if (next != kTraversalDone) stk.push(w.sp, next);
        next = refs.x;
      }
      w.cur = next != kTraversalDone ? next : (w.sp == 0 ? kTraversalDone : stk.pop(w.sp));
    } else {
      const uint32_t leaf = ~(uint32_t)w.cur;
      if (leaf & kLeafPairFlag) {
        const PairHit h = pair_intersect(q0, q1, q2, q3, w.o, w.d);
        hit = (eps < h.t0 && h.t0 < tfar) || (eps < h.t1 && h.t1 < tfar);
        w.cur = (hit || w.sp == 0) ? (int)kTraversalDone : stk.pop(w.sp);
      } else {
        const float t = tri_intersect_n(mk3(q0.x, q0.y, q0.z), mk3(q1.x, q1.y, q1.z), mk3(q2.x, q2.y, q2.z),
                                        mk3(q3.x, q3.y, q3.z), w.o, w.d);
        hit = eps < t && t < tfar;
        if (hit)
          w.cur = kTraversalDone;
        else
          w.cur = after_triangle(leaf, stk, w.sp);
      }
    }
  }
}
__device__ __forceinline__ bool any_single_phase(const Geom& g, const Stack& stk, float eps, float tfar, Walk& w) {
  bool hit = false;
  while (w.cur != kTraversalDone) any_step(g, stk, eps, tfar, w, hit);
  return hit;
}

// Closest mesh hit of one ray per lane among the hits with t <= t_limit (a hit at
// exactly t_limit counts).  best_slot is kMiss, and best_t is t_limit, when there is none.
template <int MODE>
__device__ __forceinline__ void mesh_closest(const SceneDev& sc, const Geom& g, f3 o, f3 d, const Stack& stk,
                                             float t_limit, float& best_t, uint32_t& best_slot) {
  ClosestHit best = {t_limit, kMiss, kMiss};
  if constexpr (MODE == kFewTriangles) {
    if (sc.num_tris) leaf_closest<MODE>(g, sc.num_tris - 1u, o, d, sc.eps, best.t, best.prim, best.slot);  // "leaf" of all
    best_t = best.t;
    best_slot = best.slot;
    return;
  }
  Walk w;
  walk_begin<MODE>(sc, o, d, w);
  if constexpr (MODE == kGlobalBvh4) {
    closest_single_phase(g, stk, sc.eps, w, best);
  } else {
    while (w.cur != kTraversalDone) {
      closest_nodes<MODE>(g, stk, w, best);
      closest_leaves<MODE>(g, stk, sc.eps, w, best);
    }
  }
  best_t = best.t;
  best_slot = best.slot;
}

// Any mesh hit with eps < t < tfar.
template <int MODE>
__device__ __forceinline__ bool mesh_any(const SceneDev& sc, const Geom& g, f3 o, f3 d, float tfar,
                                         const Stack& stk) {
  if constexpr (MODE == kFewTriangles) return sc.num_tris && leaf_any<MODE>(g, sc.num_tris - 1u, o, d, sc.eps, tfar);
  Walk w;
  walk_begin<MODE>(sc, o, d, w);
  if constexpr (MODE == kGlobalBvh4) {
    return any_single_phase(g, stk, sc.eps, tfar, w);
  } else {
    bool hit = false;
    while (w.cur != kTraversalDone) {
      any_nodes<MODE>(g, stk, w, tfar);
      any_leaves<MODE>(g, stk, sc.eps, tfar, w, hit);
    }
    return hit;
  }
}

// The sphere a lane hit last, kept by kernels that run a path for many bounces: a path caught
// inside a glass sphere hits that sphere every time, and its three loads would stand in the
// chain of every bounce.
// `clear` (from the host): no triangle's box comes within a thousandth of a radius of the
// sphere's box.  A ray that starts inside such a sphere and ends on it cannot meet a triangle
// on the way, so cast_ray skips the mesh for it.
struct LastSphere {
  uint32_t index = 0xffffffffu;
  float4 centre_radius, material;
  int surface, clear = 0;
};
// The analytic objects of Scene::castRay (reference Scene.h:164-188): spheres in
// order, then boxes and planes continuing the same list (t > EPS, the first of
// equal t wins); the winner replaces the mesh hit only when strictly closer.
__device__ __forceinline__ void analytic_closest(const SceneDev& sc, f3 o, f3 d, float& t_io, uint32_t& id_io) {
  float ts = INFINITY;
  uint32_t si = 0;
  for (uint32_t i = 0; i < sc.num_spheres; i++) {
    float4 s = sc.sph0[i];
    float t = sphere_intersect(mk3(s.x, s.y, s.z), s.w, o, d, sc.eps);
    if (t > sc.eps && t < ts) {
      ts = t;
      si = i;
    }
  }
  for (uint32_t i = 0; i < sc.num_boxes; i++) {
    float4 a = sc.box0[i], b = sc.box1[i];
    float t = box_intersect(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), o, d, sc.eps);
    if (t > sc.eps && t < ts) {
      ts = t;
      si = sc.num_spheres + i;
    }
  }
  for (uint32_t i = 0; i < sc.num_planes; i++) {
    float4 a = sc.pln0[i], b = sc.pln1[i];
    float t = plane_intersect(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), o, d);
    if (t > sc.eps && t < ts) {
      ts = t;
      si = sc.num_spheres + sc.num_boxes + i;
    }
  }
  if (ts < t_io) {
    t_io = ts;
    id_io = kSphereBit | si;
  }
}
// The analytic objects of Scene::isOccluded (reference Scene.h:196-202).
__device__ __forceinline__ bool analytic_any(const SceneDev& sc, f3 o, f3 d, float tfar) {
  for (uint32_t i = 0; i < sc.num_spheres; i++) {
    float4 s = sc.sph0[i];
    float t = sphere_intersect(mk3(s.x, s.y, s.z), s.w, o, d, sc.eps);
    if (t > sc.eps && t < tfar) return true;
  }
  for (uint32_t i = 0; i < sc.num_boxes; i++) {
    float4 a = sc.box0[i], b = sc.box1[i];
    float t = box_intersect(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), o, d, sc.eps);
    if (t > sc.eps && t < tfar) return true;
  }
  for (uint32_t i = 0; i < sc.num_planes; i++) {
    float4 a = sc.pln0[i], b = sc.pln1[i];
    float t = plane_intersect(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), o, d);
    if (t > sc.eps && t < tfar) return true;
  }
  return false;
}

// Scene::castRay, reference Scene.h:144-189: the closest mesh hit and the closest
// analytic object (spheres in order, the first of equal t wins); the object wins only
// when strictly closer than the mesh hit.  The objects are tested first here and the
// tree is then walked for hits up to and including the object's distance, which gives
// the same winner and spares most of the walk for rays that end on an object (every
// bounce of a chain inside a glass sphere).  Returns t and an id: kMiss,
// kSphereBit|index, or triangle slot.
template <int MODE>
__device__ __forceinline__ void cast_ray(const SceneDev& sc, const Geom& g, f3 o, f3 d, const Stack& stk,
                                         float& t_out, uint32_t& id_out, const LastSphere* last = nullptr) {
  float ts = INFINITY;
  uint32_t sid = kMiss;
  analytic_closest(sc, o, d, ts, sid);
  if (last && last->clear && sid != kMiss && sid == (kSphereBit | last->index)) {
    // From inside the sphere hit last to that sphere again (the inside of a glass ball): the whole
    // segment lies in the ball, give or take a rounding, and no triangle comes that near.
    const float4 s = last->centre_radius;
    if (vsqrnorm(vsub(o, mk3(s.x, s.y, s.z))) <= s.w * s.w * 1.0002f) {
      t_out = ts;
      id_out = sid;
      return;
    }
  }
  float t;
  uint32_t slot;
  mesh_closest<MODE>(sc, g, o, d, stk, ts, t, slot);
  t_out = t;  // = ts when the mesh has nothing at or before ts
  id_out = slot != kMiss ? slot : sid;
}

// Scene::isOccluded, reference Scene.h:191-225, for a segment whose far end
// tfar = distance - eps has been computed by the caller (distance >= 2 eps).
template <int MODE>
__device__ __forceinline__ bool occluded(const SceneDev& sc, const Geom& g, f3 o, f3 d, float tfar,
                                         const Stack& stk) {
  if (analytic_any(sc, o, d, tfar)) return true;
  return mesh_any<MODE>(sc, g, o, d, tfar, stk);
}

// Surface normal and material of a hit (reference Scene.h:169-171,181-187;
// Sphere::normal Obj.h:269-271).  The mesh normal is normalize(Ng) with
// Ng = (b-a)x(c-a), not flipped towards the ray.
__device__ __forceinline__ void hit_surface(const SceneDev& sc, const Geom& g, uint32_t id, f3 o, f3 d, float t,
                                            f3& n, f3& albedo, float& emission, int& surface,
                                            LastSphere* last = nullptr) {
  if (sc.few_surfaces && last == nullptr) {
    // A handful of surfaces: every lane reads all of them, which makes the reads scalar loads (tens of cycles
    // from the scalar cache, three per surface, requested together), and keeps the one it hit.  The per-lane
    // lookup below costs a round trip through the vector memory path, some 1 500 cycles of a wave that lives 12 000.
    float4 c = make_float4(0, 0, 0, 1), m = make_float4(0, 0, 0, 0);
    f3 nt = mk3(0, 0, 0);
    int sf = 0;
    for (uint32_t i = 0; i < sc.num_spheres; i++) {
      const float4 ci = sc.sph0[i], mi = sc.sph1[i];
      const int si = sc.sph2[i];
      if (id == (kSphereBit | i)) {
        c = ci;
        m = mi;
        sf = si;
      }
    }
    for (uint32_t k = 0; k < sc.num_tris; k++) {
      const float4 Nk = sc.tri_nrm[k], mk = sc.tri_mat[k];
      const int sk = sc.tri_surf[k];
      if (id == k) {
        nt = mk3(Nk.x, Nk.y, Nk.z);
        m = mk;
        sf = sk;
      }
    }
    n = (id & kSphereBit) ? vdivs(vsub(vadd(o, vscale(d, t)), mk3(c.x, c.y, c.z)), c.w) : nt;
    albedo = mk3(m.x, m.y, m.z);
    emission = m.w;
    surface = sf;
    return;
  }
  if (sc.tri_mat != nullptr && last == nullptr) {
    // Small scenes: everything a hit may need is requested before anything is used, from addresses that are valid
    // for every lane (a lane that hit a sphere reads triangle slot 0 for nothing and the other way round; a few
    // cached lines), so a wave whose lanes hit spheres and triangles waits for one round trip.  With the loads
    // inside the branches it waits for the sphere's, then for the triangle's normal and material index, then for
    // the material that index names.
    const bool analytic = (id & kSphereBit) != 0;
    const uint32_t nobj = sc.num_spheres + sc.num_boxes + sc.num_planes;
    const uint32_t i = analytic ? (id & ~kSphereBit) : 0u, slot = analytic ? 0u : id;
    float4 ma = make_float4(0, 0, 0, 0), sp = make_float4(0, 0, 0, 1), N = ma, mt = ma;
    int surf_a = 0, surf_t = 0;
    if (nobj != 0) {
      ma = sc.sph1[i];
      surf_a = sc.sph2[i];
    }
    if (sc.num_spheres != 0) sp = sc.sph0[i < sc.num_spheres ? i : 0u];
    if (sc.num_tris != 0) {
      N = sc.tri_nrm[slot];
      mt = sc.tri_mat[slot];
      surf_t = sc.tri_surf[slot];
    }
    if (analytic) {
      const f3 p = vadd(o, vscale(d, t));
      if (i < sc.num_spheres) {
        n = vdivs(vsub(p, mk3(sp.x, sp.y, sp.z)), sp.w);
      } else if (i < sc.num_spheres + sc.num_boxes) {
        float4 a = sc.box0[i - sc.num_spheres], b = sc.box1[i - sc.num_spheres];
        n = box_normal(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), p);
      } else {
        float4 b = sc.pln1[i - sc.num_spheres - sc.num_boxes];
        n = mk3(b.x, b.y, b.z);
      }
    } else {
      n = mk3(N.x, N.y, N.z);  // = vnormalized(Ng), bit for bit (see rtx_set_scene)
    }
    const float4 m = analytic ? ma : mt;
    albedo = mk3(m.x, m.y, m.z);
    emission = m.w;
    surface = analytic ? surf_a : surf_t;
    return;
  }
  if (id & kSphereBit) {
    uint32_t i = id & ~kSphereBit;
    f3 p = vadd(o, vscale(d, t));
    if (last && last->index == i) {
      const float4 s = last->centre_radius, m = last->material;
      n = vdivs(vsub(p, mk3(s.x, s.y, s.z)), s.w);
      albedo = mk3(m.x, m.y, m.z);
      emission = m.w;
      surface = last->surface;
      return;
    }
    float4 m = sc.sph1[i];
    if (i < sc.num_spheres) {
      float4 s = sc.sph0[i];
      n = vdivs(vsub(p, mk3(s.x, s.y, s.z)), s.w);
      if (last) {
        last->index = i;
        last->centre_radius = s;
        last->material = m;
        last->surface = sc.sph2[i];
        last->clear = sc.sph3[i];
      }
    } else if (i < sc.num_spheres + sc.num_boxes) {
      float4 a = sc.box0[i - sc.num_spheres], b = sc.box1[i - sc.num_spheres];
      n = box_normal(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), p);
    } else {
      float4 b = sc.pln1[i - sc.num_spheres - sc.num_boxes];
      n = mk3(b.x, b.y, b.z);
    }
    albedo = mk3(m.x, m.y, m.z);
    emission = m.w;
    surface = sc.sph2[i];
  } else {
    const float4 N = sc.tri_nrm[id];  // = vnormalized(Ng), bit for bit (see rtx_set_scene)
    n = mk3(N.x, N.y, N.z);
    uint32_t mi = __float_as_uint(N.w);
    float4 m = sc.mat0[mi];
    albedo = mk3(m.x, m.y, m.z);
    emission = m.w;
    surface = sc.mat1[mi];
  }
}

// ----------------------------------------------------------------------------
// Shading helpers, reference Integrator.h and Matrix3.cpp.

// skyEmission, Integrator.h:73-85 (the comparisons widen `a` to double).
__device__ __forceinline__ f3 sky_emission(f3 d) {
  float a = vdot(d, vnormalized(mk3(0.2f, -0.8f, -0.4f)));
  f3 col = mk3(0.0f, 0.0f, 0.0f);
  if ((double)a > 0.999) {
    col = splat3(5.0f);
  } else if ((double)a > 0.96) {
    col = vdivs(vdivs(vscale(vscale(splat3(5.0f), a - 0.96f), a - 0.96f), 0.999f - 0.96f),
                0.999f - 0.96f);
  }
  return vadd(col, mk3(0.5f, 0.70f, 0.8f));
}

// schlickApprox, Integrator.h:45-51
__device__ __forceinline__ float schlick(float r, float cos_t) {
  float R0 = (r - 1.0f) / (r + 1.0f);
  R0 *= R0;
  float x = 1.0f - cos_t;
  float x2 = x * x;
  return R0 + (1.0f - R0) * x2 * x2 * x;
}

// rotMatrixVectors(a, b) * v, Matrix3.cpp:4-31.  The rotation axis b x a is not
// normalised in the reference; that is kept.
__device__ __forceinline__ f3 rot_vectors_apply(f3 a, f3 b, f3 v) {
  f3 ax = vcross(b, a);
  float c = vdot(a, b);
  float s = vnorm(ax);
  float m00 = c + ax.x * ax.x * (1.0f - c);
  float m01 = ax.x * ax.y * (1.0f - c) - ax.z * s;
  float m02 = ax.x * ax.z * (1.0f - c) + ax.y * s;
  float m10 = m01 + 2 * ax.z * s;
  float m11 = c + ax.y * ax.y * (1.0f - c);
  float m12 = ax.y * ax.z * (1.0f - c) - ax.x * s;
  float m20 = m02 - 2 * ax.y * s;
  float m21 = m12 + 2 * ax.x * s;
  float m22 = c + ax.z * ax.z * (1.0f - c);
  f3 r = mk3(0.0f, 0.0f, 0.0f);
  r.x += m00 * v.x + m01 * v.y + m02 * v.z;
  r.y += m10 * v.x + m11 * v.y + m12 * v.z;
  r.z += m20 * v.x + m21 * v.y + m22 * v.z;
  return r;
}

// cosineSampleHemisphere, Integrator.h:18-25.  U*2.0f is a float product, the
// product with M_PI is taken in double and narrowed.
__device__ __forceinline__ f3 cosine_sample_hemisphere(Rng& rng) {
  float r_sqr = pcg_uniform(rng);
  float phi = (float)((double)(pcg_uniform(rng) * 2.0f) * 3.14159265358979323846);
  float r = sqrtf(r_sqr);
  float sn, cs;
  sincosf(phi, &sn, &cs);
  float x = cs * r;
  float y = sn * r;
  return mk3(x, y, sqrtf(1.0f - x * x - y * y));
}

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

// ----------------------------------------------------------------------------
// One bounce of pathTrace.

// The primary ray of the k-th sample generated in a render call (reference
// Raytracer.cpp:246-250: two jitter draws, camera ray, ray.o += pos).  Sample k is
// sample index sample_base + k / band_pixels of pixel (k mod band_pixels) of the
// band, so consecutive lanes take consecutive pixels.  The random stream of a sample
// depends only on (pixel, sample index, seed); `rng` returns its state after the
// camera's draws.  A pure function of k: a kernel can recompute a fresh path's ray
// instead of reading it back (see Primary).
__device__ __forceinline__ void primary_ray(const CameraDev& cam, const RenderParams& rp, uint32_t k, f3& o, f3& d,
                                            Rng& rng, bool streams) {
  const uint32_t in_band = k % rp.band_pixels;
  const uint32_t sample = rp.sample_base + k / rp.band_pixels;
  const uint32_t pixel = rp.row0 * cam.width + in_band;
  const uint32_t px = pixel % cam.width, py = pixel / cam.width;
  rng.s = pcg_hash(pixel ^ pcg_hash(sample ^ pcg_hash(rp.seed32)));
  rng.inc = kPcgIncrement;
  if (streams) {  // (= rp.rng_streams; from the caller's copy of the scene, which a kernel built for a scene knows)
    const unsigned long long m = stream_mix(pixel, sample, rp.seed64);
    rng.s = (uint32_t)m;
    rng.inc = (uint32_t)(m >> 32) | 1u;
  }
  float jx = pcg_uniform(rng);
  float jy = pcg_uniform(rng);
  float fx = (float)(int)px + jx - 0.5f;
  float fy = (float)(int)py + jy - 0.5f;
  const float W = (float)cam.width, H = (float)cam.height;
  o = mk3(cam.ox, cam.oy, cam.oz);
  if (cam.type == 2) {  // OrthogonalCamera, Camera.h:73-79
    o = vadd(o, mk3((2.0f * fx - W) / W * cam.stretch_x, -(2.0f * fy - H) / H * cam.stretch_y, 0.f));
    d = mk3(0.0f, 0.0f, -1.0f);
  } else {
    float x = (2.0f * fx - W) / W * cam.tan_x;
    float y = (2.0f * fy - H) / H * cam.tan_y;
    if (cam.type == 0) {  // PerspectiveCamera, Camera.h:18-25
      d = vnormalized(mk3(x, -y, -1.0f));
    } else {  // ThinLensCamera, Camera.h:44-57
      d = mk3(x, -y, -1.0f);
      float t = -cam.focal_length / d.z;
      f3 focal = vadd(o, vscale(d, t));
      float r = pcg_uniform(rng);
      float theta = (float)((double)(pcg_uniform(rng) * 2) * 3.14159265358979323846);
      float sn, cs;
      sincosf(theta, &sn, &cs);
      o = vadd(o, vscale(mk3(r * cs, r * sn, 0.0f), cam.lens_radius));
      d = vnormalized(vsub(focal, o));
    }
  }
  o = vadd(o, mk3(cam.offx, cam.offy, cam.offz));
}

// How the kernels get at the fresh paths of a queue (slots gen_slot0.. of every
// segment, filled since the last round).  With `recompute` set there is no k_generate
// launch: nothing is stored for a fresh path, and k_intersect, k_shade and k_finish
// each recompute its ray from its sample number.  That trades about a hundred
// instructions per use for 32 bytes written and 24 + 32 bytes read back per fresh
// path, on kernels that wait on memory.  Otherwise k_generate has written ray,
// ordinal and generator state, and they are loaded.
struct Primary {
  CameraDev cam;
  RenderParams rp;
  int recompute;
};

struct PathState {
  f3 o, d, att, col;
  float mis_pdf, nee_pdf;
  uint32_t ordinal, depth;
  Rng rng;
};
struct ShadowReq {
  f3 d, contrib;
  float tfar;
};

// `fresh`: the path was written by k_generate and has not bounced yet.  Its
// attenuation (1), colour (0), mis_brdf_pdf (-1) and nee_pdf are the constants
// pathTrace starts with (Integrator.h:99-106); k_generate does not write them
// and they are not read back (32 of the 64 bytes, on the most numerous paths).
__device__ __forceinline__ void load_path(const SceneDev& sc, const Queue& q, uint32_t i, bool fresh,
                                          const Primary& pr, uint32_t fresh_k, PathState& p) {
  float4 s0 = make_float4(1.0f, 1.0f, 1.0f, -1.0f);
  float4 s1 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (fresh && pr.recompute) {
    primary_ray(pr.cam, pr.rp, fresh_k, p.o, p.d, p.rng, sc.rng_streams != 0);
    p.ordinal = fresh_k;
    p.depth = 0;
  } else {
    float4 r0 = q.r0[i];
    float2 r1 = q.r1[i];
    uint2 s2 = q.s2[i];
    if (!fresh) {
      s0 = q.s0[i];
      if (sc.carries_colour) s1 = q.s1[i];
    }
    p.o = mk3(r0.x, r0.y, r0.z);
    p.d = mk3(r0.w, r1.x, r1.y);
    p.ordinal = s2.x;
    p.rng.s = s2.y;
    p.rng.inc = sc.rng_streams ? q.inc[i] : kPcgIncrement;
    p.depth = sc.max_bounces ? q.depth[i] : 0;
  }
  p.att = mk3(s0.x, s0.y, s0.z);
  p.mis_pdf = s0.w;
  p.col = mk3(s1.x, s1.y, s1.z);
  p.nee_pdf = s1.w;
}
__device__ __forceinline__ void store_path(const SceneDev& sc, const Queue& q, uint32_t i, const PathState& p) {
  q.r0[i] = make_float4(p.o.x, p.o.y, p.o.z, p.d.x);
  q.r1[i] = make_float2(p.d.y, p.d.z);
  q.s0[i] = make_float4(p.att.x, p.att.y, p.att.z, p.mis_pdf);
  if (sc.carries_colour) q.s1[i] = make_float4(p.col.x, p.col.y, p.col.z, p.nee_pdf);
  q.s2[i] = make_uint2(p.ordinal, p.rng.s);
  if (sc.max_bounces) q.depth[i] = p.depth;
  if (sc.rng_streams) q.inc[i] = p.rng.inc;
}

// The body of the loop at reference Integrator.h:108-285 for a path whose closest hit (t, id) is known, in two
// halves.  shade_arrive: miss, surface lookup, emission, Russian roulette (:113-194); returns true when the
// path goes on.  shade_leave: the surface's switch (:196-278), after which p holds the next ray; `want_shadow`
// asks for the NEE segment in `sr` to be tested for occlusion before its contribution is added to the path's
// colour.  Whether a path goes on is known after the first half: a caller can reserve its queue slot there.
struct SurfaceHit {
  f3 n3, albedo;
  float emis;
  int surface;
};
template <bool SETTLE = false>
__device__ __forceinline__ bool shade_arrive(const SceneDev& sc, const Geom& g, PathState& p, float t, uint32_t id,
                                             SurfaceHit& h, LastSphere* last = nullptr) {
  if (sc.max_bounces) p.depth++;
  if (id == kMiss) {  // Integrator.h:113-116
    if (sc.sky || sc.integrator == 1) {
      f3 sky = sky_emission(p.d);
      p.col = (sc.integrator == 1) ? sky : vadd(p.col, vmul(p.att, sky));
    }
    return false;
  }
  f3 n3, albedo;
  float emis;
  int surface;
  hit_surface(sc, g, id, p.o, p.d, t, n3, albedo, emis, surface, last);
  h.n3 = n3;
  h.albedo = albedo;
  h.emis = emis;
  h.surface = surface;
  // SETTLE: every load of the lookup has landed when this half returns.  A wait for one of them further down
  // would be a wait for everything issued in between as well, a caller's early reservation included.
  if constexpr (SETTLE)
    asm volatile("" ::"v"(surface), "v"(albedo.x), "v"(albedo.y), "v"(albedo.z), "v"(n3.x), "v"(n3.y), "v"(n3.z));
  if (sc.integrator == 1) {  // rayTraceNormals, Integrator.h:87-96
    p.col = vadd(vscale(n3, 0.5f), splat3(0.5f));
    return false;
  }
  f3 hp = vadd(vscale(p.d, t), p.o);  // :119
  if (emis > 0.0f) {                  // :135-151 (emission is a scalar, so max() is itself)
    f3 e3 = splat3(emis);
    if (p.mis_pdf > 0.0f) {
      float sa = fabsf(-vdot(p.d, n3)) / p.nee_pdf / t / t;
      f3 num = vscale(vscale(vmul(e3, p.att), p.mis_pdf), p.mis_pdf);
      p.col = vadd(p.col, vdivs(num, 1.0f / sa / sa + p.mis_pdf * p.mis_pdf));
    } else {
      p.col = vadd(p.col, vmul(e3, p.att));
    }
  }
  p.o = hp;  // :184
  // Russian roulette, :189-194
  float pr = std_min(vmax3(p.att), sc.rr_cap);
  if (pcg_uniform(p.rng) > pr) return false;
  if (sc.max_bounces && p.depth >= sc.max_bounces) return false;
  p.att = vdivs(p.att, pr);
  return true;
}
__device__ __forceinline__ void shade_leave(const SceneDev& sc, PathState& p, const SurfaceHit& h, bool& want_shadow,
                                            bool& did_nee, ShadowReq& sr) {
  f3 n3 = h.n3;
  const f3 albedo = h.albedo;
  const float emis = h.emis;
  const int surface = h.surface;
  if (surface == 2) {  // reflective, :198-205
    float cos_t = vdot(p.d, n3);
    p.d = vsub(p.d, vscale(vscale(n3, cos_t), 2.0f));
    p.att = vmul(p.att, albedo);
    p.mis_pdf = -1.0f;
  } else if (surface == 0) {  // diffuse, :206-247
    if (sc.num_lights != 0 && emis < sc.nee_threshold) {
      // sampleRandomPointOnLight, Scene.h:227-235.  The light is picked with one
      // draw of the path's generator, uniformly (the reference uses rand()%2).
      uint32_t nl = sc.num_lights;
      float ul = pcg_uniform(p.rng);
      uint32_t li = (uint32_t)(ul * (float)nl);
      if (li >= nl) li = nl - 1;
      float4 L0 = sc.light0[li], L1 = sc.light1[li], L2 = sc.light2[li], L3 = sc.light3[li];
      // (the light's emission with the rest of its record: left to itself the compiler fetches that one word
      // after the other loads have come back, a second round trip in the middle of the bounce)
      asm volatile("" ::"v"(L1.w));
      // Triangle::samplePoint, Obj.h:127-136
      float u0 = pcg_uniform(p.rng);
      float u1 = pcg_uniform(p.rng);
      float su0 = sqrtf(u0);
      float b0 = 1.0f - su0;
      float b1 = u1 * su0;
      f3 sp = vadd(vadd(vscale(mk3(L0.x, L0.y, L0.z), b0), vscale(mk3(L1.x, L1.y, L1.z), b1)),
                   vscale(mk3(L2.x, L2.y, L2.z), 1.0f - b0 - b1));
      p.nee_pdf = L0.w / (float)nl;
      f3 to = vsub(sp, p.o);
      float distance = vnorm(to);
      f3 ld = vdivs(to, distance);
      did_nee = true;
      float sa = fabsf(-vdot(ld, mk3(L3.x, L3.y, L3.z))) / p.nee_pdf / distance / distance;
      if (sa != 0.0f) {  // :219-223; denominator in double, narrowed for Vec3/float
        f3 v = vscale(vmul(vscale(p.att, L1.w), albedo), fabsf(vdot(n3, ld)));
        v = vdivs(v, 3.14159265358979323846f);
        v = vscale(v, 1.0f);
        v = vdivs(v, sa);
        double den = (double)(1.0f / sa / sa) + (double)fabsf(vdot(ld, n3)) / 3.14159265358979323846 *
                                                    (double)fabsf(vdot(ld, n3)) / 3.14159265358979323846;
        f3 contrib = vdivs(v, (float)den);
        if (distance < 2.0f * sc.eps) {
          p.col = vadd(p.col, contrib);  // isOccluded returns false at once, Scene.h:192-194
        } else {
          want_shadow = true;
          sr.d = ld;
          sr.tfar = distance - sc.eps;
          sr.contrib = contrib;
        }
      }
    }
    p.d = rot_vectors_apply(n3, mk3(0.0f, 0.0f, 1.0f), cosine_sample_hemisphere(p.rng));  // :241-242
    float cos_t = std_max(vdot(p.d, n3), 0.0f);
    p.mis_pdf = did_nee ? (float)((double)cos_t / 3.14159265358979323846) : -1.0f;  // :244
    p.att = vmul(p.att, albedo);
  } else if (surface == 1) {  // specular (glass), :248-278
    float ior = 1.0f / 2.0f;
    float cos_t1 = -vdot(n3, p.d);
    bool from_outside = cos_t1 > 0.0f;
    if (!from_outside) {
      cos_t1 *= -1;
      ior = 1.0f / ior;
      n3 = vneg(n3);
    }
    float cos_t2_sqr = 1.0f - ior * ior * (1.0f - cos_t1 * cos_t1);
    bool refract = false;
    if (cos_t2_sqr >= 0.0f) {  // no draw on total internal reflection
      float fr = schlick(ior, from_outside ? cos_t1 : sqrtf(cos_t2_sqr));
      refract = pcg_uniform(p.rng) > fr;
    }
    if (refract) {
      float cos_t2 = sqrtf(cos_t2_sqr);
      p.d = vadd(vscale(p.d, ior), vscale(n3, ior * cos_t1 - cos_t2));
      p.d = vdivs(p.d, vnorm(p.d));
    } else {
      p.d = vadd(p.d, vscale(vscale(n3, cos_t1), 2));
    }
    p.att = vmul(p.att, albedo);
    p.mis_pdf = -1.0f;
  }
  // varnish: the reference's switch has no case for it; nothing changes.
}
__device__ __forceinline__ bool shade_bounce(const SceneDev& sc, const Geom& g, PathState& p, float t,
                                             uint32_t id, bool& want_shadow, bool& did_nee, ShadowReq& sr,
                                             LastSphere* last = nullptr) {
  want_shadow = false;
  did_nee = false;
  SurfaceHit h;
  if (!shade_arrive(sc, g, p, t, id, h, last)) return false;
  shade_leave(sc, p, h, want_shadow, did_nee, sr);
  return true;
}

// A finished path's colour goes to the slot of its sample (packed RGB floats).
__device__ __forceinline__ void store_result(float* res, uint32_t ordinal, f3 col) {
  float* dst = res + 3 * (size_t)ordinal;
  dst[0] = col.x;
  dst[1] = col.y;
  dst[2] = col.z;
}

// Counts the finished paths whose colour is not finite (the reference can produce them:
// sqrt of a slightly negative number at Integrator.h:24; they are counted, not masked).
// Every lane of the wave calls this together; one atomic per wave that has any.
__device__ __forceinline__ bool colour_not_finite(f3 c) {
  return !(fabsf(c.x) < INFINITY && fabsf(c.y) < INFINITY && fabsf(c.z) < INFINITY);
}
__device__ __forceinline__ void wave_count(bool flag, uint32_t* counter) {
  const unsigned long long m = __ballot(flag);
  if (m != 0) {
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if (lane == (uint32_t)__builtin_ctzll(m)) atomicAdd(counter, (uint32_t)__popcll(m));
  }
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

// Closest mesh hit for the rays of a whole block when the tree has a handful of leaves (Cornell's seventeen).
// In a walk a ray needs about eight node and 1.2 leaf visits, but its wave runs twenty node rounds and seven
// leaf rounds, the latter with a sixth of the lanes at work, because the lanes' walks differ.  Here every lane
// tests the boxes of all leaves (scalar operands, no stack, no divergence; what an ordered walk would have
// culled behind a hit is a twentieth of the leaf visits in such a scene), the (ray, leaf) pairs go to a queue
// in LDS, and the block's waves test them in dense batches of 64; a hit is merged into its ray's record with a
// 64-bit minimum of (t, primitive id, slot): the smallest t and, among equal t, the lowest primitive id, which is
// the rule of Mesh::intersect (reference Obj.h:184-195).  Every thread of the block calls this (barriers);
// `active` lanes bring a ray, and get the closest hit with t <= t_limit (best_slot kMiss, best_t = t_limit:
// none).  The queue holds sc.leaf_task_cap pairs per thread on average (a ray meets 1.3 leaf boxes); a lane whose
// pairs do not fit tests its leaves itself.  LDS at `ws`: leaf_task_lds_bytes; the caller has set ws[0] to
// (0, queue capacity, ...) as uint32 before its last barrier.
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
// Bit l set: the ray of walk `w` meets the box of leaf l between w.tmin and tmax.  The boxes come through the
// scalar cache; each is requested one test ahead, so that a test does not begin with a wait for its operands.
__device__ __forceinline__ uint32_t leaf_boxes_met(const SceneDev& sc, const Walk& w, float tmax) {
  uint32_t mask = 0;
  const uint32_t nl = sc.num_leaf_boxes;
  float4 a = sc.leaf_box[0], b = sc.leaf_box[1];
#pragma unroll 2
  for (uint32_t l = 0; l < nl; l++) {
    const uint32_t nx = l + 1u < nl ? l + 1u : l;
    const float4 na = sc.leaf_box[2 * nx], nb = sc.leaf_box[2 * nx + 1];
    float tn;
    if (slab(a.x, a.y, a.z, a.w, b.x, b.y, w.inv, w.add_lo, w.add_hi, w.tmin, tmax, tn)) mask |= 1u << l;
    a = na;
    b = nb;
  }
  return mask;
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
                                                            uint32_t& best_slot) {
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
    uint32_t mask = leaf_boxes_met(sc, w, t_limit);
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
  __syncthreads();
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
  __syncthreads();
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
                                                        f3 o, f3 d, float tfar) {
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
    uint32_t mask = leaf_boxes_met(sc, w, tfar);
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
  __syncthreads();
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
  __syncthreads();
  return active && (hit || flag[tid] != 0);
}

// k_intersect and k_shade_wave_append in one launch, for scenes whose closest-hit search is
// a handful of tests (tree staged in LDS, a few nodes): the search then costs less than
// carrying the ray to a kernel of its own and the hit back, so each path is read once per
// bounce and the hit stays in registers.
// (eight waves per SIMD: the compiler then keeps to 96 scalar registers as well, which it passed by five
// otherwise, and a wave of this kernel mostly waits: sample2 8.8 -> 8.3 ms, Cornell's bounces 51.7 -> 49.4)
// (kLdsLeafTasks: its LDS workspace leaves room for seven blocks of 256 per CU, so seven waves per SIMD it is)
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
  if constexpr (MODE == kLdsLeafTasks) leaf_task_begin<BLOCK>(sc, task_ws);  // (ahead of the staging's barrier)
  const Geom geom = WALK == kLdsBvh2 ? stage_geom_lds(sc, lds_dyn + scene_at, threadIdx.x, BLOCK) : global_geom(sc);
  const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
  const bool active = j < n;
  const uint32_t i = s * cap + j;

  PathState p;
  ShadowReq sr;
  SurfaceHit sh;
  bool alive = false, want_shadow = false, did_nee = false, bad = false;
  float t = INFINITY;
  uint32_t id = kMiss;
  if constexpr (MODE == kLdsLeafTasks) {
    // cast_ray with the mesh part done by the whole block
    // (requesting the path ahead of the staging, as k_shadow_leaf_tasks does with its samples, measured slower here)
    p.o = p.d = mk3(0.0f, 0.0f, 0.0f);
    float ts = INFINITY;
    uint32_t sid = kMiss;
    if (active) {
      load_path(sc, in, i, j >= seg[s].gen_slot0, *prp, (uint32_t)seg[s].gen_ordinal0 + (j - seg[s].gen_slot0), p);
      analytic_closest(sc, p.o, p.d, ts, sid);
    }
    uint32_t slot;
    block_closest_by_leaf_tasks<BLOCK>(sc, geom, task_ws, active, p.o, p.d, ts, t, slot);
    id = slot != kMiss ? slot : sid;
  }
  // Without lights a bounce queues no light sample, so the append counts paths only, and which paths go on is
  // known once the roulette is through: the slots are reserved there and the atomic's round trip runs beside
  // the sampling of the next direction.
  constexpr bool reserve_early = !LIGHTS;  // (LIGHTS: the scene has any, sc.num_lights != 0)
  bool live = active;  // this lane carries a path
  if constexpr (MODE != kLdsLeafTasks) {
    if (active)
      load_path(sc, in, i, j >= seg[s].gen_slot0, *prp, (uint32_t)seg[s].gen_ordinal0 + (j - seg[s].gen_slot0), p);
    if constexpr (BOUNCES > 1) {
      // The first BOUNCES - 1 bounces of a launch: a path that goes on stays in its lane (its state is neither
      // written nor read back), the lanes of the paths that ended idle.  Counted in seg[s].more_bounces.
      static_assert(!LIGHTS, "a bounce that samples a light queues the sample against the path's slot");
#pragma unroll 1
      for (int b = 1; b < BOUNCES; b++) {
        bool goes_on = false;
        if (live) {
          cast_ray<WALK>(sc, geom, p.o, p.d, stk, t, id);
          goes_on = shade_bounce(sc, global_geom(sc), p, t, id, want_shadow, did_nee, sr);
          if (!goes_on) {
            store_result(res, p.ordinal, p.col);
            bad = bad || colour_not_finite(p.col);
          }
        }
        live = goes_on;
        wave_count(live, &seg[s].more_bounces);
      }
    }
  }
  if (live) {
    if constexpr (MODE != kLdsLeafTasks) cast_ray<WALK>(sc, geom, p.o, p.d, stk, t, id);
    alive = reserve_early ? shade_arrive<true>(sc, global_geom(sc), p, t, id, sh)
                          : shade_bounce(sc, global_geom(sc), p, t, id, want_shadow, did_nee, sr);
    if (!alive) {
      store_result(res, p.ordinal, p.col);
      bad = bad || colour_not_finite(p.col);
    }
  }
  wave_count(bad, &sum->stat_nan);
  bool queue_sample = want_shadow;
  if constexpr (INLINE_LIGHT_SAMPLES) {
    // The light sample is traced here, as a second round of tasks of the block: it is counted, not queued.
    static_assert(MODE == kLdsLeafTasks && LIGHTS, "the block's task queue is what traces the sample");
    if (!want_shadow) {  // (lanes without a sample bring defined values to the block's call)
      sr.d = mk3(0.0f, 0.0f, 0.0f);
      sr.tfar = 0.0f;
    }
    const bool try_light = want_shadow && !analytic_any(sc, p.o, sr.d, sr.tfar);
    const bool occ = block_any_by_leaf_tasks<BLOCK, true>(sc, geom, task_ws, try_light, p.o, sr.d, sr.tfar);
    if (try_light && !occ) p.col = vadd(p.col, sr.contrib);
    queue_sample = false;
  }
  uint32_t pos, spos = 0;  // (the per-block append of k_shade_block_append measured slower here)
  if (reserve_early) {
    const PathTicket tk = wave_reserve_paths(alive, &seg[s].c[qi ^ 1]);
    if (alive) shade_leave(sc, p, sh, want_shadow, did_nee, sr);
    pos = wave_path_slot(tk);
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

// An unoccluded NEE segment adds its contribution to the colour of its path.
__device__ __forceinline__ void add_contribution(const Queue& paths, const ShadowQueue& sq, uint32_t k,
                                                 uint32_t slot) {
  float4 c = sq.q2[k];
  float4 s1 = paths.s1[slot];
  s1.x += c.x;
  s1.y += c.y;
  s1.z += c.z;
  paths.s1[slot] = s1;
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
  const Geom geom = stage_geom_lds(sc, lds_dyn, threadIdx.x, BLOCK);  // (ends with a barrier)
  const bool free_so_far = active && !analytic_any(sc, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), a.w);
  const bool occ = block_any_by_leaf_tasks<BLOCK>(sc, geom, task_ws, free_so_far, mk3(a.x, a.y, a.z),
                                                  mk3(b.x, b.y, b.z), a.w);
  if (free_so_far && !occ) add_contribution(paths, sq, k, __float_as_uint(b.w));
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
      prim[i] = __float_as_uint(sc.tris[4 * (size_t)id].w);
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

__global__ void k_resolve(const float* accum, float* out, size_t n, float spp) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = accum[i] / spp;
}

}  // namespace rtx
