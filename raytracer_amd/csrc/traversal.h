// traversal.h -- geometry tests and tree walks: Triangle/Sphere/Box/Plane intersection (reference Obj.h), the two-child
// tree (global memory or LDS), the four-child quantised tree with its single-phase walk and leaf pairs, and on top of
// them Scene::castRay / Scene::isOccluded and the surface lookup of a hit (reference Scene.h:144-225).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "scene_types.h"

namespace rtx {

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
__device__ __forceinline__ void slab_interval(float lox, float hix, float loy, float hiy, float loz, float hiz,
                                              f3 inv, f3 add_lo, f3 add_hi, float tmin, float tmax, float& tnear,
                                              float& tfar_slack) {
  float ax = __builtin_fmaf(lox, inv.x, add_lo.x), bx = __builtin_fmaf(hix, inv.x, add_hi.x);
  float ay = __builtin_fmaf(loy, inv.y, add_lo.y), by = __builtin_fmaf(hiy, inv.y, add_hi.y);
  float az = __builtin_fmaf(loz, inv.z, add_lo.z), bz = __builtin_fmaf(hiz, inv.z, add_hi.z);
  tnear = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), tmin));
  tfar_slack = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tmax)) * 1.000001f;
}
// The box is met where tnear <= tfar_slack.
__device__ __forceinline__ bool slab(float lox, float hix, float loy, float hiy, float loz, float hiz,
                                     f3 inv, f3 add_lo, f3 add_hi, float tmin, float tmax, float& tnear) {
  float tf;
  slab_interval(lox, hix, loy, hiy, loz, hiz, inv, add_lo, add_hi, tmin, tmax, tnear, tf);
  return tnear <= tf;
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
//   row 0 (a.xyz, primitive id of (a, b, c))   row 1 (b.xyz, primitive id of (a, c, d))
//   row 2 (c.xyz, triangle slot of (a, b, c); the other half has the next slot)   row 3 (d.xyz, 0)
// The wide tree's leaf records are packed in leaf order (rtx_set_scene): a leaf word's upper bits number records, and
// every record, pair or single, names its triangle slot in row 2.
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
        const uint32_t slot = __float_as_uint(q2.w);
        if (eps < h.t0 && h.t0 < INFINITY && (h.t0 < best.t || (h.t0 == best.t && p0 < best.prim))) {
          best.t = h.t0;
          best.prim = p0;
          best.slot = slot;
        }
        if (eps < h.t1 && h.t1 < INFINITY && (h.t1 < best.t || (h.t1 == best.t && p1 < best.prim))) {
          best.t = h.t1;
          best.prim = p1;
          best.slot = slot + 1u;
        }
        w.cur = w.sp == 0 ? (int)kTraversalDone : stk.pop(w.sp);
      } else {
        const float t = tri_intersect_n(mk3(q0.x, q0.y, q0.z), mk3(q1.x, q1.y, q1.z), mk3(q2.x, q2.y, q2.z),
                                        mk3(q3.x, q3.y, q3.z), w.o, w.d);
        const uint32_t prim = __float_as_uint(q0.w);
        if (eps < t && t < INFINITY && (t < best.t || (t == best.t && prim < best.prim))) {
          best.t = t;
          best.prim = prim;
          best.slot = __float_as_uint(q2.w);
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

}  // namespace rtx
