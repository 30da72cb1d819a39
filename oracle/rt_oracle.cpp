// rt_oracle.cpp -- scalar CPU restatement of the reference's radiance loop.
//
// TEST INFRASTRUCTURE ONLY (see rt_oracle.h for the rules and the pinning status:
// only the Pcg generator and the oven-box answer are pinned to the reference;
// everything else is "parity unpinned").
//
// Build: g++ -O2 -std=c++17 -ffp-contract=off -fopenmp (no -march=native, no
// -ffast-math): every float operation below is one IEEE-754 binary32 operation
// in source order, which is the arithmetic the HIP kernels must reproduce.
//
// Every function cites the reference file:line it follows (paths relative to the
// reference checkout).  Deliberate deviations from the reference, all stated in
// the functions concerned:
//   (1) per-path counter-based RNG seeding instead of 8 per-thread generators
//       (Raytracer.cpp:233,244,311-313);
//   (2) the light is picked with one draw of the path's own Pcg, uniformly over
//       all lights, instead of libc rand()%2 (Scene.h:232);
//   (3) jitter draws are taken x first, then y (unspecified at Raytracer.cpp:249);
//   (4) samples are summed and divided once, instead of the running mean of
//       Raytracer.cpp:78-81 (orc_img_update restates that for comparison);
//   (5) optional max_bounces cap (0 = none, as the reference).
#include "rt_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ----------------------------------------------------------------------------
// Vec3 arithmetic, Vec3.cpp:15-114.  Note the three conventions that matter for
// bits: a-b is a+(-b) (:87-89), v/f is v*(1.0f/f) (:78-80), normalized() uses
// true division by the norm (:27-30).
struct V3 {
  float x, y, z;
};
inline V3 mk(float x, float y, float z) { return V3{x, y, z}; }
inline V3 splat(float s) { return V3{s, s, s}; }                                   // Vec3.cpp:9
inline V3 add(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }          // :63-65
inline V3 neg(V3 a) { return mk(-a.x, -a.y, -a.z); }                               // :84-86
inline V3 sub(V3 a, V3 b) { return add(a, neg(b)); }                               // :87-89
inline V3 scale(V3 a, float b) { return mk(a.x * b, a.y * b, a.z * b); }           // :66-71
inline V3 mul(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }          // :72-74
inline V3 divs(V3 a, float b) { return scale(a, 1.0f / b); }                       // :78-80
inline float dot(V3 a, V3 o) { return a.x * o.x + a.y * o.y + a.z * o.z; }         // :15-17
inline V3 cross(V3 a, V3 o) {                                                      // :18-20
  return mk(a.y * o.z - a.z * o.y, a.z * o.x - a.x * o.z, a.x * o.y - a.y * o.x);
}
inline float sqr_norm(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }          // :21-23
inline float norm(V3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }   // :24-26
inline V3 normalized(V3 a) {                                                       // :27-30
  float n = norm(a);
  return mk(a.x / n, a.y / n, a.z / n);
}
inline float max3(V3 a) { return std::max(a.x, std::max(a.y, a.z)); }              // :35-37
inline V3 ld(const float* p) { return mk(p[0], p[1], p[2]); }
inline void st(float* p, V3 v) {
  p[0] = v.x;
  p[1] = v.y;
  p[2] = v.z;
}

const float kInf = std::numeric_limits<float>::infinity();
const double kPi = 3.14159265358979323846;  // M_PI

// ----------------------------------------------------------------------------
// Sampler.h:11-42.  32-bit-state PCG (RXS-M-XS output of the OLD state).
struct Rng {
  uint32_t s;
  uint32_t inc = 2891336453u;  // the reference's constant; orc_config.rng_streams gives every sample its own (stream_rng)
  uint32_t next() {  // rand_pcg, Sampler.h:24-30
    uint32_t old = s;
    s = s * 747796405u + inc;
    uint32_t w = ((old >> ((old >> 28u) + 4u)) ^ old) * 277803737u;
    return (w >> 22u) ^ w;
  }
  float uniform() {  // Uniform, Sampler.h:39-41 (ONE_MINUS_EPSILON = 0x1.fffffep-1, :8)
    float u = next() * 0x1p-32f;
    const float one_minus_eps = 0x1.fffffep-1f;
    return (u < one_minus_eps) ? u : one_minus_eps;  // std::min(ome, u): returns u iff u < ome
  }
};
inline uint32_t pcg_hash(uint32_t in) {  // Sampler.h:32-37
  uint32_t st = in * 747796405u + 2891336453u;
  uint32_t w = ((st >> ((st >> 28u) + 4u)) ^ st) * 277803737u;
  return (w >> 22u) ^ w;
}
// Deviation (1): one stream per (pixel, sample), built from the reference's own
// pcg_hash: state = hash(pixel ^ hash(sample ^ hash(seed32))).
inline uint32_t path_seed(uint32_t pixel, uint32_t sample, uint64_t seed) {
  uint32_t s32 = (uint32_t)(seed ^ (seed >> 32));
  return pcg_hash(pixel ^ pcg_hash(sample ^ pcg_hash(s32)));
}

// orc_config.rng_streams (no counterpart in the reference): state and increment of a sample's generator are the
// two halves of a 64-bit bijective mix (xor with the seed, then MurmurHash3's finaliser) of (sample, pixel).
inline Rng stream_rng(uint32_t pixel, uint32_t sample, uint64_t seed) {
  uint64_t k = (((uint64_t)sample << 32) | pixel) ^ seed;
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  Rng g{(uint32_t)k};
  g.inc = (uint32_t)(k >> 32) | 1u;
  return g;
}

// ----------------------------------------------------------------------------
// Matrix3.cpp:4-31
struct M3 {
  float m[3][3];
};
inline V3 matvec(const M3& a, V3 v) {  // Matrix3.cpp:4-11: mv starts at 0 and is += the row sum
  V3 r = mk(0.0f, 0.0f, 0.0f);
  r.x += a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z;
  r.y += a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z;
  r.z += a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z;
  return r;
}
// Matrix3.cpp:14-31.  The axis is NOT normalised, so this is a rotation only
// when |b x a| is 0 or 1; reproduced as written.
inline M3 rot_matrix_vectors(V3 a, V3 b) {
  V3 ax = cross(b, a);
  float c = dot(a, b);
  float s = norm(ax);
  M3 r;
  r.m[0][0] = c + ax.x * ax.x * (1.0f - c);
  r.m[0][1] = ax.x * ax.y * (1.0f - c) - ax.z * s;
  r.m[0][2] = ax.x * ax.z * (1.0f - c) + ax.y * s;
  r.m[1][0] = r.m[0][1] + 2 * ax.z * s;
  r.m[1][1] = c + ax.y * ax.y * (1.0f - c);
  r.m[1][2] = ax.y * ax.z * (1.0f - c) - ax.x * s;
  r.m[2][0] = r.m[0][2] - 2 * ax.y * s;
  r.m[2][1] = r.m[1][2] + 2 * ax.x * s;
  r.m[2][2] = c + ax.z * ax.z * (1.0f - c);
  return r;
}

// ----------------------------------------------------------------------------
// Integrator.h:18-25.  phi: (U*2.0f) in float, times M_PI in double, narrowed.
inline V3 cosine_sample_hemisphere(Rng& g) {
  float r_sqr = g.uniform();
  float phi = (float)((double)(g.uniform() * 2.0f) * kPi);
  float r = std::sqrt(r_sqr);
  float x = std::cos(phi) * r;
  float y = std::sin(phi) * r;
  return mk(x, y, std::sqrt(1.0f - x * x - y * y));
}
// Integrator.h:45-51
inline float schlick(float r, float cos_t) {
  float R0 = (r - 1.0f) / (r + 1.0f);
  R0 *= R0;
  float x = 1.0f - cos_t;
  float x2 = x * x;
  return R0 + (1.0f - R0) * x2 * x2 * x;
}
// Integrator.h:73-85.  The comparisons are float-vs-double-literal (a widened).
inline V3 sky_emission(V3 d) {
  float a = dot(d, normalized(mk(0.2f, -0.8f, -0.4f)));
  V3 col = mk(0.0f, 0.0f, 0.0f);
  if ((double)a > 0.999) {
    col = splat(5.0f);
  } else if ((double)a > 0.96) {
    col = divs(divs(scale(scale(splat(5.0f), a - 0.96f), a - 0.96f), 0.999f - 0.96f), 0.999f - 0.96f);
  }
  col = add(col, mk(0.5f, 0.70f, 0.8f));
  return col;
}

// ----------------------------------------------------------------------------
// Geometry, Obj.h.
struct Tri {
  V3 a, b, c;
};
// Triangle::intersect, Obj.h:110-121.  Comparisons kept in the reference's
// form so NaN/Inf inputs fall through the same way.
inline float tri_intersect(const Tri& T, V3 o, V3 d) {
  V3 ba = sub(T.b, T.a);
  V3 ca = sub(T.c, T.a);
  V3 roa = sub(o, T.a);
  V3 n = cross(ba, ca);
  V3 q = cross(roa, d);
  float inv = 1.0f / dot(d, n);
  float u = inv * -dot(q, ca);
  float v = inv * dot(q, ba);
  if (u < 0.0f || u > 1.0f || v < 0.0f || (u + v) > 1.0f) return kInf;
  return inv * -dot(n, roa);
}
// Sphere::intersect, Obj.h:257-268 (assumes |d| = 1; d may not be).
inline float sphere_intersect(V3 c, float r, V3 o, V3 d, float eps) {
  float B = 2.0f * dot(d, sub(o, c));
  float C = sqr_norm(sub(o, c)) - r * r;
  float disc2 = B * B - 4.0f * C;
  if (disc2 < 0.0f) return kInf;
  float disc = std::sqrt(disc2);
  float t1 = (-B - disc) / 2.0f;
  return t1 > eps ? t1 : (-B + disc) / 2.0f;
}
// Box::intersect, the reference's commented-out class (Obj.h:43-68).
inline float box_intersect(V3 mn, V3 mx, V3 o, V3 d, float eps) {
  float t1 = (mn.x - o.x) / d.x;
  float t2 = (mx.x - o.x) / d.x;
  float tmin = std::min(t1, t2);
  float tmax = std::max(t1, t2);
  t1 = (mn.y - o.y) / d.y;
  t2 = (mx.y - o.y) / d.y;
  tmin = std::max(tmin, std::min(t1, t2));
  tmax = std::min(tmax, std::max(t1, t2));
  t1 = (mn.z - o.z) / d.z;
  t2 = (mx.z - o.z) / d.z;
  tmin = std::max(tmin, std::min(t1, t2));
  tmax = std::min(tmax, std::max(t1, t2));
  return tmax >= tmin ? (tmin > eps ? tmin : tmax) : kInf;
}
// sgn, Obj.h:21-23; Box::normal, Obj.h:70-87.
inline float sgn(float x) { return (float)((x > 0) - (x < 0)); }
inline V3 box_normal(V3 mn, V3 mx, V3 p) {
  V3 c = divs(add(mx, mn), 2.0f);
  V3 v = sub(p, c);
  const float cv[3] = {v.x, v.y, v.z};
  const float lo[3] = {mn.x, mn.y, mn.z}, hi[3] = {mx.x, mx.y, mx.z};
  int am = 0;
  for (int i = 1; i < 3; i++)
    if (std::fabs(cv[i]) * (hi[am] - lo[am]) > std::fabs(cv[am]) * (hi[i] - lo[i])) am = i;
  return mk((am == 0) * sgn(v.x), (am == 1) * sgn(v.y), (am == 2) * sgn(v.z));
}
// Plane(point, normal): build-defined (the reference has call sites only).
inline float plane_intersect(V3 pt, V3 n, V3 o, V3 d) { return dot(sub(pt, o), n) / dot(d, n); }

// A light = Triangle copy made in Scene::commit (Scene.h:129-141) through the
// Triangle ctor (Obj.h:107): n = normalize((b-a)x(c-b)), rcp_area = 2/|(b-a)x(c-b)|.
struct Light {
  Tri t;
  V3 n;
  float rcp_area;
  orc_material mat;
};
inline Light make_light(const Tri& t, const orc_material& m) {
  Light L;
  L.t = t;
  V3 cr = cross(sub(t.b, t.a), sub(t.c, t.b));
  L.n = normalized(cr);
  L.rcp_area = 2.0f / norm(cr);
  L.mat = m;
  return L;
}
// Triangle::samplePoint, Obj.h:127-136 (brace-init draws u0 then u1).
inline V3 tri_sample_point(const Tri& t, Rng& g) {
  float u0 = g.uniform();
  float u1 = g.uniform();
  float su0 = std::sqrt(u0);
  float b0 = 1.0f - su0;
  float b1 = u1 * su0;
  return add(add(scale(t.a, b0), scale(t.b, b1)), scale(t.c, 1.0f - b0 - b1));
}

// ----------------------------------------------------------------------------
// The oracle's own BVH (object-median split, conservative slab test).  It is a
// stand-in for Embree's BVH *as an accelerator only*: it must return exactly
// what the linear scan of Mesh::intersect returns (tests check that).
struct BNode {
  float lo[3], hi[3];
  int32_t left, right;   // children; leaf when count > 0
  uint32_t first, count; // range into order[]
};
struct Bvh {
  std::vector<BNode> nodes;
  std::vector<uint32_t> order;
};

void tri_bounds(const Tri& t, float lo[3], float hi[3]) {
  const float* p[3] = {&t.a.x, &t.b.x, &t.c.x};
  for (int k = 0; k < 3; k++) {
    lo[k] = std::min(p[0][k], std::min(p[1][k], p[2][k]));
    hi[k] = std::max(p[0][k], std::max(p[1][k], p[2][k]));
  }
}

int bvh_build_rec(Bvh& B, const std::vector<Tri>& tris, std::vector<float>& cent, uint32_t first,
                  uint32_t count, int depth) {
  int id = (int)B.nodes.size();
  B.nodes.push_back(BNode());
  float lo[3] = {kInf, kInf, kInf}, hi[3] = {-kInf, -kInf, -kInf};
  float clo[3] = {kInf, kInf, kInf}, chi[3] = {-kInf, -kInf, -kInf};
  for (uint32_t i = first; i < first + count; i++) {
    float l[3], h[3];
    tri_bounds(tris[B.order[i]], l, h);
    for (int k = 0; k < 3; k++) {
      lo[k] = std::min(lo[k], l[k]);
      hi[k] = std::max(hi[k], h[k]);
      float c = cent[3 * (size_t)B.order[i] + k];
      clo[k] = std::min(clo[k], c);
      chi[k] = std::max(chi[k], c);
    }
  }
  // Pad the box so that a hit the float triangle test accepts is never culled
  // by rounding in the slab test.
  for (int k = 0; k < 3; k++) {
    float pad = 1e-5f * std::max(std::fabs(lo[k]), std::fabs(hi[k])) + 1e-6f;
    B.nodes[id].lo[k] = lo[k] - pad;
    B.nodes[id].hi[k] = hi[k] + pad;
  }
  int axis = 0;
  for (int k = 1; k < 3; k++)
    if (chi[k] - clo[k] > chi[axis] - clo[axis]) axis = k;
  if (count <= 4 || depth > 60 || !(chi[axis] > clo[axis])) {
    B.nodes[id].left = B.nodes[id].right = -1;
    B.nodes[id].first = first;
    B.nodes[id].count = count;
    return id;
  }
  uint32_t mid = first + count / 2;
  std::nth_element(B.order.begin() + first, B.order.begin() + mid, B.order.begin() + first + count,
                   [&](uint32_t p, uint32_t q) {
                     float cp = cent[3 * (size_t)p + axis], cq = cent[3 * (size_t)q + axis];
                     return cp < cq || (cp == cq && p < q);
                   });
  B.nodes[id].count = 0;
  B.nodes[id].first = 0;
  int l = bvh_build_rec(B, tris, cent, first, mid - first, depth + 1);
  int r = bvh_build_rec(B, tris, cent, mid, first + count - mid, depth + 1);
  B.nodes[id].left = l;
  B.nodes[id].right = r;
  return id;
}

void bvh_build(Bvh& B, const std::vector<Tri>& tris) {
  B.nodes.clear();
  B.order.resize(tris.size());
  for (size_t i = 0; i < tris.size(); i++) B.order[i] = (uint32_t)i;
  if (tris.empty()) return;
  std::vector<float> cent(3 * tris.size());
  for (size_t i = 0; i < tris.size(); i++) {
    float l[3], h[3];
    tri_bounds(tris[i], l, h);
    for (int k = 0; k < 3; k++) cent[3 * i + k] = 0.5f * (l[k] + h[k]);
  }
  B.nodes.reserve(tris.size());
  bvh_build_rec(B, tris, cent, 0, (uint32_t)tris.size(), 0);
}

// Conservative slab test in double: true when the ray may touch the box for
// some t in [0, tmax].
inline bool box_maybe_hit(const BNode& n, const double o[3], const double inv[3], double tmax) {
  double t0 = 0.0, t1 = tmax;
  for (int k = 0; k < 3; k++) {
    double a = ((double)n.lo[k] - o[k]) * inv[k];
    double b = ((double)n.hi[k] - o[k]) * inv[k];
    if (a != a || b != b) continue;  // 0 * inf: origin on a slab plane of a parallel ray
    double mn = a < b ? a : b, mx = a < b ? b : a;
    if (mn > t0) t0 = mn;
    if (mx < t1) t1 = mx;
  }
  return t0 <= t1 * (1.0 + 1e-9) + 1e-12;
}

struct Scene {
  orc_config cfg;
  std::vector<Tri> tris;
  std::vector<uint32_t> mat_idx;
  std::vector<orc_material> mats;
  std::vector<orc_sphere> spheres;
  std::vector<orc_box> boxes;
  std::vector<orc_plane> planes;
  std::vector<Light> lights;
  Bvh bvh;
};

// Closest mesh hit with the rules of Mesh::intersect (Obj.h:184-195) and the
// Embree call it replaces (Scene.h:156-157: tnear = EPS, tfar = inf): a hit
// needs EPS < t && t < best; the first of equal t wins, i.e. the lowest index.
inline void mesh_closest_linear(const Scene& S, V3 o, V3 d, float& best_t, uint32_t& best_prim) {
  best_t = kInf;
  best_prim = 0xffffffffu;
  const float eps = S.cfg.eps;
  for (size_t i = 0; i < S.tris.size(); i++) {
    float t = tri_intersect(S.tris[i], o, d);
    if (eps < t && t < best_t) {
      best_t = t;
      best_prim = (uint32_t)i;
    }
  }
}
inline void mesh_closest_bvh(const Scene& S, V3 o, V3 d, float& best_t, uint32_t& best_prim) {
  best_t = kInf;
  best_prim = 0xffffffffu;
  if (S.bvh.nodes.empty()) return;
  const float eps = S.cfg.eps;
  double oo[3] = {o.x, o.y, o.z};
  double inv[3] = {1.0 / (double)d.x, 1.0 / (double)d.y, 1.0 / (double)d.z};
  int stack[128];
  int sp = 0;
  stack[sp++] = 0;
  while (sp) {
    const BNode& n = S.bvh.nodes[stack[--sp]];
    if (!box_maybe_hit(n, oo, inv, (double)best_t)) continue;
    if (n.left < 0) {
      for (uint32_t i = n.first; i < n.first + n.count; i++) {
        uint32_t p = S.bvh.order[i];
        float t = tri_intersect(S.tris[p], o, d);
        if (eps < t && t < kInf && (t < best_t || (t == best_t && p < best_prim))) {
          best_t = t;
          best_prim = p;
        }
      }
    } else {
      stack[sp++] = n.left;
      stack[sp++] = n.right;
    }
  }
}
// Any mesh hit with EPS < t && t < tfar.  (The reference asks Embree's
// rtcOccluded1 with tnear = EPS, tfar = distance - EPS, Scene.h:216-221; the
// interval ends are taken open, as in Mesh::intersect.)
inline bool mesh_any_linear(const Scene& S, V3 o, V3 d, float tfar) {
  const float eps = S.cfg.eps;
  for (size_t i = 0; i < S.tris.size(); i++) {
    float t = tri_intersect(S.tris[i], o, d);
    if (eps < t && t < tfar) return true;
  }
  return false;
}
inline bool mesh_any_bvh(const Scene& S, V3 o, V3 d, float tfar) {
  if (S.bvh.nodes.empty()) return false;
  const float eps = S.cfg.eps;
  double oo[3] = {o.x, o.y, o.z};
  double inv[3] = {1.0 / (double)d.x, 1.0 / (double)d.y, 1.0 / (double)d.z};
  int stack[128];
  int sp = 0;
  stack[sp++] = 0;
  while (sp) {
    const BNode& n = S.bvh.nodes[stack[--sp]];
    if (!box_maybe_hit(n, oo, inv, (double)tfar)) continue;
    if (n.left < 0) {
      for (uint32_t i = n.first; i < n.first + n.count; i++) {
        float t = tri_intersect(S.tris[S.bvh.order[i]], o, d);
        if (eps < t && t < tfar) return true;
      }
    } else {
      stack[sp++] = n.left;
      stack[sp++] = n.right;
    }
  }
  return false;
}

struct Hit {
  float t;
  uint32_t kind;  // 0 miss, 1 triangle, 2 sphere
  uint32_t prim;
  orc_material mat;
  V3 n;
};

// Scene::castRay, Scene.h:144-189.
inline void cast_ray(const Scene& S, V3 o, V3 d, bool use_bvh, Hit& h) {
  float tfar;
  uint32_t prim;
  if (use_bvh)
    mesh_closest_bvh(S, o, d, tfar, prim);
  else
    mesh_closest_linear(S, o, d, tfar, prim);
  h.t = kInf;
  h.kind = 0;
  h.prim = 0xffffffffu;
  h.n = mk(0, 0, 0);
  h.mat = orc_material{{0, 0, 0}, 0, 0};
  for (size_t i = 0; i < S.spheres.size(); i++) {  // Scene.h:165-173, first of equal t wins
    const orc_sphere& sp = S.spheres[i];
    float t = sphere_intersect(ld(sp.c), sp.r, o, d, S.cfg.eps);
    if (t > S.cfg.eps && t < h.t) {
      h.t = t;
      h.kind = 2;
      h.prim = (uint32_t)i;
      h.mat = sp.material;
      h.n = divs(sub(add(o, scale(d, t)), ld(sp.c)), sp.r);  // Sphere::normal, Obj.h:269-271
    }
  }
  // Boxes, then planes, continue the object list by the same rule.
  for (size_t i = 0; i < S.boxes.size(); i++) {
    const orc_box& b = S.boxes[i];
    float t = box_intersect(ld(b.min), ld(b.max), o, d, S.cfg.eps);
    if (t > S.cfg.eps && t < h.t) {
      h.t = t;
      h.kind = 3;
      h.prim = (uint32_t)i;
      h.mat = b.material;
      h.n = box_normal(ld(b.min), ld(b.max), add(o, scale(d, t)));
    }
  }
  for (size_t i = 0; i < S.planes.size(); i++) {
    const orc_plane& pl = S.planes[i];
    float t = plane_intersect(ld(pl.point), ld(pl.normal), o, d);
    if (t > S.cfg.eps && t < h.t) {
      h.t = t;
      h.kind = 4;
      h.prim = (uint32_t)i;
      h.mat = pl.material;
      h.n = ld(pl.normal);
    }
  }
  if (h.t < tfar) return;  // Scene.h:175-177: the sphere wins only when strictly closer
  if (prim != 0xffffffffu) {  // Scene.h:181-187
    const Tri& T = S.tris[prim];
    h.kind = 1;
    h.prim = prim;
    h.mat = S.mats[S.mat_idx[prim]];
    // Embree's Ng for a triangle (v0,v1,v2) is (v1-v0)x(v2-v0), not normalised.
    h.n = normalized(cross(sub(T.b, T.a), sub(T.c, T.a)));
  }
  h.t = tfar;  // Scene.h:188
}

// Scene::isOccluded, Scene.h:191-225.
inline bool is_occluded(const Scene& S, V3 o, V3 d, float distance, bool use_bvh) {
  const float eps = S.cfg.eps;
  if (distance < 2.0f * eps) return false;
  for (size_t i = 0; i < S.spheres.size(); i++) {
    const orc_sphere& sp = S.spheres[i];
    float t = sphere_intersect(ld(sp.c), sp.r, o, d, eps);
    if (t > eps && t < distance - eps) return true;
  }
  for (size_t i = 0; i < S.boxes.size(); i++) {
    float t = box_intersect(ld(S.boxes[i].min), ld(S.boxes[i].max), o, d, eps);
    if (t > eps && t < distance - eps) return true;
  }
  for (size_t i = 0; i < S.planes.size(); i++) {
    float t = plane_intersect(ld(S.planes[i].point), ld(S.planes[i].normal), o, d);
    if (t > eps && t < distance - eps) return true;
  }
  return use_bvh ? mesh_any_bvh(S, o, d, distance - eps) : mesh_any_linear(S, o, d, distance - eps);
}

// ----------------------------------------------------------------------------
// Cameras, Camera.h.  tanf's argument is evaluated in double (M_PI) and narrowed.
inline void camera_ray(const orc_config& c, float px, float py, Rng& g, V3& o, V3& d) {
  const float W = (float)c.width, H = (float)c.height;
  if (c.camera == 2) {  // OrthogonalCamera, Camera.h:73-79
    o = add(ld(c.cam_origin),
            mk((2.0f * px - W) / W * c.stretch_x, -(2.0f * py - H) / H * c.stretch_y, 0.f));
    d = mk(0.0f, 0.0f, -1.0f);
    return;
  }
  float tx = tanf((float)((double)c.fov * kPi / 180.0f / 2.0f));
  float ty = tanf((float)((double)(H / W * c.fov) * kPi / 180.0f / 2.0f));
  float x = (2.0f * px - W) / W * tx;
  float y = (2.0f * py - H) / H * ty;
  float z = -1.0f;
  o = ld(c.cam_origin);
  if (c.camera == 0) {  // PerspectiveCamera, Camera.h:18-25
    d = normalized(mk(x, -y, z));
    return;
  }
  // ThinLensCamera, Camera.h:44-57 with uniformSampleUnitDisk, :28-33
  d = mk(x, -y, z);
  float t = -c.focal_length / d.z;
  V3 focal = add(o, scale(d, t));
  float r = g.uniform();
  float theta = (float)((double)(g.uniform() * 2) * kPi);
  V3 disk = mk(r * std::cos(theta), r * std::sin(theta), 0.0f);
  o = add(o, scale(disk, c.lens_radius));
  d = normalized(sub(focal, o));
}

// ----------------------------------------------------------------------------
// pathTrace, Integrator.h:98-291.
inline V3 path_trace(const Scene& S, V3 o, V3 d, Rng& g, bool use_bvh, uint32_t& bounces,
                     uint64_t& nee) {
  const orc_config& c = S.cfg;
  V3 att = splat(1.0f);
  V3 color = splat(0.0f);
  float mis_brdf_pdf = -1.0f;
  float nee_pdf = 0.0f;  // uninitialised in the reference; only read after a NEE bounce set it
  bounces = 0;
  while (true) {
    Hit h;
    cast_ray(S, o, d, use_bvh, h);
    bounces++;
    if (std::isinf(h.t)) {  // :113-116
      if (c.sky) color = add(color, mul(att, sky_emission(d)));  // :114 un-commented
      break;
    }
    V3 hp = add(scale(d, h.t), o);  // :119
    V3 n = h.n;
    V3 emission = splat(h.mat.emission);  // :133
    if (max3(emission) > 0.0f) {          // :135-151
      if (mis_brdf_pdf > 0.0f) {
        float sa = std::fabs(-dot(d, n)) / nee_pdf / h.t / h.t;
        V3 num = scale(scale(mul(emission, att), mis_brdf_pdf), mis_brdf_pdf);
        color = add(color, divs(num, 1.0f / sa / sa + mis_brdf_pdf * mis_brdf_pdf));
      } else {
        color = add(color, mul(emission, att));
      }
    }
    o = hp;  // :184
    // Russian roulette, :189-194.  std::min(a,b) returns b iff b < a.
    float amax = max3(att);
    float p = (c.rr_cap < amax) ? c.rr_cap : amax;
    if (g.uniform() > p) break;
    // Deviation (5): a path that has made max_bounces closest-hit queries ends here.
    if (c.max_bounces && bounces >= c.max_bounces) break;
    att = divs(att, p);

    V3 albedo = ld(h.mat.albedo);
    switch (h.mat.surface) {
      case 2: {  // reflective, :198-205
        float cos_t = dot(d, n);
        d = sub(d, scale(scale(n, cos_t), 2.0f));
        att = mul(att, albedo);
        mis_brdf_pdf = -1.0f;
        break;
      }
      case 0: {  // diffuse, :206-247
        bool did_nee = false;
        if (!S.lights.empty() && h.mat.emission < c.nee_threshold) {
          // Scene::sampleRandomPointOnLight, Scene.h:227-235, with deviation (2).
          size_t nl = S.lights.size();
          float ul = g.uniform();
          size_t li = (size_t)(ul * (float)nl);
          if (li >= nl) li = nl - 1;
          const Light& L = S.lights[li];
          V3 sp = tri_sample_point(L.t, g);
          nee_pdf = L.rcp_area / (float)nl;
          float distance = norm(sub(sp, o));
          d = divs(sub(sp, o), distance);
          did_nee = true;
          nee++;
          if (!is_occluded(S, o, d, distance, use_bvh)) {
            float sa = std::fabs(-dot(d, L.n)) / nee_pdf / distance / distance;
            if (sa != 0.0f) {
              // :222-223.  The denominator is evaluated in double (abs(..)/M_PI),
              // then narrowed to float by operator/(Vec3,float).
              V3 v = scale(mul(scale(att, L.mat.emission), albedo), std::fabs(dot(n, d)));
              v = divs(v, (float)kPi);
              v = scale(v, 1.0f);
              v = divs(v, sa);
              double den = (double)(1.0f / sa / sa) +
                           (double)std::fabs(dot(d, n)) / kPi * (double)std::fabs(dot(d, n)) / kPi;
              color = add(color, divs(v, (float)den));
            }
          }
        }
        M3 R = rot_matrix_vectors(n, mk(0.0f, 0.0f, 1.0f));  // :241
        d = matvec(R, cosine_sample_hemisphere(g));          // :242
        float dn = dot(d, n);
        float cos_t = (dn < 0.0f) ? 0.0f : dn;  // std::max(dn, 0.0f): returns 0 iff dn < 0
        mis_brdf_pdf = did_nee ? (float)((double)cos_t / kPi) : -1.0f;  // :244
        att = mul(att, albedo);                                          // :245
        break;
      }
      case 1: {  // specular (glass), :248-278
        float ior = 1.0f / 2.0f;
        float cos_t1 = -dot(n, d);
        bool from_outside = cos_t1 > 0.0f;
        if (!from_outside) {
          cos_t1 *= -1;
          ior = 1.0f / ior;
          n = neg(n);
        }
        float cos_t2_sqr = 1.0f - ior * ior * (1.0f - cos_t1 * cos_t1);
        // Short-circuit: no draw on total internal reflection.
        if (cos_t2_sqr >= 0.0f &&
            g.uniform() > schlick(ior, from_outside ? cos_t1 : std::sqrt(cos_t2_sqr))) {
          float cos_t2 = std::sqrt(cos_t2_sqr);
          d = add(scale(d, ior), scale(n, ior * cos_t1 - cos_t2));
          d = divs(d, norm(d));
        } else {
          d = add(d, scale(scale(n, cos_t1), 2));
        }
        att = mul(att, albedo);
        mis_brdf_pdf = -1.0f;
        break;
      }
      default:  // varnish: no case in the switch -> direction and throughput unchanged
        break;
    }
  }
  return color;
}

// rayTraceNormals, Integrator.h:87-96
inline V3 trace_normals(const Scene& S, V3 o, V3 d, bool use_bvh, uint32_t& bounces) {
  Hit h;
  cast_ray(S, o, d, use_bvh, h);
  bounces = 1;
  if (std::isinf(h.t)) return sky_emission(d);
  return add(scale(h.n, 0.5f), splat(0.5f));
}

// One sample of render<I>'s inner loop, Raytracer.cpp:246-251, with deviations
// (1) and (3).  px/py are int + float - 0.5f, evaluated in float.
inline V3 sample_pixel(const Scene& S, uint32_t px, uint32_t py, uint32_t sample, uint64_t seed,
                       V3 cam_offset, bool use_bvh, uint32_t& bounces, uint64_t& nee) {
  Rng g{path_seed(px + py * S.cfg.width, sample, seed)};
  if (S.cfg.rng_streams) g = stream_rng(px + py * S.cfg.width, sample, seed);
  float jx = g.uniform();
  float jy = g.uniform();
  float fx = (float)(int)px + jx - 0.5f;
  float fy = (float)(int)py + jy - 0.5f;
  V3 o, d;
  camera_ray(S.cfg, fx, fy, g, o, d);
  o = add(o, cam_offset);  // ray.o += pos, Raytracer.cpp:250
  if (S.cfg.integrator == 1) return trace_normals(S, o, d, use_bvh, bounces);
  return path_trace(S, o, d, g, use_bvh, bounces, nee);
}

}  // namespace

struct orc_scene {
  Scene s;
};

extern "C" {

uint32_t orc_pcg_hash(uint32_t input) { return pcg_hash(input); }
void orc_pcg_rand(uint32_t* state, uint32_t* out, size_t n) {
  Rng g{*state};
  for (size_t i = 0; i < n; i++) out[i] = g.next();
  *state = g.s;
}
void orc_pcg_uniform(uint32_t* state, float* out, size_t n) {
  Rng g{*state};
  for (size_t i = 0; i < n; i++) out[i] = g.uniform();
  *state = g.s;
}
uint32_t orc_path_seed(uint32_t pixel_index, uint32_t sample_index, uint64_t seed) {
  return path_seed(pixel_index, sample_index, seed);
}

void orc_rot_matrix_vectors(const float a[3], const float b[3], float m_out[9]) {
  M3 r = rot_matrix_vectors(ld(a), ld(b));
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) m_out[3 * i + j] = r.m[i][j];
}
void orc_mat_vec(const float m[9], const float v[3], float out[3]) {
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = m[3 * i + j];
  st(out, matvec(r, ld(v)));
}
void orc_cosine_sample_hemisphere(uint32_t* rng_state, float out[3]) {
  Rng g{*rng_state};
  st(out, cosine_sample_hemisphere(g));
  *rng_state = g.s;
}
float orc_schlick(float r, float cos_t) { return schlick(r, cos_t); }
void orc_sky_emission(const float d[3], float out[3]) { st(out, sky_emission(ld(d))); }
float orc_triangle_intersect(const float a[3], const float b[3], const float c[3], const float o[3],
                             const float d[3]) {
  Tri t{ld(a), ld(b), ld(c)};
  return tri_intersect(t, ld(o), ld(d));
}
float orc_sphere_intersect(const float c[3], float r, const float o[3], const float d[3]) {
  return sphere_intersect(ld(c), r, ld(o), ld(d), 1.0e-3f);
}
float orc_box_intersect(const float mn[3], const float mx[3], const float o[3], const float d[3]) {
  return box_intersect(ld(mn), ld(mx), ld(o), ld(d), 1.0e-3f);
}
void orc_box_normal(const float mn[3], const float mx[3], const float p[3], float n_out[3]) {
  st(n_out, box_normal(ld(mn), ld(mx), ld(p)));
}
float orc_plane_intersect(const float point[3], const float normal[3], const float o[3], const float d[3]) {
  return plane_intersect(ld(point), ld(normal), ld(o), ld(d));
}
float orc_triangle_sample_point(const float a[3], const float b[3], const float c[3],
                                uint32_t* rng_state, float p_out[3], float n_out[3]) {
  Tri t{ld(a), ld(b), ld(c)};
  orc_material m{{0, 0, 0}, 0, 0};
  Light L = make_light(t, m);
  Rng g{*rng_state};
  st(p_out, tri_sample_point(t, g));
  st(n_out, L.n);
  *rng_state = g.s;
  return L.rcp_area;
}
void orc_camera_ray(const orc_config* cfg, float px, float py, uint32_t* rng_state, float o_out[3],
                    float d_out[3]) {
  Rng g{*rng_state};
  V3 o, d;
  camera_ray(*cfg, px, py, g, o, d);
  st(o_out, o);
  st(d_out, d);
  *rng_state = g.s;
}
void orc_img_update(float pixel[3], uint32_t sample_count, const float color[3]) {
  // (get*n + color) / (n+1), Raytracer.cpp:80
  V3 v = divs(add(scale(ld(pixel), (float)sample_count), ld(color)), (float)(sample_count + 1));
  st(pixel, v);
}

orc_scene* orc_scene_create(const float* tris, size_t ntri, const uint32_t* mat_idx,
                            const orc_material* mats, size_t nmat, const orc_sphere* spheres,
                            size_t nsph, const orc_box* boxes, size_t nbox, const orc_plane* planes,
                            size_t nplane, const orc_config* cfg) {
  orc_scene* h = new orc_scene();
  Scene& S = h->s;
  S.cfg = *cfg;
  S.tris.resize(ntri);
  S.mat_idx.assign(mat_idx, mat_idx + ntri);
  S.mats.assign(mats, mats + nmat);
  if (nsph) S.spheres.assign(spheres, spheres + nsph);
  if (nbox) S.boxes.assign(boxes, boxes + nbox);
  if (nplane) S.planes.assign(planes, planes + nplane);
  for (size_t i = 0; i < ntri; i++) {
    S.tris[i] = Tri{ld(tris + 9 * i), ld(tris + 9 * i + 3), ld(tris + 9 * i + 6)};
    if (S.mat_idx[i] >= nmat) {
      delete h;
      return nullptr;
    }
  }
  // Scene::commit, Scene.h:129-141: every triangle whose material emission exceeds
  // the threshold becomes a light, in triangle order.
  for (size_t i = 0; i < ntri; i++) {
    const orc_material& m = S.mats[S.mat_idx[i]];
    if (m.emission > S.cfg.nee_threshold) S.lights.push_back(make_light(S.tris[i], m));
  }
  bvh_build(S.bvh, S.tris);
  return h;
}
void orc_scene_destroy(orc_scene* s) { delete s; }
size_t orc_scene_num_lights(const orc_scene* s) { return s->s.lights.size(); }

void orc_trace_closest(const orc_scene* sc, const float* rays, size_t n, int use_bvh, uint32_t* kind,
                       uint32_t* prim, float* t, float* normal) {
  const Scene& S = sc->s;
#pragma omp parallel for schedule(static)
  for (long long i = 0; i < (long long)n; i++) {
    Hit h;
    cast_ray(S, ld(rays + 6 * i), ld(rays + 6 * i + 3), use_bvh != 0, h);
    kind[i] = h.kind;
    prim[i] = h.prim;
    t[i] = h.t;
    if (normal) st(normal + 3 * i, h.n);
  }
}
void orc_trace_occluded(const orc_scene* sc, const float* rays, const float* dist, size_t n,
                        int use_bvh, uint8_t* occluded) {
  const Scene& S = sc->s;
#pragma omp parallel for schedule(static)
  for (long long i = 0; i < (long long)n; i++)
    occluded[i] = is_occluded(S, ld(rays + 6 * i), ld(rays + 6 * i + 3), dist[i], use_bvh != 0) ? 1 : 0;
}

void orc_path_trace(const orc_scene* sc, uint32_t pixel_index, uint32_t sample_index, uint64_t seed,
                    const float cam_offset[3], int use_bvh, float color_out[3],
                    uint32_t* bounces_out) {
  const Scene& S = sc->s;
  uint32_t b = 0;
  uint64_t nee = 0;
  V3 off = cam_offset ? ld(cam_offset) : mk(0, 0, 0);
  V3 c = sample_pixel(S, pixel_index % S.cfg.width, pixel_index / S.cfg.width, sample_index, seed,
                      off, use_bvh != 0, b, nee);
  st(color_out, c);
  if (bounces_out) *bounces_out = b;
}

// render<I>, Raytracer.cpp:236-277: OpenMP parallel-for over rows with dynamic
// scheduling; per-sample work as sample_pixel.  Samples are added into sum_rgb
// in sample order per pixel (deviation (4)).
void orc_render(const orc_scene* sc, uint32_t spp, uint32_t sample_base, uint64_t seed,
                const float cam_offset[3], int use_bvh, int nthreads, uint32_t row0, uint32_t row1,
                float* sum_rgb, orc_stats* stats) {
  const Scene& S = sc->s;
  const uint32_t W = S.cfg.width, H = S.cfg.height;
  if (row1 == 0 || row1 > H) row1 = H;
  V3 off = cam_offset ? ld(cam_offset) : mk(0, 0, 0);
  uint64_t tot_b = 0, tot_nee = 0, tot_nan = 0;
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
  nthreads = 1;
#endif
#pragma omp parallel for schedule(dynamic) num_threads(nthreads) reduction(+ : tot_b, tot_nee, tot_nan)
  for (long long py = row0; py < (long long)row1; py++) {
    for (uint32_t px = 0; px < W; px++) {
      float* dst = sum_rgb + 3 * ((size_t)px + (size_t)py * W);
      for (uint32_t s = 0; s < spp; s++) {
        uint32_t b = 0;
        uint64_t nee = 0;
        V3 c = sample_pixel(S, px, (uint32_t)py, sample_base + s, seed, off, use_bvh != 0, b, nee);
        tot_b += b;
        tot_nee += nee;
        if (!std::isfinite(c.x) || !std::isfinite(c.y) || !std::isfinite(c.z)) tot_nan++;
        dst[0] += c.x;
        dst[1] += c.y;
        dst[2] += c.z;
      }
    }
  }
  if (stats) {
    stats->paths += (uint64_t)(row1 - row0) * W * spp;
    stats->bounces += tot_b;
    stats->nee_samples += tot_nee;
    stats->nan_paths += tot_nan;
  }
}

}  // extern "C"
