// shading.h -- one bounce of pathTrace (reference Integrator.h:98-291): sky, Schlick, hemisphere sampling, the primary
// ray of a sample (Camera.h, Raytracer.cpp:246-250), path state in the queues, emission + MIS, Russian roulette, the
// surface's switch with next-event estimation, and where a finished path's colour goes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "traversal.h"

namespace rtx {

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

}  // namespace rtx
