/*
 * rt_oracle.h -- C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is a scalar CPU restatement of the
 * reference renderer's radiance loop.  It exists to CHECK the HIP library and
 * to be timed as the CPU baseline.  Nothing in the shipped product path may
 * include, link or load it: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py do.
 *
 * Pinning status (what ties this restatement to the reference itself):
 *   - Pcg (rand_pcg / pcg_hash / Uniform): PINNED.  Checked bit-for-bit
 *     against the reference's own Sampler.h compiled unchanged into
 *     oracle/_ref/libref_pcg.so (see oracle/Makefile) and against the golden
 *     vectors generated from that build (tests/golden/pcg_kat.json).
 *   - EPS and NEE_EMISSION_THRESHOLD: PINNED.  The reference's globals.h (macros only) is
 *     compiled unchanged into the same library; tests/golden/pcg_kat.json records the values.
 *   - Oven-box radiance (closed room, emission 0.5, albedo 0.5 -> 1.0):
 *     PINNED to the analytic answer stated at Raytracer.cpp:203-214; the same room
 *     with a glass and a mirror sphere of albedo 1 inside keeps that answer and
 *     pins the energy balance of the specular branches and of Russian roulette.
 *   - Everything else (Vec3/Matrix3 arithmetic, camera, triangle/sphere
 *     intersection, castRay/isOccluded rules, light sampling, pathTrace):
 *     PARITY UNPINNED.  Vec3.h includes SFML and Scene.h/Obj.h include Embree 3;
 *     neither exists in this image and neither is vendored, so none of the
 *     reference's own rendering files can be compiled here without writing
 *     stand-ins for the missing headers.  The reference ships no tests, fixtures
 *     or golden data for this path.  These functions follow the reference source
 *     line by line (each cites the file:line it restates).
 *   - Outside this library, the file formats either side of the loop: PINNED.  The
 *     third-party sources the reference vendors and calls for them (tiny_obj_loader.h in
 *     Scene::load_mesh, tinyexr.h + miniz in Img::saveEXR) are self-contained and compile
 *     unchanged into oracle/_ref/libref_io.so (ref_io_wrap.cpp); tests/test_formats.py checks
 *     the host layer's OBJ front end and EXR writer against it and against the fixtures
 *     generated from it (tests/golden/formats).
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Material.h:4-15 -- POD, 20 bytes. surface: 0 diffuse, 1 specular (glass),
 * 2 reflective (mirror), 3 varnish (no case in the integrator's switch). */
typedef struct {
  float albedo[3];
  float emission;
  int32_t surface;
} orc_material;

/* Obj.h:249-256 */
typedef struct {
  float c[3];
  float r;
  orc_material material;
} orc_sphere;

/* The commented-out class Box (Obj.h:35-93). */
typedef struct {
  float min[3];
  float max[3];
  orc_material material;
} orc_box;

/* Plane(point, normal): only call sites exist in the reference (Raytracer.cpp:184-221);
 * t = ((point - o).n)/(d.n), surface normal = `normal` as given.  Build-defined. */
typedef struct {
  float point[3];
  float normal[3];
  orc_material material;
} orc_plane;

/* Compile-time constants of the reference turned into fields
 * (globals.h:3-8, Integrator.h:12, Raytracer.cpp:145, Camera.h:23,37-39,64-65). */
typedef struct {
  uint32_t width, height;
  float fov;            /* degrees; PerspectiveCamera(50.0f) / ThinLens 60 */
  float cam_origin[3];  /* (0,1,3) */
  float eps;            /* EPS 1e-3 */
  float nee_threshold;  /* NEE_EMISSION_THRESHOLD 0.05 */
  float rr_cap;         /* MAX_BOUNCE_PROB 0.99 */
  int32_t sky;          /* 1 = add attenuation*skyEmission(d) on a miss (Integrator.h:114 un-commented) */
  uint32_t max_bounces; /* 0 = unbounded, as the reference */
  int32_t camera;       /* 0 perspective, 1 thin lens, 2 orthogonal */
  int32_t integrator;   /* 0 pathTrace, 1 rayTraceNormals */
  float focal_length, lens_radius; /* thin lens: 2.4, 0.09 */
  float stretch_x, stretch_y;      /* orthogonal */
  int32_t rng_streams;  /* 0: the reference's Pcg; 1: an odd increment of its own per sample (the library's opt-in
                           mode for renders of more than 2^32 samples; not in the reference) */
} orc_config;

typedef struct {
  uint64_t paths;       /* primary samples */
  uint64_t bounces;     /* iterations of the loop at Integrator.h:108 = closest-hit queries */
  uint64_t nee_samples; /* calls of isOccluded (Integrator.h:217) */
  uint64_t nan_paths;   /* paths whose returned colour has a NaN/Inf component */
} orc_stats;

typedef struct orc_scene orc_scene;

/* ---- Sampler.h ---- */
uint32_t orc_pcg_hash(uint32_t input);
/* Runs n draws of rand_pcg from *state (updated in place). */
void orc_pcg_rand(uint32_t* state, uint32_t* out, size_t n);
void orc_pcg_uniform(uint32_t* state, float* out, size_t n);
/* The build's counter-based stream seed for (pixel, sample, seed). */
uint32_t orc_path_seed(uint32_t pixel_index, uint32_t sample_index, uint64_t seed);

/* ---- per-function restatements (for unit vectors) ---- */
void orc_rot_matrix_vectors(const float a[3], const float b[3], float m_out[9]);
void orc_mat_vec(const float m[9], const float v[3], float out[3]);
void orc_cosine_sample_hemisphere(uint32_t* rng_state, float out[3]);
float orc_schlick(float r, float cos_t);
void orc_sky_emission(const float d[3], float out[3]);
float orc_triangle_intersect(const float a[3], const float b[3], const float c[3],
                             const float o[3], const float d[3]);
float orc_sphere_intersect(const float c[3], float r, const float o[3], const float d[3]);
float orc_box_intersect(const float mn[3], const float mx[3], const float o[3], const float d[3]);
void orc_box_normal(const float mn[3], const float mx[3], const float p[3], float n_out[3]);
float orc_plane_intersect(const float point[3], const float normal[3], const float o[3], const float d[3]);
/* Triangle ctor + samplePoint: returns rcp_area; writes the point and n. */
float orc_triangle_sample_point(const float a[3], const float b[3], const float c[3],
                                uint32_t* rng_state, float p_out[3], float n_out[3]);
void orc_camera_ray(const orc_config* cfg, float px, float py, uint32_t* rng_state,
                    float o_out[3], float d_out[3]);
/* Img::update running mean (Raytracer.cpp:78-81). */
void orc_img_update(float pixel[3], uint32_t sample_count, const float color[3]);

/* ---- scene ---- */
/* tris: ntri*9 floats (a,b,c), prim id = index (Obj.h:218-226 order). */
orc_scene* orc_scene_create(const float* tris, size_t ntri, const uint32_t* mat_idx,
                            const orc_material* mats, size_t nmat,
                            const orc_sphere* spheres, size_t nsph, const orc_box* boxes, size_t nbox,
                            const orc_plane* planes, size_t nplane, const orc_config* cfg);
void orc_scene_destroy(orc_scene*);
size_t orc_scene_num_lights(const orc_scene*);

/* kind: 0 miss, 1 triangle, 2 sphere, 3 box, 4 plane. prim: index in its own array.
 * use_bvh = 0 runs the first-party linear scan (Obj.h:184-195), 1 the oracle's BVH. */
void orc_trace_closest(const orc_scene*, const float* rays /* n*6 */, size_t n, int use_bvh,
                       uint32_t* kind, uint32_t* prim, float* t, float* normal /* n*3 or NULL */);
void orc_trace_occluded(const orc_scene*, const float* rays /* n*6 */, const float* dist, size_t n,
                        int use_bvh, uint8_t* occluded);

/* One path: returns colour; *bounces_out = closest-hit queries. */
void orc_path_trace(const orc_scene*, uint32_t pixel_index, uint32_t sample_index, uint64_t seed,
                    const float cam_offset[3], int use_bvh, float color_out[3],
                    uint32_t* bounces_out);

/* Adds spp samples/pixel (sample indices sample_base .. sample_base+spp-1) into
 * sum_rgb[W*H*3] (idx = px + py*W, Raytracer.cpp:57-59).  nthreads<=0: OpenMP default.
 * rows [row0,row1) only (row1 = 0 -> all rows). */
void orc_render(const orc_scene*, uint32_t spp, uint32_t sample_base, uint64_t seed,
                const float cam_offset[3], int use_bvh, int nthreads, uint32_t row0, uint32_t row1,
                float* sum_rgb, orc_stats* stats);

#ifdef __cplusplus
}
#endif
#endif
