/*
 * rtx_capi.h -- C ABI of the MI355X wavefront path tracer (librtx_hip.so).
 *
 * This is the drop-in boundary for the reference renderer's Monte-Carlo loop.
 * The reference has no plugin or FFI interface; its seam is the call
 * `render(img, pathTrace)` at Raytracer.cpp:319 (template render<I>,
 * Raytracer.cpp:236-286), which reads the globals `scene` (:148), `camera`
 * (:145), `gens` (:233) and `pos` (:235) and writes `Img::pixels` (:49).
 * Each entry point below cites the reference code it replaces.
 *
 * Conventions
 *  - Plain C: pointers and sizes only.  No C++ or torch types cross the boundary.
 *  - Every function returns RTX_OK (0) or a negative rtx_status; it never calls
 *    exit() (the reference does, Scene.h:66) and never throws.
 *    rtx_last_error() returns a human-readable message for the last failure on
 *    the calling thread.
 *  - The library is HIP-only.  There is no CPU fallback: rtx_create fails with
 *    RTX_ERR_NO_DEVICE when no gfx950 device is usable.
 *  - A context is used from one thread at a time (the reference calls render()
 *    from exactly one render thread, Raytracer.cpp:462).
 *  - "device pointer" arguments are addresses in the memory of the context's GPU
 *    (hipMalloc, or a torch tensor's data_ptr()).  "host pointer" arguments are
 *    ordinary memory, borrowed only for the duration of the call.
 *  - `stream` arguments are a hipStream_t passed as void*; NULL = the default
 *    stream.  Calls that take a stream only enqueue work unless stated otherwise.
 */
#ifndef RTX_CAPI_H
#define RTX_CAPI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  RTX_OK = 0,
  RTX_ERR_INVALID_ARGUMENT = -1,
  RTX_ERR_NO_DEVICE = -2,
  RTX_ERR_HIP = -3,
  RTX_ERR_OUT_OF_MEMORY = -4,
  RTX_ERR_NO_SCENE = -5,
  RTX_ERR_UNSUPPORTED = -6,
  RTX_ERR_COMM = -7
} rtx_status;

/* Surface enum, Material.h:4-9 (same values). */
enum { RTX_DIFFUSE = 0, RTX_SPECULAR = 1, RTX_REFLECTIVE = 2, RTX_VARNISH = 3 };

/* struct Material, Material.h:11-15: same 20-byte layout. emission is a scalar
 * broadcast to RGB (Integrator.h:133). */
typedef struct {
  float albedo[3];
  float emission;
  int32_t surface;
} rtx_material;

/* class Sphere, Obj.h:249-256 (centre, radius, material). */
typedef struct {
  float c[3];
  float r;
  rtx_material material;
} rtx_sphere;

/* Axis-aligned box, the reference's commented-out class Box (Obj.h:35-93). */
typedef struct {
  float min[3];
  float max[3];
  rtx_material material;
} rtx_box;

/* Plane through `point` with normal `normal`: the reference's scenes construct
 * `new Plane(point, normal)` (Raytracer.cpp:184-221) but the class itself is not in its
 * repository; t = ((point - o).n)/(d.n) and the surface normal is `normal` as given. */
typedef struct {
  float point[3];
  float normal[3];
  rtx_material material;
} rtx_plane;

/* The reference's compile-time constants as run-time fields:
 * WIDTH/HEIGHT/EPS/NEE_EMISSION_THRESHOLD globals.h:3-8, MAX_BOUNCE_PROB
 * Integrator.h:12, camera type and fov Raytracer.cpp:145-147, camera origin
 * Camera.h:23, thin-lens defaults Camera.h:37-39, orthogonal stretch Camera.h:64-65. */
typedef struct {
  uint32_t width, height;
  float fov;             /* degrees */
  float cam_origin[3];   /* (0,1,3) in the reference */
  float eps;             /* 1e-3 */
  float nee_threshold;   /* 0.05 */
  float rr_cap;          /* 0.99 */
  int32_t sky;           /* 0: a miss adds nothing (the live reference, Integrator.h:113-116);
                            1: a miss adds attenuation*skyEmission(d) (Integrator.h:114 un-commented) */
  uint32_t max_bounces;  /* 0 = unbounded (the reference has no cap) */
  int32_t camera;        /* 0 PerspectiveCamera, 1 ThinLensCamera, 2 OrthogonalCamera */
  int32_t integrator;    /* 0 pathTrace (Integrator.h:98), 1 rayTraceNormals (Integrator.h:87) */
  float focal_length, lens_radius; /* thin lens */
  float stretch_x, stretch_y;      /* orthogonal */
  int32_t device;        /* HIP device ordinal; -1 = the calling thread's current device */
  uint32_t pool_paths;   /* capacity of the in-flight path pool; 0 = library default (16 Mi paths) */
  uint32_t finish_paths; /* once the samples are exhausted and at most this many paths are left, they
                            are run to their end by one launch (one path per lane) instead of further
                            rounds of small kernels; 0 = library default (524288), 1 = never */
  uint32_t result_buffer_mib; /* budget of the per-sample result buffer (12 bytes per sample); a call
                            with more samples is rendered in several chunks; 0 = default (8192) */
  int32_t primary_rays;  /* how the kernels get the primary rays of new samples: 0 = the library chooses
                            (recomputed from the sample number by each kernel that needs them where that
                            is cheaper than a round trip through memory), 1 = always written by the
                            generate kernel and read back.  Both give the same bits. */
  int32_t fused_trace;   /* closest-hit search and shading of a bounce: 0 = the library chooses (one kernel
                            where the tree is small enough to be staged in LDS: the search then costs less
                            than passing the ray and the hit through memory; two kernels otherwise),
                            1 = always two kernels, 2 = as 0 today.  All give the same bits. */
  int32_t rng_streams;   /* 0 = the reference's generator, bit for bit: 32 bits of state on ONE cycle of 2^32 values
                            (Sampler.h:11-42), every sample a starting point on that cycle -- 2^32 distinct starting
                            points in all, so a render of more than 2^32 samples (3840x2160 at 4096 per pixel is
                            3.4e10) repeats samples, and long paths of different samples share stretches of the
                            cycle.  1 = the same generator with an odd increment of its own per sample: state and
                            increment are the two halves of a 64-bit bijective mix of (sample index, pixel, seed), so
                            that the samples of such a render start from distinct (state, increment) pairs (about
                            31 coinciding pairs expected among 3.4e10), each on its own cycle.  Same estimator, other
                            random numbers: images agree with mode 0 in distribution, not bit for bit.  No counterpart
                            in the reference.  Costs 4 bytes more per stored path and the kernels' general build. */
} rtx_config;

/* Scene geometry, borrowed for the duration of rtx_set_scene.
 * triangles: num_triangles*9 floats (a,b,c); the primitive id of a triangle is
 * its index here, which is the order Mesh::triangles has and EmbreeMesh copies
 * (Obj.h:218-226).  material_index[i] selects materials[] for triangle i
 * (Mesh::material_idxs, Obj.h:168). */
typedef struct {
  const float* triangles;
  size_t num_triangles;
  const uint32_t* material_index;
  const rtx_material* materials;
  size_t num_materials;
  const rtx_sphere* spheres; /* Scene::spheres, Scene.h:37; may be NULL when num_spheres==0 */
  size_t num_spheres;
  /* Analytic boxes and planes.  castRay tests them after the spheres, by the rule it applies
   * to spheres (Scene.h:165-177): t > EPS, the first of equal t wins, and an analytic object
   * beats the mesh only when strictly closer.  Like spheres they are never sampled as lights. */
  const rtx_box* boxes;
  size_t num_boxes;
  const rtx_plane* planes;
  size_t num_planes;
} rtx_scene;

typedef struct {
  uint64_t paths;        /* primary samples started */
  uint64_t bounces;      /* closest-hit queries = iterations of the loop at Integrator.h:108 */
  uint64_t nee_samples;  /* light samples taken (one isOccluded call each in the reference, Integrator.h:217) */
  uint64_t shadow_rays;  /* shadow rays actually traced (light samples with a zero contribution are dropped) */
  uint64_t iterations;   /* wavefront iterations (one intersect+shade round each) */
  uint64_t launches;     /* kernel launches */
  /* Filled only while profiling is on (rtx_set_profiling): device time per stage,
   * measured with HIP events on the render stream, and launches timed. */
  double ms_generate, ms_intersect, ms_shade, ms_shadow, ms_finish, ms_reduce, ms_other;
  uint64_t timed_launches_intersect, timed_launches_shade, timed_launches_shadow,
      timed_launches_generate;
  uint64_t nan_paths;        /* finished paths whose colour holds a NaN or an infinity.  The reference can
                                produce them (sqrt of a slightly negative number, Integrator.h:24) and paints
                                them magenta under RT_DEBUG (Raytracer.cpp:259-264); they are counted here,
                                not masked: such a sample goes into the sum as it is */
  uint64_t truncated_paths;  /* paths the drain kernel gave up on after 65536 bounces (the reference's loop
                                has no bound, Integrator.h:108; at rr_cap 0.99 such a path has probability
                                below 1e-286); their colour so far is stored as if they had ended */
  uint64_t drain_launches;   /* chunks that ended in the drain kernel (one launch that runs the last paths to
                                their ends) */
  uint64_t overlapped_sums;  /* ... of which the ordered sum ran beside that kernel on a second stream */
  uint64_t stored_paths;     /* paths a bounce wrote to a queue and the next launch read back (a bounce follows a fresh
                                sample, such a round trip through memory, or, where a launch makes two bounces, the
                                bounce before it in the same lane) */
  uint64_t device_allocations; /* hipMalloc calls this library has made in this process so far, over all contexts
                                (not cleared by rtx_reset_stats): unchanged across a call means the call allocated
                                no device memory */
} rtx_stats;

typedef struct {
  uint32_t num_nodes, num_triangles, num_lights, num_spheres, max_depth, stack_entries;
  double build_ms;
  uint32_t bvh_width;    /* children per BVH node: 2 (scenes staged in LDS) or 4 (quantised, large scenes) */
  uint32_t stack_bound;  /* worst-case traversal stack entries (those beyond stack_entries spill to memory) */
} rtx_scene_info;

typedef struct rtx_ctx rtx_ctx;

/* Message for the last failing call on this thread ("" if none). */
const char* rtx_last_error(void);

/* Replaces the construction of the globals camera/scene/gens (Raytracer.cpp:145,148,233):
 * selects the GPU, allocates the path pool and queues. */
int rtx_create(const rtx_config* cfg, rtx_ctx** out);
void rtx_destroy(rtx_ctx* ctx);

/* Replaces Scene::addMesh + Scene::commit (Scene.h:117-142): uploads the
 * geometry, builds the BVH (in place of rtcCommitScene, Scene.h:123) and
 * extracts the light list with the rule of Scene.h:129-141 (every triangle
 * whose material emission exceeds nee_threshold, in triangle order).
 * Synchronous. */
int rtx_set_scene(rtx_ctx* ctx, const rtx_scene* scene);
int rtx_get_scene_info(rtx_ctx* ctx, rtx_scene_info* out);

/* Replaces one or more calls of render(img, pathTrace) (Raytracer.cpp:236-277,319):
 * adds `spp` samples per pixel -- sample indices sample_base .. sample_base+spp-1 --
 * of rows [row0,row1) into accum_dev, a device buffer of width*height*3 floats
 * holding per-pixel SUMS (idx = px + py*width, the layout of Img::pixels,
 * Raytracer.cpp:49,57-59).  Each pixel's samples are added in sample-index order, so
 * the result is bit-reproducible from run to run.  row1 = 0 means all rows.  cam_offset is the global
 * `pos` added to every ray origin (Raytracer.cpp:250); NULL = (0,0,0).
 * The per-path random stream depends only on (pixel, sample index, seed), so
 * disjoint sample ranges rendered on different GPUs add up to the same image as
 * one GPU rendering the union.  The generator is the reference's Pcg (Sampler.h:11-42): 32 bits
 * of state on one cycle of 2^32 values.  The seed is folded to 32 bits (low word xor high word)
 * and a path starts at state pcg_hash(pixel ^ pcg_hash(sample ^ pcg_hash(seed32))): there are 2^32
 * distinct starting points in all, paths are shifted copies of one sequence (as the reference's
 * eight per-thread generators are), and a render of more than 2^32 samples repeats samples
 * (rtx_config.rng_streams = 1 lifts that limit).  Returns after the work has finished on `stream`
 * (it polls the path queue's length while it drains). */
int rtx_render(rtx_ctx* ctx, float* accum_dev, uint32_t spp, uint32_t sample_base, uint64_t seed,
               const float cam_offset[3], uint32_t row0, uint32_t row1, void* stream);

/* rtx_render that does not wait: for loops that render frame after frame (the reference's render_loop,
 * Raytracer.cpp:307-331, one render() per pass).  A frame ends with a few long paths (chains of
 * hundreds of bounces inside glass) that occupy a handful of waves for about a millisecond; the next
 * frame's work fills the GPU meanwhile.  The context keeps a second set of path queues and result
 * slots (allocated at the first call: as much device memory again) and its own streams; frames
 * alternate between the two sets, at most one in flight per set (a third call waits for the first
 * frame).  Same arguments and the same sums, bit for bit, as rtx_render: frames that add to one
 * buffer add in call order.  The work is ordered behind what `stream` holds at the time of the call,
 * but NOT before what is enqueued on it afterwards: the sums are complete once rtx_wait, or any other
 * call on this context that is given the buffer (rtx_resolve, rtx_comm_reduce, rtx_accum_*), or
 * rtx_stream_synchronize / rtx_get_stats / rtx_render / rtx_set_scene has returned.  With per-launch
 * timing switched on (rtx_set_profiling) it behaves as rtx_render. */
int rtx_render_async(rtx_ctx* ctx, float* accum_dev, uint32_t spp, uint32_t sample_base, uint64_t seed,
                     const float cam_offset[3], uint32_t row0, uint32_t row1, void* stream);
/* Waits (host) for the frames of rtx_render_async that add to accum_dev; NULL = for all of them. */
int rtx_wait(rtx_ctx* ctx, const float* accum_dev);

/* Replaces the division in Img::set/Img::update (Raytracer.cpp:70-81) and the
 * hand-over of Img::pixels to the caller: rgb_out_host[i] = accum_dev[i] / total_spp,
 * width*height*3 floats.  Synchronous. */
int rtx_resolve(rtx_ctx* ctx, const float* accum_dev, uint32_t total_spp, float* rgb_out_host,
                void* stream);

/* Fixed-ray-set entry points (parity tests and tools).
 * rays_host: n*6 floats (o, d); d is used as given (not normalised), as
 * Scene::castRay does (Scene.h:144-189).  Outputs are host arrays of n elements
 * (normal_out: n*3, may be NULL): kind 0 miss / 1 triangle / 2 sphere / 3 box / 4 plane; prim =
 * triangle index as passed to rtx_set_scene, or the index in its own array; t = hit distance
 * (+inf on a miss).  Synchronous. */
int rtx_trace_closest(rtx_ctx* ctx, const float* rays_host, size_t n, uint32_t* kind_out,
                      uint32_t* prim_out, float* t_out, float* normal_out);
/* rtx_trace_closest for callers that must not allocate device memory (tight serving loops,
 * harnesses that forbid hipMalloc, code that cannot tolerate an allocation's implicit
 * synchronisation).  It has no counterpart in the reference.  Same arguments, same results
 * bit for bit, but no hipMalloc/hipFree on the call path: rays and results are staged in a
 * 12.6 MB array that is part of the library's device image on every GPU it runs on (held
 * until the process ends).  n is at most 262144 rays per call; a larger n fails with
 * RTX_ERR_INVALID_ARGUMENT.  All contexts on one GPU share that GPU's array: calls that
 * overlap in time on one GPU (from two threads or two contexts) race.  Synchronous. */
int rtx_trace_closest_noalloc(rtx_ctx* ctx, const float* rays_host, size_t n, uint32_t* kind_out,
                              uint32_t* prim_out, float* t_out, float* normal_out);
/* Scene::isOccluded (Scene.h:191-225) for n segments: occluded_out[i] in {0,1}. */
int rtx_trace_occluded(rtx_ctx* ctx, const float* rays_host, const float* distance_host, size_t n,
                       uint8_t* occluded_out);

/* Multi-GPU exchange step.  The reference renders with one OpenMP loop on one host
 * (Raytracer.cpp:242) and has no counterpart.  Samples are independent and a sample's random
 * stream depends only on (pixel, sample index, seed), so a render is spread over GPUs by giving
 * every GPU -- one process and one context each -- its own range of sample indices (and, for
 * large frames, its own band of rows); the per-GPU sum buffers are then added onto one root
 * with a single RCCL reduce over xGMI, and the root calls rtx_resolve.
 *   rtx_comm_create_id   on one rank: a fresh communicator id (ncclGetUniqueId); the caller
 *                        carries its RTX_COMM_ID_BYTES bytes to the other ranks (a file, a socket);
 *   rtx_comm_init        on every rank, with the same id: joins the communicator as `rank` of
 *                        `world_size` (ncclCommInitRank on the context's GPU).  Blocks until all
 *                        ranks have called it;
 *   rtx_comm_reduce      on every rank: enqueues on `stream` the element-wise sum of all ranks'
 *                        accum_dev buffers (width*height*3 floats) into the root's, in place
 *                        (ncclReduce, ncclSum).  Ordered after the renders enqueued on that stream;
 *   rtx_comm_destroy     leaves the communicator (rtx_destroy does it too).
 * RCCL is loaded on first use; a host without it gets RTX_ERR_COMM from these calls only. */
#define RTX_COMM_ID_BYTES 128
int rtx_comm_create_id(char id_out[RTX_COMM_ID_BYTES]);
int rtx_comm_init(rtx_ctx* ctx, const char id[RTX_COMM_ID_BYTES], int rank, int world_size);
int rtx_comm_reduce(rtx_ctx* ctx, float* accum_dev, int root, void* stream);
int rtx_comm_destroy(rtx_ctx* ctx);
/* Waits for the work enqueued on `stream` of the context's GPU (for hosts that do not link HIP). */
int rtx_stream_synchronize(rtx_ctx* ctx, void* stream);

/* Known-answer hook for the device's generator (class Pcg, Sampler.h:11-42), all outputs host
 * arrays of n elements, any of them may be NULL: rand_out[i] = the i-th rand_pcg() of Pcg(seed),
 * uniform_out[i] = the i-th Uniform() of another Pcg(seed), hash_out[i] = pcg_hash(seed + i),
 * each computed by the device functions the render kernels use.  n <= 2^20.  Synchronous. */
int rtx_debug_pcg(rtx_ctx* ctx, uint32_t seed, uint32_t n, uint32_t* rand_out, float* uniform_out,
                  uint32_t* hash_out);

/* The generator each sample of a call rtx_render(spp, sample_base, seed) over all rows starts with, as the render
 * kernels set it up: out_dev[k] = state | increment << 32 for sample sample_base + k / (width*height) of pixel
 * k mod (width*height); out_dev is a device array of width*height*spp 64-bit words.  With rng_streams = 0 the
 * increment is the reference's constant 2891336453 (Sampler.h:26).  Enqueues on `stream`.  For tests of
 * rtx_config.rng_streams; no counterpart in the reference. */
int rtx_debug_streams(rtx_ctx* ctx, uint32_t spp, uint32_t sample_base, uint64_t seed, uint64_t* out_dev,
                      void* stream);

/* Counters since rtx_create or the last rtx_reset_stats. */
int rtx_get_stats(rtx_ctx* ctx, rtx_stats* out);
int rtx_reset_stats(rtx_ctx* ctx);
/* on != 0: time every kernel launch of rtx_render with HIP events on its stream
 * (adds a few microseconds per launch). */
int rtx_set_profiling(rtx_ctx* ctx, int on);

/* Device-memory helpers for callers that do not link HIP themselves (a host
 * front end built with g++).  rtx_accum_clear replaces Img::clear
 * (Raytracer.cpp:61-68). */
int rtx_accum_alloc(rtx_ctx* ctx, float** accum_dev_out);
int rtx_accum_free(rtx_ctx* ctx, float* accum_dev);
int rtx_accum_clear(rtx_ctx* ctx, float* accum_dev, void* stream);
/* Restores the state Img carries between passes (Img::pixels and Img::sample_count,
 * Raytracer.cpp:49-59) from a checkpoint: copies width*height*3 per-pixel SUMS from the host
 * into accum_dev.  The sums are what rtx_resolve returns for total_spp = 1; a render resumed
 * with sample_base = the number of samples they hold gives the bits of an uninterrupted one.
 * Synchronous. */
int rtx_accum_upload(rtx_ctx* ctx, float* accum_dev, const float* sums_host, void* stream);

/*
 * Environment variables (test and tuning hooks, not part of the contract).
 *
 * rtx_create reads the following variables once and keeps what it found in the context; no other
 * entry point looks at the environment.  Each selects between two code paths that produce the
 * same bits (the GPU tests render with both and compare), or sets a tuning constant; none changes
 * what the library computes.  Production callers leave them unset.
 *   RTX_NO_LEAF_TASKS        rooms of a few surfaces: per-lane walk instead of the block's task queue
 *   RTX_LEAF_TASKS_ALWAYS    ... the task queue for sparse small scenes as well
 *   RTX_LEAF_TASK_CAP=n      (ray, leaf) pairs per thread that queue holds (default 8; 1 forces the fallback)
 *   RTX_QUEUE_LIGHT_SAMPLES  rooms: light samples queued for a kernel of their own instead of traced in the bounce kernel
 *   RTX_NO_FEW_TRIANGLES     meshes of up to 8 triangles get a tree like larger ones
 *   RTX_NO_LDS_SCENE         small trees stay in global memory instead of LDS
 *   RTX_LDS_SCENE_MAX_KB=n   size up to which a tree is staged in LDS (default 24)
 *   RTX_NO_PLAIN_KERNELS     the general build of the bounce and drain kernels for every scene
 *   RTX_FEW_BOUNCES=1        one bounce per launch where two are the default
 *   RTX_NO_LEAF_PAIRS        large meshes: one record per triangle where the halves of a quad share one by default
 *   RTX_WIDE_REFILL_LANES=n, RTX_SHADOW_REFILL_LANES=n
 *                            idle lanes at which a wave of the wide-tree kernels takes new rays
 *   RTX_REDUCE_BLOCKS=n      width of the ordered sum of a frame left in flight (0: full width)
 *   RTX_OVERLAP_MIN_SAMPLES=n  chunk size from which the ordered sum runs beside the drain kernel
 */

#ifdef __cplusplus
}
#endif
#endif
