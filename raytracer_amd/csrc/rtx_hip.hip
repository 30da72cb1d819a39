// rtx_hip.hip -- context management and the C ABI of librtx_hip.so
// (declarations and the reference code each entry point replaces: include/rtx_capi.h).
//
// HIP only: there is no CPU path in this library.  Every entry point fails with
// an error when no usable GPU is present.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/rtx_capi.h"
#include "bvh_build.h"
#include "kernels.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}
int fail_hip(hipError_t e, const char* what) {
  g_last_error = std::string(what) + ": " + hipGetErrorString(e);
  return e == hipErrorOutOfMemory ? RTX_ERR_OUT_OF_MEMORY : RTX_ERR_HIP;
}

// For catch blocks: records the message if it can, never throws.
int fail_nothrow(int code, const char* what, const char* detail) noexcept {
  try {
    g_last_error = std::string(what) + ": " + detail;
  } catch (...) {
  }
  return code;
}

#define HIP_TRY(expr)                                  \
  do {                                                 \
    hipError_t e_ = (expr);                            \
    if (e_ != hipSuccess) return fail_hip(e_, #expr);  \
  } while (0)

// k_intersect_wide: waves a launch starts (flat from 28 672 to 250 000 on the 1 M-triangle mesh; with only as
// many as the chip holds, 7 168, every segment ends at the same moment and the launch takes 25 % longer),
// and the idle lanes at which a wave stops to supply them: 24 where the tree is larger than one L2 (mesh, with
// the halves of a quad in one leaf record: 24 lanes 26.9 ms, 32: 27.6, 40: 30.3 per 32-spp frame; with one record
// per triangle it was 8 lanes 31.4, 16: 30.8, 24: 29.8, 32: 29.1, 40: 30.7, 48: 33.8), 16 below (3 000-triangle
// soup 35.1 / 34.3 / 35.5 / 36.6 for 8 / 16 / 24 / 32; 3 200-triangle terrain 15.97 / 15.91 / 16.4 / 16.8).
constexpr uint32_t kWideWaves = 40960;
constexpr size_t kWideLargeTreeBytes = 4u << 20;

constexpr uint32_t kDefaultPool = 1u << 24;      // 16 Mi paths in flight: 2 queues * 64 B + hits + shadow = 3.1 GB
constexpr uint32_t kDefaultFinish = 1u << 19;    // at most this many paths left: finish them in one launch
constexpr uint32_t kDefaultResultMiB = 8192;     // per-sample result buffer budget
constexpr uint32_t kSegmentTarget = 8192;        // aim for this many slots per queue segment
constexpr uint32_t kMaxSegments = 512;
constexpr int kTraceBlock = 128;                 // threads per block of the traversal kernels
constexpr size_t kLdsSceneMaxBytes = 24 * 1024;  // stage BVH + triangles in LDS up to this size (80-byte LDS records: about 200
                                                 // triangles).  Measured crossover with the four-wide tree in global memory on random soups
                                                 // (tools/lds_vs_global.sh): LDS 1.67x faster at 100 triangles, 1.18x at 150, equal at 220,
                                                 // 0.62x at 300
constexpr size_t kLdsLaunchMaxBytes = 64 * 1024;  // dynamic LDS a launch may ask for without raising the kernel's limit
constexpr int kWideLdsStack = 16;                // LDS stack entries per lane for the four-child tree
constexpr int kFinishBlock = 64;
#ifndef RTX_OVERLAP_REDUCE
#define RTX_OVERLAP_REDUCE 1
#endif
constexpr bool kOverlapReduce = RTX_OVERLAP_REDUCE != 0;
constexpr unsigned long long kOverlapMinSamples = 1ull << 26;
#ifndef RTX_TIGHT_END
#define RTX_TIGHT_END 1
#endif
constexpr bool kTightEnd = RTX_TIGHT_END != 0;
constexpr size_t kFusedSmallSceneBytes = 4096;  // k_trace_shade: scenes up to this size run 128-thread blocks
constexpr double kLeafTaskMinLeafArea = 1.2;  // leaf boxes' surfaces / root box's surface from which the leaf-task kernels are used
constexpr uint32_t kTraceBatches = 2;  // ray batches per block of k_intersect/k_shadow (the next batch's loads overlap the current traversal)
constexpr int kPollSlots = 4;

// rtx_trace_closest_noalloc: staging for the rays and the results of one call, in device memory
// the library owns from load time on (no hipMalloc on the call path).  One copy per GPU; every
// call on a GPU shares it.  Layout in 32-bit words, N = kNoAllocMaxRays:
//   [0, 6N) rays | [6N, 7N) t | [7N, 8N) kind | [8N, 9N) primitive | [9N, 12N) normal
constexpr size_t kNoAllocMaxRays = 1u << 18;
constexpr size_t kScratchWords = 12 * kNoAllocMaxRays;  // 12.6 MB per GPU
constexpr int kMaxDevices = 64;
static __device__ __align__(256) uint32_t scratch[kScratchWords];
// The array's address on each device, looked up once per device with that device current
// (in host code the name `scratch` is only a handle for hipGetSymbolAddress, not an address).
uint32_t* g_scratch_ptr[kMaxDevices] = {};
std::mutex g_scratch_mutex;

int resolve_scratch(int dev) {  // the device `dev` is current
  if (dev < 0 || dev >= kMaxDevices) return fail(RTX_ERR_UNSUPPORTED, "device ordinal above 63");
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (g_scratch_ptr[dev]) return RTX_OK;
  void* ptr = nullptr;
  hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
  if (e != hipSuccess || !ptr) return fail_hip(e, "hipGetSymbolAddress(scratch)");
  g_scratch_ptr[dev] = static_cast<uint32_t*>(ptr);
  return RTX_OK;
}

// hipMalloc calls made by this library in this process (rtx_stats.device_allocations): what the test of the
// entry point that allocates nothing watches.
std::atomic<uint64_t> g_device_allocations{0};

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  // n is the capacity the buffer really has: it stays 0 when the allocation fails, so a
  // later "large enough already?" test cannot pass on a null pointer.
  hipError_t alloc(size_t count) {
    release();
    if (count == 0) return hipSuccess;
    g_device_allocations.fetch_add(1, std::memory_order_relaxed);
    hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
    if (e != hipSuccess) {
      p = nullptr;
      (void)hipGetLastError();  // the runtime also remembers the failure; a later hipGetLastError() check must not see it
      return e;
    }
    n = count;
    return hipSuccess;
  }
  hipError_t upload(const std::vector<T>& h) {
    hipError_t e = alloc(h.size());
    if (e != hipSuccess || h.empty()) return e;
    return hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
};

// RCCL is loaded on first use (dlopen), so that single-GPU users of the library neither link nor
// load it.  Only the five entry points the exchange step needs.
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};
Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
    if (!r.lib) {
      const char* e = dlerror();
      r.error = std::string("cannot load librccl.so.1: ") + (e ? e : "unknown error");
      return;
    }
    auto sym = [&](const char* n) {
      void* p = dlsym(r.lib, n);
      if (!p && r.error.empty()) r.error = std::string("librccl has no symbol ") + n;
      return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.Reduce = reinterpret_cast<decltype(r.Reduce)>(sym("ncclReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
  });
  return r.error.empty() ? &r : nullptr;
}
int fail_rccl(ncclResult_t e, const char* what) {
  Rccl* r = rccl();
  g_last_error = std::string(what) + ": " + (r && r->GetErrorString ? r->GetErrorString(e) : "RCCL error");
  return RTX_ERR_COMM;
}

struct StageTimer {
  std::vector<hipEvent_t> ev;  // pairs (start, stop)
  std::vector<int> kind;
  std::vector<uint64_t> units;
};

// Test and tuning hooks (include/rtx_capi.h lists them).  The environment is read once, by rtx_create, into this
// record; nothing on the render or set-scene path looks at the environment.  Each hook selects between two code
// paths that give the same bits, or sets a tuning constant: none changes what the library computes.
struct Hooks {
  bool no_leaf_tasks = false;        // RTX_NO_LEAF_TASKS: rooms of a few surfaces keep the per-lane walk
  bool leaf_tasks_always = false;    // RTX_LEAF_TASKS_ALWAYS: ... and sparse scenes use the block's task queue too
  int leaf_task_cap = 0;             // RTX_LEAF_TASK_CAP: (ray, leaf) pairs per thread in that queue (0: default)
  bool no_few_triangles = false;     // RTX_NO_FEW_TRIANGLES: meshes of up to 8 triangles get a tree like any other
  bool no_lds_scene = false;         // RTX_NO_LDS_SCENE: small trees stay in global memory
  bool queue_light_samples = false;  // RTX_QUEUE_LIGHT_SAMPLES: rooms queue their light samples for a kernel of its own
  bool no_plain_kernels = false;     // RTX_NO_PLAIN_KERNELS: the general build of the bounce and drain kernels
  int few_bounces = 2;               // RTX_FEW_BOUNCES=1: one bounce per launch where two are the default
  long lds_scene_max_kb = -1;        // RTX_LDS_SCENE_MAX_KB: size up to which a tree is staged in LDS
  int wide_refill_lanes = -1;        // RTX_WIDE_REFILL_LANES, RTX_SHADOW_REFILL_LANES: idle lanes at which a wave of the
  int shadow_refill_lanes = -1;      //   wide-tree kernels stops to supply them
  bool no_leaf_pairs = false;        // RTX_NO_LEAF_PAIRS: wide tree: the two halves of a quad as two records
  long reduce_blocks = -1;           // RTX_REDUCE_BLOCKS: width of the ordered sum of a frame left in flight
  bool have_overlap_min = false;     // RTX_OVERLAP_MIN_SAMPLES: chunk size from which the sum runs beside the drain kernel
  unsigned long long overlap_min_samples = 0;
};
Hooks read_hooks() {
  Hooks h;
  auto flag = [](const char* name) { return std::getenv(name) != nullptr; };
  auto num = [](const char* name, long none) {
    const char* e = std::getenv(name);
    return e ? std::atol(e) : none;
  };
  h.no_leaf_tasks = flag("RTX_NO_LEAF_TASKS");
  h.leaf_tasks_always = flag("RTX_LEAF_TASKS_ALWAYS");
  h.leaf_task_cap = (int)num("RTX_LEAF_TASK_CAP", 0);
  h.no_few_triangles = flag("RTX_NO_FEW_TRIANGLES");
  h.no_lds_scene = flag("RTX_NO_LDS_SCENE");
  h.queue_light_samples = flag("RTX_QUEUE_LIGHT_SAMPLES");
  h.no_plain_kernels = flag("RTX_NO_PLAIN_KERNELS");
  h.few_bounces = num("RTX_FEW_BOUNCES", 2) == 1 ? 1 : 2;
  h.lds_scene_max_kb = num("RTX_LDS_SCENE_MAX_KB", -1);
  h.wide_refill_lanes = (int)num("RTX_WIDE_REFILL_LANES", -1);
  h.shadow_refill_lanes = (int)num("RTX_SHADOW_REFILL_LANES", -1);
  h.no_leaf_pairs = flag("RTX_NO_LEAF_PAIRS");
  h.reduce_blocks = num("RTX_REDUCE_BLOCKS", -1);
  if (const char* e = std::getenv("RTX_OVERLAP_MIN_SAMPLES")) {
    h.have_overlap_min = true;
    h.overlap_min_samples = std::strtoull(e, nullptr, 10);
  }
  return h;
}

}  // namespace

struct rtx_ctx {
  rtx_config cfg;
  Hooks hooks;  // the environment's test hooks as rtx_create found them
  int device = 0;
  uint32_t pool = 0, nseg = 1, cap = 0, finish_paths = kDefaultFinish;
  size_t result_budget_floats = 0;
  float tan_x = 0, tan_y = 0;

  // scene
  bool have_scene = false;
  bool specular_surfaces = false;  // any surface that is not diffuse: paths make long chains of bounces
  DevBuf<float4> nodes, tris, tri_nrm, tri_mat, leaf_box, mat0, sph0, sph1, box0, box1, pln0, pln1, light0, light1, light2, light3;
  DevBuf<int> mat1, tri_surf, sph2, sph3;
  DevBuf<uint32_t> slot_prim;
  rtx::SceneDev scene{};
  rtx_scene_info info{};
  int stack_entries = 8;    // stack entries per lane kept in LDS
  int trace_mode = rtx::kGlobalBvh2;
  size_t lds_scene_bytes = 0;  // kLdsBvh2: bytes of BVH + triangles every block stages
  DevBuf<int> spill;           // stack entries beyond stack_entries, one column per launched thread

  // path pool
  DevBuf<float4> r0[2], s0[2], s1[2], sq0, sq1, sq2;
  DevBuf<float2> r1[2];
  DevBuf<uint2> s2[2], hits;
  DevBuf<uint32_t> depth[2], inc[2];
  DevBuf<rtx::SegCounter> seg;
  DevBuf<rtx::Summary> sum;
  DevBuf<rtx::Primary> primary;  // written by k_begin at the start of every chunk
  rtx::Summary* h_sum = nullptr;  // pinned, kPollSlots entries
  hipEvent_t poll_ev[kPollSlots]{};
  DevBuf<float> resolve_tmp;
  DevBuf<float> res;  // per-sample results of the chunk being rendered (grown on demand)
  // The ordered sum runs beside the drain kernel on a stream of its own (see k_reduce_head).
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  DevBuf<uint4> stop4;  // per float of the image: the sample k_reduce_head stopped at
  unsigned long long overlap_min_samples = 0;  // chunks of at least this many samples split the sum
  uint32_t wide_refill_lanes = 16;  // k_intersect_wide: idle lanes at which a wave stops to supply them
  uint32_t shadow_refill_lanes = 32; // k_shadow_wide likewise; 0: k_shadow, one light sample per lane

  // multi-GPU exchange step (rtx_comm_*): this context's rank in an RCCL communicator
  ncclComm_t comm = nullptr;
  int comm_rank = -1, comm_world = 0;

  // stats
  rtx_stats stats{};
  bool profiling = false;
  std::vector<hipEvent_t> ev_pool;

  // Frames in flight (rtx_render_async).  A frame ends in a drain kernel that keeps a few waves busy for
  // about a millisecond; the next frame's rounds can fill the chip meanwhile if they have queues, result
  // slots and a stream of their own.  `twin` is that second set: a context of the same configuration whose
  // scene fields mirror this one's (it owns no scene memory).  Frames alternate between the two sets.
  rtx_ctx* twin = nullptr;
  rtx_ctx* owner = nullptr;         // the context the caller holds (itself for that context): statistics go there
  hipStream_t own = nullptr;        // the stream this set's asynchronous frames run on
  hipEvent_t ev_in = nullptr;       // the caller's stream at the time of the call
  hipEvent_t ev_done = nullptr;     // the frame in flight has finished (its summary is on the host)
  hipEvent_t ev_reduced = nullptr;  // ... has been added to the sums
  bool pending = false;             // a frame is in flight whose summary has not been absorbed
  const float* pending_accum = nullptr;
  uint64_t pending_iterations = 0, pending_launches = 0;
  bool pending_drained = false;
  int next_set = 0;
};

namespace {

using rtx::Queue;

int wait_frames(rtx_ctx* c, const float* accum = nullptr);

Queue queue_of(rtx_ctx* c, int i) {
  Queue q;
  q.r0 = c->r0[i].p;
  q.r1 = c->r1[i].p;
  q.s0 = c->s0[i].p;
  q.s1 = c->s1[i].p;
  q.s2 = c->s2[i].p;
  q.depth = c->depth[i].p;
  q.inc = c->inc[i].p;
  return q;
}

int pick_stack_entries(uint32_t depth) {
  const int sizes[] = {8, 16, 24, 32, 48, 64, 96, 128};
  for (int s : sizes)
    if ((uint32_t)s >= depth + 1) return s;
  return -1;
}

struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() {
    int cur = -1;
    if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
  }
};

void free_ctx(rtx_ctx* c) {
  if (!c) return;
  DeviceGuard g(c->device);
  if (c->pending && c->ev_done) (void)hipEventSynchronize(c->ev_done);
  if (c->twin) {
    free_ctx(c->twin);
    c->twin = nullptr;
  }
  if (c->own) (void)hipStreamSynchronize(c->own);
  if (c->ev_in) (void)hipEventDestroy(c->ev_in);
  if (c->ev_done) (void)hipEventDestroy(c->ev_done);
  if (c->ev_reduced) (void)hipEventDestroy(c->ev_reduced);
  if (c->own) (void)hipStreamDestroy(c->own);

  if (c->comm) {
    if (Rccl* r = rccl()) (void)r->CommDestroy(c->comm);
    c->comm = nullptr;
  }
  c->nodes.release(); c->tris.release(); c->tri_nrm.release(); c->slot_prim.release(); c->tri_mat.release(); c->tri_surf.release(); c->leaf_box.release();
  c->mat0.release(); c->mat1.release(); c->sph0.release(); c->sph1.release(); c->sph2.release(); c->sph3.release();
  c->box0.release(); c->box1.release(); c->pln0.release(); c->pln1.release();
  c->light0.release(); c->light1.release(); c->light2.release(); c->light3.release();
  for (int i = 0; i < 2; i++) {
    c->r0[i].release(); c->r1[i].release(); c->s0[i].release(); c->s1[i].release();
    c->s2[i].release(); c->depth[i].release(); c->inc[i].release();
  }
  c->sq0.release(); c->sq1.release(); c->sq2.release(); c->hits.release(); c->seg.release(); c->sum.release(); c->primary.release(); c->spill.release();
  c->resolve_tmp.release(); c->res.release();
  if (c->h_sum) (void)hipHostFree(c->h_sum);
  for (int i = 0; i < kPollSlots; i++)
    if (c->poll_ev[i]) (void)hipEventDestroy(c->poll_ev[i]);
  for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
  c->stop4.release();
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->side) (void)hipStreamDestroy(c->side);
  delete c;
}

}  // namespace

extern "C" {

const char* rtx_last_error(void) { return g_last_error.c_str(); }

static int impl_create(const rtx_config* cfg, rtx_ctx** out) {
  if (!cfg || !out) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_create: null argument");
  *out = nullptr;
  if (cfg->width == 0 || cfg->height == 0 || (uint64_t)cfg->width * cfg->height > (1ull << 30))
    return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_create: bad image size");
  if (cfg->camera < 0 || cfg->camera > 2 || cfg->integrator < 0 || cfg->integrator > 1)
    return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_create: bad camera or integrator");
  if (cfg->primary_rays < 0 || cfg->primary_rays > 1)
    return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_create: bad primary_rays");
  if (cfg->fused_trace < 0 || cfg->fused_trace > 2)
    return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_create: bad fused_trace");
  if (cfg->rng_streams < 0 || cfg->rng_streams > 1)
    return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_create: bad rng_streams");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0)
    return fail(RTX_ERR_NO_DEVICE, "rtx_create: no HIP device (this library has no CPU path)");
  int dev = cfg->device;
  if (dev < 0) HIP_TRY(hipGetDevice(&dev));
  if (dev >= ndev) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_create: device ordinal out of range");
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, dev));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(RTX_ERR_NO_DEVICE,
                std::string("rtx_create: device is ") + prop.gcnArchName + ", kernels are built for gfx950");

  rtx_ctx* c = new (std::nothrow) rtx_ctx();
  if (!c) return fail(RTX_ERR_OUT_OF_MEMORY, "rtx_create: host allocation failed");
  c->cfg = *cfg;
  c->hooks = read_hooks();
  c->device = dev;
  DeviceGuard g(dev);
  if (!g.ok) {
    delete c;
    return fail(RTX_ERR_HIP, "rtx_create: hipSetDevice failed");
  }
  if (int rc = resolve_scratch(dev)) {
    delete c;
    return rc;
  }
  // Loop-invariant camera terms (Camera.h:19-20): the tanf argument is evaluated in
  // double (M_PI) and narrowed, as in the reference.
  {
    const double pi = 3.14159265358979323846;
    float W = (float)cfg->width, H = (float)cfg->height;
    c->tan_x = tanf((float)((double)cfg->fov * pi / 180.0f / 2.0f));
    c->tan_y = tanf((float)((double)(H / W * cfg->fov) * pi / 180.0f / 2.0f));
  }
  uint64_t npix = (uint64_t)cfg->width * cfg->height;
  // Queue geometry: nseg segments of cap slots (cap a multiple of 256, below 2^20 so
  // that a segment's fill fits the 21-bit fields of its packed counter).
  uint64_t want = cfg->pool_paths ? cfg->pool_paths : kDefaultPool;
  if (want > (1ull << 28)) {
    delete c;
    return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_create: pool_paths above 2^28");
  }
  uint32_t nseg = (uint32_t)std::min<uint64_t>(kMaxSegments, std::max<uint64_t>(1, want / kSegmentTarget));
  uint32_t cap = (uint32_t)((want + nseg - 1) / nseg);
  cap = (cap + 255u) & ~255u;
  c->nseg = nseg;
  c->cap = cap;
  const uint32_t pool = nseg * cap;
  c->pool = pool;
  c->finish_paths = cfg->finish_paths ? cfg->finish_paths : kDefaultFinish;
  c->result_budget_floats =
      (size_t)(cfg->result_buffer_mib ? cfg->result_buffer_mib : kDefaultResultMiB) * (1u << 20) / sizeof(float);

#define CTX_TRY(expr)                     \
  do {                                    \
    hipError_t e_ = (expr);               \
    if (e_ != hipSuccess) {               \
      int rc_ = fail_hip(e_, #expr);      \
      free_ctx(c);                        \
      return rc_;                         \
    }                                     \
  } while (0)

  for (int i = 0; i < 2; i++) {
    CTX_TRY(c->r0[i].alloc(pool));
    CTX_TRY(c->r1[i].alloc(pool));
    CTX_TRY(c->s0[i].alloc(pool));
    CTX_TRY(c->s1[i].alloc(pool));
    CTX_TRY(c->s2[i].alloc(pool));
    if (cfg->max_bounces) CTX_TRY(c->depth[i].alloc(pool));
    if (cfg->rng_streams) CTX_TRY(c->inc[i].alloc(pool));
  }
  CTX_TRY(c->sq0.alloc(pool));
  CTX_TRY(c->sq1.alloc(pool));
  CTX_TRY(c->sq2.alloc(pool));
  CTX_TRY(c->hits.alloc(pool));
  CTX_TRY(c->seg.alloc(nseg));
  CTX_TRY(c->sum.alloc(1));
  CTX_TRY(c->primary.alloc(1));
  CTX_TRY(hipMemset(c->seg.p, 0, sizeof(rtx::SegCounter) * nseg));
  CTX_TRY(hipMemset(c->sum.p, 0, sizeof(rtx::Summary)));
  CTX_TRY(hipHostMalloc((void**)&c->h_sum, sizeof(rtx::Summary) * kPollSlots, hipHostMallocDefault));
  for (int i = 0; i < kPollSlots; i++) CTX_TRY(hipEventCreateWithFlags(&c->poll_ev[i], hipEventDisableTiming));
  CTX_TRY(c->resolve_tmp.alloc(npix * 3));
  CTX_TRY(c->stop4.alloc((npix * 3 + 3) / 4));
  c->overlap_min_samples = c->hooks.have_overlap_min ? c->hooks.overlap_min_samples : kOverlapMinSamples;
  CTX_TRY(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
  CTX_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
  CTX_TRY(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
#undef CTX_TRY
  *out = c;
  return RTX_OK;
}

void rtx_destroy(rtx_ctx* ctx) { free_ctx(ctx); }

static void mirror_scene(rtx_ctx* c);
static int impl_set_scene(rtx_ctx* c, const rtx_scene* s) {
  if (!c || !s) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_set_scene: null argument");
  if (s->num_triangles && (!s->triangles || !s->material_index || !s->materials))
    return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_set_scene: null geometry arrays");
  if ((s->num_spheres && !s->spheres) || (s->num_boxes && !s->boxes) || (s->num_planes && !s->planes))
    return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_set_scene: null analytic-object array");
  if (s->num_triangles >= (1u << 28))
    return fail(RTX_ERR_UNSUPPORTED, "rtx_set_scene: at most 2^28-1 triangles");
  if (s->num_spheres + s->num_boxes + s->num_planes >= (1u << 20))
    return fail(RTX_ERR_UNSUPPORTED, "rtx_set_scene: too many analytic objects");
  for (size_t i = 0; i < s->num_triangles; i++)
    if (s->material_index[i] >= s->num_materials)
      return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_set_scene: material index out of range");
  if (int rc = wait_frames(c)) return rc;
  DeviceGuard g(c->device);
  if (!g.ok) return fail(RTX_ERR_HIP, "rtx_set_scene: hipSetDevice failed");
  if (c->twin) c->twin->have_scene = false;

  auto t_begin = std::chrono::steady_clock::now();
  // Small scenes are staged in LDS by every block and get 2-triangle leaves; with
  // those a scene of n triangles has about n/2 inner nodes of 64 bytes.
  // Larger scenes stream their nodes from L2/HBM: four-child quantised nodes (two levels per
  // 64-byte fetch).  Leaves hold at most 2 triangles in both forms (a box test costs about a
  // third of a triangle test; measured best on the 1M-triangle mesh as well).
  const Hooks& hk = c->hooks;
  const size_t lds_scene_max = hk.lds_scene_max_kb >= 0 ? (size_t)hk.lds_scene_max_kb * 1024 : kLdsSceneMaxBytes;
  const bool small_scene = s->num_triangles > 0 && s->num_triangles * (16 * rtx::kLdsRecordStride * 3 / 2 + 8) <= lds_scene_max;
  rtx::FlatBvh bvh;
  rtx::build_bvh(s->triangles, s->num_triangles, 2u, small_scene ? 2u : 4u, bvh);
  // Stack: two-child trees keep all of it in LDS; the wide tree keeps kWideLdsStack entries
  // there and spills deeper ones (rare) to a global column per thread.
  int stack_entries;
  size_t spill_entries = 0;
  if (bvh.width == 4) {
    stack_entries = kWideLdsStack;
    if (bvh.max_stack + 1 > (uint32_t)kWideLdsStack) spill_entries = bvh.max_stack + 1 - kWideLdsStack;
  } else {
    stack_entries = pick_stack_entries(bvh.max_stack);
    if (stack_entries < 0) return fail(RTX_ERR_UNSUPPORTED, "rtx_set_scene: BVH too deep");
  }

  const size_t nt = s->num_triangles;
  std::vector<float4> nodes(bvh.nodes.size() / 4), tr(4 * nt), tn(nt);
  std::memcpy(nodes.data(), bvh.nodes.data(), bvh.nodes.size() * sizeof(float));
  for (size_t slot = 0; slot < nt; slot++) {
    uint32_t prim = bvh.order[slot];
    const float* p = s->triangles + 9 * (size_t)prim;
    // b-a and c-a with the reference's subtraction a + (-b) (Vec3.cpp:87-89); IEEE
    // subtraction gives the same bits.
    float bax = p[3] + (-p[0]), bay = p[4] + (-p[1]), baz = p[5] + (-p[2]);
    float cax = p[6] + (-p[0]), cay = p[7] + (-p[1]), caz = p[8] + (-p[2]);
    uint32_t mi = s->material_index[prim];
    float fprim, fmi;
    std::memcpy(&fprim, &prim, 4);
    std::memcpy(&fmi, &mi, 4);
    tr[4 * slot] = make_float4(p[0], p[1], p[2], fprim);
    tr[4 * slot + 1] = make_float4(bax, bay, baz, fmi);
    // (b-a)x(c-a) with the operations of the device's cross product (no contraction on either side)
    const float nx = bay * caz - baz * cay, ny = baz * cax - bax * caz, nz = bax * cay - bay * cax;
    tr[4 * slot + 3] = make_float4(nx, ny, nz, 0.f);
    // What this record shares with the one before it (the walk looks at the flags only for the second and later
    // triangles of a leaf): the two halves of a quad have the same first vertex, bit for bit, and those of an
    // axis-aligned rectangle the same normal.  The normals are compared as floats, so a zero component may
    // differ in sign (the cross products of a rectangle's halves give (4, 0, 0) and (4, -0, 0)): x + 0 and
    // x + -0 are the same float for every x that is not itself a zero, so d.n and n.(o-a) come out the same
    // bits unless they are zero, and a zero d.n (t infinite or NaN) or a zero n.(o-a) (t = 0, not beyond eps) is
    // no hit with either sign.
    uint32_t shared = 0;
    if (slot > 0) {
      const float4 &pa = tr[4 * slot - 4], &pn = tr[4 * slot - 1];
      const float prev[3] = {pa.x, pa.y, pa.z};
      if (std::memcmp(prev, p, 12) == 0)
        shared = (pn.x == nx && pn.y == ny && pn.z == nz) ? (rtx::kSameVertex | rtx::kSamePlane) : rtx::kSameVertex;
    }
    float fshared;
    std::memcpy(&fshared, &shared, 4);
    tr[4 * slot + 2] = make_float4(cax, cay, caz, fshared);
    // The shading normal of a mesh hit, normalize(Ng) (Scene.h:181-187, Vec3.cpp:24-30): the square
    // root of ((x*x + y*y) + z*z), then three divisions; the same IEEE operations the device would
    // perform per hit, done once here (this file is compiled without contraction as well).
    const float len = std::sqrt(nx * nx + ny * ny + nz * nz);
    tn[slot] = make_float4(nx / len, ny / len, nz / len, fmi);
  }
  // Wide tree: the triangle records are re-packed as leaf records, in leaf order and without gaps, and a leaf word
  // counts in records.  A leaf whose two triangles are the halves of a fan-tessellated quad, (a, b, c) then (a, c, d)
  // bit for bit, is ONE record of the four vertices (rtx::pair_intersect; kLeafPairFlag in its leaf word); any other
  // triangle keeps a record of its own.  Row 2 of a record carries the triangle slot of its (first) triangle, which
  // is what a hit reports (the per-slot arrays tri_nrm and slot_prim are unchanged).  Dense records matter because the
  // L2 fetches 128-byte lines: two neighbouring leaves then share a line, where a pair record used to share it with the
  // unused record of its second slot.
  if (bvh.width == 4) {
    std::vector<uint8_t> leaf_count(nt, 0);  // by first slot
    for (size_t k = 3; k < nodes.size(); k += 4) {
      int refs[4];
      std::memcpy(refs, &nodes[k], 16);
      for (int r : refs)
        if (r < 0 && r != rtx::kTraversalDone) leaf_count[~(uint32_t)r >> 3] = (uint8_t)((~(uint32_t)r & 7u) + 1u);
    }
    std::vector<uint32_t> word_of_slot(nt, 0);  // new leaf word, by first slot
    std::vector<float4> packed;
    packed.reserve(tr.size());
    for (size_t first = 0; first < nt;) {
      const uint32_t cnt = leaf_count[first];
      if (cnt == 0) return fail(RTX_ERR_HIP, "rtx_set_scene: internal error, a triangle slot outside every leaf");
      const uint32_t rec = (uint32_t)(packed.size() / 4);
      const uint32_t p0 = bvh.order[first];
      const float* t0 = s->triangles + 9 * (size_t)p0;
      bool pair = false;
      if (cnt == 2 && !hk.no_leaf_pairs) {
        const uint32_t p1 = bvh.order[first + 1];
        const float* t1 = s->triangles + 9 * (size_t)p1;
        if (std::memcmp(t1, t0, 12) == 0 && std::memcmp(t1 + 3, t0 + 6, 12) == 0) {
          float f0, f1, fs;
          const uint32_t slot = (uint32_t)first;
          std::memcpy(&f0, &p0, 4);
          std::memcpy(&f1, &p1, 4);
          std::memcpy(&fs, &slot, 4);
          packed.push_back(make_float4(t0[0], t0[1], t0[2], f0));
          packed.push_back(make_float4(t0[3], t0[4], t0[5], f1));
          packed.push_back(make_float4(t0[6], t0[7], t0[8], fs));
          packed.push_back(make_float4(t1[6], t1[7], t1[8], 0.f));
          word_of_slot[first] = (rec << 3) | 1u | rtx::kLeafPairFlag;
          pair = true;
        }
      }
      if (!pair) {
        for (uint32_t k = 0; k < cnt; k++) {
          const uint32_t slot = (uint32_t)first + k;
          float fs;
          std::memcpy(&fs, &slot, 4);
          packed.push_back(tr[4 * (size_t)slot]);
          packed.push_back(tr[4 * (size_t)slot + 1]);
          packed.push_back(make_float4(tr[4 * (size_t)slot + 2].x, tr[4 * (size_t)slot + 2].y, tr[4 * (size_t)slot + 2].z, fs));
          packed.push_back(tr[4 * (size_t)slot + 3]);
        }
        word_of_slot[first] = (rec << 3) | (cnt - 1u);
      }
      first += cnt;
    }
    for (size_t k = 3; k < nodes.size(); k += 4) {
      int refs[4];
      std::memcpy(refs, &nodes[k], 16);
      for (int& r : refs)
        if (r < 0 && r != rtx::kTraversalDone) r = (int)~word_of_slot[~(uint32_t)r >> 3];
      std::memcpy(&nodes[k], refs, 16);
    }
    tr.swap(packed);
  }
  std::vector<float4> m0(s->num_materials);
  std::vector<int> m1(s->num_materials);
  for (size_t i = 0; i < s->num_materials; i++) {
    const rtx_material& m = s->materials[i];
    m0[i] = make_float4(m.albedo[0], m.albedo[1], m.albedo[2], m.emission);
    m1[i] = m.surface;
  }
  c->specular_surfaces = false;
  for (size_t i = 0; i < s->num_materials; i++) c->specular_surfaces |= s->materials[i].surface != RTX_DIFFUSE;
  // Analytic objects: one material list in the order spheres, boxes, planes.
  const size_t nobj = s->num_spheres + s->num_boxes + s->num_planes;
  std::vector<float4> sp0(s->num_spheres), sp1(nobj), bx0(s->num_boxes), bx1(s->num_boxes), pl0(s->num_planes),
      pl1(s->num_planes);
  std::vector<int> sp2(nobj);
  auto put_material = [&](size_t i, const rtx_material& m) {
    sp1[i] = make_float4(m.albedo[0], m.albedo[1], m.albedo[2], m.emission);
    sp2[i] = m.surface;
    c->specular_surfaces |= m.surface != RTX_DIFFUSE;
  };
  for (size_t i = 0; i < s->num_spheres; i++) {
    const rtx_sphere& q = s->spheres[i];
    sp0[i] = make_float4(q.c[0], q.c[1], q.c[2], q.r);
    put_material(i, q.material);
  }
  for (size_t i = 0; i < s->num_boxes; i++) {
    const rtx_box& q = s->boxes[i];
    bx0[i] = make_float4(q.min[0], q.min[1], q.min[2], 0.f);
    bx1[i] = make_float4(q.max[0], q.max[1], q.max[2], 0.f);
    put_material(s->num_spheres + i, q.material);
  }
  for (size_t i = 0; i < s->num_planes; i++) {
    const rtx_plane& q = s->planes[i];
    pl0[i] = make_float4(q.point[0], q.point[1], q.point[2], 0.f);
    pl1[i] = make_float4(q.normal[0], q.normal[1], q.normal[2], 0.f);
    put_material(s->num_spheres + s->num_boxes + i, q.material);
  }
  // Spheres no triangle comes near (LastSphere::clear): every triangle's box stays outside the
  // sphere's box grown by a thousandth of the radius (and a little more for large coordinates).
  std::vector<int> sp3(s->num_spheres, 0);
  if ((double)s->num_spheres * (double)nt <= 1.0e8) {
    for (size_t i = 0; i < s->num_spheres; i++) {
      const rtx_sphere& q = s->spheres[i];
      if (!(q.r > 0.0f) || !std::isfinite(q.r)) continue;
      float lo[3], hi[3];
      bool clear = true;
      for (int k = 0; k < 3; k++) {
        const float grow = q.r * 1.001f + 1.0e-5f * (std::fabs(q.c[k]) + q.r) + 1.0e-30f;
        lo[k] = q.c[k] - grow;
        hi[k] = q.c[k] + grow;
        clear = clear && std::isfinite(lo[k]) && std::isfinite(hi[k]);
      }
      for (size_t t = 0; clear && t < nt; t++) {
        const float* p = s->triangles + 9 * t;
        bool apart = false;
        for (int k = 0; k < 3; k++) {
          const float tlo = std::min(p[k], std::min(p[3 + k], p[6 + k])), thi = std::max(p[k], std::max(p[3 + k], p[6 + k]));
          if (thi < lo[k] || tlo > hi[k]) apart = true;   // (a NaN coordinate compares false: not apart)
        }
        clear = apart;
      }
      sp3[i] = clear ? 1 : 0;
    }
  }
  // Light list, Scene::commit (Scene.h:129-141): triangles in order whose material
  // emission exceeds the threshold; per light the Triangle ctor's n and rcp_area
  // (Obj.h:107): n = normalize((b-a)x(c-b)), rcp_area = 2/|(b-a)x(c-b)|.
  std::vector<float4> l0, l1, l2, l3;
  for (size_t i = 0; i < nt; i++) {
    const rtx_material& m = s->materials[s->material_index[i]];
    if (!(m.emission > c->cfg.nee_threshold)) continue;
    const float* p = s->triangles + 9 * i;
    float bax = p[3] + (-p[0]), bay = p[4] + (-p[1]), baz = p[5] + (-p[2]);
    float cbx = p[6] + (-p[3]), cby = p[7] + (-p[4]), cbz = p[8] + (-p[5]);
    float nx = bay * cbz - baz * cby, ny = baz * cbx - bax * cbz, nz = bax * cby - bay * cbx;
    float len = std::sqrt(nx * nx + ny * ny + nz * nz);
    l0.push_back(make_float4(p[0], p[1], p[2], 2.0f / len));
    l1.push_back(make_float4(p[3], p[4], p[5], m.emission));
    l2.push_back(make_float4(p[6], p[7], p[8], 0.f));
    l3.push_back(make_float4(nx / len, ny / len, nz / len, 0.f));
  }

  HIP_TRY(hipDeviceSynchronize());
  // From here on the old scene's buffers are being replaced: until every upload has succeeded
  // the context has no scene (a failure below must not leave renders running on freed memory).
  c->have_scene = false;
  HIP_TRY(c->spill.alloc(spill_entries * (size_t)c->pool));
  HIP_TRY(c->nodes.upload(nodes));
  HIP_TRY(c->tris.upload(tr));
  HIP_TRY(c->tri_nrm.upload(tn));
  HIP_TRY(c->slot_prim.upload(bvh.order));
  HIP_TRY(c->mat0.upload(m0));
  HIP_TRY(c->mat1.upload(m1));
  // Small scenes (everything but the wide tree): the material copied out per triangle slot, so that the loads of
  // a hit do not wait for its material index (hit_surface).  Large meshes keep the one table: their shading is
  // bound by memory traffic, and a table of a few materials stays in cache where a per-slot copy would not.
  std::vector<float4> tm;
  std::vector<int> ts;
  if (bvh.width != 4) {
    tm.resize(nt);
    ts.resize(nt);
    for (size_t slot = 0; slot < nt; slot++) {
      const uint32_t mi = s->material_index[bvh.order[slot]];
      tm[slot] = m0[mi];
      ts[slot] = m1[mi];
    }
  }
  HIP_TRY(c->tri_mat.upload(tm));
  HIP_TRY(c->tri_surf.upload(ts));
  // Trees of a handful of leaves in scenes where a ray meets leaves at every turn (a closed room): the leaves' boxes as
  // a list, for the kernels that test them all and hand the triangle tests out as tasks of their block
  // (block_closest_by_leaf_tasks; 16-bit primitive ids and slots there).  The sum of the leaf boxes' surfaces over the
  // root's is the number of leaf boxes a ray through the root's box meets on average.  Measured (time with the tasks
  // over time with the walk; tools/leaf_task_calibration.py): Cornell's seventeen leaves, 2.3 times the root's surface,
  // 0.87; soups of 30 triangles of growing size under the sky: 0.36 -> 1.10, 0.89 -> 0.99, 1.75 -> 0.93, 3.2 -> 0.89,
  // 4.6 -> 0.91.  Where most rays leave the tree after a node or two the list makes them test every leaf box for nothing.
  std::vector<float4> lb;
  if (bvh.width == 2 && nt < 65536 && !hk.no_leaf_tasks) {
    auto half_area = [](float lx, float hx, float ly, float hy, float lz, float hz) {
      const double dx = std::max(0.f, hx - lx), dy = std::max(0.f, hy - ly), dz = std::max(0.f, hz - lz);
      return dx * dy + dy * dz + dz * dx;
    };
    double leaf_area = 0.0, root_area = 0.0;
    for (size_t k = 0; k + 3 < nodes.size() && lb.size() <= 2 * rtx::kLeafTaskMaxLeaves; k += 4) {
      const float4 &q0 = nodes[k], &q1 = nodes[k + 1], &q2 = nodes[k + 2], &q3 = nodes[k + 3];
      if (k == 0)
        root_area = half_area(std::min(q0.x, q1.x), std::max(q0.y, q1.y), std::min(q0.z, q1.z), std::max(q0.w, q1.w),
                              std::min(q2.x, q2.z), std::max(q2.y, q2.w));
      int c0, c1;
      std::memcpy(&c0, &q3.x, 4);
      std::memcpy(&c1, &q3.y, 4);
      if (c0 < 0 && c0 != rtx::kTraversalDone) {
        float w;
        const uint32_t leaf = ~(uint32_t)c0;
        std::memcpy(&w, &leaf, 4);
        lb.push_back(make_float4(q0.x, q0.y, q0.z, q0.w));
        lb.push_back(make_float4(q2.x, q2.y, w, 0.f));
        leaf_area += half_area(q0.x, q0.y, q0.z, q0.w, q2.x, q2.y);
      }
      if (c1 < 0 && c1 != rtx::kTraversalDone) {
        float w;
        const uint32_t leaf = ~(uint32_t)c1;
        std::memcpy(&w, &leaf, 4);
        lb.push_back(make_float4(q1.x, q1.y, q1.z, q1.w));
        lb.push_back(make_float4(q2.z, q2.w, w, 0.f));
        leaf_area += half_area(q1.x, q1.y, q1.z, q1.w, q2.z, q2.w);
      }
    }
    const bool dense = leaf_area >= kLeafTaskMinLeafArea * root_area || hk.leaf_tasks_always;
    // (the list takes the node records' place in the kernels' LDS: stage_leaf_task_geom_lds)
    if (lb.size() > 2 * rtx::kLeafTaskMaxLeaves || lb.size() > rtx::kLdsRecordStride * (nodes.size() / 4) || !dense) lb.clear();
  }
  HIP_TRY(c->leaf_box.upload(lb));
  HIP_TRY(c->sph0.upload(sp0));
  HIP_TRY(c->sph1.upload(sp1));
  HIP_TRY(c->sph2.upload(sp2));
  HIP_TRY(c->sph3.upload(sp3));
  HIP_TRY(c->box0.upload(bx0));
  HIP_TRY(c->box1.upload(bx1));
  HIP_TRY(c->pln0.upload(pl0));
  HIP_TRY(c->pln1.upload(pl1));
  HIP_TRY(c->light0.upload(l0));
  HIP_TRY(c->light1.upload(l1));
  HIP_TRY(c->light2.upload(l2));
  HIP_TRY(c->light3.upload(l3));

  rtx::SceneDev& d = c->scene;
  d.nodes = c->nodes.p; d.tris = c->tris.p; d.tri_nrm = c->tri_nrm.p; d.slot_prim = c->slot_prim.p;
  d.mat0 = c->mat0.p; d.mat1 = c->mat1.p;
  d.tri_mat = tm.empty() ? nullptr : c->tri_mat.p;
  d.tri_surf = tm.empty() ? nullptr : c->tri_surf.p;
  d.leaf_box = lb.empty() ? nullptr : c->leaf_box.p;
  d.num_leaf_boxes = (uint32_t)lb.size() / 2;
  d.leaf_task_cap = hk.leaf_task_cap > 0 ? (uint32_t)std::min(32, hk.leaf_task_cap) : 8u;  // (1 makes lanes fall back)
  d.sph0 = c->sph0.p; d.sph1 = c->sph1.p; d.sph2 = c->sph2.p; d.sph3 = c->sph3.p;
  d.box0 = c->box0.p; d.box1 = c->box1.p; d.pln0 = c->pln0.p; d.pln1 = c->pln1.p;
  d.num_boxes = (uint32_t)s->num_boxes;
  d.num_planes = (uint32_t)s->num_planes;
  d.light0 = c->light0.p; d.light1 = c->light1.p; d.light2 = c->light2.p; d.light3 = c->light3.p;
  d.num_nodes = (uint32_t)nodes.size() / 4;
  d.num_tris = (uint32_t)nt;
  d.num_spheres = (uint32_t)s->num_spheres;
  d.num_lights = (uint32_t)l0.size();
  d.eps = c->cfg.eps;
  d.nee_threshold = c->cfg.nee_threshold;
  d.rr_cap = c->cfg.rr_cap;
  d.sky = c->cfg.sky;
  d.max_bounces = c->cfg.max_bounces;
  d.integrator = c->cfg.integrator;
  d.rng_streams = c->cfg.rng_streams;
  bool emits = !l0.empty();
  for (const float4& m : m0) emits = emits || m.w > 0.0f;
  for (const float4& m : sp1) emits = emits || m.w > 0.0f;
  d.carries_colour = emits ? 1 : 0;
  c->stack_entries = stack_entries;
  c->lds_scene_bytes = (nodes.size() / 4 + nt) * rtx::kLdsRecordStride * sizeof(float4);  // as stage_geom_lds lays it out
  c->wide_refill_lanes = (nodes.size() + (size_t)nt * 4) * sizeof(float4) > kWideLargeTreeBytes ? 24 : 16;
  if (hk.wide_refill_lanes >= 0) c->wide_refill_lanes = (uint32_t)std::min(64, std::max(1, hk.wide_refill_lanes));
  if (hk.shadow_refill_lanes >= 0) c->shadow_refill_lanes = (uint32_t)std::min(64, hk.shadow_refill_lanes);
  if (bvh.width == 4)
    c->trace_mode = rtx::kGlobalBvh4;
  else if (nt > 0 && nt <= rtx::kFewTrianglesMax && !hk.no_few_triangles)
    c->trace_mode = rtx::kFewTriangles;
  else if (nt > 0 && c->lds_scene_bytes <= lds_scene_max && !hk.no_lds_scene &&
           (size_t)stack_entries * kTraceBlock * sizeof(int) + c->lds_scene_bytes <= kLdsLaunchMaxBytes)
    c->trace_mode = rtx::kLdsBvh2;  // (a very deep small tree leaves the LDS to the stacks and stays in global memory)
  else
    c->trace_mode = rtx::kGlobalBvh2;
  // The kernels that run triangle tests as tasks of the block stage the tree in LDS and put their workspace behind
  // it: they exist for kLdsBvh2 only.  A tree that ended up elsewhere (a mesh of up to eight triangles, a tree kept in
  // global memory) has no leaf-box list, whatever its leaves' surfaces add up to.
  if (c->trace_mode != rtx::kLdsBvh2) {
    d.leaf_box = nullptr;
    d.num_leaf_boxes = 0;
  }
  // (a handful of surfaces and nothing else: hit_surface reads them all through the scalar cache)
  d.few_surfaces = c->trace_mode == rtx::kFewTriangles && s->num_spheres <= 4 && s->num_boxes == 0 && s->num_planes == 0;
  c->have_scene = true;
  if (c->twin) {
    mirror_scene(c);
    if (c->spill.n != c->twin->spill.n) HIP_TRY(c->twin->spill.alloc(c->spill.n));
  }
  c->info.num_nodes = d.num_nodes;
  c->info.num_triangles = d.num_tris;
  c->info.num_lights = d.num_lights;
  c->info.num_spheres = d.num_spheres;
  c->info.max_depth = bvh.max_depth;
  c->info.stack_entries = (uint32_t)stack_entries;
  c->info.bvh_width = bvh.width;
  c->info.stack_bound = bvh.max_stack;
  c->info.build_ms =
      std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  return RTX_OK;
}

static int impl_get_scene_info(rtx_ctx* c, rtx_scene_info* out) {
  if (!c || !out) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_get_scene_info: null argument");
  if (!c->have_scene) return fail(RTX_ERR_NO_SCENE, "rtx_get_scene_info: no scene set");
  *out = c->info;
  return RTX_OK;
}

}  // extern "C"

// Launches kernel<BLOCK, mode> for the context's traversal mode.
#define RTX_LAUNCH(K, B, M, CFG, ARGS) hipLaunchKernelGGL((K<B, M>), RTX_UNPACK CFG, RTX_UNPACK ARGS)
#define RTX_UNPACK(...) __VA_ARGS__
#define RTX_BY_MODE(mode, K, B, CFG, ARGS)                                   \
  do {                                                                       \
    if ((mode) == rtx::kLdsBvh2)                                             \
      RTX_LAUNCH(K, B, rtx::kLdsBvh2, CFG, ARGS);                            \
    else if ((mode) == rtx::kFewTriangles)                                   \
      RTX_LAUNCH(K, B, rtx::kFewTriangles, CFG, ARGS);                       \
    else if ((mode) == rtx::kGlobalBvh4)                                     \
      RTX_LAUNCH(K, B, rtx::kGlobalBvh4, CFG, ARGS);                         \
    else                                                                     \
      RTX_LAUNCH(K, B, rtx::kGlobalBvh2, CFG, ARGS);                         \
  } while (0)

namespace {

enum StageKind { kGen = 0, kIsect = 1, kShade = 2, kShadow = 3, kFinish = 4, kReduce = 5, kOther = 6 };

// Records start/stop events around a launch when profiling is on.
struct LaunchTimer {
  rtx_ctx* c;
  hipStream_t st;
  StageTimer* tm;
  LaunchTimer(rtx_ctx* c_, hipStream_t st_, StageTimer* tm_) : c(c_), st(st_), tm(tm_) {}
  hipEvent_t get_event() {
    if (!c->ev_pool.empty()) {
      hipEvent_t e = c->ev_pool.back();
      c->ev_pool.pop_back();
      return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
  }
  void begin(int kind, uint64_t units) {
    if (!tm) return;
    hipEvent_t a = get_event(), b = get_event();
    tm->ev.push_back(a);
    tm->ev.push_back(b);
    tm->kind.push_back(kind);
    tm->units.push_back(units);
    (void)hipEventRecord(a, st);
  }
  void end() {
    if (!tm) return;
    (void)hipEventRecord(tm->ev.back(), st);
  }
};

uint32_t div_up(uint32_t n, uint32_t d) { return n == 0 ? 1u : (n + d - 1) / d; }

// Renders samples [sample_base, sample_base+spp) of the band into the result
// buffer and folds them into accum.  One chunk = one fill/drain of the queues.
// Takes the summary of a finished chunk (in h_sum[0]) into the statistics.
int absorb_summary(rtx_ctx* c, uint64_t iterations, uint64_t launches, bool drained, bool overlapped) {
  const rtx::Summary& h = c->h_sum[0];
  if (h.next_sample != h.total_samples)
    return fail(RTX_ERR_HIP, "rtx_render: internal error, samples left ungenerated");
  rtx_stats& st = (c->owner ? c->owner : c)->stats;
  st.paths += h.stat_paths;
  st.bounces += h.stat_bounces;
  st.nee_samples += h.stat_nee;
  st.shadow_rays += h.stat_shadow;
  st.stored_paths += h.stat_stored;
  st.nan_paths += h.stat_nan;
  st.truncated_paths += h.stat_truncated;
  st.iterations += iterations;
  st.launches += launches;
  st.drain_launches += drained ? 1 : 0;
  st.overlapped_sums += overlapped ? 1 : 0;
  return RTX_OK;
}

// Waits for the frame rtx_render_async left in flight on this set, if any.
int finish_pending(rtx_ctx* c) {
  if (!c || !c->pending) return RTX_OK;
  c->pending = false;
  c->pending_accum = nullptr;
  HIP_TRY(hipEventSynchronize(c->ev_done));
  HIP_TRY(hipGetLastError());
  return absorb_summary(c, c->pending_iterations, c->pending_launches, c->pending_drained, false);
}
// ... on either set (accum == nullptr), or for the frames that add to `accum`.
int wait_frames(rtx_ctx* c, const float* accum) {
  if (!c) return RTX_OK;
  for (rtx_ctx* s : {c, c->twin}) {
    if (!s || !s->pending) continue;
    if (accum && s->pending_accum != accum) continue;
    DeviceGuard g(s->device);
    if (int rc = finish_pending(s)) return rc;
  }
  return RTX_OK;
}

// `detach`: leave the chunk in flight (its last launches enqueued, ev_done recorded behind them) instead of
// waiting for it.  `after_reduce`: an event the chunk's additions to the sums have to wait for (the other
// set's frame adds to the same sums first), or null.
int render_chunk(rtx_ctx* c, float* accum_band, const rtx::CameraDev& cam, rtx::RenderParams rp,
                 uint32_t spp, hipStream_t st, StageTimer* timer, bool detach = false,
                 hipEvent_t after_reduce = nullptr) {
  const uint32_t nseg = c->nseg, cap = c->cap, pool = c->pool;
  const unsigned long long total = (unsigned long long)rp.band_pixels * spp;
  rtx::SegCounter* seg = c->seg.p;
  rtx::Summary* sum = c->sum.p;
  rtx::ShadowQueue sq{c->sq0.p, c->sq1.p, c->sq2.p};
  const int use_depth = c->cfg.max_bounces ? 1 : 0;
  const size_t scene_lds = c->trace_mode == rtx::kLdsBvh2 ? c->lds_scene_bytes : 0;
  const size_t stack_lds = c->trace_mode == rtx::kFewTriangles ? 0 : (size_t)c->stack_entries * sizeof(int);  // per thread
  const size_t lds = stack_lds * kTraceBlock + scene_lds;
  const size_t lds_fin = stack_lds * kFinishBlock + scene_lds;
  const rtx::TraceParams tp{c->stack_entries, c->spill.p, pool};
  const bool shadows = c->scene.num_lights != 0 && c->cfg.integrator == 0;
  // One kernel for search and shading where the tree is staged in LDS.  Every block stages the
  // scene, so larger scenes get larger blocks (measured: 2 % for 128 threads below the limit,
  // 14 % for 256 at a scene of 16 KB).
  const bool few = c->trace_mode == rtx::kFewTriangles;
  const bool fused = c->cfg.fused_trace != 1 && (c->trace_mode == rtx::kLdsBvh2 || few);
  // Rooms whose triangle tests run as tasks of the block trace a bounce's light sample in the bounce kernel as
  // well, as a second round of tasks: no light-sample queue, no launch for it (MIS Cornell 57.4 -> 51.4 ms per
  // frame, same bits).  RTX_QUEUE_LIGHT_SAMPLES keeps the queue and k_shadow_leaf_tasks (test hook).
  const bool inline_light_samples = fused && !few && shadows && c->scene.num_leaf_boxes != 0 &&
                                    !c->hooks.queue_light_samples;
  // Light-free scenes without a tree make two bounces per launch (k_trace_shade, BOUNCES); RTX_FEW_BOUNCES=1: one
  // (the hook of the test that compares the two).
  const int few_bounces = c->hooks.few_bounces;
  // (a scene of triangles and spheres, path-traced without a bounce limit: kernels built for exactly that)
  const bool plain = c->scene.num_boxes == 0 && c->scene.num_planes == 0 && c->scene.integrator == 0 &&
                     c->scene.max_bounces == 0 && c->scene.rng_streams == 0 && !c->hooks.no_plain_kernels;
  const bool open_sky = c->scene.sky == 1 && c->scene.carries_colour == 0 && c->scene.few_surfaces == 1;
  const int fused_block = c->lds_scene_bytes <= kFusedSmallSceneBytes ||
                                  (size_t)c->stack_entries * 256 * sizeof(int) + scene_lds > kLdsLaunchMaxBytes
                              ? 128
                              : 256;
  const size_t lds_fused = (size_t)c->stack_entries * fused_block * sizeof(int) + scene_lds;
  const uint32_t nfloats = rp.band_pixels * 3;
  // (not while launches are being timed one by one: the two halves of the sum would not add up to a stage)
  // (and not for small chunks: the extra launches and events cost more than the sum)
  // (nor for a frame left in flight: the next frame's rounds run beside its drain kernel and its sum alike, and a
  // third stream would have to share one of the process's four hardware queues with something)
  const bool overlap_reduce =
      kOverlapReduce && !timer && !detach && (nfloats & 3u) == 0 && total >= c->overlap_min_samples;
  bool reduce_split = false;
  const uint32_t plan_threads = (nseg + 63u) / 64u * 64u;  // k_plan: one thread per segment, whole waves
  LaunchTimer lt(c, st, timer);
  uint64_t launches = 0, iterations = 0;

  // Fresh paths: recomputed by the kernels that need them, or written by k_generate (the thin
  // lens draws sinf/cosf per sample, and the wide-tree kernel prefetches rays by the 64).
  rtx::Primary pr{cam, rp, 0};
  pr.recompute = c->cfg.primary_rays != 1 && c->trace_mode != rtx::kGlobalBvh4 && c->cfg.camera != 1;
  const rtx::Primary* prp = c->primary.p;
  rtx::k_begin<<<div_up(nseg, 256), 256, 0, st>>>(seg, sum, nseg, total, pr, c->primary.p);
  rtx::k_plan<<<1, plan_threads, 0, st>>>(seg, sum, 0, -1, nseg, cap);
  const uint32_t first = (uint32_t)std::min<unsigned long long>(total, pool);
  launches += 2;
  if (!pr.recompute) {
    lt.begin(kGen, first);
    rtx::k_generate<<<dim3(div_up(std::min(first, cap), 256), nseg), 256, 0, st>>>(queue_of(c, 0), seg, cam, rp,
                                                                                use_depth);
    lt.end();
    launches++;
  }

  int cur = 0;
  // What the host knows: upper bounds, refreshed from snapshots of the summary.
  uint32_t seg_max_upper = std::min(first, cap);
  uint32_t active_upper = first;
  unsigned long long left_upper = total - first;
  int poll_head = 0, poll_pending = 0;
  uint64_t since_poll = 0;
  bool finished = false;
  // Towards the end of the samples the host stops running far ahead of the device: rounds
  // enqueued on stale knowledge keep their full grids and go on after the paths are gone.  It
  // follows how fast the samples go (from consecutive snapshots) and, once they are about to run
  // out at the round it is enqueuing, takes a snapshot every round and waits for the one before.
  uint64_t snap_round[kPollSlots] = {};
  uint64_t known_round = 0, prev_round = 0;
  unsigned long long known_left = left_upper, prev_next = first, rate = 0;
  bool have_prev = false;

  auto absorb = [&](const rtx::Summary& h, uint64_t round) {
    left_upper = h.total_samples - h.next_sample;
    if (left_upper == 0) {
      seg_max_upper = std::min(seg_max_upper, h.active_max);
      active_upper = std::min(active_upper, h.active_total);
    }
    if (have_prev && round > prev_round && h.next_sample >= prev_next)
      rate = std::max(rate, (h.next_sample - prev_next) / (round - prev_round));
    have_prev = true;
    prev_round = round;
    prev_next = h.next_sample;
    known_round = round;
    known_left = left_upper;
  };
  auto near_the_end = [&]() {
    if (known_left == 0 || total <= 4ull * pool) return true;
    if (rate == 0) return false;
    return known_left <= rate * ((iterations - known_round) + 3);
  };

  while (true) {
    if (left_upper == 0 && active_upper == 0) break;
    if (left_upper == 0 && active_upper <= c->finish_paths) {
      if (after_reduce) {
        HIP_TRY(hipStreamWaitEvent(st, after_reduce, 0));
        after_reduce = nullptr;
      }
      if (overlap_reduce) {
        // Most of the ordered sum runs on the side stream while the last paths run to their ends.
        rtx::k_mark_unfinished<<<dim3(div_up(seg_max_upper, 64), nseg), 64, 0, st>>>(queue_of(c, cur), seg, cur, cap,
                                                                                   c->res.p, prp);
        HIP_TRY(hipEventRecord(c->ev_fork, st));
        launches++;
        reduce_split = true;
      }
      lt.begin(kFinish, active_upper);
      if (few && plain && open_sky && c->scene.num_lights == 0)
        rtx::k_finish<kFinishBlock, rtx::kFewTriangles, 2><<<dim3(div_up(seg_max_upper, kFinishBlock), nseg), kFinishBlock,
                                                             lds_fin, st>>>(c->scene, queue_of(c, cur), seg, cur, cap,
                                                                            c->res.p, sum, tp, prp);
      else
        RTX_BY_MODE(c->trace_mode, rtx::k_finish, kFinishBlock,
                    (dim3(div_up(seg_max_upper, kFinishBlock), nseg), kFinishBlock, lds_fin, st),
                    (c->scene, queue_of(c, cur), seg, cur, cap, c->res.p, sum, tp, prp));
      lt.end();
      launches++;
      if (reduce_split) {
        // (enqueued behind the drain kernel: the other way round the sum's twelve thousand blocks can hold
        // the drain kernel's few back until they are all out, and the two run one after the other)
        HIP_TRY(hipStreamWaitEvent(c->side, c->ev_fork, 0));
        rtx::k_reduce_head<<<div_up(nfloats / 4, 256), 256, 0, c->side>>>(c->res.p, accum_band, c->stop4.p, nfloats, spp);
        HIP_TRY(hipEventRecord(c->ev_join, c->side));
        launches++;
      }
      finished = true;
      break;
    }
    const int nxt = cur ^ 1;
    Queue qc = queue_of(c, cur), qn = queue_of(c, nxt);
    // Traversal kernels loop over their segment, kTraceBatches batches of rays per block.
    uint32_t gx128 = div_up(seg_max_upper, kTraceBlock);
    gx128 = std::max(1u, std::min(gx128, div_up(gx128, kTraceBatches)));
    const uint32_t gx256 = div_up(seg_max_upper, 256);
    if (fused) {
      lt.begin(kShade, active_upper);
      // (two builds of each: with lights a bounce queues light samples, without it reserves its queue slots early)
#define RTX_TRACE_SHADE(B, M, LDS)                                                                                \
  do {                                                                                                            \
    if (c->scene.num_lights != 0)                                                                                 \
      rtx::k_trace_shade<B, M, true><<<dim3(div_up(seg_max_upper, B), nseg), B, LDS, st>>>(                       \
          c->scene, qc, qn, sq, seg, cur, cap, c->res.p, tp, prp, sum);                                           \
    else                                                                                                          \
      rtx::k_trace_shade<B, M, false><<<dim3(div_up(seg_max_upper, B), nseg), B, LDS, st>>>(                      \
          c->scene, qc, qn, sq, seg, cur, cap, c->res.p, tp, prp, sum);                                           \
  } while (0)
      // Scenes without a tree: no barrier and no LDS, so a block is one wave where waves live long (chains of
      // specular bounces), and two where they are short enough for the dispatch of blocks to be what limits the
      // launch.  sample2's frame (glass): 5.54 ms at 128 threads, 5.46 at 64, 5.9 at 256, 6.8 at 512 (waves of a
      // block leave their slots together); sample1's (diffuse only): 0.42 at 128, 0.44 at 64.
#define RTX_FEW_LAUNCH(B)                                                                                         \
  do {                                                                                                            \
    const dim3 grid(div_up(seg_max_upper, B), nseg);                                                              \
    if (c->scene.num_lights == 0 && few_bounces == 2 && plain && open_sky) /* two bounces per launch */           \
      rtx::k_trace_shade<B, rtx::kFewTriangles, false, false, 2, 2><<<grid, B, 0, st>>>(                          \
          c->scene, qc, qn, sq, seg, cur, cap, c->res.p, tp, prp, sum);                                           \
    else if (c->scene.num_lights == 0 && few_bounces == 2 && plain)                                               \
      rtx::k_trace_shade<B, rtx::kFewTriangles, false, false, 2, 1><<<grid, B, 0, st>>>(                          \
          c->scene, qc, qn, sq, seg, cur, cap, c->res.p, tp, prp, sum);                                           \
    else if (c->scene.num_lights == 0 && few_bounces == 2)                                                        \
      rtx::k_trace_shade<B, rtx::kFewTriangles, false, false, 2><<<grid, B, 0, st>>>(                             \
          c->scene, qc, qn, sq, seg, cur, cap, c->res.p, tp, prp, sum);                                           \
    else                                                                                                          \
      RTX_TRACE_SHADE(B, rtx::kFewTriangles, 0);                                                                  \
  } while (0)
      if (few && c->specular_surfaces)
        RTX_FEW_LAUNCH(64);
      else if (few)
        RTX_FEW_LAUNCH(128);
#undef RTX_FEW_LAUNCH
      else if (inline_light_samples && plain)  // ... and the light sample as a second round of tasks
        rtx::k_trace_shade<256, rtx::kLdsLeafTasks, true, true, 1, 1>
            <<<dim3(div_up(seg_max_upper, 256), nseg), 256,
               scene_lds + rtx::leaf_task_lds_bytes(256, c->scene.leaf_task_cap), st>>>(
                c->scene, qc, qn, sq, seg, cur, cap, c->res.p, tp, prp, sum);
      else if (inline_light_samples)
        rtx::k_trace_shade<256, rtx::kLdsLeafTasks, true, true>
            <<<dim3(div_up(seg_max_upper, 256), nseg), 256,
               scene_lds + rtx::leaf_task_lds_bytes(256, c->scene.leaf_task_cap), st>>>(
                c->scene, qc, qn, sq, seg, cur, cap, c->res.p, tp, prp, sum);
      else if (c->scene.num_leaf_boxes != 0)   // a handful of leaves: their triangle tests as tasks of the block
        RTX_TRACE_SHADE(256, rtx::kLdsLeafTasks, scene_lds + rtx::leaf_task_lds_bytes(256, c->scene.leaf_task_cap));
      else if (fused_block == 128)
        RTX_TRACE_SHADE(128, rtx::kLdsBvh2, lds_fused);
      else
        RTX_TRACE_SHADE(256, rtx::kLdsBvh2, lds_fused);
#undef RTX_TRACE_SHADE
      lt.end();
      launches++;
    } else {
    lt.begin(kIsect, active_upper);
    if (c->trace_mode == rtx::kGlobalBvh4) {
      // the launch's waves spread over the segments; never more blocks than a segment has chunks of rays
      const uint32_t per_seg = std::max(1u, div_up(kWideWaves, nseg * (kTraceBlock / 64)));
      const uint32_t gxw = std::max(1u, std::min(per_seg, div_up(seg_max_upper, kTraceBlock)));
      rtx::k_intersect_wide<kTraceBlock><<<dim3(gxw, nseg), kTraceBlock, lds, st>>>(c->scene, qc, c->hits.p, seg, cur,
                                                                                    cap, tp, c->wide_refill_lanes);
    } else {
      RTX_BY_MODE(c->trace_mode, rtx::k_intersect, kTraceBlock, (dim3(gx128, nseg), kTraceBlock, lds, st),
                  (c->scene, qc, c->hits.p, seg, cur, cap, tp, prp));
    }
    lt.end();
    lt.begin(kShade, active_upper);
    if (shadows)
      rtx::k_shade_block_append<<<dim3(gx256, nseg), 256, 0, st>>>(c->scene, qc, qn, sq, c->hits.p, seg, cur, cap,
                                                                  c->res.p, prp, sum);
    else
      rtx::k_shade_wave_append<<<dim3(gx256, nseg), 256, 0, st>>>(c->scene, qc, qn, sq, c->hits.p, seg, cur, cap,
                                                                 c->res.p, prp, sum);
    lt.end();
    launches += 2;
    }
    if (shadows && !inline_light_samples) {
      lt.begin(kShadow, active_upper);
      if (c->trace_mode == rtx::kGlobalBvh4 && c->shadow_refill_lanes > 0) {
        const uint32_t per_seg = std::max(1u, div_up(kWideWaves, nseg * (kTraceBlock / 64)));
        const uint32_t gxw = std::max(1u, std::min(per_seg, div_up(seg_max_upper, kTraceBlock)));
        rtx::k_shadow_wide<kTraceBlock><<<dim3(gxw, nseg), kTraceBlock, lds, st>>>(c->scene, qn, sq, seg, nxt, cap, tp,
                                                                                   c->shadow_refill_lanes);
      } else if (c->scene.num_leaf_boxes != 0 && fused) {
        rtx::k_shadow_leaf_tasks<256><<<dim3(gx256, nseg), 256,
                                        scene_lds + rtx::leaf_task_lds_bytes(256, c->scene.leaf_task_cap), st>>>(
            c->scene, qn, sq, seg, nxt, cap);
      } else {
        RTX_BY_MODE(c->trace_mode, rtx::k_shadow, kTraceBlock, (dim3(gx128, nseg), kTraceBlock, lds, st),
                    (c->scene, qn, sq, seg, nxt, cap, tp));
      }
      lt.end();
      launches++;
    }
    rtx::k_plan<<<1, plan_threads, 0, st>>>(seg, sum, nxt, cur, nseg, cap);
    launches++;
    if (left_upper > 0) {
      if (!pr.recompute) {
        lt.begin(kGen, (uint64_t)std::min<unsigned long long>(left_upper, pool));
        rtx::k_generate<<<dim3(cap / 256, nseg), 256, 0, st>>>(qn, seg, cam, rp, use_depth);
        lt.end();
        launches++;
      }
      seg_max_upper = cap;  // the refill may top every segment up
      active_upper = pool;
    }
    iterations++;
    since_poll++;
    cur = nxt;

    const bool tight = kTightEnd && near_the_end();
    const uint64_t poll_every = tight ? 1 : 4;
    if (since_poll >= poll_every) {
      since_poll = 0;
      const int max_pending = tight ? 1 : kPollSlots - 1;
      while (poll_pending >= max_pending) {  // wait for the oldest snapshot
        int slot = (poll_head - poll_pending + 2 * kPollSlots) % kPollSlots;
        HIP_TRY(hipEventSynchronize(c->poll_ev[slot]));
        absorb(c->h_sum[slot], snap_round[slot]);
        poll_pending--;
      }
      HIP_TRY(hipMemcpyAsync(&c->h_sum[poll_head], sum, sizeof(rtx::Summary), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipEventRecord(c->poll_ev[poll_head], st));
      snap_round[poll_head] = iterations;
      poll_head = (poll_head + 1) % kPollSlots;
      poll_pending++;
    }
    while (poll_pending > 0) {  // take every snapshot that has already landed
      int slot = (poll_head - poll_pending + 2 * kPollSlots) % kPollSlots;
      if (hipEventQuery(c->poll_ev[slot]) != hipSuccess) break;
      absorb(c->h_sum[slot], snap_round[slot]);
      poll_pending--;
    }
  }
  // Fold the chunk's samples into the sums, in sample order.
  if (after_reduce) HIP_TRY(hipStreamWaitEvent(st, after_reduce, 0));
  lt.begin(kReduce, total);
  if (reduce_split) {
    HIP_TRY(hipStreamWaitEvent(st, c->ev_join, 0));
    rtx::k_reduce_tail<<<div_up(nfloats / 4, 256), 256, 0, st>>>(c->res.p, accum_band, c->stop4.p, nfloats, spp);
  } else {
    // The sum of a frame left in flight runs beside the next frame's rounds.  At full width it takes the memory
    // system for itself (5.7 TB/s) and the bounce kernel beside it, which lives on short round trips, stands still
    // for as long: 0.6 of sample2's 5.8 ms.  On 64 blocks it trickles along at about 1 TB/s (its lanes wait for
    // their own loads), more than any workload produces results at, and the rounds beside it lose little:
    // sample2 5.79 -> 5.63 ms per frame; 3072 to 256 blocks change nothing, 32 are no better.  Same additions in
    // the same order.  RTX_REDUCE_BLOCKS: another width (0: the full one).
    uint32_t blocks = div_up(nfloats, 256);
    if (detach && (nfloats & 3u) == 0) {
      const uint32_t narrow = c->hooks.reduce_blocks >= 0 ? (uint32_t)c->hooks.reduce_blocks : 64u;
      if (narrow != 0) blocks = std::min(blocks, narrow);
    }
    rtx::k_reduce<<<blocks, 256, 0, st>>>(c->res.p, accum_band, nfloats, spp);
  }
  lt.end();
  launches++;
  if (c->ev_reduced) HIP_TRY(hipEventRecord(c->ev_reduced, st));
  // (snapshots still on their way were enqueued before this copy: it is the last to land in h_sum[0])
  HIP_TRY(hipMemcpyAsync(&c->h_sum[0], sum, sizeof(rtx::Summary), hipMemcpyDeviceToHost, st));
  if (detach) {
    HIP_TRY(hipEventRecord(c->ev_done, st));
    c->pending = true;
    c->pending_iterations = iterations;
    c->pending_launches = launches;
    c->pending_drained = finished;
    return RTX_OK;
  }
  HIP_TRY(hipStreamSynchronize(st));
  HIP_TRY(hipGetLastError());
  return absorb_summary(c, iterations, launches, finished, reduce_split);
}

}  // namespace

extern "C" {

static int ensure_twin(rtx_ctx* c);

// rtx_render (async = false: on the caller's stream, returns when the frame is done) and rtx_render_async
// (the frame goes to one of the two sets in turn, on that set's stream, and is left in flight).
static int render_frame(rtx_ctx* c, float* accum_dev, uint32_t spp, uint32_t sample_base, uint64_t seed,
                        const float cam_offset[3], uint32_t row0, uint32_t row1, void* stream, bool async) {
  if (!c || !accum_dev) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_render: null argument");
  if (!c->have_scene) return fail(RTX_ERR_NO_SCENE, "rtx_render: no scene set");
  if (row1 == 0) row1 = c->cfg.height;
  if (row0 >= row1 || row1 > c->cfg.height)
    return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_render: bad row range");
  if (spp == 0) return RTX_OK;
  DeviceGuard g(c->device);
  if (!g.ok) return fail(RTX_ERR_HIP, "rtx_render: hipSetDevice failed");
  rtx_ctx* const caller = c;
  hipStream_t st = (hipStream_t)stream;
  hipEvent_t after_reduce = nullptr;
  if (c->profiling) async = false;  // launches are timed one by one on the caller's stream
  if (!async) {
    if (int rc = wait_frames(c)) return rc;
  } else {
    if (int rc = ensure_twin(c)) return rc;
    rtx_ctx* const set = caller->next_set ? caller->twin : caller;
    rtx_ctx* const other = caller->next_set ? caller : caller->twin;
    caller->next_set ^= 1;
    if (int rc = finish_pending(set)) return rc;  // at most one frame per set
    // behind what the caller has enqueued so far, and, where it adds to the sums, behind the other set's frame
    // (an idle stream needs no marker: a marker would queue up behind whatever shares the caller's stream's
    // hardware queue, four per process here, and the frame with it)
    const hipError_t busy = hipStreamQuery(st);
    if (busy == hipErrorNotReady) {
      HIP_TRY(hipEventRecord(set->ev_in, st));
      HIP_TRY(hipStreamWaitEvent(set->own, set->ev_in, 0));
    } else if (busy != hipSuccess) {
      return fail_hip(busy, "hipStreamQuery");
    }
    if (other->pending) after_reduce = other->ev_reduced;
    st = set->own;
    c = set;
  }

  rtx::CameraDev cam{};
  cam.width = c->cfg.width;
  cam.height = c->cfg.height;
  cam.tan_x = c->tan_x;
  cam.tan_y = c->tan_y;
  cam.ox = c->cfg.cam_origin[0];
  cam.oy = c->cfg.cam_origin[1];
  cam.oz = c->cfg.cam_origin[2];
  cam.offx = cam_offset ? cam_offset[0] : 0.f;
  cam.offy = cam_offset ? cam_offset[1] : 0.f;
  cam.offz = cam_offset ? cam_offset[2] : 0.f;
  cam.type = c->cfg.camera;
  cam.focal_length = c->cfg.focal_length;
  cam.lens_radius = c->cfg.lens_radius;
  cam.stretch_x = c->cfg.stretch_x;
  cam.stretch_y = c->cfg.stretch_y;

  rtx::RenderParams rp{};
  rp.nseg = c->nseg;
  rp.cap = c->cap;
  rp.row0 = row0;
  rp.band_pixels = (row1 - row0) * c->cfg.width;
  rp.seed32 = (uint32_t)(seed ^ (seed >> 32));
  rp.seed64 = seed;
  rp.rng_streams = c->cfg.rng_streams;

  // The result buffer holds 3 floats per sample of a chunk; a call with more
  // samples than fit is rendered as several chunks of whole samples-per-pixel.
  const size_t per_spp = (size_t)rp.band_pixels * 3;
  size_t chunk_spp = c->result_budget_floats / per_spp;
  const size_t ordinal_limit = 0xffffffffull / rp.band_pixels;  // ordinals are 32-bit
  chunk_spp = std::min(chunk_spp, ordinal_limit);
  if (chunk_spp == 0)
    return fail(RTX_ERR_OUT_OF_MEMORY, "rtx_render: result_buffer_mib too small for one sample per pixel");
  chunk_spp = std::min<size_t>(chunk_spp, spp);
  if (c->res.n < chunk_spp * per_spp) {
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(c->res.alloc(chunk_spp * per_spp));
  }

  StageTimer timer;
  float* accum_band = accum_dev + (size_t)row0 * c->cfg.width * 3;
  int rc = RTX_OK;
  for (uint32_t done = 0; done < spp && rc == RTX_OK;) {
    const uint32_t n = (uint32_t)std::min<size_t>(chunk_spp, spp - done);
    rp.sample_base = sample_base + done;
    const bool last = done + n >= spp;  // (earlier chunks share the result slots with it: they are waited for)
    rc = render_chunk(c, accum_band, cam, rp, n, st, c->profiling ? &timer : nullptr, async && last, after_reduce);
    after_reduce = nullptr;
    done += n;
  }
  if (async && rc == RTX_OK) c->pending_accum = accum_dev;
  if (c->profiling) {
    for (size_t i = 0; i < timer.kind.size(); i++) {
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, timer.ev[2 * i], timer.ev[2 * i + 1]);
      switch (timer.kind[i]) {
        case kGen:
          c->stats.ms_generate += ms;
          c->stats.timed_launches_generate++;
          break;
        case kIsect:
          c->stats.ms_intersect += ms;
          c->stats.timed_launches_intersect++;
          break;
        case kShade:
          c->stats.ms_shade += ms;
          c->stats.timed_launches_shade++;
          break;
        case kShadow:
          c->stats.ms_shadow += ms;
          c->stats.timed_launches_shadow++;
          break;
        case kFinish:
          c->stats.ms_finish += ms;
          break;
        case kReduce:
          c->stats.ms_reduce += ms;
          break;
        default:
          c->stats.ms_other += ms;
      }
    }
    for (hipEvent_t e : timer.ev) c->ev_pool.push_back(e);
  }
  return rc;
}

static int impl_render(rtx_ctx* c, float* accum_dev, uint32_t spp, uint32_t sample_base, uint64_t seed,
                       const float cam_offset[3], uint32_t row0, uint32_t row1, void* stream) {
  return render_frame(c, accum_dev, spp, sample_base, seed, cam_offset, row0, row1, stream, false);
}
static int impl_render_async(rtx_ctx* c, float* accum_dev, uint32_t spp, uint32_t sample_base, uint64_t seed,
                             const float cam_offset[3], uint32_t row0, uint32_t row1, void* stream) {
  return render_frame(c, accum_dev, spp, sample_base, seed, cam_offset, row0, row1, stream, true);
}
static int impl_wait(rtx_ctx* c, const float* accum_dev) {
  if (!c) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_wait: null argument");
  return wait_frames(c, accum_dev);
}

// The scene fields of the second set: the same device memory, nothing owned.
static void mirror_scene(rtx_ctx* c) {
  rtx_ctx* t = c->twin;
  if (!t) return;
  t->have_scene = c->have_scene;
  t->scene = c->scene;
  t->info = c->info;
  t->stack_entries = c->stack_entries;
  t->trace_mode = c->trace_mode;
  t->specular_surfaces = c->specular_surfaces;
  t->lds_scene_bytes = c->lds_scene_bytes;
  t->wide_refill_lanes = c->wide_refill_lanes;
  t->shadow_refill_lanes = c->shadow_refill_lanes;
}
static int ensure_twin(rtx_ctx* c) {
  auto events = [](rtx_ctx* s) -> int {
    if (s->own) return RTX_OK;
    HIP_TRY(hipStreamCreateWithFlags(&s->own, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&s->ev_in, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&s->ev_done, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&s->ev_reduced, hipEventDisableTiming));
    return RTX_OK;
  };
  if (int rc = events(c)) return rc;
  c->owner = c;
  if (!c->twin) {
    rtx_config cfg = c->cfg;
    cfg.device = c->device;
    rtx_ctx* t = nullptr;
    if (int rc = impl_create(&cfg, &t)) return rc;
    t->owner = c;
    t->hooks = c->hooks;
    t->finish_paths = c->finish_paths;
    t->overlap_min_samples = c->overlap_min_samples;
    c->twin = t;
  }
  // (outside the block: a twin whose stream or events could not be created stays attached, and the next call
  // tries again instead of enqueuing on a null stream)
  if (int rc = events(c->twin)) return rc;
  mirror_scene(c);
  // the stacks' spill columns are per launched thread: the second set needs its own
  if (c->spill.n != c->twin->spill.n) HIP_TRY(c->twin->spill.alloc(c->spill.n));
  return RTX_OK;
}

static int impl_resolve(rtx_ctx* c, const float* accum_dev, uint32_t total_spp, float* rgb_out_host, void* stream) {
  if (!c || !accum_dev || !rgb_out_host) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_resolve: null argument");
  if (total_spp == 0) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_resolve: total_spp is 0");
  if (int rc = wait_frames(c, accum_dev)) return rc;
  DeviceGuard g(c->device);
  if (!g.ok) return fail(RTX_ERR_HIP, "rtx_resolve: hipSetDevice failed");
  hipStream_t st = (hipStream_t)stream;
  size_t n = (size_t)c->cfg.width * c->cfg.height * 3;
  rtx::k_resolve<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(accum_dev, c->resolve_tmp.p, n, (float)total_spp);
  HIP_TRY(hipMemcpyAsync(rgb_out_host, c->resolve_tmp.p, n * sizeof(float), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return RTX_OK;
}

static int impl_trace_closest(rtx_ctx* c, const float* rays_host, size_t n, uint32_t* kind_out, uint32_t* prim_out,
                      float* t_out, float* normal_out) {
  if (!c || (n && (!rays_host || !kind_out || !prim_out || !t_out)))
    return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_trace_closest: null argument");
  if (int rc = wait_frames(c)) return rc;  // (the traversal stacks' spill columns are shared with the frames)
  if (!c->have_scene) return fail(RTX_ERR_NO_SCENE, "rtx_trace_closest: no scene set");
  if (n == 0) return RTX_OK;
  if (n > 0x7fffffffu) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_trace_closest: too many rays");
  DeviceGuard g(c->device);
  if (!g.ok) return fail(RTX_ERR_HIP, "rtx_trace_closest: hipSetDevice failed");
  DevBuf<float> rays, t, nrm;
  DevBuf<uint32_t> kind, prim;
  int rc = RTX_OK;
  do {
#define TRY_BREAK(expr)               \
  {                                   \
    hipError_t e_ = (expr);           \
    if (e_ != hipSuccess) {           \
      rc = fail_hip(e_, #expr);       \
      break;                          \
    }                                 \
  }
    TRY_BREAK(rays.alloc(n * 6));
    TRY_BREAK(t.alloc(n));
    TRY_BREAK(kind.alloc(n));
    TRY_BREAK(prim.alloc(n));
    if (normal_out) TRY_BREAK(nrm.alloc(n * 3));
    TRY_BREAK(hipMemcpy(rays.p, rays_host, n * 6 * sizeof(float), hipMemcpyHostToDevice));
    size_t lds = (size_t)c->stack_entries * kTraceBlock * sizeof(int) +
                 (c->trace_mode == rtx::kLdsBvh2 ? c->lds_scene_bytes : 0);
    const rtx::TraceParams tp{c->stack_entries, c->spill.p, c->pool};
    for (size_t first = 0; first < n; first += c->pool) {  // at most pool rays per launch (spill columns)
      const uint32_t m = (uint32_t)std::min<size_t>(c->pool, n - first);
      RTX_BY_MODE(c->trace_mode, rtx::k_trace_closest, kTraceBlock, (div_up(m, kTraceBlock), kTraceBlock, lds, 0),
                  (c->scene, rays.p + 6 * first, m, kind.p + first, prim.p + first, t.p + first,
                   nrm.p ? nrm.p + 3 * first : nullptr, tp));
    }
    TRY_BREAK(hipGetLastError());
    TRY_BREAK(hipDeviceSynchronize());
    TRY_BREAK(hipMemcpy(kind_out, kind.p, n * 4, hipMemcpyDeviceToHost));
    TRY_BREAK(hipMemcpy(prim_out, prim.p, n * 4, hipMemcpyDeviceToHost));
    TRY_BREAK(hipMemcpy(t_out, t.p, n * 4, hipMemcpyDeviceToHost));
    if (normal_out) TRY_BREAK(hipMemcpy(normal_out, nrm.p, n * 12, hipMemcpyDeviceToHost));
  } while (0);
  rays.release(); t.release(); nrm.release(); kind.release(); prim.release();
  return rc;
}

// rtx_trace_closest without an allocation on the call path: the rays and the results are staged
// in this GPU's `scratch` array (through the pointer looked up at rtx_create, never through the
// array's name), the traversal stacks use the spill columns rtx_set_scene allocated.
static int impl_trace_closest_noalloc(rtx_ctx* c, const float* rays_host, size_t n, uint32_t* kind_out,
                                      uint32_t* prim_out, float* t_out, float* normal_out) {
  if (!c || (n && (!rays_host || !kind_out || !prim_out || !t_out)))
    return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_trace_closest_noalloc: null argument");
  if (int rc = wait_frames(c)) return rc;  // (the traversal stacks' spill columns are shared with the frames)
  if (!c->have_scene) return fail(RTX_ERR_NO_SCENE, "rtx_trace_closest_noalloc: no scene set");
  if (n == 0) return RTX_OK;
  if (n > kNoAllocMaxRays)
    return fail(RTX_ERR_INVALID_ARGUMENT,
                "rtx_trace_closest_noalloc: more than 262144 rays in one call (use rtx_trace_closest, or split the call)");
  DeviceGuard g(c->device);
  if (!g.ok) return fail(RTX_ERR_HIP, "rtx_trace_closest_noalloc: hipSetDevice failed");
  uint32_t* const base = g_scratch_ptr[c->device];
  if (!base) return fail(RTX_ERR_HIP, "rtx_trace_closest_noalloc: scratch address not resolved for this device");
  constexpr size_t N = kNoAllocMaxRays;
  float* rays = reinterpret_cast<float*>(base);
  float* t = reinterpret_cast<float*>(base + 6 * N);
  uint32_t* kind = base + 7 * N;
  uint32_t* prim = base + 8 * N;
  float* nrm = reinterpret_cast<float*>(base + 9 * N);
  // Every word a call reads back is written by its own kernel first (one thread per ray writes
  // t, kind, primitive and normal), so nothing of the previous call's contents has to be cleared.
  HIP_TRY(hipMemcpy(rays, rays_host, n * 6 * sizeof(float), hipMemcpyHostToDevice));
  const size_t lds = (size_t)c->stack_entries * kTraceBlock * sizeof(int) +
                     (c->trace_mode == rtx::kLdsBvh2 ? c->lds_scene_bytes : 0);
  const rtx::TraceParams tp{c->stack_entries, c->spill.p, c->pool};
  for (size_t first = 0; first < n; first += c->pool) {  // at most pool rays per launch (spill columns)
    const uint32_t m = (uint32_t)std::min<size_t>(c->pool, n - first);
    RTX_BY_MODE(c->trace_mode, rtx::k_trace_closest, kTraceBlock, (div_up(m, kTraceBlock), kTraceBlock, lds, 0),
                (c->scene, rays + 6 * first, m, kind + first, prim + first, t + first,
                 normal_out ? nrm + 3 * first : nullptr, tp));
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(kind_out, kind, n * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(prim_out, prim, n * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(t_out, t, n * 4, hipMemcpyDeviceToHost));
  if (normal_out) HIP_TRY(hipMemcpy(normal_out, nrm, n * 12, hipMemcpyDeviceToHost));
  return RTX_OK;
}

static int impl_trace_occluded(rtx_ctx* c, const float* rays_host, const float* distance_host, size_t n,
                       uint8_t* occluded_out) {
  if (!c || (n && (!rays_host || !distance_host || !occluded_out)))
    return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_trace_occluded: null argument");
  if (int rc = wait_frames(c)) return rc;  // (the traversal stacks' spill columns are shared with the frames)
  if (!c->have_scene) return fail(RTX_ERR_NO_SCENE, "rtx_trace_occluded: no scene set");
  if (n == 0) return RTX_OK;
  if (n > 0x7fffffffu) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_trace_occluded: too many rays");
  DeviceGuard g(c->device);
  if (!g.ok) return fail(RTX_ERR_HIP, "rtx_trace_occluded: hipSetDevice failed");
  DevBuf<float> rays, dist;
  DevBuf<uint8_t> occ;
  int rc = RTX_OK;
  do {
    TRY_BREAK(rays.alloc(n * 6));
    TRY_BREAK(dist.alloc(n));
    TRY_BREAK(occ.alloc(n));
    TRY_BREAK(hipMemcpy(rays.p, rays_host, n * 6 * sizeof(float), hipMemcpyHostToDevice));
    TRY_BREAK(hipMemcpy(dist.p, distance_host, n * sizeof(float), hipMemcpyHostToDevice));
    size_t lds = (size_t)c->stack_entries * kTraceBlock * sizeof(int) +
                 (c->trace_mode == rtx::kLdsBvh2 ? c->lds_scene_bytes : 0);
    const rtx::TraceParams tp{c->stack_entries, c->spill.p, c->pool};
    for (size_t first = 0; first < n; first += c->pool) {
      const uint32_t m = (uint32_t)std::min<size_t>(c->pool, n - first);
      RTX_BY_MODE(c->trace_mode, rtx::k_trace_occluded, kTraceBlock, (div_up(m, kTraceBlock), kTraceBlock, lds, 0),
                  (c->scene, rays.p + 6 * first, dist.p + first, m, occ.p + first, tp));
    }
    TRY_BREAK(hipGetLastError());
    TRY_BREAK(hipDeviceSynchronize());
    TRY_BREAK(hipMemcpy(occluded_out, occ.p, n, hipMemcpyDeviceToHost));
  } while (0);
#undef TRY_BREAK
  rays.release(); dist.release(); occ.release();
  return rc;
}

// ---- multi-GPU exchange step ----------------------------------------------------------------
static int impl_comm_create_id(char* id_out) {
  if (!id_out) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_comm_create_id: null argument");
  Rccl* r = rccl();
  if (!r) return fail(RTX_ERR_COMM, "rtx_comm_create_id: RCCL is not available on this host");
  static_assert(RTX_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
  ncclUniqueId id;
  ncclResult_t e = r->GetUniqueId(&id);
  if (e != ncclSuccess) return fail_rccl(e, "ncclGetUniqueId");
  std::memcpy(id_out, id.internal, RTX_COMM_ID_BYTES);
  return RTX_OK;
}
static int impl_comm_init(rtx_ctx* c, const char* id_bytes, int rank, int world) {
  if (!c || !id_bytes) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_comm_init: null argument");
  if (world < 1 || rank < 0 || rank >= world) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_comm_init: bad rank or world size");
  if (c->comm) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_comm_init: this context already has a communicator");
  Rccl* r = rccl();
  if (!r) return fail(RTX_ERR_COMM, "rtx_comm_init: RCCL is not available on this host");
  DeviceGuard g(c->device);
  if (!g.ok) return fail(RTX_ERR_HIP, "rtx_comm_init: hipSetDevice failed");
  ncclUniqueId id;
  std::memcpy(id.internal, id_bytes, RTX_COMM_ID_BYTES);
  ncclResult_t e = r->CommInitRank(&c->comm, world, id, rank);
  if (e != ncclSuccess) {
    c->comm = nullptr;
    return fail_rccl(e, "ncclCommInitRank");
  }
  c->comm_rank = rank;
  c->comm_world = world;
  return RTX_OK;
}
static int impl_comm_reduce(rtx_ctx* c, float* accum_dev, int root, void* stream) {
  if (!c || !accum_dev) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_comm_reduce: null argument");
  if (!c->comm) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_comm_reduce: no communicator (rtx_comm_init)");
  if (root < 0 || root >= c->comm_world) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_comm_reduce: bad root");
  if (int rc = wait_frames(c, accum_dev)) return rc;
  DeviceGuard g(c->device);
  if (!g.ok) return fail(RTX_ERR_HIP, "rtx_comm_reduce: hipSetDevice failed");
  const size_t n = (size_t)c->cfg.width * c->cfg.height * 3;
  ncclResult_t e = rccl()->Reduce(accum_dev, accum_dev, n, ncclFloat, ncclSum, root, c->comm, (hipStream_t)stream);
  if (e != ncclSuccess) return fail_rccl(e, "ncclReduce");
  return RTX_OK;
}
static int impl_comm_destroy(rtx_ctx* c) {
  if (!c) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_comm_destroy: null argument");
  if (!c->comm) return RTX_OK;
  DeviceGuard g(c->device);
  ncclResult_t e = rccl()->CommDestroy(c->comm);
  c->comm = nullptr;
  c->comm_rank = -1;
  c->comm_world = 0;
  if (e != ncclSuccess) return fail_rccl(e, "ncclCommDestroy");
  return RTX_OK;
}
static int impl_stream_sync(rtx_ctx* c, void* stream) {
  if (!c) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_stream_synchronize: null argument");
  if (int rc = wait_frames(c)) return rc;
  DeviceGuard g(c->device);
  if (!g.ok) return fail(RTX_ERR_HIP, "rtx_stream_synchronize: hipSetDevice failed");
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return RTX_OK;
}

static int impl_debug_pcg(rtx_ctx* c, uint32_t seed, uint32_t n, uint32_t* rand_out, float* uniform_out,
                          uint32_t* hash_out) {
  if (!c) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_debug_pcg: null argument");
  if (n == 0) return RTX_OK;
  if (n > (1u << 20)) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_debug_pcg: at most 2^20 draws");
  DeviceGuard g(c->device);
  if (!g.ok) return fail(RTX_ERR_HIP, "rtx_debug_pcg: hipSetDevice failed");
  DevBuf<uint32_t> buf;  // [rand | uniform bits | hash]
  HIP_TRY(buf.alloc(3 * (size_t)n));
  int rc = RTX_OK;
  do {
#define TRY_BREAK(expr)               \
  {                                   \
    hipError_t e_ = (expr);           \
    if (e_ != hipSuccess) {           \
      rc = fail_hip(e_, #expr);       \
      break;                          \
    }                                 \
  }
    rtx::k_debug_pcg<<<1, 64>>>(seed, n, buf.p, reinterpret_cast<float*>(buf.p + n), buf.p + 2 * (size_t)n);
    TRY_BREAK(hipGetLastError());
    TRY_BREAK(hipDeviceSynchronize());
    if (rand_out) TRY_BREAK(hipMemcpy(rand_out, buf.p, 4 * (size_t)n, hipMemcpyDeviceToHost));
    if (uniform_out) TRY_BREAK(hipMemcpy(uniform_out, buf.p + n, 4 * (size_t)n, hipMemcpyDeviceToHost));
    if (hash_out) TRY_BREAK(hipMemcpy(hash_out, buf.p + 2 * (size_t)n, 4 * (size_t)n, hipMemcpyDeviceToHost));
#undef TRY_BREAK
  } while (0);
  buf.release();
  return rc;
}

static int impl_debug_streams(rtx_ctx* c, uint32_t spp, uint32_t sample_base, uint64_t seed, uint64_t* out_dev,
                              void* stream) {
  if (!c || !out_dev) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_debug_streams: null argument");
  DeviceGuard g(c->device);
  if (!g.ok) return fail(RTX_ERR_HIP, "rtx_debug_streams: hipSetDevice failed");
  rtx::CameraDev cam{};
  cam.width = c->cfg.width;
  cam.height = c->cfg.height;
  rtx::RenderParams rp{};
  rp.row0 = 0;
  rp.band_pixels = c->cfg.width * c->cfg.height;
  rp.sample_base = sample_base;
  rp.seed32 = (uint32_t)(seed ^ (seed >> 32));
  rp.seed64 = seed;
  rp.rng_streams = c->cfg.rng_streams;
  const unsigned long long n = (unsigned long long)rp.band_pixels * spp;
  if (n == 0) return RTX_OK;
  if ((n + 255) / 256 > 0x7fffffffull) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_debug_streams: too many samples in one call");
  rtx::k_debug_streams<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      cam, rp, n, reinterpret_cast<unsigned long long*>(out_dev));
  HIP_TRY(hipGetLastError());
  return RTX_OK;
}

static int impl_get_stats(rtx_ctx* c, rtx_stats* out) {
  if (!c || !out) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_get_stats: null argument");
  if (int rc = wait_frames(c)) return rc;
  *out = c->stats;
  out->device_allocations = g_device_allocations.load(std::memory_order_relaxed);
  return RTX_OK;
}
static int impl_reset_stats(rtx_ctx* c) {
  if (!c) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_reset_stats: null argument");
  if (int rc = wait_frames(c)) return rc;
  c->stats = rtx_stats{};
  return RTX_OK;
}
static int impl_set_profiling(rtx_ctx* c, int on) {
  if (!c) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_set_profiling: null argument");
  if (int rc = wait_frames(c)) return rc;
  c->profiling = on != 0;
  return RTX_OK;
}

static int impl_accum_alloc(rtx_ctx* c, float** out) {
  if (!c || !out) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_accum_alloc: null argument");
  DeviceGuard g(c->device);
  if (!g.ok) return fail(RTX_ERR_HIP, "rtx_accum_alloc: hipSetDevice failed");
  size_t bytes = (size_t)c->cfg.width * c->cfg.height * 3 * sizeof(float);
  g_device_allocations.fetch_add(1, std::memory_order_relaxed);
  HIP_TRY(hipMalloc((void**)out, bytes));
  HIP_TRY(hipMemset(*out, 0, bytes));
  return RTX_OK;
}
static int impl_accum_free(rtx_ctx* c, float* p) {
  if (!c) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_accum_free: null argument");
  if (int rc = wait_frames(c, p)) return rc;
  DeviceGuard g(c->device);
  if (p) HIP_TRY(hipFree(p));
  return RTX_OK;
}
static int impl_accum_clear(rtx_ctx* c, float* p, void* stream) {
  if (!c || !p) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_accum_clear: null argument");
  if (int rc = wait_frames(c, p)) return rc;
  DeviceGuard g(c->device);
  size_t bytes = (size_t)c->cfg.width * c->cfg.height * 3 * sizeof(float);
  HIP_TRY(hipMemsetAsync(p, 0, bytes, (hipStream_t)stream));
  return RTX_OK;
}

static int impl_accum_upload(rtx_ctx* c, float* p, const float* sums_host, void* stream) {
  if (!c || !p || !sums_host) return fail(RTX_ERR_INVALID_ARGUMENT, "rtx_accum_upload: null argument");
  if (int rc = wait_frames(c, p)) return rc;
  DeviceGuard g(c->device);
  if (!g.ok) return fail(RTX_ERR_HIP, "rtx_accum_upload: hipSetDevice failed");
  size_t bytes = (size_t)c->cfg.width * c->cfg.height * 3 * sizeof(float);
  HIP_TRY(hipMemcpyAsync(p, sums_host, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return RTX_OK;
}

}  // extern "C"

#ifdef RTX_PHASE_STATS
// Diagnostic build only: reads and clears the phase clocks of the leaf-task bounce kernel (g_phase_stats).
extern "C" int rtx_debug_phase_stats(unsigned long long out[16]) {
  static const unsigned long long zero[16] = {0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rtx::g_phase_stats), sizeof(zero)) != hipSuccess) return RTX_ERR_HIP;
  if (hipMemcpyToSymbol(HIP_SYMBOL(rtx::g_phase_stats), zero, sizeof(zero)) != hipSuccess) return RTX_ERR_HIP;
  return RTX_OK;
}
#endif
#ifdef RTX_WALK_STATS
// Diagnostic build only: reads and clears the counters of k_intersect_wide's waves (kernels.h, g_walk_stats).
extern "C" int rtx_debug_walk_stats(unsigned long long out[8]) {
  unsigned long long zero[8] = {};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rtx::g_walk_stats), sizeof(zero)) != hipSuccess) return RTX_ERR_HIP;
  if (hipMemcpyToSymbol(HIP_SYMBOL(rtx::g_walk_stats), zero, sizeof(zero)) != hipSuccess) return RTX_ERR_HIP;
  return RTX_OK;
}
#endif

// The exported entry points: nothing thrown inside the library (std::bad_alloc from a vector,
// std::system_error from a worker thread) crosses the C boundary; it becomes a status code.
namespace {
template <typename F>
int guarded(const char* what, F&& body) {
  try {
    return body();
  } catch (const std::bad_alloc&) {
    return fail_nothrow(RTX_ERR_OUT_OF_MEMORY, what, "host allocation failed");
  } catch (const std::exception& e) {
    return fail_nothrow(RTX_ERR_HIP, what, e.what());
  } catch (...) {
    return fail_nothrow(RTX_ERR_HIP, what, "unknown exception");
  }
}
}  // namespace

extern "C" {

int rtx_create(const rtx_config* cfg, rtx_ctx** out) {
  return guarded("rtx_create", [&] { return impl_create(cfg, out); });
}
int rtx_set_scene(rtx_ctx* c, const rtx_scene* s) {
  return guarded("rtx_set_scene", [&] { return impl_set_scene(c, s); });
}
int rtx_get_scene_info(rtx_ctx* c, rtx_scene_info* out) {
  return guarded("rtx_get_scene_info", [&] { return impl_get_scene_info(c, out); });
}
int rtx_render(rtx_ctx* c, float* accum_dev, uint32_t spp, uint32_t sample_base, uint64_t seed, const float cam_offset[3], uint32_t row0, uint32_t row1, void* stream) {
  return guarded("rtx_render", [&] { return impl_render(c, accum_dev, spp, sample_base, seed, cam_offset, row0, row1, stream); });
}
int rtx_render_async(rtx_ctx* c, float* accum_dev, uint32_t spp, uint32_t sample_base, uint64_t seed, const float cam_offset[3], uint32_t row0, uint32_t row1, void* stream) {
  return guarded("rtx_render_async", [&] { return impl_render_async(c, accum_dev, spp, sample_base, seed, cam_offset, row0, row1, stream); });
}
int rtx_wait(rtx_ctx* c, const float* accum_dev) {
  return guarded("rtx_wait", [&] { return impl_wait(c, accum_dev); });
}
int rtx_resolve(rtx_ctx* c, const float* accum_dev, uint32_t total_spp, float* rgb_out_host, void* stream) {
  return guarded("rtx_resolve", [&] { return impl_resolve(c, accum_dev, total_spp, rgb_out_host, stream); });
}
int rtx_trace_closest(rtx_ctx* c, const float* rays_host, size_t n, uint32_t* kind_out, uint32_t* prim_out, float* t_out, float* normal_out) {
  return guarded("rtx_trace_closest", [&] { return impl_trace_closest(c, rays_host, n, kind_out, prim_out, t_out, normal_out); });
}
int rtx_trace_closest_noalloc(rtx_ctx* c, const float* rays_host, size_t n, uint32_t* kind_out, uint32_t* prim_out, float* t_out, float* normal_out) {
  return guarded("rtx_trace_closest_noalloc", [&] { return impl_trace_closest_noalloc(c, rays_host, n, kind_out, prim_out, t_out, normal_out); });
}
int rtx_trace_occluded(rtx_ctx* c, const float* rays_host, const float* distance_host, size_t n, uint8_t* occluded_out) {
  return guarded("rtx_trace_occluded", [&] { return impl_trace_occluded(c, rays_host, distance_host, n, occluded_out); });
}
int rtx_comm_create_id(char id_out[RTX_COMM_ID_BYTES]) {
  return guarded("rtx_comm_create_id", [&] { return impl_comm_create_id(id_out); });
}
int rtx_comm_init(rtx_ctx* c, const char id[RTX_COMM_ID_BYTES], int rank, int world_size) {
  return guarded("rtx_comm_init", [&] { return impl_comm_init(c, id, rank, world_size); });
}
int rtx_comm_reduce(rtx_ctx* c, float* accum_dev, int root, void* stream) {
  return guarded("rtx_comm_reduce", [&] { return impl_comm_reduce(c, accum_dev, root, stream); });
}
int rtx_comm_destroy(rtx_ctx* c) {
  return guarded("rtx_comm_destroy", [&] { return impl_comm_destroy(c); });
}
int rtx_stream_synchronize(rtx_ctx* c, void* stream) {
  return guarded("rtx_stream_synchronize", [&] { return impl_stream_sync(c, stream); });
}
int rtx_debug_pcg(rtx_ctx* c, uint32_t seed, uint32_t n, uint32_t* rand_out, float* uniform_out, uint32_t* hash_out) {
  return guarded("rtx_debug_pcg", [&] { return impl_debug_pcg(c, seed, n, rand_out, uniform_out, hash_out); });
}
int rtx_debug_streams(rtx_ctx* c, uint32_t spp, uint32_t sample_base, uint64_t seed, uint64_t* out_dev, void* stream) {
  return guarded("rtx_debug_streams", [&] { return impl_debug_streams(c, spp, sample_base, seed, out_dev, stream); });
}
int rtx_get_stats(rtx_ctx* c, rtx_stats* out) {
  return guarded("rtx_get_stats", [&] { return impl_get_stats(c, out); });
}
int rtx_reset_stats(rtx_ctx* c) {
  return guarded("rtx_reset_stats", [&] { return impl_reset_stats(c); });
}
int rtx_set_profiling(rtx_ctx* c, int on) {
  return guarded("rtx_set_profiling", [&] { return impl_set_profiling(c, on); });
}
int rtx_accum_alloc(rtx_ctx* c, float** out) {
  return guarded("rtx_accum_alloc", [&] { return impl_accum_alloc(c, out); });
}
int rtx_accum_free(rtx_ctx* c, float* p) {
  return guarded("rtx_accum_free", [&] { return impl_accum_free(c, p); });
}
int rtx_accum_clear(rtx_ctx* c, float* p, void* stream) {
  return guarded("rtx_accum_clear", [&] { return impl_accum_clear(c, p, stream); });
}
int rtx_accum_upload(rtx_ctx* c, float* p, const float* sums_host, void* stream) {
  return guarded("rtx_accum_upload", [&] { return impl_accum_upload(c, p, sums_host, stream); });
}

}  // extern "C"
