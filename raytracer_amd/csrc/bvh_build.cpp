// bvh_build.cpp -- binned-SAH BVH2 builder and flattener (host code).
// See bvh_build.h for the node format and what it replaces in the reference.
#include "bvh_build.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <limits>
#include <new>
#include <thread>

namespace rtx {
namespace {

constexpr int kBins = 16;
constexpr uint32_t kMaxSahDepth = 40;  // below this depth, fall back to median splits
constexpr size_t kParallelMinTriangles = 1u << 16;  // smaller meshes build on the calling thread
constexpr size_t kParallelSubtrees = 128;           // subtrees of at most ntri/128 triangles go to workers

struct Box {
  float lo[3], hi[3];
  void reset() {
    for (int k = 0; k < 3; k++) {
      lo[k] = std::numeric_limits<float>::infinity();
      hi[k] = -std::numeric_limits<float>::infinity();
    }
  }
  void grow(const Box& b) {
    for (int k = 0; k < 3; k++) {
      lo[k] = std::min(lo[k], b.lo[k]);
      hi[k] = std::max(hi[k], b.hi[k]);
    }
  }
  void grow_point(const float* p) {
    for (int k = 0; k < 3; k++) {
      lo[k] = std::min(lo[k], p[k]);
      hi[k] = std::max(hi[k], p[k]);
    }
  }
  float half_area() const {
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    if (!(dx >= 0.f) || !(dy >= 0.f) || !(dz >= 0.f)) return 0.f;
    return dx * dy + dy * dz + dz * dx;
  }
};

struct TmpNode {
  Box box;
  int32_t left = -1, right = -1;  // inner when left >= 0
  uint32_t first = 0, count = 0;
};

struct Builder {
  const std::vector<Box>& tb;      // per-triangle bounds
  const std::vector<float>& cent;  // per-triangle centroid (3 floats)
  std::vector<uint32_t>& order;
  std::vector<TmpNode> nodes;

  uint32_t kMaxLeaf;  // a node with more triangles is always split

  // Top of a parallel build: a subtree of at most `defer_below` triangles is not built but
  // noted in `deferred` (its root stays a placeholder) for a worker to build on its own.
  struct Deferred {
    int32_t node;
    uint32_t first, count, depth;
  };
  uint32_t defer_below = 0;
  std::vector<Deferred> deferred;

  Builder(const std::vector<Box>& tb_, const std::vector<float>& c_, std::vector<uint32_t>& o_, uint32_t max_leaf)
      : tb(tb_), cent(c_), order(o_), kMaxLeaf(max_leaf) {}

  int32_t build(uint32_t first, uint32_t count, uint32_t depth) {
    int32_t id = (int32_t)nodes.size();
    nodes.emplace_back();
    if (count <= defer_below && count > kMaxLeaf) {
      deferred.push_back({id, first, count, depth});
      return id;
    }
    Box box, cbox;
    box.reset();
    cbox.reset();
    for (uint32_t i = first; i < first + count; i++) {
      box.grow(tb[order[i]]);
      cbox.grow_point(&cent[3 * (size_t)order[i]]);
    }
    nodes[id].box = box;
    nodes[id].first = first;
    nodes[id].count = count;
    if (count <= kMaxLeaf) return id;

    uint32_t mid = 0;
    bool have_split = false;
    if (depth < kMaxSahDepth) {
      // Binned SAH over the three axes of the centroid box.
      float best_cost = std::numeric_limits<float>::infinity();
      int best_axis = -1, best_bin = -1;
      for (int axis = 0; axis < 3; axis++) {
        float ext = cbox.hi[axis] - cbox.lo[axis];
        if (!(ext > 0.f)) continue;
        float scale = (float)kBins / ext;
        Box bb[kBins];
        uint32_t bc[kBins];
        for (int b = 0; b < kBins; b++) {
          bb[b].reset();
          bc[b] = 0;
        }
        for (uint32_t i = first; i < first + count; i++) {
          uint32_t p = order[i];
          int b = (int)((cent[3 * (size_t)p + axis] - cbox.lo[axis]) * scale);
          b = std::min(std::max(b, 0), kBins - 1);
          bb[b].grow(tb[p]);
          bc[b]++;
        }
        float right_area[kBins];
        uint32_t right_cnt[kBins];
        Box acc;
        acc.reset();
        uint32_t n = 0;
        for (int b = kBins - 1; b > 0; b--) {
          acc.grow(bb[b]);
          n += bc[b];
          right_area[b] = acc.half_area();
          right_cnt[b] = n;
        }
        acc.reset();
        n = 0;
        for (int b = 0; b < kBins - 1; b++) {
          acc.grow(bb[b]);
          n += bc[b];
          if (n == 0 || right_cnt[b + 1] == 0) continue;
          float cost = acc.half_area() * (float)n + right_area[b + 1] * (float)right_cnt[b + 1];
          if (cost < best_cost) {
            best_cost = cost;
            best_axis = axis;
            best_bin = b;
          }
        }
      }
      if (best_axis >= 0) {
        float ext = cbox.hi[best_axis] - cbox.lo[best_axis];
        float scale = (float)kBins / ext;
        float lo = cbox.lo[best_axis];
        auto it = std::partition(order.begin() + first, order.begin() + first + count,
                                 [&](uint32_t p) {
                                   int b = (int)((cent[3 * (size_t)p + best_axis] - lo) * scale);
                                   b = std::min(std::max(b, 0), kBins - 1);
                                   return b <= best_bin;
                                 });
        mid = (uint32_t)(it - order.begin());
        have_split = mid > first && mid < first + count;
      }
    }
    if (!have_split) {
      // Median split on the widest centroid axis (ties and degenerate inputs by index).
      int axis = 0;
      for (int k = 1; k < 3; k++)
        if (cbox.hi[k] - cbox.lo[k] > cbox.hi[axis] - cbox.lo[axis]) axis = k;
      mid = first + count / 2;
      std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count,
                       [&](uint32_t p, uint32_t q) {
                         float cp = cent[3 * (size_t)p + axis], cq = cent[3 * (size_t)q + axis];
                         return cp < cq || (cp == cq && p < q);
                       });
    }
    int32_t l = build(first, mid - first, depth + 1);
    int32_t r = build(mid, first + count - mid, depth + 1);
    nodes[id].left = l;
    nodes[id].right = r;
    return id;
  }
};

inline float as_float(int32_t v) {
  float f;
  std::memcpy(&f, &v, 4);
  return f;
}

struct Flattener {
  const std::vector<TmpNode>& tmp;
  std::vector<float>& out;
  uint32_t max_depth = 0;

  static void padded(const Box& b, float lo[3], float hi[3]) {
    float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    float diag = std::sqrt(dx * dx + dy * dy + dz * dz);
    for (int k = 0; k < 3; k++) {
      float pad = 2e-6f * std::max(std::fabs(b.lo[k]), std::fabs(b.hi[k])) + 1e-5f * diag + 1e-30f;
      lo[k] = b.lo[k] - pad;
      hi[k] = b.hi[k] + pad;
    }
  }
  int32_t child_ref(int32_t t, uint32_t depth) {
    const TmpNode& n = tmp[t];
    if (n.left < 0) {
      uint32_t v = (n.first << 3) | (n.count - 1);
      return ~(int32_t)v;
    }
    return emit_inner(t, depth);
  }
  // Emits tmp node t (an inner node) and its subtree; returns its flat index.
  int32_t emit_inner(int32_t t, uint32_t depth) {
    int32_t idx = (int32_t)(out.size() / 16);
    out.resize(out.size() + 16, 0.f);
    max_depth = std::max(max_depth, depth + 1);
    const TmpNode& n = tmp[t];
    float l0[3], h0[3], l1[3], h1[3];
    padded(tmp[n.left].box, l0, h0);
    padded(tmp[n.right].box, l1, h1);
    int32_t c0 = child_ref(n.left, depth + 1);
    int32_t c1 = child_ref(n.right, depth + 1);
    float* q = &out[(size_t)idx * 16];
    q[0] = l0[0]; q[1] = h0[0]; q[2] = l0[1]; q[3] = h0[1];
    q[4] = l1[0]; q[5] = h1[0]; q[6] = l1[1]; q[7] = h1[1];
    q[8] = l0[2]; q[9] = h0[2]; q[10] = l1[2]; q[11] = h1[2];
    q[12] = as_float(c0);
    q[13] = as_float(c1);
    return idx;
  }
};

// Width-4 flattening with 8-bit child boxes (format: bvh_build.h).
struct Flattener4 {
  const std::vector<TmpNode>& tmp;
  std::vector<float>& out;
  uint32_t max_depth = 0;

  static int32_t leaf_ref(const TmpNode& n) { return ~(int32_t)((n.first << 3) | (n.count - 1)); }

  int32_t emit(int32_t t, uint32_t depth) {
    const int32_t idx = (int32_t)(out.size() / 16);
    out.resize(out.size() + 16, 0.f);
    max_depth = std::max(max_depth, depth + 1);
    // Pull grandchildren up, largest box first, until there are four children.
    int32_t kids[4] = {tmp[t].left, tmp[t].right, -1, -1};
    int nk = 2;
    while (nk < 4) {
      int best = -1;
      float best_area = -1.f;
      for (int k = 0; k < nk; k++)
        if (tmp[kids[k]].left >= 0 && tmp[kids[k]].box.half_area() > best_area) {
          best_area = tmp[kids[k]].box.half_area();
          best = k;
        }
      if (best < 0) break;
      const int32_t inner = kids[best];
      kids[best] = tmp[inner].left;
      kids[nk++] = tmp[inner].right;
    }
    float lo[4][3], hi[4][3], plo[3], phi[3];
    for (int k = 0; k < 3; k++) {
      plo[k] = std::numeric_limits<float>::infinity();
      phi[k] = -std::numeric_limits<float>::infinity();
    }
    for (int c = 0; c < nk; c++) {
      Flattener::padded(tmp[kids[c]].box, lo[c], hi[c]);
      for (int k = 0; k < 3; k++) {
        plo[k] = std::min(plo[k], lo[c][k]);
        phi[k] = std::max(phi[k], hi[c][k]);
      }
    }
    uint32_t ebyte[3];
    float scale[3];
    for (int k = 0; k < 3; k++) {
      // smallest power of two with 255 * scale >= extent (in float arithmetic)
      int e = 0;
      float ext = phi[k] - plo[k];
      std::frexp(ext / 255.0f, &e);  // ext/255 = m * 2^e, m in [0.5, 1)  ->  2^e > ext/255
      int biased = std::min(std::max(e + 127, 1), 254);
      uint32_t bits = (uint32_t)biased << 23;
      std::memcpy(&scale[k], &bits, 4);
      while (plo[k] + 255.0f * scale[k] < phi[k] && biased < 254) {  // rounding safety
        biased++;
        bits = (uint32_t)biased << 23;
        std::memcpy(&scale[k], &bits, 4);
      }
      ebyte[k] = (uint32_t)biased;
    }
    uint32_t qlo[3] = {0, 0, 0}, qhi[3] = {0, 0, 0};
    for (int c = 0; c < 4; c++) {
      for (int k = 0; k < 3; k++) {
        uint32_t a = 255, b = 0;  // absent child: inverted box
        if (c < nk) {
          int ia = (int)std::floor((lo[c][k] - plo[k]) / scale[k]);
          int ib = (int)std::ceil((hi[c][k] - plo[k]) / scale[k]);
          ia = std::min(std::max(ia, 0), 255);
          ib = std::min(std::max(ib, 0), 255);
          // The decoded planes must enclose the child box in float arithmetic too.
          while (ia > 0 && plo[k] + (float)ia * scale[k] > lo[c][k]) ia--;
          while (ib < 255 && plo[k] + (float)ib * scale[k] < hi[c][k]) ib++;
          a = (uint32_t)ia;
          b = (uint32_t)ib;
        }
        qlo[k] |= a << (8 * c);
        qhi[k] |= b << (8 * c);
      }
    }
    int32_t refs[4];
    for (int c = 0; c < 4; c++) {
      if (c >= nk)
        refs[c] = (int32_t)0x80000000u;
      else if (tmp[kids[c]].left < 0)
        refs[c] = leaf_ref(tmp[kids[c]]);
      else
        refs[c] = emit(kids[c], depth + 1);
    }
    float* q = &out[(size_t)idx * 16];
    auto put_u = [&](int i, uint32_t v) { std::memcpy(&q[i], &v, 4); };
    q[0] = plo[0];
    q[1] = plo[1];
    q[2] = plo[2];
    put_u(3, ebyte[0] | (ebyte[1] << 8) | (ebyte[2] << 16));
    put_u(4, qlo[0]);
    put_u(5, qlo[1]);
    put_u(6, qlo[2]);
    put_u(7, qhi[0]);
    put_u(8, qhi[1]);
    put_u(9, qhi[2]);
    put_u(10, 0);
    put_u(11, 0);
    for (int c = 0; c < 4; c++) put_u(12 + c, (uint32_t)refs[c]);
    return idx;
  }
};

}  // namespace

void build_bvh(const float* tris, size_t ntri, uint32_t max_leaf, uint32_t width, FlatBvh& out) {
  const uint32_t kMaxLeaf = max_leaf < 1 ? 1u : (max_leaf > 8 ? 8u : max_leaf);
  out.nodes.clear();
  out.order.clear();
  out.max_depth = 0;
  out.max_stack = 0;
  out.width = width == 4 ? 4 : 2;
  if (ntri == 0) return;  // no mesh: the kernels skip the traversal when there are no nodes
  std::vector<Box> tb(ntri);
  std::vector<float> cent(3 * ntri);
  for (size_t i = 0; i < ntri; i++) {
    tb[i].reset();
    tb[i].grow_point(tris + 9 * i);
    tb[i].grow_point(tris + 9 * i + 3);
    tb[i].grow_point(tris + 9 * i + 6);
    for (int k = 0; k < 3; k++) cent[3 * i + k] = 0.5f * tb[i].lo[k] + 0.5f * tb[i].hi[k];
  }
  out.order.resize(ntri);
  for (size_t i = 0; i < ntri; i++) out.order[i] = (uint32_t)i;
  Builder b(tb, cent, out.order, kMaxLeaf);
  b.nodes.reserve(ntri);
  // Large meshes: the top of the tree here, the subtrees below it on worker threads.  Each worker
  // permutes only its own range of `order`, and the subtrees are spliced in in a fixed order, so
  // the tree does not depend on the number of threads.
  const uint32_t workers = ntri >= kParallelMinTriangles ? std::min(16u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
  if (workers > 1) b.defer_below = (uint32_t)(ntri / kParallelSubtrees);
  int32_t root = b.build(0, (uint32_t)ntri, 0);
  if (!b.deferred.empty()) {
    std::vector<std::vector<TmpNode>> sub(b.deferred.size());
    std::atomic<size_t> next_task{0};
    // A worker never lets an exception escape (that would end the process): it notes the
    // failure and stops taking tasks; the calling thread rethrows after every worker is joined.
    std::atomic<bool> failed{false};
    auto work = [&]() noexcept {
      try {
        for (size_t k; !failed.load() && (k = next_task.fetch_add(1)) < b.deferred.size();) {
          const Builder::Deferred& d = b.deferred[k];
          Builder w(tb, cent, out.order, kMaxLeaf);
          w.nodes.reserve(d.count);
          w.build(d.first, d.count, d.depth);
          sub[k] = std::move(w.nodes);
        }
      } catch (...) {
        failed.store(true);
      }
    };
    // Thread creation can be refused (a process limit): the workers that did start, and this
    // thread, then share the tasks; the shared task counter makes any number of takers correct.
    std::vector<std::thread> pool;
    try {
      pool.reserve(workers);
      for (uint32_t t = 1; t < workers; t++) pool.emplace_back(work);
    } catch (...) {
    }
    work();
    for (std::thread& t : pool) t.join();
    if (failed.load()) throw std::bad_alloc();
    for (size_t k = 0; k < sub.size(); k++) {
      // local node 0 takes the placeholder's place, local node i > 0 goes to base + i - 1
      const int32_t base = (int32_t)b.nodes.size();
      auto moved = [&](int32_t child) { return child < 0 ? child : base + child - 1; };
      for (size_t i = 0; i < sub[k].size(); i++) {
        TmpNode n = sub[k][i];
        n.left = moved(n.left);
        n.right = moved(n.right);
        if (i == 0)
          b.nodes[b.deferred[k].node] = n;
        else
          b.nodes.push_back(n);
      }
    }
  }

  // Leaves keep their triangles in ascending primitive order (helps the
  // lowest-index tie-break find the winner first).
  for (const TmpNode& n : b.nodes)
    if (n.left < 0) std::sort(out.order.begin() + n.first, out.order.begin() + n.first + n.count);

  if (out.width == 4 && b.nodes[root].left >= 0) {
    Flattener4 f4{b.nodes, out.nodes};
    out.nodes.reserve(16 * ntri / 4 + 16);
    f4.emit(root, 0);
    out.max_depth = f4.max_depth;
    out.max_stack = 3 * f4.max_depth;
    return;
  }
  out.width = 2;  // (a mesh that fits one leaf keeps the two-child root below)
  Flattener f{b.nodes, out.nodes};
  out.nodes.reserve(16 * ntri / 2 + 16);
  if (b.nodes[root].left < 0) {
    // The whole mesh fits one leaf.  The root is always an inner node, so cut
    // the leaf in two (a single triangle is referenced by both children; testing
    // it twice cannot change the result).
    uint32_t cnt = b.nodes[root].count, h = cnt > 1 ? cnt / 2 : 1;
    uint32_t first1 = cnt > 1 ? h : 0, cnt1 = cnt > 1 ? cnt - h : 1;
    Box b0, b1;
    b0.reset();
    b1.reset();
    for (uint32_t i = 0; i < h; i++) b0.grow(tb[out.order[i]]);
    for (uint32_t i = first1; i < first1 + cnt1; i++) b1.grow(tb[out.order[i]]);
    float l0[3], h0[3], l1[3], h1[3];
    Flattener::padded(b0, l0, h0);
    Flattener::padded(b1, l1, h1);
    out.nodes.assign(16, 0.f);
    float* q = out.nodes.data();
    q[0] = l0[0]; q[1] = h0[0]; q[2] = l0[1]; q[3] = h0[1];
    q[4] = l1[0]; q[5] = h1[0]; q[6] = l1[1]; q[7] = h1[1];
    q[8] = l0[2]; q[9] = h0[2]; q[10] = l1[2]; q[11] = h1[2];
    q[12] = as_float(~(int32_t)((0u << 3) | (h - 1)));
    q[13] = as_float(~(int32_t)((first1 << 3) | (cnt1 - 1)));
    out.max_depth = 1;
    out.max_stack = 1;
    return;
  }
  f.emit_inner(root, 0);
  out.max_depth = f.max_depth;
  out.max_stack = f.max_depth;
}

}  // namespace rtx
