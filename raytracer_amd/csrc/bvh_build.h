// bvh_build.h -- host-side builder of the flat SoA BVH the HIP kernels traverse.
//
// Stands in for rtcCommitScene (reference Scene.h:123): the reference hands its
// triangles to Embree 3; CDNA4 has no ray-tracing units, so the library builds
// its own binned-SAH BVH2 and flattens it into 64-byte nodes that carry both
// children's boxes (one node fetch = two box tests).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace rtx {

// One flattened node = 4 x float4 (64 bytes):
//   q0 = (c0.lo.x, c0.hi.x, c0.lo.y, c0.hi.y)
//   q1 = (c1.lo.x, c1.hi.x, c1.lo.y, c1.hi.y)
//   q2 = (c0.lo.z, c0.hi.z, c1.lo.z, c1.hi.z)
//   q3 = (child0, child1, 0, 0) as int bits.
// child >= 0: index of an inner node.  child < 0: leaf, v = ~child,
// first triangle slot = v >> 3, triangle count = (v & 7) + 1.
// The root is always an inner node with two children; a scene without
// triangles has no nodes at all.
struct FlatBvh {
  std::vector<float> nodes;     // 16 floats per node; node 0 is the root
  std::vector<uint32_t> order;  // triangle slot -> primitive id (index given by the caller)
  uint32_t max_depth = 0;       // longest root-to-leaf path in inner nodes (bounds the traversal stack)
};

// tris: ntri*9 floats (a,b,c).  max_leaf (1..8): a node with more triangles is
// always split.  A box test costs about a third of a triangle test, so scenes
// that are staged in LDS (instruction-bound) use 2; scenes whose nodes stream
// from L2/HBM (bandwidth-bound) use 4, which halves the node count.  Boxes are padded by a few 1e-6 relative so that
// a hit accepted by the float triangle test is never culled by rounding in the
// slab test.  Deterministic for a given input.
void build_bvh(const float* tris, size_t ntri, uint32_t max_leaf, FlatBvh& out);

}  // namespace rtx
