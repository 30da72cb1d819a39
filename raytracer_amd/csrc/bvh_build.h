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
//
// Wide form (width 4), used for scenes that stream from L2/HBM: the same tree
// with every inner node's grandchildren pulled up until it has up to four
// children, and the children's boxes quantised to 8 bits per plane relative to
// the node's own box, so a node is still 64 bytes but covers two levels:
//   q0 = (origin.x, origin.y, origin.z, ex | ey<<8 | ez<<16)   scale_k = 2^(e_k-127)
//   q1 = (lo.x[4], lo.y[4], lo.z[4], hi.x[4])   one byte per child, child i in bits 8i..8i+7
//   q2 = (hi.y[4], hi.z[4], 0, 0)
//   q3 = child0..child3 (same encoding as above; 0x80000000 = no child)
// child plane = origin_k + byte * scale_k; lo bytes are rounded down and hi
// bytes up, so the decoded box contains the (padded) child box.
struct FlatBvh {
  std::vector<float> nodes;     // 16 floats per node; node 0 is the root
  std::vector<uint32_t> order;  // triangle slot -> primitive id (index given by the caller)
  uint32_t max_depth = 0;       // longest root-to-leaf path in inner nodes
  uint32_t max_stack = 0;       // bound on the traversal stack: max_depth pushes of width-1 entries
  uint32_t width = 2;
};

// tris: ntri*9 floats (a,b,c).  max_leaf (1..8): a node with more triangles is
// always split.  A box test costs about a third of a triangle test; the library
// uses 2 for every scene.  Boxes are padded by a few 1e-6 relative so that
// a hit accepted by the float triangle test is never culled by rounding in the
// slab test.  Deterministic for a given input.
void build_bvh(const float* tris, size_t ntri, uint32_t max_leaf, uint32_t width, FlatBvh& out);

}  // namespace rtx
