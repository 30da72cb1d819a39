// Test-only probe around the product's host-side BVH builder
// (raytracer_amd/csrc/bvh_build.cpp): builds a tree and copies the flat arrays out
// so that a CPU test can check the structure without a GPU.
#include <cstring>

#include "bvh_build.h"

extern "C" {
// Returns the number of nodes; call once with null outputs to size the buffers.
int bvh_probe(const float* tris, size_t ntri, unsigned max_leaf, unsigned width, float* nodes_out,
              unsigned* order_out, unsigned* depth_out, unsigned* stack_out) {
  rtx::FlatBvh b;
  rtx::build_bvh(tris, ntri, max_leaf, width, b);
  if (nodes_out) std::memcpy(nodes_out, b.nodes.data(), b.nodes.size() * sizeof(float));
  if (order_out) std::memcpy(order_out, b.order.data(), b.order.size() * sizeof(unsigned));
  if (depth_out) *depth_out = b.max_depth;
  if (stack_out) *stack_out = b.max_stack;
  return (int)(b.nodes.size() / 16);
}
}
