// rt_bvh.h — host-side acceleration-structure build for librt_mi355x.
// Replaces BVHNode::inner_from_vec (reference src/hittable/containers.rs:58-152).  The reference splits at the
// mean centroid of the widest axis; closest-hit results do not depend on topology (containers.rs:177-191), so
// the library is free to build a better tree: binned-SAH BVH2, one primitive per leaf, children boxes stored
// in the parent (one 64-byte fetch per traversal step), boxes rounded OUTWARD to f32 and padded so the f32
// slab test is conservative with respect to the f64 geometry.
#pragma once
#include <cstdint>
#include <vector>

namespace rtb {

struct Box {
  double lo[3], hi[3];
};
struct FlatNode {  // mirrors rtd::BvhNode (64 B)
  float l_lo[3], l_hi[3], r_lo[3], r_hi[3];
  int32_t left, right;
  int32_t pad[2];
};
static const int32_t EMPTY_CHILD = INT32_MIN;

struct BuildResult {
  std::vector<FlatNode> nodes;
  int32_t root = EMPTY_CHILD;  // >= 0: node index; < 0: ~prim (single primitive)
  int depth = 0;
};

// boxes[i] = f64 bounds of primitive i.  max_depth bounds the tree depth (traversal stack size).
// optimise_passes > 0: that many rounds of the reinsertion pass afterwards (rt_bvh.cpp); the depth never grows.
BuildResult build_bvh(const std::vector<Box>& boxes, int max_depth, int optimise_passes = 0);


// Four-wide form of the same tree for the instances that traverse out of HBM/L2 (one 128-byte line per step, about
// half as many dependent fetches per ray as the binary tree).  Structure of arrays: per axis the four children's
// planes sit in one float4.  Unused slots have child == EMPTY_CHILD and an inverted (+inf, -inf) box.
struct WideNode {  // mirrors the device layout (128 B)
  float lo[3][4], hi[3][4];
  int32_t child[4];  // >= 0: wide node index; < 0: ~prim; EMPTY_CHILD: unused slot
  int32_t pad[4];
};
struct WideResult {
  std::vector<WideNode> nodes;  // root is node 0 (when the binary tree has an inner root)
  int stack_need = 0;           // most entries a nearest-first traversal can have pending
};
// Collapses the binary tree: each wide node adopts the grandchildren of its largest inner children until it has four.
WideResult collapse_bvh4(const BuildResult& bvh);

}  // namespace rtb
