// Test helper (not part of the product): builds a BVH with the library's host builder (raytracer_amd/csrc/rt_bvh.cpp)
// over caller-supplied boxes and reports structural facts, so that the builder is covered by the CPU test suite.
#include <cmath>
#include <cstdint>
#include <vector>

#include "../raytracer_amd/csrc/rt_bvh.h"

extern "C" {

// out[0] = nodes, out[1] = depth, out[2] = leaves reached, out[3] = primitives referenced more or less than once,
// out[4] = child boxes that fail to contain their subtree's f64 primitive boxes, out[5] = root (as int64),
// out[6] = max stack entries a depth-first traversal that defers one child per node can hold
int bvh_check(const double* lohi, uint64_t n, int max_depth, int64_t* out) {
  std::vector<rtb::Box> boxes(n);
  for (uint64_t i = 0; i < n; i++)
    for (int a = 0; a < 3; a++) { boxes[i].lo[a] = lohi[6 * i + a]; boxes[i].hi[a] = lohi[6 * i + 3 + a]; }
  rtb::BuildResult r = rtb::build_bvh(boxes, max_depth);
  std::vector<int> seen(n, 0);
  int64_t leaves = 0, bad_box = 0, max_stack = 0;
  // recursive walk: returns the f64 bounds of the subtree
  struct Walk {
    const rtb::BuildResult& r; const std::vector<rtb::Box>& boxes; std::vector<int>& seen;
    int64_t &leaves, &bad_box, &max_stack;
    rtb::Box go(int32_t ref, int64_t pending) {
      if (ref < 0) {
        const uint32_t p = (uint32_t)~ref;
        seen[p]++; leaves++;
        if (pending > max_stack) max_stack = pending;
        return boxes[p];
      }
      const rtb::FlatNode& nd = r.nodes[(size_t)ref];
      rtb::Box lb = go(nd.left, pending + 1), rb = go(nd.right, pending);
      for (int a = 0; a < 3; a++) {
        if (!((double)nd.l_lo[a] <= lb.lo[a] && (double)nd.l_hi[a] >= lb.hi[a])) bad_box++;
        if (!((double)nd.r_lo[a] <= rb.lo[a] && (double)nd.r_hi[a] >= rb.hi[a])) bad_box++;
      }
      rtb::Box u;
      for (int a = 0; a < 3; a++) { u.lo[a] = std::fmin(lb.lo[a], rb.lo[a]); u.hi[a] = std::fmax(lb.hi[a], rb.hi[a]); }
      return u;
    }
  };
  if (n) {
    Walk w{r, boxes, seen, leaves, bad_box, max_stack};
    w.go(r.root, 0);
  }
  int64_t wrong = 0;
  for (uint64_t i = 0; i < n; i++) wrong += seen[i] != 1;
  out[0] = (int64_t)r.nodes.size(); out[1] = r.depth; out[2] = leaves; out[3] = wrong; out[4] = bad_box;
  out[5] = r.root; out[6] = max_stack;
  return 0;
}

// Four-wide collapse of the same tree.  out[0] = wide nodes, out[1] = stack_need reported, out[2] = leaves reached,
// out[3] = primitives referenced more or less than once, out[4] = slot boxes that fail to contain their subtree,
// out[5] = most entries any visiting order can have pending (children - 1 summed along a path), out[6] = unused
// slots whose box is not inverted (+inf, -inf), out[7] = average children per wide node x 1000
int bvh4_check(const double* lohi, uint64_t n, int max_depth, int64_t* out) {
  std::vector<rtb::Box> boxes(n);
  for (uint64_t i = 0; i < n; i++)
    for (int a = 0; a < 3; a++) { boxes[i].lo[a] = lohi[6 * i + a]; boxes[i].hi[a] = lohi[6 * i + 3 + a]; }
  rtb::BuildResult r = rtb::build_bvh(boxes, max_depth);
  rtb::WideResult w = rtb::collapse_bvh4(r);
  std::vector<int> seen(n, 0);
  int64_t leaves = 0, bad_box = 0, max_pending = 0, bad_empty = 0, children = 0;
  struct Walk {
    const rtb::WideResult& w; const std::vector<rtb::Box>& boxes; std::vector<int>& seen;
    int64_t &leaves, &bad_box, &max_pending, &bad_empty, &children;
    rtb::Box go(int32_t ref, int64_t pending) {
      if (ref < 0) {
        const uint32_t p = (uint32_t)~ref;
        seen[p]++; leaves++;
        if (pending > max_pending) max_pending = pending;
        return boxes[p];
      }
      const rtb::WideNode& nd = w.nodes[(size_t)ref];
      int used = 0;
      for (int k = 0; k < 4; k++) used += nd.child[k] != rtb::EMPTY_CHILD;
      children += used;
      rtb::Box u;
      for (int a = 0; a < 3; a++) { u.lo[a] = INFINITY; u.hi[a] = -INFINITY; }
      for (int k = 0; k < 4; k++) {
        if (nd.child[k] == rtb::EMPTY_CHILD) {
          for (int a = 0; a < 3; a++)
            if (!(nd.lo[a][k] == INFINITY && nd.hi[a][k] == -INFINITY)) bad_empty++;
          continue;
        }
        rtb::Box b = go(nd.child[k], pending + used - 1);
        for (int a = 0; a < 3; a++) {
          if (!((double)nd.lo[a][k] <= b.lo[a] && (double)nd.hi[a][k] >= b.hi[a])) bad_box++;
          u.lo[a] = std::fmin(u.lo[a], b.lo[a]); u.hi[a] = std::fmax(u.hi[a], b.hi[a]);
        }
      }
      return u;
    }
  };
  if (r.root >= 0) {
    Walk wk{w, boxes, seen, leaves, bad_box, max_pending, bad_empty, children};
    wk.go(0, 0);
  } else if (n == 1) {
    seen[0]++; leaves++;
  }
  int64_t wrong = 0;
  for (uint64_t i = 0; i < n; i++) wrong += seen[i] != 1;
  out[0] = (int64_t)w.nodes.size(); out[1] = w.stack_need; out[2] = leaves; out[3] = wrong; out[4] = bad_box;
  out[5] = max_pending; out[6] = bad_empty;
  out[7] = w.nodes.empty() ? 0 : children * 1000 / (int64_t)w.nodes.size();
  return 0;
}
}
