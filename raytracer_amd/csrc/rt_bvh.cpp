// rt_bvh.cpp — binned-SAH BVH2 builder (host).  See rt_bvh.h.
#include "rt_bvh.h"

#include <algorithm>
#include <cmath>
#include <limits>
#include <numeric>

namespace rtb {
namespace {

const double INF = std::numeric_limits<double>::infinity();

struct Builder {
  const std::vector<Box>& boxes;
  std::vector<uint32_t> idx;
  std::vector<FlatNode> nodes;
  int max_depth;
  int depth_seen = 0;

  Builder(const std::vector<Box>& b, int md) : boxes(b), idx(b.size()), max_depth(md) {
    std::iota(idx.begin(), idx.end(), 0u);
  }

  static Box empty_box() {
    Box b;
    for (int a = 0; a < 3; a++) { b.lo[a] = INF; b.hi[a] = -INF; }
    return b;
  }
  static void grow(Box& b, const Box& o) {
    for (int a = 0; a < 3; a++) { b.lo[a] = std::fmin(b.lo[a], o.lo[a]); b.hi[a] = std::fmax(b.hi[a], o.hi[a]); }
  }
  static double area(const Box& b) {
    double dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    if (!(dx >= 0) || !(dy >= 0) || !(dz >= 0)) return 0.0;
    return 2.0 * (dx * dy + dy * dz + dz * dx);
  }
  double centroid(uint32_t i, int a) const { return 0.5 * (boxes[i].lo[a] + boxes[i].hi[a]); }

  // conservative f32 image of an f64 box: pad by 2^-20 relative to the coordinate magnitude (covers the f32
  // rounding of the plane itself and of plane*inv in the slab test), then round outward.
  static float down(double v) {
    double p = v - (std::fabs(v) * 9.5367431640625e-07 + 1e-30);
    float f = (float)p;
    if ((double)f > p) f = std::nextafterf(f, -std::numeric_limits<float>::infinity());
    return f;
  }
  static float up(double v) {
    double p = v + (std::fabs(v) * 9.5367431640625e-07 + 1e-30);
    float f = (float)p;
    if ((double)f < p) f = std::nextafterf(f, std::numeric_limits<float>::infinity());
    return f;
  }
  static void store(const Box& b, float* lo, float* hi) {
    for (int a = 0; a < 3; a++) { lo[a] = down(b.lo[a]); hi[a] = up(b.hi[a]); }
  }

  // builds idx[begin,end); returns child reference and its bounds
  int32_t build(size_t begin, size_t end, int depth, Box& out_box) {
    depth_seen = std::max(depth_seen, depth);
    size_t n = end - begin;
    if (n == 1) {
      out_box = boxes[idx[begin]];
      return ~(int32_t)idx[begin];
    }
    Box cb = empty_box();
    for (size_t i = begin; i < end; i++)
      for (int a = 0; a < 3; a++) {
        double c = centroid(idx[i], a);
        cb.lo[a] = std::fmin(cb.lo[a], c);
        cb.hi[a] = std::fmax(cb.hi[a], c);
      }
    size_t mid = begin + n / 2;
    bool split_done = false;
    // depth budget: a balanced median split of n items needs ceil(log2 n) levels
    int need = 0;
    while (((size_t)1 << need) < n) need++;
    bool force_median = depth + need + 1 >= max_depth;
    if (!force_median) {
      const int NB = 32;
      double best_cost = INF;
      int best_axis = -1, best_bin = -1;
      for (int a = 0; a < 3; a++) {
        double ext = cb.hi[a] - cb.lo[a];
        if (!(ext > 0)) continue;
        Box bb[NB];
        size_t cnt[NB];
        for (int b = 0; b < NB; b++) { bb[b] = empty_box(); cnt[b] = 0; }
        double scale = NB / ext;
        for (size_t i = begin; i < end; i++) {
          int b = (int)((centroid(idx[i], a) - cb.lo[a]) * scale);
          b = std::min(std::max(b, 0), NB - 1);
          grow(bb[b], boxes[idx[i]]);
          cnt[b]++;
        }
        double ra[NB];
        size_t rc[NB];
        Box acc = empty_box();
        size_t c = 0;
        for (int b = NB - 1; b > 0; b--) {
          grow(acc, bb[b]);
          c += cnt[b];
          ra[b] = area(acc);
          rc[b] = c;
        }
        acc = empty_box();
        c = 0;
        for (int b = 0; b < NB - 1; b++) {
          grow(acc, bb[b]);
          c += cnt[b];
          if (c == 0 || rc[b + 1] == 0) continue;
          double cost = area(acc) * (double)c + ra[b + 1] * (double)rc[b + 1];
          if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; }
        }
      }
      if (best_axis >= 0) {
        int a = best_axis;
        double ext = cb.hi[a] - cb.lo[a];
        double scale = 32 / ext;
        auto it = std::partition(idx.begin() + begin, idx.begin() + end, [&](uint32_t i) {
          int b = (int)((centroid(i, a) - cb.lo[a]) * scale);
          b = std::min(std::max(b, 0), 31);
          return b <= best_bin;
        });
        size_t m = (size_t)(it - idx.begin());
        if (m > begin && m < end) { mid = m; split_done = true; }
      }
    }
    if (!split_done) {
      // median split along the widest centroid axis (or index order if all centroids coincide)
      int a = 0;
      double best = -1;
      for (int k = 0; k < 3; k++) {
        double e = cb.hi[k] - cb.lo[k];
        if (e > best) { best = e; a = k; }
      }
      mid = begin + n / 2;
      if (best > 0)
        std::nth_element(idx.begin() + begin, idx.begin() + mid, idx.begin() + end,
                         [&](uint32_t x, uint32_t y) { return centroid(x, a) < centroid(y, a); });
    }
    int32_t me = (int32_t)nodes.size();
    nodes.emplace_back();
    Box lb, rb;
    int32_t l = build(begin, mid, depth + 1, lb);
    int32_t r = build(mid, end, depth + 1, rb);
    FlatNode& nd = nodes[(size_t)me];
    store(lb, nd.l_lo, nd.l_hi);
    store(rb, nd.r_lo, nd.r_hi);
    nd.left = l;
    nd.right = r;
    nd.pad[0] = nd.pad[1] = 0;
    out_box = lb;
    grow(out_box, rb);
    return me;
  }
};

}  // namespace

BuildResult build_bvh(const std::vector<Box>& boxes, int max_depth) {
  BuildResult res;
  if (boxes.empty()) return res;
  Builder b(boxes, max_depth);
  b.nodes.reserve(boxes.size());
  Box root_box;
  res.root = b.build(0, boxes.size(), 0, root_box);
  res.nodes = std::move(b.nodes);
  res.depth = b.depth_seen;
  return res;
}


namespace {
struct Cand {
  int32_t ref;
  float lo[3], hi[3];
};
float half_area(const Cand& c) {
  const float dx = c.hi[0] - c.lo[0], dy = c.hi[1] - c.lo[1], dz = c.hi[2] - c.lo[2];
  return dx * dy + dy * dz + dz * dx;
}
void children_of(const FlatNode& n, Cand& l, Cand& r) {
  l.ref = n.left;
  r.ref = n.right;
  for (int a = 0; a < 3; a++) {
    l.lo[a] = n.l_lo[a]; l.hi[a] = n.l_hi[a];
    r.lo[a] = n.r_lo[a]; r.hi[a] = n.r_hi[a];
  }
}
// returns the wide node's index; `need` = stack entries a traversal below this node can have pending
int32_t collapse_rec(const BuildResult& bvh, int32_t n2, WideResult& out, int& need) {
  Cand c[4];
  int nc = 2;
  children_of(bvh.nodes[(size_t)n2], c[0], c[1]);
  while (nc < 4) {
    int best = -1;
    float best_area = -1.f;
    for (int k = 0; k < nc; k++)
      if (c[k].ref >= 0 && half_area(c[k]) > best_area) { best = k; best_area = half_area(c[k]); }
    if (best < 0) break;
    const int32_t opened = c[best].ref;
    children_of(bvh.nodes[(size_t)opened], c[best], c[nc]);
    nc++;
  }
  const int32_t me = (int32_t)out.nodes.size();
  out.nodes.emplace_back();
  int below = 0, used = 0;
  int32_t child[4];
  for (int k = 0; k < 4; k++) {
    child[k] = EMPTY_CHILD;
    if (k >= nc || c[k].ref == EMPTY_CHILD) continue;
    used++;
    if (c[k].ref >= 0) {
      int sub = 0;
      child[k] = collapse_rec(bvh, c[k].ref, out, sub);
      below = std::max(below, sub);
    } else {
      child[k] = c[k].ref;
    }
  }
  WideNode& w = out.nodes[(size_t)me];  // taken after the recursion: the vector may have grown
  for (int k = 0; k < 4; k++) {
    const bool live = child[k] != EMPTY_CHILD;
    for (int a = 0; a < 3; a++) {
      w.lo[a][k] = live ? c[k].lo[a] : std::numeric_limits<float>::infinity();
      w.hi[a][k] = live ? c[k].hi[a] : -std::numeric_limits<float>::infinity();
    }
    w.child[k] = child[k];
    w.pad[k] = 0;
  }
  need = std::max(used - 1, 0) + below;
  return me;
}
}  // namespace

WideResult collapse_bvh4(const BuildResult& bvh) {
  WideResult out;
  if (bvh.root < 0) return out;  // empty scene or a single primitive: nothing to collapse
  out.nodes.reserve(bvh.nodes.size() / 2 + 1);
  collapse_rec(bvh, bvh.root, out, out.stack_need);
  // Renumber breadth-first: node indices grow with depth, so "index < n" selects the top of the tree (the part the
  // fused kernel keeps in LDS) and siblings share cache lines' neighbourhoods.
  std::vector<int32_t> order, new_of(out.nodes.size(), -1);
  order.reserve(out.nodes.size());
  order.push_back(0);
  for (size_t head = 0; head < order.size(); head++)
    for (int k = 0; k < 4; k++) {
      const int32_t c = out.nodes[(size_t)order[head]].child[k];
      if (c >= 0) order.push_back(c);
    }
  for (size_t i = 0; i < order.size(); i++) new_of[(size_t)order[i]] = (int32_t)i;
  std::vector<WideNode> sorted(order.size());
  for (size_t i = 0; i < order.size(); i++) {
    sorted[i] = out.nodes[(size_t)order[i]];
    for (int k = 0; k < 4; k++)
      if (sorted[i].child[k] >= 0) sorted[i].child[k] = new_of[(size_t)sorted[i].child[k]];
  }
  out.nodes.swap(sorted);
  return out;
}

}  // namespace rtb
