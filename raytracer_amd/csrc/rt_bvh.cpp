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

// ---- Reinsertion pass (after Bittner, Hapala, Havran: "Fast insertion-based optimization of bounding volume
// hierarchies", CGF 2013).  The top-down binned build fixes every split before it knows what lies below it; this pass
// takes subtrees out, largest boxes first, and puts each back where it enlarges the tree's boxes least — found by
// branch and bound over the induced cost (the growth of the ancestors' areas), the old position among the candidates,
// so the summed area of the inner nodes (SAH with one primitive per leaf) never grows.  The tree's depth is held to the
// built tree's: the traversal stacks in LDS are sized by it.  Closest hits do not depend on the topology.
struct Optimiser {
  struct TN {
    Box box;
    int parent, child[2];
    int32_t prim;  // >= 0: leaf
    int height;    // levels below this node (leaf: 0), kept current by refit_up
  };
  const std::vector<Box>& boxes;
  std::vector<TN> t;
  int root = -1;

  explicit Optimiser(const std::vector<Box>& b) : boxes(b) {}
  static double area(const Box& b) { return Builder::area(b); }
  static Box merged(const Box& a, const Box& b) { Box m = a; Builder::grow(m, b); return m; }

  int import(const std::vector<FlatNode>& nodes, int32_t ref, int parent) {
    const int me = (int)t.size();
    t.emplace_back();
    t[(size_t)me].parent = parent;
    if (ref < 0) {
      t[(size_t)me].prim = ~ref;
      t[(size_t)me].child[0] = t[(size_t)me].child[1] = -1;
      t[(size_t)me].box = boxes[(size_t)~ref];
      t[(size_t)me].height = 0;
      return me;
    }
    const int l = import(nodes, nodes[(size_t)ref].left, me), r = import(nodes, nodes[(size_t)ref].right, me);
    TN& n = t[(size_t)me];
    n.prim = -1; n.child[0] = l; n.child[1] = r;
    n.box = merged(t[(size_t)l].box, t[(size_t)r].box);
    n.height = 1 + std::max(t[(size_t)l].height, t[(size_t)r].height);
    return me;
  }
  void refit_up(int n) {
    for (; n >= 0; n = t[(size_t)n].parent) {
      const TN &l = t[(size_t)t[(size_t)n].child[0]], &r = t[(size_t)t[(size_t)n].child[1]];
      t[(size_t)n].box = merged(l.box, r.box);
      t[(size_t)n].height = 1 + std::max(l.height, r.height);
    }
  }
  int depth_of(int n) const { int d = 0; for (n = t[(size_t)n].parent; n >= 0; n = t[(size_t)n].parent) d++; return d; }
  int height_of(int n) const { return t[(size_t)n].height; }
  double inner_area() const {
    double s = 0;
    for (const TN& n : t) if (n.prim < 0 && n.parent >= -1) s += area(n.box);
    return s;
  }
  // takes n (and its parent p) out, returns p for reuse; n's sibling takes p's place
  int detach(int n) {
    const int p = t[(size_t)n].parent, g = t[(size_t)p].parent;
    const int s = t[(size_t)p].child[0] == n ? t[(size_t)p].child[1] : t[(size_t)p].child[0];
    t[(size_t)s].parent = g;
    if (g < 0) root = s;
    else { t[(size_t)g].child[t[(size_t)g].child[0] == p ? 0 : 1] = s; refit_up(g); }
    return p;
  }
  // p becomes the parent of x and n, in x's place
  void attach(int n, int p, int x) {
    const int g = t[(size_t)x].parent;
    t[(size_t)p].parent = g;
    t[(size_t)p].child[0] = x; t[(size_t)p].child[1] = n;
    t[(size_t)x].parent = p; t[(size_t)n].parent = p;
    if (g < 0) root = p;
    else t[(size_t)g].child[t[(size_t)g].child[0] == x ? 0 : 1] = p;
    refit_up(p);
  }
  // where n's subtree (height h) costs least, among positions that keep the tree within max_depth levels
  int best_position(int n, int h, int max_levels) const {
    struct Item { double induced; int node, depth; };
    auto later = [](const Item& a, const Item& b) { return a.induced > b.induced; };
    std::vector<Item> heap;
    heap.push_back({0.0, root, 0});
    const Box& nb = t[(size_t)n].box;
    const double na = area(nb);
    double best_cost = INF;
    int best = -1;
    while (!heap.empty()) {
      std::pop_heap(heap.begin(), heap.end(), later);
      const Item it = heap.back();
      heap.pop_back();
      if (it.induced + na >= best_cost) break;
      const TN& x = t[(size_t)it.node];
      const double direct = area(merged(x.box, nb));
      // the new parent sits at x's depth; below it: x's subtree and n's, each one level deeper
      if (it.induced + direct < best_cost && it.depth + 1 + h <= max_levels &&
          (x.prim >= 0 || it.depth + 1 + height_of(it.node) <= max_levels)) {
        best_cost = it.induced + direct;
        best = it.node;
      }
      if (x.prim < 0) {
        const double below = it.induced + direct - area(x.box);
        if (below + na < best_cost)
          for (int k = 0; k < 2; k++) {
            heap.push_back({below, x.child[k], it.depth + 1});
            std::push_heap(heap.begin(), heap.end(), later);
          }
      }
    }
    return best;
  }
  void run(int passes, int max_levels) {
    std::vector<int> order;
    for (int pass = 0; pass < passes; pass++) {
      order.clear();
      for (int i = 0; i < (int)t.size(); i++)
        if (i != root && t[(size_t)i].parent != root) order.push_back(i);
      std::sort(order.begin(), order.end(), [&](int a, int b) {
        const double aa = area(t[(size_t)a].box), ab = area(t[(size_t)b].box);
        return aa != ab ? aa > ab : a < b;
      });
      const double before = inner_area();
      for (int n : order) {
        if (n == root || t[(size_t)n].parent == root || t[(size_t)n].parent < 0) continue;
        const int h = height_of(n);
        const int old_sibling = t[(size_t)t[(size_t)n].parent].child[0] == n ? t[(size_t)t[(size_t)n].parent].child[1]
                                                                              : t[(size_t)t[(size_t)n].parent].child[0];
        const int p = detach(n);
        int x = best_position(n, h, max_levels);
        if (x < 0) x = old_sibling;  // (depth limit left no candidate: back where it was)
        attach(n, p, x);
      }
      if (!(inner_area() < before * (1.0 - 1e-4))) break;
    }
  }
  int depth() const { return height_of(root); }
  // back to the flat form, numbered in the order the builder numbers (pre-order, left subtree first)
  int32_t flatten(int n, std::vector<FlatNode>& out) const {
    const TN& tn = t[(size_t)n];
    if (tn.prim >= 0) return ~tn.prim;
    const int32_t me = (int32_t)out.size();
    out.emplace_back();
    const int32_t l = flatten(tn.child[0], out), r = flatten(tn.child[1], out);
    FlatNode& nd = out[(size_t)me];
    Builder::store(t[(size_t)tn.child[0]].box, nd.l_lo, nd.l_hi);
    Builder::store(t[(size_t)tn.child[1]].box, nd.r_lo, nd.r_hi);
    nd.left = l; nd.right = r;
    nd.pad[0] = nd.pad[1] = 0;
    return me;
  }
};

}  // namespace

BuildResult build_bvh(const std::vector<Box>& boxes, int max_depth, int optimise_passes) {
  BuildResult res;
  if (boxes.empty()) return res;
  Builder b(boxes, max_depth);
  b.nodes.reserve(boxes.size());
  Box root_box;
  res.root = b.build(0, boxes.size(), 0, root_box);
  res.nodes = std::move(b.nodes);
  res.depth = b.depth_seen;
  // (≈25 ms for 1 400 primitives, ≈190 ms for 8 192 on the build host: larger scenes keep the built tree)
  if (optimise_passes > 0 && res.root >= 0 && boxes.size() >= 8 && boxes.size() <= 8192) {
    Optimiser o(boxes);
    o.t.reserve(2 * boxes.size());
    o.root = o.import(res.nodes, res.root, -1);
    o.run(optimise_passes, res.depth);
    if (o.depth() <= res.depth) {
      std::vector<FlatNode> flat;
      flat.reserve(res.nodes.size());
      const int32_t r = o.flatten(o.root, flat);
      if (r >= 0 && flat.size() == res.nodes.size()) { res.nodes.swap(flat); res.root = r; res.depth = o.depth(); }
    }
  }
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
