"""The reinsertion pass of the host BVH builder (rt_bvh.cpp, option bvh_optimise), on the CPU: a small C++ harness
compiled against rt_bvh.cpp checks, for random boxes, that the optimised tree (1) holds every primitive exactly once,
(2) has the node count and at most the depth of the built tree (the traversal stacks in LDS are sized by the depth),
(3) encloses every primitive in every ancestor's stored f32 box (conservative rounding), (4) has a smaller or equal sum
of inner-node areas, and (5) is built in bounded time."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HARNESS = r"""
#include "rt_bvh.h"
#include <chrono>
#include <cstdio>
#include <random>
#include <vector>
using namespace rtb;
static double area(const float* lo, const float* hi) {
  double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
  return 2.0 * (dx * dy + dy * dz + dz * dx);
}
struct Check { const std::vector<Box>& b; const BuildResult& r; std::vector<int> seen; int bad_box = 0; double inner = 0; };
// returns the depth below `ref`; lo / hi: the box the parent stores for this child
static int walk(Check& c, int32_t ref, const float* lo, const float* hi) {
  if (ref < 0) {
    const int p = ~ref;
    c.seen[p]++;
    for (int a = 0; a < 3; a++) if (lo && (c.b[p].lo[a] < lo[a] || c.b[p].hi[a] > hi[a])) c.bad_box++;
    return 0;
  }
  const FlatNode& n = c.r.nodes[(size_t)ref];
  if (lo) {
    c.inner += area(lo, hi);
    for (int a = 0; a < 3; a++) if (n.l_lo[a] < lo[a] || n.r_lo[a] < lo[a] || n.l_hi[a] > hi[a] || n.r_hi[a] > hi[a]) c.bad_box++;
  }
  const int dl = walk(c, n.left, n.l_lo, n.l_hi), dr = walk(c, n.right, n.r_lo, n.r_hi);
  return 1 + (dl > dr ? dl : dr);
}
int main() {
  int fail = 0;
  for (int n : {9, 64, 484, 1409, 5000}) {
    std::mt19937 rng(11 + n);
    std::uniform_real_distribution<double> U(-100, 100), R(0.1, 2.0);
    std::vector<Box> b(n);
    for (auto& x : b) { double c[3] = {U(rng), U(rng) * 0.05, U(rng)}, r = R(rng); for (int a = 0; a < 3; a++) { x.lo[a] = c[a] - r; x.hi[a] = c[a] + r; } }
    b[0] = Box{{-1000, -2000, -1000}, {1000, 0, 1000}};  // one huge primitive
    BuildResult r0 = build_bvh(b, 32, 0);
    auto t0 = std::chrono::steady_clock::now();
    BuildResult r3 = build_bvh(b, 32, 3);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    Check c0{b, r0, std::vector<int>(n, 0)}, c3{b, r3, std::vector<int>(n, 0)};
    const int d0 = walk(c0, r0.root, nullptr, nullptr), d3 = walk(c3, r3.root, nullptr, nullptr);
    int missing = 0;
    for (int s : c3.seen) missing += s != 1;
    const bool ok = missing == 0 && c3.bad_box == 0 && r3.nodes.size() == r0.nodes.size() && d3 <= d0 && r3.depth == d3 &&
                    c3.inner <= c0.inner * (1 + 1e-9) && ms < 2000;
    printf("n=%d depth %d -> %d inner area %.1f -> %.1f  %.1f ms  %s\n", n, d0, d3, c0.inner, c3.inner, ms, ok ? "ok" : "FAIL");
    fail += !ok;
  }
  return fail;
}
"""


def test_reinsertion_pass_keeps_the_tree_valid_and_no_deeper():
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "bvh_check.cpp")
        exe = os.path.join(td, "bvh_check")
        open(src, "w").write(HARNESS)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "raytracer_amd", "csrc"), src,
                               os.path.join(ROOT, "raytracer_amd", "csrc", "rt_bvh.cpp"), "-o", exe])
        out = subprocess.run([exe], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
        assert out.stdout.count(" ok") == 5, out.stdout
