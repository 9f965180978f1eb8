"""CPU tests of the library's host-side BVH builder (raytracer_amd/csrc/rt_bvh.cpp), through a small C helper
compiled here (tests/bvh_check.cpp): every primitive in exactly one leaf, child boxes conservative with respect to
the f64 primitive boxes, depth within the traversal stack, degenerate inputs."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def chk(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("bvh") / "libbvh_check.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, os.path.join(HERE, "bvh_check.cpp"),
                           os.path.join(ROOT, "raytracer_amd", "csrc", "rt_bvh.cpp")])
    lib = C.CDLL(so)
    lib.bvh_check.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]

    def run(lohi, max_depth=32):
        lohi = np.ascontiguousarray(lohi, dtype=np.float64).reshape(-1, 6)
        out = np.zeros(7, dtype=np.int64)
        assert lib.bvh_check(lohi.ctypes.data, len(lohi), max_depth, out.ctypes.data) == 0
        return dict(zip(["nodes", "depth", "leaves", "wrong", "bad_box", "root", "max_stack"], out.tolist()))
    lib.bvh4_check.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]

    def run4(lohi, max_depth=32):
        lohi = np.ascontiguousarray(lohi, dtype=np.float64).reshape(-1, 6)
        out = np.zeros(8, dtype=np.int64)
        assert lib.bvh4_check(lohi.ctypes.data, len(lohi), max_depth, out.ctypes.data) == 0
        return dict(zip(["nodes", "stack_need", "leaves", "wrong", "bad_box", "max_pending", "bad_empty", "fill_x1000"],
                        out.tolist()))
    run.wide = run4
    return run


def sphere_boxes(centers, radii):
    centers, radii = np.asarray(centers, float), np.asarray(radii, float).reshape(-1, 1)
    return np.hstack([centers - radii, centers + radii])


def test_book_cover_like_field(chk):
    rng = np.random.default_rng(1)
    a, b = np.meshgrid(np.arange(-11, 11), np.arange(-11, 11))
    c = np.stack([a.ravel() + 0.9 * rng.random(484), np.full(484, 0.2), b.ravel() + 0.9 * rng.random(484)], axis=1)
    boxes = np.vstack([sphere_boxes(c, np.full(484, 0.2)), sphere_boxes([[0, -1000, 0]], [1000])])
    r = chk(boxes)
    assert r["nodes"] == 484 and r["leaves"] == 485 and r["wrong"] == 0 and r["bad_box"] == 0
    assert r["depth"] < 32 and r["max_stack"] <= r["depth"]  # what the device stack is sized by
    assert r["depth"] <= 16  # SAH keeps a regular field shallow (the C2 scene builds to depth 11)


def test_single_and_pair(chk):
    r = chk(sphere_boxes([[1, 2, 3]], [0.5]))
    assert r["nodes"] == 0 and r["root"] == ~0 and r["leaves"] == 1 and r["wrong"] == 0  # a lone primitive is the root leaf
    r = chk(sphere_boxes([[0, 0, 0], [5, 0, 0]], [1, 1]))
    assert r["nodes"] == 1 and r["leaves"] == 2 and r["depth"] == 1 and r["bad_box"] == 0


def test_coincident_centroids_and_flat_boxes(chk):
    # all centroids equal (concentric shells) and zero-thickness boxes (axis-aligned quads): no split plane exists
    shells = sphere_boxes(np.zeros((64, 3)), np.linspace(0.1, 6.4, 64))
    r = chk(shells)
    assert r["leaves"] == 64 and r["wrong"] == 0 and r["bad_box"] == 0 and r["depth"] <= 8  # median splits by index
    quads = np.array([[0, 0, i, 1, 1, i] for i in range(33)], float)
    r = chk(quads)
    assert r["leaves"] == 33 and r["wrong"] == 0 and r["bad_box"] == 0


def test_depth_budget_is_respected_on_adversarial_input(chk):
    # geometrically growing boxes make SAH peel one primitive per level; the builder must stay inside the stack
    n = 200
    c = np.stack([2.0 ** np.arange(n) * 1e-20, np.zeros(n), np.zeros(n)], axis=1)
    boxes = sphere_boxes(c, 2.0 ** np.arange(n) * 0.4e-20)
    for cap in (32, 16, 12):
        r = chk(boxes, max_depth=cap)
        assert r["leaves"] == n and r["wrong"] == 0 and r["bad_box"] == 0 and r["depth"] < cap


def test_large_coordinates_stay_conservative_in_f32(chk):
    rng = np.random.default_rng(7)
    c = rng.uniform(-1, 1, (300, 3)) * np.array([4e7, 4e7, 4e7]) + 0.123456789
    boxes = sphere_boxes(c, rng.uniform(0.01, 1e6, 300))
    r = chk(boxes)
    assert r["leaves"] == 300 and r["wrong"] == 0 and r["bad_box"] == 0


def test_four_wide_collapse(chk):
    """rt_bvh.cpp collapse_bvh4: what the HBM-resident fused instances traverse."""
    rng = np.random.default_rng(3)
    a, b = np.meshgrid(np.arange(-11, 11), np.arange(-11, 11))
    c = np.stack([a.ravel() + 0.9 * rng.random(484), np.full(484, 0.2), b.ravel() + 0.9 * rng.random(484)], axis=1)
    field = np.vstack([sphere_boxes(c, np.full(484, 0.2)), sphere_boxes([[0, -1000, 0]], [1000])])
    cloud = sphere_boxes(rng.uniform(-100, 100, (3000, 3)), rng.uniform(0.1, 10, 3000))
    quads = np.array([[0, 0, i, 1, 1, i] for i in range(33)], float)
    for boxes in (field, cloud, quads, sphere_boxes([[0, 0, 0], [5, 0, 0]], [1, 1]), sphere_boxes(np.zeros((3, 3)), [1, 2, 3])):
        r2, r = chk(boxes), chk.wide(boxes)
        assert r["leaves"] == len(boxes) and r["wrong"] == 0 and r["bad_box"] == 0 and r["bad_empty"] == 0
        assert r["nodes"] <= r2["nodes"] and r["max_pending"] <= r["stack_need"] <= 3 * r2["depth"]
        if len(boxes) > 100:
            # one primitive per leaf: the bottom level's nodes hold 2..4 leaves, 3 on average
            assert r["fill_x1000"] > 2900 and r["nodes"] < 0.55 * r2["nodes"]
    r = chk.wide(sphere_boxes([[1, 2, 3]], [0.5]))  # a lone primitive stays the root leaf: nothing to collapse
    assert r["nodes"] == 0 and r["leaves"] == 1 and r["wrong"] == 0
    # adversarial chain (one primitive peeled per level): the stack bound still holds under the depth cap
    n = 200
    cc = np.stack([2.0 ** np.arange(n) * 1e-20, np.zeros(n), np.zeros(n)], axis=1)
    r = chk.wide(sphere_boxes(cc, 2.0 ** np.arange(n) * 0.4e-20), max_depth=16)
    assert r["leaves"] == n and r["wrong"] == 0 and r["bad_box"] == 0 and r["max_pending"] <= r["stack_need"] <= 48
