"""Scenes far beyond the LDS-resident sizes, described by hand through the C ABI (rt_scene_desc): tens of thousands
of random spheres.  Covers the host builder and the HBM-resident instances at scale — the four-wide tree with only
its top levels in LDS (20 000 spheres) and the f32 fallback when references do not fit the 16-bit stack (40 000) —
against a brute-force closest hit in numpy and against each other."""
import ctypes as C

import numpy as np
import pytest

import raytracer_amd as ra

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = ra.Context(0)
    yield c
    c.close()


def sphere_cloud(n, seed):
    rng = np.random.default_rng(seed)
    centers = rng.uniform(-60.0, 60.0, (n, 3))
    radii = rng.uniform(0.2, 1.2, n)
    prims = (ra.RtPrim * n)()
    for i in range(n):
        prims[i].type = 0
        prims[i].material_id = i % 3
        prims[i].object_id = i
        for a in range(3):
            prims[i].a[a] = centers[i, a]
            prims[i].b[a] = centers[i, a]
        prims[i].radius = radii[i]
    mats = (ra.RtMaterial * 3)()
    for k, (kind, rgb, fuzz, ior) in enumerate([(0, (0.7, 0.3, 0.2), 0.0, 0.0), (1, (0.8, 0.8, 0.9), 0.1, 0.0),
                                                (2, (1.0, 1.0, 1.0), 0.0, 1.5)]):
        mats[k].kind = kind
        mats[k].tex = 0 if kind == 0 else -1
        for a in range(3):
            mats[k].albedo[a] = rgb[a]
        mats[k].fuzz = fuzz
        mats[k].ior = ior
    texs = (ra.RtTexture * 1)()
    texs[0].kind = 0
    for a, v in enumerate((0.7, 0.3, 0.2)):
        texs[0].rgb[a] = v
    desc = ra.RtSceneDesc()
    desc.prims, desc.n_prims = prims, n
    desc.materials, desc.n_materials = mats, 3
    desc.textures, desc.n_textures = texs, 1
    desc.texels, desc.n_texel_bytes = None, 0
    return desc, (prims, mats, texs), centers, radii


def brute_force(rays, centers, radii, tmin=1e-6):
    """closest sphere per ray, the reference's quadratic (geometry.rs:36-57) in f64"""
    best_t = np.full(len(rays), np.inf)
    best_i = np.full(len(rays), -1)
    for r in range(len(rays)):
        o, d = rays["o"][r], rays["d"][r]
        oc = o - centers
        a = d @ d
        hb = oc @ d
        c = np.einsum("ij,ij->i", oc, oc) - radii * radii
        disc = hb * hb - a * c
        ok = disc >= 0
        sq = np.sqrt(np.where(ok, disc, 0.0))
        t1, t2 = (-hb - sq) / a, (-hb + sq) / a
        t = np.where(ok & (t1 > tmin), t1, np.where(ok & (t2 > tmin), t2, np.inf))
        i = int(np.argmin(t))
        if np.isfinite(t[i]):
            best_t[r], best_i[r] = t[i], i
    return best_i, best_t


@pytest.mark.parametrize("n", [20000, 40000])
def test_large_sphere_cloud(ctx, n):
    desc, keep, centers, radii = sphere_cloud(n, seed=n)
    dev = ra.DeviceScene(ctx, desc)
    rng = np.random.default_rng(1)
    rays = np.zeros(1500, dtype=ra.RAY_DTYPE)
    rays["o"] = rng.uniform(-70, 70, (len(rays), 3))
    rays["d"] = rng.normal(size=(len(rays), 3))
    want_i, want_t = brute_force(rays, centers, radii)
    h = dev.intersect(rays)
    assert (h["object_id"] == want_i).mean() > 0.999  # grazing / tie rays aside
    same = (h["object_id"] == want_i) & (want_i >= 0)
    assert same.sum() > 100 and np.allclose(h["t"][same], want_t[same], rtol=1e-9)
    for tree in (1, 2):  # the four-wide forms return the binary tree's hits exactly (when the probe's stack allows)
        ctx.set_option("probe_tree", tree)
        try:
            h2 = dev.intersect(rays)
            assert np.array_equal(h2["object_id"], h["object_id"]) and np.array_equal(h2["t"], h["t"])
        except ra.RtError as e:
            assert "deeper stack" in str(e)
        finally:
            ctx.set_option("probe_tree", 0)
    # a small frame through the render kernels: the tree forms trace the same paths
    host = ra.Scene("two_spheres", 160, 16 / 9, "uniform", 16, 12)  # only for a camera block
    cam = host.camera_copy()
    for a, v in enumerate((0.0, 0.0, 150.0)):
        cam.center[a] = v
    look = np.array([0.0, 0.0, -1.0])
    right, up = np.array([1.0, 0.0, 0.0]), np.array([0.0, 1.0, 0.0])
    px = 2.0 * np.tan(np.radians(25.0)) / cam.width
    for a in range(3):
        cam.du[a] = right[a] * px
        cam.dv[a] = -up[a] * px
        cam.pixel00[a] = cam.center[a] + look[a] - 0.5 * px * (cam.width - 1) * right[a] + 0.5 * px * (cam.height - 1) * up[a]
    cam.defocus_angle = 0.0
    a0 = dev.render_sums(cam, seed=3).astype(np.float64)
    assert np.isfinite(a0).all() and a0.min() >= 0 and np.array_equal(a0, dev.render_sums(cam, seed=3))
    frames = []
    try:
        ctx.set_option("fused_half", 0)
        frames.append(dev.render_sums(cam, seed=3).astype(np.float64))
        ctx.set_option("fused_wide", 0)
        frames.append(dev.render_sums(cam, seed=3).astype(np.float64))
    finally:
        ctx.set_option("fused_half", 1); ctx.set_option("fused_wide", 1)
    for f in frames:
        assert np.isclose(a0, f, rtol=1e-4, atol=1e-3).all(axis=2).mean() > 0.97
        assert abs(f.sum() - a0.sum()) < 5e-3 * a0.sum()
    dev.close()
