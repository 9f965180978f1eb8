"""Malformed or extreme arguments at the C ABI: every entry point returns a status, never aborts or reads out of
range (include/rt_abi.h preamble).  Cases a C caller produces by accident: a zero-initialised METAL material in a
scene without textures, an image texture whose width * height * 4 wraps 64 bits, a sample stride of 2^32 - 1."""
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


def one_sphere(kind, n_textures=0, tex_fields=None):
    prims = (ra.RtPrim * 2)()
    for i, (y, r) in enumerate(((0.0, 2.0), (-1002.0, 1000.0))):
        prims[i].type, prims[i].material_id, prims[i].object_id = 0, 0, i
        prims[i].a[1] = prims[i].b[1] = y
        prims[i].radius = r
    mats = (ra.RtMaterial * 1)()  # zero-initialised as a C caller's `rt_material m = {0}`: tex == 0
    mats[0].kind = kind
    for a in range(3):
        mats[0].albedo[a] = 0.8
    mats[0].ior = 1.5
    texs = (ra.RtTexture * max(n_textures, 1))()
    for k, v in (tex_fields or {}).items():
        setattr(texs[0], k, v)
    desc = ra.RtSceneDesc()
    desc.prims, desc.n_prims = prims, 2
    desc.materials, desc.n_materials = mats, 1
    desc.textures, desc.n_textures = (texs if n_textures else None), n_textures
    desc.texels, desc.n_texel_bytes = None, 0
    return desc, (prims, mats, texs)


def camera():
    return ra.Scene("earth", 64, 16 / 9, "uniform", 16, 8).camera_copy()  # looks at the origin from (12, 0, 0)


@pytest.mark.parametrize("kind", [1, 2])  # METAL, DIELECTRIC: kinds that take no texture
def test_zeroed_textureless_material_in_a_scene_without_textures(ctx, kind):
    desc, keep = one_sphere(kind)
    dev = ra.DeviceScene(ctx, desc)
    sums = dev.render_sums(camera(), seed=5)
    assert np.isfinite(sums).all() and sums.min() >= 0 and sums.max() > 0
    dev.close()


def test_lambertian_with_missing_texture_is_rejected(ctx):
    desc, keep = one_sphere(0)
    with pytest.raises(ra.RtError) as e:
        ra.DeviceScene(ctx, desc)
    assert e.value.code == -1  # RT_E_INVALID_ARG


@pytest.mark.parametrize("w,h,off", [(0x80000000, 0x80000000, 0),  # width * height * 4 == 2^64: wraps to 0
                                     (1, 1, 0xFFFFFFFFFFFFFFFC),  # offset + bytes wraps
                                     (4, 4, 0)])                   # plainly larger than the (empty) blob
def test_image_texture_range_is_checked_without_wrapping(ctx, w, h, off):
    desc, keep = one_sphere(0, n_textures=1, tex_fields=dict(kind=2, width=w, height=h, texel_offset=off))
    with pytest.raises(ra.RtError) as e:
        ra.DeviceScene(ctx, desc)
    assert e.value.code == -1 and "texel range" in str(e.value)


def test_huge_sample_stride_selects_one_sample(ctx):
    g = ra.Scene("two_spheres", 64, 16 / 9, "uniform", 16, 8)
    dev = ra.DeviceScene(ctx, C.byref(g.desc))
    cam = g.camera_copy()
    one = dev.render_sums(cam, seed=2, sample_begin=3, sample_end=4, sample_stride=1)
    assert one.max() > 0
    for stride in (0xFFFFFFFF, 0x80000000, 13):  # K = ceil((16 - 3) / stride) = 1, computed in 64 bits
        assert np.array_equal(dev.render_sums(cam, seed=2, sample_begin=3, sample_end=16, sample_stride=stride), one)
    dev.close()


def test_frame_cut_short_by_the_watchdog_is_an_error_not_a_frame(ctx):
    """the scheduler's watchdog (a wave that finds nothing to do for ~2^24 polls gives up instead of hanging
    the GPU) used to be visible in rt_get_stats only.  With the poll limit shrunk to one poll the flag is raised on any
    frame whose waves ever wait for a tile to be opened: the blocking entry points must then return RT_E_INTERNAL, and
    the asynchronous one must report it through rt_get_stats; the next frame with the normal limit is fine again."""
    import torch
    g = ra.Scene("book_cover", 320, 16 / 9, "uniform", 16, 50)
    dev = ra.DeviceScene(ctx, C.byref(g.desc))
    cam = g.camera_copy()
    good = dev.render_sums(cam, seed=3)
    ctx.set_option("dbg_watchdog_polls", 1)
    try:
        with pytest.raises(ra.RtError) as e:
            dev.render_sums(cam, seed=3)
        assert e.value.code == ra.RT_E_INTERNAL and "watchdog" in str(e.value)
        d = torch.zeros((g.height, g.width, 3), dtype=torch.float32, device="cuda")
        dev.render_device(cam, d.data_ptr(), seed=3)  # asynchronous: the call itself cannot know
        ctx.synchronize()
        with pytest.raises(ra.RtError) as e:
            ctx.stats()
        assert e.value.code == ra.RT_E_INTERNAL
        m = ra.MultiContext([0, 0])
        try:
            m.set_option("dbg_watchdog_polls", 1)
            m.upload(C.byref(g.desc))
            with pytest.raises(ra.RtError) as e:
                m.render_sums(cam, seed=3)
            assert e.value.code == ra.RT_E_INTERNAL and "watchdog" in str(e.value)
            with pytest.raises(ra.RtError):
                m.stats()  # rt_multi_get_stats propagates the failing rt_get_stats
        finally:
            m.close()
    finally:
        ctx.set_option("dbg_watchdog_polls", 0)
    again = dev.render_sums(cam, seed=3)
    assert np.array_equal(again, good)
    assert ctx.stats()["rays"] > 0
    dev.close()


def test_no_sample_is_dropped_or_clamped_on_ordinary_frames(ctx):
    """rt_stats.samples_nan / samples_clamped count the radiance channels that did not reach the frame as computed (NaN
    -> 0, > 2^20 -> 2^20).  Zero on the shipped scenes, lights included; non-zero when a light of 4e6 is looked at."""
    for name, spp in (("book_cover", 16), ("cornell_smoke", 16)):
        g = ra.Scene(name, 160, 16 / 9, "uniform", spp, 50)
        dev = ra.DeviceScene(ctx, C.byref(g.desc))
        sums = dev.render_sums(g.camera_copy(), seed=9)
        st = ctx.stats()
        assert np.isfinite(sums).all()
        assert st["samples_nan"] == 0 and st["samples_clamped"] == 0, (name, st["samples_nan"], st["samples_clamped"])
        dev.close()
    # a sphere that emits 4e6: every path that sees it directly is clamped to 2^20 per channel, and counted
    desc, keep = one_sphere(3, n_textures=1, tex_fields=dict(kind=0))
    for a in range(3):
        keep[2][0].rgb[a] = 4.0e6
    keep[1][0].tex = 0
    dev = ra.DeviceScene(ctx, desc)
    cam = camera()
    cam.background_mode = 1
    sums = dev.render_sums(cam, seed=1)
    st = ctx.stats()
    assert st["samples_clamped"] > 0 and st["samples_nan"] == 0
    assert sums.max() <= 16 * 1048576.0 * (1 + 1e-6)  # 16 spp, each clamped to 2^20
    # sums of K clamped samples cannot overflow the fixed-point slot, whatever K: 4096 spp of the same view
    cam.sampler_n = 64
    big = dev.render_sums(cam, seed=1)
    assert np.isfinite(big).all() and big.max() <= 4096 * 1048576.0 * (1 + 1e-6) and big.max() > 1e9
    dev.close()
