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
