"""Image textures on spheres: the checked f32 guess of the texel (rt_device.h: sphere_texel_checked, tables made at scene
upload) against the reference's own f64 formulas (get_sphere_uv, geometry.rs:30-34, + ImageTexture::value,
texture.rs:79-93), which stay in the kernels as the path every failed check takes and which option uv_tables = 0 (read at
upload) makes the only path.  Frames must be BIT-IDENTICAL either way: on the earth scene, on the final scene, and on a
unit sphere seen along each axis — normals exactly on the poles, the seam (longitude +-pi), the axes' texel boundaries
— under images whose sizes are and are not multiples of four."""
import ctypes as C

import numpy as np
import pytest

import raytracer_amd as ra

pytestmark = pytest.mark.gpu

ASPECT = 16.0 / 9.0


@pytest.fixture(scope="module")
def ctx():
    c = ra.Context(0)
    yield c
    c.close()


def earth_texels():
    from golden_data import load_earth_rgba
    return load_earth_rgba()


def both_ways(ctx, upload, render):
    frames = []
    try:
        for tables in (1, 0):
            ctx.set_option("uv_tables", tables)
            dev = upload()
            frames.append(render(dev))
            dev.close()
    finally:
        ctx.set_option("uv_tables", 1)
    return frames


@pytest.mark.parametrize("name,width,spp", [("earth", 640, 64), ("final_scene_small", 240, 16), ("final_scene", 320, 16)])
def test_scene_frames_do_not_depend_on_the_texel_tables(ctx, name, width, spp):
    g = ra.Scene(name, width, ASPECT, "uniform", spp, 50, texels=earth_texels())
    cam = g.camera_copy()
    fast, formula = both_ways(ctx, lambda: g.upload(ctx), lambda dev: dev.render_sums(cam, seed=11))
    assert fast.max() > 0
    assert np.array_equal(fast, formula), float(np.mean(fast != formula))


def textured_sphere(width, height, seed):
    rng = np.random.default_rng(seed)
    texels = rng.integers(0, 256, size=(height, width, 4), dtype=np.uint8)
    texels[..., 3] = 255
    blob = np.ascontiguousarray(texels).reshape(-1)
    prims = (ra.RtPrim * 1)()
    prims[0].type, prims[0].material_id, prims[0].object_id = 0, 0, 0
    prims[0].radius = 1.0
    mats = (ra.RtMaterial * 1)()
    mats[0].kind, mats[0].tex = 0, 0
    texs = (ra.RtTexture * 1)()
    texs[0].kind, texs[0].width, texs[0].height, texs[0].texel_offset = 2, width, height, 0
    desc = ra.RtSceneDesc()
    desc.prims, desc.n_prims = prims, 1
    desc.materials, desc.n_materials = mats, 1
    desc.textures, desc.n_textures = texs, 1
    desc.texels = blob.ctypes.data_as(C.POINTER(C.c_uint8))
    desc.n_texel_bytes = blob.size
    return desc, (prims, mats, texs, blob)


def axis_camera(axis, sign, n_side=65, sampler_n=3):
    """Pinhole camera on a coordinate axis, 4 from the origin, looking at it; the pixel grid is dyadic and centred, so the
    centre sample of the centre pixel travels exactly along the axis and its neighbours straddle the coordinate planes."""
    cam = ra.RtCamera()
    cam.width = cam.height = n_side
    cam.sampler_kind, cam.sampler_n, cam.max_depth, cam.background_mode = 0, sampler_n, 3, 0
    step = 2.0 ** -6
    half = (n_side // 2) * step
    e = [np.zeros(3), np.zeros(3), np.zeros(3)]
    for a in range(3):
        e[a][a] = 1.0
    fwd = -sign * e[axis]
    right, up = e[(axis + 1) % 3], e[(axis + 2) % 3]
    center = -4.0 * fwd
    p00 = center + fwd - half * right + half * up
    for a in range(3):
        cam.center[a] = center[a]
        cam.pixel00[a] = p00[a]
        cam.du[a] = step * right[a]
        cam.dv[a] = -step * up[a]
    return cam


@pytest.mark.parametrize("width,height", [(1024, 512), (36, 18), (37, 19), (2, 3), (1, 1), (5, 2)])
def test_poles_seam_and_axes_take_the_formulas_answer(ctx, width, height):
    desc, keep = textured_sphere(width, height, seed=width * 1000 + height)
    for axis in range(3):
        for sign in (1.0, -1.0):
            cam = axis_camera(axis, sign)
            fast, formula = both_ways(ctx, lambda: ra.DeviceScene(ctx, desc), lambda dev: dev.render_sums(cam, seed=4))
            assert fast.max() > 0
            assert np.array_equal(fast, formula), (axis, sign, float(np.mean(fast != formula)))
