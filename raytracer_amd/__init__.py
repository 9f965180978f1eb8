"""raytracer_amd — MI355X-native path-tracing backend behind the reference's Camera::render.

Python here is plumbing only (ctypes over the C ABI of include/rt_abi.h and over the C++ host layer in
raytracer_amd/host/).  The compute path is librt_mi355x.so (hand-written HIP for gfx950); there is NO CPU
fallback: loading fails loudly when the library is missing, and every entry point needs a GPU.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_GPU_PATH = os.environ.get("RT_LIB_GPU") or os.path.join(HERE, "librt_mi355x.so")  # (override: A/B runs of two builds)
LIB_HOST_PATH = os.path.join(HERE, "libraytracer_host.so")

RT_OK = 0
RT_E_INVALID_ARG, RT_E_HIP, RT_E_OOM, RT_E_UNSUPPORTED, RT_E_INTERNAL = -1, -2, -3, -4, -5
RT_NOALLOC_MAX_PIXELS = 1 << 24  # include/rt_abi.h

# every symbol include/rt_abi.h declares
ABI_SYMBOLS = [
    "rt_abi_version", "rt_init", "rt_shutdown", "rt_last_error", "rt_scene_upload", "rt_scene_free", "rt_intersect",
    "rt_render", "rt_render_device", "rt_resolve", "rt_resolve_device", "rt_get_stats", "rt_set_option",
    "rt_philox4x32_10", "rt_half_round_bits", "rt_synchronize", "rt_render_device_noalloc", "rt_render_batch_device",
    "rt_multi_init", "rt_multi_shutdown", "rt_multi_last_error", "rt_multi_device_count", "rt_multi_scene_upload",
    "rt_multi_render", "rt_multi_render_device", "rt_multi_set_option", "rt_multi_get_stats",
]


class RtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rt error {code}: {msg}")
        self.code = code


# ---- POD mirrors of include/rt_abi.h ------------------------------------------------------------------
class RtPrim(C.Structure):
    _fields_ = [("type", C.c_uint32), ("material_id", C.c_uint32), ("object_id", C.c_int32), ("reserved", C.c_uint32),
                ("a", C.c_double * 3), ("b", C.c_double * 3), ("c", C.c_double * 3), ("d", C.c_double * 3),
                ("radius", C.c_double), ("density", C.c_double)]


class RtMaterial(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("tex", C.c_int32), ("albedo", C.c_double * 3), ("fuzz", C.c_double),
                ("ior", C.c_double)]


class RtTexture(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("odd", C.c_int32), ("even", C.c_int32), ("width", C.c_uint32),
                ("height", C.c_uint32), ("reserved", C.c_uint32), ("texel_offset", C.c_uint64),
                ("rgb", C.c_double * 3), ("inv_scale", C.c_double)]


class RtSceneDesc(C.Structure):
    _fields_ = [("prims", C.POINTER(RtPrim)), ("n_prims", C.c_uint64), ("materials", C.POINTER(RtMaterial)),
                ("n_materials", C.c_uint64), ("textures", C.POINTER(RtTexture)), ("n_textures", C.c_uint64),
                ("texels", C.POINTER(C.c_uint8)), ("n_texel_bytes", C.c_uint64)]


class RtCamera(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("sampler_kind", C.c_uint32), ("sampler_n", C.c_uint32),
                ("max_depth", C.c_uint32), ("background_mode", C.c_uint32), ("center", C.c_double * 3),
                ("pixel00", C.c_double * 3), ("du", C.c_double * 3), ("dv", C.c_double * 3),
                ("defocus_u", C.c_double * 3), ("defocus_v", C.c_double * 3), ("defocus_angle", C.c_double),
                ("background", C.c_double * 3)]


class RtStats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("rays", C.c_uint64), ("passes", C.c_uint32), ("bounces_run", C.c_uint32),
                ("ms_total", C.c_float), ("ms_generate", C.c_float), ("ms_intersect", C.c_float),
                ("ms_shade", C.c_float), ("ms_accumulate", C.c_float), ("intersect_launches", C.c_uint64),
                ("probe_rays", C.c_uint64), ("probe_node_visits", C.c_uint64), ("probe_prim_tests", C.c_uint64),
                ("dbg", C.c_float * 16), ("samples_nan", C.c_uint64), ("samples_clamped", C.c_uint64), ("dbg_counts", C.c_uint64 * 12),
                ("lib_alloc_calls", C.c_uint64)]


class RtMultiStats(C.Structure):
    _fields_ = [("n_devices", C.c_uint32), ("reserved", C.c_uint32), ("paths", C.c_uint64), ("rays", C.c_uint64),
                ("ms_render", C.c_float * 16), ("ms_reduce", C.c_float), ("ms_copy", C.c_float)]


RAY_DTYPE = np.dtype([("o", "<f8", 3), ("d", "<f8", 3), ("time", "<f8")])
HIT_DTYPE = np.dtype([("object_id", "<i4"), ("front_face", "<i4"), ("t", "<f8"), ("p", "<f8", 3), ("n", "<f8", 3),
                      ("u", "<f8"), ("v", "<f8")])

_gpu = None
_host = None


def lib():
    """librt_mi355x.so (HIP).  Raises if the extension is not built — there is no fallback."""
    global _gpu
    if _gpu is not None:
        return _gpu
    if not os.path.exists(LIB_GPU_PATH):
        raise ImportError(f"{LIB_GPU_PATH} is missing: build it with `python -m raytracer_amd.build` "
                          "(hipcc --offload-arch=gfx950). raytracer_amd has no CPU fallback.")
    L = C.CDLL(LIB_GPU_PATH, mode=C.RTLD_GLOBAL)
    vp = C.c_void_p
    L.rt_abi_version.restype = C.c_int
    L.rt_init.argtypes = [C.c_int, C.POINTER(vp)]
    L.rt_shutdown.argtypes = [vp]
    L.rt_last_error.argtypes = [vp]
    L.rt_last_error.restype = C.c_char_p
    L.rt_scene_upload.argtypes = [vp, C.POINTER(RtSceneDesc), C.POINTER(vp)]
    L.rt_scene_free.argtypes = [vp, vp]
    L.rt_intersect.argtypes = [vp, vp, vp, C.c_uint64, vp]
    L.rt_render.argtypes = [vp, vp, C.POINTER(RtCamera), C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, vp]
    L.rt_render_device.argtypes = [vp, vp, C.POINTER(RtCamera), C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp]
    L.rt_resolve.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, vp, vp]
    L.rt_resolve_device.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, vp, vp, vp]
    L.rt_get_stats.argtypes = [vp, C.POINTER(RtStats)]
    L.rt_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    L.rt_philox4x32_10.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.rt_philox4x32_10.restype = None
    L.rt_half_round_bits.argtypes = [C.c_float, C.c_int]
    L.rt_half_round_bits.restype = C.c_uint32
    L.rt_synchronize.argtypes = [vp]
    if not hasattr(L, "rt_multi_init") or not hasattr(L, "rt_render_device_noalloc"):
        if not os.environ.get("RT_LIB_GPU"):
            raise ImportError(f"{LIB_GPU_PATH} lacks entry points of include/rt_abi.h: rebuild it")
        _gpu = L  # an older build loaded on purpose (A/B timing): only the entry points it has are bound
        return L
    L.rt_render_device_noalloc.argtypes = L.rt_render_device.argtypes
    if hasattr(L, "rt_render_batch_device"):
        L.rt_render_batch_device.argtypes = [vp, vp, C.POINTER(RtCamera), C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32),
                                             C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p), vp]
    L.rt_multi_init.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    L.rt_multi_shutdown.argtypes = [vp]
    L.rt_multi_last_error.argtypes = [vp]
    L.rt_multi_last_error.restype = C.c_char_p
    L.rt_multi_device_count.argtypes = [vp]
    L.rt_multi_scene_upload.argtypes = [vp, C.POINTER(RtSceneDesc)]
    L.rt_multi_render.argtypes = [vp, C.POINTER(RtCamera), C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, vp]
    L.rt_multi_render_device.argtypes = L.rt_multi_render.argtypes
    L.rt_multi_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    L.rt_multi_get_stats.argtypes = [vp, C.POINTER(RtMultiStats)]
    _gpu = L
    return L


def hostlib():
    """libraytracer_host.so: the C++ mirror of the reference's scene/camera API (flattening, scene builders)."""
    global _host
    if _host is not None:
        return _host
    lib()
    if not os.path.exists(LIB_HOST_PATH):
        raise ImportError(f"{LIB_HOST_PATH} is missing: build it with `python -m raytracer_amd.build`.")
    H = C.CDLL(LIB_HOST_PATH)
    vp = C.c_void_p
    H.rth_last_error.restype = C.c_char_p
    H.rth_scene_create.argtypes = [C.c_char_p, C.c_uint64, C.c_double, C.c_int, C.c_uint64, C.c_uint64, C.c_double, vp,
                                   C.c_uint32, C.c_uint32]
    H.rth_scene_create.restype = vp
    H.rth_scene_free.argtypes = [vp]
    H.rth_scene_desc.argtypes = [vp]
    H.rth_scene_desc.restype = C.POINTER(RtSceneDesc)
    H.rth_scene_camera.argtypes = [vp]
    H.rth_scene_camera.restype = C.POINTER(RtCamera)
    H.rth_scene_spp.argtypes = [vp]
    H.rth_scene_spp.restype = C.c_uint64
    H.rth_scene_render.argtypes = [vp, vp, C.c_uint64, C.c_char_p, vp, vp]
    H.rth_scene_render_partial.argtypes = [vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, vp]
    H.rth_scene_render_progressive.argtypes = [vp, vp, C.c_uint64, C.c_uint64, C.c_char_p, C.c_char_p, vp, vp,
                                               C.POINTER(C.c_uint64)]
    H.rth_write_image.argtypes = [C.c_char_p, vp, C.c_uint64, C.c_uint64]
    H.rth_rng_next_f64_seq.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_int, vp, vp]
    _host = H
    return H


TILE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_double))


def write_image(path, rgb):
    """Writes an (H, W, 3) array of gamma-corrected colours with the host layer's writer: the reference's plain-text
    PPM (camera/mod.rs:216-227), or PNG when `path` ends in ".png".  Needs no GPU."""
    import numpy as np
    a = np.ascontiguousarray(rgb, dtype=np.float64)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError("rgb must have shape (H, W, 3)")
    H = hostlib()
    if H.rth_write_image(str(path).encode(), a.ctypes.data, a.shape[1], a.shape[0]) != 0:
        raise RuntimeError(H.rth_last_error().decode())


def philox4x32_10(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().rt_philox4x32_10(c, k, o)
    return list(o)


class Context:
    """One GPU (rt_ctx)."""

    def __init__(self, device_id=0):
        self.L = lib()
        h = C.c_void_p()
        rc = self.L.rt_init(device_id, C.byref(h))
        if rc != RT_OK:
            raise RtError(rc, self.L.rt_last_error(None).decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.rt_shutdown(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def check(self, rc):
        if rc != RT_OK:
            raise RtError(rc, self.L.rt_last_error(self.h).decode())

    def set_option(self, key, value):
        self.check(self.L.rt_set_option(self.h, key.encode(), int(value)))

    def stats(self):
        s = RtStats()
        self.check(self.L.rt_get_stats(self.h, C.byref(s)))
        d = {f: getattr(s, f) for f, _ in RtStats._fields_}
        d["dbg"] = list(s.dbg)
        d["dbg_counts"] = list(s.dbg_counts)
        return d

    def synchronize(self):
        self.check(self.L.rt_synchronize(self.h))


class DeviceScene:
    """A flattened world resident in HBM (rt_scene)."""

    def __init__(self, ctx, desc):
        self.ctx = ctx
        h = C.c_void_p()
        ctx.check(ctx.L.rt_scene_upload(ctx.h, desc, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None) and self.ctx.h:
            self.ctx.L.rt_scene_free(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        self.close()

    def intersect(self, rays):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        hits = np.zeros(rays.shape[0], dtype=HIT_DTYPE)
        self.ctx.check(self.ctx.L.rt_intersect(self.ctx.h, self.h, rays.ctypes.data, rays.shape[0], hits.ctypes.data))
        return hits

    def render_sums(self, cam, seed=0, sample_begin=0, sample_end=None, sample_stride=1):
        """Un-normalised linear RGB sums (H, W, 3) f32 of the selected sample indices."""
        if sample_end is None:
            sample_end = cam.sampler_n * cam.sampler_n if cam.sampler_kind == 0 else cam.sampler_n
        out = np.zeros((cam.height, cam.width, 3), dtype=np.float32)
        self.ctx.check(self.ctx.L.rt_render(self.ctx.h, self.h, C.byref(cam), seed, sample_begin, sample_end,
                                            sample_stride, out.ctypes.data))
        return out

    def render_device(self, cam, d_out_ptr, seed=0, sample_begin=0, sample_end=None, sample_stride=1, stream=None,
                      no_alloc=False):
        """rt_render_device, or rt_render_device_noalloc (scratch in the library's fixed device array)."""
        if sample_end is None:
            sample_end = cam.sampler_n * cam.sampler_n if cam.sampler_kind == 0 else cam.sampler_n
        fn = self.ctx.L.rt_render_device_noalloc if no_alloc else self.ctx.L.rt_render_device
        self.ctx.check(fn(self.ctx.h, self.h, C.byref(cam), seed, sample_begin, sample_end, sample_stride, d_out_ptr,
                          stream))


def render_batch_device(dev_scene, cam, d_out_ptrs, seeds, sample_begins, sample_end=None, sample_stride=1, stream=None):
    """rt_render_batch_device: len(d_out_ptrs) frames of one scene and camera in one persistent launch."""
    n = len(d_out_ptrs)
    if sample_end is None:
        sample_end = cam.sampler_n * cam.sampler_n if cam.sampler_kind == 0 else cam.sampler_n
    ctx = dev_scene.ctx
    ctx.check(ctx.L.rt_render_batch_device(ctx.h, dev_scene.h, C.byref(cam), n, (C.c_uint64 * n)(*[s & 0xFFFFFFFFFFFFFFFF for s in seeds]),
                                           (C.c_uint32 * n)(*sample_begins), sample_end, sample_stride,
                                           (C.c_void_p * n)(*d_out_ptrs), stream))


class MultiContext:
    """Several GPUs of one node behind one render call (rt_multi): one context + host thread per listed device, sample
    indices split by residue class, partial frames summed on the first device over xGMI."""

    def __init__(self, device_ids):
        self.L = lib()
        ids = (C.c_int * len(device_ids))(*device_ids)
        h = C.c_void_p()
        rc = self.L.rt_multi_init(ids, len(device_ids), C.byref(h))
        if rc != RT_OK:
            raise RtError(rc, self.L.rt_multi_last_error(None).decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.rt_multi_shutdown(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def check(self, rc):
        if rc != RT_OK:
            raise RtError(rc, self.L.rt_multi_last_error(self.h).decode())

    def device_count(self):
        return self.L.rt_multi_device_count(self.h)

    def set_option(self, key, value):
        self.check(self.L.rt_multi_set_option(self.h, key.encode(), int(value)))

    def upload(self, desc):
        self.check(self.L.rt_multi_scene_upload(self.h, desc))

    def render_sums(self, cam, seed=0, sample_begin=0, sample_end=None, sample_stride=1):
        if sample_end is None:
            sample_end = cam.sampler_n * cam.sampler_n if cam.sampler_kind == 0 else cam.sampler_n
        out = np.zeros((cam.height, cam.width, 3), dtype=np.float32)
        self.check(self.L.rt_multi_render(self.h, C.byref(cam), seed, sample_begin, sample_end, sample_stride,
                                          out.ctypes.data))
        return out

    def render_device(self, cam, d_out_ptr, seed=0, sample_begin=0, sample_end=None, sample_stride=1):
        if sample_end is None:
            sample_end = cam.sampler_n * cam.sampler_n if cam.sampler_kind == 0 else cam.sampler_n
        self.check(self.L.rt_multi_render_device(self.h, C.byref(cam), seed, sample_begin, sample_end, sample_stride,
                                                 d_out_ptr))

    def stats(self):
        s = RtMultiStats()
        self.check(self.L.rt_multi_get_stats(self.h, C.byref(s)))
        n = s.n_devices
        return {"n_devices": n, "paths": s.paths, "rays": s.rays, "ms_render": list(s.ms_render)[:n],
                "ms_reduce": s.ms_reduce, "ms_copy": s.ms_copy}


def resolve(ctx, sums, spp, gamma=2.2, want_u8=False):
    sums = np.ascontiguousarray(sums, dtype=np.float32)
    h, w = sums.shape[:2]
    rgb = np.zeros((h, w, 3), dtype=np.float64)
    u8 = np.zeros((h, w, 3), dtype=np.uint8) if want_u8 else None
    ctx.check(ctx.L.rt_resolve(ctx.h, sums.ctypes.data, w, h, spp, gamma, rgb.ctypes.data,
                               u8.ctypes.data if want_u8 else None))
    return (rgb, u8) if want_u8 else rgb


class Scene:
    """A reference scene builder (scene.rs) + CameraBuilder settings, built by the C++ host layer."""

    def __init__(self, name, width, aspect=16.0 / 9.0, sampler="uniform", spp=1, depth=50, defocus_override=-1.0,
                 texels=None):
        self.H = hostlib()
        tex_ptr, tw, th = None, 0, 0
        if texels is not None:
            self._texels = np.ascontiguousarray(texels, dtype=np.uint8)
            assert self._texels.ndim == 3 and self._texels.shape[2] == 4
            th, tw = self._texels.shape[:2]
            tex_ptr = self._texels.ctypes.data
        kind = {"uniform": 0, "random": 1}[sampler]
        self.h = self.H.rth_scene_create(name.encode(), width, aspect, kind, spp, depth, defocus_override, tex_ptr, tw,
                                         th)
        if not self.h:
            raise ValueError(self.H.rth_last_error().decode())
        self.desc = self.H.rth_scene_desc(self.h).contents
        self.camera = self.H.rth_scene_camera(self.h).contents
        self.width, self.height = self.camera.width, self.camera.height
        self.spp = int(self.H.rth_scene_spp(self.h))
        self.name = name

    def close(self):
        if getattr(self, "h", None):
            self.H.rth_scene_free(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def upload(self, ctx):
        return DeviceScene(ctx, C.byref(self.desc))

    def camera_copy(self):
        c = RtCamera()
        C.memmove(C.byref(c), C.byref(self.camera), C.sizeof(RtCamera))
        return c

    def prims(self):
        n = self.desc.n_prims
        return np.ctypeslib.as_array(C.cast(self.desc.prims, C.POINTER(C.c_uint8)), shape=(n * C.sizeof(RtPrim),)) \
            .view(PRIM_DTYPE).copy()

    def render(self, ctx, seed=0x853C49E6748FEA9B, ppm_path=None, on_tile=None):
        """Scene::render -> Camera::render (GPU).  Returns the gamma-corrected frame (H, W, 3) f64 assembled from
        the tile messages; `on_tile(top, left, h, w, rgb)` may return False to hang up (cancel)."""
        frame = np.zeros((self.height, self.width, 3), dtype=np.float64)
        tiles = []

        def cb(_user, top, left, h, w, ptr):
            arr = np.ctypeslib.as_array(ptr, shape=(h * w * 3,)).reshape(h, w, 3)
            frame[top:top + h, left:left + w] = arr
            tiles.append((top, left, h, w))
            if on_tile is not None and on_tile(top, left, h, w, arr) is False:
                return 0
            return 1

        cfn = TILE_CB(cb)
        rc = self.H.rth_scene_render(self.h, ctx.h, seed, ppm_path.encode() if ppm_path else None, cfn, None)
        if rc != 0:
            raise RuntimeError(self.H.rth_last_error().decode())
        return frame, tiles

    def render_progressive(self, ctx, passes, seed=0x853C49E6748FEA9B, checkpoint_path=None, ppm_path=None,
                           on_tile=None):
        """Camera::render_progressive: every tile is re-sent after each of `passes` sample passes (live preview),
        sums are checkpointed to `checkpoint_path` after each pass and a matching checkpoint is resumed.
        Returns (last frame (H, W, 3) f64, passes rendered by this call, tile messages received)."""
        if not getattr(ctx, "h", None):
            raise RuntimeError("GPU backend is required (no CPU fallback)")
        frame = np.zeros((self.height, self.width, 3), dtype=np.float64)
        count = [0]

        def cb(_user, top, left, h, w, ptr):
            arr = np.ctypeslib.as_array(ptr, shape=(h * w * 3,)).reshape(h, w, 3)
            frame[top:top + h, left:left + w] = arr
            count[0] += 1
            if on_tile is not None and on_tile(top, left, h, w, arr) is False:
                return 0
            return 1

        cfn = TILE_CB(cb)
        n = C.c_uint64(0)
        rc = self.H.rth_scene_render_progressive(self.h, ctx.h, seed, passes,
                                                 checkpoint_path.encode() if checkpoint_path else None,
                                                 ppm_path.encode() if ppm_path else None, cfn, None, C.byref(n))
        if rc != 0:
            raise RuntimeError(self.H.rth_last_error().decode())
        return frame, int(n.value), count[0]

    def render_partial(self, ctx, seed, rank, world_size):
        out = np.zeros((self.height, self.width, 3), dtype=np.float32)
        rc = self.H.rth_scene_render_partial(self.h, ctx.h, seed, rank, world_size, out.ctypes.data)
        if rc != 0:
            raise RuntimeError(self.H.rth_last_error().decode())
        return out


PRIM_DTYPE = np.dtype([("type", "<u4"), ("material_id", "<u4"), ("object_id", "<i4"), ("reserved", "<u4"),
                       ("a", "<f8", 3), ("b", "<f8", 3), ("c", "<f8", 3), ("d", "<f8", 3), ("radius", "<f8"),
                       ("density", "<f8")])
assert PRIM_DTYPE.itemsize == C.sizeof(RtPrim)
