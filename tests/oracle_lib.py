"""ctypes binding of the CPU oracle (oracle/liboracle*.so) — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the product
package `raytracer_amd` never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

COUNTER_FIELDS = [
    "samples", "box_tests", "sphere_tests", "sphere_disc_pos", "moving_tests", "quad_tests", "medium_tests",
    "bounce_hits", "misses", "depth_cut", "scat_lambert", "scat_metal", "scat_dielectric", "scat_isotropic",
    "emit_light", "tex_solid", "tex_checker", "tex_image", "tex_noise", "rays",
]


class OrcRay(C.Structure):
    _fields_ = [("o", C.c_double * 3), ("d", C.c_double * 3), ("time", C.c_double)]


class OrcHit(C.Structure):
    _fields_ = [("object_id", C.c_int32), ("front_face", C.c_int32), ("t", C.c_double), ("p", C.c_double * 3),
                ("n", C.c_double * 3), ("u", C.c_double), ("v", C.c_double)]


RAY_DTYPE = np.dtype([("o", "<f8", 3), ("d", "<f8", 3), ("time", "<f8")])
HIT_DTYPE = np.dtype([("object_id", "<i4"), ("front_face", "<i4"), ("t", "<f8"), ("p", "<f8", 3), ("n", "<f8", 3),
                      ("u", "<f8"), ("v", "<f8")])
assert RAY_DTYPE.itemsize == C.sizeof(OrcRay) and HIT_DTYPE.itemsize == C.sizeof(OrcHit)


def build_oracle():
    """Compile oracle/ (gcc) if the libraries are missing or stale."""
    src = os.path.join(ORACLE_DIR, "oracle.cpp")
    out = os.path.join(ORACLE_DIR, "liboracle.so")
    if not os.path.exists(out) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(out)):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-j2"], stdout=subprocess.DEVNULL)


_libs = {}


def load(counting=False):
    key = "count" if counting else "plain"
    if key in _libs:
        return _libs[key]
    build_oracle()
    lib = C.CDLL(os.path.join(ORACLE_DIR, "liboracle_count.so" if counting else "liboracle.so"))
    u64p = C.POINTER(C.c_uint64)
    f64p = C.POINTER(C.c_double)
    lib.orc_rng_new.argtypes = [u64p]
    lib.orc_rng_next_u64.argtypes = [u64p]
    lib.orc_rng_next_u64.restype = C.c_uint64
    lib.orc_rng_next_f64.argtypes = [u64p]
    lib.orc_rng_next_f64.restype = C.c_double
    lib.orc_rng_short_jump.argtypes = [u64p]
    lib.orc_rng_long_jump.argtypes = [u64p]
    lib.orc_noise_at.argtypes = [C.c_int32] * 3
    lib.orc_noise_at.restype = C.c_double
    lib.orc_noise_value.argtypes = [C.c_double] * 4 + [f64p]
    lib.orc_pixel_stream_state.argtypes = [C.c_uint64] * 3 + [u64p]
    lib.orc_book_cover_census.argtypes = [C.POINTER(C.c_int32), f64p]
    lib.orc_scene_create.argtypes = [C.c_char_p, C.c_uint64, C.c_double, C.c_int, C.c_uint64, C.c_uint64, C.c_double,
                                     C.c_void_p, C.c_uint32, C.c_uint32]
    lib.orc_scene_create.restype = C.c_void_p
    lib.orc_scene_ok.argtypes = [C.c_void_p]
    lib.orc_scene_error.argtypes = [C.c_void_p]
    lib.orc_scene_error.restype = C.c_char_p
    lib.orc_scene_free.argtypes = [C.c_void_p]
    lib.orc_scene_dims.argtypes = [C.c_void_p, u64p]
    lib.orc_scene_n_objects.argtypes = [C.c_void_p]
    lib.orc_scene_set_defocus_angle.argtypes = [C.c_void_p, C.c_double]
    lib.orc_scene_camera.argtypes = [C.c_void_p, f64p]
    lib.orc_render.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_uint64] * 4
    lib.orc_render.restype = C.c_double
    lib.orc_primary_rays.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_double, C.c_void_p]
    lib.orc_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    lib.orc_resolve_u8.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    lib.orc_write_ppm.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_uint64]
    _libs[key] = lib
    return lib


class OracleScene:
    """A reference scene (scene.rs builder name) + camera, rendered by the CPU oracle."""

    def __init__(self, name, width, aspect=16.0 / 9.0, sampler="uniform", spp=1, depth=50, defocus_override=-1.0,
                 texels=None, counting=False):
        self.lib = load(counting)
        tex_ptr, tw, th = None, 0, 0
        if texels is not None:
            texels = np.ascontiguousarray(texels, dtype=np.uint8)
            assert texels.ndim == 3 and texels.shape[2] == 4
            th, tw = texels.shape[:2]
            tex_ptr = texels.ctypes.data
        kind = {"uniform": 0, "random": 1}[sampler]
        self.h = self.lib.orc_scene_create(name.encode(), width, aspect, kind, spp, depth, defocus_override, tex_ptr,
                                           tw, th)
        if not self.lib.orc_scene_ok(self.h):
            msg = self.lib.orc_scene_error(self.h).decode()
            self.lib.orc_scene_free(self.h)
            self.h = None
            raise ValueError(msg)
        wh = (C.c_uint64 * 2)()
        self.lib.orc_scene_dims(self.h, wh)
        self.width, self.height = int(wh[0]), int(wh[1])
        self.n_objects = self.lib.orc_scene_n_objects(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.orc_scene_free(self.h)
            self.h = None

    def camera_block(self):
        out = (C.c_double * 19)()
        self.lib.orc_scene_camera(self.h, out)
        return np.array(out[:], dtype=np.float64)

    def set_defocus_angle(self, a):
        self.lib.orc_scene_set_defocus_angle(self.h, a)

    def render(self, threads=0, window=None, want_counters=False):
        """Returns dict(gamma=HxWx3 f64, linear=HxWx3 f64, seconds=float[, counters=dict])."""
        g = np.zeros((self.height, self.width, 3), dtype=np.float64)
        l = np.zeros((self.height, self.width, 3), dtype=np.float64)
        cnt = (C.c_uint64 * len(COUNTER_FIELDS))()
        x0, y0, w, h = window if window else (0, 0, 0, 0)
        secs = self.lib.orc_render(self.h, threads, g.ctypes.data, l.ctypes.data, C.addressof(cnt), x0, y0, w, h)
        res = dict(gamma=g, linear=l, seconds=secs)
        if want_counters:
            res["counters"] = dict(zip(COUNTER_FIELDS, [int(v) for v in cnt]))
        return res

    def primary_rays(self, n_side=1, yi=0, xi=0, time=0.5):
        rays = np.zeros(self.width * self.height, dtype=RAY_DTYPE)
        self.lib.orc_primary_rays(self.h, n_side, yi, xi, time, rays.ctypes.data)
        return rays

    def intersect(self, rays):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        hits = np.zeros(rays.shape[0], dtype=HIT_DTYPE)
        self.lib.orc_intersect(self.h, rays.ctypes.data, rays.shape[0], hits.ctypes.data)
        return hits


def resolve_u8(gamma):
    lib = load()
    g = np.ascontiguousarray(gamma, dtype=np.float64)
    out = np.zeros(g.shape, dtype=np.uint8)
    lib.orc_resolve_u8(g.ctypes.data, g.size, out.ctypes.data)
    return out
