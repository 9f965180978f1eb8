"""CPU-only tests: the C-ABI library loads and exports every declared symbol, the host layer (C++ mirror of the
reference API) flattens the reference scenes exactly like the oracle builds them, error behaviour, Philox KAT.
No compute entry point is called here (no GPU in this container)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import raytracer_amd as ra
from oracle_lib import OracleScene, load as load_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_header_symbols_all_exported():
    hdr = open(os.path.join(ROOT, "include", "rt_abi.h")).read()
    declared = set(re.findall(r"^\s*(?:int|void|uint32_t|const char\*)\s+(rt_[a-z0-9_]+)\s*\(", hdr, flags=re.M))
    assert declared == set(ra.ABI_SYMBOLS), declared ^ set(ra.ABI_SYMBOLS)
    lib = ra.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.rt_abi_version() == 2


def test_abi_struct_sizes_match_header_layout():
    # natural C layout of the POD structs in rt_abi.h
    assert C.sizeof(ra.RtPrim) == 16 + 12 * 8 + 16 == 128
    assert C.sizeof(ra.RtMaterial) == 8 + 5 * 8
    assert C.sizeof(ra.RtTexture) == 24 + 8 + 4 * 8
    assert C.sizeof(ra.RtCamera) == 24 + 18 * 8 + 8 + 24
    assert ra.RAY_DTYPE.itemsize == 56 and ra.HIT_DTYPE.itemsize == 8 + 8 + 24 + 24 + 16


def test_rt_init_without_gpu_fails_cleanly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ra.RtError) as e:
        ra.Context(0)
    assert e.value.code == ra.RT_E_HIP and "no HIP device" in str(e.value)
    # NULL handling never crashes
    lib = ra.lib()
    assert lib.rt_shutdown(None) == 0
    assert lib.rt_get_stats(None, None) == ra.RT_E_INVALID_ARG
    assert lib.rt_set_option(None, b"mode", 0) == ra.RT_E_INVALID_ARG


def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10."""
    assert ra.philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert ra.philox4x32_10([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert ra.philox4x32_10([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_host_rng_matches_reference_vectors():
    H = ra.hostlib()
    out = (C.c_double * 3)()
    st = (C.c_uint64 * 2)()
    H.rth_rng_next_f64_seq(1, 1, 2, 3, out, st)  # pixel (0,0) stream: SURVEY §8c
    assert [st[0], st[1]] == [0xC2DA8E3B2E1E5948, 0xAA11E4D611D754CC]
    assert list(out) == [0.42548293276919114, 0.19595313318494761, 0.7615421707530103]
    H.rth_rng_next_f64_seq(42, 1337, 0, 3, out, st)
    assert list(out) == [7.475573979287553e-17, 9.678347009227015e-06, 0.5183144114499367]


def test_book_cover_flattening_matches_reference_census():
    s = ra.Scene("book_cover", 400, 16 / 9, "random", 10, 50)
    p = s.prims()
    assert len(p) == 484 and (p["object_id"] == np.arange(484)).all()
    assert (p["type"][1:481] == 1).all() and (p["type"][[0, 481, 482, 483]] == 0).all()
    mats = np.ctypeslib.as_array(C.cast(s.desc.materials, C.POINTER(C.c_uint8)),
                                 shape=(s.desc.n_materials * C.sizeof(ra.RtMaterial),))
    kinds = np.frombuffer(mats.tobytes(), dtype=np.dtype([("kind", "<u4"), ("tex", "<i4"), ("albedo", "<f8", 3),
                                                          ("fuzz", "<f8"), ("ior", "<f8")]))
    small = kinds[p["material_id"][1:481]]["kind"]
    assert [(small == k).sum() for k in (0, 1, 2)] == [317, 105, 58]  # diffuse / metal / glass
    assert tuple(p["a"][1]) == (-10.999991289487692, 0.2, -10.533517029695057)
    assert tuple(p["a"][2]) == (-10.384960083338036, 0.2, -9.736323990780543)
    # destination = center + (0, 0.5*(1-choose_mat), 0)  with choose_mat = 0.8651562194246568 (scene.rs:137-140)
    assert p["b"][2][1] == 0.2 + 0.5 * (1.0 - 0.8651562194246568)
    assert p["radius"][0] == 1000.0 and (p["radius"][1:481] == 0.2).all()
    # ground: Lambertian over a checker of two solids (odd = green first: scene.rs:102-106)
    tex = np.frombuffer(np.ctypeslib.as_array(C.cast(s.desc.textures, C.POINTER(C.c_uint8)),
                                              shape=(s.desc.n_textures * C.sizeof(ra.RtTexture),)).tobytes(),
                        dtype=np.dtype([("kind", "<u4"), ("odd", "<i4"), ("even", "<i4"), ("w", "<u4"), ("h", "<u4"),
                                        ("r", "<u4"), ("off", "<u8"), ("rgb", "<f8", 3), ("inv_scale", "<f8")]))
    g = kinds[p["material_id"][0]]
    chk = tex[g["tex"]]
    assert g["kind"] == 0 and chk["kind"] == 1 and chk["inv_scale"] == 1.0 / 0.32
    assert tuple(tex[chk["odd"]]["rgb"]) == (0.2, 0.3, 0.1) and tuple(tex[chk["even"]]["rgb"]) == (0.9, 0.9, 0.9)


@pytest.mark.parametrize("name", ["composition", "book_cover", "two_spheres", "something_blocky"])
def test_host_camera_matches_oracle_camera(name):
    o = OracleScene(name, 400, 16 / 9, "uniform", 16, 50)
    g = ra.Scene(name, 400, 16 / 9, "uniform", 16, 50)
    cam = g.camera
    host = np.concatenate([np.array(getattr(cam, f)[:]) for f in ("center", "pixel00", "du", "dv", "defocus_u",
                                                                  "defocus_v")] + [[cam.defocus_angle]])
    assert np.array_equal(host, o.camera_block())  # bit-identical frame vectors
    assert (cam.width, cam.height, cam.sampler_kind, cam.sampler_n, cam.max_depth) == (400, 225, 0, 4, 50)
    assert g.desc.n_prims == o.n_objects


def test_reference_panics_become_errors():
    with pytest.raises(ValueError, match="square number"):
        ra.Scene("book_cover", 64, 16 / 9, "uniform", 10, 50)  # builder.rs:63-67
    with pytest.raises(ValueError, match="unknown scene"):
        ra.Scene("cornell_box_of_doom", 64)
    # random sampler accepts any count (builder.rs:70)
    assert ra.Scene("book_cover", 64, 16 / 9, "random", 10, 50).spp == 10


def test_main_rs_defocus_is_overridden_by_scene_builders():
    # main.rs:41 sets defocus 0.2 on the builder; book_cover overrides to 0.3, two_spheres keeps 0.2,
    # composition forces 0 (scene.rs:41,95)
    assert ra.Scene("book_cover", 64, 16 / 9, "uniform", 4, 8, defocus_override=0.2).camera.defocus_angle == 0.3
    assert ra.Scene("two_spheres", 64, 16 / 9, "uniform", 4, 8, defocus_override=0.2).camera.defocus_angle == 0.2
    assert ra.Scene("composition", 64, 16 / 9, "uniform", 4, 8, defocus_override=0.2).camera.defocus_angle == 0.0


def test_earth_scene_carries_texels():
    from golden_data import load_earth_rgba, earth_sha256, load_json
    assert earth_sha256() == load_json("earthmap.json")["rgba8_sha256"]
    tex = load_earth_rgba()
    s = ra.Scene("earth", 320, 16 / 9, "uniform", 4, 50, texels=tex)
    assert s.desc.n_prims == 1 and s.desc.n_texel_bytes == 1024 * 512 * 4
    blob = np.ctypeslib.as_array(s.desc.texels, shape=(s.desc.n_texel_bytes,))
    assert np.array_equal(blob.reshape(512, 1024, 4), tex)


def test_render_without_gpu_context_is_an_error_not_a_fallback():
    s = ra.Scene("two_spheres", 32, 16 / 9, "uniform", 1, 4)

    class NullCtx:
        h = None
    with pytest.raises(RuntimeError, match="GPU backend is required"):
        s.render(NullCtx())


def test_oracle_is_not_linked_into_the_product():
    import subprocess
    for lib in ("librt_mi355x.so", "libraytracer_host.so"):
        out = subprocess.run(["ldd", os.path.join(ROOT, "raytracer_amd", lib)], capture_output=True, text=True).stdout
        assert "oracle" not in out
    for root, _, files in os.walk(os.path.join(ROOT, "raytracer_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip")):
                src = open(os.path.join(root, f)).read()
                assert "oracle_lib" not in src and "liboracle" not in src and "oracle/" not in src.replace(
                    "under oracle/", ""), f


@pytest.mark.parametrize("name", ["cornell_box", "cornell_smoke", "final_scene_small"])
def test_ext_scenes_host_matches_oracle_structure(name):
    """EXT scenes (SURVEY aX1-aX5): same object count / camera on both sides; instances are baked on the host."""
    from golden_data import load_earth_rgba
    tex = load_earth_rgba() if name.startswith("final") else None
    o = OracleScene(name, 100, 1.0, "uniform", 4, 20, texels=tex)
    g = ra.Scene(name, 100, 1.0, "uniform", 4, 20, texels=tex)
    assert g.desc.n_prims == o.n_objects
    p = g.prims()
    assert (p["object_id"] == np.arange(len(p))).all()
    assert g.camera.background_mode == 1 and tuple(g.camera.background) == (0.0, 0.0, 0.0)
    assert (g.width, g.height) == (100, 100)
    if name == "cornell_box":
        assert (p["type"] == 2).all() and len(p) == 18
        # box1: 165x330x165 rotated 15 deg about Y then translated (265,0,295): its bottom face keeps y = 0 and
        # every edge keeps its length
        quads = p[6:12]
        assert np.allclose(np.linalg.norm(quads["b"], axis=1), [165, 165, 165, 165, 165, 165])
        assert np.allclose(sorted(np.linalg.norm(quads["c"], axis=1)), [165, 165, 330, 330, 330, 330])
        assert np.isclose(quads["a"][:, 1].min(), 0.0) and np.isclose(quads["a"][:, 1].max(), 330.0)
        c, s_ = np.cos(np.radians(15)), np.sin(np.radians(15))
        # front face Q = (0,0,165) -> (cos*0 + sin*165 + 265, 0, -sin*0 + cos*165 + 295)
        assert np.allclose(quads["a"][0], [s_ * 165 + 265, 0, c * 165 + 295])
    if name == "cornell_smoke":
        assert list(p["type"]) == [2] * 6 + [4, 4] and list(p["density"][6:]) == [0.01, 0.01]
    if name == "final_scene_small":
        assert (p["type"] == 3).sum() == 2 and (p["type"] == 1).sum() == 1


def test_cpp_example_is_built_and_fails_cleanly_without_a_gpu():
    """examples/render_scene.cpp = the reference's main() flow (main.rs:28-67) on the C++ host layer."""
    import subprocess
    import torch
    exe = os.path.join(ROOT, "examples", "render_scene")
    assert os.path.exists(exe), "build it with python -m raytracer_amd.build"
    if torch.cuda.is_available():
        pytest.skip("GPU present (the GPU run is tests/test_gpu_example.py)")
    r = subprocess.run([exe, "two_spheres", "64", "4", "4", "/dev/null"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "no HIP device" in r.stderr


def test_image_writers_ppm_and_png(tmp_path):
    """Output path: the reference's P3 text PPM (camera/mod.rs:216-227, color.rs:68-81) and the PNG
    variant carry the same saturating 8-bit quantisation; the PNG is read back with an independent decoder."""
    import zlib
    import struct
    rng = np.random.default_rng(5)
    img = rng.uniform(-0.2, 1.3, (37, 301, 3))  # odd sizes; values outside [0, 1] saturate; > 65535 B of scanlines
    ra.write_image(tmp_path / "a.ppm", img)
    ra.write_image(tmp_path / "a.png", img)
    tok = (tmp_path / "a.ppm").read_text().split()
    assert tok[:4] == ["P3", "301", "37", "255"]
    ppm = np.array(tok[4:], dtype=np.int64).reshape(37, 301, 3)
    want = np.clip(img * 256.0, 0, 255).astype(np.int64)  # `as u8` of a float saturates and truncates
    assert np.array_equal(ppm, want)
    raw = (tmp_path / "a.png").read_bytes()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(raw):
        n, typ = struct.unpack(">I4s", raw[pos:pos + 8])
        body = raw[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(typ + body)
        chunks.append((typ, body))
        pos += 12 + n
    assert [c[0] for c in chunks] == [b"IHDR", b"IDAT", b"IEND"]
    assert struct.unpack(">IIBBBBB", chunks[0][1]) == (301, 37, 8, 2, 0, 0, 0)
    lines = np.frombuffer(zlib.decompress(chunks[1][1]), dtype=np.uint8).reshape(37, 1 + 3 * 301)  # checks Adler-32 too
    assert (lines[:, 0] == 0).all() and np.array_equal(lines[:, 1:].reshape(37, 301, 3), want)
    try:
        from PIL import Image
    except ImportError:
        return
    assert np.array_equal(np.asarray(Image.open(tmp_path / "a.png").convert("RGB")), want)
    with pytest.raises(RuntimeError):
        ra.write_image(tmp_path / "no_such_dir" / "a.png", img)


def test_binary16_plane_rounding_is_outward_and_tight():
    """The four-wide tree's 64-B nodes keep their planes in binary16 (rt_lib.hip half_round): lower planes round
    towards -inf, upper planes towards +inf, by at most one binary16 step, so an encoded box can only grow."""
    L = ra.lib()
    rng = np.random.default_rng(9)
    xs = np.concatenate([rng.normal(0, 1, 4000) * 10.0 ** rng.uniform(-9, 5.2, 4000), [0.0, -0.0, 1.0, -1.0, 65504.0,
                         -65504.0, 65505.0, 70000.0, -1e30, 1e30, 5.96e-8, -5.96e-8, 1e-12, -1e-12, 6.1e-5, 0.1, -0.1,
                         2049.0, -2049.0, np.inf, -np.inf]]).astype(np.float32)
    for up in (0, 1):
        bits = np.array([L.rt_half_round_bits(float(x), up) for x in xs], dtype=np.uint16)
        h = bits.view(np.float16).astype(np.float64)
        x64 = xs.astype(np.float64)
        assert (h >= x64).all() if up else (h <= x64).all()
        # tight: the neighbouring binary16 value on the other side is already past x
        with np.errstate(over="ignore"):  # stepping from 65504 towards +inf overflows to inf, as intended
            nb = np.nextafter(bits.view(np.float16), np.float16(-np.inf if up else np.inf)).astype(np.float64)
        exact = h == x64
        assert (exact | ((nb < x64) if up else (nb > x64))).all()
    assert L.rt_half_round_bits(1.0, 0) == 0x3C00 and L.rt_half_round_bits(1.0, 1) == 0x3C00
    assert L.rt_half_round_bits(70000.0, 0) == 0x7BFF and L.rt_half_round_bits(70000.0, 1) == 0x7C00  # 65504 / +inf
    assert L.rt_half_round_bits(-70000.0, 1) == 0xFBFF and L.rt_half_round_bits(-70000.0, 0) == 0xFC00
