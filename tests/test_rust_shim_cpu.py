"""The Rust shim (integration/rust/, shipped uncompiled: no Rust toolchain in the build image) against the C header:
every entry point include/rt_abi.h declares is bound in ffi.rs under the same name with the same number of parameters,
nothing else is bound, and the #[repr(C)] structs list the header's fields in the header's order."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def _split_params(params):
    params = params.strip()
    if params in ("", "void"):
        return []
    return [p.strip() for p in params.split(",")]


def header_prototypes():
    hdr = _strip_comments(open(os.path.join(ROOT, "include", "rt_abi.h")).read())
    out = {}
    for m in re.finditer(r"\b(rt_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", hdr, flags=re.S):
        out[m.group(1)] = _split_params(m.group(2))
    return out


def rust_prototypes():
    src = _strip_comments(open(os.path.join(ROOT, "integration", "rust", "ffi.rs")).read())
    block = re.search(r'extern\s+"C"\s*\{(.*?)\n\}', src, flags=re.S).group(1)
    out = {}
    for m in re.finditer(r"pub\s+fn\s+(rt_[a-z0-9_]+)\s*\(([^)]*)\)", block, flags=re.S):
        out[m.group(1)] = _split_params(m.group(2))
    return out


def test_ffi_binds_every_entry_point_with_the_headers_arity():
    import raytracer_amd as ra
    h, r = header_prototypes(), rust_prototypes()
    assert set(h) == set(ra.ABI_SYMBOLS), set(h) ^ set(ra.ABI_SYMBOLS)
    assert set(r) == set(h), set(r) ^ set(h)
    for name in h:
        assert len(r[name]) == len(h[name]), (name, h[name], r[name])
    # parameter names agree too where the header names them (catches swapped arguments)
    for name in h:
        for hp, rp in zip(h[name], r[name]):
            hname = re.findall(r"([A-Za-z_][A-Za-z0-9_]*)\s*(?:\[\d*\])?$", hp)[0]
            rname = rp.split(":")[0].strip()
            assert hname == rname, (name, hp, rp)


def _c_struct_fields(hdr, name):
    body = re.search(r"typedef struct " + name + r"\s*\{(.*?)\}\s*" + name + r"\s*;", hdr, flags=re.S).group(1)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        decl = re.sub(r"^(const\s+)?[A-Za-z_][A-Za-z0-9_]*(\s*\*)?\s+", "", decl, count=1)  # drop the type
        for part in decl.split(","):
            fields.append(re.match(r"\s*\*?\s*([A-Za-z_][A-Za-z0-9_]*)", part).group(1))
    return fields


def _rust_struct_fields(src, name):
    body = re.search(r"pub struct " + name + r"\s*\{(.*?)\n?\}", src, flags=re.S).group(1)
    return [f.rstrip("_") for f in re.findall(r"pub\s+([A-Za-z_][A-Za-z0-9_]*)\s*:", body)]


def test_repr_c_structs_list_the_headers_fields_in_order():
    hdr = _strip_comments(open(os.path.join(ROOT, "include", "rt_abi.h")).read())
    src = _strip_comments(open(os.path.join(ROOT, "integration", "rust", "ffi.rs")).read())
    for name in ("rt_prim", "rt_material", "rt_texture", "rt_scene_desc", "rt_camera", "rt_ray", "rt_hit", "rt_stats",
                 "rt_multi_stats"):
        assert _rust_struct_fields(src, name) == _c_struct_fields(hdr, name), name


def test_shim_files_are_shipped_and_say_they_are_uncompiled():
    for f in ("build.rs", "ffi.rs", "flatten.rs", "camera_image.rs", "render.rs", "README.md"):
        text = open(os.path.join(ROOT, "integration", "rust", f)).read()
        assert "ncompiled" in text or "never been compiled" in text, f
    build = open(os.path.join(ROOT, "integration", "rust", "build.rs")).read()
    for src in ("rt_lib.hip", "rt_multi.hip", "rt_bvh.cpp", "--offload-arch=gfx950"):
        assert src in build
