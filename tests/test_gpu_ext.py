"""GPU parity for the EXT rows of SURVEY §8 (aX1-aX5: quads, boxes, translate / rotate-Y instances, constant media,
emissive materials + solid background, Cornell and final-scene builders).  These types do not exist in the reference
("parity unpinned" by it); parity here is GPU vs the build's own CPU oracle, which implements them with ray-space
instance transforms while the GPU path bakes the transforms into the primitives at flatten time."""
import numpy as np
import pytest

import raytracer_amd as ra
from oracle_lib import OracleScene
from test_gpu_parity import image_parity, psnr  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = ra.Context(0)
    yield c
    c.close()


def pair(name, width, sampler, spp, depth):
    tex = None
    if name.startswith("final"):
        from golden_data import load_earth_rgba
        tex = load_earth_rgba()
    return (OracleScene(name, width, 1.0, sampler, spp, depth, texels=tex),
            ra.Scene(name, width, 1.0, sampler, spp, depth, texels=tex))


@pytest.mark.parametrize("name,width", [("cornell_box", 200), ("cornell_smoke", 160), ("final_scene_small", 200),
                                        ("final_scene", 160)])
@pytest.mark.parametrize("form", [0, 3, 4])  # probe_tree: 0 diagnostic, 3 / 4 the render kernels' own traversals
def test_ext_primary_hit_parity(ctx, name, width, form):
    o, g = pair(name, width, "uniform", 1, 50)
    rays = o.primary_rays(time=0.25)
    ho = o.intersect(rays)
    ctx.set_option("probe_tree", form)
    try:
        hg = g.upload(ctx).intersect(rays)
    except ra.RtError as e:
        if form == 3 and e.code == ra.RT_E_UNSUPPORTED and name == "final_scene":
            pytest.skip("final_scene does not fit in LDS: its render instances walk form 4")
        raise
    finally:
        ctx.set_option("probe_tree", 0)
    prims = g.prims()
    medium_ids = set(prims["object_id"][prims["type"] >= 3].tolist())
    # constant media scatter at random distances (one RNG draw inside hit()): exclude rays where either side
    # reports a medium event; all the others must agree exactly
    is_med = np.isin(ho["object_id"], list(medium_ids)) | np.isin(hg["object_id"], list(medium_ids))
    d = rays["d"] / np.linalg.norm(rays["d"], axis=1, keepdims=True)
    graz = ((ho["object_id"] >= 0) & (np.abs(np.sum(ho["n"] * d, axis=1)) < 1e-3)) | \
           ((hg["object_id"] >= 0) & (np.abs(np.sum(hg["n"] * d, axis=1)) < 1e-3))
    same = ho["object_id"] == hg["object_id"]
    # quads that share an edge (box faces) give equal-t ties on the seam: allow a different face when t agrees
    tie = ~same & (ho["object_id"] >= 0) & (hg["object_id"] >= 0) & \
        (np.abs(ho["t"] - hg["t"]) <= 1e-9 * np.maximum(np.abs(ho["t"]), 1.0))
    bad = ~same & ~is_med & ~graz & ~tie
    assert bad.sum() == 0, (bad.sum(), np.flatnonzero(bad)[:5])
    both = same & (ho["object_id"] >= 0) & ~is_med
    assert both.sum() > 0.3 * len(rays)
    assert (np.abs(hg["t"][both] - ho["t"][both]) / np.abs(ho["t"][both])).max() <= 1e-5
    assert np.abs(hg["n"][both] - ho["n"][both]).max() <= 1e-5
    assert (hg["front_face"][both & ~graz] == ho["front_face"][both & ~graz]).all()
    if medium_ids:
        # media are hit with comparable frequency on both sides (same physics, different random streams)
        fo = np.isin(ho["object_id"], list(medium_ids)).mean()
        fg = np.isin(hg["object_id"], list(medium_ids)).mean()
        assert abs(fo - fg) < 0.02 + 0.1 * fo, (fo, fg)


@pytest.mark.parametrize("name,spp,depth", [("cornell_box", 4096, 20), ("cornell_smoke", 4096, 20),
                                            ("final_scene_small", 2304, 20),
                                            # the full scene does not fit LDS: four-wide binary16 tree, top in LDS
                                            ("final_scene", 1024, 20)])
@pytest.mark.parametrize("mode", [1, 0])
def test_ext_converged_image_parity(ctx, name, spp, depth, mode):
    if mode == 0 and name != "cornell_smoke":
        pytest.skip("wavefront mode is cross-checked on the medium scene only (time)")
    tex = None
    if name.startswith("final"):
        from golden_data import load_earth_rgba
        tex = load_earth_rgba()
    o = OracleScene(name, 48, 1.0, "uniform", spp, depth, texels=tex)
    g = ra.Scene(name, 48, 1.0, "uniform", spp, depth, texels=tex)
    ctx.set_option("mode", mode)
    try:
        dev = g.upload(ctx)
        cam = g.camera_copy()
        a = dev.render_sums(cam, seed=11).astype(np.float64) / g.spp
        b = dev.render_sums(cam, seed=12).astype(np.float64) / g.spp
    finally:
        ctx.set_option("mode", 1)
    ref = o.render(threads=0)["linear"]
    noise = float(np.sqrt(np.mean((a - b) ** 2)))
    rmse = float(np.sqrt(np.mean((a - ref) ** 2)))
    se = noise / np.sqrt(a.shape[0] * a.shape[1])
    bias = np.abs((a - ref).mean(axis=(0, 1)))
    print(name, mode, dict(noise=noise, rmse=rmse, bias=bias, se=se, mean=ref.mean()))
    # small emitters => high variance: the tolerance is the measured Monte-Carlo noise, not a fixed PSNR
    assert rmse <= 1.5 * noise + 1e-3, (rmse, noise)
    assert (bias <= 5.0 * se + 5e-4).all(), (bias, se)
    ga, gr = np.clip(a, 0, None) ** (1 / 2.2), np.clip(ref, 0, None) ** (1 / 2.2)
    # the full final scene at 1024 spp is noisier than the others at theirs (rmse == noise, checked above): 24 dB
    assert psnr(np.clip(ga, 0, 1), np.clip(gr, 0, 1)) >= (24.0 if name == "final_scene" else 28.0)


def test_background_modes(ctx):
    """mode 0 = the reference's sky gradient; mode 1 = solid colour: an empty-ish view returns exactly it."""
    o, g = pair("cornell_box", 32, "uniform", 4, 5)
    dev = g.upload(ctx)
    cam = g.camera_copy()
    cam.center[2] = -2000.0  # look from far behind, walls subtend little: most rays miss
    cam.pixel00[0] += 5000.0  # shift the window sideways so every ray misses the box
    cam.background[0], cam.background[1], cam.background[2] = 0.25, 0.5, 0.75
    s = dev.render_sums(cam, seed=1) / 4.0
    assert np.allclose(s[..., 0], 0.25, atol=1e-6) and np.allclose(s[..., 2], 0.75, atol=1e-6)
    cam.background_mode = 0
    s0 = dev.render_sums(cam, seed=1) / 4.0
    assert (s0[..., 2] >= s0[..., 0] - 1e-6).all() and s0.min() >= 0.5 - 1e-6  # sky: white..(0.5,0.7,1.0)


def test_traversal_schedule_does_not_change_ext_frames(ctx):
    """Media draw random numbers INSIDE the traversal (SURVEY aX3); a lane's traversal order is its own, so the
    resumable schedule and its thresholds must not change which numbers a path sees."""
    from golden_data import load_earth_rgba
    g = ra.Scene("final_scene_small", 160, 1.0, "uniform", 16, 30, texels=load_earth_rgba())
    dev = g.upload(ctx)
    cam = g.camera_copy()
    a = dev.render_sums(cam, seed=4)
    rays = ctx.stats()["rays"]
    try:
        ctx.set_option("fused_kb", 48); ctx.set_option("fused_kn", 3); ctx.set_option("fused_kr", 5)
        b = dev.render_sums(cam, seed=4)
        assert np.array_equal(a, b) and ctx.stats()["rays"] == rays
        ctx.set_option("fused_resume", 0)
        c = dev.render_sums(cam, seed=4).astype(np.float64)
    finally:
        ctx.set_option("fused_resume", 2)
        ctx.set_option("fused_kb", 16); ctx.set_option("fused_kn", 8); ctx.set_option("fused_kr", 16)
    a64 = a.astype(np.float64)
    assert abs(ctx.stats()["rays"] - rays) <= 2e-3 * rays
    assert np.isclose(a64, c, rtol=1e-4, atol=1e-3).all(axis=2).mean() > 0.97
    assert abs(a64.sum() - c.sum()) < 5e-3 * a64.sum()


@pytest.mark.parametrize("name", ["cornell_box", "final_scene_small"])
def test_hbm_resident_tree_forms_agree_on_ext_scenes(ctx, name):
    """The instances for scenes that do not fit LDS (four-wide tree with binary16 planes and the top cached in LDS;
    four-wide f32; binary) against the LDS-resident instance on the same scene.  Closest hits do not depend on the
    tree, so without media the frames agree path by path; with media the visiting order decides which random numbers
    a boundary test draws, so those frames agree as estimates of the same image."""
    from golden_data import load_earth_rgba
    g = ra.Scene(name, 128, 1.0, "uniform", 64, 30, texels=load_earth_rgba() if name.startswith("final") else None)
    dev = g.upload(ctx)
    cam = g.camera_copy()
    a = dev.render_sums(cam, seed=9).astype(np.float64)
    a2 = dev.render_sums(cam, seed=10).astype(np.float64)
    frames = []
    try:
        ctx.set_option("fused_lds", 0)
        frames.append(dev.render_sums(cam, seed=9).astype(np.float64))
        ctx.set_option("fused_half", 0)
        frames.append(dev.render_sums(cam, seed=9).astype(np.float64))
        ctx.set_option("fused_wide", 0)
        frames.append(dev.render_sums(cam, seed=9).astype(np.float64))
    finally:
        ctx.set_option("fused_lds", 1); ctx.set_option("fused_half", 1); ctx.set_option("fused_wide", 1)
    noise = np.sqrt(np.mean((a - a2) ** 2))
    for f in frames:
        assert abs(f.sum() - a.sum()) < 5e-3 * a.sum()
        if name == "cornell_box":
            assert np.isclose(a, f, rtol=1e-4, atol=1e-3).all(axis=2).mean() > 0.97
        else:
            assert np.sqrt(np.mean((a - f) ** 2)) < 1.2 * noise  # no further apart than two independent seeds


@pytest.mark.parametrize("name", ["final_scene_small", "final_scene", "cornell_smoke"])
def test_global_primitives_with_media_estimate_the_same_image(ctx, name):
    """Option global_prims on scenes with constant media: the final scenes' enclosing haze sphere leaves the tree and is
    tested first, Cornell smoke (eight primitives) has no tree at all.  A medium's boundary test draws from the path's
    random stream, and the order of the tests changes — so the frames are two estimates of the same image, no further
    apart than two seeds of one configuration, with the same total energy."""
    from golden_data import load_earth_rgba
    tex = load_earth_rgba() if name.startswith("final") else None
    g = ra.Scene(name, 128, 1.0, "uniform", 64, 30, texels=tex)
    cam = g.camera_copy()
    frames = {}
    for on in (1, 0):
        ctx.set_option("global_prims", on)
        try:
            dev = g.upload(ctx)
        finally:
            ctx.set_option("global_prims", 1)
        frames[on] = [dev.render_sums(cam, seed=s).astype(np.float64) for s in ((9, 10) if on else (9,))]
        dev.close()
    a, a2 = frames[1]
    b = frames[0][0]
    noise = np.sqrt(np.mean((a - a2) ** 2))
    assert abs(b.sum() - a.sum()) < 5e-3 * a.sum()
    assert np.sqrt(np.mean((a - b) ** 2)) < 1.2 * noise


@pytest.mark.parametrize("name,width", [("book_cover", 300), ("cornell_box", 200), ("final_scene", 240)])
def test_every_tree_form_returns_the_same_hits(ctx, name, width):
    """rt_intersect over the binary tree is held against the oracle above; here the four-wide trees (f32 and binary16
    planes) that the HBM-resident render kernels walk must return the very same hits, ray by ray: primary rays and
    rays leaving surfaces in random directions.  Media draw random numbers, so rays on which any form reports a
    medium are only counted."""
    if name == "book_cover":
        from test_gpu_parity import make_pair
        o, g = make_pair(name, width, "uniform", 1, 50)
    else:
        o, g = pair(name, width, "uniform", 1, 50)
    dev = g.upload(ctx)
    prims = g.prims()
    medium_ids = list(set(prims["object_id"][prims["type"] >= 3].tolist()))
    rays = o.primary_rays(time=0.4)
    base = dev.intersect(rays)
    hit = base["object_id"] >= 0
    rng = np.random.default_rng(11)
    sec = np.zeros(int(hit.sum()), dtype=ra.RAY_DTYPE)
    u = rng.normal(size=(len(sec), 3))
    sec["o"] = base["p"][hit]
    sec["d"] = base["n"][hit] + u / np.linalg.norm(u, axis=1, keepdims=True)
    sec["time"] = 0.4
    for batch in (rays, sec):
        ctx.set_option("probe_tree", 0)
        h0 = dev.intersect(batch)
        try:
            for tree in (1, 2):
                ctx.set_option("probe_tree", tree)
                h = dev.intersect(batch)
                med = np.isin(h0["object_id"], medium_ids) | np.isin(h["object_id"], medium_ids)
                same_t = h["t"] == h0["t"]  # bit-equal: the same f64 primitive test on the same primitive
                tie = same_t & (h["object_id"] != h0["object_id"])  # seams of box faces: equal t on two quads
                ok = med | (same_t & ((h["object_id"] == h0["object_id"]) | tie))
                assert ok.all(), (tree, int((~ok).sum()), np.flatnonzero(~ok)[:5])
                assert tie.sum() <= 1e-3 * len(batch) and med.mean() < 0.5
                same = ~med & (h["object_id"] == h0["object_id"])
                assert np.array_equal(h["n"][same], h0["n"][same]) and np.array_equal(h["p"][same], h0["p"][same])
        finally:
            ctx.set_option("probe_tree", 0)


@pytest.mark.parametrize("name,width", [("final_scene_small", 200), ("final_scene", 200)])
@pytest.mark.parametrize("form", [0, 4])
def test_merged_boxes_return_the_quads_hits(ctx, name, width, form):
    """rt_scene_upload merges six quads that form an axis-aligned box into one primitive of the tree (option merge_boxes,
    default on; final scene: 3 409 -> 1 409 tree primitives).  A hit reports the quad of the face that was hit, and its t
    is the same division the quad's own test performs: against the unmerged scene the object ids agree except on seams
    between faces (equal t), t is BIT-IDENTICAL wherever the ids agree, and the frames estimate the same image."""
    from golden_data import load_earth_rgba
    tex = load_earth_rgba()
    o = OracleScene(name, width, 1.0, "uniform", 1, 50, texels=tex)
    g = ra.Scene(name, width, 1.0, "uniform", 16, 50, texels=tex)
    rays = o.primary_rays(time=0.25)
    # secondary-like rays too: from points on the ground boxes upwards and sideways (origins ON box faces)
    rng = np.random.default_rng(5)
    extra = rays[: len(rays) // 2].copy()
    ho = o.intersect(rays)
    sel = np.flatnonzero(ho["object_id"] >= 0)[: len(extra)]
    extra = extra[: len(sel)]
    extra["o"] = ho["p"][sel]
    extra["d"] = rng.normal(size=(len(sel), 3))
    allrays = np.concatenate([rays, extra])
    hits = {}
    for merge in (0, 1):
        ctx.set_option("merge_boxes", merge)
        ctx.set_option("probe_tree", form)
        try:
            dev = g.upload(ctx)
            hits[merge] = dev.intersect(allrays)
            if merge == 1:
                frame_m = dev.render_sums(g.camera_copy(), seed=3)
            else:
                frame_q = dev.render_sums(g.camera_copy(), seed=3)
            dev.close()
        finally:
            ctx.set_option("probe_tree", 0)
            ctx.set_option("merge_boxes", 1)
    a, b = hits[0], hits[1]
    prims = g.prims()
    medium_ids = list(set(prims["object_id"][prims["type"] >= 3].tolist()))
    is_med = np.isin(a["object_id"], medium_ids) | np.isin(b["object_id"], medium_ids)  # media scatter at random
    same = a["object_id"] == b["object_id"]
    tie = ~same & (a["object_id"] >= 0) & (b["object_id"] >= 0) & (np.abs(a["t"] - b["t"]) <= 1e-9 * np.maximum(np.abs(a["t"]), 1.0))
    bad = ~same & ~tie & ~is_med
    assert bad.sum() <= 2e-5 * len(allrays), (int(bad.sum()), np.flatnonzero(bad)[:5])  # (rays through a box edge within rounding)
    both = same & (a["object_id"] >= 0) & ~is_med
    assert both.sum() > 0.3 * len(allrays)
    assert np.array_equal(a["t"][both], b["t"][both])  # the same division
    assert np.array_equal(a["n"][both], b["n"][both]) and np.array_equal(a["front_face"][both], b["front_face"][both])
    # whole frames: different trees give the media's free-path draws a different order => statistical agreement
    assert abs(frame_m.mean() - frame_q.mean()) <= 0.02 * frame_q.mean()
