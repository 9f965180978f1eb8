"""GPU tests of the progressive / checkpointed form of Camera::render: per-pass tile
refresh for the preview channel (ui.rs:56-77 consumes the same tuples), checkpoint + resume of the accumulated
sums, and equality of the final frame with the single-shot render."""
import os

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


def test_progressive_final_frame_equals_single_shot(ctx, tmp_path):
    g = ra.Scene("book_cover", 80, ASPECT, "uniform", 16, 8)
    ref, tiles = g.render(ctx, seed=3)
    ppm = tmp_path / "image.ppm"
    frame, rendered, n_msgs = g.render_progressive(ctx, passes=4, seed=3, ppm_path=str(ppm))
    assert rendered == 4 and n_msgs == 4 * len(tiles)  # every tile re-sent after every pass
    # counter-based sample keys: the partition into passes changes only the f32 summation order
    assert np.allclose(frame, ref, rtol=0, atol=2e-6)
    assert ppm.read_text().split("\n")[1] == "80 45"
    # passes need not divide the sample count; the random sampler works too
    g2 = ra.Scene("two_spheres", 64, ASPECT, "random", 10, 8)
    ref2, _ = g2.render(ctx, seed=5)
    frame2, rendered2, _ = g2.render_progressive(ctx, passes=3, seed=5)
    assert rendered2 == 3 and np.allclose(frame2, ref2, rtol=0, atol=2e-6)


def test_progressive_preview_converges_pass_by_pass(ctx):
    g = ra.Scene("book_cover", 64, ASPECT, "uniform", 64, 8)
    ref, tiles = g.render(ctx, seed=9)
    snaps, cur = [], np.zeros_like(ref)
    state = {"n": 0}

    def on_tile(top, left, h, w, rgb):
        cur[top:top + h, left:left + w] = rgb
        state["n"] += 1
        if state["n"] % len(tiles) == 0:
            snaps.append(cur.copy())

    g.render_progressive(ctx, passes=8, seed=9, on_tile=on_tile)
    assert len(snaps) == 8
    err = [float(np.sqrt(np.mean((s - ref) ** 2))) for s in snaps]
    assert err[-1] < 2e-6 and err[0] > 4 * err[-2] > 0  # 8 spp preview is noisy, 56 spp is close


def test_checkpoint_resume_after_cancel(ctx, tmp_path):
    g = ra.Scene("book_cover", 80, ASPECT, "uniform", 16, 8)
    ref, tiles = g.render(ctx, seed=11)
    ck = str(tmp_path / "render.ckpt")
    state = {"n": 0}

    def hang_up_in_third_pass(*_a):
        state["n"] += 1
        return state["n"] <= 2 * len(tiles)  # False on the first tile of pass 3

    ppm = tmp_path / "image.ppm"
    _, rendered, _ = g.render_progressive(ctx, passes=4, seed=11, checkpoint_path=ck, ppm_path=str(ppm),
                                          on_tile=hang_up_in_third_pass)
    assert rendered == 3 and not ppm.exists() and os.path.exists(ck)  # pass 3 was rendered + saved, then cancelled
    assert os.path.getsize(ck) == 72 + 80 * 45 * 3 * 4 and not os.path.exists(ck + ".tmp")
    frame, rendered, n_msgs = g.render_progressive(ctx, passes=4, seed=11, checkpoint_path=ck, ppm_path=str(ppm))
    assert rendered == 1 and n_msgs == len(tiles) and ppm.exists()  # only the missing pass is rendered
    assert np.allclose(frame, ref, rtol=0, atol=2e-6)
    # a finished checkpoint renders nothing more but still delivers the frame
    frame, rendered, n_msgs = g.render_progressive(ctx, passes=4, seed=11, checkpoint_path=ck)
    assert rendered == 0 and n_msgs == len(tiles) and np.allclose(frame, ref, rtol=0, atol=2e-6)
    # a checkpoint of another job (different seed) is ignored, not resumed
    frame, rendered, _ = g.render_progressive(ctx, passes=4, seed=12, checkpoint_path=ck)
    assert rendered == 4 and not np.allclose(frame, ref, atol=1e-3)
    # nor is one of another scene, or of the same scene at another depth, with the same size / spp / passes / seed:
    # the header carries a hash of the flattened scene and of the camera block
    other = ra.Scene("two_spheres", 80, ASPECT, "uniform", 16, 8)
    other_ref, _ = other.render(ctx, seed=12)
    frame, rendered, _ = other.render_progressive(ctx, passes=4, seed=12, checkpoint_path=ck)
    assert rendered == 4 and np.allclose(frame, other_ref, rtol=0, atol=2e-6)
    deeper = ra.Scene("two_spheres", 80, ASPECT, "uniform", 16, 9)
    _, rendered, _ = deeper.render_progressive(ctx, passes=4, seed=12, checkpoint_path=ck)
    assert rendered == 4


def test_progressive_argument_errors(ctx):
    g = ra.Scene("two_spheres", 32, ASPECT, "uniform", 4, 4)
    with pytest.raises(RuntimeError, match="passes must be in 1"):
        g.render_progressive(ctx, passes=5)
    with pytest.raises(RuntimeError, match="passes must be in 1"):
        g.render_progressive(ctx, passes=0)


def test_ten_thousand_samples_per_pixel_in_one_launch_and_in_slices(ctx, tmp_path):
    """BASELINE config 5 is quoted at 10 000 spp (a 100 x 100 grid).  The bookkeeping that only such a K exercises — a
    tile's path count `rem = n_valid * K` (640 000), the fixed-point format chosen from K (2^-29 at K = 10 000, 2^-31 at
    2500), 32-bit sample arithmetic, chunks of a 10 000-deep tile — on a small frame of the final scene's types (lights,
    media, glass, image and noise textures): ONE launch of all 10 000 sample indices equals the sum of four stride-4
    slices of 2500 (a path depends only on (seed, pixel, sample index)), up to f32 rounding of the four partial sums and
    the slices' finer fixed-point grid; and the checkpointed progressive form arrives at the same frame."""
    from golden_data import load_earth_rgba
    g = ra.Scene("final_scene_small", 64, ASPECT, "uniform", 10000, 50, texels=load_earth_rgba())
    assert g.spp == 10000 and g.camera.sampler_n == 100
    dev = g.upload(ctx)
    cam = g.camera_copy()
    whole = dev.render_sums(cam, seed=21)
    st = ctx.stats()
    assert st["paths"] == g.width * g.height * 10000 and st["samples_nan"] == 0 and st["samples_clamped"] == 0
    parts = [dev.render_sums(cam, seed=21, sample_begin=r, sample_stride=4) for r in range(4)]
    assert ctx.stats()["paths"] == g.width * g.height * 2500
    total = np.sum(np.stack(parts).astype(np.float64), axis=0)
    assert whole.max() > 100.0  # thousands of samples really were added up
    # 10 000 truncations of at most 2^-29 each + f32 rounding of sums of magnitude <= ~1e5
    assert np.allclose(whole, total, rtol=3e-7, atol=10000 * 2.0 ** -29 + 1e-6)
    # two launches of the same sample set are identical bit for bit (integer sums)
    assert np.array_equal(dev.render_sums(cam, seed=21, sample_begin=1, sample_stride=4), parts[1])
    # checkpointed passes of 2500 (Camera::render_progressive) resolve to the single-shot frame
    ref = ra.resolve(ctx, whole, 10000)
    ck = str(tmp_path / "k10000.ckpt")
    frame, rendered, _ = g.render_progressive(ctx, passes=4, seed=21, checkpoint_path=ck)
    assert rendered == 4 and np.allclose(frame, ref, rtol=0, atol=2e-6)
    frame, rendered, _ = g.render_progressive(ctx, passes=4, seed=21, checkpoint_path=ck)
    assert rendered == 0 and np.allclose(frame, ref, rtol=0, atol=2e-6)
    dev.close()
