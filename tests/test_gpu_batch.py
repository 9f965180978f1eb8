"""rt_render_batch_device — several frames of one scene and camera in ONE persistent launch — against rt_render_device,
through the C ABI: every frame of a batch must be BIT-IDENTICAL to the single launch of the same (seed, sample_begin,
sample_end, sample_stride), whatever the batch's size, on the LDS-resident instance with and without its partner
workgroups, on the HBM-resident one, and with tile-order feedback primed."""
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


def _single(dev, cam, w, h, seed, begin, stride, end=None):
    import torch
    d = torch.full((h, w, 3), -1.0, dtype=torch.float32, device="cuda")
    dev.render_device(cam, d.data_ptr(), seed=seed, sample_begin=begin, sample_end=end, sample_stride=stride,
                      stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return d.cpu().numpy()


def _batch(dev, cam, w, h, seeds, begins, stride, end=None):
    import torch
    outs = [torch.full((h, w, 3), -1.0, dtype=torch.float32, device="cuda") for _ in seeds]  # poisoned
    ra.render_batch_device(dev, cam, [o.data_ptr() for o in outs], seeds, begins, sample_end=end, sample_stride=stride,
                           stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    st = dev.ctx.stats()  # raises if the watchdog cut a frame short
    return [o.cpu().numpy() for o in outs], st


@pytest.mark.parametrize("name,width,spp,stride", [
    ("book_cover", 320, 16, 1),      # LDS-resident lean instance, whole frames with different seeds
    ("book_cover", 1200, 64, 8),     # C2's frame split 8 ways (8 sample indices per frame): strong-scaling slices
    ("cornell_smoke", 200, 16, 4),   # general scene in LDS, media (one RNG draw inside hit())
    ("final_scene", 240, 16, 2),     # HBM-resident four-wide instance, merged boxes
])
def test_batch_frames_equal_single_launches(ctx, name, width, spp, stride):
    tex = None
    if name.startswith("final"):
        from golden_data import load_earth_rgba
        tex = load_earth_rgba()
    g = ra.Scene(name, width, ASPECT, "uniform", spp, 50, texels=tex)
    dev, cam = g.upload(ctx), g.camera_copy()
    n = 5
    seeds = [1000 + 7 * i for i in range(n)]
    begins = [i % stride for i in range(n)]
    ref = [_single(dev, cam, g.width, g.height, seeds[i], begins[i], stride) for i in range(n)]
    got, st = _batch(dev, cam, g.width, g.height, seeds, begins, stride)
    assert st["paths"] == g.width * g.height * (spp // stride) * n
    for i in range(n):
        assert got[i].min() >= 0.0, f"frame {i}: a value was never written"
        assert np.array_equal(got[i], ref[i]), f"frame {i} of the batch differs from its single launch"
    # a batch of one, and a second batch right after the first (tile-order feedback primed by it)
    one, _ = _batch(dev, cam, g.width, g.height, seeds[:1], begins[:1], stride)
    assert np.array_equal(one[0], ref[0])
    again, _ = _batch(dev, cam, g.width, g.height, seeds[::-1], begins[::-1], stride)
    for i in range(n):
        assert np.array_equal(again[i], ref[n - 1 - i])
    dev.close()


def test_batch_argument_errors(ctx):
    import torch
    g = ra.Scene("two_spheres", 64, ASPECT, "uniform", 16, 8)
    dev, cam = g.upload(ctx), g.camera_copy()
    d = [torch.zeros((g.height, g.width, 3), dtype=torch.float32, device="cuda") for _ in range(2)]
    ptrs = [t.data_ptr() for t in d]
    with pytest.raises(ra.RtError) as e:  # 16 indices from 0 with stride 3 -> 6 samples, from 1 -> 5: unequal
        ra.render_batch_device(dev, cam, ptrs, [1, 2], [0, 1], sample_stride=3)
    assert e.value.code == ra.RT_E_INVALID_ARG and "split the batch" in str(e.value)
    with pytest.raises(ra.RtError):       # a frame that selects nothing
        ra.render_batch_device(dev, cam, ptrs, [1, 2], [16, 16])
    ctx.set_option("mode", 0)
    try:
        with pytest.raises(ra.RtError) as e:
            ra.render_batch_device(dev, cam, ptrs, [1, 2], [0, 0])
        assert e.value.code == ra.RT_E_UNSUPPORTED
    finally:
        ctx.set_option("mode", 1)
    ra.render_batch_device(dev, cam, ptrs, [1, 2], [0, 0])
    ctx.synchronize()
    assert float(d[0].sum()) > 0 and float(d[1].sum()) > 0 and not torch.equal(d[0], d[1])
    dev.close()
