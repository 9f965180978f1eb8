"""rt_render_device_noalloc — the entry point that allocates nothing — against rt_render_device, through the C ABI.

Both run the same kernels; the only difference is where the render's scratch (camera block of the frame in flight,
tile counter, watchdog flag, ray counter) lives: library-allocated memory vs the library's fixed file-scope device
array.  Fused frames are sums of integers, so the two must agree BIT FOR BIT (the bar of the other render tests).  Two
calls in a row, the second with a smaller frame: the array keeps the first call's contents (a tile counter that has
run past the first frame's tile list), and the second must not see them.
"""
import ctypes as C

import numpy as np
import pytest

import raytracer_amd as ra

pytestmark = pytest.mark.gpu

ASPECT = 16.0 / 9.0


def _render(dev, cam, w, h, seed, no_alloc, **kw):
    import torch
    d = torch.full((h, w, 3), -1.0, dtype=torch.float32, device="cuda")  # poisoned: the call must write every value
    stream = torch.cuda.current_stream().cuda_stream
    dev.render_device(cam, d.data_ptr(), seed=seed, stream=stream, no_alloc=no_alloc, **kw)
    torch.cuda.synchronize()
    return d.cpu().numpy()


def test_noalloc_entry_matches_allocating_entry_twice_second_smaller():
    import torch
    ctx = ra.Context(0)
    try:
        big = ra.Scene("book_cover", 384, ASPECT, "uniform", 36, 50)   # 384 x 216: partner workgroups not engaged
        small = ra.Scene("book_cover", 200, ASPECT, "random", 10, 50)  # 200 x 112, ragged 8x8 tiles, other sampler
        dev_big, dev_small = big.upload(ctx), small.upload(ctx)
        cam_big, cam_small = big.camera_copy(), small.camera_copy()
        ref_big = _render(dev_big, cam_big, big.width, big.height, 11, False)
        ref_small = _render(dev_small, cam_small, small.width, small.height, 12, False)
        ref_part = _render(dev_small, cam_small, small.width, small.height, 12, False, sample_begin=1, sample_stride=3)
        assert ref_big.min() >= 0 and ref_big.std() > 0 and ref_small.std() > 0
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        # the checkable signal: the library counts every hipMalloc / hipFree / stream / event creation it makes, in any
        # context of the process (rt_stats.lib_alloc_calls); the count must not move across the no-alloc calls
        calls0 = ctx.stats()["lib_alloc_calls"]
        assert calls0 > 0  # rt_init and the uploads allocated
        got_big = _render(dev_big, cam_big, big.width, big.height, 11, True)
        rays_big = ctx.stats()["rays"]
        got_small = _render(dev_small, cam_small, small.width, small.height, 12, True)
        got_part = _render(dev_small, cam_small, small.width, small.height, 12, True, sample_begin=1, sample_stride=3)
        assert ctx.stats()["lib_alloc_calls"] == calls0, "rt_render_device_noalloc allocated or created a runtime object"
        # (torch's caching allocator serves the output tensors from its pool: device memory in use must not have grown)
        assert torch.cuda.mem_get_info()[0] >= free0
        # ... and the counter does see an allocating path: the first feedback-ordered frame of the allocating entry point
        # allocates its tile lists
        _render(dev_big, cam_big, big.width, big.height, 11, False)
        ctx2 = ra.Context(0)
        assert ctx.stats()["lib_alloc_calls"] > calls0
        ctx2.close()
        assert np.array_equal(got_big, ref_big)
        assert np.array_equal(got_small, ref_small)
        assert np.array_equal(got_part, ref_part)
        assert rays_big > big.width * big.height * 36  # stats come from the fixed array's ray counter
        # the C2-class frame (1200 x 675) with the partner workgroups, both entries
        c2 = ra.Scene("book_cover", 1200, ASPECT, "uniform", 4, 50)
        dev_c2, cam_c2 = c2.upload(ctx), c2.camera_copy()
        assert c2.width * c2.height <= ra.RT_NOALLOC_MAX_PIXELS
        assert np.array_equal(_render(dev_c2, cam_c2, c2.width, c2.height, 5, True),
                              _render(dev_c2, cam_c2, c2.width, c2.height, 5, False))
        # and once more the small one after the large one
        assert np.array_equal(_render(dev_small, cam_small, small.width, small.height, 12, True), ref_small)
    finally:
        ctx.close()


def test_noalloc_entry_rejects_what_it_cannot_hold():
    import torch
    ctx = ra.Context(0)
    try:
        sc = ra.Scene("two_spheres", 5600, ASPECT, "uniform", 1, 8)  # 5600 x 3150 > 2^24 pixels
        assert sc.width * sc.height > ra.RT_NOALLOC_MAX_PIXELS
        dev, cam = sc.upload(ctx), sc.camera_copy()
        d = torch.zeros((sc.height, sc.width, 3), dtype=torch.float32, device="cuda")
        with pytest.raises(ra.RtError) as e:
            dev.render_device(cam, d.data_ptr(), seed=1, no_alloc=True)
        assert e.value.code == ra.RT_E_UNSUPPORTED and "exceeds the stated maximum" in str(e.value)
        dev.render_device(cam, d.data_ptr(), seed=1)  # the allocating entry still takes it
        ctx.synchronize()
        assert float(d.sum()) > 0
        small = ra.Scene("two_spheres", 64, ASPECT, "uniform", 1, 8)
        dev_s, cam_s = small.upload(ctx), small.camera_copy()
        ctx.set_option("mode", 0)  # the wavefront pipeline needs queues the library allocates
        with pytest.raises(ra.RtError) as e:
            dev_s.render_device(cam_s, d.data_ptr(), seed=1, no_alloc=True)
        assert e.value.code == ra.RT_E_UNSUPPORTED
        ctx.set_option("mode", 1)
    finally:
        ctx.close()
