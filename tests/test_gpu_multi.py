"""Multi-GPU paths on the one-GPU box: rt_multi_* (library-owned contexts and host threads, frames summed by the
peer-reading kernel) with one device listed several times, and the one-process-per-GPU path (render_split + reduce)
with two gloo ranks sharing GPU 0.  Bar: the N-way split reproduces the single-context frame — bit for bit where the
summation order is the same, to f32 rounding otherwise."""
import os
import socket

import numpy as np
import pytest

import raytracer_amd as ra

pytestmark = pytest.mark.gpu

ASPECT = 16.0 / 9.0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("n_dev,name,sampler,spp", [(3, "book_cover", "uniform", 36), (4, "book_cover", "random", 10),
                                                    (2, "cornell_smoke", "uniform", 16), (8, "two_spheres", "uniform", 4)])
def test_rt_multi_equals_the_single_context_frame(n_dev, name, sampler, spp):
    sc = ra.Scene(name, 120, ASPECT, sampler, spp, 20)
    cam = sc.camera_copy()
    ctx = ra.Context(0)
    dev = sc.upload(ctx)
    whole = dev.render_sums(cam, seed=21)
    parts = [dev.render_sums(cam, seed=21, sample_begin=i, sample_stride=n_dev) for i in range(n_dev)]
    m = ra.MultiContext([0] * n_dev)
    try:
        assert m.device_count() == n_dev
        m.upload(sc.desc)
        got = m.render_sums(cam, seed=21)
        # the summing kernel adds the devices' frames in list order: the same f32 additions as here
        expect = parts[0].copy()
        for p in parts[1:]:
            expect += p
        assert np.array_equal(got, expect)
        assert np.allclose(got, whole, rtol=2e-6, atol=1e-6)
        st = m.stats()
        assert st["n_devices"] == n_dev and st["paths"] == sc.width * sc.height * sc.spp
        assert all(t > 0 for t in st["ms_render"][:min(n_dev, spp)]) and st["ms_reduce"] > 0
        # an outer split composes (what a multi-node caller would pass): odd sample indices only
        odd = m.render_sums(cam, seed=21, sample_begin=1, sample_stride=2)
        assert np.allclose(odd, dev.render_sums(cam, seed=21, sample_begin=1, sample_stride=2), rtol=2e-6, atol=1e-6)
        # device-pointer form: the sum stays on the root GPU
        import torch
        d = torch.full((sc.height, sc.width, 3), -1.0, dtype=torch.float32, device="cuda")
        m.render_device(cam, d.data_ptr(), seed=21)
        torch.cuda.synchronize()
        assert np.array_equal(d.cpu().numpy(), got)
        # a second scene replaces the first
        sc2 = ra.Scene("two_spheres", 64, ASPECT, "uniform", 4, 8)
        m.upload(sc2.desc)
        dev2 = sc2.upload(ctx)
        assert np.allclose(m.render_sums(sc2.camera_copy(), seed=3), dev2.render_sums(sc2.camera_copy(), seed=3),
                           rtol=2e-6, atol=1e-6)
    finally:
        m.close()
        ctx.close()


def test_rt_multi_argument_errors():
    L = ra.lib()
    import ctypes as C
    h = C.c_void_p()
    assert L.rt_multi_init(None, 0, C.byref(h)) == ra.RT_E_INVALID_ARG
    ids = (C.c_int * 1)(999)
    assert L.rt_multi_init(ids, 1, C.byref(h)) == ra.RT_E_INVALID_ARG and b"out of range" in L.rt_multi_last_error(None)
    m = ra.MultiContext([0, 0])
    try:
        sc = ra.Scene("two_spheres", 32, ASPECT, "uniform", 4, 8)
        with pytest.raises(ra.RtError) as e:
            m.render_sums(sc.camera_copy())
        assert "no scene uploaded" in str(e.value)
        m.upload(sc.desc)
        with pytest.raises(ra.RtError):
            m.render_sums(sc.camera_copy(), sample_stride=0)
        with pytest.raises(ra.RtError):
            m.set_option("no_such_option", 1)
        m.set_option("fused_ku", 2)
        assert m.render_sums(sc.camera_copy(), seed=1).sum() > 0
    finally:
        m.close()


def test_render_split_at_world_size_one_equals_render_sums():
    """render_split on its default stream (torch's current stream): the frame must be complete when the reduce — here a
    no-op — and the copy back are issued right behind it."""
    import torch
    from raytracer_amd import multi_gpu as mg
    ctx = ra.Context(0)
    try:
        sc = ra.Scene("book_cover", 160, ASPECT, "uniform", 16, 50)
        dev, cam = sc.upload(ctx), sc.camera_copy()
        ref = dev.render_sums(cam, seed=9)
        for _ in range(3):  # on the legacy default stream (handle 0: render_split waits for the library's stream)
            d = torch.zeros((sc.height, sc.width, 3), dtype=torch.float32, device="cuda")
            out = mg.render_split(dev, cam, d, 9, 0, 1)
            assert np.array_equal(out.cpu().numpy(), ref)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):  # on a stream of torch's own: everything is stream-ordered, no host wait
            assert side.cuda_stream != 0
            for _ in range(3):
                d = torch.zeros((sc.height, sc.width, 3), dtype=torch.float32, device="cuda")
                out = mg.render_split(dev, cam, d, 9, 0, 1)
                assert np.array_equal(out.cpu().numpy(), ref)
    finally:
        ctx.close()


def _split_worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from raytracer_amd import multi_gpu as mg
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    ctx = ra.Context(0)
    sc = ra.Scene("book_cover", 96, ASPECT, "uniform", 25, 50)
    dev, cam = sc.upload(ctx), sc.camera_copy()
    d = torch.zeros((sc.height, sc.width, 3), dtype=torch.float32, device="cuda")
    out = mg.render_split(dev, cam, d, 5, rank, world, via_host=True)
    torch.cuda.synchronize()
    if rank == 0:
        np.save(out_path, out.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()


def test_two_gloo_ranks_on_one_gpu_reproduce_the_frame(tmp_path):
    """The one-process-per-GPU path end to end (renderer + partition + reduce), two ranks sharing GPU 0."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "split.npy")
    mp.spawn(_split_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    ctx = ra.Context(0)
    try:
        sc = ra.Scene("book_cover", 96, ASPECT, "uniform", 25, 50)
        dev, cam = sc.upload(ctx), sc.camera_copy()
        whole = dev.render_sums(cam, seed=5)
        halves = dev.render_sums(cam, seed=5, sample_begin=0, sample_stride=2) + \
            dev.render_sums(cam, seed=5, sample_begin=1, sample_stride=2)
        assert np.array_equal(got, halves)
        assert np.allclose(got, whole, rtol=2e-6, atol=1e-6)
    finally:
        ctx.close()
