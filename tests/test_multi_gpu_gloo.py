"""world_size-2 gloo test of the multi-GPU decomposition (raytracer_amd/multi_gpu.py): residue classes of the
sample index cover every sample exactly once, and the reduce of the ranks' un-normalised partial frames equals
the single-process frame.  The per-sample contribution is a deterministic stand-in function of (pixel, sample):
the decomposition logic is what is under test, not the renderer (that is the GPU parity suite)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracer_amd import multi_gpu as mg

W, H, SPP = 24, 10, 37


def contribution(s):
    """stand-in radiance of sample s for every pixel: (H, W, 3) f32, exactly representable => order independent"""
    yy, xx = np.mgrid[0:H, 0:W]
    base = ((xx * 7 + yy * 13 + s * 3) % 16).astype(np.float32) / 16.0
    return np.stack([base, base * 0.5, 1.0 - base], axis=2).astype(np.float32)


def partial_frame(rank, world):
    b, e, st = mg.sample_slice(rank, world, SPP)
    f = np.zeros((H, W, 3), dtype=np.float32)
    for s in range(b, e, st):
        f += contribution(s)
    return f, mg.samples_in_slice(b, e, st)


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    f, n = partial_frame(rank, world)
    t = torch.from_numpy(f.copy())
    counts = torch.tensor([n], dtype=torch.int64)
    dist.all_reduce(counts)
    mg.reduce_partial_frames(t, dst=0)
    dist.barrier()
    if rank == 0:
        np.save(out_path, np.concatenate([t.numpy().ravel(), [float(counts.item())]]))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_sample_partition_covers_every_sample_once():
    for world in (1, 2, 3, 4, 8):
        seen = []
        for r in range(world):
            b, e, st = mg.sample_slice(r, world, SPP)
            seen += list(range(b, e, st))
            assert mg.samples_in_slice(b, e, st) == len(range(b, e, st))
        assert sorted(seen) == list(range(SPP))


def test_nested_partition_composes():
    """A node-level split (rt_multi_render: begin, stride given by an outer split across nodes) inside an outer split:
    every sample index is still rendered exactly once."""
    for outer in (1, 2, 3):
        for inner in (1, 2, 4, 8):
            seen = []
            for o in range(outer):
                ob, oe, ost = mg.sample_slice(o, outer, SPP)
                for i in range(inner):
                    b, e, st = mg.sample_slice(i, inner, oe, begin=ob, stride=ost)
                    seen += list(range(b, e, st))
            assert sorted(seen) == list(range(SPP)), (outer, inner)
    # more ranks than samples: the surplus ranks get empty ranges, nothing is rendered twice
    seen = []
    for r in range(8):
        b, e, st = mg.sample_slice(r, 8, 5)
        seen += list(range(b, e, st))
    assert sorted(seen) == list(range(5))


def test_two_rank_gloo_reduce_equals_single_process(tmp_path):
    out = str(tmp_path / "sum.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    whole, n = partial_frame(0, 1)
    assert int(got[-1]) == SPP == n
    assert np.array_equal(got[:-1].reshape(H, W, 3), whole)
    g = mg.resolve_reference(torch.from_numpy(whole), SPP).numpy()
    assert g.min() >= 0 and g.max() <= 1.0 + 1e-12


def test_reduce_is_noop_without_process_group():
    t = torch.ones(3)
    assert mg.reduce_partial_frames(t) is t
