"""Multi-GPU decomposition of Camera::render: one process per GPU, sample-index partition, one RCCL reduce.

The reference sums independent samples per pixel (src/camera/mod.rs:151-172), so the per-pixel sample indices
can be split into residue classes: rank r of n renders {s : s % n == r} for the WHOLE frame (perfect load balance:
every GPU sees every pixel), and the un-normalised partial frames are summed on rank 0 with a single
`torch.distributed.reduce` (RCCL over xGMI on GPUs, gloo in the CPU tests).  No other collective is needed.
"""
import torch
import torch.distributed as dist


def sample_slice(rank, world_size, spp):
    """(begin, end, stride) of the sample indices rank `rank` renders."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    return rank, spp, world_size


def samples_in_slice(begin, end, stride):
    return 0 if begin >= end else (end - begin + stride - 1) // stride


def reduce_partial_frames(partial, dst=0):
    """Sum the ranks' un-normalised partial frames onto `dst` (in place on dst).  No-op without a process group."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(partial, dst=dst, op=dist.ReduceOp.SUM)
    return partial


def resolve_reference(total_sum, spp_total, gamma=2.2):
    """Host-side mean + gamma (color.rs:58-61) used by the CPU tests; the GPU path uses rt_resolve_device."""
    return (total_sum.double() / float(spp_total)).clamp_min(0).pow(1.0 / gamma)


def render_split(dev_scene, cam, d_sum, seed, rank, world_size, stream=None):
    """Strong split of one frame's samples across ranks: every rank renders its residue class, then reduce."""
    begin, end, stride = sample_slice(rank, world_size, cam.sampler_n ** 2 if cam.sampler_kind == 0 else cam.sampler_n)
    dev_scene.render_device(cam, d_sum.data_ptr(), seed=seed, sample_begin=begin, sample_end=end, sample_stride=stride,
                            stream=stream)
    return reduce_partial_frames(d_sum, dst=0)
