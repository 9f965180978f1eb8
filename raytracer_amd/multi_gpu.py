"""Multi-GPU decomposition of Camera::render, one process per GPU: sample-index partition + one RCCL reduce.

The reference sums independent samples per pixel (src/camera/mod.rs:151-172), so the per-pixel sample indices can be
split into residue classes: rank r of n renders {s : s % n == r} for the WHOLE frame (perfect load balance: every GPU
sees every pixel), and the un-normalised partial frames are summed on rank 0 with a single `torch.distributed.reduce`
(RCCL over xGMI on GPUs, gloo in the CPU tests).  No other collective is needed.  This is the partition rt_multi_render
(include/rt_abi.h) applies inside one process; `sample_slice` is its Python statement and the tests hold the two equal.
"""
import torch
import torch.distributed as dist


def sample_slice(rank, world_size, spp, begin=0, stride=1):
    """(begin, end, stride) of the sample indices rank `rank` of `world_size` renders out of
    {begin + k * stride < spp}: begin + (rank + k * world_size) * stride — rt_multi.hip's per-device range."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    if stride < 1:
        raise ValueError("stride must be >= 1")
    return min(begin + rank * stride, spp), spp, stride * world_size


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


def render_split(dev_scene, cam, d_sum, seed, rank, world_size, stream=None, via_host=False):
    """Strong split of one frame's samples across ranks: every rank renders its residue class into `d_sum` (a CUDA
    tensor), then the partial frames are summed onto rank 0.

    The render is issued on torch's CURRENT stream unless `stream` says otherwise — the stream the reduce is ordered
    against (the library's own stream is non-blocking: a reduce issued beside it would sum an unfinished frame); when
    that is the legacy default stream (handle 0, which the ABI cannot name) the call waits for the render instead.
    `via_host` stages the reduce through host memory (gloo rehearsals on one GPU)."""
    spp = cam.sampler_n ** 2 if cam.sampler_kind == 0 else cam.sampler_n
    begin, end, stride = sample_slice(rank, world_size, spp)
    if stream is None:
        stream = torch.cuda.current_stream().cuda_stream
    dev_scene.render_device(cam, d_sum.data_ptr(), seed=seed, sample_begin=begin, sample_end=end, sample_stride=stride,
                            stream=stream)
    if not stream:
        # Handle 0 is HIP's legacy default stream, and the ABI reads a NULL stream as "the context's own stream": the
        # render then ran beside torch's stream, not on it.  Order the two by waiting for the render here.  (Callers
        # that want no host wait run under `with torch.cuda.stream(torch.cuda.Stream())`, as bench.py does.)
        dev_scene.ctx.synchronize()
    if via_host:
        h = d_sum.cpu()  # (synchronises with the current stream)
        reduce_partial_frames(h, dst=0)
        d_sum.copy_(h)
        return d_sum
    return reduce_partial_frames(d_sum, dst=0)
