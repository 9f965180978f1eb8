"""Diagnostic: how much of a frame's time is its tail?  Renders 2N frames of C2 either back to back on one context
(each kernel starts when the previous one has drained) or alternating between two contexts on two streams (the next
frame's workgroups move in as the previous frame's retire).   python scripts/overlap_probe.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as ra

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
STRIDE = int(sys.argv[2]) if len(sys.argv) > 2 else 1   # sample stride: 8 = the share of one GPU of eight
ctxs = [ra.Context(0), ra.Context(0)]
host = ra.Scene("book_cover", 1200, 16 / 9, "uniform", 100, 50)
devs = [host.upload(c) for c in ctxs]
cam = host.camera_copy()
bufs = [torch.zeros((host.height, host.width, 3), dtype=torch.float32, device="cuda") for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def run(two):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(2 * N):
        k = i & 1 if two else 0
        devs[k].render_device(cam, bufs[k].data_ptr(), seed=100 + i, sample_begin=0, sample_stride=STRIDE, stream=streams[k].cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (2 * N) * 1e3


for _ in range(2):
    run(False); run(True)
for rep in range(3):
    print(f"one context: {run(False):.3f} ms/frame   two contexts alternating: {run(True):.3f} ms/frame", flush=True)
