"""C2 frame timing (1200x675x100 spp) under rt_set_option overrides given as key=value arguments; prints ms per frame
(wall, 20 frames back to back), the library's hipEvent time of the last frame and the tiles taken at the two ends of
the tile list.   python scripts/frame_probe.py [key=value ...] [--stride=N] [--frames=R]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as ra

ctx = ra.Context(0)
opts = {}
stride, frames = 1, 20
for a in sys.argv[1:]:
    if a.startswith("--stride="):
        stride = int(a.split("=")[1])
    elif a.startswith("--frames="):
        frames = int(a.split("=")[1])
    elif "=" in a:
        k, v = a.split("=")
        ctx.set_option(k, int(v))
        opts[k] = int(v)
sc = ra.Scene("book_cover", 1200, 16.0 / 9.0, "uniform", 100, 50)
dev = sc.upload(ctx)
cam = sc.camera_copy()
d = torch.zeros((sc.height, sc.width, 3), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    dev.render_device(cam, d.data_ptr(), seed=1, sample_begin=0, sample_stride=stride, stream=st)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(frames):
    dev.render_device(cam, d.data_ptr(), seed=1 + i, sample_begin=i % stride, sample_stride=stride, stream=st)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / frames * 1e3
s = ctx.stats()
print(json.dumps({"opts": opts, "stride": stride, "ms": round(ms, 3), "kernel_ms": round(s["ms_total"], 3),
                  "front": s["dbg"][14], "back": s["dbg"][15]}))
