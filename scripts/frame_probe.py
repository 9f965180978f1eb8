"""Frame timing of a bench workload (--workload=c2 default: 1200x675x100 spp) under rt_set_option overrides given as key=value arguments; prints ms per frame
(wall, 20 frames back to back), the library's hipEvent time of the last frame and the tiles taken at the two ends of
the tile list.   python scripts/frame_probe.py [key=value ...] [--stride=N] [--frames=R]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as ra

ctx = ra.Context(0)
opts = {}
stride, frames, batch = 1, 20, 0
for a in sys.argv[1:]:
    if a.startswith("--stride="):
        stride = int(a.split("=")[1])
    elif a.startswith("--frames="):
        frames = int(a.split("=")[1])
    elif a.startswith("--batch="):
        batch = int(a.split("=")[1])  # frames per rt_render_batch_device launch (0: one rt_render_device launch per frame)
    elif a.startswith("--workload="):
        pass
    elif "=" in a:
        k, v = a.split("=")
        ctx.set_option(k, int(v))
        opts[k] = int(v)
WL = {"c2": ("book_cover", 1200, "uniform", 100), "c5": ("final_scene", 3840, "uniform", 16), "c4": ("cornell_smoke", 1920, "uniform", 64),
      "c3a": ("earth", 1920, "uniform", 256), "c3b": ("something_blocky", 1920, "uniform", 256), "c1": ("book_cover", 400, "random", 10)}
wl = next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("--workload=")), "c2")
name, width, sampler, spp = WL[wl]
texels = None
if name in ("earth", "final_scene"):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from golden_data import load_earth_rgba
    texels = load_earth_rgba()
sc = ra.Scene(name, width, 16.0 / 9.0, sampler, spp, 50, texels=texels)
dev = sc.upload(ctx)
cam = sc.camera_copy()
d = torch.zeros((sc.height, sc.width, 3), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    dev.render_device(cam, d.data_ptr(), seed=1, sample_begin=0, sample_stride=stride, stream=st)
torch.cuda.synchronize()
end = (spp // stride) * stride  # (every frame of a batch must select the same number of sample indices)
if batch:
    outs = [torch.zeros_like(d) for _ in range(batch)]
    ptrs = [o.data_ptr() for o in outs]
    ra.render_batch_device(dev, cam, ptrs, list(range(batch)), [i % stride for i in range(batch)], sample_end=end, sample_stride=stride, stream=st)
    torch.cuda.synchronize()
t0 = time.perf_counter()
if batch:
    for b0 in range(0, frames, batch):
        ra.render_batch_device(dev, cam, ptrs, [1 + b0 + i for i in range(batch)], [(b0 + i) % stride for i in range(batch)],
                               sample_end=end, sample_stride=stride, stream=st)
    frames = (frames + batch - 1) // batch * batch
else:
    for i in range(frames):
        dev.render_device(cam, d.data_ptr(), seed=1 + i, sample_begin=i % stride, sample_end=end, sample_stride=stride, stream=st)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / frames * 1e3
s = ctx.stats()
print(json.dumps({"workload": wl, "opts": opts, "stride": stride, "batch": batch, "ms": round(ms, 3), "kernel_ms": round(s["ms_total"], 3),
                  "front": s["dbg"][14], "back": s["dbg"][15]}))
