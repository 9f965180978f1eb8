"""One-GPU rehearsal of the strong split of C2: times one rank's share (sample_begin = 0, sample_stride = N) of the
1200x675x100-spp frame for N = 1, 2, 4, 8 — what each GPU of an N-GPU node renders per frame — and the fixed
per-frame overheads around the kernel.  Prints per-N: ms per frame, ideal = t(1)/N, efficiency."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as ra

ctx = ra.Context(0)
for k, v in [a.split("=") for a in sys.argv[1:] if "=" in a]:
    ctx.set_option(k, int(v))
sc = ra.Scene("book_cover", 1200, 16.0 / 9.0, "uniform", 100, 50)
dev = sc.upload(ctx)
cam = sc.camera_copy()
d = torch.zeros((sc.height, sc.width, 3), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
res = {}
for n in (1, 2, 4, 8):
    for _ in range(3):
        dev.render_device(cam, d.data_ptr(), seed=1, sample_begin=0, sample_stride=n, stream=st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    R = 20
    for i in range(R):
        dev.render_device(cam, d.data_ptr(), seed=1 + i, sample_begin=i % n, sample_stride=n, stream=st)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / R * 1e3
    dev.render_device(cam, d.data_ptr(), seed=1, sample_begin=0, sample_stride=n, stream=st)
    torch.cuda.synchronize()
    res[n] = {"ms": ms, "kernel_ms": ctx.stats()["ms_total"]}
for n in res:
    res[n]["efficiency"] = res[1]["ms"] / n / res[n]["ms"]
print(json.dumps(res))
st = ctx.stats()
print("tiles taken at the front / back of the list in the last frame:", st["dbg"][14], st["dbg"][15])
