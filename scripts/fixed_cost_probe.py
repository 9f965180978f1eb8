"""Fixed cost of a fused frame: kernel time (library hipEvents) of book_cover at 1 sample per pixel for growing image
widths, LDS-resident and HBM-resident instances.   python scripts/fixed_cost_probe.py [key=value ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as ra

ctx = ra.Context(0)
for a in sys.argv[1:]:
    if "=" in a and not a.startswith("--"):
        k, v = a.split("=")
        ctx.set_option(k, int(v))
st = torch.cuda.current_stream().cuda_stream
depth = int(next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("--depth=")), 50))
spp = int(next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("--spp=")), 1))
for width in (16, 64, 200, 400, 800, 1200, 1920):
    sc = ra.Scene("book_cover", width, 16.0 / 9.0, "random", spp, depth)
    dev = sc.upload(ctx)
    cam = sc.camera_copy()
    d = torch.zeros((sc.height, sc.width, 3), dtype=torch.float32, device="cuda")
    best = 1e9
    for i in range(6):
        dev.render_device(cam, d.data_ptr(), seed=i, stream=st)
        torch.cuda.synchronize()
        best = min(best, ctx.stats()["ms_total"])
    print(f"depth {depth} spp {spp} width {width:5d}  pixels {sc.width * sc.height:8d}  kernel {best * 1e3:8.1f} us", flush=True)
    dev.close()
