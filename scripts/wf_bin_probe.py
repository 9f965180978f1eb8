"""Wavefront mode (option mode = 0) with and without material binning (option wf_bin): frame time, per-stage times
(option profile) and whether the frame changes.   python scripts/wf_bin_probe.py [scene] [width] [spp]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as ra

name = sys.argv[1] if len(sys.argv) > 1 else "book_cover"
width = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 16
ctx = ra.Context(0)
sc = ra.Scene(name, width, 16 / 9, "uniform", spp, 50)
dev, cam = sc.upload(ctx), sc.camera_copy()
ctx.set_option("mode", 0)
frames = {}
for wb in (0, 1):
    ctx.set_option("wf_bin", wb)
    ctx.set_option("profile", 0)
    dev.render_sums(cam, seed=3)
    f = dev.render_sums(cam, seed=3)
    ms = ctx.stats()["ms_total"]
    ctx.set_option("profile", 1)
    dev.render_sums(cam, seed=3)
    st = ctx.stats()
    frames[wb] = f
    print(json.dumps({"wf_bin": wb, "ms": round(ms, 2), "stage_ms": {"intersect": round(st["ms_intersect"], 2),
          "bin": round(st["dbg"][11], 2), "shade": round(st["ms_shade"], 2), "generate": round(st["ms_generate"], 2),
          "accumulate": round(st["ms_accumulate"], 2)}, "rays": st["rays"]}), flush=True)
print("frames identical:", bool(np.array_equal(frames[0], frames[1])))
