"""Repeatability of the production kernels inside one context: the same frame rendered N times (and after other scenes
and instrumented frames have run) must be bit-identical with identical ray counts."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, raytracer_amd as ra
from golden_data import load_earth_rgba
ctx = ra.Context(0)
def scene(name, w, spp):
    g = ra.Scene(name, w, 16/9, "uniform", spp, 50, texels=load_earth_rgba() if name.startswith("final") else None)
    return g, g.upload(ctx), g.camera_copy()
ok = True
for name, w, spp in (("final_scene", 480, 16), ("final_scene", 240, 9), ("book_cover", 320, 16), ("final_scene", 960, 4)):
    g, dev, cam = scene(name, w, spp)
    ref = dev.render_sums(cam, seed=3); r0 = ctx.stats()["rays"]
    for i in range(8):
        if i == 4:
            ctx.set_option("fused_stats", 1); dev.render_sums(cam, seed=3); ctx.set_option("fused_stats", 0)
        f = dev.render_sums(cam, seed=3); r = ctx.stats()["rays"]
        if not np.array_equal(f, ref) or r != r0:
            ok = False
            print("MISMATCH", name, w, i, r, r0, float((f != ref).mean()))
    print(name, w, spp, "rays", r0, "repeatable" if ok else "NOT repeatable")
    dev.close()
print("ALL OK" if ok else "FAILED")
