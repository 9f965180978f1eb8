"""Render a small frame of a scene and save it: python scripts/frame_dump.py SCENE WIDTH SPP OUT.npy [key=value ...]
(A/B of two library builds: frames of the same seed must be bit-identical when only code generation differs.)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import raytracer_amd as ra
name, width, spp, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
ctx = ra.Context(0)
for a in sys.argv[5:]:
    k, v = a.split("=")
    try:
        ctx.set_option(k, int(v))
    except ra.RtError:
        pass  # an older build that does not know the option
tex = None
if name in ("earth", "final_scene", "final_scene_small"):
    from golden_data import load_earth_rgba
    tex = load_earth_rgba()
sc = ra.Scene(name, width, 16 / 9, "uniform", spp, int(os.environ.get("DEPTH", "50")), texels=tex)
dev = sc.upload(ctx)
f = dev.render_sums(sc.camera_copy(), seed=5)
st = ctx.stats()
np.save(out, f)
print(name, os.environ.get("RT_LIB_GPU", "default"), "mean", float(f.mean()) / spp, "rays/path", st["rays"] / st["paths"], "max", float(f.max()))
