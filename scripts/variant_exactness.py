"""Diagnostic: do the kernel instances (LDS-resident, HBM-resident tree forms, with / without partner workgroups,
both traversal schedules) produce bit-identical frames?   python scripts/variant_exactness.py [scene] [width] [spp]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as ra

name = sys.argv[1] if len(sys.argv) > 1 else "book_cover"
width = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 100
texels = None
if name in ("earth", "final_scene", "final_scene_small"):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from golden_data import load_earth_rgba
    texels = load_earth_rgba()
ctx = ra.Context(0)
sc = ra.Scene(name, width, 16 / 9, "uniform", spp, 50, texels=texels)
dev = sc.upload(ctx)
cam = sc.camera_copy()
defaults = dict(fused_partner=1, fused_lds=1, fused_half=1, fused_wide=1, fused_resume=2)
base = None
for label, opts in [("default", {}), ("again", {}), ("no partner", dict(fused_partner=0)),
                    ("hbm half", dict(fused_lds=0)), ("hbm f32 wide", dict(fused_lds=0, fused_half=0)),
                    ("hbm binary", dict(fused_lds=0, fused_wide=0)), ("bounce-sync", dict(fused_resume=0, fused_partner=0))]:
    for k, v in {**defaults, **opts}.items():
        ctx.set_option(k, v)
    f = dev.render_sums(cam, seed=3)
    st = ctx.stats()
    if base is None:
        base = f
    diff = (f != base).any(axis=2)
    print(f"{label:14s} ms {st['ms_total']:8.3f} rays {st['rays']} differing pixels {int(diff.sum())} "
          f"max rel {float(np.abs(f.astype(np.float64) - base).max() / max(base.max(), 1e-30)):.3e}", flush=True)
