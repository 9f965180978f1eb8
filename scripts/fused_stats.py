"""Diagnostic: per-phase SIMD lane utilisation of the fused kernel (instrumented build, 'fused_stats')."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra

ctx = ra.Context(0)
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 100
sc = ra.Scene("book_cover", 1200, 16 / 9, "uniform", spp, 50)
dev = sc.upload(ctx)
cam = sc.camera_copy()
ctx.set_option("fused_stats", 1)
for ku in ([int(a) for a in sys.argv[2:]] or [0]):
    ctx.set_option("fused_ku", ku)
    dev.render_sums(cam, seed=1)
    st = ctx.stats()
    d = st["dbg"]
    names = ["node", "leaf", "shade", "regen"]
    out = {"ku": ku, "ms": st["ms_total"], "rays": st["rays"], "paths": st["paths"]}
    for i, n in enumerate(names):
        lane, wave = d[2 * i], d[2 * i + 1]
        out[n] = {"per_ray": lane / st["rays"], "wave_trips_per_ray_x64": wave * 64 / st["rays"],
                  "lane_util": lane / (64 * wave) if wave else None}
    tot = d[12]
    out["cycles_share"] = {"node": d[8] / tot, "leaf": d[9] / tot, "shade": d[10] / tot, "regen": d[11] / tot}
    out["wave_cycles_per_trip"] = {"node": d[8] / d[1], "leaf": d[9] / d[3], "shade": d[10] / d[5], "regen": d[11] / d[7]}
    print(json.dumps(out))
