"""Diagnostic: per-phase SIMD lane utilisation of the fused kernels (instrumented build, 'fused_stats') and
timing sweeps of the scheduler thresholds.   python scripts/fused_stats.py [spp] [--sweep]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra

ctx = ra.Context(0)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
spp = int(args[0]) if args else 100
scene_name = next((a.split("=")[1] for a in sys.argv if a.startswith("--scene=")), "book_cover")
width = int(next((a.split("=")[1] for a in sys.argv if a.startswith("--width=")), 1200))
texels = None
if scene_name in ("earth", "final_scene", "final_scene_small"):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from golden_data import load_earth_rgba
    texels = load_earth_rgba()
sc = ra.Scene(scene_name, width, 16 / 9, "uniform", spp, 50, texels=texels)
dev = sc.upload(ctx)
cam = sc.camera_copy()


def run(stats, **opts):
    ctx.set_option("fused_stats", 1 if stats else 0)
    for k, v in opts.items():
        ctx.set_option(k, v)
    dev.render_sums(cam, seed=1)
    best = 1e9
    for _ in range(1 if stats else 3):
        dev.render_sums(cam, seed=1)
        st = ctx.stats()
        best = min(best, st["ms_total"])
    out = dict(opts, ms=round(best, 3), msamples_s=round(st["paths"] / best / 1e3, 1))
    if stats:
        d = st["dbg"]
        for i, n in enumerate(["node", "leaf", "shade", "regen"]):
            lane, wave = d[2 * i], d[2 * i + 1]
            out[n] = {"per_ray": round(lane / st["rays"], 3), "trips_x64_per_ray": round(wave * 64 / st["rays"], 3),
                      "lane_util": round(lane / (64 * wave), 3) if wave else None}
        if d[12] > 0:  # cycle shares of the phases (s_memtime deltas, per wave)
            out["cycles"] = {n: round(d[8 + i] / d[12], 3) for i, n in enumerate(["node", "leaf", "shade", "regen"])}
    print(json.dumps(out), flush=True)


run(False)
if "--occ" in sys.argv:
    run(False, fused_block=512)
    run(False, fused_block=256)
    run(False, fused_block=1024, fused_lds=0)
    ctx.set_option("fused_lds", 1)
if "--stats" in sys.argv:
    run(True)
if "--ku" in sys.argv:  # work-unit size (samples of one 8x8 tile per unit)
    for ku in (0, 2, 4, 7, 10, 13, 20, 25, 34, 50, 100):
        run(False, fused_ku=ku)
