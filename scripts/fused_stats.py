"""Diagnostic: per-phase SIMD lane utilisation of the fused kernels (instrumented build, 'fused_stats') and
timing sweeps of the scheduler thresholds.   python scripts/fused_stats.py [spp] [--sweep]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra

ctx = ra.Context(0)
args = [a for a in sys.argv[1:] if not a.startswith("--") and "=" not in a]
for a in sys.argv[1:]:
    if "=" in a and not a.startswith("--"):
        ctx.set_option(a.split("=")[0], int(a.split("=")[1]))
spp = int(args[0]) if args else 100
scene_name = next((a.split("=")[1] for a in sys.argv if a.startswith("--scene=")), "book_cover")
width = int(next((a.split("=")[1] for a in sys.argv if a.startswith("--width=")), 1200))
texels = None
if scene_name in ("earth", "final_scene", "final_scene_small"):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from golden_data import load_earth_rgba
    texels = load_earth_rgba()
sampler = next((a.split("=")[1] for a in sys.argv if a.startswith("--sampler=")), "uniform")
sc = ra.Scene(scene_name, width, 16 / 9, sampler, spp, 50, texels=texels)
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
            out["cycles"]["wait_for_work"] = round(d[14] / d[12], 3)
            out["mean_exhaustion_age_over_max_lifetime"] = round(d[15] / int(next((a.split("=")[1] for a in sys.argv if a.startswith("--waves=")), 4096)) / d[13], 3)
            # mean wave lifetime over the longest one: what the frame's tail costs
            n_waves = int(next((a.split("=")[1] for a in sys.argv if a.startswith("--waves=")), 4096))  # workgroups x waves (C5: 256 x 12)
            out["mean_over_max_lifetime"] = round(d[12] / n_waves / d[13], 3)
    print(json.dumps(out), flush=True)


run(False)
if "--occ" in sys.argv:
    run(False, fused_lds=0)
    ctx.set_option("fused_lds", 1)
if "--stats" in sys.argv:
    run(True)
if "--ku" in sys.argv:  # work-unit size (samples of one 8x8 tile per unit)
    for ku in (0, 2, 4, 7, 10, 13, 20, 25, 34, 50, 100):
        run(False, fused_ku=ku)
if "--resume" in sys.argv:  # RESUME variant of k_fused: hand-back (kb) and node-loop exit (kn) thresholds
    for kb, kn in ((1, 1), (16, 1), (8, 8), (16, 8), (24, 8), (16, 4), (12, 4), (16, 12), (24, 16), (32, 16)):
        run(False, fused_resume=1, fused_kb=kb, fused_kn=kn)
    run(True, fused_resume=1, fused_kb=16, fused_kn=8)
    run(False, fused_resume=0)
if "--kr" in sys.argv:
    for res, kr in ((0, 1), (0, 4), (0, 8), (0, 16), (1, 1), (1, 4), (1, 8), (1, 12), (1, 16), (1, 24)):
        run(False, fused_resume=res, fused_kb=16, fused_kn=8, fused_kr=kr)
    run(True, fused_resume=1, fused_kb=16, fused_kn=8, fused_kr=8)
if "--grid" in sys.argv:
    for kb in (12, 16, 24, 32):
        for kn in (4, 8, 12):
            for kr in (16, 24, 32, 40, 48):
                run(False, fused_resume=1, fused_kb=kb, fused_kn=kn, fused_kr=kr)
    run(False, fused_resume=0, fused_kr=1)
if "--ab" in sys.argv:  # RESUME on/off on the chosen scene
    for res in (0, 1, 0, 1):
        run(False, fused_resume=res)
if "--grid5" in sys.argv:
    for kb in (8, 16, 24, 32):
        for kn in (4, 8, 16):
            for kr in (8, 16, 32):
                run(False, fused_resume=1, fused_kb=kb, fused_kn=kn, fused_kr=kr)
    run(False, fused_resume=0, fused_kr=1)
    run(False, fused_resume=0, fused_kr=16)
if "--lds" in sys.argv:  # LDS-resident vs HBM-resident instance on the chosen scene
    for v in (1, 0, 1, 0):
        run(False, fused_lds=v)
if "--wide" in sys.argv:  # HBM-resident instances: four-wide tree vs binary tree
    for v in (1, 0, 1, 0):
        run(False, fused_wide=v)
    run(True, fused_wide=1)
if "--half" in sys.argv:  # HBM-resident instances: binary16 planes + LDS cache of the top vs f32 four-wide tree
    for h in (1, 0, 1, 0):
        run(False, fused_half=h)
    run(True, fused_half=1)
if "--partner" in sys.argv:  # LDS-resident spheres instance with / without its 256-thread partner workgroups
    for v in (1, 0, 1, 0, 1, 0):
        run(False, fused_partner=v)
if "--tail" in sys.argv:  # heights (per mille of the frame's rows) of the 4x4 / 2x2 bands at the end of the tile list
    for t4, t2 in ((50, 30), (0, 0), (50, 0), (100, 30), (25, 15), (50, 60), (50, 30), (0, 0)):
        run(False, fused_tail4=t4, fused_tail2=t2)
if "--ku5" in sys.argv:
    for ku in (0, 8, 4, 2, 0, 8, 4):
        run(False, fused_ku=ku)
