"""Diagnostic: average wave lifetime against the kernel's duration (instrumented build) — the difference is the
frame's tail, during which waves that have run out of work leave their SIMDs idle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as ra
ctx = ra.Context(0)
sc = ra.Scene("book_cover", 1200, 16 / 9, "uniform", 100, 50)
dev = sc.upload(ctx)
cam = sc.camera_copy()
ctx.set_option("fused_stats", 1)
for _ in range(3):
    dev.render_sums(cam, seed=1)
    st = ctx.stats()
    waves = 256 * 16
    ticks_per_wave = st["dbg"][12] / waves  # sum over lanes of cycles / 64
    print(f"kernel {st['ms_total']:.3f} ms; mean wave lifetime {ticks_per_wave:.0f} ticks, longest {st['dbg'][13]:.0f} "
          f"(mean / longest = {ticks_per_wave / st['dbg'][13]:.3f})", flush=True)
