"""Generate tests/golden/workload_<config>.json: per-sample operation counts of the reference algorithm
(reference-topology BVH, src/hittable/containers.rs:58-152) measured by the instrumented CPU oracle, and the
algorithmic flops/sample they imply under the per-operation constants of SURVEY.md §8(d).

    python tests/make_workload_fixture.py c2 [spp_for_counting]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle_lib import OracleScene  # noqa: E402

# add/sub/mul/min/max/cmp = 1, FMA = 2, div/sqrt/transcendental = 1, integer/RNG = 0
F = dict(cam=35, ray=3, box=18, sph=23, sph_root=4, mov=10, hit=26, lam=29, met=38, die=50, iso=20, tex_solid=0,
         tex_chk=6, tex_img=6, tex_noise=50, mul=3, sky=21, quad=30, medium=20)

CONFIGS = {
    # name: (scene, width, sampler, spp of the config, depth, spp used for counting)
    "c1": ("book_cover", 400, "random", 10, 50, 10),
    "c2": ("book_cover", 1200, "uniform", 100, 50, 9),
    "c3a": ("earth", 1920, "uniform", 256, 50, 4),
    "c3b": ("something_blocky", 1920, "uniform", 256, 50, 4),
    "c4": ("cornell_smoke", 1920, "uniform", 1024, 50, 4),
    "c5": ("final_scene", 3840, "uniform", 10000, 50, 1),
}


def flops_per_sample(c):
    n = c["samples"]
    total = (F["cam"] * n + F["ray"] * c["rays"] + F["box"] * c["box_tests"] + F["sph"] * c["sphere_tests"] +
             F["sph_root"] * c["sphere_disc_pos"] + F["mov"] * c["moving_tests"] + F["quad"] * c["quad_tests"] +
             F["medium"] * c["medium_tests"] + F["hit"] * c["bounce_hits"] + F["lam"] * c["scat_lambert"] +
             F["met"] * c["scat_metal"] + F["die"] * c["scat_dielectric"] + F["iso"] * c["scat_isotropic"] +
             F["tex_solid"] * c["tex_solid"] + F["tex_chk"] * c["tex_checker"] + F["tex_img"] * c["tex_image"] +
             F["tex_noise"] * c["tex_noise"] + F["mul"] * c["bounce_hits"] + F["sky"] * c["misses"])
    return total / n


def main():
    cfg = sys.argv[1]
    scene, width, sampler, spp, depth, cspp = CONFIGS[cfg]
    if len(sys.argv) > 2:
        cspp = int(sys.argv[2])
    tex = None
    if scene in ("earth", "final_scene"):
        from golden_data import load_earth_rgba
        tex = load_earth_rgba()
    sc = OracleScene(scene, width, 16.0 / 9.0, sampler, cspp, depth, texels=tex, counting=True)
    r = sc.render(threads=0, want_counters=True)
    c = r["counters"]
    n = c["samples"]
    out = {
        "config": cfg, "scene": scene, "width": sc.width, "height": sc.height, "sampler": sampler,
        "config_spp": spp, "counted_spp": cspp, "depth": depth, "samples_counted": n,
        "per_sample": {k: v / n for k, v in c.items() if k != "samples"},
        "flop_constants": F,
        "flops_per_sample": flops_per_sample(c),
        "note": "counts from the instrumented CPU oracle traversing the reference-topology BVH; "
                "throughput-independent of spp (per-sample averages)",
    }
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"workload_{cfg}.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(path, "flops/sample =", out["flops_per_sample"])


if __name__ == "__main__":
    main()
