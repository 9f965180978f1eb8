"""Generate the committed golden fixtures from the CPU oracle (run in the authoring container):

    python tests/make_golden.py

* tests/golden/hits_<scene>.npz   primary rays (64 px wide frame, time 0.5) + the oracle's closest hits
* tests/golden/image_<scene>.npz  converged oracle frames (48 px wide, linear f32) with spp / depth metadata
The reference itself cannot run here (Rust, incomplete tree); these pin the ORACLE so that a later change to it
cannot silently move the parity target, and let the GPU suite check against data that does not depend on the
oracle binary of the day."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from golden_data import GOLDEN, load_earth_rgba  # noqa: E402
from oracle_lib import OracleScene  # noqa: E402

SCENES = {
    # name: (aspect, image spp, depth)
    "book_cover": (16 / 9, 1024, 50), "composition": (16 / 9, 1024, 16), "two_spheres": (16 / 9, 1024, 50),
    "earth": (16 / 9, 256, 50), "something_blocky": (16 / 9, 256, 50),
    "cornell_box": (1.0, 1024, 20), "cornell_smoke": (1.0, 1024, 20), "final_scene_small": (1.0, 576, 20),
}


def tex_for(name):
    return load_earth_rgba() if name in ("earth", "final_scene_small", "final_scene") else None


def main():
    for name, (aspect, spp, depth) in SCENES.items():
        o = OracleScene(name, 64, aspect, "uniform", 1, depth, texels=tex_for(name))
        rays = o.primary_rays(time=0.5)
        hits = o.intersect(rays)
        np.savez_compressed(os.path.join(GOLDEN, f"hits_{name}.npz"), rays=rays, hits=hits, width=o.width,
                            height=o.height)
        o2 = OracleScene(name, 48, aspect, "uniform", spp, depth, texels=tex_for(name))
        r = o2.render(threads=0)
        np.savez_compressed(os.path.join(GOLDEN, f"image_{name}.npz"), linear=r["linear"].astype(np.float32),
                            spp=spp, depth=depth, width=o2.width, height=o2.height, aspect=aspect)
        print(name, o.width, o.height, "hits", int((hits["object_id"] >= 0).sum()), "image mean", r["linear"].mean())


if __name__ == "__main__":
    main()
