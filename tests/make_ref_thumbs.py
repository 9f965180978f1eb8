"""Builds tests/golden/ref_thumbs.npz from the images the reference's authors committed under renders/ (their own
output of the scenes on this path; spp / depth of those runs are not recorded, the scene seeds are fixed in
scene.rs).  Each image is box-filtered to a 160-wide thumbnail (uint8, gamma-corrected as stored).  Run here only:
/root/reference does not exist on the GPU box; the tests read the committed .npz.

    python tests/make_ref_thumbs.py
"""
import os

import numpy as np
from PIL import Image

REF = "/root/reference/renders"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_thumbs.npz")
# thumbnail key -> (file, scene builder it shows at HEAD / at the commit that produced it)
SOURCES = {
    "book_cover": "week/2-checker-texture.png",           # scene.rs:89-164 (checker ground, bouncing spheres)
    "earth": "week/3-earth.png",                          # scene.rs:243-260
    "two_spheres": "week/2-1-two-spheres.png",            # scene.rs:216-241
}
# Not usable: week/3-2-hermite-cubic-noise.png, week/4-blocky-noise.png and weekend/fixed-seed/1-*-composition*.png
# were rendered from earlier revisions of their scene builders / cameras (correlation with HEAD's scenes 0.4-0.6).


def thumb(path, w=160):
    im = Image.open(path).convert("RGB")
    h = int(round(im.size[1] * w / im.size[0]))
    return np.asarray(im.resize((w, h), Image.BOX), dtype=np.uint8)


if __name__ == "__main__":
    arrs = {k: thumb(os.path.join(REF, v)) for k, v in SOURCES.items()}
    np.savez_compressed(OUT, **arrs)
    for k, a in arrs.items():
        print(k, a.shape, a.mean(axis=(0, 1)).round(1))
