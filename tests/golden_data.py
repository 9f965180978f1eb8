"""Golden data loaders (fixtures under tests/golden/)."""
import hashlib
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_earth_rgba():
    """The reference's images/earthmap.jpg (data file, 1024x512) decoded as image::open(..).to_rgba8() would
    (scene.rs:250).  Decoder: Pillow (libjpeg); the Rust `jpeg-decoder` crate may differ by +-1-2/255 per texel
    (parity unpinned at texel level — SURVEY §8c); the SHA-256 of OUR decode is pinned in earthmap.json so both
    sides of every test sample identical texels."""
    from PIL import Image
    im = Image.open(os.path.join(GOLDEN, "earthmap.jpg")).convert("RGBA")
    arr = np.ascontiguousarray(np.asarray(im, dtype=np.uint8))
    return arr


def earth_sha256():
    return hashlib.sha256(load_earth_rgba().tobytes()).hexdigest()


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)
