"""Shared by tests/make_ref_pins.py (which reads the reference's committed renders, here only) and
tests/test_reference_pins.py (which reads the derived fixture tests/golden/ref_pins.npz, anywhere).

Edge map: luminance of the gamma-corrected image, Gaussian sigma 1.5 px (the committed renders are noisy: their spp is
not recorded), Sobel gradient, non-maximum suppression along the quantised gradient direction, threshold 0.04 / px.
What survives are sphere silhouettes, checker transitions and coast lines: one-pixel-wide curves whose POSITION depends
on the sphere centres and radii (scene.rs:114-143), the camera (builder.rs:74-135), the checker's inv_scale and
parity (texture.rs:50-60) and the sphere uv mapping (hittable/geometry.rs:30-34), and not on spp or the RNG stream."""
import numpy as np
from scipy import ndimage as ndi

SIGMA, THRESHOLD, ERODE_PX, MIN_PIXELS = 1.5, 0.04, 3, 40
SCENES = {  # fixture key -> (file under /root/reference/renders, scene builder, image width of the committed render)
    "book_cover": ("week/2-checker-texture.png", 1280),   # scene.rs:89-164
    "earth": ("week/3-earth.png", 1280),                  # scene.rs:243-260
    "two_spheres": ("week/2-1-two-spheres.png", 1920),    # scene.rs:216-241
}


def edge_map(img):
    """img: HxWx3 in [0, 1] (gamma-corrected). Returns a boolean HxW map of thinned strong edges."""
    y = np.asarray(img, dtype=np.float64) @ np.array([0.2126, 0.7152, 0.0722])
    y = ndi.gaussian_filter(y, SIGMA)
    gx, gy = ndi.sobel(y, axis=1) / 8.0, ndi.sobel(y, axis=0) / 8.0
    mag = np.hypot(gx, gy)
    direction = np.round(np.arctan2(gy, gx) / (np.pi / 4)).astype(int) % 4
    keep = np.zeros(mag.shape, dtype=bool)
    for k, (dy, dx) in enumerate(((0, 1), (1, 1), (1, 0), (1, -1))):
        ahead, behind = np.roll(mag, (dy, dx), (0, 1)), np.roll(mag, (-dy, -dx), (0, 1))
        keep |= (direction == k) & (mag >= ahead) & (mag >= behind)
    keep &= mag > THRESHOLD
    keep[:3] = keep[-3:] = False
    keep[:, :3] = keep[:, -3:] = False
    return keep


def edge_distance(src, dst):
    """For every edge pixel of `src`: distance (px) to the nearest edge pixel of `dst`. Returns (mean, frac <= 1.5 px)."""
    d = ndi.distance_transform_edt(~dst)[src]
    return float(d.mean()), float((d <= 1.5).mean())


def eroded_id_map(ids):
    """ids: HxW object id of the pinhole primary hit (-1 = sky). Each object's region is eroded by ERODE_PX so that
    silhouette pixels, defocus and motion blur of the neighbours stay out; objects left with < MIN_PIXELS are dropped."""
    out = np.full(ids.shape, -1, dtype=np.int16)
    for i in np.unique(ids):
        if i < 0:
            continue
        m = ndi.binary_erosion(ids == i, iterations=ERODE_PX, border_value=1)
        if m.sum() >= MIN_PIXELS:
            out[m] = i
    return out


def object_means(img, idmap, obj_ids):
    img = np.asarray(img, dtype=np.float64)
    return np.array([img[idmap == i].mean(axis=0) for i in obj_ids])
