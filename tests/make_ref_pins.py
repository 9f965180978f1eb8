"""Builds tests/golden/ref_pins.npz from the three renders the reference's authors committed of scenes HEAD still
builds identically (see make_ref_thumbs.py for why only these three), at their NATIVE resolution:

  <scene>_edges / _shape    thinned edge map of the committed image (ref_pins.edge_map), bit-packed
  <scene>_idmap             which object the pinhole ray through each pixel centre hits (oracle, time 0.5), each region
                            eroded by 3 px; -1 elsewhere.  It only says WHERE to average, the colours below are the
                            reference's own pixels
  <scene>_obj_ids / _obj_rgb  mean colour of the committed image inside each object's eroded region

Run here only (/root/reference does not exist on the GPU box):   python tests/make_ref_pins.py"""
import os

import numpy as np
from PIL import Image

from oracle_lib import OracleScene
from ref_pins import SCENES, edge_map, eroded_id_map, object_means

REF = "/root/reference/renders"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_pins.npz")

if __name__ == "__main__":
    out = {}
    for name, (path, width) in SCENES.items():
        ref = np.asarray(Image.open(os.path.join(REF, path)).convert("RGB"), dtype=np.float64) / 255.0
        sc = OracleScene(name, width, 16.0 / 9.0, "uniform", 1, 50)
        assert ref.shape == (sc.height, sc.width, 3), (name, ref.shape)
        e = edge_map(ref)
        ids = sc.intersect(sc.primary_rays(1, 0, 0, 0.5))["object_id"].reshape(sc.height, sc.width)
        idmap = eroded_id_map(ids)
        obj = np.array([i for i in np.unique(idmap) if i >= 0], dtype=np.int32)
        out[name + "_edges"] = np.packbits(e)
        out[name + "_shape"] = np.array(e.shape, dtype=np.int32)
        out[name + "_idmap"] = idmap
        out[name + "_obj_ids"] = obj
        out[name + "_obj_rgb"] = object_means(ref, idmap, obj).astype(np.float32)
        print(name, e.shape, "edge pixels", int(e.sum()), "objects", len(obj))
    np.savez_compressed(OUT, **out)
    print(OUT, os.path.getsize(OUT) >> 10, "KiB")
