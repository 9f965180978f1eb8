#!/bin/bash
# The earth frame at C3's full size (1920 x 1080 x 256 spp: 530 M texel lookups) with the checked f32 texel guess and with
# the f64 formulas (option uv_tables = 0): scripts/uv_full_compare.sh LIB -> equal / differing samples
L=$1
mkdir -p gpurun_out/uvc
RT_LIB_GPU=ab/librt_$L.so timeout -k 10 200 python scripts/frame_dump.py earth 1920 256 gpurun_out/uvc/a.npy uv_tables=1 || exit 1
RT_LIB_GPU=ab/librt_$L.so timeout -k 10 200 python scripts/frame_dump.py earth 1920 256 gpurun_out/uvc/b.npy uv_tables=0 || exit 1
RT_LIB_GPU=ab/librt_$L.so timeout -k 10 200 python scripts/frame_dump.py final_scene 800 64 gpurun_out/uvc/c.npy uv_tables=1 || exit 1
RT_LIB_GPU=ab/librt_$L.so timeout -k 10 200 python scripts/frame_dump.py final_scene 800 64 gpurun_out/uvc/d.npy uv_tables=0 || exit 1
python - <<'PY'
import numpy as np
for x, y, n in (("a", "b", "earth 1920x1080x256"), ("c", "d", "final_scene 800x450x64")):
    a = np.load(f"gpurun_out/uvc/{x}.npy"); b = np.load(f"gpurun_out/uvc/{y}.npy")
    print(n, "equal" if np.array_equal(a, b) else f"DIFFERS in {int((a != b).sum())} of {a.size} values")
PY
rm -rf gpurun_out/uvc
