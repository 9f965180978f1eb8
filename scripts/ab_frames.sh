#!/bin/bash
# scripts/ab_frames.sh REF "lib1 lib2 ..." [key=value ...]: small frames of every scene type under ab/librt_REF.so and under
# each other build; codegen-only changes must give BIT-IDENTICAL frames (prints equal / differing per scene and build).
REF=$1; LIBS=$2; shift 2
SCENES="book_cover:200:16 cornell_box:160:16 cornell_smoke:160:16 final_scene_small:160:16 final_scene:240:9 earth:160:16 something_blocky:160:16 composition:160:16"
mkdir -p gpurun_out/abf
for L in $REF $LIBS; do
  for S in $SCENES; do
    IFS=: read N W P <<< "$S"
    RT_LIB_GPU=ab/librt_$L.so timeout -k 10 120 python scripts/frame_dump.py $N $W $P gpurun_out/abf/${L}_$N.npy "$@" > /dev/null 2>&1 || { echo "FAILED $L $N"; exit 1; }
  done
done
python - "$REF" "$LIBS" <<'PY'
import sys, numpy as np
ref, libs = sys.argv[1], sys.argv[2].split()
for L in libs:
    out = []
    for n in ("book_cover", "cornell_box", "cornell_smoke", "final_scene_small", "final_scene", "earth", "something_blocky", "composition"):
        a = np.load(f"gpurun_out/abf/{ref}_{n}.npy"); b = np.load(f"gpurun_out/abf/{L}_{n}.npy")
        out.append(f"{n}: " + ("equal" if np.array_equal(a, b) else f"DIFFERS {float((a != b).mean()):.4f} mean {a.mean():.5f} vs {b.mean():.5f}"))
    print(L, "vs", ref, "|", "; ".join(out))
PY
rm -rf gpurun_out/abf
