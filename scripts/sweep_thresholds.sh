#!/bin/bash
OUT=gpurun_out/sweep_asm2.log; rm -f $OUT
for kn in 4 8 12 16; do for kb in 12 16 24; do
  echo "== kn=$kn kb=$kb" >> $OUT
  RT_LIB_GPU=ab/librt_asm2.so timeout -k 10 100 python scripts/frame_probe.py fused_kn=$kn fused_kb=$kb 2>/dev/null >> $OUT || exit 1
done; done
for kr in 8 24 32; do
  echo "== kr=$kr" >> $OUT
  RT_LIB_GPU=ab/librt_asm2.so timeout -k 10 100 python scripts/frame_probe.py fused_kr=$kr 2>/dev/null >> $OUT || exit 1
done
