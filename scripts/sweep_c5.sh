#!/bin/bash
export RT_LIB_GPU=ab/librt_med1.so
OUT=gpurun_out/sweep_c5b.log; rm -f $OUT
for kr in 6 8 12 16; do for kb in 12 16 20; do
  echo "== kr=$kr kb=$kb" >> $OUT
  timeout -k 10 100 python scripts/frame_probe.py --workload=c5 --frames=8 fused_kr=$kr fused_kb=$kb 2>/dev/null >> $OUT || exit 1
done; done
grep -o "^== .*\|\"ms\": [0-9.]*" $OUT | paste - -
