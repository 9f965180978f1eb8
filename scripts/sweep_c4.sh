#!/bin/bash
OUT=gpurun_out/sweep_c4.log; rm -f $OUT
for kr in 1 2 4 8 12 16 24; do
  echo "== kr_sync=$kr" >> $OUT
  timeout -k 10 100 python scripts/frame_probe.py --workload=c4 --frames=8 fused_kr_sync=$kr 2>/dev/null >> $OUT || exit 1
done
grep -o "fused_kr_sync\": [0-9]*\|\"ms\": [0-9.]*" $OUT | paste - -
