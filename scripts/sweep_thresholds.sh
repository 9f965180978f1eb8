#!/bin/bash
# scripts/sweep_thresholds.sh WORKLOAD [LIB]: frame time over the traversal thresholds (kn x kb, then kr)
WL=${1:-c2}; LIB=${2:-}
OUT=gpurun_out/sweep_$WL.log; rm -f $OUT
[ -n "$LIB" ] && export RT_LIB_GPU=ab/librt_$LIB.so
for kn in 4 8 12 16; do for kb in 8 16 24; do
  echo "== kn=$kn kb=$kb" >> $OUT
  timeout -k 10 100 python scripts/frame_probe.py --workload=$WL --frames=8 fused_kn=$kn fused_kb=$kb 2>/dev/null >> $OUT || exit 1
done; done
for kr in 8 24 32; do
  echo "== kr=$kr" >> $OUT
  timeout -k 10 100 python scripts/frame_probe.py --workload=$WL --frames=8 fused_kr=$kr 2>/dev/null >> $OUT || exit 1
done
grep -o "fused_k[a-z]\": [0-9]*\|\"ms\": [0-9.]*" $OUT | paste - - - 
