#!/bin/bash
OUT=gpurun_out/sweep_ku.log; rm -f $OUT
for o in "fused_ku=0" "fused_ku=4" "fused_ku=6" "fused_ku=10" "fused_ku=12" "fused_ku=16" "fused_slots=16" "fused_slots=24" "fused_tail4=30" "fused_tail4=80" "fused_tail2=0" "fused_tail2=60" "fused_partner_min=16"; do
  echo "== $o" >> $OUT
  timeout -k 10 100 python scripts/frame_probe.py --frames=16 $o 2>/dev/null >> $OUT || exit 1
done
grep -o "^== .*\|\"ms\": [0-9.]*" $OUT | paste - -
