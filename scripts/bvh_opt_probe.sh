#!/bin/bash
export RT_LIB_GPU=ab/librt_bvh1.so
OUT=gpurun_out/bvh_opt.log; rm -f $OUT
for p in 0 1 3 6; do
  echo "== c2 bvh_optimise=$p" >> $OUT
  timeout -k 10 100 python scripts/frame_probe.py bvh_optimise=$p 2>/dev/null >> $OUT || exit 1
  timeout -k 10 100 python scripts/fused_stats.py 100 --stats bvh_optimise=$p 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('node','leaf')})" >> $OUT || exit 1
done
for p in 0 3; do
  echo "== c5 bvh_optimise=$p" >> $OUT
  timeout -k 10 100 python scripts/frame_probe.py --workload=c5 --frames=8 bvh_optimise=$p 2>/dev/null >> $OUT || exit 1
done
cat $OUT
