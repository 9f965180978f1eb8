#!/bin/bash
# scripts/ab_run.sh OUT "lib1 lib2 ..." probe-args...   : frame_probe.py under each ab/librt_<lib>.so, stops at the first failure
OUT=$1; LIBS=$2; shift 2
for L in $LIBS; do
  echo "== $L $*" >> $OUT
  if [ "$L" = "default" ]; then
    timeout -k 10 150 python scripts/frame_probe.py "$@" 2>/dev/null >> $OUT || { echo "FAILED $L rc $?" >> $OUT; exit 1; }
  else
    RT_LIB_GPU=ab/librt_$L.so timeout -k 10 150 python scripts/frame_probe.py "$@" 2>/dev/null >> $OUT || { echo "FAILED $L rc $?" >> $OUT; exit 1; }
  fi
done
