#!/bin/bash
# A/B builds of the GPU library: scripts/build_variant.sh NAME [extra hipcc flags] -> ab/librt_NAME.so
# (run with RT_LIB_GPU=ab/librt_NAME.so; ab/ is git-ignored and travels to the GPU box)
set -e
cd "$(dirname "$0")/.."
mkdir -p ab
NAME=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function "$@" \
  -o ab/librt_$NAME.so raytracer_amd/csrc/rt_lib.hip raytracer_amd/csrc/rt_multi.hip raytracer_amd/csrc/rt_bvh.cpp -lpthread
echo built ab/librt_$NAME.so
