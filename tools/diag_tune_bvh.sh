#!/bin/bash
# A/B experimental builds, BVH kernel only (GPU box): tools/diag_tune_bvh.sh name1 name2 ...
for n in "$@"; do
  if [ "$n" = default ]; then unset DMT_HIP_LIB; else export DMT_HIP_LIB=$PWD/cuda-optix-pathtracing_amd/csrc/variants/libdmt_hip_$n.so; fi
  echo "== $n"
  python tools/diag_speed_bvh.py 1024 64 2>&1 | tail -n 4 | cut -c1-200 | grep -v "^{"
done
