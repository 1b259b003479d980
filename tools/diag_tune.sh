#!/bin/bash
# A/B experimental builds on both kernels (GPU box): tools/diag_tune.sh name1 name2 ...   ("default" = the in-tree build)
for n in "$@"; do
  if [ "$n" = default ]; then unset DMT_HIP_LIB; else export DMT_HIP_LIB=$PWD/cuda-optix-pathtracing_amd/csrc/variants/libdmt_hip_$n.so; fi
  echo "== $n"
  python tools/diag_speed.py 1024 256 8 2>&1 | tail -n 1
  python tools/diag_speed_bvh.py 1024 64 2>&1 | tail -n 1 | cut -c1-60
done
