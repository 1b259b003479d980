#!/bin/bash
# A/B wavefront builds (GPU box): tools/diag_wf_variants.sh <spp> name1 name2 ...
spp=$1; shift
for n in "$@"; do
  if [ "$n" = default ]; then unset DMT_HIP_LIB; else export DMT_HIP_LIB=$PWD/cuda-optix-pathtracing_amd/csrc/variants/libdmt_hip_$n.so; fi
  echo "== $n"; python tools/diag_wf.py $spp 2>&1 | tail -n 1
done
