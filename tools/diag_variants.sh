#!/bin/bash
# A/B experimental builds (GPU box): tools/diag_variants.sh <res> <spp> <depths> name1 name2 ...
res=$1; spp=$2; depths=$3; shift 3
echo "== default"; python tools/diag_speed.py $res $spp $depths 2>&1 | tail -n +2
for n in "$@"; do echo "== $n"; DMT_HIP_LIB=$PWD/cuda-optix-pathtracing_amd/csrc/variants/libdmt_hip_$n.so python tools/diag_speed.py $res $spp $depths 2>&1 | tail -n +2; done
