#!/bin/bash
set -e -o pipefail
run() { # lib workload steps
  DMT_HIP_LIB=$1 timeout -k 5 200 python bench.py --no-cpu-baseline --no-secondary --workload $2 --steps $3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1'.split('/')[-1], '$2', round(d['value'],1), round(d['ms_per_step'],2))"
}
V=cuda-optix-pathtracing_amd/csrc/variants
A=cuda-optix-pathtracing_amd/csrc/libdmt_hip.so
X=$V/libdmt_hip_$1.so
DMT_HIP_LIB=$PWD/$X timeout -k 10 300 python -m pytest tests/test_parity_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q -k "bvh or c4 or c3" 2>&1 | tail -2
for l in $A $X $A $X; do run $l random1M_1024x1024_512spp_8bounces 3; done
for l in $A $X; do run $l random16M_1024x1024_64spp_8bounces 3; done
for l in $A $X; do run $l sphere_fbx_veranda_256x256_2048spp_12bounces 20; done
for l in $A $X; do run $l sphere_envmap_1024x1024_2048spp_8bounces 3; done
