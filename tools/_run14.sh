mkdir -p gpurun_out/r3
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "sampler or film or resum or chunk or fullsize or c1 or c2" > gpurun_out/r3/tests_g.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3/tests_g.log
V=$PWD/cuda-optix-pathtracing_amd/csrc/variants
for lib in cur r02head cur; do
  if [ $lib = cur ]; then unset DMT_HIP_LIB; else export DMT_HIP_LIB=$V/libdmt_hip_$lib.so; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --steps 10 > gpurun_out/r3/ab6_c2_${lib}.log 2>&1; echo "c2 $lib rc=$? $(grep -ao '"value": [0-9.]*' gpurun_out/r3/ab6_c2_${lib}.log | head -1)"
done
unset DMT_HIP_LIB
timeout -k 10 300 python bench.py --no-cpu-baseline --workload random1M_1024x1024_512spp_8bounces --steps 3 --warmup 1 > gpurun_out/r3/ab6_c4_cur.log 2>&1; echo "c4 cur rc=$? $(grep -ao '"value": [0-9.]*' gpurun_out/r3/ab6_c4_cur.log | head -1)"
