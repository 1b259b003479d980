mkdir -p gpurun_out/r3
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r3/tests_c.log 2>&1; echo "tests rc=$?" ; tail -3 gpurun_out/r3/tests_c.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r3/bench_c2_c.log 2>&1; echo "bench rc=$?"; grep -ao '"value": [0-9.]*' gpurun_out/r3/bench_c2_c.log; grep -ao '"fold_handover.*' gpurun_out/r3/bench_c2_c.log
timeout -k 10 300 python bench.py --no-cpu-baseline --workload random1M_1024x1024_512spp_8bounces --steps 3 --warmup 1 > gpurun_out/r3/bench_c4_c.log 2>&1; echo "bench rc=$?"; grep -ao '"value": [0-9.]*' gpurun_out/r3/bench_c4_c.log; grep -ao '"fold_handover.*' gpurun_out/r3/bench_c4_c.log
timeout -k 10 300 python bench.py --no-cpu-baseline --workload cornell_256x256_2048spp_32bounces > gpurun_out/r3/bench_256_c.log 2>&1; echo "bench rc=$?"; grep -ao '"value": [0-9.]*' gpurun_out/r3/bench_256_c.log; grep -ao '"fold_handover.*' gpurun_out/r3/bench_256_c.log
tools/rehearse_multirank_one_gpu.sh 4 > gpurun_out/r3/gloo_n4_handover.log 2>&1; echo "rehearsal(new) rc=$?"; grep -ao '"value": [0-9.]*' gpurun_out/r3/gloo_n4_handover.log;  grep -ao '"per_rank.*' gpurun_out/r3/gloo_n4_handover.log
true
