mkdir -p gpurun_out/r3
timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --steps 5 --cornell-bvh > gpurun_out/r3/c2_through_bvh.log 2>&1; echo "c2 via bvh rc=$? $(grep -ao '"value": [0-9.]*' gpurun_out/r3/c2_through_bvh.log | head -1) $(grep -ao '"film_ok": [a-z]*' gpurun_out/r3/c2_through_bvh.log)"
timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --steps 5 > gpurun_out/r3/c2_brute.log 2>&1; echo "c2 brute rc=$? $(grep -ao '"value": [0-9.]*' gpurun_out/r3/c2_brute.log | head -1)"
tools/rehearse_multirank_one_gpu.sh 4 > gpurun_out/r3/gloo_n4_final.log 2>&1; echo "rehearsal n4 rc=$?"; grep -ao '"value": [0-9.]*' gpurun_out/r3/gloo_n4_final.log | head -1; grep -ao '"fold_handover.*' gpurun_out/r3/gloo_n4_final.log
DMT_COMBINE=gather tools/rehearse_multirank_one_gpu.sh 2 > gpurun_out/r3/gloo_n2_gather_final.log 2>&1; echo "rehearsal n2 gather rc=$?"; grep -ao '"value": [0-9.]*' gpurun_out/r3/gloo_n2_gather_final.log | head -1
