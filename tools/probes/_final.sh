cd /root/repo
timeout 900 python -m pytest tests/test_gpu_memory.py tests/test_gpu_affinity_served_sizes.py -x -q 2>&1 | tail -3 > gpurun_out/final_gpu_tests2.txt
cat gpurun_out/final_gpu_tests2.txt
bash tools/collect_final.sh > gpurun_out/collect_final.log 2>&1
tail -3 gpurun_out/collect_final.log
