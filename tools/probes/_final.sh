cd /root/repo
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/final_gpu_tests.txt
cat gpurun_out/final_gpu_tests.txt
bash tools/collect_final.sh > gpurun_out/collect_final.log 2>&1
tail -3 gpurun_out/collect_final.log
