mkdir -p gpurun_out/c20 && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_bench_region.py -q -m gpu -x > gpurun_out/c20/t.log 2>&1; tail -15 gpurun_out/c20/t.log
