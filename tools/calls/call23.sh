mkdir -p gpurun_out/c23 && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_a*.py tests/test_gpu_b*.py tests/test_gpu_c*.py tests/test_gpu_e2e.py -q -m gpu -x --durations=5 > gpurun_out/c23/t.log 2>&1; tail -14 gpurun_out/c23/t.log
