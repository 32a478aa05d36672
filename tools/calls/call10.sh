mkdir -p gpurun_out/c10 && cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu -x --durations=15 > gpurun_out/c10/pytest_gpu.log 2>&1; tail -30 gpurun_out/c10/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/c10/smoke.log 2>&1; tail -2 gpurun_out/c10/smoke.log
