# final validation at the shipping tree: the driver's round-end sequence (pytest -m gpu, smoke)
mkdir -p gpurun_out/c18 && cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu -x --durations=12 > gpurun_out/c18/pytest_gpu.log 2>&1; tail -22 gpurun_out/c18/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/c18/smoke.log 2>&1; tail -2 gpurun_out/c18/smoke.log
