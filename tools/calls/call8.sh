mkdir -p gpurun_out/c8 && cd $GRAFT_REPO_ROOT
O=gpurun_out/c8
python -m pytest tests/test_gpu_fp16_loop.py -q -s -k "end_to_end or stage" > $O/pytest_fp16.log 2>&1; grep -n "fp16 loop" $O/pytest_fp16.log | head -30; tail -3 $O/pytest_fp16.log
python -m pytest tests/test_gpu_e2e.py -x -q -s -k "three_objects or noise_floor" > $O/pytest_c3.log 2>&1; tail -4 $O/pytest_c3.log
