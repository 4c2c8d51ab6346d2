cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
(time python -m pytest tests/test_e2e_la_gpu.py tests/test_host_batch_gpu.py tests/test_lookahead_gpu.py tests/test_lookahead_golden.py -x -q -m gpu -s 2>&1 | tail -40) > gpurun_out/r03_gputest3.txt 2>&1
cat gpurun_out/r03_gputest3.txt
