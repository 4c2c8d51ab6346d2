cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
(time python -m pytest tests/test_e2e_tme_gpu.py tests/test_e2e_la_gpu.py -x -q -m gpu 2>&1 | tail -8) > gpurun_out/r03_gputest6.txt 2>&1
cat gpurun_out/r03_gputest6.txt
for i in 1 2; do for t in 0 1; do X265TMEGPU=$t MALLOC_PERTURB_=85 oracle/_ref/x265tmegpu_8 x265-mod-by-patman_amd/libx265hip_8.so 1920 1088 8 medium /tmp/t$t.hevc; md5sum /tmp/t$t.hevc; done; done > gpurun_out/r03_tme_1080p.txt 2>&1
for t in 0 1; do X265TMEGPU=$t MALLOC_PERTURB_=85 oracle/_ref/x265tmegpu_8 x265-mod-by-patman_amd/libx265hip_8.so 1920 1088 6 slow /tmp/t$t.hevc; md5sum /tmp/t$t.hevc; done >> gpurun_out/r03_tme_1080p.txt 2>&1
cat gpurun_out/r03_tme_1080p.txt
