cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
(time python -m pytest tests/test_e2e_la_gpu.py -x -q -m gpu -s 2>&1 | tail -25) > gpurun_out/r03_gputest5.txt 2>&1
cat gpurun_out/r03_gputest5.txt
for la in 0 1; do X265TME=0 X265TMEGPU=0 X265LAGPU=$la MALLOC_PERTURB_=85 oracle/_ref/x265e2e_8 x265-mod-by-patman_amd/libx265hip_8.so 1920 1088 12 medium /tmp/la$la.hevc; md5sum /tmp/la$la.hevc; done > gpurun_out/r03_la_1080p.txt 2>&1
for la in 0 1; do X265TME=0 X265TMEGPU=0 X265LAGPU=$la MALLOC_PERTURB_=85 oracle/_ref/x265e2e_8 x265-mod-by-patman_amd/libx265hip_8.so 1920 1088 24 ultrafast /tmp/la$la.hevc bframes=4 b-adapt=2 rc-lookahead=20; md5sum /tmp/la$la.hevc; done >> gpurun_out/r03_la_1080p.txt 2>&1
cat gpurun_out/r03_la_1080p.txt
