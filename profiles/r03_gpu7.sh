cd $GRAFT_REPO_ROOT
for t in 1 1; do X265TMEGPU=$t MALLOC_PERTURB_=85 oracle/_ref/x265tmegpu_8 x265-mod-by-patman_amd/libx265hip_8.so 1920 1088 12 medium /tmp/t$t.hevc; done > gpurun_out/r03_tme_1080p_b.txt 2>&1
X265TMEGPU=1 MALLOC_PERTURB_=85 oracle/_ref/x265tmegpu_8 x265-mod-by-patman_amd/libx265hip_8.so 1920 1088 12 medium /tmp/t.hevc pools=8 >> gpurun_out/r03_tme_1080p_b.txt 2>&1
cat gpurun_out/r03_tme_1080p_b.txt
