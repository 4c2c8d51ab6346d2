cd $GRAFT_REPO_ROOT
bash profiles/e2e_tme.sh 1920 1088 12 medium > gpurun_out/r03_e2e_tme_prof.txt 2>&1
bash profiles/e2e_tme.sh 1920 1088 6 slow >> gpurun_out/r03_e2e_tme_prof.txt 2>&1
cat gpurun_out/r03_e2e_tme_prof.txt
