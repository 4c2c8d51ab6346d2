cd $GRAFT_REPO_ROOT
(time python -m pytest tests/test_e2e_tme_gpu.py tests/test_e2e_la_gpu.py -x -q -m gpu 2>&1 | tail -5) > gpurun_out/r03_gputest9.txt 2>&1
cat gpurun_out/r03_gputest9.txt
bash profiles/e2e_tme.sh 1920 1088 12 medium > gpurun_out/r03_e2e_tme_prof.txt 2>&1
bash profiles/e2e_tme.sh 1920 1088 6 slow >> gpurun_out/r03_e2e_tme_prof.txt 2>&1
bash profiles/e2e_tme.sh 1280 704 16 medium >> gpurun_out/r03_e2e_tme_prof.txt 2>&1
bash profiles/e2e_tme.sh 1280 704 8 slow >> gpurun_out/r03_e2e_tme_prof.txt 2>&1
cat gpurun_out/r03_e2e_tme_prof.txt
