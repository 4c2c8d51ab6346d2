cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
(time python -m pytest tests -x -q -m gpu 2>&1 | tail -8) > gpurun_out/r03_gputest13.txt 2>&1
(time python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench13.json 2> gpurun_out/r03_bench13.err) 2>> gpurun_out/r03_gputest13.txt
cat gpurun_out/r03_gputest13.txt; tail -c 300 gpurun_out/r03_bench13.err
python -c "
import json
d=json.loads(open('gpurun_out/r03_bench13.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['all_kernels_ms']); print(d['streams']); print(d['preset_exact']['ms_per_pass'], d['preset_exact']['vs_headline_per_searched_pu']); print(d['e2e_fps']['fps'], d['e2e_fps']['bitstream_identical']); print(d['roofline_step']['traffic_source'])"
