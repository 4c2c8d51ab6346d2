cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
(time python -m pytest tests -x -q -m gpu 2>&1 | tail -15) > gpurun_out/r03_gputest4.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench4.json 2> gpurun_out/r03_bench4.err
cat gpurun_out/r03_gputest4.txt; tail -c 400 gpurun_out/r03_bench4.err
python -c "
import json
d=json.loads(open('gpurun_out/r03_bench4.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms']); print(json.dumps(d['e2e_fps'])[:2500])"
