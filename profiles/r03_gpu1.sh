set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
(time python -m pytest tests -x -q -m gpu 2>&1 | tail -15) > gpurun_out/r03_gputest1.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench1.json 2> gpurun_out/r03_bench1.err
tail -c 1500 gpurun_out/r03_bench1.err
cat gpurun_out/r03_gputest1.txt
python -c "
import json; d=json.loads(open('gpurun_out/r03_bench1.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step']); print(json.dumps(d['e2e_fps'])[:1500]); print(json.dumps(d['tme_producer'])[:1200])"
