B="python bench.py --cpu-ctus 0 --steps 12 --warmup 3"
pick() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms'])
"; }
$B 2>/dev/null | pick 2160p10
$B --workload 1080p8_medium 2>/dev/null | pick 1080p8
$B --workload 4320p10_slower --frames 2 2>/dev/null | pick 4320p10
