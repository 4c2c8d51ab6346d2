B="python bench.py --cpu-ctus 0 --steps 12 --warmup 3"
pick() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', d['value'], d['ms_per_step'], d.get('kernels_ms'))
"; }
$B 2>/dev/null | pick base
X265HIP_STAR64_DBG=32 $B 2>/dev/null | pick contiguous
X265HIP_STAR64_DBG=64 $B 2>/dev/null | pick chunk60
X265HIP_STAR64_DBG=5 $B 2>/dev/null | pick winonly
X265HIP_STAR64_DBG=37 $B 2>/dev/null | pick winonly_contig
X265HIP_STAR64_DBG=69 $B 2>/dev/null | pick winonly_chunk
$B --no-planes 2>/dev/null | pick noplanes
