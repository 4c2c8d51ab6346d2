for sp in 2 3 4 2 3 4; do python bench.py --splits $sp --steps 8 --warmup 2 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('splits $sp ms per pass %.3f'%(d['ms_per_step']/5))"; done
