for v in release exp_tiled2 release exp_tiled2; do
if [ $v = release ]; then L=A=1; else L=X265HIP_LIBDIR=$GRAFT_REPO_ROOT/x265-mod-by-patman_amd/$v; fi
env $L python bench.py --splits 1 --steps 8 --warmup 2 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'ms per pass %.3f'%(d['ms_per_step']/5), d['roofline']['all_kernels_ms'])"
done
X265HIP_LIBDIR=$GRAFT_REPO_ROOT/x265-mod-by-patman_amd/exp_tiled2 python -m pytest tests/test_host_batch_gpu.py -x -q -k "fused or oracle" 2>&1 | tail -2
