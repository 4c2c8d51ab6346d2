for dir in "" exp_nic16 "" exp_nic16; do
if [ -n "$dir" ]; then export X265HIP_LIBDIR=$GRAFT_REPO_ROOT/x265-mod-by-patman_amd/$dir; else unset X265HIP_LIBDIR; fi
python bench.py --workload 4320p10_slower --frames 2 --steps 6 --warmup 2 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg 2>/dev/null | python -c "
import json,sys,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(os.environ.get('X265HIP_LIBDIR','default')[-10:], '8K ms per pass %.3f'%(d['ms_per_step']/5), d['roofline']['all_kernels_ms'])"
done
X265HIP_LIBDIR=$GRAFT_REPO_ROOT/x265-mod-by-patman_amd/exp_nic16 python -m pytest tests/test_me_gpu.py -x -q -k merange_128 2>&1 | tail -2
