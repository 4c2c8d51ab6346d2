# candidates of a costing round paired by LDS bank parity (star64_kernel, pair_slots): release library (before) against x265-mod-by-patman_amd/exp_pair (after)
run() { name=$1; sp=$2; shift 2; env "$@" python bench.py --splits $sp --steps 10 --warmup 3 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg > gpurun_out/ps_$name.json 2> gpurun_out/ps_$name.err; python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
d=json.loads(open("gpurun_out/ps_%s.json"%n).read().strip().splitlines()[-1])
print(n, "ms per pass %.3f" % (d["ms_per_step"]/5), d["roofline"]["all_kernels_ms"])
PY
}
P=X265HIP_LIBDIR=$GRAFT_REPO_ROOT/x265-mod-by-patman_amd/exp_pair
run before_1 1 A=1
run after_1 1 $P
run before_2 2 A=1
run after_2 2 $P
run before_1b 1 A=1
run after_1b 1 $P
X265HIP_LIBDIR=$GRAFT_REPO_ROOT/x265-mod-by-patman_amd/exp_pair python -m pytest tests/test_me_gpu.py tests/test_pipeline_gpu.py -x -q 2>&1 | tail -2
