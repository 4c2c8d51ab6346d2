export X265HIP_LIBDIR=$GRAFT_REPO_ROOT/x265-mod-by-patman_amd/exp
run() { name=$1; sp=$2; shift 2; env "$@" python bench.py --splits $sp --steps 10 --warmup 3 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg > gpurun_out/ring_$name.json 2> gpurun_out/ring_$name.err; python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
d=json.loads(open("gpurun_out/ring_%s.json"%n).read().strip().splitlines()[-1])
print(n, "ms per pass %.3f" % (d["ms_per_step"]/5))
PY
}
run two 2 A=1
run three_lockstep 3 A=1
run three_ring 3 X265HIP_RING=1
run four_ring 4 X265HIP_RING=1
run eight_ring 8 X265HIP_RING=1
run two_again 2 A=1
