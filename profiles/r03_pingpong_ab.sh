# A/B of the alternating two-stream schedule (experiment build: make OUT=$PWD/exp/ OBJ=$PWD/exp/obj EXPERIMENTS=1 in x265-mod-by-patman_amd)
export X265HIP_LIBDIR=$GRAFT_REPO_ROOT/x265-mod-by-patman_amd/exp
B="python bench.py --steps 10 --warmup 3 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg"
run() { name=$1; shift; env "$@" $B > gpurun_out/pp_$name.json 2> gpurun_out/pp_$name.err; python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
d=json.loads(open("gpurun_out/pp_%s.json"%n).read().strip().splitlines()[-1])
print(n, "ms per pass %.3f" % (d["ms_per_step"]/5), d["roofline"]["all_kernels_ms"])
PY
}
run one_stream A=1
B="$B --splits 2"
run two_streams A=1
run pingpong X265HIP_PINGPONG=1
run pingpong_pad8k X265HIP_PINGPONG=1 X265HIP_STAR64_LDSPAD=8192
run two_streams_pad8k X265HIP_STAR64_LDSPAD=8192
B="python bench.py --steps 10 --warmup 3 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg"
run one_stream_pad8k X265HIP_STAR64_LDSPAD=8192
