# cumulative stage times of the search kernels at 4K 10 bit (experiment build, X265HIP_ME_DBG: 1 = set-up only, 2 = + start stage, 4 = + full-pel search, 0 = everything);
# one stream, so that the per-kernel times are those of kernels that own the GPU.  Results of the cut runs are wrong by construction (timing only).
run() { name=$1; shift; env X265HIP_LIBDIR=$GRAFT_REPO_ROOT/x265-mod-by-patman_amd/exp "$@" python bench.py --splits 1 --steps 6 --warmup 2 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg > gpurun_out/st_$name.json 2> gpurun_out/st_$name.err; python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
try:
    d=json.loads(open("gpurun_out/st_%s.json"%n).read().strip().splitlines()[-1])
    print(n, "ms per pass %.3f" % (d["ms_per_step"]/5), d["roofline"]["all_kernels_ms"])
except Exception as e:
    print(n, "failed", e); print(open("gpurun_out/st_%s.err"%n).read()[-600:])
PY
}
run full A=1
run dbg4 X265HIP_ME_DBG=4
run dbg2 X265HIP_ME_DBG=2
run dbg1 X265HIP_ME_DBG=1
run full_b A=1
