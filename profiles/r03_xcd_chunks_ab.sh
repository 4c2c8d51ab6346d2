# lower-level search kernels: workgroup -> XCD mapping (experiment build of the STAR kernels; X265HIP_ME_DBG bit 32 = chunks of 8 consecutive workgroups per XCD, 32 + 64 = chunks of 32;
# default = the hardware's round robin, neighbouring workgroups on different XCDs / L2s)
for dbg in 0 32 96 0 32 96; do
X265HIP_LIBDIR=$GRAFT_REPO_ROOT/x265-mod-by-patman_amd/exp_dbg X265HIP_ME_DBG=$dbg python bench.py --splits 1 --steps 8 --warmup 2 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg > gpurun_out/xc.json 2> gpurun_out/xc.err
python - $dbg <<'PY'
import json,sys
d=json.loads(open("gpurun_out/xc.json").read().strip().splitlines()[-1])
print("dbg", sys.argv[1], "ms per pass %.3f" % (d["ms_per_step"]/5), d["roofline"]["all_kernels_ms"])
PY
done
