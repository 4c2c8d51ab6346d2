# what do the cold phase planes cost the sub-pel stage?  Experiment build of the STAR kernels, X265HIP_ME_DBG=256: every sub-pel candidate is read from plane 0 at the same
# offsets (same instruction stream, same number of loads; results wrong by construction: timing only)
for dbg in 0 256 0 256; do
X265HIP_LIBDIR=$GRAFT_REPO_ROOT/x265-mod-by-patman_amd/exp_dbg X265HIP_ME_DBG=$dbg python bench.py --splits 1 --steps 8 --warmup 2 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg > gpurun_out/cp.json 2> gpurun_out/cp.err
python - $dbg <<'PY'
import json,sys
d=json.loads(open("gpurun_out/cp.json").read().strip().splitlines()[-1])
print("dbg", sys.argv[1], "ms per pass %.3f" % (d["ms_per_step"]/5), d["roofline"]["all_kernels_ms"])
PY
done
