# r04: what part of star64_kernel's pattern passes is SAD arithmetic: the kernel cut off behind its first pattern pass (X265HIP_STAR64_DBG=1) and whole (0), with the passes' SAD
# loops over all 32 rows of a PU half (xs_a) and over ONE row (xs_b: wrong results, same control flow up to the decisions) -- experiment objects of kern_star64.hip
for v in xs_a xs_b; do for d in 5 1 0; do
  X265HIP_STAR64_DBG=$d X265HIP_LIBDIR=$GRAFT_REPO_ROOT/x265-mod-by-patman_amd/$v python bench.py --splits 1 --steps 6 --warmup 2 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg > gpurun_out/r04_s64o_${v}_$d.json 2> gpurun_out/r04_s64o_${v}_$d.err
  python - $v $d <<'PY'
import json,sys
v,d=sys.argv[1:3]
try:
    j=json.loads(open("gpurun_out/r04_s64o_%s_%s.json"%(v,d)).read().strip().splitlines()[-1])
    print(v, "dbg", d, "me64 ms", j["roofline"]["all_kernels_ms"]["me64"])
except Exception as e:
    print(v, d, "failed", e)
PY
done; done
