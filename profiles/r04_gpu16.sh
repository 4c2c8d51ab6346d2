# r04: non-temporal stores of the phase planes (xp_nt) against the release library, two runs each
for v in ${VARIANTS:-release xp_nt release xp_nt}; do
  if [ $v = release ]; then unset X265HIP_LIBDIR; else export X265HIP_LIBDIR=$GRAFT_REPO_ROOT/x265-mod-by-patman_amd/$v; fi
  python bench.py --steps 10 --warmup 3 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg > gpurun_out/r04_nt_$v.json 2> gpurun_out/r04_nt_$v.err
  python - $v <<'PY'
import json,sys
v=sys.argv[1]
j=json.loads(open("gpurun_out/r04_nt_%s.json"%v).read().strip().splitlines()[-1])
print(v, "Mpx/s", j["value"], "ms per pass", round(j["ms_per_step"]/5,4), j["roofline"]["all_kernels_ms"])
PY
done
