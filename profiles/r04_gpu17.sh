# r04: non-temporal plane stores at 8 bit (1080p8_medium): xp_nt8 (streaming) against xp_t8 (plain), two runs each
for v in xp_t8 xp_nt8 xp_t8 xp_nt8; do
  X265HIP_LIBDIR=$GRAFT_REPO_ROOT/x265-mod-by-patman_amd/$v python bench.py --workload 1080p8_medium --steps 20 --warmup 5 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg > gpurun_out/r04_nt8_$v.json 2> gpurun_out/r04_nt8_$v.err
  python - $v <<'PY'
import json,sys
v=sys.argv[1]
j=json.loads(open("gpurun_out/r04_nt8_%s.json"%v).read().strip().splitlines()[-1])
print(v, "Mpx/s", j["value"], "ms per pass", round(j["ms_per_step"]/5,4), j["roofline"]["all_kernels_ms"])
PY
done
