# r04: star64_kernel variant against the release library: 10-bit parity of the 64x64 STAR paths (exhaustive full-size tests), section clocks (xs_p), me64 time (xs_n)
X265HIP_LIBDIR=$GRAFT_REPO_ROOT/x265-mod-by-patman_amd/xs_n timeout 600 python -m pytest tests/test_host_batch_gpu.py tests/test_me_gpu.py -q -m gpu -n 4 --timeout 400 -k "every_pu or (merange and 10) or (matches_oracle and 10) or (outside and 10) or (extreme and 10)" 2>&1 | tail -3
X265HIP_LIBDIR=$GRAFT_REPO_ROOT/x265-mod-by-patman_amd/xs_p timeout 300 python bench.py --splits 1 --steps 1 --warmup 0 --inner 1 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg 2>&1 | grep s64prof | head -6
for v in release xs_n release xs_n; do
  if [ $v = release ]; then unset X265HIP_LIBDIR; else export X265HIP_LIBDIR=$GRAFT_REPO_ROOT/x265-mod-by-patman_amd/$v; fi
  python bench.py --steps 10 --warmup 3 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg > gpurun_out/r04_s64n_$v.json 2> gpurun_out/r04_s64n_$v.err
  python - $v <<'PY'
import json,sys
v=sys.argv[1]
j=json.loads(open("gpurun_out/r04_s64n_%s.json"%v).read().strip().splitlines()[-1])
print(v, "Mpx/s", j["value"], "ms per pass", round(j["ms_per_step"]/5,4), j["roofline"]["all_kernels_ms"])
PY
done
