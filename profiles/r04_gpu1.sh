# r04: lookahead batch binding tests, the default bench with its auxiliary legs in child processes, star64_kernel's stages (experiment build, X265HIP_STAR64_DBG)
python -m pytest tests/test_e2e_la_gpu.py -q -x -s 2>&1 | grep -E "passed|failed|e2e la batch|Error|assert" | tail -12
python bench.py > gpurun_out/r04_b4.json 2> gpurun_out/r04_b4.err; echo "bench rc=$?"; grep -v amdgpu.ids gpurun_out/r04_b4.err | tail -4
for d in 0 5 1 2 4; do
  X265HIP_STAR64_DBG=$d X265HIP_LIBDIR=$GRAFT_REPO_ROOT/x265-mod-by-patman_amd/exp python bench.py --splits 1 --steps 6 --warmup 2 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg > gpurun_out/r04_s64_dbg$d.json 2> gpurun_out/r04_s64_dbg$d.err
  python - $d <<'PY'
import json,sys
d=sys.argv[1]
try:
    j=json.loads(open("gpurun_out/r04_s64_dbg%s.json"%d).read().strip().splitlines()[-1])
    print("star64 dbg", d, "me64 ms", j["roofline"]["all_kernels_ms"]["me64"], "pass", round(j["ms_per_step"]/5,3))
except Exception as e:
    print("dbg", d, "failed", e)
PY
done
