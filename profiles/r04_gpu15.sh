# r04: the other two BASELINE workloads with the final code (one-stream per-stage times + the default two-stream value)
for w in 1080p8_medium 4320p10_slower; do
  F=8; [ $w = 4320p10_slower ] && F=2
  python bench.py --workload $w --frames $F --steps 10 --warmup 3 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg > gpurun_out/r04_final_$w.json 2> gpurun_out/r04_final_$w.err
  python - $w <<'PY'
import json,sys
w=sys.argv[1]
j=json.loads(open("gpurun_out/r04_final_%s.json"%w).read().strip().splitlines()[-1])
print(w, "Mpx/s", j["value"], "ms per pass", round(j["ms_per_step"]/5,4), j["roofline"]["all_kernels_ms"], "frac", j["roofline"]["frac"])
PY
done
