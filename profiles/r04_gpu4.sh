# r04: the chain kernels with the references of a PU side by side (tme_chain.inc RP): parity (chain kernels against the launch-per-stage path, recorded reference calls,
# bitstreams), then the in-encode producer time (bench's e2e leg: 1080p medium, ref 3, B pictures)
python -m pytest tests/test_tme_producer_gpu.py tests/test_tme_gpu.py tests/test_e2e_tme_gpu.py tests/test_e2e_la_gpu.py tests/test_e2e_ff_gpu.py -q -x 2>&1 | tail -5
python bench.py --no-preset-exact --no-streams-leg --cpu-ctus 0 > gpurun_out/r04_b5.json 2> gpurun_out/r04_b5.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_b5.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"]); print(json.dumps(d["e2e_fps"]["tme"])); print(d["e2e_fps"]["fps"]); print(json.dumps(d["e2e_fps"].get("lookahead")))
print({k:(v.get("ms"), v.get("pictures_per_s_4_threads")) for k,v in d["tme_producer"]["presets"].items()})
PY
bash profiles/e2e_tme.sh 1920 1088 12 medium 2>&1 | tail -8; bash profiles/e2e_tme.sh 1920 1088 8 slow 2>&1 | tail -8
