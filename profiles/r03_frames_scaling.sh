# per-level time against the number of pictures in the batch (one stream): does the working set (16 phase planes of 340 MB per 4K 10-bit picture) matter?
for f in 1 2 4 8; do
python bench.py --frames $f --splits 1 --steps 8 --warmup 2 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg > gpurun_out/fs_$f.json 2> gpurun_out/fs_$f.err
python - $f <<'PY'
import json,sys
f=int(sys.argv[1])
d=json.loads(open("gpurun_out/fs_%d.json"%f).read().strip().splitlines()[-1])
k=d["roofline"]["all_kernels_ms"]
print("frames", f, "ms per pass %.3f" % (d["ms_per_step"]/5), "per picture:", {a: round(b/f,4) for a,b in k.items()})
PY
done
