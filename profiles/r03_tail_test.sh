# is the fixed cost of a level's launch (~0.15 ms whatever the batch size, profiles/r03_frames_scaling.txt) the tail of the few PUs that run the raster refinement?
# experiment build, X265HIP_ME_DBG=128 skips the raster in the lower levels (results wrong by construction: timing only)
for dbg in 0 128; do for f in 1 8; do
X265HIP_LIBDIR=$GRAFT_REPO_ROOT/x265-mod-by-patman_amd/exp X265HIP_ME_DBG=$dbg python bench.py --frames $f --splits 1 --steps 8 --warmup 2 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg > gpurun_out/tt_$f.json 2> gpurun_out/tt_$f.err
python - $f $dbg <<'PY'
import json,sys
f=int(sys.argv[1])
d=json.loads(open("gpurun_out/tt_%d.json"%f).read().strip().splitlines()[-1])
k=d["roofline"]["all_kernels_ms"]
print("dbg", sys.argv[2], "frames", f, "ms per pass %.3f" % (d["ms_per_step"]/5), k)
PY
done; done
