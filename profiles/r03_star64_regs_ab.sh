# star64_kernel's registers (LDS arrays reordered; raster row groups of 6 / 4 / 3): 161 / 148 / 144 vector registers -- what fits beside two of its workgroups on a SIMD
run() { name=$1; dir=$2; sp=$3; X265HIP_LIBDIR=$GRAFT_REPO_ROOT/x265-mod-by-patman_amd/$dir python bench.py --splits $sp --steps 10 --warmup 3 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg > gpurun_out/sr_$name.json 2> gpurun_out/sr_$name.err; python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
d=json.loads(open("gpurun_out/sr_%s.json"%n).read().strip().splitlines()[-1])
print(n, "ms per pass %.3f" % (d["ms_per_step"]/5), d["roofline"]["all_kernels_ms"])
PY
}
for sp in 2 1; do
run gr6_$sp exp_gr6 $sp
run gr4_$sp exp_gr4 $sp
run gr3_$sp exp_gr3 $sp
done
run gr6_2b exp_gr6 2
