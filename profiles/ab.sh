#!/bin/bash
# ab.sh <tag> <variant dir | release> ...: the headline pass (4K 10 bit) with each library, one stream and the two-stream schedule, interleaved twice; then the parity
# tests of the search kernels with the LAST variant.  Variants are built by profiles/quick_variant.sh (or make OUT=...); "release" = the in-tree library.
tag=$1; shift
run() { name=$1; dir=$2; sp=$3; if [ $dir = release ]; then L=A=1; else L=X265HIP_LIBDIR=$GRAFT_REPO_ROOT/x265-mod-by-patman_amd/$dir; fi
  env $L python bench.py --splits $sp --steps 8 --warmup 2 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg ${WORKLOAD:+--workload $WORKLOAD} > gpurun_out/${tag}_$name.json 2> gpurun_out/${tag}_$name.err; python - "$tag" "$name" <<'PY'
import json,sys
t,n=sys.argv[1:3]
try:
    d=json.loads(open("gpurun_out/%s_%s.json"%(t,n)).read().strip().splitlines()[-1])
    print(n, "ms per pass %.3f" % (d["ms_per_step"]/5), d["roofline"]["all_kernels_ms"])
except Exception as e:
    print(n, "failed", e); print(open("gpurun_out/%s_%s.err"%(t,n)).read()[-800:])
PY
}
for rep in a b; do for v in "$@"; do for sp in 1 2; do run ${v}_${sp}$rep $v $sp; done; done; done
last=${@: -1}
if [ $last != release ]; then X265HIP_LIBDIR=$GRAFT_REPO_ROOT/x265-mod-by-patman_amd/$last python -m pytest tests/test_me_gpu.py tests/test_pipeline_gpu.py tests/test_host_batch_gpu.py -x -q 2>&1 | tail -3; fi
