#!/bin/bash
# profiles/kstats.sh <tag> [bench args]: rocprofv3 --kernel-trace --stats of a short bench run, per-kernel averages to gpurun_out/<tag>_kernel_stats.csv
tag=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_$tag -- python bench.py --steps 10 --warmup 2 --cpu-ctus 0 "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}.err
find gpurun_out/ks_$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${tag}_kernel_stats.csv
find gpurun_out/ks_$tag -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} /tmp/${tag}_trace.csv
rm -rf gpurun_out/ks_$tag
python profiles/kstats.py gpurun_out/${tag}_kernel_stats.csv /tmp/${tag}_trace.csv | tee gpurun_out/${tag}_kstats.txt
