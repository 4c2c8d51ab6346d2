#!/bin/bash
# profiles/pmc.sh <tag> "<counters pass 1>" "<counters pass 2>" ... : one rocprofv3 --pmc run per counter set, summary to gpurun_out/<tag>.txt
tag=$1; shift
export TMPDIR=/tmp
i=0
: > gpurun_out/$tag.txt
for set in "$@"; do
  i=$((i+1))
  mkdir -p gpurun_out/pmc_$tag; timeout -k 5 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_$tag/p$i -- python bench.py --steps 3 --warmup 1 --cpu-ctus 0 $BENCH_ARGS > /dev/null 2> gpurun_out/pmc_$tag/p$i.err
  python profiles/summarize_pmc.py gpurun_out/pmc_$tag/p$i | grep -v "at::native" >> gpurun_out/$tag.txt
done
rm -rf gpurun_out/pmc_$tag
cat gpurun_out/$tag.txt
