#!/bin/bash
# Collects the per-round evidence on the MI355X box (run through gpurun from the repo root):
#   profiles/collect.sh <tag> [bench args]      -> gpurun_out/<tag>/{stats,fetch,write,sq1,sq2}/..., gpurun_out/<tag>/bench.json
# The counter passes and the per-launch statistics run the batch as ONE sub-batch with one pass per step (--splits 1 --inner 1): a launch then covers the whole batch and the
# per-launch figures in profiles/{traffic,valu}_<workload>.json are per full-batch launch (bench.py scales them by the share of the batch a timed launch covers).
# Counter passes are separate runs with no tracing besides --kernel-trace (see the prompt's rocprofv3 rules); each pass has its own time limit.
set -u
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 "$@" > $out/bench.json 2> $out/bench.err
# the same command under the profiler (default schedule: sub-batches on their own streams, stages of different sub-batches overlap)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_default -- python bench.py --steps 4 --warmup 1 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg "$@" > $out/stats_default_bench.json 2> $out/stats_default.err
find $out/stats_default -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats_default.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python bench.py --steps 10 --warmup 2 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg --splits 1 --inner 1 "$@" > $out/stats_bench.json 2> $out/stats.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python bench.py --steps 3 --warmup 1 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg --splits 1 --inner 1 "$@" > /dev/null 2> $out/fetch.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python bench.py --steps 3 --warmup 1 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg --splits 1 --inner 1 "$@" > /dev/null 2> $out/write.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $out/sq1 -- python bench.py --steps 3 --warmup 1 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg --splits 1 --inner 1 "$@" > /dev/null 2> $out/sq1.err
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $out/sq2 -- python bench.py --steps 3 --warmup 1 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg --splits 1 --inner 1 "$@" > /dev/null 2> $out/sq2.err
src="$tag, code $(cat profiles/.commit 2>/dev/null || echo unknown)"
python profiles/summarize_pmc.py --traffic $out/fetch $out/write "$src" > $out/traffic.json
python profiles/summarize_pmc.py --valu $out/sq1 "$src" > $out/valu.json
python profiles/summarize_pmc.py $out/sq1 $out/sq2 > $out/sq.txt
find $out/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
find $out/stats -name "*kernel_trace.csv" | head -1 | xargs -I{} python profiles/kstats.py $out/kernel_stats.csv {} > $out/kstats.txt
# keep the merged-back payload small: the raw traces are not needed once summarised
find $out -name "*kernel_trace.csv" -delete; find $out -name "*counter_collection.csv" -delete
cat $out/bench.json; cat $out/traffic.json; cat $out/valu.json; cat $out/kstats.txt
