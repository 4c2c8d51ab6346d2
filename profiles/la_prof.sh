#!/bin/bash
# profiles/la_prof.sh <tag>: rocprofv3 evidence for the lookahead leg (bench.py --lookahead): kernel stats, then separate counter passes
tag=$1
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
A="--steps 4 --warmup 1 --cpu-ctus 0 --lookahead"
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python bench.py $A > $out/bench.json 2> $out/stats.err
timeout -k 5 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TA_FLAT_READ_WAVEFRONTS_sum TA_TA_BUSY_sum --output-format csv -d $out/ta -- python bench.py $A > /dev/null 2> $out/ta.err
timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD --output-format csv -d $out/sq -- python bench.py $A > /dev/null 2> $out/sq.err
python profiles/summarize_pmc.py $out/ta $out/sq | grep "la_\|lowres\|extend" > $out/pmc.txt
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
find $out -name "*kernel_trace.csv" -delete; find $out -name "*counter_collection.csv" -delete
grep "la_\|lowres\|extend\|Name" $out/kernel_stats.csv; cat $out/pmc.txt
