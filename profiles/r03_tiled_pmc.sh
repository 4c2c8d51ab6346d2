# traffic and instruction counts of the pass with tiled and with row-major phase planes
S1="FETCH_SIZE"
S2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
BENCH_ARGS="--splits 1 --inner 1 --no-tme --no-e2e --no-preset-exact --no-streams-leg" bash profiles/pmc.sh pmc_rows "$S1" "$S2" | grep "me_kernel<\(8, 64\|16, 256\|64, 1024\|128\)\|tq_kernel\|subpel"
BENCH_ARGS="--splits 1 --inner 1 --no-tme --no-e2e --no-preset-exact --no-streams-leg --fused 8" bash profiles/pmc.sh pmc_tiled "$S1" "$S2" | grep "me_kernel<\(8, 64\|16, 256\|64, 1024\|128\)\|tq_kernel\|subpel"
