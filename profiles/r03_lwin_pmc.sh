# counters of the lower-level search kernels with and without the per-wavefront LDS window (release library against x265-mod-by-patman_amd/$1)
export BENCH_ARGS="--splits 1 --inner 1 --no-tme --no-e2e --no-preset-exact --no-streams-leg"
S1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES"
S2="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE TCP_TCP_LATENCY_sum"
S3="SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_IFETCH SQ_LDS_IDX_ACTIVE"
bash profiles/pmc.sh pmc_release "$S1" "$S2" "$S3" | grep "me_kernel<\(8, 64\|16, 256\|64, 1024\)"
X265HIP_LIBDIR=$GRAFT_REPO_ROOT/x265-mod-by-patman_amd/$1 bash profiles/pmc.sh pmc_$1 "$S1" "$S2" "$S3" | grep "me_kernel<\(8, 64\|16, 256\|64, 1024\)"
