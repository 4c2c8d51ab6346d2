# r04: (1) the release evidence of the round: profiles/collect.sh on the headline configuration (stats, FETCH / WRITE, SQ passes)
#      (2) vector instructions of the full-pel part alone (experiment build, X265HIP_ME_DBG=4: stop behind the full-pel search) -> the sub-pel stages' share, for DESIGN's
#          account of the "vertical taps inside the search kernel" proposal
#      (3) SQ_INSTS_VALU of the preset-exact legs
export TMPDIR=/tmp
bash profiles/collect.sh r04_v1_2160p10 > gpurun_out/r04_v1_collect.log 2>&1; tail -30 gpurun_out/r04_v1_collect.log
for d in 0 4; do
  mkdir -p gpurun_out/r04_dbg$d
  X265HIP_ME_DBG=$d X265HIP_LIBDIR=$GRAFT_REPO_ROOT/x265-mod-by-patman_amd/exp timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/r04_dbg$d/sq -- python bench.py --steps 3 --warmup 1 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg --splits 1 --inner 1 > /dev/null 2> gpurun_out/r04_dbg$d/err.txt
  python profiles/summarize_pmc.py --valu gpurun_out/r04_dbg$d/sq "X265HIP_ME_DBG=$d, experiment build" > gpurun_out/r04_valu_dbg$d.json; cat gpurun_out/r04_valu_dbg$d.json
  rm -rf gpurun_out/r04_dbg$d
done
bash profiles/collect_preset_exact.sh r04_pe > gpurun_out/r04_pe_collect.log 2>&1; tail -40 gpurun_out/r04_pe_collect.log
