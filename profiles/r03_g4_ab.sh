cd $GRAFT_REPO_ROOT
export X265HIP_LIBDIR=$PWD/x265-mod-by-patman_amd/exp
X265HIP_ME_VARIANT=24576 python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu -k "small_frames and 10-" 2>&1 | tail -3
for v in 0 8192 16384 24576 0; do
  X265HIP_ME_VARIANT=$v python bench.py --steps 8 --warmup 2 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant $v', d['value'], round(d['ms_per_step']/5,4), d['roofline']['all_kernels_ms'])"
done
