#!/bin/bash
# after the masked AMP kernels: (1) SQ_INSTS_VALU of the 8K preset-exact leg again (its kernels changed), (2) the bench line, (3) the search / batch-host tests that reach the
# masked kernels under the fence (end mode: the masked lanes read up to 16 pixels right of the PU -- inside the plane's margin; a read past the plane buffer would fault here)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
# (0) first what only this session can do: X265TME_AHEAD on the real producer (bitstream tests + A/B), a few minutes
timeout 500 bash profiles/r05_ahead_ab.sh > gpurun_out/r05_ahead_ab.log 2>&1; tail -15 gpurun_out/r05_ahead_ab.log
timeout 420 bash profiles/collect_preset_exact.sh r05_pe 4320p10_slower > gpurun_out/r05_pe.log 2>&1
if python -c "import json,sys; d=json.load(open('gpurun_out/r05_pe/preset_exact_valu.json')); sys.exit(0 if 'r05_pe' in d['4320p10_slower']['source'] else 1)"; then cp gpurun_out/r05_pe/preset_exact_valu.json profiles/preset_exact_valu.json; fi
tail -12 gpurun_out/r05_pe.log
timeout 420 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_final2_bench.json 2> gpurun_out/r05_final2_bench.err
tail -c 600 gpurun_out/r05_final2_bench.json
rm -f gpurun_out/fence_*.log
( timeout 420 tools/fence_run.sh end python -m pytest tests/test_me_gpu.py tests/test_host_batch_gpu.py -m gpu -q -p no:cacheprovider --timeout=400 -k "not every_pu and not beyond_4gb and not whole_4k" > /tmp/part.out 2>&1 )
echo "== fence end | test_me_gpu + test_host_batch_gpu: $(grep -E 'passed|failed' /tmp/part.out | tail -1) $(grep -c 'Memory access fault' /tmp/part.out) faults" > gpurun_out/r05_amp_fence.txt
tail -5 /tmp/part.out >> gpurun_out/r05_amp_fence.txt
rm -f gpurun_out/fence_*.log
cat gpurun_out/r05_amp_fence.txt
