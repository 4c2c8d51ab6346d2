# r04: ThreadedME chain kernels, LDS cost slice and workgroups per CU: the producer leg of bench.py (1920x1080 P picture, presets medium / slow / slower) with variant libraries
#   release: XH_COST_R 512 (65 KB of MVD cost slices per 8-lane-group workgroup: two workgroups per CU), HEX / UMH built for 2 workgroups per CU, STAR for 4 (128 registers)
#   xc_b: cost slice +-256, HEX / UMH for 3;  xc_c: +-128, HEX / UMH for 3;  xc_d: +-256 only;  xc_e: +-256, every search for 3
export TMPDIR=/tmp
for v in release xc_b xc_c xc_d xc_e release; do
  if [ $v = release ]; then unset X265HIP_LIBDIR; else export X265HIP_LIBDIR=$GRAFT_REPO_ROOT/x265-mod-by-patman_amd/$v; fi
  timeout 300 python bench.py --leg tme_producer > gpurun_out/r04_chain_$v.json 2> gpurun_out/r04_chain_$v.err
  python - gpurun_out/r04_chain_$v.json $v <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], {k: (v["ms"], v.get("pictures_per_s_4_threads")) for k,v in d["presets"].items()})
except Exception as e: print(sys.argv[2], "failed", e)
PY
done
