# FETCH_SIZE calibration (run through gpurun from the repo root): profiles/micro/fetch_calib.sh
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib profiles/micro/fetch_calib.hip
rm -rf gpurun_out/fetch_calib; timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/fetch_calib -- /tmp/fetch_calib
python profiles/summarize_pmc.py gpurun_out/fetch_calib | tee gpurun_out/r03_fetch_calib.txt
rm -rf gpurun_out/fetch_calib
