# r04: the whole -m gpu suite and the driver's bench command on the build of the commit
python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -16
python bench.py > gpurun_out/r04_final_bench.json 2> gpurun_out/r04_final_bench.err; echo "bench rc=$?"; grep -v amdgpu.ids gpurun_out/r04_final_bench.err | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
