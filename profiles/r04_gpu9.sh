# r04: --hme with diamond / exhaustive levels: kernel against the oracle, the encode against the CPU lookahead
python -m pytest tests/test_lookahead_gpu.py tests/test_e2e_la_gpu.py -q -x 2>&1 | tail -5
