"""The host producer x265hip_tme_picture (include/x265hip_ctx.h) on synthetic pictures: the chain kernels (tme_chain.inc: a PU shape's entries inside one kernel, entries of a
level side by side, a placement per lane in the STAR rounds) must write the table the one-launch-per-stage path writes (kern_tme.hip, pinned to the reference's recorded
calls in test_tme_gpu.py and to its bitstreams in test_e2e_tme_gpu.py) -- record for record, P and B pictures, with the partition sets of presets medium and slow."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def run(depth, preset, kind, extra=(), **env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(HERE, "tme_producer_run.py"), str(depth), preset, kind] + [str(x) for x in extra], capture_output=True, text=True, env=e, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("table ")][-1].split()
    return line[1], int(line[2]), int(line[3])


@pytest.mark.gpu
@pytest.mark.parametrize("depth,preset,kind", [(8, "medium", "P"), (8, "slower", "B"), (10, "slow", "P"), (10, "medium", "B"), (10, "slower", "B")])
def test_chain_kernels_write_the_table_of_the_launch_path(depth, preset, kind):
    chains = run(depth, preset, kind)
    launches = run(depth, preset, kind, TME_RUN_FLAGS="1")
    packed = run(depth, preset, kind, TME_RUN_FLAGS="2")            # the small shapes in the batched kernels' lane packing (several PUs per wavefront)
    assert chains[1] > 1000, "hardly any record written: %s" % (chains,)
    if kind == "B" and preset == "slower":
        assert chains[2] > 0, "no bidirectional record in a B picture"
    assert chains == launches, "chain kernels %s != launch path %s" % (chains, launches)
    assert packed == launches, "packed chain kernels %s != launch path %s" % (packed, launches)


@pytest.mark.gpu
@pytest.mark.parametrize("depth,preset,kind,method,merange", [(8, "slower", "P", 0, 57),      # DIA
                                                              (10, "medium", "B", 5, 5),    # FULL (a small window: every position is costed)
                                                              (8, "medium", "P", 2, 24)])   # UMH
def test_other_search_methods_through_the_producer(depth, preset, kind, method, merange):
    chains = run(depth, preset, kind, extra=(method, merange))
    launches = run(depth, preset, kind, extra=(method, merange), TME_RUN_FLAGS="1")
    assert chains[1] > 1000 and chains == launches, "method %d: %s != %s" % (method, chains, launches)
