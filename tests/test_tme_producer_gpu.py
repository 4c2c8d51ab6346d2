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


@pytest.mark.gpu
@pytest.mark.parametrize("depth,preset,kind,method,merange,step", [(8, "medium", "P", 1, 16, 1), (10, "slow", "B", 3, 24, 2), (8, "slower", "B", 3, 57, 1), (10, "medium", "P", 2, 16, 3)])
def test_bands_of_ctu_rows_on_growing_references_give_the_pictures_table(depth, preset, kind, method, merange, step):
    """ThreadedME under frame threads (threadedme.cpp:121-150): a picture goes through the producer in bands of CTU rows while its references are still being reconstructed --
    each band sees only the reference rows FrameEncoder::m_refLagRows releases for it (the rest of the host planes is garbage), planes are uploaded and phase-interpolated
    incrementally under their keys.  The table must be the one a single call on complete references writes (both with the window of m_refLagPixels, desc.frameThreads = 3)."""
    whole = run(depth, preset, kind, extra=(method, merange), TME_RUN_BANDS="whole", TME_RUN_HEIGHT="616")
    banded = run(depth, preset, kind, extra=(method, merange), TME_RUN_BANDS=str(step), TME_RUN_HEIGHT="616")
    one_thread = run(depth, preset, kind, extra=(method, merange), TME_RUN_HEIGHT="616")
    assert whole[1] > 1000 and banded == whole, "bands of %d rows: %s != %s" % (step, banded, whole)
    if merange > 24:
        assert whole != one_thread, "the frame-parallel window did not change a single record: is it modelled?"
