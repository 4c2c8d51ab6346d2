"""x265hip_tme_schedule (host): the order of ThreadedME's PU stage inside a CTU -- slots, neighbour slots, partitions -- against the sequence of
Search::puMotionEstimation calls a reference encode made (tests/golden/tme_sched.npz, oracle/ref_tme.cpp kind 9), for presets without rectangles, with
rectangles, and with rectangles + AMP."""
import ctypes as C
import os

import numpy as np
import pytest

import x265hip

STEP = np.dtype([("part", "<i2"), ("cuSize", "<i2"), ("cuX", "<i2"), ("cuY", "<i2"), ("puOffset", "<i2"), ("finalIdx", "<i2"), ("neighbor", "<i2", 5), ("numPart", "<i2"),
                 ("pu", "<i2", (2, 4))])
assert STEP.itemsize == 40


@pytest.mark.parametrize("name,rect,amp,per_ctu", [("medium", 0, 0, 85), ("slow", 1, 0, 255), ("slower", 1, 1, 339)])
def test_schedule_is_the_references_call_order(name, rect, amp, per_ctu):
    lib = x265hip.HipLib(8, fill_table=False).lib
    n = lib.x265hip_tme_schedule(64, 8, rect, amp, None, 0)
    assert n == per_ctu
    steps = np.zeros(n, STEP)
    assert lib.x265hip_tme_schedule(64, 8, rect, amp, steps.ctypes.data_as(C.c_void_p), n) == n
    rows = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tme_sched.npz"))[name]
    for ctu in range(4):
        r = rows[rows[:, 1] == ctu]
        assert len(r) == n
        ox, oy = (ctu % 2) * 64, (ctu // 2) * 64
        for k in range(n):
            s, e = steps[k], r[k]
            area = 0 if s["cuSize"] == 64 else int(ox + s["cuX"] >= 32) + 2 * int(oy + s["cuY"] >= 32) + 1
            got = [int(s["part"]), int(s["cuSize"]), ox + int(s["cuX"]), oy + int(s["cuY"]), int(s["puOffset"]), area, int(s["finalIdx"])] + [int(v) for v in s["neighbor"]] + [int(s["numPart"])]
            assert got == [int(v) for v in e[2:15]], "CTU %d entry %d: schedule %s reference %s" % (ctu, k, got, list(e[2:15]))
            for p in range(int(s["numPart"])):
                assert [ox + int(s["pu"][p][0]), oy + int(s["pu"][p][1]), int(s["pu"][p][2]), int(s["pu"][p][3])] == [int(v) for v in e[15 + 4 * p:19 + 4 * p]], "CTU %d entry %d partition %d" % (ctu, k, p)
