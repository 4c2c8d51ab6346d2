"""tools/fence_report.py -- the reader of a fence-build crash (csrc/xh_fence.h): given the allocation log and the stderr of a run that died with a GPU page fault it names
the block the address belongs to, which side of it was overrun and the launches in flight.  Checked on the committed evidence of the round-5 find (profiles/) and on a
synthetic under-run; CPU only."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "fence_report.py")


def report(tmp_path, log, err):
    a, b = tmp_path / "fence.log", tmp_path / "stderr.txt"
    a.write_text(log)
    b.write_text(err)
    r = subprocess.run([sys.executable, TOOL, str(a), str(b)], capture_output=True, text=True)
    return r.returncode, r.stdout


def test_overrun_is_attributed_to_the_block_before_the_fault_address(tmp_path):
    log = ("[fence] pid 1 mode end align 16 sync 1\n"
           "[fence] alloc #0 0x7181c6031000..0x7181c6040000 (61440 B) mapped 0x7181c6031000..0x7181c6040000 tag torch\n"
           "[fence] alloc #1 0x71811eb01010..0x71811ebf1000 (983024 B) mapped 0x71811eb01000..0x71811ebf1000 tag xh_tme.cpp:304\n")
    err = ("Memory access fault by GPU node-2 (Agent handle: 0x58ab752d7480) on address 0x71811ebf1000. Reason: Unknown.\n"
           "[fence] launch #0 kern_planes.hip subpel_planes_kernel:428   <-- last\n")
    rc, out = report(tmp_path, log, err)
    assert rc == 0
    assert "block #1" in out and "983024 bytes" in out and "xh_tme.cpp:304" in out and "after the end" in out and "+0 bytes" in out
    assert "subpel_planes_kernel:428" in out and "<-- last" in out


def test_underrun_and_wild_pointer(tmp_path):
    log = "[fence] alloc #7 0x700000400000..0x700000400100 (256 B) mapped 0x700000400000..0x700000600000 tag xh_ctx.cpp:88\n"
    rc, out = report(tmp_path, log, "Memory access fault by GPU node-1 (Agent handle: 0x1) on address 0x7000003ff000. Reason: Page not present.\n")
    assert rc == 0 and "block #7" in out and "before the start" in out and "-4096 bytes" in out
    rc, out = report(tmp_path, log, "Memory access fault by GPU node-1 (Agent handle: 0x1) on address 0x123456000. Reason: Page not present.\n")
    assert rc == 0 and "wild pointer" in out
    rc, out = report(tmp_path, log, "clean exit\n")
    assert rc == 1 and "no memory fault" in out


def test_release_library_has_no_fence_code():
    """the release build's allocator is hipMalloc: none of the fence's strings may be in the shipped libraries"""
    import x265hip
    for depth in (8, 10):
        path = x265hip.lib_path(depth)
        if not os.path.exists(path) or os.environ.get("X265HIP_LIBDIR"):
            continue
        data = open(path, "rb").read()
        assert b"[fence] alloc" not in data and b"X265HIP_FENCE_LOG" not in data
