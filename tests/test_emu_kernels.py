"""The kernel SOURCES on the CPU: a fixed selection of the `-m gpu` tests against the emulated library of tests/emu (x265-mod-by-patman_amd/csrc compiled for the host against
tests/emu/hip/hip_runtime.h: a workgroup as fibers, the cross-lane operations, MFMA, LDS and atomics modelled -- tests/emu/README.md).  What runs here is the library's own code
(every table slot, the batched entry points, the ME / TQ batch pipeline, the lookahead, the filter producer in bands, the ThreadedME producer inside the compiled reference encoder)
against the oracle, the reference's golden vectors and the reference encoder's bitstream -- without a GPU.  It says nothing about speed, about races between wavefronts or about
the compiler's gfx950 code: the GPU tests proper stay the `-m gpu` run.  tests/emu/run_gpu_suite.sh runs everything that is not full-size (profiles/r06_emu_gpu_suite.txt)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"

pytestmark = pytest.mark.skipif(not os.path.exists(CLANG), reason="no host clang++ to compile the sources with")


@pytest.fixture(scope="module")
def emulated_library():
    subprocess.check_call(["make", "-s", "-j%d" % (os.cpu_count() or 2), "-C", EMU, "all"])
    for d in (8, 10):
        assert os.path.exists(os.path.join(EMU, "_build", "libx265hip_%d.so" % d))
    return os.path.join(EMU, "_build")


def run_gpu_tests(libdir, node_ids, jobs=4, timeout=900, order=None):
    env = dict(os.environ, X265HIP_EMU="1", X265HIP_LIBDIR=libdir)
    if order:                                   # the order in which the emulation gives ready work-items their turn: a result that depends on it is a race
        env["X265HIP_EMU_ORDER"] = order
    r = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-p", "no:cacheprovider", "-n", str(jobs)] + list(node_ids), cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=timeout)
    tail = r.stdout[-3000:] + r.stderr[-1000:]
    assert r.returncode == 0 and " passed" in r.stdout and "failed" not in r.stdout.splitlines()[-1], tail
    return r.stdout.splitlines()[-1]


def test_the_emulated_library_is_the_librarys_own_sources(emulated_library):
    """the emulated library exports the C ABI of include/*.h like the real one (same export map), and nothing in the package or the bench can reach it"""
    import ctypes as C
    import re
    lib = C.CDLL(os.path.join(emulated_library, "libx265hip_8.so"))
    declared = set()
    for hdr in os.listdir(os.path.join(ROOT, "include")):
        declared |= set(re.findall(r"\b(x265hip_\w+)\s*\(", open(os.path.join(ROOT, "include", hdr)).read()))
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, missing[:5]
    pkg = os.path.join(ROOT, "x265-mod-by-patman_amd")
    for path in [os.path.join(ROOT, "bench.py")] + [os.path.join(pkg, f) for f in os.listdir(pkg) if f.endswith(".py")] + [os.path.join(ROOT, "integration", f) for f in os.listdir(os.path.join(ROOT, "integration"))]:
        src = open(path).read()
        assert "X265HIP_EMU" not in src and "tests/emu" not in src, path + " must not know the emulation"


def test_kernels_against_the_oracle_and_the_golden_vectors(emulated_library):
    """SURVEY 8(a) a2-a25 and f2 / f4 on the CPU: every table slot against the oracle (test_hip_parity.py) and the reference's golden vectors (test_golden.py), the batched entry
    points (test_batch_api_gpu.py), the C++ host's batch -- phase planes, the ME pyramid in HEX and STAR, rect / AMP, P and B pictures, the choice among references, TQ -- against
    the oracle (three of test_host_batch_gpu.py's configurations), a lowres frame-cost batch, and this round's band form of the filter producer (two pictures interleaved through
    one producer; 4:2:0 with slices, 4:2:2, 4:4:4) against the oracle's row forms -- with the work-items of a workgroup taking their turns in a RANDOM order (a kernel whose
    result depended on the order would have an LDS exchange without its barrier).  (Everything that is not full-size: tests/emu/run_gpu_suite.sh.)"""
    print(run_gpu_tests(emulated_library, ["tests/test_hip_parity.py", "tests/test_golden.py", "tests/test_batch_api_gpu.py",
                                           "tests/test_host_batch_gpu.py::test_host_batch_matches_oracle[8-1-2-1-False-1-False-0]",
                                           "tests/test_host_batch_gpu.py::test_host_batch_matches_oracle[10-3-3-1-True-1-False-0]",
                                           "tests/test_host_batch_gpu.py::test_host_batch_matches_oracle[8-3-3-2-True-2-True-1]",
                                           "tests/test_lookahead_gpu.py::test_lookahead_batch_matches_oracle[size2-1-shift2-8]",
                                           "tests/test_ff_host_gpu.py::test_ff_picture_in_bands_of_ctu_rows[8-256-320-64-cut0-3-0-1-slices0]",
                                           "tests/test_ff_host_gpu.py::test_ff_picture_in_bands_of_ctu_rows[8-256-320-64-cut10-3-0-1-slices10]",
                                           "tests/test_ff_host_gpu.py::test_ff_picture_in_bands_of_ctu_rows[10-200-168-32-cut7-3-0-2-slices7]",
                                           "tests/test_ff_host_gpu.py::test_ff_picture_in_bands_of_ctu_rows[8-256-192-64-cut8-3-1-3-slices8]"], jobs=6, order="random:5"))


def test_the_producers_inside_the_reference_encoder(emulated_library):
    """f1 / f4 end to end on the CPU: the compiled reference encoder with the EMULATED library as its ThreadedME producer (one frame thread; this round's --intra-refresh windows;
    bands under frame threads) and as its filter producer (bands inside slices under frame threads) -- the bitstream of the encoder's own producers / filters"""
    print(run_gpu_tests(emulated_library, ["tests/test_e2e_tme_gpu.py::test_bitstream_identical_with_gpu_producer[8-args0]", "tests/test_e2e_tme_gpu.py::test_bitstream_identical_with_gpu_producer[8-args12]",
                                           "tests/test_e2e_tme_gpu.py::test_bitstream_identical_with_gpu_producer_under_frame_threads[8-args1]",
                                           "tests/test_e2e_ff_gpu.py::test_slices_under_frame_threads_go_through_the_producer_in_bands_of_their_own[args0-None]"], jobs=5))


def test_the_bench_line_on_a_few_ctus(emulated_library, tmp_path):
    """bench.py itself (the driver's contract: ONE JSON line with metric / value / roofline / cpu_baseline) on a 256x128 plumbing workload against the emulated library -- the
    script's plumbing and the CPU-baseline leg (the reference's C table at -O2 and at -O3 for the host's vector ISA, its results cross-checked against the library's) run without
    a GPU.  The numbers mean nothing (the emulation has no clock worth reading); the line's shape and the cross-check do."""
    import json
    env = dict(os.environ, X265HIP_EMU="1", X265HIP_LIBDIR=emulated_library, PYTHONPATH=EMU + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "plumbing_256x128", "--frames", "2", "--inner", "1", "--steps", "2", "--warmup", "1", "--no-streams-leg",
                        "--no-tme", "--no-preset-exact", "--no-e2e", "--cpu-ctus", "16"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "bench.py prints ONE JSON line"
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["unit"] == "Mpixels/s" and d["dtype"] == "u8" and d["config"]["workload"] == "plumbing_256x128"
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    cb = d["cpu_baseline"]
    if cb["kind"] == "reference":
        assert "identical" in cb["sample"] and "O2" in cb["builds"] and cb["cores"] >= 1
        assert all("failed" not in b for b in cb["builds"].values()), cb["builds"]


def test_the_drivers_smoke_entry_point(emulated_library):
    """__graft_entry__.smoke() -- what the driver runs on the GPU box before the bench: the table slots and the frame-level ME + TQ path against the oracle -- on the emulated library"""
    env = dict(os.environ, X265HIP_EMU="1", X265HIP_LIBDIR=emulated_library, PYTHONPATH=EMU + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "smoke ok" in r.stdout, r.stderr[-1500:]
