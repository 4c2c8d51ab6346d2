"""The drop-in boundary: the slot offsets in include/x265hip.h / binding.py must equal offsetof() in the
REAL reference struct (source/common/primitives.h:239-432), and the built libraries must export every
symbol the header declares (no compute calls here -- this runs without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

from depths import DEPTHS

import x265hip
from x265hip import binding as B
from refproc import RefProc, ref_available

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def expected_layout():
    pu = lambda i, s: B.OFF_PU + (i * B.PU_PTRS + B.PU_SLOT[s]) * 8  # noqa: E731
    cu = lambda i, s: B.OFF_CU + (i * B.CU_PTRS + B.CU_SLOT[s]) * 8  # noqa: E731
    ch = B.OFF_CHROMA
    cpu = lambda s: ch + B.CHROMA_PU_SLOT[s] * 8  # noqa: E731
    ccu0 = ch + B.NUM_PU * B.CHROMA_PU_PTRS * 8
    ccu = lambda s: ccu0 + B.CHROMA_CU_SLOT[s] * 8  # noqa: E731
    v = [pu(0, "sad"), pu(1, "sad")]
    v += [pu(0, s) for s in ("sad", "sad_x3", "sad_x4", "ads", "satd", "luma_hpp", "luma_hps", "luma_vpp", "luma_vps",
                             "luma_vsp", "luma_vss", "luma_hvpp", "pixelavg_pp", "addAvg", "copy_pp", "convert_p2s")]
    v += [cu(0, "dct"), cu(1, "dct")]
    v += [cu(0, s) for s in ("dct", "idct", "standard_dct", "lowpass_dct", "calcresidual", "sub_ps", "add_ps", "blockfill_s",
                             "copy_cnt", "count_nonzero", "cpy2Dto1D_shl", "cpy2Dto1D_shr", "cpy1Dto2D_shl", "cpy1Dto2D_shr",
                             "copy_sp", "copy_ps", "copy_ss", "copy_pp", "var", "sse_pp", "sse_ss", "psy_cost_pp", "ssd_s",
                             "sa8d", "transpose", "intra_pred_allangs", "intra_filter", "intra_pred")]
    v += [B.SCALAR_OFF[s] for s in ("dst4x4", "idst4x4", "quant", "nquant", "dequant_scaling", "dequant_normal", "denoiseDct",
                                    "scale1D_128to64", "scale2D_64to32", "weight_sp", "weight_pp")]
    v += [ch, ch + B.CHROMA_BYTES, ch + B.CHROMA_PU_PTRS * 8]
    v += [cpu(s) for s in ("satd", "filter_vpp", "filter_vps", "filter_vsp", "filter_vss", "filter_hpp", "filter_hps",
                           "addAvg", "copy_pp", "p2s")]
    v += [ccu0, ccu0 + B.CHROMA_CU_PTRS * 8]
    v += [ccu(s) for s in ("sa8d", "sse_pp", "sub_ps", "add_ps", "copy_ps", "copy_sp", "copy_ss", "copy_pp")]
    v += [B.SCALAR_OFF["extendRowBorder"], B.SCALAR_OFF["frameInitLowres"], B.SCALAR_OFF["frameInitLowerRes"]]
    v += [B.SCALAR_OFF[s] for s in ("propagateCost", "fix8Unpack", "fix8Pack", "integral_initv", "integral_inith")]
    v += [B.SCALAR_OFF[s] for s in ("saoCuStatsBO", "saoCuStatsE0", "saoCuStatsE1", "saoCuStatsE2", "saoCuStatsE3")] + [B.SIZEOF_TABLE]
    return v


@pytest.mark.parametrize("depth", DEPTHS)
def test_slot_offsets_match_reference_struct(depth):
    if not ref_available(depth):
        pytest.skip("reference binary not built here")
    r = RefProc(depth)
    try:
        got = np.frombuffer(r.call("layout")[0], np.int32).tolist()
        info = r.call("info")
        assert RefProc.i32(info[0]) == depth and RefProc.i32(info[1]) == B.SIZEOF_TABLE
        assert RefProc.i32(info[2]) == (4 if depth == 8 else 8)     # sizeof(sse_t), common.h:145-149
    finally:
        r.close()
    assert got == expected_layout()


def test_header_constants_agree_with_binding():
    h = open(os.path.join(ROOT, "include", "x265hip.h")).read()
    d = dict(re.findall(r"#define (X265HIP_\w+) (-?\d+)", h))
    assert int(d["X265HIP_SIZEOF_TABLE"]) == B.SIZEOF_TABLE
    assert (int(d["X265HIP_OFF_CU"]), int(d["X265HIP_CU_PTRS"]), int(d["X265HIP_PU_PTRS"])) == (B.OFF_CU, B.CU_PTRS, B.PU_PTRS)
    assert (int(d["X265HIP_OFF_CHROMA"]), int(d["X265HIP_CHROMA_BYTES"])) == (B.OFF_CHROMA, B.CHROMA_BYTES)
    for name, key in (("dst4x4", "DST4X4"), ("quant", "QUANT"), ("nquant", "NQUANT"), ("dequant_normal", "DEQUANT_NORMAL"),
                      ("dequant_scaling", "DEQUANT_SCALING"), ("weight_sp", "WEIGHT_SP"), ("weight_pp", "WEIGHT_PP"),
                      ("scale2D_64to32", "SCALE2D_64TO32"), ("denoiseDct", "DENOISEDCT"), ("extendRowBorder", "EXTENDROWBORDER"), ("frameInitLowres", "FRAMEINITLOWRES"),
                      ("frameInitLowerRes", "FRAMEINITLOWERRES"), ("propagateCost", "PROPAGATECOST"), ("fix8Unpack", "FIX8UNPACK"), ("fix8Pack", "FIX8PACK"),
                      ("integral_initv", "INTEGRAL_INITV"), ("integral_inith", "INTEGRAL_INITH"), ("saoCuStatsBO", "SAOCUSTATSBO")):
        assert int(d["X265HIP_OFF_" + key]) == B.SCALAR_OFF[name]


@pytest.mark.parametrize("depth", DEPTHS)
def test_library_exports_every_declared_symbol(depth):
    x265hip.build_libraries()
    lib = x265hip.HipLib(depth, fill_table=False).lib
    names = set()
    for hdr in sorted(os.listdir(os.path.join(ROOT, "include"))):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        names |= set(re.findall(r"\b(x265hip_\w+)\s*\(", text))
    assert len(names) >= 15
    for n in sorted(names):
        assert hasattr(lib, n), "%s declared in include/ but not exported by %s" % (n, B.lib_path(depth))
    assert lib.x265hip_bit_depth() == depth
    assert lib.x265hip_abi_check(ctypes.c_size_t(B.SIZEOF_TABLE), depth) == 0
    assert lib.x265hip_abi_check(ctypes.c_size_t(B.SIZEOF_TABLE), 12 if depth != 12 else 10) != 0      # another depth's table
    assert lib.x265hip_abi_check(ctypes.c_size_t(100), depth) != 0


@pytest.mark.parametrize("depth", DEPTHS)
def test_library_exports_nothing_but_the_c_abi(depth):
    """-fvisibility=hidden + csrc/exports.map: every defined dynamic symbol is an x265hip_* C entry point the headers declare -- no mangled C++ (runtime classes,
    launcher functions, device stubs, template instantiations of the standard library)."""
    import subprocess
    x265hip.build_libraries()
    out = subprocess.check_output(["nm", "-D", "--defined-only", B.lib_path(depth)], text=True)
    syms = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
    assert len(syms) >= 100
    declared = set()
    for hdr in sorted(os.listdir(os.path.join(ROOT, "include"))):
        declared |= set(re.findall(r"\b(x265hip_\w+)\s*\(", open(os.path.join(ROOT, "include", hdr)).read()))
    bad = [s_ for s_ in syms if s_.startswith("_Z") or not s_.startswith("x265hip_")]
    assert not bad, "exported beside the C ABI: %s" % bad[:8]
    undeclared = sorted(set(syms) - declared)
    assert not undeclared, "exported but declared in no header of include/: %s" % undeclared[:8]


def test_no_cpu_fallback_without_gpu():
    """Without a device the table setup must fail loudly (never silently fall back)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = x265hip.HipLib(8, fill_table=False).lib
    table = (ctypes.c_void_p * (B.SIZEOF_TABLE // 8))()
    assert lib.x265hip_setup_primitives(table, 8, 0) == -2      # X265HIP_EDEVICE
    assert all(not p for p in table)


def test_python_mirrors_of_the_c_records_have_the_c_size(tmp_path):
    """The tests and the Python plumbing describe the C ABI's records with ctypes.  A field added to a header and not to its mirror makes the library read past the caller's
    record (found in round 6 by AddressSanitizer on the emulated library: x265hip_tme_args had grown by two ints).  sizeof of every mirrored record, from the headers through gcc,
    against ctypes.sizeof of the mirror."""
    import ctypes as C
    import subprocess
    import x265hip  # noqa: F401
    from x265hip_pkg import frame, host_batch, tme_host
    import deblock_util
    import test_ff_host_gpu
    import test_filters_batch_gpu
    pairs = [("x265hip_tme_args", frame.TmeArgs), ("x265hip_tme_ref", frame.TmeRef), ("x265hip_la_hme", frame.LaHme), ("x265hip_batch_desc", host_batch.BatchDesc),
             ("x265hip_tme_host_ref", tme_host.HostRef), ("x265hip_tme_picture_desc", tme_host.PictureDesc), ("x265hip_deblock_pic", deblock_util.DeblockPic),
             ("x265hip_ff_picture_desc", test_ff_host_gpu.FfDesc), ("x265hip_deblock_job", test_filters_batch_gpu.Job)]
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "x265hip.h"\n#include "x265hip_frame.h"\n#include "x265hip_ctx.h"\nint main(void) {\n' +
                   "".join('    printf("%s %%zu\\n", sizeof(%s));\n' % (n, n) for n, _ in pairs) + "    return 0;\n}\n")
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    c_sizes = dict(line.split() for line in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    wrong = {n: (int(c_sizes[n]), C.sizeof(m)) for n, m in pairs if int(c_sizes[n]) != C.sizeof(m)}
    assert not wrong, "record: (C size, ctypes mirror's size) %s" % wrong
