"""ctypes loader for libx265hip_{8,10}.so (see include/x265hip.h)."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SIZEOF_TABLE = 18240
OFF_PU, PU_PTRS = 0, 19
OFF_CU, CU_PTRS = 3800, 73
OFF_CHROMA, CHROMA_BYTES, CHROMA_PU_PTRS, CHROMA_CU_PTRS = 7200, 2760, 12, 9
NUM_PU = 25

PU_SLOT = {n: i for i, n in enumerate(
    ["sad", "sad_x3", "sad_x4", "ads", "satd", "luma_hpp", "luma_hps", "luma_vpp", "luma_vps", "luma_vsp", "luma_vss",
     "luma_hvpp", "pixelavg_pp", "pixelavg_pp_aligned", "addAvg", "addAvg_aligned", "copy_pp", "convert_p2s",
     "convert_p2s_aligned"])}
CU_SLOT = {n: i for i, n in enumerate(
    ["dct", "idct", "standard_dct", "lowpass_dct", "calcresidual", "calcresidual_aligned", "sub_ps", "add_ps",
     "add_ps_aligned", "blockfill_s", "blockfill_s_aligned", "copy_cnt", "count_nonzero", "cpy2Dto1D_shl",
     "cpy2Dto1D_shr", "cpy1Dto2D_shl", "cpy1Dto2D_shl_aligned", "cpy1Dto2D_shr", "copy_sp", "copy_ps", "copy_ss",
     "copy_pp", "var", "sse_pp", "sse_ss", "psy_cost_pp", "ssd_s", "ssd_s_aligned", "sa8d", "transpose",
     "intra_pred_allangs", "intra_filter", "intra_pred"])}
SCALAR_OFF = {"dst4x4": 6720, "idst4x4": 6728, "quant": 6736, "nquant": 6744, "dequant_scaling": 6752,
              "dequant_normal": 6760, "denoiseDct": 6768, "scale1D_128to64": 6776, "scale2D_64to32": 6792,
              "saoCuStatsBO": 6888, "saoCuStatsE0": 6896, "saoCuStatsE1": 6904, "saoCuStatsE2": 6912, "saoCuStatsE3": 6920,
              "frameInitLowres": 6928, "frameInitLowerRes": 6936, "propagateCost": 6944, "fix8Unpack": 6952, "fix8Pack": 6960, "extendRowBorder": 6968,
              "integral_initv": 7104, "integral_inith": 7152, "weight_sp": 7016, "weight_pp": 7024}
CHROMA_PU_SLOT = {n: i for i, n in enumerate(
    ["satd", "filter_vpp", "filter_vps", "filter_vsp", "filter_vss", "filter_hpp", "filter_hps", "addAvg",
     "addAvg_aligned", "copy_pp", "p2s", "p2s_aligned"])}
CHROMA_CU_SLOT = {n: i for i, n in enumerate(
    ["sa8d", "sse_pp", "sub_ps", "add_ps", "add_ps_aligned", "copy_ps", "copy_sp", "copy_ss", "copy_pp"])}
LUMA_PU = [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (8, 4), (4, 8), (16, 8), (8, 16), (32, 16), (16, 32),
           (64, 32), (32, 64), (16, 12), (12, 16), (16, 4), (4, 16), (32, 24), (24, 32), (32, 8), (8, 32),
           (64, 48), (48, 64), (64, 16), (16, 64)]


def lib_path(depth):
    # X265HIP_LIBDIR: the experiment build of profiles/*.sh (make OUT=<dir>/ OBJ=<dir>/obj EXPERIMENTS=1), never set by the tests or the default bench
    return os.path.join(os.environ.get("X265HIP_LIBDIR", HERE), "libx265hip_%d.so" % depth)


def build_libraries(jobs=8):
    """Compile every HIP source for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-s", "-j%d" % jobs, "-C", HERE])


_runtime_pinned = False


def _pin_hip_runtime():
    """torch ships its own libamdhip64.so.7 (same SONAME as /opt/rocm's).  Whichever copy is mapped first serves
    every later user in the process, and torch stops seeing the GPU when the system copy got there first.  Inside
    Python we therefore always map torch's runtime before libx265hip (a C/C++ host simply uses the system one)."""
    global _runtime_pinned
    if _runtime_pinned:
        return
    _runtime_pinned = True
    try:
        with open("/proc/self/maps") as f:
            if "libamdhip64" in f.read():
                return                      # some copy is already mapped: never map a second one
    except OSError:
        pass
    try:
        import torch
        cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except ImportError:
        pass


class HipLib:
    """One loaded libx265hip_<depth>.so plus an EncoderPrimitives-sized table it has filled."""

    def __init__(self, depth, fill_table=True):
        path = lib_path(depth)
        if not os.path.exists(path):
            raise RuntimeError("x265hip: %s missing -- run `make -C %s` (no fallback exists)" % (path, HERE))
        self.depth = depth
        _pin_hip_runtime()
        self.lib = C.CDLL(path)
        self.lib.x265hip_last_error.restype = C.c_char_p
        if self.lib.x265hip_bit_depth() != depth:
            raise RuntimeError("x265hip: library depth mismatch")
        self.table = (C.c_void_p * (SIZEOF_TABLE // 8))()
        if fill_table:
            self.check(self.lib.x265hip_abi_check(C.c_size_t(SIZEOF_TABLE), depth))
            self.check(self.lib.x265hip_setup_primitives(C.byref(self.table), depth, 0))

    def check(self, rc):
        if rc != 0:
            raise RuntimeError("x265hip error %d: %s" % (rc, (self.lib.x265hip_last_error() or b"").decode()))

    # ---- table access ----
    def _fn(self, index, restype, argtypes):
        addr = self.table[index]
        if not addr:
            raise NotImplementedError("table slot %d is NULL" % index)
        return C.CFUNCTYPE(restype, *argtypes)(addr)

    def pu(self, w, h, name, restype, argtypes):
        return self._fn(OFF_PU // 8 + LUMA_PU.index((w, h)) * PU_PTRS + PU_SLOT[name], restype, argtypes)

    def cu(self, n, name, restype, argtypes, extra=0):
        i = {4: 0, 8: 1, 16: 2, 32: 3, 64: 4}[n]
        return self._fn(OFF_CU // 8 + i * CU_PTRS + CU_SLOT[name] + extra, restype, argtypes)

    def scalar(self, name, restype, argtypes, extra=0):
        return self._fn(SCALAR_OFF[name] // 8 + extra, restype, argtypes)

    def chroma_pu(self, lw, lh, name, restype, argtypes, csp=1):
        base = (OFF_CHROMA + csp * CHROMA_BYTES) // 8
        return self._fn(base + LUMA_PU.index((lw, lh)) * CHROMA_PU_PTRS + CHROMA_PU_SLOT[name], restype, argtypes)

    def chroma_cu(self, n, name, restype, argtypes, csp=1):
        i = {4: 0, 8: 1, 16: 2, 32: 3, 64: 4}[n]
        base = (OFF_CHROMA + csp * CHROMA_BYTES) // 8 + NUM_PU * CHROMA_PU_PTRS
        return self._fn(base + i * CHROMA_CU_PTRS + CHROMA_CU_SLOT[name], restype, argtypes)
