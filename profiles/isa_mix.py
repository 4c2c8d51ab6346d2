"""Static instruction mix of the kernels in a HIP object / shared library (gfx950 code object -> llvm-objdump), per kernel: how much of the vector work is the compare units
themselves (v_sad_*), how much is address arithmetic, alignment (v_alignbyte / v_alignbit: unaligned LDS / register windows), reductions (DPP), LDS and memory traffic.
Static counts (every instruction once, loops not weighted): what the code is made of, not what a run issues -- the counters of a run (SQ_INSTS_VALU ...) need the GPU.
usage: python profiles/isa_mix.py <file.o | lib.so> [kernel-name-substring ...]"""
import collections, os, re, subprocess, sys, tempfile

B = "/opt/rocm/lib/llvm/bin"


def disassemble(path):
    with tempfile.TemporaryDirectory() as t:
        fat, co = os.path.join(t, "fat.bin"), os.path.join(t, "k.co")
        subprocess.run([B + "/llvm-objcopy", "--dump-section=.hip_fatbin=" + fat, path], check=True)
        subprocess.run([B + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
        return subprocess.run([B + "/llvm-objdump", "-d", "--no-show-raw-insn", "-C", co], check=True, capture_output=True, text=True).stdout


def classify(op):
    if "sad_u" in op: return "compare units (v_sad_*)"
    if op.startswith(("v_alignbyte", "v_alignbit", "v_perm")): return "alignment / byte shuffles"
    if op.endswith("_dpp") or op.startswith(("v_readlane", "v_readfirstlane", "v_writelane", "ds_swizzle", "ds_bpermute", "ds_permute")): return "cross-lane (DPP, lane reads)"
    if op.startswith("v_mfma"): return "MFMA"
    if op.startswith("ds_"): return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "memory"
    if op.startswith(("v_cmp", "v_cndmask", "v_min", "v_max", "v_med3")): return "compare / select / min-max"
    if op.startswith(("v_add", "v_sub", "v_lshl", "v_lshr", "v_ashr", "v_mad", "v_mul", "v_and", "v_or", "v_xor", "v_bfe", "v_bfi", "v_mov", "v_lshl_add", "v_add_lshl", "v_pk_")): return "integer arithmetic / moves"
    if op.startswith("v_"): return "other vector"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "waits / nops"
    if op.startswith("s_"): return "scalar"
    return "other"


def main():
    text = disassemble(sys.argv[1])
    want = sys.argv[2:]
    kern, name = collections.OrderedDict(), None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            name = m.group(1); kern[name] = collections.Counter(); continue
        m = re.match(r"\s+([a-z_0-9]+)(\s|$)", line)
        if m and name:
            kern[name][m.group(1)] += 1
    for name, ops in kern.items():
        short = re.sub(r"\(anonymous namespace\)::|^void ", "", name)
        if want and not any(w in short for w in want):
            continue
        tot = sum(ops.values())
        if tot < 50:
            continue
        cls = collections.Counter()
        for o, c in ops.items():
            cls[classify(o)] += c
        vec = sum(c for k, c in cls.items() if k not in ("scalar", "waits / nops", "LDS", "memory", "other"))
        print("%s\n  %d instructions, %d vector ALU" % (short[:150], tot, vec))
        for k, c in cls.most_common():
            share = " (%.0f %% of the vector ALU instructions)" % (100.0 * c / vec) if k not in ("scalar", "waits / nops", "LDS", "memory", "other") and vec else ""
            print("    %-34s %6d%s" % (k, c, share))
        print("    most frequent: " + ", ".join("%s %d" % kv for kv in ops.most_common(8)))


if __name__ == "__main__":
    main()
