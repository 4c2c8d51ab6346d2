"""TEST INFRASTRUCTURE ONLY: client for oracle/_ref/x265ref_{8,10} (see ref_driver.cpp).

The binary is the REAL reference C primitives (built from /root/reference sources by
oracle/Makefile) behind a tiny stdin/stdout protocol.  Used by tests and by
tests/golden/make_golden.py; never by the product path.
"""
import os
import struct
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def ref_binary(depth, variant=""):
    """variant "" = the -O2 build; "v3" / "v4" = the same sources at -O3 -march=x86-64-v3 / -v4 (oracle/Makefile)"""
    return os.path.join(HERE, "_ref", "x265ref_%d%s" % (depth, "_" + variant if variant else ""))


def ref_available(depth, variant=""):
    return os.access(ref_binary(depth, variant), os.X_OK)


def widest_variant(depth):
    """The widest -O3 build of the reference table this host can run (by /proc/cpuinfo), or None"""
    try:
        flags = set()
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags"):
                flags = set(line.split(":", 1)[1].split())
                break
    except OSError:
        return None
    v4 = {"avx512f", "avx512bw", "avx512cd", "avx512dq", "avx512vl"}
    v3 = {"avx2", "bmi1", "bmi2", "fma", "movbe", "f16c", "abm"}
    if v4 <= flags and v3 <= flags and ref_available(depth, "v4"):
        return "v4"
    if v3 <= flags and ref_available(depth, "v3"):
        return "v3"
    return None


class RefProc:
    def __init__(self, depth, variant=""):
        self.depth = depth
        self.pixel = np.uint8 if depth == 8 else np.uint16
        self.p = subprocess.Popen([ref_binary(depth, variant)], stdin=subprocess.PIPE, stdout=subprocess.PIPE)

    def close(self):
        if self.p:
            try:
                self.p.stdin.close()
                self.p.wait(timeout=5)
            except Exception:
                self.p.kill()
            self.p = None

    def __del__(self):
        self.close()

    def call(self, op, ints=(), bufs=()):
        """ints: python ints; bufs: numpy arrays / bytes. Returns list of bytes objects."""
        opb = op.encode()
        msg = [struct.pack("<I", len(opb)), opb, struct.pack("<I", len(ints))]
        msg.append(struct.pack("<%dq" % len(ints), *[int(i) for i in ints]))
        msg.append(struct.pack("<I", len(bufs)))
        for b in bufs:
            raw = b.tobytes() if isinstance(b, np.ndarray) else bytes(b)
            msg.append(struct.pack("<Q", len(raw)))
            msg.append(raw)
        self.p.stdin.write(b"".join(msg))
        self.p.stdin.flush()
        (n,) = struct.unpack("<I", self._read(4))
        if n == 0xFFFFFFFF:
            raise ValueError("x265ref: unknown op %r" % op)
        out = []
        for _ in range(n):
            (l,) = struct.unpack("<Q", self._read(8))
            out.append(self._read(l))
        return out

    def _read(self, n):
        chunks = []
        while n:
            c = self.p.stdout.read(n)
            if not c:
                raise RuntimeError("x265ref died")
            chunks.append(c)
            n -= len(c)
        return b"".join(chunks)

    # convenience decoders
    @staticmethod
    def i32(b):
        return int(np.frombuffer(b, np.int32)[0])

    @staticmethod
    def u32(b):
        return int(np.frombuffer(b, np.uint32)[0])

    @staticmethod
    def u64(b):
        return int(np.frombuffer(b, np.uint64)[0])
