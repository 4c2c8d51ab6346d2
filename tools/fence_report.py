#!/usr/bin/env python3
"""Name the block and the kernel of a page fault caught by the fence build (csrc/xh_fence.h).

usage: fence_report.py <fence log> <stderr of the dead run>

The HSA runtime reports "Memory access fault by GPU node-N ... on address 0x...": the address is looked up among the blocks of the fence log (it falls into the unmapped
granule after or before one of them), and the "[fence] launch" lines the abort handler wrote name the kernels in flight (the last one when launches are waited for)."""
import re
import sys


def main():
    log, err = open(sys.argv[1]).read(), open(sys.argv[2]).read()
    m = re.search(r"Memory access fault by GPU.*?on address (0x[0-9a-f]+)", err, re.S)
    if not m:
        print("no memory fault in", sys.argv[2])
        return 1
    addr = int(m.group(1), 16)
    print("fault address 0x%x" % addr)
    best = None
    for a in re.finditer(r"\[fence\] alloc #(\d+) (0x[0-9a-f]+)\.\.(0x[0-9a-f]+) \((\d+) B\) mapped (0x[0-9a-f]+)\.\.(0x[0-9a-f]+) tag (.*)", log):
        lo, hi, mlo, mhi = (int(a.group(k), 16) for k in (2, 3, 5, 6))
        gran = 1 << 21
        if mlo - gran <= addr < mhi + gran:
            side = "after the end" if addr >= mhi else ("before the start" if addr < mlo else "inside the mapped range (a freed block?)")
            best = (a.group(1), lo, hi, int(a.group(4)), side, a.group(7), addr - hi if addr >= hi else addr - lo)
    if best:
        print("block #%s %#x..%#x (%d bytes), allocated at %s: the access is %s, %+d bytes from the block's %s" %
              (best[0], best[1], best[2], best[3], best[5], best[4], best[6], "end" if best[6] >= 0 else "start"))
    else:
        print("address is in no fenced block's guard (a wild pointer)")
        near = []
        for a in re.finditer(r"\[fence\] alloc #(\d+) (0x[0-9a-f]+)\.\.(0x[0-9a-f]+) \((\d+) B\) mapped (0x[0-9a-f]+)\.\.(0x[0-9a-f]+) tag (.*)", log):
            lo, hi = int(a.group(2), 16), int(a.group(3), 16)
            near.append((min(abs(addr - lo), abs(addr - hi)), a.group(1), lo, hi, int(a.group(4)), a.group(7)))
        for dist, idx, lo, hi, size, tag in sorted(near)[:4]:
            print("  nearest: block #%s %#x..%#x (%d bytes, %s): %+d bytes from its %s" % (idx, lo, hi, size, tag, addr - hi if addr >= hi else addr - lo, "end" if addr >= hi else "start"))
    for line in (err + log).splitlines():
        if line.startswith("[fence] launch #"):
            print(line)
    return 0


if __name__ == "__main__":
    sys.exit(main())
