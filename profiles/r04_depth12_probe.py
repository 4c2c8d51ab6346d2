"""Probe of a 12-bit build: every slot family of the drop-in table (tests/cases.py) against the oracle at depth 12; prints the labels that differ."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from backends import Hip, Oracle
from cases import FAMILIES, run_case, same
depth = 12
rng = np.random.default_rng(0xBADC0DE + depth)
hip, ora = Hip(depth), Oracle(depth)
total = 0
for fam in sorted(FAMILIES):
    fails, n = [], 0
    for label, method, args in FAMILIES[fam](depth, rng):
        a = run_case(ora, method, args); b = run_case(hip, method, args)
        if not same(a, b): fails.append(label)
        n += 1
    total += len(fails)
    print(fam, n, ("FAIL %d: %s" % (len(fails), sorted(set(f.split()[0] for f in fails))[:40])) if fails else "ok", flush=True)
print("total failing cases", total)
