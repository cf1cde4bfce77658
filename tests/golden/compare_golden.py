"""compare_golden.py <oracle.npz> <reference.npz> -- diff of the oracle-generated golden vectors against the ones the
real CompressedSensing.jl produced (make_golden_reference.jl).  Supports must be identical; coefficients within the
north_star tolerance (1e-6 relative); selection orders identical where both files carry one.  Exit code 0 = the oracle is
pinned by the reference on every committed case."""
import sys

import numpy as np

a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
bad = 0
for name in (str(n) for n in a["names"]):
    ia, ib = a[name + ".idx"], b[name + ".idx"]
    va, vb = a[name + ".val"], b[name + ".val"]
    oa, ob = a[name + ".order"], b[name + ".order"]
    ok = np.array_equal(ia, ib)
    if ok and np.all(np.isfinite(va)):
        ok = np.allclose(va, vb, rtol=1e-6, atol=1e-6 * max(1e-300, float(np.abs(va).max()) if len(va) else 0.0))
    if ok and len(oa) and len(ob):
        ok = np.array_equal(oa, ob)
    print(("ok      " if ok else "MISMATCH") + " " + name)
    bad += not ok
print(f"{bad} mismatching case(s)")
sys.exit(1 if bad else 0)
