"""Invalid 5-tuples through the one-shot entry points: every call must come back with an error (an exception of the
Python mirror), never crash, hang or return garbage silently."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_package
pkg = load_package()
U = pkg.umfpack
good = pkg.Matrix(3, 3, [0, 2, 3, 4], [0, 2, 1, 2], [1.0, 2.0, 3.0, 4.0])
bads = {
    "row index out of range": pkg.Matrix(3, 3, [0, 2, 3, 4], [0, 5, 1, 2], [1.0, 2.0, 3.0, 4.0]),
    "negative row index": pkg.Matrix(3, 3, [0, 2, 3, 4], [0, -1, 1, 2], [1.0, 2.0, 3.0, 4.0]),
    "decreasing pointers": pkg.Matrix(3, 3, [0, 3, 2, 4], [0, 2, 1, 2], [1.0, 2.0, 3.0, 4.0]),
    "pointers start at 1": pkg.Matrix(3, 3, [1, 2, 3, 4], [0, 2, 1, 2], [1.0, 2.0, 3.0, 4.0]),
    "last pointer beyond the arrays": pkg.Matrix(3, 3, [0, 2, 3, 9], [0, 2, 1, 2], [1.0, 2.0, 3.0, 4.0]),
}
ops = {
    "mulV": lambda M: pkg.mulV(M, np.ones(3)),
    "mulVT": lambda M: pkg.mulVT(M, np.ones(3)),
    "mulM": lambda M: pkg.mulM(M, np.ones((3, 2))),
    "mm left": lambda M: pkg.mm(M, good),
    "mm right": lambda M: pkg.mm(good, M),
    "lin": lambda M: pkg.lin(1.0, M, 1.0, good),
    "transpose": lambda M: pkg.transpose(M),
    "kronecker": lambda M: pkg.kronecker(M, good),
    "takeDiag": lambda M: pkg.takeDiag(M),
    "hcat": lambda M: pkg.hcat([M, good]),
    "analyze": lambda M: U.analyze(M),
    "solve": lambda M: U.solve(M, np.ones(3)),
}
failures = 0
for bname, B in bads.items():
    for oname, op in ops.items():
        try:
            r = op(B)
            print("NOT REFUSED: %s with %s -> %r" % (oname, bname, r)); failures += 1
        except Exception as e:
            pass
    # complex variants
    Z = pkg.Matrix(B.ncols, B.nrows, B.pointers, B.indices, B.values.astype(np.complex128))
    for oname, op in (("mulV z", lambda M: pkg.mulV(M, np.ones(3, dtype=complex))), ("mm z", lambda M: pkg.mm(M, M)),
                      ("lin z", lambda M: pkg.lin(1j, M, 1.0, M)), ("analyze z", lambda M: U.analyze(M))):
        try:
            r = op(Z)
            print("NOT REFUSED: %s with %s" % (oname, bname)); failures += 1
        except Exception:
            pass
for name, args in (("compress row", (3, 3, [0, 7], [0, 1], [1.0, 2.0])), ("compress col", (3, 3, [0, 1], [0, -2], [1.0, 2.0]))):
    try:
        pkg.compress(*args); print("NOT REFUSED:", name); failures += 1
    except Exception:
        pass
print("invalid_inputs: %d not refused" % failures)
sys.exit(1 if failures else 0)
