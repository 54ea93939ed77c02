"""LPStandardForm: the API input type of the reference (LPStandardForm.java:10-66):
    max/min  c.x   subject to   A x <= b,  x >= 0
with optional variable-name maps (`variables`: index -> name, `coefficients`: name -> index)."""
import ctypes as C

import numpy as np

from . import _lib


class LPStandardForm:
    def __init__(self, A, b, c, variables=None, coefficients=None, m=None, n=None, maximize=True):
        """Mirrors LPStandardForm(A, b, c, variables, coefficients, m, n, maximize)  (LPStandardForm.java:18-31)
        and the name-less constructor (:57-65) when both maps are None."""
        self.b = np.ascontiguousarray(np.asarray(b, dtype=np.float64)).reshape(-1)
        self.c = np.ascontiguousarray(np.asarray(c, dtype=np.float64)).reshape(-1)
        self.m = int(self.b.size if m is None else m)
        self.n = int(self.c.size if n is None else n)
        A = np.asarray(A, dtype=np.float64)
        if A.size != self.m * self.n:
            if self.m * self.n != 0:
                raise ValueError("LPStandardForm: A has %d entries, expected m*n = %d*%d" % (A.size, self.m, self.n))
            A = np.zeros((self.m, self.n))
        self.A = np.ascontiguousarray(A.reshape(self.m, self.n))
        self.variables = None if variables is None else dict(variables)
        self.coefficients = None if coefficients is None else dict(coefficients)
        self.maximize = bool(maximize)

    def has_variable_names(self):                               # LPStandardForm.java:154-156
        return self.variables is not None and self.coefficients is not None

    hasVariableNames = has_variable_names

    def get_dual(self, device=0):
        """LPStandardForm.getDual() (LPStandardForm.java:129-152): transpose A, swap b and c, flip max/min,
        name the dual variables x1..xm.  The transpose runs on the device (lpx_transpose)."""
        L = _lib.lib()
        m, n = self.m, self.n
        At = np.zeros((n, m), dtype=np.float64)
        if m and n:
            rc = L.lpx_transpose(m, n, self.A.ctypes.data_as(_lib.dp), n, At.ctypes.data_as(_lib.dp), m, device)
            if rc:
                raise RuntimeError("lpx_transpose: " + _lib.last_error())
        if self.has_variable_names():
            # the reference loops i = 1..n here although the dual has m variables (:139-142); reproduced
            variables = {i - 1: "x%d" % i for i in range(1, n + 1)}
            coefficients = {"x%d" % i: i - 1 for i in range(1, n + 1)}
            return LPStandardForm(At, self.c.copy(), self.b.copy(), variables, coefficients, n, m, not self.maximize)
        return LPStandardForm(At, self.c.copy(), self.b.copy(), None, None, n, m, not self.maximize)

    getDual = get_dual
