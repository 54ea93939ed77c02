"""LPSolver: `BigDecimal LPSolver.solve(LPStandardForm) throws LPException` (LPSolver.java:78) over
liblpx.so.  Returns the objective as a decimal.Decimal with 6 fractional digits — the analogue of
`v.setScale(6, RoundingMode.HALF_UP)` (LPSolver.java:113) — and raises the reference's exceptions with the
reference's messages."""
import ctypes as C
from decimal import Decimal

import numpy as np

from . import _lib
from .errors import raise_for_status
from .java_compat import hashmap_key_order
from .lp_state import LPState


class SolveInfo:
    """What the reference only logs: pivot counts, phase-1 use, the unrounded objective, basis, x*."""

    def __init__(self, res, perm, x):
        self.status = res.status
        self.phase1_used = bool(res.phase1_used)
        self.objective = res.objective
        self.objective_text = res.objective_text.decode()
        self.pivots_phase1 = res.pivots_phase1
        self.pivots_phase2 = res.pivots_phase2
        self.x0_slot = res.x0_slot
        self.seconds_total = res.seconds_total
        self.seconds_pivots = res.seconds_pivots
        self.perm = perm
        self.x = x


class LPSolver:
    def __init__(self, device=0, max_pivots=-1, pricing="reference", devices=None, fused=None):
        """pricing="reference": the reference's first-positive rule (default, parity with the Java solver);
        pricing="dantzig": opt-in largest-coefficient rule (same optimum, ~10x fewer pivots, no pivot parity).
        devices=[d0, d1, ...]: solve with the row blocks of the tableau on several GPUs (lpx_solve_multi).
        fused=True: every update x - c*r as one fused multiply-add (LPX_OPT_FUSED; one binary rounding where the
        reference has two decimal ones, LPState.java:162 — checked against the oracle's fused instantiation);
        fused=False: product and difference rounded separately; None: the library's choice by size (fused from 0.5 GiB)."""
        self.fused = _lib.DEFAULT_FUSED if fused is None else bool(fused)
        self.device = int(device)
        self.devices = None if devices is None else [int(d) for d in devices]
        self.max_pivots = int(max_pivots)
        self.pricing = _lib.PRICING[pricing]
        self.last = None

    def _arith_options(self):
        return {} if self.fused is None else {"fused": int(self.fused)}

    def solve(self, st_form, restore_order=None):
        """LPSolver.solve(stForm).  Unlike the reference this never modifies `st_form` (the reference
        negates stForm.c in place for `min`, :86-89, and pivots inside stForm.A/b/c, :267)."""
        L = _lib.lib()
        m, n = st_form.m, st_form.n
        A = np.ascontiguousarray(st_form.A, dtype=np.float64)
        b = np.ascontiguousarray(st_form.b, dtype=np.float64)
        c = np.ascontiguousarray(st_form.c, dtype=np.float64)
        opts = _lib.SolveOptions()
        opts.device = self.device
        opts.has_variable_names = 1 if st_form.has_variable_names() else 0
        opts.max_pivots = self.max_pivots
        opts.pricing = self.pricing
        opts.fused = 0 if self.fused is None else (1 if self.fused else -1)
        order = None
        if restore_order is not None:
            order = np.ascontiguousarray(np.asarray(restore_order, dtype=np.int32))
        elif st_form.has_variable_names() and n > 0:
            order = self._key_set_order(st_form)
        if order is not None:   # an empty key set substitutes nothing (LPSolver.java:217 iterates zero names)
            keep = order if order.size else np.zeros(1, dtype=np.int32)
            opts.restore_order = keep.ctypes.data_as(_lib.ip)
            opts.restore_order_len = int(order.size)
        perm = np.zeros(n + m, dtype=np.int32)
        x = np.zeros(max(n, 1), dtype=np.float64)
        opts.perm_out = perm.ctypes.data_as(_lib.ip)
        opts.x_out = x.ctypes.data_as(_lib.dp)
        res = _lib.SolveResult()
        if self.devices is None:
            rc = L.lpx_solve(m, n, A.ctypes.data_as(_lib.dp) if A.size else None, max(n, 1),
                             b.ctypes.data_as(_lib.dp) if m else None, c.ctypes.data_as(_lib.dp) if n else None,
                             1 if st_form.maximize else 0, C.byref(opts), C.byref(res))
        else:
            dev = np.array(self.devices, dtype=np.int32)
            rc = L.lpx_solve_multi(m, n, A.ctypes.data_as(_lib.dp) if A.size else None, max(n, 1),
                                   b.ctypes.data_as(_lib.dp) if m else None, c.ctypes.data_as(_lib.dp) if n else None,
                                   1 if st_form.maximize else 0, C.byref(opts), dev.ctypes.data_as(_lib.ip), len(dev),
                                   C.byref(res))
        self.last = SolveInfo(res, perm[: n + m].copy(), x[:n].copy())
        if rc != _lib.OPTIMAL:
            raise_for_status(rc)
        return Decimal(res.objective_text.decode())

    @staticmethod
    def _key_set_order(form):
        """Iteration order of `initial.coefficients.keySet()` in restoreInitialLP (LPSolver.java:213-217) as
        variable indices.  Only the names PRESENT are visited — a named form from getDual() with m > n names
        just min(n, m) of its variables (LPStandardForm.java:139-142) and the reference substitutes only those.
        Keys enter the HashMap in index order (LPInputReader.processObjective/processConstraint)."""
        keys = sorted(form.coefficients, key=lambda k: form.coefficients[k])
        return np.array([form.coefficients[k] for k in hashmap_key_order(keys)
                         if 0 <= form.coefficients[k] < form.n], dtype=np.int32)

    # ---- the reference's public/package-private helpers around solve() -----------------------------------
    @staticmethod
    def min_in_b(b):
        """LPSolver.minInB (LPSolver.java:375-386): first index of the strict minimum of b, -1 if empty
        (the scan starts from 1e50, so values >= 1e50 are never selected)."""
        mn, idx = 1e50, -1
        for i, x in enumerate(np.asarray(b, dtype=np.float64).reshape(-1)):
            if mn > x:
                mn, idx = float(x), i
        return idx

    @staticmethod
    def get_name_for_x0(coefficients):
        """LPSolver.getNameForX0 (LPSolver.java:323-342): "x0", else "auxVar", else "auxVar<k>"."""
        if "x0" not in coefficients:
            return "x0"
        if "auxVar" not in coefficients:
            return "auxVar"
        i = 1
        while "auxVar%d" % i in coefficients:
            i += 1
        return "auxVar%d" % i

    @staticmethod
    def slack_names(coefficients, m):
        """The slack naming of convertIntoSlackForm (LPSolver.java:255-266): x<k> for the smallest unused k."""
        names, used, k = [], set(coefficients), 1
        while len(names) < m:
            nm = "x%d" % k
            if nm not in used:
                names.append(nm)
                used.add(nm)
            k += 1
        return names

    def convert_into_slack_form(self, st_form):
        """LPSolver.convertIntoSlackForm (LPSolver.java:248-272): device-resident LPState of the slack form.
        The reference aliases stForm's arrays and mutates its name maps; here both are copied."""
        if st_form.has_variable_names():
            variables = dict(st_form.variables)
            coefficients = dict(st_form.coefficients)
            for i, nm in enumerate(self.slack_names(coefficients, st_form.m)):
                variables[st_form.n + i] = nm
                coefficients[nm] = st_form.n + i
            return LPState(st_form.A, st_form.b, st_form.c, 0.0, variables, coefficients, st_form.m, st_form.n,
                           device=self.device, options=self._arith_options())
        return LPState(st_form.A, st_form.b, st_form.c, 0.0, None, None, st_form.m, st_form.n, device=self.device,
                       options=self._arith_options())

    def convert_into_aux_lp(self, st_form):
        """LPSolver.convertIntoAuxLP (LPSolver.java:283-321): extra column of -1, objective -x0, x0 named by
        getNameForX0 and placed in slot n; slacks named as in convertIntoSlackForm."""
        m, n = st_form.m, st_form.n
        auxA = np.hstack([np.asarray(st_form.A, dtype=np.float64).reshape(m, n), -np.ones((m, 1))])
        auxc = np.zeros(n + 1)
        auxc[n] = -1.0
        if st_form.has_variable_names():
            variables, coefficients = dict(st_form.variables), dict(st_form.coefficients)
        else:                                                  # addDefaultVariables (LPSolver.java:388-400)
            variables = {i: "x%d" % (i + 1) for i in range(n)}
            coefficients = {v: k for k, v in variables.items()}
        x0 = self.get_name_for_x0(coefficients)
        variables[n] = x0
        coefficients[x0] = n
        for i, nm in enumerate(self.slack_names(coefficients, m)):
            variables[n + 1 + i] = nm
            coefficients[nm] = n + 1 + i
        return LPState(auxA, st_form.b, auxc, 0.0, variables, coefficients, m, n + 1, device=self.device,
                       options=self._arith_options())

    def restore_initial_lp(self, aux_lp, initial, index_of_x0, restore_order=None):
        """LPSolver.restoreInitialLP (LPSolver.java:200-246), in place on the auxiliary LPState `aux_lp`, which
        becomes the restored m x n LPState (returned).  `initial` supplies c (and, with names, the keySet()
        iteration order of its `coefficients`)."""
        L = _lib.lib()
        n = initial.n
        c0 = np.ascontiguousarray(initial.c, dtype=np.float64)
        order = None
        if restore_order is not None:
            order = np.ascontiguousarray(np.asarray(restore_order, dtype=np.int32))
        elif initial.has_variable_names() and n > 0:
            order = self._key_set_order(initial)
        # an EMPTY key set substitutes nothing (LPSolver.java:217 iterates zero names): a non-NULL pointer with length 0,
        # as solve() passes it — NULL would select the default-name order over all n variables
        keep = None if order is None else (order if order.size else np.zeros(1, dtype=np.int32))
        rc = L.lpx_restore_initial_lp(aux_lp._h, c0.ctypes.data_as(_lib.dp), n, int(index_of_x0),
                                      None if keep is None else keep.ctypes.data_as(_lib.ip),
                                      0 if order is None else int(order.size))
        if rc:
            raise_for_status(rc)
        aux_lp.n = n                    # names stay keyed by variable id; x0's id simply no longer occurs in perm
        return aux_lp
