"""LPInputReader: the reference's text format -> LPStandardForm (LPInputReader.java:19-224).

    max | min
    <objective>            e.g.  x1 + 2x2 - 0.5*x3
    <constraint> ...       e.g.  x1 + x2 <= 4 ,  2x1 - x3 >= 1 ,  x1 + x3 = 2

`>=` rows are negated (:191-197), `=` rows become two `<=` rows (:198-207), variables first seen in a
constraint are appended with objective coefficient 0 (:172-178), short rows are zero-padded (:215-223).
Host-side plumbing only (string parsing); coefficients are parsed exactly (decimal.Decimal, the analogue of
`new BigDecimal(text)`) and kept in `form.exact`, the fp64 arrays the device consumes are rounded from them.
"""
import os
import re
from decimal import Decimal

import numpy as np

from .errors import LPException
from .lp_standard_form import LPStandardForm

# LPInputReader.java:26-30
_OBJECTIVE = re.compile(r"^((\s*[+-]?\s*\d*\.?\d*)\*?([a-zA-Z]+\d*))+\s*$", re.ASCII)
_CONSTRAINT = re.compile(
    r"^((\s*[+-]?\s*\d*\.?\d*)\*?([a-zA-Z]+\d*))+\s*(=|==|<=|>=)\s*(-?\s*\d+(\.\d+)?)\s*$", re.ASCII)
_TOKEN = re.compile(r"(([+-]?\s*\d*\.?\d*)\*?([a-zA-Z]+\d*))", re.ASCII)
_WS = re.compile(r"\s+")


def _negate(x):
    """BigDecimal.negate(): exact (Python's unary minus would round to the context precision), no -0."""
    return Decimal(0) if x == 0 else x.copy_negate()


def _coefficient(text):
    t = _WS.sub("", text)
    if t == "" or t == "+":
        t = "1"
    elif t == "-":
        t = "-1"
    try:
        return Decimal(t)
    except Exception:
        raise ValueError("NumberFormatException: %r" % t)   # what `new BigDecimal(".")` would throw


class LPInputReader:
    def __init__(self):
        self._reload()

    def _reload(self):                                          # LPInputReader.java:41-49
        self.A, self.b, self.c = [], [], []
        self.variables, self.coefficients = {}, {}
        self.num_vars = 0
        self.num_ineq = 0

    # ---- public API ----------------------------------------------------------------------------------
    def read_lp(self, source):
        """readLP(File) when `source` is a path to an existing file or an os.PathLike, else readLP(String)."""
        if isinstance(source, os.PathLike) or (isinstance(source, str) and "\n" not in source and os.path.exists(source)):
            return self.read_lp_file(source)
        return self.read_lp_string(source)

    readLP = read_lp

    def read_lp_file(self, path):                               # LPInputReader.java:52-93
        if path is None:
            raise ValueError("IllegalArgumentException")        # @NotNull
        if not os.path.isfile(path) or not os.access(path, os.R_OK):
            raise ValueError("IllegalArgumentException")        # :54-61
        self._reload()
        constraint_counter = 0
        with open(path, "r") as f:
            lines = f.read().splitlines()
        if not lines:
            raise LPException("Input file is empty")            # :67-70
        maximized = self._max_min(lines[0])
        objective = lines[1] if len(lines) > 1 else None
        if objective is None:
            raise ValueError("IllegalArgumentException")        # @NotNull objective
        self.c = self._objective(objective)
        for line in lines[2:]:
            if line.strip() != "":                              # !StringUtils.isBlank
                self._constraint(line)
                constraint_counter += 1
            elif constraint_counter > 0:
                break                                           # only the FIRST block is consumed (:76-81)
            else:
                raise LPException("No constraints in the input file")
        return self._finish(maximized)

    def read_lp_string(self, lp):                               # LPInputReader.java:96-114
        if lp is None:
            raise ValueError("IllegalArgumentException")
        self._reload()
        lines = lp.split("\n")
        while lines and lines[-1] == "":                        # String.split drops trailing empty strings
            lines.pop()
        if len(lines) < 3:
            raise LPException("Incomplete lp")
        maximized = self._max_min(lines[0])
        self.c = self._objective(lines[1])
        for line in lines[2:]:
            self._constraint(line)
        return self._finish(maximized)

    # ---- pieces --------------------------------------------------------------------------------------
    @staticmethod
    def _max_min(text):                                         # :117-128
        t = text.strip().lower()
        if t == "min":
            return False
        if t == "max":
            return True
        raise LPException("Incorrect max/min parameter")

    def _objective(self, objective):                            # :131-155
        if not _OBJECTIVE.fullmatch(objective):
            raise LPException("Can't recognize objective")
        out = []
        for i, tok in enumerate(_TOKEN.finditer(objective)):
            name = tok.group(3)
            self.variables[i] = name
            self.coefficients[name] = i
            out.append(_coefficient(tok.group(2).strip()))
        self.num_vars = len(self.variables)
        return out

    def _constraint(self, constraint):                          # :158-213
        mt = _CONSTRAINT.search(constraint)
        if not mt:
            raise LPException("Can't recognize constraint")
        row = [Decimal(0)] * self.num_vars
        for tok in _TOKEN.finditer(constraint):
            name = tok.group(3)
            if name not in self.coefficients:                   # late-appearing variable (:172-178)
                self.variables[self.num_vars] = name
                self.coefficients[name] = self.num_vars
                self.num_vars += 1
                row.append(None)
                self.c.append(Decimal(0))
            row[self.coefficients[name]] = _coefficient(tok.group(2))
        sign = mt.group(4).strip()
        number = Decimal(_WS.sub("", mt.group(5)))
        if sign == ">=":                                        # :191-197
            self.A.append([_negate(x) for x in row])
            self.b.append(_negate(number))
            self.num_ineq += 1
        elif sign in ("=", "=="):                               # :198-207
            self.A.append(row)
            self.A.append([_negate(x) for x in row])
            self.b.append(number)
            self.b.append(_negate(number))
            self.num_ineq += 2
        else:
            self.A.append(row)
            self.b.append(number)
            self.num_ineq += 1

    def _finish(self, maximized):
        for row in self.A:                                      # normalizeConstraintMatrix :215-223
            row.extend([Decimal(0)] * (self.num_vars - len(row)))
        m, n = self.num_ineq, self.num_vars
        A = np.array([[float(x) for x in r] for r in self.A], dtype=np.float64).reshape(m, n)
        b = np.array([float(x) for x in self.b], dtype=np.float64)
        c = np.array([float(x) for x in self.c], dtype=np.float64)
        form = LPStandardForm(A, b, c, dict(self.variables), dict(self.coefficients), m, n, maximized)
        form.exact = {"A": [list(r) for r in self.A], "b": list(self.b), "c": list(self.c)}
        return form
