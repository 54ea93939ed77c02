"""Exception types of the reference (LPException.java:4, SolutionException.java:3) and the mapping from
lpx_status codes back to the exact class + message the reference throws (its tests assert on the text,
LPSolverSpec.groovy:162,176,187)."""
from . import _lib


class LPException(Exception):
    """lpsolver.LPException (checked exception in the reference)."""


class SolutionException(LPException):
    """lpsolver.SolutionException extends LPException."""


def raise_for_status(status):
    """Raise what the reference would have thrown for a non-OPTIMAL lpx_status."""
    if status == _lib.OPTIMAL:
        return
    msg = _lib.status_message(status)
    if status in (_lib.UNBOUNDED, _lib.AUX_UNBOUNDED, _lib.NO_DEGENERATE_PIVOT):
        raise SolutionException(msg)                 # LPSolver.java:105, :149, :193
    if status == _lib.INFEASIBLE:
        raise LPException(msg)                       # LPSolver.java:173
    if status == _lib.BAD_ARGUMENT:
        raise ValueError(_lib.last_error() or msg)   # IllegalArgumentException, LPState.java:288
    if status == _lib.RESTORE_INDEX_FAULT:
        raise IndexError(msg)                        # ArrayIndexOutOfBoundsException, LPSolver.java:231
    if status == _lib.DIVIDE_BY_ZERO:
        raise ZeroDivisionError(msg)                 # ArithmeticException, LPState.java:139
    if status == _lib.PIVOT_LIMIT:
        raise RuntimeError("pivot limit reached")
    raise RuntimeError("%s: %s" % (msg, _lib.last_error()))
