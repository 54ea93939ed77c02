"""linear_programming_solver_amd — MI355X-native dense simplex pivot engine.

Host-side mirror of the reference's operator interface for ONE path (the simplex pivot loop), over the
C ABI of liblpx.so (include/lpx.h): LPStandardForm, LPState {get_entering, get_leaving, pivot},
LPSolver.solve, LPInputReader.  There is no CPU fallback: without the HIP library every compute call raises.
"""
from ._lib import set_default_arithmetic  # noqa: F401
from .errors import LPException, SolutionException  # noqa: F401
from .lp_input_reader import LPInputReader  # noqa: F401
from .lp_multi import LPMulti  # noqa: F401
from .lp_solver import LPSolver  # noqa: F401
from .lp_standard_form import LPStandardForm  # noqa: F401
from .lp_state import LPState  # noqa: F401

__all__ = ["LPException", "SolutionException", "LPInputReader", "LPMulti", "LPSolver", "LPStandardForm", "LPState",
           "set_default_arithmetic"]
