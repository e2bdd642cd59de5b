"""Drop-in for the reference's `neural_dae` package (same exported names, neural_dae/__init__.py:1-3)."""
from .neural_base import ODE_Curves_Sample, ODE_Event, DE_Func, ODE_Base
from .neural_base import DAE_Curves_Sample, DAE_Event, AE_Func, DAE_Base
from .my_fixed_grid import Euler, Midpoint, RK4
from .my_solvers import FixedGridODESolver, NotFusableError

__all__ = ["ODE_Curves_Sample", "ODE_Event", "DE_Func", "ODE_Base", "DAE_Curves_Sample", "DAE_Event", "AE_Func",
           "DAE_Base", "Euler", "Midpoint", "RK4", "FixedGridODESolver", "NotFusableError"]
