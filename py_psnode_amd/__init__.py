"""py_psnode_amd -- MI355X (gfx950) native fixed-grid neural-ODE/DAE integrator.

Hot path: py_psnode_amd.fused -> libpsnode_hip.so (C ABI in include/psnode_hip.h, HIP kernels in csrc/).
Drop-in surface: py_psnode_amd.neural_dae mirrors the reference's `neural_dae` package;
`install_as_neural_dae()` registers it under that name so the four training scripts import it unchanged.
"""
import sys

__version__ = "0.1.0"


def install_as_neural_dae():
    """Make `import neural_dae` (and neural_dae.neural_base / my_solvers / my_fixed_grid) resolve to this package."""
    from . import neural_dae as nd
    from .neural_dae import my_fixed_grid, my_solvers, neural_base
    sys.modules["neural_dae"] = nd
    sys.modules["neural_dae.neural_base"] = neural_base
    sys.modules["neural_dae.my_solvers"] = my_solvers
    sys.modules["neural_dae.my_fixed_grid"] = my_fixed_grid
    return nd


def accelerate(model):
    """Encoders / decoders of a direct_encode model (the reference's own script classes included) onto the fused HIP row kernels;
    see py_psnode_amd.models.accelerate."""
    from .models import accelerate as _acc
    return _acc(model)

