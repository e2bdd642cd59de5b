"""Round-3 goldens, generated from the REAL reference (/root/reference) in the build container -- data only, never source:

  g7_grad_<tag>.npz for tag in ode01_h128, ode01_h32, dae01_h128, dae01_h32: what the reference's `loss.backward()` produces through
  integrate_ODE / integrate_DAE at the scripts' argparse default --hidden 128 (neural_00_ODE_01_no_encode.py:245-246) and at
  --hidden 32 -- the widths the fused backward kernels K4f / K7w serve.  Same recipe, shapes (B=8, T=21, two events, per-trajectory
  clocks) and file layout as make_goldens_r2.py:g7, which this script calls with other model constructors.

    python tests/golden/make_goldens_r3.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens_r2 as r2  # noqa: E402


def cases(ode01, ode02, dae01, dae02):
    xd, zd, vd, idim = 8, 2, 2, 2
    return (("ode01_h128", 80, lambda z_: ode01.ODE_Model(xd, zd, 128)), ("ode01_h32", 81, lambda z_: ode01.ODE_Model(xd, zd, 32)),
            ("dae01_h128", 82, lambda z_: dae01.DAE_Model(xd, zd, vd, idim, 128)), ("dae01_h32", 83, lambda z_: dae01.DAE_Model(xd, zd, vd, idim, 32)))


if __name__ == "__main__":
    if not os.path.isdir(r2.REF):
        sys.exit("reference not mounted; goldens can only be regenerated in the build container")
    nd_, mods_ = r2.load_reference()
    r2.g7(nd_, mods_, cases)
