"""Euler / Midpoint / RK4 (3/8 rule) with the interface of the reference's neural_dae/my_fixed_grid.py.

`method` names the formula the fused HIP kernel runs for the class; `_step_func` is the same formula for
the callback walk (user callables / autograd).  Callback convention (my_fixed_grid.py:16-17): ODE branch
`func(t0=, xt=, zt=, all_initial=)` when v0 is None, else `func(t0=, xt=, zt=, vt=, it=, all_initial=)`.
"""
from .my_solvers import FixedGridODESolver

_one_third = 1 / 3
_two_thirds = 2 / 3


def _rhs(func, z0, v0, i0, all_initial):
    """f(t, x) with the step's external inputs frozen."""
    if v0 is None:
        return lambda tt, xx: func(t0=tt, xt=xx, zt=z0, all_initial=all_initial)
    return lambda tt, xx: func(t0=tt, xt=xx, zt=z0, vt=v0, it=i0, all_initial=all_initial)


class Euler(FixedGridODESolver):
    order = 1
    method = "euler"

    def _step_func(self, func, t0, dt, t1, x0, z0=None, v0=None, i0=None, all_initial=None):
        f0 = _rhs(func, z0, v0, i0, all_initial)(t0, x0)
        return dt * f0, f0


class Midpoint(FixedGridODESolver):
    order = 2
    method = "midpoint"

    def _step_func(self, func, t0, dt, t1, x0, z0=None, v0=None, i0=None, all_initial=None):
        f = _rhs(func, z0, v0, i0, all_initial)
        half_dt = 0.5 * dt
        f0 = f(t0, x0)
        return dt * f(t0 + half_dt, x0 + f0 * half_dt), f0


class RK4(FixedGridODESolver):
    order = 4
    method = "rk4"

    def rk4_alt_step_func(self, func, t0, dt, t1, x0, z0=None, v0=None, i0=None, all_initial=None, f0=None, perturb=False):
        """3/8-rule increment (k1 + 3(k2+k3) + k4)*dt/8 (my_fixed_grid.py:38-51)."""
        f = _rhs(func, z0, v0, i0, all_initial)
        k1 = f(t0, x0) if f0 is None else f0
        k2 = f(t0 + dt * _one_third, x0 + dt * k1 * _one_third)
        k3 = f(t0 + dt * _two_thirds, x0 + dt * (k2 - k1 * _one_third))
        k4 = f(t1, x0 + dt * (k1 - k2 + k3))
        return (k1 + 3 * (k2 + k3) + k4) * dt * 0.125

    def _step_func(self, func, t0, dt, t1, x0, z0=None, v0=None, i0=None, all_initial=None):
        f0 = _rhs(func, z0, v0, i0, all_initial)(t0, x0)
        return self.rk4_alt_step_func(func=func, t0=t0, dt=dt, t1=t1, x0=x0, z0=z0, v0=v0, i0=i0,
                                      all_initial=all_initial, f0=f0), f0
