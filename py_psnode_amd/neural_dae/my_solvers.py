"""Fixed-grid solver base with the call surface of the reference's neural_dae/my_solvers.py.

`integrate_ODE` / `integrate_DAE` keep the reference's keyword names, tensor layout (time-major views in,
fresh contiguous [T,B,D] out) and error behaviour (my_solvers.py:11-29, 52-131).  When the right-hand
sides are the reference's ELU-MLPs on a HIP device the whole time loop runs in ONE fused HIP launch
(py_psnode_amd.fused -> libpsnode_hip.so); with autograd in play the call becomes a torch.autograd.Function
(fused forward kernel + one fused backward kernel, py_psnode_amd/autograd.py).  Arbitrary Python callbacks --
which no kernel can execute -- teacher-forced training and shapes no backward kernel covers are stepped through
the user's own callables by `_walk_*` below.

`solver.fused` selects the route: "auto" (default; fused whenever the call is fusable, and it then FAILS
LOUDLY if libpsnode_hip.so is missing -- never a silent substitute; a call on a HIP device that has to walk
says so once with a RuntimeWarning), "require" (raise if the call is not fusable), "off" (always the walk).
"""
import abc
import os
import warnings

import torch
import torch.nn as nn

from .. import fused as _fused
from .._lib import UnsupportedShapeError


class NotFusableError(RuntimeError):
    pass


def _autograd():
    from .. import autograd       # (imported on first use: autograd imports fused, which this module imports too)
    return autograd


class FixedGridODESolver(metaclass=abc.ABCMeta):
    order: int
    method: str = ""

    def __init__(self, step_size=None, grid_constructor=None, interp="linear"):
        # public attributes of the reference (my_solvers.py:13-18)
        self.step_size = step_size
        self.interp = interp
        self.enable_cal_time = False
        self.assert_time = 0
        self.cal_time = 0
        self.total_time = 0
        self.fused = os.environ.get("PSNODE_FUSED", "auto")
        self.kernel = os.environ.get("PSNODE_KERNEL", "auto")
        # Two events at one time stamp make the reference's jump_change_fn raise (neural_base.py:61); the fused event table would
        # take the first.  True: read the table kernel's duplicate flag back (one 4-byte D2H per call, skipped when the event list
        # has fewer than two entries or the stream is being captured into a HIP graph) and raise like the reference.
        self.check_events = os.environ.get("PSNODE_CHECK_EVENTS", "1") != "0"
        if step_size is not None and grid_constructor is not None:
            raise ValueError("step_size and grid_constructor are mutually exclusive arguments.")
        if grid_constructor is not None:
            self.grid_constructor = grid_constructor
        elif step_size is None:
            self.grid_constructor = lambda func, x0, t: t
        else:
            self.grid_constructor = self._grid_constructor_from_step_size(step_size)

    @staticmethod
    def _grid_constructor_from_step_size(step_size):
        # unused by either integrate_* in the reference as well (my_solvers.py:31-42, :54 commented out)
        def _grid_constructor(t):
            n = torch.ceil((t[-1] - t[0]) / step_size + 1).item()
            grid = torch.arange(0, n, dtype=t.dtype, device=t.device) * step_size + t[0]
            grid[-1] = t[-1]
            return grid
        return _grid_constructor

    def _note_walk(self, what, tensor):
        """The walk is the reference's own route, but on a HIP device it is ~100x slower than the fused one: never silent."""
        if self.fused == "auto" and tensor.device.type == "cuda" and not getattr(self, "_walk_warned", False):
            self._walk_warned = True
            warnings.warn(f"{what}: this call is not fusable (needs fp32 HIP tensors, DE_Func/AE_Func-style ELU-MLPs, "
                          "ODE_Event/DAE_Event callbacks; under autograd also a shape with a backward kernel and no teacher "
                          "forcing) -- stepping through the Python callables instead", RuntimeWarning, stacklevel=3)

    def _check_events_now(self, event_t):
        return (self.check_events and event_t is not None and event_t.dim() == 3 and event_t.shape[1] > 1
                and not torch.cuda.is_current_stream_capturing())

    @abc.abstractmethod
    def _step_func(self, func, t0, dt, t1, x0, z0=None, v0=None, i0=None, all_initial=None):
        """-> (dx, f0)"""

    def step_integrate(self, func, t0, dt, t1, x0, z0=None, v0=None, i0=None, all_initial=None):
        dx, f0 = self._step_func(func=func, t0=t0, dt=dt, t1=t1, x0=x0, z0=z0, v0=v0, i0=i0, all_initial=all_initial)
        return x0 + dx, f0

    # ------------------------------------------------------------------ ODE
    def integrate_ODE(self, x_func, t, x, z, all_initial, event_fn=None, jump_change_fn=None, input_true_x=False, x_init=None):
        """my_solvers.py:53-79.  `x_init` (an extension, default None = upstream: the integration starts from x[0]) hands the [B,x_dim]
        initial state over on its own: a caller whose x is a big differentiable tensor (the direct_encode models' Xh) otherwise gets
        d loss / d x back as a [T,B,x_dim] tensor that is zero except for row 0 -- allocated, filled and ADDED to x's other gradient."""
        if x_init is not None and input_true_x:
            raise ValueError("integrate_ODE: x_init and input_true_x exclude each other (teacher forcing starts every step from x[k])")
        if self.fused != "off":
            plan = _fused.plan_ode(x_func, x, z, all_initial, event_fn, jump_change_fn, t=t, x_init=x_init)
            if plan is not None:
                layers, event_t, z_jump, needs_grad = plan
                if not needs_grad:
                    try:
                        return _fused.ode_integrate(self.method, layers, t, x if x_init is None else x_init.unsqueeze(0), z, all_initial,
                                                    event_t=event_t, z_jump=z_jump,
                                                    input_true_x=input_true_x, kernel=self.kernel,
                                                    check_events=self._check_events_now(event_t))
                    except UnsupportedShapeError:      # no kernel covers the shape (too wide for LDS): user callables it is
                        if self.fused == "require":
                            raise
                # training: fused forward + fused backward when the backward kernel covers the shape
                elif not input_true_x and _autograd().ode_training_supported(self.method, layers, x.shape[-1], z.shape[-1], t.shape[0],
                                                                             t.shape[1], kernel=self.kernel):
                    from ..autograd import fused_ode_integrate
                    return fused_ode_integrate(self.method, self.kernel, layers, t, x, z, all_initial, event_t, z_jump,
                                               check_events=self._check_events_now(event_t), x_init=x_init)
                # teacher-forced training (my_solvers.py:72-74): K4f in its recompute form; the dataset x gets no gradient
                elif input_true_x and not x.requires_grad and self.kernel in ("auto", "mfma") and \
                        _fused.ode_backward_supported(self.method, layers, x.shape[-1], z.shape[-1], "wide"):
                    from ..autograd import fused_ode_integrate
                    return fused_ode_integrate(self.method, self.kernel, layers, t, x, z, all_initial, event_t, z_jump,
                                               check_events=self._check_events_now(event_t), input_true_x=True)
            if self.fused == "require":
                raise NotFusableError("integrate_ODE: call is not fusable (needs fp32 HIP tensors, a DE_Func-style ELU-MLP "
                                      "`x_dot`, ODE_Event callbacks; with autograd: a shape with a backward kernel, no teacher forcing)")
            self._note_walk("integrate_ODE", x)
        return self._walk_ode(x_func, t, x, z, all_initial, event_fn, jump_change_fn, input_true_x, x_init)

    def _walk_ode(self, x_func, t, x, z, all_initial, event_fn, jump_change_fn, input_true_x, x_init=None):
        n_grid = t.shape[0]
        xs = torch.zeros(x.shape, dtype=x.dtype, device=x.device)
        cur = x[0] if x_init is None else x_init
        xs[0] = cur
        for k in range(n_grid - 1):
            t0, t1, zk = t[k], t[k + 1], z[k]
            if event_fn is not None and event_fn(t0) == True:  # noqa: E712 (callbacks may return tensors)
                zk = jump_change_fn(t0, zk)
            start = x[k] if input_true_x else cur
            cur, _ = self.step_integrate(func=x_func, t0=t0, dt=t1 - t0, t1=t1, x0=start, z0=zk, all_initial=all_initial)
            xs[k + 1] = cur
        return xs

    # ------------------------------------------------------------------ DAE
    def integrate_DAE(self, x_init, x_func, i_func, t, x, z, v, i, all_initial, event_fn=None, jump_change_fn=None,
                      input_true_x=False, input_true_i=False):
        if self.fused != "off":
            plan = _fused.plan_dae(x_init, x_func, i_func, z, v, i, all_initial, event_fn, jump_change_fn, t=t)
            if plan is not None:
                de, ae, event_t, z_jump, v_jump, needs_grad = plan
                if not needs_grad:
                    try:
                        return _fused.dae_integrate(self.method, de, ae, x_init, t, x, z, v, i, all_initial, event_t=event_t,
                                                    z_jump=z_jump, v_jump=v_jump, input_true_x=input_true_x,
                                                    input_true_i=input_true_i, kernel=self.kernel,
                                                    check_events=self._check_events_now(event_t))
                    except UnsupportedShapeError:
                        if self.fused == "require":
                            raise
                elif not (input_true_x or input_true_i) and _autograd().dae_training_supported(
                        self.method, de, ae, x_init.shape[-1], z.shape[-1], v.shape[-1], i.shape[-1], t.shape[0], t.shape[1]):
                    from ..autograd import fused_dae_integrate
                    return fused_dae_integrate(self.method, self.kernel, de, ae, x_init, t, z, v, i, all_initial, event_t, z_jump, v_jump,
                                               check_events=self._check_events_now(event_t))
                # teacher-forced training (my_solvers.py:111-121): K7f in its recompute form; the dataset rows get no gradient
                elif (input_true_x or input_true_i) and not (input_true_x and x.requires_grad) and not (input_true_i and i.requires_grad) \
                        and x.shape[-1] == x_init.shape[-1] and self.kernel in ("auto", "mfma") and t.shape[0] >= 2 and \
                        _fused.dae_backward_wide_supported(self.method, de, ae, x_init.shape[-1], z.shape[-1], v.shape[-1], i.shape[-1]):
                    from ..autograd import fused_dae_integrate
                    return fused_dae_integrate(self.method, self.kernel, de, ae, x_init, t, z, v, i, all_initial, event_t, z_jump, v_jump,
                                               check_events=self._check_events_now(event_t), x=x, input_true_x=input_true_x,
                                               input_true_i=input_true_i)
            if self.fused == "require":
                raise NotFusableError("integrate_DAE: call is not fusable (needs fp32 HIP tensors, DE_Func/AE_Func-style "
                                      "ELU-MLPs, DAE_Event callbacks; with autograd: a shape with a backward kernel; teacher forcing: hidden <= 128, dataset rows without grad)")
            self._note_walk("integrate_DAE", z if z.numel() else v)
        return self._walk_dae(x_init, x_func, i_func, t, x, z, v, i, all_initial, event_fn, jump_change_fn,
                              input_true_x, input_true_i)

    def _walk_dae(self, x_init, x_func, i_func, t, x, z, v, i, all_initial, event_fn, jump_change_fn, input_true_x, input_true_i):
        n_grid = t.shape[0]
        cur_x = x_init
        cur_i = i_func(xt=x[0] if input_true_x else cur_x, zt=z[0], vt=v[0], all_initial=all_initial)
        x_shape = (*x.shape[0:2], x_init.shape[-1]) if x.shape[-1] == 0 else x.shape
        xs = torch.zeros(x_shape, dtype=x_init.dtype if x.shape[-1] == 0 else x.dtype, device=x.device)
        is_ = torch.zeros(i.shape, dtype=i.dtype, device=i.device)
        xs[0], is_[0] = cur_x, cur_i
        for k in range(n_grid - 1):
            t0, t1, zk, vk = t[k], t[k + 1], z[k], v[k]
            if event_fn is not None and event_fn(t0) == True:  # noqa: E712
                zk, vk = jump_change_fn(t0, zk, vk)
                cur_i = i_func(xt=cur_x, zt=zk, vt=vk, all_initial=all_initial)
            start = x[k] if input_true_x else cur_x
            i_in = i[k] if input_true_i else cur_i
            cur_x, _ = self.step_integrate(func=x_func, t0=t0, dt=t1 - t0, t1=t1, x0=start, z0=zk, v0=vk, i0=i_in,
                                           all_initial=all_initial)
            cur_i = i_func(xt=x[k + 1] if input_true_x else cur_x, zt=z[k + 1], vt=v[k + 1], all_initial=all_initial)
            xs[k + 1], is_[k + 1] = cur_x, cur_i
        return xs, is_

    # dead helpers of the reference kept for API completeness (my_solvers.py:177-192); nothing calls them:
    # the integrators hold external inputs constant over a step (zero-order hold).
    def _cubic_hermite_interp(self, t0, x0, f0, t1, x1, f1, t):
        h = (t - t0) / (t1 - t0)
        dt = t1 - t0
        return ((1 + 2 * h) * (1 - h) ** 2) * x0 + (h * (1 - h) ** 2) * dt * f0 + (h * h * (3 - 2 * h)) * x1 + (h * h * (h - 1)) * dt * f1

    def _linear_interp(self, t0, t1, x0, x1, t):
        if t == t0:
            return x0
        if t == t1:
            return x1
        return x0 + (t - t0) / (t1 - t0) * (x1 - x0)
