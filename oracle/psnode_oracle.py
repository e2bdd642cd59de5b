"""CPU ORACLE for the fixed-grid neural-ODE/DAE integration path.  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (PyTorch-CPU fp32 ops, in the reference's op order) of the
algorithm implemented by the reference repository xxh0523/Py_PSNODE:

  * time loops            /root/reference/neural_dae/my_solvers.py:52-80   (integrate_ODE)
                          /root/reference/neural_dae/my_solvers.py:82-131  (integrate_DAE)
  * step formulas         /root/reference/neural_dae/my_fixed_grid.py:12-59 (Euler / Midpoint / RK4 3/8-rule)
  * right-hand sides      /root/reference/neural_00_ODE_01_no_encode.py:58-68   (DE, ODE recipe)
                          /root/reference/neural_01_DAE_01_no_encode.py:61-83   (DE + AE, DAE recipe)
  * events                /root/reference/neural_dae/neural_base.py:43-65,169-196

Who may use it: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
-- as the checker / the reported CPU baseline, never as the product path.  Nothing under
``py_psnode_amd/`` imports this module; the product path is the HIP library and it fails loudly
when that library is missing.

Parity pinning: the functions below are checked against golden vectors captured from the *imported
reference itself* in the build container (tests/golden/make_goldens.py, tests/test_oracle_golden.py).
The reference ships no tests, golden vectors or fixtures of its own, so these
captured outputs are the only pin that exists.

Inputs are raw tensors: an MLP is a list of (W[out,in], b[out]) pairs exactly as nn.Linear stores them,
with ELU(alpha=1) between consecutive Linear layers and none after the last.
All tensors are time-major: t[T,B,1], x[T,B,xd], z[T,B,zd], ... like the solver seam of the reference.
"""
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Layers = Sequence[Tuple[torch.Tensor, torch.Tensor]]

# my_fixed_grid.py:8-9 -- python doubles multiplied into fp32 tensors
_ONE_THIRD = 1 / 3
_TWO_THIRDS = 2 / 3

METHODS = ("euler", "midpoint", "rk4")


def mlp_forward(layers: Layers, u: torch.Tensor) -> torch.Tensor:
    """nn.Sequential(Linear, ELU, ..., Linear) forward (neural_00_ODE_01_no_encode.py:61-64)."""
    n = len(layers)
    for k, (w, b) in enumerate(layers):
        u = F.linear(u, w, b)
        if k + 1 < n:
            u = F.elu(u)
    return u


def de_rhs(layers: Layers, xt: torch.Tensor, ext: Sequence[torch.Tensor], all_initial: torch.Tensor) -> torch.Tensor:
    """DE_Func.forward: MLP(cat(a0, s - a0, s)), s = cat(xt, *ext).

    ODE recipe ext=(zt,)          neural_00_ODE_01_no_encode.py:66-68
    DAE recipe ext=(zt, vt, it)   neural_01_DAE_01_no_encode.py:69-71
    """
    s = torch.cat((xt, *ext), dim=-1)
    return mlp_forward(layers, torch.cat((all_initial, s - all_initial, s), dim=-1))


def ae_rhs(layers: Layers, xt: torch.Tensor, zt: torch.Tensor, vt: torch.Tensor, all_initial: torch.Tensor) -> torch.Tensor:
    """AE_Func.forward: MLP(cat(a0, xt, zt, vt)) (neural_01_DAE_01_no_encode.py:82-83)."""
    return mlp_forward(layers, torch.cat((all_initial, xt, zt, vt), dim=-1))


def step(method: str, f, t0, dt, t1, x0):
    """One fixed-grid step; returns (x1, f0) like step_integrate (my_solvers.py:48-50).

    f(x) is the right-hand side with the step's external inputs frozen (zero-order hold,
    my_fixed_grid.py:43-50 pass z0/v0/i0 unchanged to every stage).
    """
    if method == "euler":                       # my_fixed_grid.py:15-18
        f0 = f(x0)
        dx = dt * f0
    elif method == "midpoint":                  # my_fixed_grid.py:23-32
        half_dt = 0.5 * dt
        f0 = f(x0)
        x_mid = x0 + f0 * half_dt
        dx = dt * f(x_mid)
    elif method == "rk4":                       # my_fixed_grid.py:38-59 (3/8 rule)
        f0 = f(x0)
        k1 = f0
        k2 = f(x0 + dt * k1 * _ONE_THIRD)
        k3 = f(x0 + dt * (k2 - k1 * _ONE_THIRD))
        k4 = f(x0 + dt * (k1 - k2 + k3))
        dx = (k1 + 3 * (k2 + k3) + k4) * dt * 0.125
    else:
        raise ValueError(f"unknown method {method!r}")
    return x0 + dx, f0


def _event_hit(event_t: Optional[torch.Tensor], t0: torch.Tensor) -> bool:
    """ODE_Event.event_fn (neural_base.py:52-57): trajectory 0's clock vs trajectory 0's event list."""
    if event_t is None:
        return False
    return bool((event_t[0] == t0[0]).any())


def _jump(event_t: torch.Tensor, jump: torch.Tensor, t0: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    """jump_change_fn (neural_base.py:59-62): replace the whole batch's input by jump[:, e]."""
    sel = (event_t[0] == t0[0][0]).view(-1)
    out = like.clone()
    out[:] = jump[:, sel].view(like.shape)
    return out


@torch.no_grad()
def integrate_ode(method: str, de_layers: Layers, t, x, z, all_initial,
                  event_t=None, z_jump=None, input_true_x: bool = False) -> torch.Tensor:
    """integrate_ODE (my_solvers.py:52-80)."""
    x0 = x[0]
    xs = torch.zeros(x.shape, dtype=x.dtype)
    xs[0] = x0
    for j in range(1, t.shape[0]):
        t0, t1, z0 = t[j - 1], t[j], z[j - 1]
        dt = t1 - t0
        if _event_hit(event_t, t0):
            z0 = _jump(event_t, z_jump, t0, z0)
        src = x[j - 1] if input_true_x else x0
        x1, _ = step(method, lambda xx: de_rhs(de_layers, xx, (z0,), all_initial), t0, dt, t1, src)
        xs[j] = x1
        x0 = x1
    return xs


@torch.no_grad()
def integrate_dae(method: str, de_layers: Layers, ae_layers: Layers, x_init, t, x, z, v, i, all_initial,
                  event_t=None, z_jump=None, v_jump=None,
                  input_true_x: bool = False, input_true_i: bool = False):
    """integrate_DAE (my_solvers.py:82-131)."""
    x0 = x_init
    i0 = ae_rhs(ae_layers, x[0] if input_true_x else x0, z[0], v[0], all_initial)
    if x.shape[-1] == 0:
        xs = torch.zeros((*x.shape[0:2], x_init.shape[-1]), dtype=x_init.dtype)
    else:
        xs = torch.zeros(x.shape, dtype=x.dtype)
    xs[0] = x0
    is_ = torch.zeros(i.shape, dtype=i.dtype)
    is_[0] = i0
    for j in range(1, t.shape[0]):
        t0, t1 = t[j - 1], t[j]
        z0, z1, v0, v1 = z[j - 1], z[j], v[j - 1], v[j]
        dt = t1 - t0
        if _event_hit(event_t, t0):
            z0 = _jump(event_t, z_jump, t0, z0)
            v0 = _jump(event_t, v_jump, t0, v0)
            i0 = ae_rhs(ae_layers, x0, z0, v0, all_initial)
        src = x[j - 1] if input_true_x else x0
        i_in = i[j - 1] if input_true_i else i0
        x1, _ = step(method, lambda xx: de_rhs(de_layers, xx, (z0, v0, i_in), all_initial), t0, dt, t1, src)
        i1 = ae_rhs(ae_layers, x[j] if input_true_x else x1, z1, v1, all_initial)
        xs[j] = x1
        is_[j] = i1
        x0 = x1
        i0 = i1
    return xs, is_


def event_step_table(t: torch.Tensor, event_t: Optional[torch.Tensor]) -> List[int]:
    """Per-step event index (-1 = none) as the reference's event_fn/jump_change_fn pair resolves it.

    Used by tests to check the host-side table the product builds for the kernel.
    """
    tab = []
    for j in range(t.shape[0] - 1):
        if event_t is None:
            tab.append(-1)
            continue
        sel = (event_t[0] == t[j][0][0]).view(-1).nonzero().view(-1).tolist()
        tab.append(sel[0] if len(sel) else -1)
    return tab


# ---- the loss expressions that follow the path in the scripts' training loops (Loss_func = nn.functional.mse_loss,
#      neural_00_ODE_01_no_encode.py:49).  Tensors in the scripts' [B,T,D] shape.
def ode_loss(x_pred: torch.Tensor, x: torch.Tensor, mask: torch.Tensor):
    """neural_00_ODE_01_no_encode.py:354-355 (same lines in neural_00_ODE_02_direct_encode.py:268,270):
    returns (loss, x_loss[D])."""
    se = torch.nn.functional.mse_loss(x_pred, x, reduction="none")
    x_loss = torch.sum(torch.sum(se * mask, dim=1), dim=0) / torch.sum(mask)
    return torch.sum(x_loss), x_loss


def dae_loss(x_pred, x, i_pred, i, mask):
    """neural_01_DAE_01_no_encode.py:414-419: returns (loss, x_loss, i_loss, x0_term, i0_term)."""
    mse = torch.nn.functional.mse_loss
    x_loss = (torch.sum(mse(x_pred, x, reduction="none") * mask)
              + torch.sum(mse(x_pred[:, :, 1:2], x[:, :, 1:2], reduction="none") * mask) * 9) / torch.sum(mask)
    i_loss = torch.sum(mse(i_pred, i, reduction="none") * mask) / torch.sum(mask)
    x0, i0 = mse(x[:, 0, :], x_pred[:, 0, :]), mse(i[:, 0, :], i_pred[:, 0, :])
    return x_loss + i_loss + x0 + i0, x_loss, i_loss, x0, i0


def recon_loss(x_re: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """neural_00_ODE_02_direct_encode.py:269: Loss_func(x_re, x)."""
    return torch.nn.functional.mse_loss(x_re, x)


def ode02_loss(x_pred, x_re, x, mask):
    """neural_00_ODE_02_direct_encode.py:267-270: x0_loss + sum(x_loss) + x_recon_loss."""
    mse = torch.nn.functional.mse_loss
    x0_loss = mse(x[:, 0, :], x_pred[:, 0, :]).view(1)
    x_loss = torch.sum(torch.sum(mse(x_pred, x, reduction="none") * mask, dim=1), dim=0) / torch.sum(mask)
    x_recon_loss = mse(x_re, x).view(1)
    return torch.sum(x0_loss) + torch.sum(x_loss) + torch.sum(x_recon_loss)


def dae02_loss(x_pred, i_pred, x_re, i_re, x, i, mask):
    """neural_01_DAE_02_direct_encode.py:359-365: no extra column weight (commented out upstream), two reconstruction terms."""
    mse = torch.nn.functional.mse_loss
    x_loss = torch.sum(mse(x_pred, x, reduction="none") * mask) / torch.sum(mask)
    i_loss = torch.sum(mse(i_pred, i, reduction="none") * mask) / torch.sum(mask)
    recon = mse(x_re, x) + mse(i_re, i)
    return x_loss + i_loss + mse(x[:, 0, :], x_pred[:, 0, :]) + mse(i[:, 0, :], i_pred[:, 0, :]) + recon

