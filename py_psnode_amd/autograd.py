"""torch.autograd bridge for the fused integrator: forward = psnode_ode_integrate_f32, backward = psnode_ode_backward_f32.

This is what lets the reference's training loops (`loss.backward()` through the integrator,
neural_00_ODE_01_no_encode.py:358-360) run on the fused HIP path instead of an unrolled T-step autograd graph.
"""
import os
import warnings

import torch

from . import fused

# Save the stage activations in the training forward instead of recomputing them in the backward?  "auto": at every hidden width the
# MFMA integrators take (measured per 4096 x 1000 RK4 batch, end of round 3: hidden 128 48.2 -> 34.5 ms per ODE_01 training step, 64
# 17.5 -> 14.3, 32 10.4 -> 9.6; DAE_01 77.0 -> 52.2, 23.0 -> 21.4, 16.9 -> 14.7) whenever the rows (6 KB per state-step at 128: 25 GB for
# that batch; 3.1 KB / 13 GB at 64) fit into half of the free HBM; "1" / "0" force it on / off.
SAVE_ACTIVATIONS = os.environ.get("PSNODE_SAVE_ACTIVATIONS", "auto")
# bytes of stage activations the most recent training forward kept for its backward (0: the backward recomputes) -- bench.py reports it
last_saved_bytes = 0
_warned_generic_override = False


def latent_wide_training_fits(method, de, ae, hidden, T, B, dev) -> bool:
    """Training at the latent-wide hidden widths exists in ONE form: K3w saves its rows, K9w writes as many adjoint rows again and the
    host contracts them (fused.latent_backward_wide) -- there is no recompute form to fall back to.  True if all of that fits half of the
    free HBM and PSNODE_SAVE_ACTIVATIONS is not "0"; otherwise the solver takes the walk through the user's callables (with its warning)."""
    if SAVE_ACTIVATIONS == "0":
        return False
    if SAVE_ACTIVATIONS == "1" or T < 2:         # (T = 1: no step, nothing to save)
        return True
    S = {"euler": 1, "midpoint": 2, "rk4": 4}[method]
    rows = (T - 1) * S * B * hidden * 4              # one [T-1,S,B,H] tensor
    grid = T * B * hidden * 4                        # one [T,B,H] tensor
    need = 4 * rows + (6 if ae is None else 12) * grid       # saved act + xst, gk + d1; d1s, the where / contiguous copies of the external blocks, (DAE) gi, da1, s_ae
    free, _ = torch.cuda.mem_get_info(dev)
    return need <= free // 2


def ode_training_supported(method, layers, x_dim, z_dim, T, B, kernel="auto") -> bool:
    """What the solver asks before it routes a call that needs autograd to the fused forward + backward pair."""
    if kernel in ("auto", "mfma") and fused.latent_wide_shape(layers, None, x_dim, z_dim):
        return latent_wide_training_fits(method, layers, None, x_dim, T, B, layers[0][0].device)
    return fused.ode_backward_supported(method, layers, x_dim, z_dim, kernel)


def dae_training_supported(method, de, ae, x_dim, z_dim, v_dim, i_dim, T, B) -> bool:
    if fused.latent_wide_shape(de, ae, x_dim, z_dim, v_dim, i_dim):
        return latent_wide_training_fits(method, de, ae, x_dim, T, B, de[0][0].device)
    return fused.dae_backward_supported(method, de, ae, x_dim, z_dim, v_dim, i_dim)


def _want_saved(method, kernel, layers, x_dim, z_dim, T, B):
    if fused.latent_wide_shape(layers, None, x_dim, z_dim):      # K3w saves, K9w reads: the only fused backward at these widths (the solver
        return True                                              # asked latent_wide_training_fits before it came here)
    if SAVE_ACTIVATIONS == "0" or T < 2 or kernel not in ("auto", "mfma", "wave", "tile"):
        return False
    Hp = fused.ode_save_hidden(method, layers, x_dim, z_dim, kernel)
    latent = len(layers) == 2            # the direct_encode latent shape at hidden 64: K3c saves, K9 reads
    if Hp <= 0 or not (latent or fused.ode_backward_supported(method, layers, x_dim, z_dim, "wide")):
        return False
    if SAVE_ACTIVATIONS == "1":
        return True
    S = {"euler": 1, "midpoint": 2, "rk4": 4}[method]
    need = (T - 1) * S * B * ((len(layers) - 1) * Hp + x_dim) * 4
    free, _ = torch.cuda.mem_get_info(layers[0][0].device)
    return need <= free // 2


def _want_saved_dae(method, kernel, de, ae, x_dim, z_dim, v_dim, i_dim, T, B):
    """The same policy for the DAE: saved rows are read by the fused-DE backward K7f, i.e. at hidden widths other than 64 (K7, the
    one-launch kernel there, recomputes)."""
    if fused.latent_wide_shape(de, ae, x_dim, z_dim, v_dim, i_dim):
        return True
    if SAVE_ACTIVATIONS == "0" or T < 2 or kernel not in ("auto", "mfma") or len(de) not in (2, 4):
        return False
    if len(de) == 2:                     # the direct_encode latent shape at hidden 64 (K3c saves, K9 reads); hidden 16 (K3a / K8) recomputes
        if fused.dae_save_hidden(method, de, ae, x_dim, z_dim, v_dim, i_dim, kernel) != 64:
            return False
        if SAVE_ACTIVATIONS == "1":
            return True
        S = {"euler": 1, "midpoint": 2, "rk4": 4}[method]
        free, _ = torch.cuda.mem_get_info(de[0][0].device)
        return ((T - 1) * S * 2 + T) * 64 * B * 4 <= free // 2
    # hidden 64 also has the one-launch kernel K7 (recompute).  The saved form beats it at every method since round 4 (K7f requests its
    # per-step inputs a step ahead: training step at 4096 x 1000 RK4 19.7 vs 23.0 ms, Euler 9.6 vs 10.6 -- gpurun_out/r04g.log; round 3:
    # Euler 11.0 vs 10.65, which kept K7 for Euler)
    Hp = fused.dae_save_hidden(method, de, ae, x_dim, z_dim, v_dim, i_dim, kernel)
    if Hp <= 0 or not fused.dae_backward_wide_supported(method, de, ae, x_dim, z_dim, v_dim, i_dim):
        return False
    if SAVE_ACTIVATIONS == "1":
        return True
    S = {"euler": 1, "midpoint": 2, "rk4": 4}[method]
    need = ((T - 1) * S * (3 * Hp + x_dim) + 3 * T * Hp) * B * 4
    free, _ = torch.cuda.mem_get_info(de[0][0].device)
    return need <= free // 2


class _FusedOde(torch.autograd.Function):
    @staticmethod
    def forward(ctx, method, kernel, event_idx, t, x0, z, all_initial, z_jump, *params):
        layers = [(params[k], params[k + 1]) for k in range(0, len(params), 2)]
        global last_saved_bytes
        ctx.x_true = False
        if x0.dim() == 3:        # teacher forcing (my_solvers.py:72-74): x0 is the whole dataset x [T,B,xd]; nothing is saved, K4f recomputes
            x_true = x0.detach().contiguous()
            xs = fused.ode_integrate(method, layers, t, x_true, z, all_initial, z_jump=z_jump, event_idx=event_idx, kernel=kernel,
                                     input_true_x=True)
            last_saved_bytes = 0
            ctx.x_true = True
            ctx.method, ctx.has_jump, ctx.has_saved, ctx.event_idx = method, z_jump is not None, False, event_idx
            # (the dataset rows go through save_for_backward like everything else the backward reads: autograd's version counter then
            #  catches an in-place edit of x between forward and backward)
            ctx.save_for_backward(t, z, all_initial, xs, *((z_jump,) if z_jump is not None else ()), x_true, *params)
            return xs
        if kernel == "generic" and fused.latent_wide_shape(layers, None, x0.shape[-1], z.shape[-1]):
            global _warned_generic_override
            if not _warned_generic_override:
                _warned_generic_override = True
                warnings.warn("kernel='generic' is ignored for training at the latent hidden widths other than 16 / 64: the only backward "
                              "there reads the rows K3w saves (K0 saves nothing)", RuntimeWarning, stacklevel=3)
            kernel = "auto"      # training at these widths exists on K3w + K9w only (K0 saves nothing)
        save = _want_saved(method, kernel, layers, x0.shape[-1], z.shape[-1], t.shape[0], t.shape[1])
        res = fused.ode_integrate(method, layers, t, x0.unsqueeze(0), z, all_initial, z_jump=z_jump, event_idx=event_idx, kernel=kernel,
                                  save=save)
        xs, saved = res if save else (res, None)
        last_saved_bytes = sum(q.numel() * q.element_size() for q in saved) if saved is not None else 0
        ctx.method = method
        ctx.bwd_kernel = {"wave": "wave", "tile": "tile"}.get(kernel, "auto")      # a forced forward form forces its backward counterpart (K4x / K4f)
        if ctx.bwd_kernel != "auto" and not fused.ode_backward_supported(method, layers, x0.shape[-1], z.shape[-1], ctx.bwd_kernel):
            ctx.bwd_kernel = "auto"                                                    # ... where that counterpart exists for the shape
        ctx.has_jump = z_jump is not None
        ctx.has_saved = saved is not None
        ctx.event_idx = event_idx
        ctx.save_for_backward(t, z, all_initial, xs, *((z_jump,) if z_jump is not None else ()), *(saved if saved is not None else ()),
                              *params)
        return xs

    @staticmethod
    def backward(ctx, grad_xs):
        saved = ctx.saved_tensors
        t, z, a0, xs = saved[:4]
        pos = 4
        z_jump = saved[pos] if ctx.has_jump else None
        pos += 1 if ctx.has_jump else 0
        acts = (saved[pos], saved[pos + 1]) if ctx.has_saved else None
        pos += 2 if ctx.has_saved else 0
        x_true = saved[pos] if ctx.x_true else None
        pos += 1 if ctx.x_true else 0
        params = saved[pos:]
        layers = [(params[k], params[k + 1]) for k in range(0, len(params), 2)]
        need_z = ctx.needs_input_grad[5]
        if x_true is not None:   # teacher forcing: every step started from a dataset row -- K4f with the dataset as `xs`, no carried adjoint
            gx0, gz, gzj, ga0, gpar = fused.ode_backward(ctx.method, layers, t, z, a0, x_true, grad_xs, event_idx=ctx.event_idx,
                                                         z_jump=z_jump, need_grad_z=need_z, kernel="wide", input_true_x=True)
            if gz is None and need_z:
                gz = torch.zeros_like(z)
            return (None, None, None, None, None, gz, ga0, gzj if ctx.needs_input_grad[7] else None, *gpar)   # (no gradient for the dataset x)
        gx0, gz, gzj, ga0, gpar = fused.ode_backward(ctx.method, layers, t, z, a0, xs, grad_xs, event_idx=ctx.event_idx, z_jump=z_jump,
                                                     need_grad_z=need_z, saved=acts, need_grad_zj=bool(ctx.needs_input_grad[7]),
                                                     kernel=ctx.bwd_kernel if acts is not None else "auto")
        if gz is None and need_z:
            gz = torch.zeros_like(z)
        return (None, None, None, None, gx0, gz, ga0, gzj if ctx.needs_input_grad[7] else None, *gpar)


def fused_ode_integrate(method, kernel, layers, t, x, z, all_initial, event_t=None, z_jump=None, check_events=False, input_true_x=False,
                        x_init=None):
    """Differentiable fused integrate_ODE: gradients flow to x[0], z, all_initial, z_jump and the MLP.  input_true_x (teacher forcing,
    my_solvers.py:72-74): every step starts from the dataset row x[k]; gradients flow to z, all_initial, z_jump and the MLP (the dataset
    x gets none: callers whose x requires grad take the callback walk)."""
    with torch.no_grad():
        event_idx = fused.event_table(t, event_t, check_events)
    if event_idx is None:
        z_jump = None
    params = [p for wb in layers for p in wb]
    x0 = x.detach() if input_true_x else (x[0] if x_init is None else x_init)     # (x_init: integrate_ODE's extension -- no SelectBackward)
    return _FusedOde.apply(method, kernel, event_idx, t, x0, z, all_initial, z_jump, *params)


class _FusedDae(torch.autograd.Function):
    @staticmethod
    def forward(ctx, method, kernel, event_idx, n_de, t, x_init, z, v, i_shape_like, all_initial, z_jump, v_jump, *params):
        de = [(params[k], params[k + 1]) for k in range(0, 2 * n_de, 2)]
        ae = [(params[k], params[k + 1]) for k in range(2 * n_de, len(params), 2)]
        T, B = t.shape[0], t.shape[1]
        x_dummy = x_init.new_zeros((1, B, 0))
        if kernel == "generic" and fused.latent_wide_shape(de, ae, x_init.shape[-1], z.shape[-1], v.shape[-1], i_shape_like.shape[-1]):
            kernel = "auto"
        save = _want_saved_dae(method, kernel, de, ae, x_init.shape[-1], z.shape[-1], v.shape[-1], i_shape_like.shape[-1], T, B)
        res = fused.dae_integrate(method, de, ae, x_init, t, x_dummy, z, v, i_shape_like, all_initial, z_jump=z_jump, v_jump=v_jump,
                                  event_idx=event_idx, kernel=kernel, save=save)
        xs, is_ = res[0], res[1]
        acts = [q for q in res[2] if q is not None] if save else []
        global last_saved_bytes
        last_saved_bytes = sum(q.numel() * q.element_size() for q in acts)
        ctx.method, ctx.n_de, ctx.event_idx = method, n_de, event_idx
        ctx.has_zj, ctx.has_vj = z_jump is not None, v_jump is not None
        ctx.n_saved = len(acts)
        ctx.save_for_backward(t, z, v, all_initial, xs, is_, *((z_jump,) if z_jump is not None else ()),
                              *((v_jump,) if v_jump is not None else ()), *acts, *params)
        return xs, is_

    @staticmethod
    def backward(ctx, grad_xs, grad_is):
        sv = list(ctx.saved_tensors)
        t, z, v, a0, xs, is_ = sv[:6]
        k = 6
        z_jump = sv[k] if ctx.has_zj else None
        k += int(ctx.has_zj)
        v_jump = sv[k] if ctx.has_vj else None
        k += int(ctx.has_vj)
        acts = None
        if ctx.n_saved:
            acts = tuple(sv[k:k + ctx.n_saved]) + (None,) * (5 - ctx.n_saved)
            k += ctx.n_saved
        params = sv[k:]
        de = [(params[q], params[q + 1]) for q in range(0, 2 * ctx.n_de, 2)]
        ae = [(params[q], params[q + 1]) for q in range(2 * ctx.n_de, len(params), 2)]
        g = fused.dae_backward(ctx.method, de, ae, t, z, v, a0, xs, is_, grad_xs, grad_is, event_idx=ctx.event_idx, z_jump=z_jump, v_jump=v_jump,
                               saved=acts)
        gz = g["z"] if g["z"] is not None else (torch.zeros_like(z) if ctx.needs_input_grad[6] else None)
        gv = g["v"] if g["v"] is not None else (torch.zeros_like(v) if ctx.needs_input_grad[7] else None)
        return (None, None, None, None, None, g["x_init"], gz, gv, None, g["all_initial"],
                g["z_jump"] if ctx.needs_input_grad[10] else None, g["v_jump"] if ctx.needs_input_grad[11] else None, *g["de"], *g["ae"])


class _FusedDaeTeacherForced(torch.autograd.Function):
    """integrate_DAE with input_true_x and / or input_true_i (my_solvers.py:111-121): forward K2 with the flags (nothing saved), backward
    K7f in its recompute form with the dataset rows (psnode_dae_bwd_wide_args_f32::x_true / i_true).  The dataset rows get no gradient."""

    @staticmethod
    def forward(ctx, method, kernel, event_idx, n_de, tx, ti, t, x_init, x, z, v, i, all_initial, z_jump, v_jump, *params):
        de = [(params[k], params[k + 1]) for k in range(0, 2 * n_de, 2)]
        ae = [(params[k], params[k + 1]) for k in range(2 * n_de, len(params), 2)]
        xs, is_ = fused.dae_integrate(method, de, ae, x_init, t, x, z, v, i, all_initial, z_jump=z_jump, v_jump=v_jump, event_idx=event_idx,
                                      kernel=kernel, input_true_x=tx, input_true_i=ti)[:2]
        ctx.method, ctx.n_de, ctx.event_idx, ctx.tx, ctx.ti = method, n_de, event_idx, tx, ti
        ctx.has_zj, ctx.has_vj = z_jump is not None, v_jump is not None
        ctx.save_for_backward(t, z, v, all_initial, xs, is_, x, i, *((z_jump,) if z_jump is not None else ()),
                              *((v_jump,) if v_jump is not None else ()), *params)
        return xs, is_

    @staticmethod
    def backward(ctx, grad_xs, grad_is):
        sv = list(ctx.saved_tensors)
        t, z, v, a0, xs, is_, x, i = sv[:8]
        k = 8
        z_jump = sv[k] if ctx.has_zj else None
        k += int(ctx.has_zj)
        v_jump = sv[k] if ctx.has_vj else None
        k += int(ctx.has_vj)
        params = sv[k:]
        de = [(params[q], params[q + 1]) for q in range(0, 2 * ctx.n_de, 2)]
        ae = [(params[q], params[q + 1]) for q in range(2 * ctx.n_de, len(params), 2)]
        g = fused.dae_backward_wide(ctx.method, de, ae, t, z, v, a0, xs, is_, grad_xs, grad_is, event_idx=ctx.event_idx, z_jump=z_jump,
                                    v_jump=v_jump, x_true=x if ctx.tx else None, i_true=i if ctx.ti else None)
        gz = g["z"] if g["z"] is not None else (torch.zeros_like(z) if ctx.needs_input_grad[9] else None)
        gv = g["v"] if g["v"] is not None else (torch.zeros_like(v) if ctx.needs_input_grad[10] else None)
        return (None, None, None, None, None, None, None, g["x_init"], None, gz, gv, None, g["all_initial"],
                g["z_jump"] if ctx.needs_input_grad[13] else None, g["v_jump"] if ctx.needs_input_grad[14] else None, *g["de"], *g["ae"])


def fused_dae_integrate(method, kernel, de_layers, ae_layers, x_init, t, z, v, i, all_initial, event_t=None, z_jump=None, v_jump=None,
                        check_events=False, x=None, input_true_x=False, input_true_i=False):
    """Differentiable fused integrate_DAE: gradients flow to x_init, z, v, all_initial, the jump inputs and both MLPs.  Without teacher
    forcing `i` only provides the width of the algebraic variable.  input_true_x / input_true_i: `x` / `i` are the dataset rows the DE
    and the heads are fed (my_solvers.py:111-121); they get no gradient."""
    with torch.no_grad():
        event_idx = fused.event_table(t, event_t, check_events)
    if event_idx is None:
        z_jump = v_jump = None
    else:
        z_jump = z_jump if (z_jump is not None and z_jump.shape[-1] > 0) else None
        v_jump = v_jump if (v_jump is not None and v_jump.shape[-1] > 0) else None
    params = [p for wb in list(de_layers) + list(ae_layers) for p in wb]
    if input_true_x or input_true_i:
        xd_ = x.detach() if input_true_x else x_init.new_zeros((1, t.shape[1], 0))
        return _FusedDaeTeacherForced.apply(method, kernel, event_idx, len(de_layers), bool(input_true_x), bool(input_true_i), t, x_init, xd_,
                                            z, v, i.detach(), all_initial, z_jump, v_jump, *params)
    return _FusedDae.apply(method, kernel, event_idx, len(de_layers), t, x_init, z, v, i.detach(), all_initial, z_jump, v_jump, *params)
