"""torch.autograd bridge for the fused integrator: forward = psnode_ode_integrate_f32, backward = psnode_ode_backward_f32.

This is what lets the reference's training loops (`loss.backward()` through the integrator,
neural_00_ODE_01_no_encode.py:358-360) run on the fused HIP path instead of an unrolled T-step autograd graph.
"""
import torch

from . import fused


class _FusedOde(torch.autograd.Function):
    @staticmethod
    def forward(ctx, method, kernel, event_idx, t, x0, z, all_initial, z_jump, *params):
        layers = [(params[k], params[k + 1]) for k in range(0, len(params), 2)]
        xs = fused.ode_integrate(method, layers, t, x0.unsqueeze(0), z, all_initial, z_jump=z_jump, event_idx=event_idx, kernel=kernel)
        ctx.method = method
        ctx.has_jump = z_jump is not None
        ctx.event_idx = event_idx
        ctx.save_for_backward(t, z, all_initial, xs, *( (z_jump,) if z_jump is not None else () ), *params)
        return xs

    @staticmethod
    def backward(ctx, grad_xs):
        saved = ctx.saved_tensors
        t, z, a0, xs = saved[:4]
        z_jump = saved[4] if ctx.has_jump else None
        params = saved[5 if ctx.has_jump else 4:]
        layers = [(params[k], params[k + 1]) for k in range(0, len(params), 2)]
        need_z = ctx.needs_input_grad[5]
        gx0, gz, gzj, ga0, gpar = fused.ode_backward(ctx.method, layers, t, z, a0, xs, grad_xs, event_idx=ctx.event_idx, z_jump=z_jump,
                                                     need_grad_z=need_z)
        if gz is None and need_z:
            gz = torch.zeros_like(z)
        return (None, None, None, None, gx0, gz, ga0, gzj if ctx.needs_input_grad[7] else None, *gpar)


def fused_ode_integrate(method, kernel, layers, t, x, z, all_initial, event_t=None, z_jump=None):
    """Differentiable fused integrate_ODE (no teacher forcing): gradients flow to x[0], z, all_initial, z_jump and the MLP."""
    with torch.no_grad():
        event_idx = fused.event_table(t, event_t)
    if event_idx is None:
        z_jump = None
    params = [p for wb in layers for p in wb]
    return _FusedOde.apply(method, kernel, event_idx, t, x[0], z, all_initial, z_jump, *params)
