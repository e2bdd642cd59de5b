"""Run-to-run bit-identity of every backward / reduction kernel (K4, K5, K7, K8, K9, row-MLP backward, K6) at a batch that
fills the chip: all of them sum per-workgroup partials in a fixed order, so any difference between two launches on the
same inputs means a data race inside a kernel (LDS exchange buffers, published tiles, transposes)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _grads(model, inputs, loss_fn):
    model.zero_grad()
    outs = model(**inputs)
    loss_fn(outs).backward()
    return [p.grad.clone() for p in model.parameters()]


@pytest.mark.parametrize("tag,H,direct,method", [("ode", 64, False, "rk4"), ("dae", 64, False, "rk4"), ("dae", 64, False, "midpoint"),
                                                 ("ode", 16, True, "rk4"), ("dae", 16, True, "rk4"), ("ode", 64, True, "euler"),
                                                 ("dae", 64, True, "rk4"), ("ode", 32, False, "rk4"), ("dae", 128, False, "euler")])
def test_training_step_is_bit_reproducible(tag, H, direct, method):
    from py_psnode_amd import models, neural_dae as nd
    torch.manual_seed(3)
    B, T = 1024 + 7, 12
    g = torch.Generator().manual_seed(4)
    r = lambda *s: (0.1 * torch.randn(*s, generator=g)).cuda()
    t = (torch.arange(T, dtype=torch.float32) * 0.01).view(1, T, 1).repeat(B, 1, 1).cuda()
    solver = {"euler": nd.Euler, "midpoint": nd.Midpoint, "rk4": nd.RK4}[method]()
    ev = t[:, [3, 8], :].contiguous()
    if tag == "ode":
        model = models.ODE_Model(8, 2, H, direct_encode=direct, solver=solver).cuda()
        inputs = dict(t=t, x=r(B, T, 8), z=r(B, T, 2), event_t=ev, z_jump=r(B, 2, 2))
    else:
        model = models.DAE_Model(8, 2, 2, 2, H, direct_encode=direct, solver=solver).cuda()
        inputs = dict(t=t, x=r(B, T, 8), z=r(B, T, 2), v=r(B, T, 2), i=r(B, T, 2), event_t=ev, z_jump=r(B, 2, 2), v_jump=r(B, 2, 2))
    model.solver.fused = "require"
    loss_fn = lambda outs: sum(((o - 0.03) ** 2).sum() for o in (outs if isinstance(outs, tuple) else (outs,)))
    first = _grads(model, inputs, loss_fn)
    for _ in range(3):
        again = _grads(model, inputs, loss_fn)
        for k, (a, b) in enumerate(zip(first, again)):
            assert torch.equal(a, b), f"parameter {k}: gradients differ between two identical launches"


def test_loss_kernel_is_bit_reproducible():
    from py_psnode_amd import loss as L
    g = torch.Generator().manual_seed(9)
    B, T, D = 2048 + 3, 65, 8
    x = torch.randn(B, T, D, generator=g).cuda()
    pred = (x.permute(1, 0, 2) + 0.1).contiguous().permute(1, 0, 2)
    mask = (torch.rand(B, T, 1, generator=g) > 0.2).float().cuda()
    ref = L.masked_mse_terms(pred, x, mask, want_grad=True)
    for _ in range(3):
        out = L.masked_mse_terms(pred, x, mask, want_grad=True)
        assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])
