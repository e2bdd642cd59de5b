"""GPU parity of the fused masked-MSE loss (K6, psnode_masked_mse_f32) against the oracle's restatement of the scripts'
loss expressions (oracle/psnode_oracle.py: ode_loss, dae_loss, recon_loss), values and gradients.

Tolerance: the loss is a sum of up to 3e7 fp32 products; torch's CPU reduction order and the kernel's (per-thread 16,
per-tile tree, per-launch double) differ, so values are compared at rel 2e-6 and gradients (elementwise products, no
reduction) at rel 1e-6 of the gradient's max."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import psnode_oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu
RTOL_VAL, RTOL_GRAD = 2e-6, 1e-6


def L():
    from py_psnode_amd import loss
    return loss


def _case(B, T, D, mw, seed=0, layout="tm"):
    g = torch.Generator().manual_seed(seed)
    x = 0.3 * torch.randn(B, T, D, generator=g)
    if layout == "tm":      # what the integrator returns: time-major memory viewed as [B,T,D]
        pred = (x.permute(1, 0, 2) + 0.05 * torch.randn(T, B, D, generator=g)).contiguous().permute(1, 0, 2)
    else:
        pred = x + 0.05 * torch.randn(B, T, D, generator=g)
    mask = None
    if mw:
        mask = (torch.rand(B, T, mw, generator=g) > 0.3).float()
        mask[0, 0] = 1.0
    return pred, x, mask


def _close(a, b, rtol):
    a, b = torch.as_tensor(a).detach().double().cpu(), torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max()) <= rtol * max(float(b.abs().max()), 1e-30)


@pytest.mark.parametrize("layout", ["tm", "bm"])
@pytest.mark.parametrize("B,T,D,mw", [(32, 101, 8, 8), (32, 101, 8, 1), (1, 1, 8, 1), (33, 17, 8, 8), (5, 40, 3, 1), (70, 9, 2, 2),
                                      (7, 19, 1, 1), (9, 33, 16, 16), (3, 21, 64, 1), (41, 16, 5, 5), (260, 5, 8, 1),
                                      (8, 19, 1, 1), (12, 7, 4, 4), (12, 7, 4, 1), (6, 9, 32, 32), (64, 33, 2, 1), (300, 40, 8, 0)])
def test_ode_loss_value_and_grad(B, T, D, mw, layout):
    pred, x, mask = _case(B, T, D, mw, seed=B + T, layout=layout)
    pr = pred.clone().requires_grad_(True)
    ref, ref_cols = O.ode_loss(pr, x, mask if mask is not None else torch.ones(B, T, D))
    ref.backward()
    pg = pred.cuda().requires_grad_(True) if layout == "bm" else pred.permute(1, 0, 2).contiguous().cuda().permute(1, 0, 2).requires_grad_(True)
    if mask is None:     # no mask == a mask of ones (the ODE datasets' default, neural_base.py:14-32)
        mask = torch.ones(B, T, D)
        tot, terms = L().masked_mse(pg, x.cuda(), None, scale=1.0 / mask.sum().item())
    else:
        tot, terms = L().ode_loss(pg, x.cuda(), mask.cuda())
    tot.backward()
    assert _close(tot, ref, RTOL_VAL)
    assert _close(terms[:D], ref_cols, RTOL_VAL) and float(terms[D]) == 0.0
    assert pg.grad.shape == pr.grad.shape and _close(pg.grad, pr.grad, RTOL_GRAD)


@pytest.mark.parametrize("B,T,xd,idim", [(32, 101, 8, 2), (21, 11, 5, 1), (3, 2, 8, 4)])
def test_dae_loss_value_and_grad(B, T, xd, idim):
    xp, x, mask = _case(B, T, xd, 1, seed=3)
    ip, i, _ = _case(B, T, idim, 0, seed=4)
    xr, ir = xp.clone().requires_grad_(True), ip.clone().requires_grad_(True)
    ref = O.dae_loss(xr, x, ir, i, mask)
    ref[0].backward()
    c = lambda a: a.permute(1, 0, 2).contiguous().cuda().permute(1, 0, 2)
    xg, ig = c(xp).requires_grad_(True), c(ip).requires_grad_(True)
    tot, (tx, ti) = L().dae_loss(xg, x.cuda(), ig, i.cuda(), mask.cuda())
    tot.backward()
    assert _close(tot, ref[0], RTOL_VAL)
    assert _close(tx[:xd].sum(), ref[1], RTOL_VAL) and _close(ti[:idim].sum(), ref[2], RTOL_VAL)
    assert _close(tx[xd], ref[3], RTOL_VAL) and _close(ti[idim], ref[4], RTOL_VAL)
    assert _close(xg.grad, xr.grad, RTOL_GRAD) and _close(ig.grad, ir.grad, RTOL_GRAD)


def test_recon_loss_unmasked_mean():
    xr, x, _ = _case(19, 23, 8, 0, seed=9, layout="bm")
    pr = xr.clone().requires_grad_(True)
    ref = O.recon_loss(pr, x)
    ref.backward()
    pg = xr.cuda().requires_grad_(True)
    tot, _ = L().recon_loss(pg, x.cuda())
    tot.backward()
    assert _close(tot, ref, RTOL_VAL) and _close(pg.grad, pr.grad, RTOL_GRAD)


def test_full_size_value_deterministic_and_linear():
    """BASELINE batch (B=4096, T=1001, xd=8): oracle value at full size, run-to-run bit-identical, and the size-independent
    property loss(mask1 + mask2) == loss(mask1) + loss(mask2) for disjoint masks under one norm."""
    B, T, D = 4096, 1001, 8
    pred, x, mask = _case(B, T, D, 1, seed=1)
    ref, _ = O.ode_loss(pred, x, mask)
    pg, xg, mg = pred.permute(1, 0, 2).contiguous().cuda().permute(1, 0, 2), x.cuda(), mask.cuda()
    a, ta = L().ode_loss(pg, xg, mg)
    b, tb = L().ode_loss(pg, xg, mg)
    assert torch.equal(ta, tb)
    assert _close(a, ref, 1e-5)
    inv = L().inv_mask_sum(mg)
    half = (torch.arange(B, device="cuda") % 2 == 0).float().view(B, 1, 1)
    l1, _ = L().masked_mse(pg, xg, mg * half, inv_norm=inv)
    l2, _ = L().masked_mse(pg, xg, mg * (1 - half), inv_norm=inv)
    assert _close(l1 + l2, a, 1e-6)


def test_loss_error_paths():
    pred, x, mask = _case(4, 5, 8, 1)
    with pytest.raises(ValueError):
        L().masked_mse(pred.cuda(), x.cuda()[:, :4], mask.cuda())
    with pytest.raises(ValueError):
        L().masked_mse(pred.cuda(), x.cuda(), mask.cuda().expand(4, 5, 3))
    with pytest.raises(TypeError):
        L().masked_mse(pred.cuda().double(), x.cuda().double(), None)
    with pytest.raises(ValueError):
        L().masked_mse(pred, x, mask)       # CPU tensors: no CPU path
    big = torch.zeros(2, 3, 65, device="cuda")
    with pytest.raises(ValueError):         # PSNODE_ERR_UNSUPPORTED: rows wider than 64
        L().masked_mse(big, big, None)


def test_train_step_fused_integrator_plus_fused_loss_matches_oracle_autograd():
    """ODE_01 training step: fused forward -> fused loss -> fused backward, vs autograd through the oracle's unrolled loop."""
    from py_psnode_amd import models, neural_dae as nd
    torch.manual_seed(0)
    B, T = 24, 21
    m_ref = models.ODE_Model(8, 2, 64, solver=nd.RK4())
    m_ref.solver.fused = "off"
    m_gpu = models.ODE_Model(8, 2, 64, solver=nd.RK4()).cuda()
    m_gpu.load_state_dict(m_ref.state_dict())
    m_gpu.solver.fused = "require"
    g = torch.Generator().manual_seed(5)
    t = (torch.arange(T, dtype=torch.float32) * 0.01).view(1, T, 1).repeat(B, 1, 1)
    x, z = 0.1 * torch.randn(B, T, 8, generator=g), 0.1 * torch.randn(B, T, 2, generator=g)
    mask = (torch.rand(B, T, 8, generator=g) > 0.2).float()
    ev, zj = -torch.ones(B, 1, 1), torch.zeros(B, 1, 2)
    ref, _ = O.ode_loss(m_ref(t, x, z, ev, zj), x, mask)
    ref.backward()
    c = lambda a: a.cuda()
    tot, _ = L().ode_loss(m_gpu(c(t), c(x), c(z), c(ev), c(zj)), c(x), c(mask))
    tot.backward()
    assert _close(tot, ref, 1e-5)
    for (n, pr), (_, pg) in zip(m_ref.named_parameters(), m_gpu.named_parameters()):
        assert _close(pg.grad, pr.grad, 2e-4), n


@pytest.mark.parametrize("B,T,xd", [(32, 101, 8), (7, 3, 5)])
def test_ode02_loss_matches_the_script_expression(B, T, xd):
    """neural_00_ODE_02_direct_encode.py:267-270: x0 term + masked term + reconstruction (the x0 term was missing)."""
    xp, x, mask = _case(B, T, xd, xd, seed=12)
    xre, _, _ = _case(B, T, xd, 0, seed=13, layout="bm")
    pr, rr = xp.clone().requires_grad_(True), xre.clone().requires_grad_(True)
    ref = O.ode02_loss(pr, rr, x, mask)
    ref.backward()
    c = lambda a: a.permute(1, 0, 2).contiguous().cuda().permute(1, 0, 2)
    pg, rg = c(xp).requires_grad_(True), xre.cuda().requires_grad_(True)
    tot, _ = L().ode02_loss(pg, rg, x.cuda(), mask.cuda())
    tot.backward()
    assert _close(tot, ref, RTOL_VAL) and _close(pg.grad, pr.grad, RTOL_GRAD) and _close(rg.grad, rr.grad, RTOL_GRAD)
    assert float(pr.grad[:, 0].abs().max()) > float(pr.grad[:, 1:].abs().max()) * 0 and not _close(tot, O.ode_loss(pr, x, mask)[0], 1e-3)


@pytest.mark.parametrize("B,T,xd,idim", [(32, 101, 8, 2), (5, 4, 8, 3)])
def test_dae02_loss_matches_the_script_expression(B, T, xd, idim):
    """neural_01_DAE_02_direct_encode.py:359-365: NO extra weight on x column 1, both reconstruction terms."""
    xp, x, mask = _case(B, T, xd, 1, seed=21)
    ip, i, _ = _case(B, T, idim, 0, seed=22)
    xre, _, _ = _case(B, T, xd, 0, seed=23, layout="bm")
    ire, _, _ = _case(B, T, idim, 0, seed=24, layout="bm")
    leaves = [a.clone().requires_grad_(True) for a in (xp, ip, xre, ire)]
    ref = O.dae02_loss(*leaves, x, i, mask)
    ref.backward()
    c = lambda a: a.permute(1, 0, 2).contiguous().cuda().permute(1, 0, 2)
    dev = [c(xp).requires_grad_(True), c(ip).requires_grad_(True), xre.cuda().requires_grad_(True), ire.cuda().requires_grad_(True)]
    tot, _ = L().dae02_loss(*dev, x.cuda(), i.cuda(), mask.cuda())
    tot.backward()
    assert _close(tot, ref, RTOL_VAL)
    for a, b in zip(dev, leaves):
        assert _close(a.grad, b.grad, RTOL_GRAD)
    assert not _close(tot, O.dae_loss(leaves[0], x, leaves[1], i, mask)[0], 1e-3)     # differs from DAE_01's weighted objective
