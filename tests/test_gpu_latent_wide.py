"""K3w (psnode_latent_wide.hip): the latent shapes of the direct_encode models at every hidden_dim <= 128 other than 16 / 64 -- in
particular the argparse default --hidden 128 of neural_00_ODE_02_direct_encode.py:160-162 / neural_01_DAE_02_direct_encode.py:246-248 --
against the CPU oracle: events, ragged clocks and tiles, B-major strided views, zero-padded widths; and at model level against the
module-by-module route."""
import pytest
import torch
import torch.nn as nn

from helpers import TOL_GPU, traj_rel_err as rel_err
from oracle import psnode_oracle as O

pytestmark = pytest.mark.gpu
METHODS = ["euler", "midpoint", "rk4"]


def fused():
    from py_psnode_amd import fused as f
    return f


def dl(ls):
    return [(w.cuda(), b.cuda()) for w, b in ls]


def _mk(dims):
    return [(l.weight.detach(), l.bias.detach()) for l in [nn.Linear(dims[k], dims[k + 1]) for k in range(len(dims) - 1)]]


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("H", [128, 32, 100, 8, 48])
def test_latent_wide_ode_kernel(method, H):
    B, Tn = 37, 13
    g = torch.Generator().manual_seed(31 + H)
    torch.manual_seed(31 + H)
    ls = _mk([6 * H, H, H])
    t = (torch.arange(Tn, dtype=torch.float32) * 0.01).view(Tn, 1, 1).repeat(1, B, 1)
    t = t * (0.5 + torch.rand(1, B, 1, generator=g))
    t[:, 0] = torch.arange(Tn, dtype=torch.float32).view(Tn, 1) * 0.01
    r = lambda *s: 0.1 * torch.randn(*s, generator=g)
    x, z = r(Tn, B, H), r(Tn, B, H)
    a0 = torch.cat((x[0], z[0]), -1)
    ev = torch.stack([t[2, :, :], t[9, :, :]], dim=1).contiguous()
    zj = r(B, 2, H)
    ref = O.integrate_ode(method, ls, t, x, z, a0, ev, zj)
    bm = lambda a: a.permute(1, 0, 2).contiguous().cuda().permute(1, 0, 2)      # B-major memory, time-major view (as the scripts pass it)
    out = fused().ode_integrate(method, dl(ls), bm(t), bm(x), bm(z), a0.cuda(), event_t=ev.cuda(), z_jump=zj.cuda(), kernel="mfma")
    assert rel_err(out.cpu(), ref) <= TOL_GPU


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("H,with_z", [(128, True), (128, False), (32, True), (100, False), (20, True)])
def test_latent_wide_dae_kernel(method, H, with_z):
    B, Tn = 21, 9
    zd = H if with_z else 0
    g = torch.Generator().manual_seed(41 + H)
    torch.manual_seed(41 + H)
    nblk = 4 if zd else 3
    de, ae = _mk([3 * nblk * H, H, H]), _mk([(2 * nblk - 1) * H, H, H])
    r = lambda *s: 0.1 * torch.randn(*s, generator=g)
    t = (torch.arange(Tn, dtype=torch.float32) * 0.01).view(Tn, 1, 1).repeat(1, B, 1)
    t[:, 1:] = t[:, 1:] * (0.5 + torch.rand(1, B - 1, 1, generator=g))
    x, z, v, i, xi = r(Tn, B, H), r(Tn, B, zd), r(Tn, B, H), r(Tn, B, H), r(B, H)
    a0 = torch.cat((xi, z[0], v[0], i[0]), -1)
    ev = torch.stack([t[3, :, :], t[6, :, :]], dim=1).contiguous()
    zj, vj = r(B, 2, zd), r(B, 2, H)
    ref_x, ref_i = O.integrate_dae(method, de, ae, xi, t, x, z, v, i, a0, ev, zj, vj)
    c = lambda a: a.cuda()
    xs, is_ = fused().dae_integrate(method, dl(de), dl(ae), c(xi), c(t), c(x), c(z), c(v), c(i), c(a0), event_t=c(ev), z_jump=c(zj),
                                    v_jump=c(vj), kernel="mfma")
    assert rel_err(xs.cpu(), ref_x) <= TOL_GPU
    assert rel_err(is_.cpu(), ref_i) <= TOL_GPU


def test_latent_wide_widths_that_stay_on_k0():
    """H % 4 != 0 (rows not 16-byte granular) and H > 128 are not this kernel's: kernel='mfma' refuses, AUTO still matches the oracle."""
    from py_psnode_amd import _lib
    for H in (30, 144):
        torch.manual_seed(H)
        ls = _mk([6 * H, H, H])
        t = (torch.arange(5, dtype=torch.float32) * 0.01).view(5, 1, 1).repeat(1, 6, 1)
        x, z = 0.1 * torch.randn(5, 6, H), 0.1 * torch.randn(5, 6, H)
        a0 = torch.cat((x[0], z[0]), -1)
        with pytest.raises(_lib.UnsupportedShapeError):
            fused().ode_integrate("rk4", dl(ls), t.cuda(), x.cuda(), z.cuda(), a0.cuda(), kernel="mfma")
        if H > 128:          # beyond K0's LDS budget too: no kernel (the model level walks the modules)
            with pytest.raises(_lib.UnsupportedShapeError):
                fused().ode_integrate("rk4", dl(ls), t.cuda(), x.cuda(), z.cuda(), a0.cuda())
            continue
        out = fused().ode_integrate("rk4", dl(ls), t.cuda(), x.cuda(), z.cuda(), a0.cuda())
        assert rel_err(out.cpu(), O.integrate_ode("rk4", ls, t, x, z, a0)) <= TOL_GPU


@pytest.mark.parametrize("kind", ["ode02", "dae02", "dae02_z0"])
def test_models_at_the_argparse_default_hidden_128_run_fused_and_match_the_module_route(kind):
    """The direct_encode models at --hidden 128 under no_grad: encoders / decoders as nn.Sequential (no row kernel at 128), the latent
    loop on K3w (solver.fused = 'require' accepts it) -- equal to the module-by-module walk (solver.fused = 'off')."""
    from py_psnode_amd import models
    from py_psnode_amd import neural_dae as nd
    torch.manual_seed(5)
    B, Tn, H = 33, 12, 128
    zd = 0 if kind.endswith("z0") else 2
    r = lambda *s: (0.1 * torch.randn(*s)).cuda()
    t = (torch.arange(Tn, dtype=torch.float32) * 0.01).view(1, Tn, 1).repeat(B, 1, 1).cuda()
    ev = t[:, [2, 8], :].contiguous()
    if kind == "ode02":
        m = models.ODE_Model(8, 2, H, direct_encode=True, solver=nd.RK4()).cuda()
        kw = dict(t=t, x=r(B, Tn, 8), z=r(B, Tn, 2), event_t=ev, z_jump=r(B, 2, 2))
    else:
        m = models.DAE_Model(8, zd, 2, 2, H, direct_encode=True, solver=nd.RK4()).cuda()
        kw = dict(t=t, x=r(B, Tn, 8), z=r(B, Tn, zd), v=r(B, Tn, 2), i=r(B, Tn, 2), event_t=ev, z_jump=r(B, 2, zd), v_jump=r(B, 2, 2))
    with torch.no_grad():
        m.solver.fused = "require"
        out = m(**kw)
        m.solver.fused = "off"
        ref = m(**kw)
    for o, q in zip(out, ref):
        assert rel_err(o.cpu(), q.cpu(), bdim=0) <= TOL_GPU
