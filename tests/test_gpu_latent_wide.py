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


def _close(a, b, tol, what):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    scale = max(float(b.abs().max()), 1e-6)
    err = float((a - b).abs().max())
    assert err <= tol * scale, f"{what}: err {err:.3e} vs scale {scale:.3e}"


def _train_case(tag, H, zd, method, events, B, T):
    """loss.backward() through encoders -> K3w (saving) -> decoders, backward K9w + library GEMMs, vs the fp64 autograd walk on the CPU;
    solver.fused = 'require': the fused route must take the call."""
    from py_psnode_amd import models, neural_dae as nd
    torch.manual_seed(11)
    g = torch.Generator().manual_seed(12)
    r = lambda *s: 0.1 * torch.randn(*s, generator=g)
    t = (torch.arange(T, dtype=torch.float32) * 0.01).view(1, T, 1).repeat(B, 1, 1)
    x, z, v, i = r(B, T, 8), r(B, T, zd), r(B, T, 2), r(B, T, 2)
    ev = t[:, [2, 6], :].contiguous() if (events and T > 7) else -torch.ones(B, 2, 1)
    zj, vj = r(B, 2, zd), r(B, 2, 2)
    cls = {"euler": nd.Euler, "midpoint": nd.Midpoint, "rk4": nd.RK4}[method]
    if tag == "ode02":
        mk = lambda: models.ODE_Model(8, zd, H, direct_encode=True, solver=cls())
    else:
        mk = lambda: models.DAE_Model(8, zd, 2, 2, H, direct_encode=True, solver=cls())
    m32, m64 = mk(), mk().double()
    m64.load_state_dict({k: v_.double() for k, v_ in m32.state_dict().items()})
    m64.solver.fused = "off"
    m32 = m32.cuda()
    m32.solver.fused = "require"

    def run(model, cast, dev):
        c = lambda a: cast(a).to(dev)
        if tag == "ode02":
            outs = model(t=c(t), x=c(x), z=c(z), event_t=c(ev), z_jump=c(zj))
        else:
            outs = model(t=c(t), x=c(x), z=c(z), v=c(v), i=c(i), event_t=c(ev), z_jump=c(zj), v_jump=c(vj))
        loss = sum(((o - 0.05) ** 2).sum() for o in outs)
        loss.backward()
        return [o.detach() for o in outs]

    ref = run(m64, lambda a: a.double(), "cpu")
    out = run(m32, lambda a: a, "cuda")
    for a, b in zip(out, ref):
        _close(a, b, 1e-5, "model output")
    for (n, p), (_, q) in zip(m32.named_parameters(), m64.named_parameters()):
        if q.grad is None or p.grad is None:
            assert (p.grad is None or float(p.grad.abs().max()) == 0.0) and (q.grad is None or float(q.grad.abs().max()) == 0.0), n
            continue
        _close(p.grad, q.grad, 5e-4, n)


@pytest.mark.parametrize("events", [False, True])
@pytest.mark.parametrize("method", ["euler", "midpoint", "rk4"])
@pytest.mark.parametrize("tag,H,zd", [("ode02", 128, 2), ("dae02", 128, 2), ("dae02", 128, 0), ("ode02", 32, 2), ("dae02", 32, 2), ("dae02", 100, 0)])
def test_direct_encode_training_at_other_hidden_widths_runs_fused_and_matches_fp64(tag, H, zd, method, events):
    """The scripts' argparse default --hidden 128 (and 32 / 100, zero-padded): every parameter gradient of the direct_encode models --
    encoders, decoders, Init_Func, the latent DE and AE -- against the fp64 autograd walk."""
    _train_case(tag, H, zd, method, events, 19, 9)


@pytest.mark.parametrize("B,T", [(3, 2), (17, 1), (1, 3), (33, 4)])
@pytest.mark.parametrize("tag", ["ode02", "dae02"])
def test_latent_wide_backward_edge_sizes(tag, B, T):
    """K9w at the edges: single step, no step at all (T = 1), single trajectory, ragged third tile."""
    _train_case(tag, 128, 2, "rk4", False, B, T)
