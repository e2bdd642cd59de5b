"""Gradients of the fused backward kernel vs an fp64 autograd walk of the same algorithm (the reference's training path:
loss.backward() through integrate_ODE).  Tolerance: 2e-4 of each gradient tensor's max magnitude (fp32 accumulation over
steps x stages x trajectories against an fp64 truth)."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
TOL = 2e-4


def _case(B, Tn, xd, zd, seed, events):
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    n = xd + zd
    dims = [3 * n, 64, 64, 64, xd]
    lin = [nn.Linear(dims[k], dims[k + 1]) for k in range(4)]
    t = (torch.arange(Tn, dtype=torch.float32) * 0.01).view(Tn, 1, 1).repeat(1, B, 1)
    if B > 1:
        t[:, 1:] = t[:, 1:] * (0.5 + torch.rand(1, B - 1, 1, generator=g))
    x = 0.1 * torch.randn(Tn, B, xd, generator=g)
    z = 0.1 * torch.randn(Tn, B, zd, generator=g)
    ev = torch.stack([t[2, :, :], t[Tn - 3, :, :]], dim=1).contiguous() if events else None
    zj = 0.1 * torch.randn(B, 2, zd, generator=g) if events else None
    G = torch.randn(Tn, B, xd, generator=g)
    return lin, t, x, z, ev, zj, G


def _truth(method, lin, t, x, z, ev, zj, G):
    """fp64 autograd through the callback walk (py_psnode_amd's own generic loop == the reference's loop)."""
    from py_psnode_amd import models
    from py_psnode_amd import neural_dae as nd
    xd, zd = x.shape[-1], z.shape[-1]
    de = models.DE_Func(xd + zd, (64, 64, 64), xd).double()
    with torch.no_grad():
        for k, l in enumerate(lin):
            de.x_dot[2 * k].weight.copy_(l.weight.double())
            de.x_dot[2 * k].bias.copy_(l.bias.double())
    solver = {"euler": nd.Euler, "midpoint": nd.Midpoint, "rk4": nd.RK4}[method]()
    solver.fused = "off"
    xq, zq = x.double().requires_grad_(True), z.double().requires_grad_(True)
    zjq = zj.double().requires_grad_(True) if zj is not None else None
    event = nd.ODE_Event()
    if ev is not None:
        event.set_event(ev.double(), zjq)
    a0 = torch.cat((xq[0], zq[0]), -1)
    xs = solver.integrate_ODE(x_func=de, t=t.double(), x=xq, z=zq, all_initial=a0, event_fn=event.event_fn if ev is not None else None,
                              jump_change_fn=event.jump_change_fn if ev is not None else None)
    (xs * G.double()).sum().backward()
    return xs.detach(), xq.grad, zq.grad, (zjq.grad if zjq is not None else None), [p.grad for p in de.x_dot.parameters()]


def _close(a, b, what):
    if b is None:                      # parameter unused by the reference graph (e.g. T = 1): gradient must be zero
        b = torch.zeros(a.shape, dtype=torch.float64)
    scale = float(b.abs().max())
    err = float((a.double().cpu() - b).abs().max())
    assert err <= TOL * max(scale, 1e-6), f"{what}: err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("method", ["euler", "midpoint", "rk4"])
@pytest.mark.parametrize("xd,zd,events", [(8, 2, True), (8, 2, False), (5, 3, True), (3, 0, False), (8, 4, True)])
def test_fused_backward_matches_fp64_autograd(method, xd, zd, events):
    from py_psnode_amd import models
    from py_psnode_amd import neural_dae as nd
    B, Tn = 21, 12
    lin, t, x, z, ev, zj, G = _case(B, Tn, xd, zd, seed=100 + xd * 10 + zd, events=events)
    xs_ref, gx_ref, gz_ref, gzj_ref, gp_ref = _truth(method, lin, t, x, z, ev, zj, G)

    de = models.DE_Func(xd + zd, (64, 64, 64), xd)
    with torch.no_grad():
        for k, l in enumerate(lin):
            de.x_dot[2 * k].weight.copy_(l.weight)
            de.x_dot[2 * k].bias.copy_(l.bias)
    de = de.cuda()
    solver = {"euler": nd.Euler, "midpoint": nd.Midpoint, "rk4": nd.RK4}[method]()
    solver.fused = "require"                      # training must take the fused forward + fused backward route
    xg, zg = x.cuda().requires_grad_(True), z.cuda().requires_grad_(True)
    zjg = zj.cuda().requires_grad_(True) if zj is not None else None
    event = nd.ODE_Event()
    if ev is not None:
        event.set_event(ev.cuda(), zjg)
    a0 = torch.cat((xg[0], zg[0]), -1)
    xs = solver.integrate_ODE(x_func=de, t=t.cuda(), x=xg, z=zg, all_initial=a0, event_fn=event.event_fn if ev is not None else None,
                              jump_change_fn=event.jump_change_fn if ev is not None else None)
    assert xs.grad_fn is not None and type(xs.grad_fn).__name__.startswith("_FusedOde")
    (xs * G.cuda()).sum().backward()
    _close(xs.detach(), xs_ref, "xs")
    _close(xg.grad, gx_ref, "grad x")           # only x[0] gets gradient (directly and through all_initial)
    if zd:
        _close(zg.grad, gz_ref, "grad z")
        if zj is not None:
            _close(zjg.grad, gzj_ref, "grad z_jump")
    for k, (p, r) in enumerate(zip(de.x_dot.parameters(), gp_ref)):
        _close(p.grad, r, f"grad param {k}")


def test_training_step_of_the_script_model_runs_fused():
    """ODE_Model forward + masked MSE + backward + Adam step on the GPU (the scripts' train loop body) on the fused route."""
    from py_psnode_amd import models
    from py_psnode_amd import neural_dae as nd
    torch.manual_seed(0)
    m = models.ODE_Model(8, 2, 64, solver=nd.RK4()).cuda()
    m.solver.fused = "require"
    B, Tn = 64, 50
    t = (torch.arange(Tn, dtype=torch.float32) * 0.01).view(1, Tn, 1).repeat(B, 1, 1).cuda()
    x, z = (0.1 * torch.randn(B, Tn, 8)).cuda(), (0.1 * torch.randn(B, Tn, 2)).cuda()
    opt = torch.optim.Adam(m.parameters(), lr=5e-3)
    losses = []
    for _ in range(5):
        pred = m(t=t, x=x, z=z, event_t=torch.full((B, 1, 1), -1.0).cuda(), z_jump=torch.zeros(B, 1, 2).cuda())
        loss = nn.functional.mse_loss(pred, x)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(torch.isfinite(p.grad).all() for p in m.parameters())
    assert losses[-1] < losses[0]


def _param_grads(seq):
    return [p.grad for p in seq.parameters()]


@pytest.mark.parametrize("method", ["euler", "midpoint", "rk4"])
@pytest.mark.parametrize("xd,zd,H", [(8, 2, 128), (8, 2, 32), (8, 2, 64), (5, 3, 128), (3, 0, 32), (8, 4, 128), (8, 6, 64), (5, 8, 128), (8, 7, 48)])
def test_wide_backward_matches_fp64_autograd(method, xd, zd, H):
    """K4f (the one-launch MFMA backward, recompute and saved-activation forms) at hidden 128 / 32 / 64 / 48 vs the fp64 autograd walk:
    ragged tile, per-trajectory clocks, two events, every NZM class."""
    from py_psnode_amd import fused
    B, Tn = 21, 12
    g = torch.Generator().manual_seed(500 + H + xd)
    torch.manual_seed(500 + H + xd)
    n = xd + zd
    dims = [3 * n, H, H, H, xd]
    lin = [nn.Linear(dims[k], dims[k + 1]) for k in range(4)]
    t = (torch.arange(Tn, dtype=torch.float32) * 0.01).view(Tn, 1, 1).repeat(1, B, 1)
    t[:, 1:] = t[:, 1:] * (0.5 + torch.rand(1, B - 1, 1, generator=g))
    x, z = 0.1 * torch.randn(Tn, B, xd, generator=g), 0.1 * torch.randn(Tn, B, zd, generator=g)
    ev = torch.stack([t[2, :, :], t[Tn - 3, :, :]], dim=1).contiguous() if zd else None
    zj = 0.1 * torch.randn(B, 2, zd, generator=g) if zd else None
    G = torch.randn(Tn, B, xd, generator=g)
    # fp64 truth through this package's walk with a DE_Func of the same width
    from py_psnode_amd import models
    from py_psnode_amd import neural_dae as nd
    de = models.DE_Func(n, (H, H, H), xd).double()
    with torch.no_grad():
        for k, l_ in enumerate(lin):
            de.x_dot[2 * k].weight.copy_(l_.weight.double()); de.x_dot[2 * k].bias.copy_(l_.bias.double())
    solver = {"euler": nd.Euler, "midpoint": nd.Midpoint, "rk4": nd.RK4}[method]()
    solver.fused = "off"
    xq, zq = x.double().requires_grad_(True), z.double().requires_grad_(True)
    zjq = zj.double().requires_grad_(True) if zj is not None else None
    event = nd.ODE_Event()
    if ev is not None:
        event.set_event(ev.double(), zjq)
    a0q = torch.cat((xq[0], zq[0]), -1)
    xs_ref = solver.integrate_ODE(x_func=de, t=t.double(), x=xq, z=zq, all_initial=a0q, event_fn=event.event_fn if ev is not None else None,
                                  jump_change_fn=event.jump_change_fn if ev is not None else None)
    (xs_ref * G.double()).sum().backward()
    layers = [(l_.weight.detach().cuda(), l_.bias.detach().cuda()) for l_ in lin]
    a0 = torch.cat((x[0], z[0]), -1).cuda()
    c = lambda a: None if a is None else a.cuda()
    xs = fused.ode_integrate(method, layers, c(t), c(x), c(z), a0, event_t=c(ev), z_jump=c(zj))
    tab = fused.event_table(c(t), c(ev)) if ev is not None else None
    _close(xs, xs_ref.detach(), "xs")
    # K4f, the width-generic ONE-launch backward (weight gradients accumulated in the kernel): every output against the fp64 truth
    fx0, fz, fzj, fa0, fp = fused.ode_backward(method, layers, c(t), c(z), a0, xs, c(G), event_idx=tab, z_jump=c(zj), kernel="wide")
    _close(fx0 + fa0[:, :xd], xq.grad[0], "K4f grad x0")
    if zd:
        fz_tot = fz.clone(); fz_tot[0] += fa0[:, xd:]
        _close(fz_tot, zq.grad, "K4f grad z")
        _close(fzj, zjq.grad, "K4f grad z_jump")
    for k, (a_, p_) in enumerate(zip(fp, de.x_dot.parameters())):
        _close(a_, p_.grad, f"K4f grad param {k}")
    # the same backward fed with the activations the FORWARD saved (ode_integrate(save=True): no recompute in K4f): the forward result must
    # not depend on saving, the gradients must meet the same truth
    xs_s, saved = fused.ode_integrate(method, layers, c(t), c(x), c(z), a0, event_t=c(ev), z_jump=c(zj), save=True)
    # (round 5: the inference forward runs its hidden layers in the log2e-scaled domain -- K1x / K1 -- the saving one in the plain domain:
    #  the same function, different roundings)
    _close(xs_s, xs.double().cpu(), "training forward (saving) vs inference forward")
    S_ = {"euler": 1, "midpoint": 2, "rk4": 4}[method]
    assert saved[0].shape == (Tn - 1, S_, 3, B, 32 if H <= 32 else (64 if H <= 64 else 128)) and saved[1].shape == (Tn - 1, S_, B, xd)
    sx0, sz, szj, sa0, sp = fused.ode_backward(method, layers, c(t), c(z), a0, xs, c(G), event_idx=tab, z_jump=c(zj), saved=saved)
    _close(sx0 + sa0[:, :xd], xq.grad[0], "K4f(saved) grad x0")
    if zd:
        sz_tot = sz.clone(); sz_tot[0] += sa0[:, xd:]
        _close(sz_tot, zq.grad, "K4f(saved) grad z")
        _close(szj, zjq.grad, "K4f(saved) grad z_jump")
    for k, (a_, p_) in enumerate(zip(sp, de.x_dot.parameters())):
        _close(a_, p_.grad, f"K4f(saved) grad param {k}")
    # the auto route is the same kernel at every width (K4 at hidden 64 is gone since round 5): bit-identical
    auto = fused.ode_backward(method, layers, c(t), c(z), a0, xs, c(G), event_idx=tab, z_jump=c(zj))
    for k, (a_, b_) in enumerate(zip(fp, auto[4])):
        assert torch.equal(a_, b_), f"auto vs wide param {k}"


@pytest.mark.parametrize("method", ["euler", "midpoint", "rk4"])
@pytest.mark.parametrize("H,xd,zd,B,Tn,events", [(64, 8, 2, 4096, 5, False), (64, 8, 2, 37, 9, True), (40, 5, 3, 18, 6, True), (32, 3, 0, 7, 4, False),
                                                 (20, 7, 8, 130, 3, False), (64, 1, 1, 4, 2, False)])
def test_training_forward_saves_the_same_rows_on_both_mfma_integrators(method, H, xd, zd, B, Tn, events):
    """The training forward (save=True) on K1x (one wave per 4 trajectories: the default up to 4608 trajectories) and on K1 (the 4-wave tile)
    writes the same rows -- stage inputs [T-1,S,B,xd], the three ELU layers [T-1,S,3,B,Hp] incl. the zero padding -- to rounding, odd x_dim,
    events, ragged tiles and the padded widths included, and K4f returns the same gradients from either."""
    from py_psnode_amd import fused
    g = torch.Generator().manual_seed(H * 7 + xd + B)
    torch.manual_seed(H * 7 + xd + B)
    lin = [nn.Linear(a_, b_) for a_, b_ in zip([3 * (xd + zd), H, H, H], [H, H, H, xd])]
    layers = [(m.weight.detach().cuda(), m.bias.detach().cuda()) for m in lin]
    t = (torch.arange(Tn, dtype=torch.float32) * 0.02).view(Tn, 1, 1).repeat(1, B, 1).cuda()
    r = lambda *s_: (0.1 * torch.randn(*s_, generator=g)).cuda()
    x, z = r(Tn, B, xd), r(Tn, B, zd)
    a0 = torch.cat((x[0], z[0]), -1)
    ev = zj = None
    if events and zd:
        ev = torch.full((B, 2, 1), -1.0, device="cuda"); ev[:, 0, 0] = 0.02 * (Tn // 2)
        zj = r(B, 2, zd)
    G = torch.randn(Tn, B, xd, generator=g).cuda()
    tab = fused.event_table(t, ev) if ev is not None else None
    out = {}
    for kern in ("tile", "wave"):
        xs, saved = fused.ode_integrate(method, layers, t, x, z, a0, event_t=ev, z_jump=zj, save=True, kernel=kern)
        grads = fused.ode_backward(method, layers, t, z, a0, xs, G, event_idx=tab, z_jump=zj, saved=saved)
        out[kern] = (xs, saved, grads)
    (xs_t, sv_t, g_t), (xs_w, sv_w, g_w) = out["tile"], out["wave"]
    _close(xs_w, xs_t.double().cpu(), "xs")
    assert sv_w[0].shape == sv_t[0].shape and sv_w[1].shape == sv_t[1].shape
    _close(sv_w[1], sv_t[1].double().cpu(), "saved stage inputs")
    _close(sv_w[0], sv_t[0].double().cpu(), "saved activations")
    hp = sv_t[0].shape[-1]
    if H < hp:
        assert float(sv_w[0][..., H:].abs().max()) == 0.0, "padding units are stored as zeros"
    for k, (a_, b_) in enumerate(zip(g_w[4], g_t[4])):
        _close(a_, b_.double().cpu(), f"grad param {k}")
    _close(g_w[0], g_t[0].double().cpu(), "grad x0")
    _close(g_w[3], g_t[3].double().cpu(), "grad all_initial")


@pytest.mark.parametrize("method", ["euler", "midpoint", "rk4"])
@pytest.mark.parametrize("H,xd", [(128, 8), (100, 8), (128, 5), (64, 8), (32, 3)])
def test_wide_backward_without_external_inputs_at_hidden_128(method, H, xd):
    """z_dim = 0 (no external-input slots: the NZM = 0 instances) on K4f, recompute and saved, against K5 -- the recompute instance
    <Midpoint, NZM = 0, 8 waves> returned a wrong dL/dall_initial / dW1 until the end of round 3 (found by the shape fuzz)."""
    from py_psnode_amd import fused
    g = torch.Generator().manual_seed(300 + H + xd)
    torch.manual_seed(300 + H + xd)
    B, Tn, zd = 48, 7, 0
    lin = [nn.Linear(a_, b_) for a_, b_ in zip([3 * (xd + zd), H, H, H], [H, H, H, xd])]
    layers = [(m.weight.detach().cuda(), m.bias.detach().cuda()) for m in lin]
    t = (torch.arange(Tn, dtype=torch.float32) * 0.02).view(Tn, 1, 1).repeat(1, B, 1).cuda()
    r = lambda *s_: (0.1 * torch.randn(*s_, generator=g)).cuda()
    x_in, z = torch.zeros(Tn, B, xd, device="cuda"), r(Tn, B, zd)
    x_in[0] = r(B, xd)
    a0 = torch.cat((x_in[0], z[0]), -1)
    G = torch.randn(Tn, B, xd, generator=g).cuda()
    xs, saved = fused.ode_integrate(method, layers, t, x_in, z, a0, save=True)
    b = fused.ode_backward(method, layers, t, z, a0, xs, G, kernel="generic")
    for name, kw in (("K4f", {}), ("K4f saved", {"saved": saved})):
        a = fused.ode_backward(method, layers, t, z, a0, xs, G, kernel="wide", **kw)
        _close(a[0], b[0].double().cpu(), f"{name} grad x0")
        _close(a[3], b[3].double().cpu(), f"{name} grad all_initial")
        for k, (p_, q_) in enumerate(zip(a[4], b[4])):
            _close(p_, q_.double().cpu(), f"{name} grad param {k}")


def test_hidden128_training_takes_the_one_launch_backward():
    """ODE_Model at --hidden 128 (the scripts' argparse default) under autograd with fused='require': forward K1, backward K4f in ONE
    launch."""
    from py_psnode_amd import fused, models
    from py_psnode_amd import neural_dae as nd
    torch.manual_seed(1)
    m = models.ODE_Model(8, 2, 128, solver=nd.RK4()).cuda()
    m.solver.fused = "require"
    B, Tn = 40, 30
    t = (torch.arange(Tn, dtype=torch.float32) * 0.01).view(1, Tn, 1).repeat(B, 1, 1).cuda()
    x, z = (0.1 * torch.randn(B, Tn, 8)).cuda(), (0.1 * torch.randn(B, Tn, 2)).cuda()
    pred = m(t=t, x=x, z=z, event_t=torch.full((B, 1, 1), -1.0).cuda(), z_jump=torch.zeros(B, 1, 2).cuda())
    nn.functional.mse_loss(pred, x).backward()
    # ... and by default (PSNODE_SAVE_ACTIVATIONS=auto) the training forward at this width saves its activations for the backward
    from py_psnode_amd import autograd as pag
    layers_ = fused.de_layers_of(m.de_func, 10, 8)
    assert pag._want_saved("rk4", "auto", layers_, 8, 2, Tn, B) is True
    assert pag._want_saved("rk4", "auto", [(w[:32, :] if k == 0 else (w[:32, :32] if k < 3 else w[:, :32]), b_[:32] if k < 3 else b_)
                                           for k, (w, b_) in enumerate(layers_)], 8, 2, Tn, B) is True      # ... and so does hidden 32
    assert pag._want_saved("rk4", "generic", layers_, 8, 2, Tn, B) is False
    assert all(p.grad is not None and torch.isfinite(p.grad).all() and float(p.grad.abs().max()) > 0 for p in m.parameters())


def test_dae_hidden128_training_takes_the_fused_backward_without_library_gemms():
    """DAE_Model at --hidden 128 (the scripts' argparse default, neural_01_DAE_01_no_encode.py) under autograd with fused='require':
    forward K2 saving its activations, backward K7f (DE gradients in the kernel) + K7h (AE head contractions): no torch matmul / bmm
    over stored rows; two runs are bit-identical."""
    from py_psnode_amd import fused, models
    from py_psnode_amd import autograd as pag
    from py_psnode_amd import neural_dae as nd
    torch.manual_seed(2)
    m = models.DAE_Model(8, 2, 2, 2, 128, solver=nd.RK4()).cuda()
    m.solver.fused = "require"
    B, Tn = 40, 24
    t = (torch.arange(Tn, dtype=torch.float32) * 0.01).view(1, Tn, 1).repeat(B, 1, 1).cuda()
    r = lambda *s_: (0.1 * torch.randn(*s_)).cuda()
    x, z, v, i = r(B, Tn, 8), r(B, Tn, 2), r(B, Tn, 2), r(B, Tn, 2)
    ev = torch.stack([t[:, 5], t[:, 17]], dim=1).contiguous()           # two event steps
    zj, vj = r(B, 2, 2), r(B, 2, 2)
    de, ae = fused.de_layers_of(m.de_func, 14, 8), fused.ae_layers_of(m.ae_func, 14, 12, 2)
    assert pag._want_saved_dae("rk4", "auto", de, ae, 8, 2, 2, 2, Tn, B) is True
    seen = {"head": 0, "mm": 0}
    orig_wide, orig_gemm = fused.backward_dae.dae_backward_wide, fused.latent._gemm_tn

    def wide(*a, **k):
        assert k.get("saved") is not None, "the training forward must have saved its activations"
        return orig_wide(*a, **k)

    grads = []
    try:
        fused.backward_dae.dae_backward_wide = wide       # (the module-level name dae_backward resolves)
        fused.latent._gemm_tn = lambda *a, **k: (seen.__setitem__("mm", seen["mm"] + 1), orig_gemm(*a, **k))[1]
        for _ in range(2):
            m.zero_grad()
            xp, ip = m(t=t, x=x, z=z, v=v, i=i, event_t=ev, z_jump=zj, v_jump=vj)
            (nn.functional.mse_loss(xp, x) + nn.functional.mse_loss(ip, i)).backward()
            grads.append([p.grad.clone() for p in m.parameters()])
    finally:
        fused.backward_dae.dae_backward_wide, fused.latent._gemm_tn = orig_wide, orig_gemm
    assert seen["mm"] == 0
    assert all(torch.isfinite(g_).all() and float(g_.abs().max()) > 0 for g_ in grads[0])
    assert all(torch.equal(p_, q_) for p_, q_ in zip(*grads)), "training step not bit-reproducible"


@pytest.mark.parametrize("hidden,R,B,nzv,ev_layout", [(128, 7, 37, 4, False), (20, 3, 16, 0, False), (64, 4, 21, 7, True), (32, 1, 5, 2, False)])
def test_head_grads_kernel_against_torch(hidden, R, B, nzv, ev_layout):
    """K7h (psnode_dae_head_grads_f32) on random rows against fp64 einsums: every block of `out`, sa1, grad_zv; zero-padded hidden
    widths, ragged tile, the [nE,3,B,Hp] row layout of the forward call's event buffers."""
    import ctypes
    from py_psnode_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(hidden + R)
    Hp = 32 if hidden <= 32 else (64 if hidden <= 64 else 128)
    def rows(*s_):
        q = torch.randn(*s_, generator=g)
        q[..., hidden:] = 0.0           # units beyond the real width are zero padding in every row the kernels write
        return q
    if ev_layout:
        hbuf = rows(R, 3, B, Hp).cuda()
        act, stride = [hbuf[:, q] for q in range(3)], 3 * B * Hp
    else:
        hbuf = rows(3, R, B, Hp).cuda()
        act, stride = [hbuf[q] for q in range(3)], B * Hp
    delta = [rows(R, B, Hp).cuda() for _ in range(3)]
    gi, u = torch.randn(R, B, 16, generator=g).cuda(), torch.randn(R, B, 16, generator=g).cuda()
    k1a, zv0 = 30, 20
    aw1 = torch.randn(hidden, k1a, generator=g).cuda()
    a = _lib.DaeHeadGradsArgsF32()
    a.R, a.B, a.hidden, a.n_zv = R, B, hidden, nzv
    for q in range(3):
        a.act[q], a.delta[q] = act[q].data_ptr(), delta[q].data_ptr()
    a.act_row_stride = stride
    a.gi, a.u, a.aw1, a.aw1_cols, a.zv_col0 = gi.data_ptr(), u.data_ptr(), aw1.data_ptr(), k1a, zv0
    gza = torch.full((R, B, 8), float("nan"), device="cuda")
    sa1 = torch.empty(B, Hp, device="cuda")
    out = torch.empty(lib.psnode_dae_head_grads_out_floats(hidden), device="cuda")
    a.grad_zv, a.sa1, a.out = (gza.data_ptr() if nzv else None), sa1.data_ptr(), out.data_ptr()
    nb = lib.psnode_dae_head_grads_workspace_bytes(ctypes.byref(a))
    ws = torch.empty(nb + 256, dtype=torch.uint8, device="cuda")
    wp = (ws.data_ptr() + 255) // 256 * 256
    _lib.check(lib.psnode_dae_head_grads_f32(ctypes.byref(a), wp, nb, torch.cuda.current_stream().cuda_stream), "head_grads")
    d64 = [q.double().cpu()[..., :hidden] for q in delta]
    h64 = [q.double().cpu()[..., :hidden] for q in act]
    gi64, u64 = gi.double().cpu(), u.double().cpu()
    o = 0
    for name, ref in [("dAW2", torch.einsum("rbu,rbv->uv", d64[1], h64[0])), ("dAW3", torch.einsum("rbu,rbv->uv", d64[2], h64[1])),
                      ("P3", torch.einsum("rbs,rbv->sv", gi64, h64[2])), ("P0", torch.einsum("rbu,rbc->uc", d64[0], u64)),
                      ("db", torch.stack([q.sum((0, 1)) for q in d64])), ("sum gi", gi64.sum((0, 1)))]:
        got = out[o:o + ref.numel()].view(ref.shape); o += ref.numel()
        _close(got, ref, name)
    assert o == out.numel()
    _close(sa1[:, :hidden], d64[0].sum(0), "sa1")
    if nzv:
        _close(gza[..., :nzv], torch.einsum("rbu,uc->rbc", d64[0], aw1.double().cpu()[:, zv0:zv0 + nzv]), "grad_zv")


@pytest.mark.parametrize("method", ["euler", "rk4"])
@pytest.mark.parametrize("xd,zd,H,nh", [(8, 2, 64, 3), (16, 16, 16, 1), (5, 3, 24, 2), (8, 2, 128, 3), (8, 2, 32, 3),
                                        # round 6, the DE's register path (<= 4 layers of <= 64 units, 3 n <= 128 input columns) and its edges:
                                        (20, 2, 64, 3), (32, 4, 48, 2), (17, 0, 33, 3), (40, 2, 64, 1), (8, 2, (33, 17, 64), 3), (5, 3, (16, 64, 16), 3),
                                        (44, 0, 40, 2),            # 132 input columns: back on the staged path
                                        (8, 2, 64, 4)])            # five Linear layers: staged path
def test_generic_backward_kernel_ode(method, xd, zd, H, nh):
    """K5 (kernel='generic') on the MFMA shape, the direct_encode latent shape, an odd shape, and the --hidden 128 / 32 shapes
    (at 128 the parameter-gradient accumulators no longer fit LDS and live in the workgroup's global partial slice), raw-tensor API."""
    from py_psnode_amd import fused
    B, Tn = 19, 9
    g = torch.Generator().manual_seed(7)
    torch.manual_seed(7)
    n = xd + zd
    dims = [3 * n] + (list(H) if isinstance(H, tuple) else [H] * nh) + [xd]
    seq64 = nn.Sequential(*[m for k in range(len(dims) - 1) for m in ([nn.Linear(dims[k], dims[k + 1])] + ([nn.ELU()] if k + 2 < len(dims) else []))]).double()
    t = (torch.arange(Tn, dtype=torch.float32) * 0.01).view(Tn, 1, 1).repeat(1, B, 1)
    x0, z = 0.1 * torch.randn(B, xd, generator=g), 0.1 * torch.randn(Tn, B, zd, generator=g)
    G = torch.randn(Tn, B, xd, generator=g)
    # fp64 truth through the oracle-style loop with autograd
    from py_psnode_amd import neural_dae as nd
    x0q, zq = x0.double().requires_grad_(True), z.double().requires_grad_(True)
    a0q = torch.cat((x0q, zq[0]), -1)
    f = lambda xx, zz: seq64(torch.cat((a0q, torch.cat((xx, zz), -1) - a0q, torch.cat((xx, zz), -1)), -1))
    solver = {"euler": nd.Euler, "rk4": nd.RK4}[method]()
    solver.fused = "off"
    xq = torch.zeros(Tn, B, xd, dtype=torch.float64)
    xq = torch.cat((x0q.unsqueeze(0), xq[1:]), 0)
    xs_ref = solver.integrate_ODE(x_func=lambda t0, xt, zt, all_initial: f(xt, zt), t=t.double(), x=xq, z=zq, all_initial=a0q)
    (xs_ref * G.double()).sum().backward()
    lin = [m for m in seq64 if isinstance(m, nn.Linear)]
    layers = [(m.weight.detach().float().cuda(), m.bias.detach().float().cuda()) for m in lin]
    x_in = torch.zeros(Tn, B, xd)
    x_in[0] = x0
    a0 = torch.cat((x0, z[0]), -1).cuda()
    xs = fused.ode_integrate(method, layers, t.cuda(), x_in.cuda(), z.cuda(), a0)
    gx0, gz, gzj, ga0, gp = fused.ode_backward(method, layers, t.cuda(), z.cuda(), a0, xs, G.cuda(), kernel="generic")
    _close(xs, xs_ref.detach(), "xs")
    # x0 and z[0] also enter through all_initial: compare the totals autograd reports
    gx0_tot = gx0 + ga0[:, :xd]
    _close(gx0_tot, x0q.grad, "grad x0")
    if zd:
        gz_tot = gz.clone()
        gz_tot[0] += ga0[:, xd:]
        _close(gz_tot, zq.grad, "grad z")
    for k, (a, m) in enumerate(zip(gp, [q for mm in lin for q in (mm.weight, mm.bias)])):
        _close(a, m.grad, f"grad param {k}")


@pytest.mark.parametrize("method", ["euler", "midpoint", "rk4"])
@pytest.mark.parametrize("events", [False, True])
def test_fused_dae_backward_matches_fp64_autograd(method, events):
    """DAE_Model-style training step: Init_Func -> integrate_DAE (fused forward + generic fused backward) vs fp64 autograd walk."""
    from py_psnode_amd import models
    from py_psnode_amd import neural_dae as nd
    B, Tn, xd, zd, vd, idim, H = 13, 8, 8, 2, 2, 2, 64
    g = torch.Generator().manual_seed(3)
    torch.manual_seed(3)
    m32 = models.DAE_Model(xd, zd, vd, idim, H)
    m64 = models.DAE_Model(xd, zd, vd, idim, H).double()
    m64.load_state_dict({k: v.double() for k, v in m32.state_dict().items()})
    cls = {"euler": nd.Euler, "midpoint": nd.Midpoint, "rk4": nd.RK4}[method]
    m64.solver = cls(); m64.solver.fused = "off"
    m32 = m32.cuda(); m32.solver = cls(); m32.solver.fused = "require"
    t = (torch.arange(Tn, dtype=torch.float32) * 0.01).view(1, Tn, 1).repeat(B, 1, 1)
    r = lambda *s: 0.1 * torch.randn(*s, generator=g)
    x, z, v, i = r(B, Tn, xd), r(B, Tn, zd), r(B, Tn, vd), r(B, Tn, idim)
    ev = (t[:, [2, 5], :] if events else torch.full((B, 2, 1), -1.0)).contiguous()
    zj, vj = r(B, 2, zd), r(B, 2, vd)
    Gx, Gi = torch.randn(B, Tn, xd, generator=g), torch.randn(B, Tn, idim, generator=g)

    def run(model, cast, dev):
        c = lambda a: cast(a).to(dev)
        zq, vq, iq = c(z).requires_grad_(True), c(v).requires_grad_(True), c(i).requires_grad_(True)
        zjq, vjq = c(zj).requires_grad_(True), c(vj).requires_grad_(True)
        xs, is_ = model(t=c(t), x=c(x), z=zq, v=vq, i=iq, event_t=c(ev), z_jump=zjq, v_jump=vjq)
        ((xs * c(Gx)).sum() + (is_ * c(Gi)).sum()).backward()
        return xs.detach(), is_.detach(), zq.grad, vq.grad, iq.grad, zjq.grad, vjq.grad, [p.grad for p in model.parameters()]

    ref = run(m64, lambda a: a.double(), "cpu")
    out = run(m32, lambda a: a, "cuda")
    names = ["xs", "is", "grad z", "grad v", "grad i", "grad z_jump", "grad v_jump"]
    for nme, a, b in zip(names, out[:7], ref[:7]):
        if nme.endswith("_jump") and not events:
            continue
        _close(a, b, nme)
    for k, (a, b) in enumerate(zip(out[7], ref[7])):
        _close(a, b, f"grad param {k}")


@pytest.mark.parametrize("kernel", ["mfma", "generic", "wide"])
@pytest.mark.parametrize("B,Tn", [(1, 2), (3, 1), (17, 2), (33, 3)])
def test_backward_edge_sizes(B, Tn, kernel):
    """T = 1 (no step at all), T = 2, single trajectory, ragged tiles -- both backward kernels against fp64 autograd."""
    from py_psnode_amd import fused
    lin, t, x, z, _, _, G = _case(B, Tn, 8, 2, seed=900 + B, events=False)
    xs_ref, gx_ref, gz_ref, _, gp_ref = _truth("rk4", lin, t, x, z, None, None, G)
    layers = [(l.weight.detach().cuda(), l.bias.detach().cuda()) for l in lin]
    a0 = torch.cat((x[0], z[0]), -1).cuda()
    xs = fused.ode_integrate("rk4", layers, t.cuda(), x.cuda(), z.cuda(), a0)
    gx0, gz, _, ga0, gp = fused.ode_backward("rk4", layers, t.cuda(), z.cuda(), a0, xs, G.cuda(), kernel=kernel)
    _close(xs, xs_ref, "xs")
    _close(gx0 + ga0[:, :8], gx_ref[0], "grad x0")
    gz_tot = gz.clone()
    gz_tot[0] += ga0[:, 8:]
    _close(gz_tot, gz_ref, "grad z")
    for k, (a, b) in enumerate(zip(gp, gp_ref)):
        _close(a, b, f"grad param {k}")


@pytest.mark.parametrize("xd,zd,vd,idim,H", [(5, 0, 3, 2, 24), (8, 4, 6, 6, 64), (12, 3, 4, 9, 40), (20, 10, 20, 10, 64)])
def test_dae_backward_without_z_and_odd_widths(xd, zd, vd, idim, H):
    """DAE generic backward with z_dim = 0 and non-multiple-of-16 widths (raw-tensor API) vs fp64 autograd of the oracle loop; round 6: the
    shapes outside K7f's classes (z + v + i > 8, x_dim > 8) with the DE on K5's register path, and 180 input columns (staged path)."""
    from oracle import psnode_oracle as O
    from py_psnode_amd import fused
    B, Tn = 9, 6
    g = torch.Generator().manual_seed(77)
    torch.manual_seed(77)
    n = xd + zd + vd + idim
    mk = lambda dims: [nn.Linear(dims[k], dims[k + 1]) for k in range(len(dims) - 1)]
    de_l, ae_l = mk([3 * n, H, H, xd]), mk([n + xd + zd + vd, H, idim])
    r = lambda *s: 0.1 * torch.randn(*s, generator=g)
    t = (torch.arange(Tn, dtype=torch.float32) * 0.02).view(Tn, 1, 1).repeat(1, B, 1)
    z, v, xi, i0 = r(Tn, B, zd), r(Tn, B, vd), r(B, xd), r(B, idim)
    Gx, Gi = torch.randn(Tn, B, xd, generator=g), torch.randn(Tn, B, idim, generator=g)
    # fp64 truth: differentiable restatement of the DAE loop (no events, no teacher forcing)
    D = lambda a: a.double()
    leaf = lambda p_: p_.detach().double().requires_grad_(True)
    de64 = [(leaf(l.weight), leaf(l.bias)) for l in de_l]
    ae64 = [(leaf(l.weight), leaf(l.bias)) for l in ae_l]
    vq, xiq = D(v).requires_grad_(True), D(xi).requires_grad_(True)
    a0q = torch.cat((xiq, D(z[0]), vq[0], D(i0)), -1)
    x_cur, i_cur = xiq, O.ae_rhs(ae64, xiq, D(z[0]), vq[0], a0q)
    xs_l, is_l = [x_cur], [i_cur]
    for k in range(Tn - 1):
        dt = D(t[k + 1] - t[k])
        x_cur, _ = O.step("rk4", lambda xx: O.de_rhs(de64, xx, (D(z[k]), vq[k], i_cur), a0q), D(t[k]), dt, D(t[k + 1]), x_cur)
        i_cur = O.ae_rhs(ae64, x_cur, D(z[k + 1]), vq[k + 1], a0q)
        xs_l.append(x_cur); is_l.append(i_cur)
    xs_ref, is_ref = torch.stack(xs_l), torch.stack(is_l)
    ((xs_ref * D(Gx)).sum() + (is_ref * D(Gi)).sum()).backward()
    c = lambda a: a.cuda()
    de = [(c(l.weight.detach()), c(l.bias.detach())) for l in de_l]
    ae = [(c(l.weight.detach()), c(l.bias.detach())) for l in ae_l]
    a0 = torch.cat((xi, z[0], v[0], i0), -1)
    xe = torch.zeros(Tn, B, 0)
    xs, is_ = fused.dae_integrate("rk4", de, ae, c(xi), c(t), c(xe), c(z), c(v), c(torch.zeros(Tn, B, idim)), c(a0))
    gr = fused.dae_backward("rk4", de, ae, c(t), c(z), c(v), c(a0), xs, is_, c(Gx), c(Gi))
    _close(xs, xs_ref.detach(), "xs"); _close(is_, is_ref.detach(), "is")
    _close(gr["x_init"] + gr["all_initial"][:, :xd], xiq.grad, "grad x_init")
    gv = gr["v"].clone(); gv[0] += gr["all_initial"][:, xd + zd:xd + zd + vd]
    _close(gv, vq.grad, "grad v")
    flat_ref = [q.grad for wb in de64 for q in wb]
    for k, (a, b) in enumerate(zip(gr["de"], flat_ref)):
        _close(a, b, f"grad de {k}")
    for k, (a, b) in enumerate(zip(gr["ae"], [q.grad for wb in ae64 for q in wb])):
        _close(a, b, f"grad ae {k}")


def _dae_raw_case(B, Tn, xd, zd, vd, idim, seed, events, H=64):
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    n = xd + zd + vd + idim
    mk = lambda dims: [(l.weight.detach().cuda(), l.bias.detach().cuda()) for l in [nn.Linear(dims[k], dims[k + 1]) for k in range(len(dims) - 1)]]
    de, ae = mk([3 * n, H, H, H, xd]), mk([n + xd + zd + vd, H, H, H, idim])
    r = lambda *s: (0.1 * torch.randn(*s, generator=g)).cuda()
    t = (torch.arange(Tn, dtype=torch.float32) * 0.02).view(Tn, 1, 1).repeat(1, B, 1)
    if B > 1:
        t[:, 1:] = t[:, 1:] * (0.5 + torch.rand(1, B - 1, 1, generator=g))      # per-trajectory clocks
    z, v, xi, i0 = r(Tn, B, zd), r(Tn, B, vd), r(B, xd), r(B, idim)
    a0 = torch.cat((xi, z[0], v[0], i0), -1)
    ev = zj = vj = None
    if events and Tn > 3:
        ev = torch.stack([t[1, :, :], t[Tn - 2, :, :]], dim=1).contiguous().cuda()
        zj, vj = r(B, 2, zd), r(B, 2, vd)
    Gx, Gi = torch.randn(Tn, B, xd, generator=g).cuda(), torch.randn(Tn, B, idim, generator=g).cuda()
    return de, ae, t.cuda(), z, v, xi, a0, ev, zj, vj, Gx, Gi


def _dae_both_kernels(method, B, Tn, xd, zd, vd, idim, seed, events, with_gi=True):
    from py_psnode_amd import fused
    de, ae, t, z, v, xi, a0, ev, zj, vj, Gx, Gi = _dae_raw_case(B, Tn, xd, zd, vd, idim, seed, events)
    xe, ie = torch.zeros(Tn, B, 0, device="cuda"), torch.zeros(Tn, B, idim, device="cuda")
    xs, is_ = fused.dae_integrate(method, de, ae, xi, t, xe, z, v, ie, a0, event_t=ev, z_jump=zj, v_jump=vj)
    tab = fused.event_table(t, ev) if ev is not None else None
    out = {}
    for kern in ("mfma", "generic"):
        out[kern] = fused.dae_backward(method, de, ae, t, z, v, a0, xs, is_, Gx, Gi if with_gi else None, event_idx=tab, z_jump=zj,
                                       v_jump=vj, kernel=kern)
    a, b = out["mfma"], out["generic"]
    for key in ("x_init", "z", "v", "z_jump", "v_jump", "all_initial"):
        if b[key] is None:
            assert a[key] is None, key
            continue
        _close(a[key], b[key].double().cpu(), f"{key} (K7 vs K5)")
    for grp in ("de", "ae"):
        for k, (p, q) in enumerate(zip(a[grp], b[grp])):
            _close(p, q.double().cpu(), f"grad {grp} {k} (K7 vs K5)")


@pytest.mark.parametrize("method", ["euler", "midpoint", "rk4"])
@pytest.mark.parametrize("xd,zd,vd,idim", [(8, 2, 2, 2), (8, 0, 2, 2), (5, 1, 1, 1), (8, 2, 2, 4), (3, 1, 0, 1), (8, 4, 3, 1), (2, 2, 4, 2)])
def test_dae_mfma_backward_matches_generic_every_shape_class(xd, zd, vd, idim, method):
    """K7 (MFMA DAE backward) against K5 (generic backward, itself checked against fp64 autograd above) on every (NZM, NZA)
    register class, with two event steps, per-trajectory clocks and a ragged tile."""
    _dae_both_kernels(method, 21, 9, xd, zd, vd, idim, seed=40 + xd + zd, events=True)


def _dae_wide_vs_generic(method, H, B, Tn, xd, zd, vd, idim, seed, events, with_gi=True, slice_step=None):
    from py_psnode_amd import fused
    de, ae, t, z, v, xi, a0, ev, zj, vj, Gx, Gi = _dae_raw_case(B, Tn, xd, zd, vd, idim, seed, events, H=H)
    xe, ie = torch.zeros(Tn, B, 0, device="cuda"), torch.zeros(Tn, B, idim, device="cuda")
    xs, is_ = fused.dae_integrate(method, de, ae, xi, t, xe, z, v, ie, a0, event_t=ev, z_jump=zj, v_jump=vj)
    tab = fused.event_table(t, ev) if ev is not None else None
    gi = Gi if with_gi else None
    b = fused.dae_backward(method, de, ae, t, z, v, a0, xs, is_, Gx, gi, event_idx=tab, z_jump=zj, v_jump=vj, kernel="generic")
    if slice_step is None:
        a = fused.dae_backward(method, de, ae, t, z, v, a0, xs, is_, Gx, gi, event_idx=tab, z_jump=zj, v_jump=vj, kernel="auto")
    else:      # the low-memory route: K7f over batch slices of `slice_step` trajectories
        a = fused._dae_backward_wide_sliced(slice_step, method, de, ae, t, z, v, a0, xs, is_, Gx, gi, tab, zj, vj, None, None)
    runs = [("K7f" if slice_step is None else "K7f in batch slices", a)]
    if slice_step is None:
        # the same backward fed with what the FORWARD saved (dae_integrate(save=True): K7f evaluates nothing forwards); the forward's
        # results must not depend on saving
        xs_s, is_s, saved = fused.dae_integrate(method, de, ae, xi, t, xe, z, v, ie, a0, event_t=ev, z_jump=zj, v_jump=vj, save=True)
        _close(xs_s, xs.double().cpu(), "xs: training forward (saving) vs inference forward")      # (scaled vs plain ELU domain: roundings differ)
        _close(is_s, is_.double().cpu(), "is: training forward (saving) vs inference forward")
        Hp = 32 if H <= 32 else (64 if H <= 64 else 128)
        assert saved[2].shape == (3, Tn, B, Hp)
        runs.append(("K7f(saved)", fused.dae_backward(method, de, ae, t, z, v, a0, xs, is_, Gx, gi, event_idx=tab, z_jump=zj, v_jump=vj,
                                                      kernel="wide", saved=saved)))
    for name, a in runs:
        for key in ("x_init", "z", "v", "z_jump", "v_jump", "all_initial"):
            if b[key] is None:
                assert a[key] is None, key
                continue
            _close(a[key], b[key].double().cpu(), f"{key} ({name} vs K5)")
        for grp in ("de", "ae"):
            for k, (p, q) in enumerate(zip(a[grp], b[grp])):
                _close(p, q.double().cpu(), f"grad {grp} {k} ({name} vs K5)")


@pytest.mark.parametrize("method", ["euler", "midpoint", "rk4"])
@pytest.mark.parametrize("H", [32, 64, 128])
@pytest.mark.parametrize("xd,zd,vd,idim", [(8, 2, 2, 2), (8, 0, 2, 2), (5, 1, 1, 1), (8, 2, 2, 4), (3, 1, 0, 1), (8, 4, 3, 1), (2, 2, 4, 2)])
def test_dae_wide_backward_matches_generic_every_shape_class(xd, zd, vd, idim, H, method):
    """K7f (one-launch sweep with the DE's parameter gradients in the kernel, AE head rows contracted on the host side; `auto` at hidden
    32 / 128, forced at 64) against K5 (generic backward, itself checked against fp64 autograd above) on every (NZM, NZA) register
    class, with two event steps, per-trajectory clocks and a ragged tile."""
    _dae_wide_vs_generic(method, H, 21, 9, xd, zd, vd, idim, seed=140 + xd + zd, events=True)


@pytest.mark.parametrize("H", [20, 48, 100])
@pytest.mark.parametrize("method", ["euler", "rk4"])
def test_wide_backwards_at_zero_padded_hidden_widths(method, H):
    """hidden widths between the kernels' 32 / 64 / 128: forward and backward run zero-padded on the next one up; gradients (sliced
    back to the real width on the host side) against the generic backward K5, ODE and DAE, with events and two time chunks."""
    from py_psnode_amd import fused
    _dae_wide_vs_generic(method, H, 21, 9, 8, 2, 2, 2, seed=340 + H, events=True)
    _dae_wide_vs_generic(method, H, 19, 9, 5, 1, 1, 1, seed=343 + H, events=True, slice_step=16)
    g = torch.Generator().manual_seed(250 + H)
    torch.manual_seed(250 + H)
    B, Tn, xd, zd = 21, 9, 8, 2
    lin = [nn.Linear(a_, b_) for a_, b_ in zip([3 * (xd + zd), H, H, H], [H, H, H, xd])]
    layers = [(m.weight.detach().cuda(), m.bias.detach().cuda()) for m in lin]
    t = (torch.arange(Tn, dtype=torch.float32) * 0.02).view(Tn, 1, 1).repeat(1, B, 1)
    t[:, 1:] = t[:, 1:] * (0.5 + torch.rand(1, B - 1, 1, generator=g))
    r = lambda *s_: (0.1 * torch.randn(*s_, generator=g)).cuda()
    x_in, z = torch.zeros(Tn, B, xd, device="cuda"), r(Tn, B, zd)
    x_in[0] = r(B, xd)
    a0 = torch.cat((x_in[0], z[0]), -1)
    ev = torch.stack([t[1, :, :], t[Tn - 2, :, :]], dim=1).contiguous().cuda()
    zj = r(B, 2, zd)
    tab = fused.event_table(t.cuda(), ev)
    G = torch.randn(Tn, B, xd, generator=g).cuda()
    xs = fused.ode_integrate(method, layers, t.cuda(), x_in, z, a0, event_t=ev, z_jump=zj, kernel="mfma")
    xs_g = fused.ode_integrate(method, layers, t.cuda(), x_in, z, a0, event_t=ev, z_jump=zj, kernel="generic")
    _close(xs, xs_g.double().cpu(), "xs (padded MFMA vs generic)")
    a = fused.ode_backward(method, layers, t.cuda(), z, a0, xs, G, event_idx=tab, z_jump=zj)           # auto: K4f
    b = fused.ode_backward(method, layers, t.cuda(), z, a0, xs, G, event_idx=tab, z_jump=zj, kernel="generic")
    for nme, p_, q_ in zip(["grad x0", "grad z", "grad z_jump", "grad all_initial"], a[:4], b[:4]):
        _close(p_, q_.double().cpu(), nme)
    for k, (p_, q_) in enumerate(zip(a[4], b[4])):
        assert p_.shape == q_.shape
        _close(p_, q_.double().cpu(), f"grad param {k}")


@pytest.mark.parametrize("H", [32, 128])
@pytest.mark.parametrize("B,Tn,chunk", [(1, 2, None), (17, 2, None), (33, 3, 16), (16, 12, None), (37, 12, 16)])
def test_dae_wide_backward_edge_sizes_and_chunks(B, Tn, chunk, H):
    """single trajectory, T = 2, ragged tiles, batch slices of the low-memory route (events, ragged last slice), grad_is = None"""
    _dae_wide_vs_generic("rk4", H, B, Tn, 8, 2, 2, 2, seed=170 + B, events=True, slice_step=chunk)
    _dae_wide_vs_generic("rk4", H, B, Tn, 8, 2, 2, 2, seed=171 + B, events=False, with_gi=False, slice_step=chunk)
    if chunk is not None:      # the same cases in one launch
        _dae_wide_vs_generic("rk4", H, B, Tn, 8, 2, 2, 2, seed=170 + B, events=True)
        _dae_wide_vs_generic("rk4", H, B, Tn, 8, 2, 2, 2, seed=171 + B, events=False, with_gi=False)


@pytest.mark.parametrize("xd,zd,vd,idim", [(4, 0, 1, 3), (4, 1, 1, 2), (3, 2, 0, 2)])
def test_dae_wide_backward_without_grad_is_on_the_four_slot_classes(xd, zd, vd, idim):
    """grad_is = None on the shapes with z + v + i = 4 (found by profiles/scripts/fuzz_backward.py: round 2's split kernel K7w got the AE
    gradients wrong there -- a uniform `if (grad_is)` branch scheduled inside an MFMA's result latency): K7f and
    K7f(saved) against K5."""
    for H, method in ((64, "euler"), (64, "rk4"), (32, "midpoint"), (128, "rk4")):
        _dae_wide_vs_generic(method, H, 24, 5, xd, zd, vd, idim, seed=900 + xd, events=False, with_gi=False)
    # ... and the one-launch kernel K7 (hidden 64) on the same shapes
    from py_psnode_amd import fused
    for method in ("euler", "rk4"):
        de, ae, t, z, v, xi, a0, ev, zj, vj, Gx, Gi = _dae_raw_case(9, 3, xd, zd, vd, idim, 902 + xd, False, H=64)
        xe, ie = torch.zeros(3, 9, 0, device="cuda"), torch.zeros(3, 9, idim, device="cuda")
        xs, is_ = fused.dae_integrate(method, de, ae, xi, t, xe, z, v, ie, a0)
        b = fused.dae_backward(method, de, ae, t, z, v, a0, xs, is_, Gx, None, kernel="generic")
        a = fused.dae_backward(method, de, ae, t, z, v, a0, xs, is_, Gx, None, kernel="mfma")
        for grp in ("de", "ae"):
            for k, (p, q) in enumerate(zip(a[grp], b[grp])):
                _close(p, q.double().cpu(), f"K7 grad_is=None grad {grp} {k}")


@pytest.mark.parametrize("B,Tn", [(1, 2), (3, 1), (17, 2), (33, 3), (16, 5)])
def test_dae_mfma_backward_edge_sizes(B, Tn):
    _dae_both_kernels("rk4", B, Tn, 8, 2, 2, 2, seed=70 + B, events=False)
    _dae_both_kernels("rk4", B, Tn, 8, 2, 2, 2, seed=71 + B, events=False, with_gi=False)


def test_dae_backward_kernel_selection():
    from py_psnode_amd import fused
    de, ae, t, z, v, xi, a0, _, _, _, Gx, Gi = _dae_raw_case(4, 3, 8, 2, 2, 2, 5, False)
    wide = [(torch.zeros(32, 42, device="cuda"), torch.zeros(32, device="cuda")), (torch.zeros(8, 32, device="cuda"), torch.zeros(8, device="cuda"))]
    xs, is_ = torch.zeros(3, 4, 8, device="cuda"), torch.zeros(3, 4, 2, device="cuda")
    with pytest.raises(ValueError):     # PSNODE_ERR_UNSUPPORTED: no MFMA backward for a 2-layer hidden-32 DE
        fused.dae_backward("rk4", wide, ae, t, z, v, a0, xs, is_, Gx, Gi, kernel="mfma")


@pytest.mark.parametrize("method", ["euler", "midpoint", "rk4"])
@pytest.mark.parametrize("B,Tn,events", [(21, 9, True), (16, 5, False), (1, 2, False), (3, 1, False), (37, 12, True),
                                         # round 5: no event -> the FAST / two-role form (>= 10 grid points): one block, a tail, many blocks, ragged tiles
                                         (16, 10, False), (5, 11, False), (37, 14, False), (130, 33, False), (2, 64, False), (19, 150, False)])
def test_latent16_ode_backward_kernel_matches_generic(method, B, Tn, events):
    """K8 (single-wave MFMA backward of the hidden-16 latent ODE, kernel='mfma') against K5 (generic, checked against fp64
    autograd in test_generic_backward_kernel_ode) with events, per-trajectory clocks, ragged tiles and T in {1, 2}; AUTO picks K8."""
    from py_psnode_amd import fused
    H = 16
    g = torch.Generator().manual_seed(300 + B)
    torch.manual_seed(300 + B)
    lin = [nn.Linear(6 * H, H), nn.Linear(H, H)]
    layers = [(m.weight.detach().cuda(), m.bias.detach().cuda()) for m in lin]
    t = (torch.arange(Tn, dtype=torch.float32) * 0.02).view(Tn, 1, 1).repeat(1, B, 1)
    if B > 1:
        t[:, 1:] = t[:, 1:] * (0.5 + torch.rand(1, B - 1, 1, generator=g))
    r = lambda *s: (0.1 * torch.randn(*s, generator=g)).cuda()
    x_in, z = torch.zeros(Tn, B, H, device="cuda"), r(Tn, B, H)
    x_in[0] = r(B, H)
    a0 = torch.cat((x_in[0], z[0]), -1)
    ev = zj = tab = None
    if events:
        ev = torch.stack([t[1, :, :], t[Tn - 2, :, :]], dim=1).contiguous().cuda()
        zj = r(B, 2, H)
        tab = fused.event_table(t.cuda(), ev)
    G = torch.randn(Tn, B, H, generator=g).cuda()
    xs = fused.ode_integrate(method, layers, t.cuda(), x_in, z, a0, event_t=ev, z_jump=zj)
    _close(xs, fused.ode_integrate(method, layers, t.cuda(), x_in, z, a0, event_t=ev, z_jump=zj, kernel="generic").double().cpu(), "forward (K3f vs K0)")
    out = {k: fused.ode_backward(method, layers, t.cuda(), z, a0, xs, G, event_idx=tab, z_jump=zj, kernel=k) for k in ("mfma", "generic", "auto")}
    names = ["grad x0", "grad z", "grad z_jump", "grad all_initial"]
    for nme, a, b, c in zip(names, out["mfma"][:4], out["generic"][:4], out["auto"][:4]):
        if b is None:
            assert a is None
            continue
        _close(a, b.double().cpu(), nme + " (K8 vs K5)")
        assert torch.equal(a, c), nme + ": AUTO must run K8 on this shape"
    for k, (a, b) in enumerate(zip(out["mfma"][4], out["generic"][4])):
        _close(a, b.double().cpu(), f"grad param {k} (K8 vs K5)")


@pytest.mark.parametrize("H", [128, 32])
def test_dae_model_training_at_other_hidden_widths_runs_fused(H):
    """DAE_Model at --hidden 128 (the argparse default) and 32: fused forward (K2) + adjoint sweep K7w with library GEMMs for the
    parameter gradients vs the fp64 autograd walk."""
    from py_psnode_amd import models
    from py_psnode_amd import neural_dae as nd
    B, Tn = 7, 5
    g = torch.Generator().manual_seed(8)
    torch.manual_seed(8)
    m32 = models.DAE_Model(8, 2, 2, 2, H, solver=nd.RK4())
    m64 = models.DAE_Model(8, 2, 2, 2, H, solver=nd.RK4()).double()
    m64.load_state_dict({k: v.double() for k, v in m32.state_dict().items()})
    m64.solver.fused = "off"
    m32 = m32.cuda(); m32.solver.fused = "require"
    t = (torch.arange(Tn, dtype=torch.float32) * 0.01).view(1, Tn, 1).repeat(B, 1, 1)
    r = lambda *s: 0.1 * torch.randn(*s, generator=g)
    x, z, v, i = r(B, Tn, 8), r(B, Tn, 2), r(B, Tn, 2), r(B, Tn, 2)
    ev, zj, vj = t[:, [2], :].contiguous(), r(B, 1, 2), r(B, 1, 2)

    def run(model, cast, dev):
        c = lambda a: cast(a).to(dev)
        xs, is_ = model(t=c(t), x=c(x), z=c(z), v=c(v), i=c(i), event_t=c(ev), z_jump=c(zj), v_jump=c(vj))
        ((xs ** 2).sum() + (is_ ** 2).sum()).backward()
        return xs.detach(), is_.detach(), [p.grad for p in model.parameters()]

    ref = run(m64, lambda a: a.double(), "cpu")
    out = run(m32, lambda a: a, "cuda")
    _close(out[0], ref[0], "xs"); _close(out[1], ref[1], "is")
    for k, (a, b) in enumerate(zip(out[2], ref[2])):
        _close(a, b, f"grad param {k}")


@pytest.mark.parametrize("kind", ["ode", "dae"])
def test_full_size_backward_two_implementations_agree(kind):
    """BASELINE's full size (B = 4096 x 1000 RK4 steps, hidden 64): the one-launch backward (K4 / K7: weight gradients accumulated
    in registers inside the sweep) and the split one (K4w / K7w: stored rows + library GEMMs, 17 time chunks) are independent
    implementations of the same adjoint; at full size every gradient tensor must agree to the usual 2e-4 of its scale, with two
    event steps.  (Size-independent property: d/dx0 of a loss that only sees the last grid point is the product of 1000 step
    Jacobians -- any drift between the two sweeps would compound.)"""
    from py_psnode_amd import fused
    torch.manual_seed(11)
    B, Tn, H = 4096, 1001, 64
    mk = lambda dims: [(l.weight.detach().cuda(), l.bias.detach().cuda()) for l in [nn.Linear(dims[k], dims[k + 1]) for k in range(len(dims) - 1)]]
    r = lambda *s: 0.1 * torch.randn(*s, device="cuda")
    t = (torch.arange(Tn, dtype=torch.float32, device="cuda") * 0.01).view(Tn, 1, 1).repeat(1, B, 1)
    ev = torch.stack([t[300, :, :], t[777, :, :]], dim=1).contiguous()
    tab = fused.event_table(t, ev)
    if kind == "ode":
        xd, zd = 8, 2
        de = mk([3 * (xd + zd), H, H, H, xd])
        x = torch.zeros(Tn, B, xd, device="cuda"); x[0] = r(B, xd)
        z, zj = r(Tn, B, zd), r(B, 2, zd)
        a0 = torch.cat((x[0], z[0]), -1)
        xs = fused.ode_integrate("rk4", de, t, x, z, a0, event_t=ev, z_jump=zj)
        G = torch.zeros(Tn, B, xd, device="cuda"); G[-1] = 1.0; G[500] = torch.randn(B, xd, device="cuda")
        a = fused.ode_backward("rk4", de, t, z, a0, xs, G, event_idx=tab, z_jump=zj, kernel="mfma")
        b = fused.ode_backward("rk4", de, t, z, a0, xs, G, event_idx=tab, z_jump=zj, kernel="wide")
        for nme, p, q in zip(["grad x0", "grad z", "grad z_jump", "grad all_initial"], a[:4], b[:4]):
            _close(p, q.double().cpu(), nme)
        for k, (p, q) in enumerate(zip(a[4], b[4])):
            _close(p, q.double().cpu(), f"grad param {k}")
    else:
        xd, zd, vd, idim = 8, 2, 2, 2
        n = xd + zd + vd + idim
        de, ae = mk([3 * n, H, H, H, xd]), mk([n + xd + zd + vd, H, H, H, idim])
        z, v, xi, i0 = r(Tn, B, zd), r(Tn, B, vd), r(B, xd), r(B, idim)
        zj, vj = r(B, 2, zd), r(B, 2, vd)
        a0 = torch.cat((xi, z[0], v[0], i0), -1)
        xe, ie = torch.zeros(Tn, B, 0, device="cuda"), torch.zeros(Tn, B, idim, device="cuda")
        xs, is_ = fused.dae_integrate("rk4", de, ae, xi, t, xe, z, v, ie, a0, event_t=ev, z_jump=zj, v_jump=vj)
        Gx = torch.zeros(Tn, B, xd, device="cuda"); Gx[-1] = 1.0
        Gi = torch.zeros(Tn, B, idim, device="cuda"); Gi[-1] = 1.0; Gi[400] = torch.randn(B, idim, device="cuda")
        a = fused.dae_backward("rk4", de, ae, t, z, v, a0, xs, is_, Gx, Gi, event_idx=tab, z_jump=zj, v_jump=vj, kernel="mfma")
        b = fused.dae_backward("rk4", de, ae, t, z, v, a0, xs, is_, Gx, Gi, event_idx=tab, z_jump=zj, v_jump=vj, kernel="wide")
        for key in ("x_init", "z", "v", "z_jump", "v_jump", "all_initial"):
            _close(a[key], b[key].double().cpu(), key)
        for grp in ("de", "ae"):
            for k, (p, q) in enumerate(zip(a[grp], b[grp])):
                _close(p, q.double().cpu(), f"grad {grp} {k}")


@pytest.mark.parametrize("events", [False, True])
def test_dae_wide_low_memory_fallback_runs_in_batch_slices(monkeypatch, events):
    """dae_backward_wide (recompute form) falls back from ONE K7f launch to K7f over batch slices when the AE head's rows of the whole grid
    would not fit half of the free HBM (trajectories are independent: parameter gradients add, per-trajectory gradients concatenate).
    Forced here through torch.cuda.mem_get_info; gradients equal the single launch's."""
    from py_psnode_amd import fused
    de, ae, t, z, v, xi, a0, ev, zj, vj, Gx, Gi = _dae_raw_case(37, 9, 8, 2, 2, 2, 4242, events, H=128)
    xe, ie = torch.zeros(9, 37, 0, device="cuda"), torch.zeros(9, 37, 2, device="cuda")
    xs, is_ = fused.dae_integrate("rk4", de, ae, xi, t, xe, z, v, ie, a0, event_t=ev, z_jump=zj, v_jump=vj)
    tab = fused.event_table(t, ev) if ev is not None else None
    kw = dict(event_idx=tab, z_jump=zj, v_jump=vj)
    ref = fused.dae_backward_wide("rk4", de, ae, t, z, v, a0, xs, is_, Gx, Gi, **kw)                   # one launch
    calls = []
    real = fused._lib.load().psnode_dae_backward_wide_f32

    class Spy:                                                                                          # counts the kernel calls of the fallback
        def __call__(self, *a_):
            calls.append(1)
            return real(*a_)
    lib = fused._lib.load()
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda *a_, **k_: (1 << 16, 1 << 40))
    monkeypatch.setattr(lib, "psnode_dae_backward_wide_f32", Spy(), raising=False)
    try:
        got = fused.dae_backward_wide("rk4", de, ae, t, z, v, a0, xs, is_, Gx, Gi, **kw)
    finally:
        monkeypatch.undo()
    assert len(calls) == 3, "37 trajectories in slices of 16"
    for key in ("x_init", "all_initial", "z", "v", "z_jump", "v_jump"):
        if ref[key] is not None:
            _close(got[key], ref[key].double().cpu(), f"fallback {key}")
    for grp in ("de", "ae"):
        for k, (p, q) in enumerate(zip(got[grp], ref[grp])):
            _close(p, q.double().cpu(), f"fallback grad {grp} {k}")


def test_saved_activations_of_another_call_are_refused():
    """The saved rows reach the kernels as raw pointers: a tuple from a call with another grid / batch / method must raise, not be read
    out of bounds."""
    from py_psnode_amd import fused
    torch.manual_seed(5)
    H, xd, zd, B = 64, 8, 2, 20
    lin = [nn.Linear(a_, b_) for a_, b_ in zip([3 * (xd + zd), H, H, H], [H, H, H, xd])]
    layers = [(m.weight.detach().cuda(), m.bias.detach().cuda()) for m in lin]
    mk = lambda Tn: ((torch.arange(Tn, dtype=torch.float32) * 0.02).view(Tn, 1, 1).repeat(1, B, 1).cuda(), (0.1 * torch.randn(Tn, B, xd)).cuda(),
                     (0.1 * torch.randn(Tn, B, zd)).cuda())
    t, x, z = mk(6)
    a0 = torch.cat((x[0], z[0]), -1)
    xs, saved = fused.ode_integrate("rk4", layers, t, x, z, a0, save=True)
    t9, x9, z9 = mk(9)
    xs9, saved9 = fused.ode_integrate("rk4", layers, t9, x9, z9, torch.cat((x9[0], z9[0]), -1), save=True)
    G = torch.randn(6, B, xd).cuda()
    fused.ode_backward("rk4", layers, t, z, a0, xs, G, saved=saved)                                     # the right tuple is fine
    with pytest.raises(ValueError, match="saved activations do not belong"):
        fused.ode_backward("rk4", layers, t, z, a0, xs, G, saved=saved9)                                # other grid
    _, saved_e = fused.ode_integrate("euler", layers, t, x, z, a0, save=True)
    with pytest.raises(ValueError, match="saved activations do not belong"):
        fused.ode_backward("rk4", layers, t, z, a0, xs, G, saved=saved_e)                               # other method (S = 1 vs 4)


