"""K4x (csrc/psnode_backward_x.hip, round 6): the exchange-free backward of the ODE integrator -- one wave = 4 trajectories, every operand
in K1x's layouts, the rows the training forward saved -- against K4f (the 4-wave tile kernel, an independent implementation of the same
adjoint: `kernel="wide"`), against the fp64 autograd walk of the reference's loop (my_solvers.py:66-78 under loss.backward(),
neural_00_ODE_01_no_encode.py:358-360), and at BASELINE's full size."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
METHODS = ["euler", "midpoint", "rk4"]


def _close(a, b, what, tol=1e-4):
    a, b = a.double().cpu(), b.double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = float(b.abs().max()) if b.numel() else 0.0
    err = float((a - b).abs().max()) if b.numel() else 0.0
    assert err <= tol * max(scale, 1e-6), f"{what}: err {err:.3e} vs scale {scale:.3e}"


def _case(B, Tn, xd, zd, H, seed, events, ragged=True):
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    lin = [nn.Linear(a_, b_) for a_, b_ in zip([3 * (xd + zd), H, H, H], [H, H, H, xd])]
    t = (torch.arange(Tn, dtype=torch.float32) * 0.02).view(Tn, 1, 1).repeat(1, B, 1)
    if ragged and B > 1:
        t[:, 1:] = t[:, 1:] * (0.5 + torch.rand(1, B - 1, 1, generator=g))
    r = lambda *s_: 0.1 * torch.randn(*s_, generator=g)
    x, z = r(Tn, B, xd), r(Tn, B, zd)
    ev = zj = None
    if events and zd and Tn > 4:
        ev = torch.stack([t[1, :, :], t[Tn - 2, :, :]], dim=1).contiguous()
        zj = r(B, 2, zd)
    G = torch.randn(Tn, B, xd, generator=g)
    return lin, t, x, z, ev, zj, G


def _both(method, lin, t, x, z, ev, zj, G, need_z=True):
    from py_psnode_amd import fused
    c = lambda a: None if a is None else a.cuda()
    layers = [(m.weight.detach().cuda(), m.bias.detach().cuda()) for m in lin]
    a0 = torch.cat((x[0], z[0]), -1).cuda()
    tab = fused.event_table(c(t), c(ev)) if ev is not None else None
    xs, saved = fused.ode_integrate(method, layers, c(t), c(x), c(z), a0, event_t=c(ev), z_jump=c(zj), save=True)
    kw = dict(event_idx=tab, z_jump=c(zj), saved=saved, need_grad_z=need_z)
    wave = fused.ode_backward(method, layers, c(t), c(z), a0, xs, c(G), kernel="wave", **kw)
    tile = fused.ode_backward(method, layers, c(t), c(z), a0, xs, c(G), kernel="wide", **kw)
    auto = fused.ode_backward(method, layers, c(t), c(z), a0, xs, c(G), **kw)
    return wave, tile, auto


def _compare(wave, tile, tol=1e-4):
    for nme, p, q in zip(["grad x0", "grad z", "grad z_jump", "grad all_initial"], wave[:4], tile[:4]):
        assert (p is None) == (q is None), nme
        if p is not None:
            _close(p, q, nme, tol)
    for k, (p, q) in enumerate(zip(wave[4], tile[4])):
        _close(p, q, f"grad param {k}", tol)


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("xd,zd,H,B,Tn,events", [(8, 2, 64, 37, 12, True), (8, 2, 64, 16, 5, False), (5, 3, 48, 21, 9, True), (3, 0, 64, 7, 4, False),
                                                 (8, 8, 33, 130, 7, True), (1, 1, 64, 1, 2, False), (7, 8, 50, 3, 3, False), (8, 4, 64, 4, 6, True),
                                                 (2, 7, 40, 9, 13, True), (8, 2, 64, 5, 70, True)])
def test_wave_backward_equals_the_tile_backward(method, xd, zd, H, B, Tn, events):
    """Every output of K4x against K4f from the SAME saved rows: every x / z width incl. odd ones and z_dim 0, padded hidden widths, ragged
    last wave (B % 4 != 0), B = 1, T = 2, per-trajectory clocks, two events (the second on the step before the last), with and without
    dL/dz wanted.  AUTO = K4x here (<= 4608 trajectories): bit-equal to the forced kernel."""
    case = _case(B, Tn, xd, zd, H, seed=100 * H + 10 * xd + zd + B, events=events)
    wave, tile, auto = _both(method, *case)
    _compare(wave, tile)
    for p, q in zip(wave[4], auto[4]):
        assert torch.equal(p, q), "AUTO must run K4x for this call"
    if zd:
        wave_nz, tile_nz, _ = _both(method, *case, need_z=False)
        assert wave_nz[1] is None
        for p, q in zip(wave_nz[4], wave[4]):
            assert torch.equal(p, q), "the parameter gradients do not depend on whether dL/dz is wanted"
        _close(wave_nz[0], tile_nz[0], "grad x0 (no dL/dz)")


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("xd,zd,H", [(8, 2, 64), (5, 3, 40), (3, 0, 64)])
def test_wave_backward_matches_fp64_autograd(method, xd, zd, H):
    """K4x against the fp64 autograd walk of this package's restated loop (== the reference's loop; golden sets G7 / G8 pin that walk to the
    reference's own loss.backward()): 2e-4 of each tensor's max, as the other backward kernels."""
    from py_psnode_amd import models
    from py_psnode_amd import neural_dae as nd
    B, Tn = 22, 11
    lin, t, x, z, ev, zj, G = _case(B, Tn, xd, zd, H, seed=7 + H + xd, events=True)
    de = models.DE_Func(xd + zd, (H, H, H), xd).double()
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
    wave, _, _ = _both(method, lin, t, x, z, ev, zj, G)
    gx0, gz, gzj, ga0, gp = wave
    _close(gx0 + ga0[:, :xd], xq.grad[0], "grad x0", 2e-4)
    if zd:
        gz_tot = gz.clone(); gz_tot[0] += ga0[:, xd:]
        _close(gz_tot, zq.grad, "grad z", 2e-4)
        if zjq is not None:
            _close(gzj, zjq.grad, "grad z_jump", 2e-4)
    for k, (a_, p_) in enumerate(zip(gp, de.x_dot.parameters())):
        _close(a_, p_.grad, f"grad param {k}", 2e-4)


def test_wave_backward_full_size_equals_the_tile_backward():
    """BASELINE config 2's training step size (B = 4096 x 1000 RK4 steps, hidden 64, two events): K4x and K4f from the same saved rows --
    d/dx0 of a loss that sees the last grid point is a product of 1000 step Jacobians; any drift between the sweeps would compound."""
    from py_psnode_amd import fused
    torch.manual_seed(12)
    B, Tn, H, xd, zd = 4096, 1001, 64, 8, 2
    de = [(l.weight.detach().cuda(), l.bias.detach().cuda()) for l in [nn.Linear(a_, b_) for a_, b_ in zip([3 * (xd + zd), H, H, H], [H, H, H, xd])]]
    r = lambda *s: 0.1 * torch.randn(*s, device="cuda")
    t = (torch.arange(Tn, dtype=torch.float32, device="cuda") * 0.01).view(Tn, 1, 1).repeat(1, B, 1)
    ev = torch.stack([t[300, :, :], t[777, :, :]], dim=1).contiguous()
    tab = fused.event_table(t, ev)
    x = torch.zeros(Tn, B, xd, device="cuda"); x[0] = r(B, xd)
    z, zj = r(Tn, B, zd), r(B, 2, zd)
    a0 = torch.cat((x[0], z[0]), -1)
    xs, saved = fused.ode_integrate("rk4", de, t, x, z, a0, event_t=ev, z_jump=zj, save=True)
    G = torch.zeros(Tn, B, xd, device="cuda"); G[-1] = 1.0; G[500] = torch.randn(B, xd, device="cuda")
    kw = dict(event_idx=tab, z_jump=zj, saved=saved)
    wave = fused.ode_backward("rk4", de, t, z, a0, xs, G, kernel="wave", **kw)
    tile = fused.ode_backward("rk4", de, t, z, a0, xs, G, kernel="wide", **kw)
    _compare(wave, tile, tol=2e-4)
    again = fused.ode_backward("rk4", de, t, z, a0, xs, G, kernel="wave", **kw)
    for p, q in zip(wave[4], again[4]):
        assert torch.equal(p, q), "K4x is deterministic (per-wave partials summed in a fixed order)"


def test_wave_backward_is_refused_without_saved_rows_and_outside_its_class():
    from py_psnode_amd import _lib, fused
    lin, t, x, z, ev, zj, G = _case(8, 6, 8, 2, 64, seed=3, events=False)
    c = lambda a: a.cuda()
    layers = [(m.weight.detach().cuda(), m.bias.detach().cuda()) for m in lin]
    a0 = torch.cat((x[0], z[0]), -1).cuda()
    xs = fused.ode_integrate("rk4", layers, c(t), c(x), c(z), a0)
    with pytest.raises(_lib.UnsupportedShapeError):
        fused.ode_backward("rk4", layers, c(t), c(z), a0, xs, c(G), kernel="wave")          # no saved rows: K4x has no recompute form
    assert fused.ode_backward_supported("rk4", layers, 8, 2, "wave")
    lin32 = _case(8, 6, 8, 2, 32, seed=3, events=False)[0]
    assert not fused.ode_backward_supported("rk4", [(m.weight.detach().cuda(), m.bias.detach().cuda()) for m in lin32], 8, 2, "wave")


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("H,xd,zd,vd,idim,B,Tn,events", [(64, 8, 2, 2, 2, 4096, 4, False), (64, 8, 2, 2, 2, 37, 9, True), (40, 5, 1, 1, 1, 18, 6, True),
                                                         (32, 3, 1, 0, 1, 7, 4, False), (64, 8, 2, 2, 4, 21, 7, True), (20, 2, 2, 4, 2, 130, 3, False),
                                                         (64, 8, 0, 2, 2, 5, 12, True), (64, 1, 1, 1, 1, 1, 2, False)])
def test_dae_training_forward_saves_the_same_rows_on_both_mfma_integrators(method, H, xd, zd, vd, idim, B, Tn, events):
    """Round 6: K2x has SAVE instances (`kernel="wave"`; AUTO keeps K2 for the saving forward, which measures faster).  Both integrators write the
    same five tensors (DE rows, stage inputs, the AE head's rows per grid point and per event, the event's i0 in slot layout) to rounding,
    incl. the zero padding, odd x_dim, z_dim == 0, two events and a ragged last wave; K7f (+ K7h) returns the same gradients from either,
    whichever wrote the rows."""
    from py_psnode_amd import fused
    g = torch.Generator().manual_seed(H * 11 + xd + B + idim)
    torch.manual_seed(H * 11 + xd + B + idim)
    n = xd + zd + vd + idim
    mk = lambda dims: [(l.weight.detach().cuda(), l.bias.detach().cuda()) for l in [nn.Linear(dims[k], dims[k + 1]) for k in range(len(dims) - 1)]]
    de, ae = mk([3 * n, H, H, H, xd]), mk([n + xd + zd + vd, H, H, H, idim])
    r = lambda *s_: (0.1 * torch.randn(*s_, generator=g)).cuda()
    t = (torch.arange(Tn, dtype=torch.float32) * 0.02).view(Tn, 1, 1).repeat(1, B, 1).cuda()
    z, v, xi, i0 = r(Tn, B, zd), r(Tn, B, vd), r(B, xd), r(B, idim)
    a0 = torch.cat((xi, z[0], v[0], i0), -1)
    ev = zj = vj = None
    if events and Tn > 3:
        ev = torch.stack([t[1, :, :], t[Tn - 2, :, :]], dim=1).contiguous()
        zj, vj = r(B, 2, zd), r(B, 2, vd)
    Gx, Gi = torch.randn(Tn, B, xd, generator=g).cuda(), torch.randn(Tn, B, idim, generator=g).cuda()
    xe, ie = torch.zeros(Tn, B, 0, device="cuda"), torch.zeros(Tn, B, idim, device="cuda")
    tab = fused.event_table(t, ev) if ev is not None else None
    out = {}
    for kern in ("tile", "wave", "auto"):
        xs, is_, saved = fused.dae_integrate(method, de, ae, xi, t, xe, z, v, ie, a0, event_t=ev, z_jump=zj, v_jump=vj, save=True, kernel=kern)
        grads = fused.dae_backward(method, de, ae, t, z, v, a0, xs, is_, Gx, Gi, event_idx=tab, z_jump=zj, v_jump=vj, saved=saved)
        out[kern] = (xs, is_, saved, grads)
    (xs_t, is_t, sv_t, g_t), (xs_w, is_w, sv_w, g_w) = out["tile"], out["wave"]
    _close(xs_w, xs_t, "xs", 1e-5); _close(is_w, is_t, "is", 1e-5)
    names = ["DE rows", "stage inputs", "AE head rows", "event AE rows", "event i0 (slot layout)"]
    for nme, p, q in zip(names, sv_w, sv_t):
        assert (p is None) == (q is None), nme
        if p is not None:
            assert p.shape == q.shape, nme
            if nme.startswith("event i0"):      # K2 fills every slot, K2x the algebraic ones (what K7f reads): compare those
                nzv, ne = zd + vd, zd + vd + idim
                cols = [nzv + d for d in range(idim)] + [ne + nzv + d for d in range(idim)]
                _close(p[..., cols], q[..., cols], nme, 1e-5)
            else:
                _close(p, q, nme, 1e-5)
    hp = sv_t[0].shape[-1]
    if H < hp and Tn > 1:
        assert float(sv_w[0][..., H:].abs().max()) == 0.0 and float(sv_w[2][..., H:].abs().max()) == 0.0, "padding units are stored as zeros"
    for key in ("x_init", "z", "v", "z_jump", "v_jump", "all_initial"):
        assert (g_w[key] is None) == (g_t[key] is None), key
        if g_w[key] is not None:
            _close(g_w[key], g_t[key], key)
    for grp in ("de", "ae"):
        for k, (p, q) in enumerate(zip(g_w[grp], g_t[grp])):
            _close(p, q, f"grad {grp} {k}")
    # AUTO keeps K2 for the SAVING forward (K2x's is forced-only: measured slower, psnode_mfma_xd.hip: mfma_x_dae_preferred)
    assert torch.equal(out["auto"][0], xs_t) and torch.equal(out["auto"][2][2], sv_t[2]), "AUTO must run K2's saving instance"
