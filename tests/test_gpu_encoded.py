"""K3f: the whole direct_encode ODE forward in one launch (psnode_ode_encoded_integrate_f32, neural_00_ODE_02_direct_encode.py:74-89)
and the DPP latent integrator behind psnode_ode_integrate_f32 at hidden 16 -- against the CPU oracle (encoders/decoders as
plain fp32 nn.functional ops, oracle.integrate_ode for the latent loop) and the golden model forwards G4-ode02."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from helpers import TOL_GPU, T, load, traj_rel_err
from oracle import psnode_oracle as O

pytestmark = pytest.mark.gpu
METHODS = ["euler", "midpoint", "rk4"]


def fused():
    from py_psnode_amd import fused as f
    return f


def _mlp2(din, dout, H=16):
    l1, l2 = nn.Linear(din, H), nn.Linear(H, dout)
    return [(l1.weight.detach(), l1.bias.detach()), (l2.weight.detach(), l2.bias.detach())]


def _apply(ls, a):
    return F.linear(F.elu(F.linear(a, *ls[0])), *ls[1])


def _case(B, Tn, xd, zd, seed, events=True, ragged_clock=True):
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    H = 16
    xe, ze, xdec, de = _mlp2(xd, H), _mlp2(zd, H), _mlp2(H, xd), _mlp2(6 * H, H)
    t = (torch.arange(Tn, dtype=torch.float32) * 0.01).view(1, Tn, 1).repeat(B, 1, 1)
    if ragged_clock and B > 1:
        t[1:] = t[1:] * (0.5 + torch.rand(B - 1, 1, 1, generator=g))
    x, z = 0.3 * torch.randn(B, Tn, xd, generator=g), 0.3 * torch.randn(B, Tn, zd, generator=g)
    ev = zj = None
    if events and Tn > 4:
        ev = t[:, [1, Tn - 2], :].contiguous()
        zj = 0.3 * torch.randn(B, 2, zd, generator=g)
    return xe, ze, xdec, de, t, x, z, ev, zj


def _oracle(method, xe, ze, xdec, de, t, x, z, ev, zj):
    P = lambda a: a.permute(1, 0, 2)
    Xh, Zh = _apply(xe, x), _apply(ze, z)
    a0 = torch.cat((Xh[:, 0], Zh[:, 0]), -1)
    Zhj = _apply(ze, zj) if zj is not None else None
    Xs = O.integrate_ode(method, de, P(t), P(Xh), P(Zh), a0, ev, Zhj)
    return _apply(xdec, Xs).permute(1, 0, 2), _apply(xdec, Xh), Xs


def _dev(ls):
    return [(w.cuda(), b.cuda()) for w, b in ls]


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("B,Tn,xd,zd", [(37, 23, 8, 2), (16, 9, 5, 3), (1, 2, 1, 1), (21, 1, 8, 2), (64, 12, 16, 16), (3, 40, 12, 9)])
def test_encoded_forward_matches_oracle(method, B, Tn, xd, zd):
    xe, ze, xdec, de, t, x, z, ev, zj = _case(B, Tn, xd, zd, seed=B * 100 + Tn)
    ref_pred, ref_re, ref_xs = _oracle(method, xe, ze, xdec, de, t, x, z, ev, zj)
    c = lambda a: None if a is None else a.cuda()
    pred, re, xh = fused().ode_encoded_integrate(method, _dev(xe), _dev(ze), _dev(xdec), _dev(de), c(t), c(x), c(z), event_t=c(ev),
                                                 z_jump=c(zj), want_latent=True)
    assert pred.shape == (B, Tn, xd) and re.shape == (B, Tn, xd) and xh.shape == (Tn, B, 16)
    assert traj_rel_err(pred.cpu(), ref_pred, bdim=0) <= TOL_GPU
    assert traj_rel_err(re.cpu(), ref_re, bdim=0) <= TOL_GPU
    assert traj_rel_err(xh.cpu(), ref_xs, bdim=1) <= TOL_GPU


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("B,Tn,xd,zd", [(37, 23, 8, 2), (16, 16, 5, 3), (1, 2, 1, 1), (19, 17, 8, 2), (64, 33, 16, 16), (3, 150, 12, 9), (130, 48, 8, 2)])
def test_two_role_form_equals_the_one_role_form_and_the_oracle(method, B, Tn, xd, zd):
    """Round 5: a call without events runs K3f in its two-role form (the z encoder, the decoder and the reconstruction on the partner wave's MFMA
    tiles, handed over through LDS rings, one barrier per 16 steps); asking for the latent trajectory (want_latent) keeps the one-role path.
    Both against each other and against the oracle -- grids that end inside a 16-row block, one block, more blocks than the ring holds,
    ragged trajectory tiles, every input width."""
    xe, ze, xdec, de, t, x, z, _, _ = _case(B, Tn, xd, zd, seed=B * 7 + Tn, events=False)
    want = _oracle(method, xe, ze, xdec, de, t, x, z, None, None)
    args = (method, _dev(xe), _dev(ze), _dev(xdec), _dev(de), t.cuda(), x.cuda(), z.cuda())
    p2, r2, _ = fused().ode_encoded_integrate(*args)
    p1, r1, xh1 = fused().ode_encoded_integrate(*args, want_latent=True)
    assert xh1 is not None
    for got, ref, what in ((p2, want[0], "x_pred two-role"), (r2, want[1], "x_re two-role"), (p1, want[0], "x_pred one-role"), (r1, want[1], "x_re one-role")):
        assert traj_rel_err(got.cpu(), ref, bdim=0) <= TOL_GPU, what
    assert traj_rel_err(p2.cpu(), p1.cpu(), bdim=0) <= 2e-6 and traj_rel_err(r2.cpu(), r1.cpu(), bdim=0) <= 2e-6
    pn, rn, _ = fused().ode_encoded_integrate(*args, want_recon=False)      # no reconstruction wanted: the partner still encodes z and decodes
    assert rn is None and torch.equal(pn, p2)


def test_encoded_forward_strided_inputs_and_no_recon():
    """Inputs as non-contiguous slices of wider tensors (element strides travel through the C ABI); reconstruction skipped."""
    B, Tn, xd, zd = 19, 15, 8, 2
    xe, ze, xdec, de, t, x, z, ev, zj = _case(B, Tn, xd, zd, seed=5)
    ref_pred, _, _ = _oracle("rk4", xe, ze, xdec, de, t, x, z, ev, zj)
    big = torch.zeros(B, Tn + 3, xd + zd + 5).cuda()
    big[:, 1:Tn + 1, 2:2 + xd] = x.cuda()
    big[:, 1:Tn + 1, 2 + xd:2 + xd + zd] = z.cuda()
    xv, zv = big[:, 1:Tn + 1, 2:2 + xd], big[:, 1:Tn + 1, 2 + xd:2 + xd + zd]
    assert not xv.is_contiguous()
    pred, re, xh = fused().ode_encoded_integrate("rk4", _dev(xe), _dev(ze), _dev(xdec), _dev(de), t.cuda(), xv, zv, event_t=ev.cuda(),
                                                 z_jump=zj.cuda(), want_recon=False)
    assert re is None and xh is None
    assert traj_rel_err(pred.cpu(), ref_pred, bdim=0) <= TOL_GPU


@pytest.mark.parametrize("method", METHODS)
def test_model_forward_takes_the_single_launch_route(method):
    """models.ODE_Model(direct_encode) under no_grad: one fused launch, equal (to tolerance) to the row-kernel + solver route that
    training uses, and to the reference's golden forward G4-ode02."""
    from py_psnode_amd import models
    from py_psnode_amd import neural_dae as nd
    d = load("g4_model_ode02.npz")
    m = models.ODE_Model(8, 2, 16, direct_encode=True)
    m.load_state_dict({k[4:].replace("__", "."): T(v) for k, v in d.items() if k.startswith("sd__")})
    m = m.cuda()
    m.solver = {"euler": nd.Euler, "midpoint": nd.Midpoint, "rk4": nd.RK4}[method]()
    m.solver.fused = "require"
    g = lambda k: T(d[k]).cuda()
    calls = []
    orig = fused().ode_encoded_integrate
    try:
        fused().ode_encoded_integrate = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        with torch.no_grad():
            pred, re = m(t=g("t"), x=g("x"), z=g("z"), event_t=g("event_t"), z_jump=g("z_jump"))
    finally:
        fused().ode_encoded_integrate = orig
    assert calls == [1], "the no-grad direct_encode forward must be ONE fused launch"
    assert traj_rel_err(pred.cpu(), d[f"{method}_out0"], bdim=0) <= TOL_GPU
    assert traj_rel_err(re.cpu(), d[f"{method}_out1"], bdim=0) <= TOL_GPU
    # the training route (autograd on): row kernels + latent integrator + row kernels -- same numbers to tolerance
    pred2, re2 = m(t=g("t"), x=g("x"), z=g("z"), event_t=g("event_t"), z_jump=g("z_jump"))
    assert pred2.requires_grad
    assert traj_rel_err(pred2.detach().cpu(), pred.cpu(), bdim=0) <= TOL_GPU and traj_rel_err(re2.detach().cpu(), re.cpu(), bdim=0) <= TOL_GPU


def test_single_launch_route_respects_hooks_and_the_generic_kernel_request():
    """Round-2 ADVICE: the one-launch route must step aside when a user hook sits on a module it would swallow (hooks have to fire)
    and when solver.kernel asks for the generic kernel."""
    from py_psnode_amd import models
    from py_psnode_amd import neural_dae as nd
    d = load("g4_model_ode02.npz")
    m = models.ODE_Model(8, 2, 16, direct_encode=True)
    m.load_state_dict({k[4:].replace("__", "."): T(v) for k, v in d.items() if k.startswith("sd__")})
    m = m.cuda()
    m.solver = nd.RK4()
    g = lambda k: T(d[k]).cuda()
    calls, fired = [], []
    orig = fused().ode_encoded_integrate
    try:
        fused().ode_encoded_integrate = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        h = m.x_encoder.register_forward_hook(lambda mod, i, o: fired.append(1))
        with torch.no_grad():
            pred_h, _ = m(t=g("t"), x=g("x"), z=g("z"), event_t=g("event_t"), z_jump=g("z_jump"))
        assert calls == [] and fired, "a forward hook on x_encoder must fire: no single-launch route"
        h.remove()
        m.solver.kernel = "generic"
        with torch.no_grad():
            pred_g, _ = m(t=g("t"), x=g("x"), z=g("z"), event_t=g("event_t"), z_jump=g("z_jump"))
        assert calls == [], "solver.kernel = 'generic' must not run the DPP kernel"
        m.solver.kernel = "auto"
        with torch.no_grad():
            pred, _ = m(t=g("t"), x=g("x"), z=g("z"), event_t=g("event_t"), z_jump=g("z_jump"))
        assert calls == [1]
    finally:
        fused().ode_encoded_integrate = orig
    for other in (pred_h, pred_g):
        assert traj_rel_err(other.cpu(), pred.cpu(), bdim=0) <= TOL_GPU


def test_encoded_forward_full_size_matches_unfused_route():
    """BASELINE config 3 size (B=4096, T=1001): the single launch against the round-1 route (row kernels + latent integrator) on the
    GPU, and 32 of the trajectories against the CPU oracle."""
    B, Tn, xd, zd = 4096, 1001, 8, 2
    xe, ze, xdec, de, t, x, z, ev, zj = _case(B, Tn, xd, zd, seed=77, ragged_clock=False)
    f = fused()
    c = lambda a: a.cuda()
    pred, re, xh = f.ode_encoded_integrate("rk4", _dev(xe), _dev(ze), _dev(xdec), _dev(de), c(t), c(x), c(z), event_t=c(ev), z_jump=c(zj),
                                           want_latent=True)
    Xh, Zh = f.mlp_rows(_dev(xe), c(x)), f.mlp_rows(_dev(ze), c(z))
    a0 = torch.cat((Xh[:, 0], Zh[:, 0]), -1)
    Xs = f.ode_integrate("rk4", _dev(de), c(t).permute(1, 0, 2), Xh.permute(1, 0, 2), Zh.permute(1, 0, 2), a0, event_t=c(ev),
                         z_jump=f.mlp_rows(_dev(ze), c(zj)))
    assert traj_rel_err(xh.cpu(), Xs.cpu(), bdim=1) <= TOL_GPU
    assert traj_rel_err(pred.cpu(), f.mlp_rows(_dev(xdec), Xs).permute(1, 0, 2).cpu(), bdim=0) <= TOL_GPU
    assert traj_rel_err(re.cpu(), f.mlp_rows(_dev(xdec), Xh).cpu(), bdim=0) <= TOL_GPU
    sel = torch.arange(0, B, B // 32)
    ref_pred, ref_re, _ = _oracle("rk4", xe, ze, xdec, de, t[sel], x[sel], z[sel], ev[sel], zj[sel])
    # events are decided by GLOBAL trajectory 0, which is in `sel`
    assert traj_rel_err(pred.cpu()[sel], ref_pred, bdim=0) <= TOL_GPU and traj_rel_err(re.cpu()[sel], ref_re, bdim=0) <= TOL_GPU


@pytest.mark.parametrize("method", ["rk4", "euler"])
def test_two_role_full_size_subset_vs_oracle(method):
    """BASELINE config 3 AS BENCHMARKED: B=4096, T=1001, no event, no latent trajectory wanted -- the call
    bench.py's `ode02` line times, which K3f runs in its TWO-ROLE form (psnode_latent_dpp.hip: enc_two_role; the full-size test above
    passes events and want_latent and therefore takes the one-role path).  32 of the trajectories against the CPU oracle, the whole
    batch against the one-role path; that the two-role form is what ran is read off the bits: its decoder / z encoder run on MFMA
    tiles, the one-role form's on VALU + DPP, so the two agree to rounding and NOT bit for bit."""
    B, Tn, xd, zd = 4096, 1001, 8, 2
    xe, ze, xdec, de, t, x, z, _, _ = _case(B, Tn, xd, zd, seed=78, events=False, ragged_clock=False)
    f = fused()
    args = (method, _dev(xe), _dev(ze), _dev(xdec), _dev(de), t.cuda(), x.cuda(), z.cuda())
    # "no events" as the scripts and bench.py encode it: an event list whose times are never on the grid
    no_ev = dict(event_t=torch.full((B, 2, 1), -1.0).cuda(), z_jump=torch.zeros(B, 2, zd).cuda())
    p2, r2, xh2 = f.ode_encoded_integrate(*args, **no_ev)
    assert xh2 is None
    p1, r1, xh1 = f.ode_encoded_integrate(*args, **no_ev, want_latent=True)
    assert xh1 is not None
    assert not torch.equal(p2, p1), "the plain call must take the two-role form (bit-equal to the one-role path: it did not)"
    assert traj_rel_err(p2.cpu(), p1.cpu(), bdim=0) <= 2e-6 and traj_rel_err(r2.cpu(), r1.cpu(), bdim=0) <= 2e-6
    assert bool(torch.isfinite(p2).all()) and bool(torch.isfinite(r2).all())
    sel = torch.arange(0, B, B // 32)
    ref_pred, ref_re, _ = _oracle(method, xe, ze, xdec, de, t[sel], x[sel], z[sel], None, None)
    assert traj_rel_err(p2.cpu()[sel], ref_pred, bdim=0) <= TOL_GPU and traj_rel_err(r2.cpu()[sel], ref_re, bdim=0) <= TOL_GPU


def test_encoded_unsupported_shape_raises():
    from py_psnode_amd import _lib
    xe, ze, xdec, de, t, x, z, ev, zj = _case(4, 5, 8, 2, seed=1, events=False)
    bad = _mlp2(6 * 16, 16, H=32)
    assert not fused().ode_encoded_supported(_dev(xe), _dev(ze), _dev(xdec), _dev(bad))
    with pytest.raises(_lib.UnsupportedShapeError):
        fused().ode_encoded_integrate("rk4", _dev(xe), _dev(ze), _dev(xdec), _dev(bad), t.cuda(), x.cuda(), z.cuda())


class _ScriptStyleOde02(nn.Module):
    """The shape of the reference's ODE_Model (neural_00_ODE_02_direct_encode.py:60-89) restated for the GPU box, where the
    reference's files do not exist: plain nn.Sequential encoders/decoder called as modules, the script's forward order."""

    def __init__(self, xd, zd, H, nd_):
        super().__init__()
        from py_psnode_amd import models
        self.x_encoder = nn.Sequential(nn.Linear(xd, H), nn.ELU(), nn.Linear(H, H))
        self.x_decoder = nn.Sequential(nn.Linear(H, H), nn.ELU(), nn.Linear(H, xd))
        self.z_encoder = nn.Sequential(nn.Linear(zd, H), nn.ELU(), nn.Linear(H, H))
        self.de_func = models.DE_Func(2 * H, (H,), H)
        self.solver = nd_.Euler()
        self.event = nd_.ODE_Event()

    def forward(self, t, x, z, event_t, z_jump):
        Xh = self.x_encoder(x).permute(1, 0, 2)
        Zh = self.z_encoder(z).permute(1, 0, 2)
        all_initial = torch.cat((Xh[0], Zh[0]), dim=-1)
        self.event.set_event(t=event_t, z=self.z_encoder(z_jump))
        sol = self.solver.integrate_ODE(x_func=self.de_func, t=t.permute(1, 0, 2), x=Xh, z=Zh, all_initial=all_initial,
                                        event_fn=self.event.event_fn, jump_change_fn=self.event.jump_change_fn)
        return self.x_decoder(sol).permute(1, 0, 2), self.x_decoder(Xh).permute(1, 0, 2)


@pytest.mark.parametrize("method", ["euler", "rk4"])
def test_accelerate_puts_script_style_encoders_on_the_row_kernels(method):
    """accelerate(model) on a model written like the reference's script class: forward equals golden G4-ode02, gradients equal the
    reference's (G7-ode02), and the row kernels really take the encoder / decoder calls (forward and backward)."""
    import py_psnode_amd
    from py_psnode_amd import neural_dae as nd
    from py_psnode_amd.models import RowsSequential
    cls = {"euler": nd.Euler, "rk4": nd.RK4}[method]
    d, dg = load("g4_model_ode02.npz"), load("g7_grad_ode02.npz")
    f = fused()
    calls = {"fwd": 0, "bwd": 0}
    of, ob = f.mlp_rows, f.mlp_rows_backward

    def run(dd, grad):
        m = _ScriptStyleOde02(8, 2, 16, nd)
        m.load_state_dict({k[4:].replace("__", "."): T(v) for k, v in dd.items() if k.startswith("sd__")})
        m = py_psnode_amd.accelerate(m.cuda())
        assert all(isinstance(getattr(m, n), RowsSequential) for n in ("x_encoder", "x_decoder", "z_encoder"))
        m.solver = cls()
        m.solver.fused = "require"
        g = lambda k: T(dd[k]).cuda()
        if not grad:
            with torch.no_grad():
                return m, m(t=g("t"), x=g("x"), z=g("z"), event_t=g("event_t"), z_jump=g("z_jump"))
        out = m(t=g("t"), x=g("x"), z=g("z"), event_t=g("event_t"), z_jump=g("z_jump"))
        sum((o * g(f"G{k}")).sum() for k, o in enumerate(out)).backward()
        return m, out

    try:
        # (both names: models.py calls fused.mlp_rows through the package, the autograd bridge in fused/rows.py its module-level names)
        f.mlp_rows = f.rows.mlp_rows = lambda *a, **k: (calls.__setitem__("fwd", calls["fwd"] + 1), of(*a, **k))[1]
        f.mlp_rows_backward = f.rows.mlp_rows_backward = lambda *a, **k: (calls.__setitem__("bwd", calls["bwd"] + 1), ob(*a, **k))[1]
        m, out = run(d, grad=False)
        assert calls["fwd"] == 5                        # enc x, enc z, enc z_jump, dec solution, dec reconstruction
        for k, o in enumerate(out):
            assert traj_rel_err(o.cpu(), d[f"{method}_out{k}"], bdim=0) <= TOL_GPU
        m, out = run(dg, grad=True)
        assert calls["bwd"] == 5
    finally:
        f.mlp_rows = f.rows.mlp_rows = of
        f.mlp_rows_backward = f.rows.mlp_rows_backward = ob
    for name, p in m.named_parameters():
        ref = torch.as_tensor(dg[f"{method}_gp__" + name.replace(".", "__")], dtype=torch.float64)
        err = float((p.grad.double().cpu() - ref).abs().max())
        assert err <= 2e-4 * max(float(ref.abs().max()), 1e-6), (name, err)


@pytest.mark.parametrize("method", ["euler", "rk4"])
def test_enc_hidden_extension_matches_the_oracle_and_trains_fused(method):
    """models.ODE_Model(..., enc_hidden=64): BASELINE's "enc/dec 64 -> 16 latent" reading of the direct_encode config (an extension:
    upstream has one hidden_dim).  Encoders / decoder of hidden 64 on the MFMA row kernels around the hidden-16 latent integrator:
    forward against the oracle, gradients against the same model walked on the CPU."""
    from py_psnode_amd import models
    from py_psnode_amd import neural_dae as nd
    B, Tn, xd, zd = 21, 12, 8, 2
    g = torch.Generator().manual_seed(77)
    torch.manual_seed(77)
    cls = {"euler": nd.Euler, "rk4": nd.RK4}[method]
    m_cpu = models.ODE_Model(xd, zd, 16, direct_encode=True, solver=cls(), enc_hidden=64)
    assert m_cpu.x_encoder[0].out_features == 64 and m_cpu.x_encoder[2].out_features == 16 and m_cpu.x_decoder[0].in_features == 16
    m_gpu = models.ODE_Model(xd, zd, 16, direct_encode=True, solver=cls(), enc_hidden=64)
    m_gpu.load_state_dict(m_cpu.state_dict())
    m_gpu = m_gpu.cuda()
    m_cpu.solver.fused = "off"
    m_gpu.solver.fused = "require"
    t = (torch.arange(Tn, dtype=torch.float32) * 0.01).view(1, Tn, 1).repeat(B, 1, 1)
    x, z = 0.3 * torch.randn(B, Tn, xd, generator=g), 0.3 * torch.randn(B, Tn, zd, generator=g)
    ev, zj = t[:, [1, Tn - 2], :].contiguous(), 0.3 * torch.randn(B, 2, zd, generator=g)
    seq = lambda s: [(s[0].weight.detach(), s[0].bias.detach()), (s[2].weight.detach(), s[2].bias.detach())]
    de = [(l.weight.detach(), l.bias.detach()) for l in m_cpu.de_func.x_dot if isinstance(l, nn.Linear)]
    ref_pred, ref_re, _ = _oracle(method, seq(m_cpu.x_encoder), seq(m_cpu.z_encoder), seq(m_cpu.x_decoder), de, t, x, z, ev, zj)
    c = lambda a: a.cuda()
    with torch.no_grad():
        pred, re = m_gpu(t=c(t), x=c(x), z=c(z), event_t=c(ev), z_jump=c(zj))
    assert traj_rel_err(pred.cpu().permute(1, 0, 2), ref_pred.permute(1, 0, 2)) <= TOL_GPU
    assert traj_rel_err(re.cpu().permute(1, 0, 2), ref_re.permute(1, 0, 2)) <= TOL_GPU
    G1, G2 = torch.randn(B, Tn, xd, generator=g), torch.randn(B, Tn, xd, generator=g)
    outs = m_cpu(t=t, x=x, z=z, event_t=ev, z_jump=zj)
    ((outs[0] * G1).sum() + (outs[1] * G2).sum()).backward()
    outs = m_gpu(t=c(t), x=c(x), z=c(z), event_t=c(ev), z_jump=c(zj))
    ((outs[0] * c(G1)).sum() + (outs[1] * c(G2)).sum()).backward()
    for (name, pc), pg in zip(m_cpu.named_parameters(), m_gpu.parameters()):
        scale = float(pc.grad.abs().max())
        err = float((pg.grad.cpu() - pc.grad).abs().max())
        assert err <= 5e-4 * max(scale, 1e-6), f"{name}: err {err:.3e} vs scale {scale:.3e}"
