"""K3g: the whole direct_encode DAE forward at hidden_dim 64 in one launch (psnode_dae_encoded_integrate_f32,
neural_01_DAE_02_direct_encode.py:125-153) -- against the CPU oracle (encoders / decoders as plain fp32 nn.functional ops around
oracle.integrate_dae), the reference's own golden forward at hidden 64 (g7_grad_dae02_h64.npz: out0..out3 of the reference's
DAE_Model) and the row-kernel + K3c route it replaces."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from helpers import TOL_GPU, T, load, traj_rel_err
from oracle import psnode_oracle as O

pytestmark = pytest.mark.gpu
METHODS = ["euler", "midpoint", "rk4"]
H = 64


def fused():
    from py_psnode_amd import fused as f
    return f


def _mlp2(din, dout):
    l1, l2 = nn.Linear(din, H), nn.Linear(H, dout)
    return [(l1.weight.detach(), l1.bias.detach()), (l2.weight.detach(), l2.bias.detach())]


def _apply(ls, a):
    return F.linear(F.elu(F.linear(a, *ls[0])), *ls[1])


def _case(B, Tn, xd, zd, vd, idim, seed, events=True, ragged_clock=True):
    g = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    nblk = 4 if zd else 3
    m = dict(xe=_mlp2(xd, H), ze=_mlp2(zd, H) if zd else None, ve=_mlp2(vd, H), ie=_mlp2(idim, H), xdec=_mlp2(H, xd), idec=_mlp2(H, idim),
             de=_mlp2(3 * nblk * H, H), ae=_mlp2((2 * nblk - 1) * H, H))
    t = (torch.arange(Tn, dtype=torch.float32) * 0.01).view(1, Tn, 1).repeat(B, 1, 1)
    if ragged_clock and B > 1:
        t[1:] = t[1:] * (0.5 + torch.rand(B - 1, 1, 1, generator=g))
    r = lambda *s: 0.3 * torch.randn(*s, generator=g)
    x, z, v, i, x0 = r(B, Tn, xd), r(B, Tn, zd), r(B, Tn, vd), r(B, Tn, idim), r(B, xd)
    ev = zj = vj = None
    if events and Tn > 4:
        ev = t[:, [1, Tn - 2], :].contiguous()
        zj, vj = r(B, 2, zd), r(B, 2, vd)
    return m, t, x, z, v, i, x0, ev, zj, vj


def _oracle(method, m, t, x, z, v, i, x0, ev, zj, vj):
    """neural_01_DAE_02_direct_encode.py:125-153 with the oracle's integrate_dae in the middle."""
    P = lambda a: a.permute(1, 0, 2)
    encz = (lambda a: a) if m["ze"] is None else (lambda a: _apply(m["ze"], a))
    Xh0, Xh, Zh, Vh, Ih = _apply(m["xe"], x0), _apply(m["xe"], x), encz(z), _apply(m["ve"], v), _apply(m["ie"], i)
    a0 = torch.cat((Xh0, Zh[:, 0], Vh[:, 0], Ih[:, 0]), -1)
    Zj = encz(zj) if ev is not None else None
    Vj = _apply(m["ve"], vj) if ev is not None else None
    Xs, Is = O.integrate_dae(method, m["de"], m["ae"], Xh0, P(t), P(Xh), P(Zh), P(Vh), P(Ih), a0, ev, Zj, Vj)
    x_pred = _apply(m["xdec"], Xs)
    x_pred[0] = x0
    return P(x_pred), P(_apply(m["idec"], Is)), _apply(m["xdec"], Xh), _apply(m["idec"], Ih)


def _dev(ls):
    return None if ls is None else [(w.cuda(), b.cuda()) for w, b in ls]


def _run(method, m, t, x, z, v, i, x0, ev, zj, vj, **kw):
    c = lambda a: None if a is None else a.cuda()
    return fused().dae_encoded_integrate(method, _dev(m["xe"]), _dev(m["ze"]), _dev(m["ve"]), _dev(m["ie"]), _dev(m["xdec"]), _dev(m["idec"]),
                                         _dev(m["de"]), _dev(m["ae"]), c(x0), c(t), c(x), c(z), c(v), c(i), event_t=c(ev), z_jump=c(zj),
                                         v_jump=c(vj), **kw)


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("B,Tn,xd,zd,vd,idim", [(37, 23, 8, 2, 2, 2), (16, 9, 5, 0, 3, 1), (1, 2, 1, 1, 1, 1), (21, 1, 8, 2, 2, 2), (20, 12, 16, 8, 8, 8),
                                                 (3, 40, 12, 0, 5, 6), (33, 7, 8, 3, 2, 4)])
def test_dae_encoded_forward_matches_oracle(method, B, Tn, xd, zd, vd, idim):
    case = _case(B, Tn, xd, zd, vd, idim, seed=B * 100 + Tn)
    ref = _oracle(method, *case)
    out = _run(method, *case)
    for k, (o, r_, name) in enumerate(zip(out, ref, ("x_pred", "i_pred", "x_re", "i_re"))):
        assert o.shape == r_.shape, (name, o.shape, r_.shape)
        assert traj_rel_err(o.cpu(), r_, bdim=0) <= TOL_GPU, (name, method)
    assert torch.equal(out[0][:, 0].cpu(), case[6])          # x_pred[0] = x0, bit for bit (neural_01_DAE_02_direct_encode.py:150)


def test_dae_encoded_forward_strided_inputs_and_no_recon():
    """Inputs as non-contiguous slices of one wide tensor (element strides travel through the C ABI); reconstructions skipped."""
    B, Tn, xd, zd, vd, idim = 19, 15, 8, 2, 2, 2
    m, t, x, z, v, i, x0, ev, zj, vj = _case(B, Tn, xd, zd, vd, idim, seed=5)
    ref = _oracle("rk4", m, t, x, z, v, i, x0, ev, zj, vj)
    big = torch.zeros(B, Tn + 3, xd + zd + vd + idim + 5).cuda()
    o = 2
    views = []
    for q in (x, z, v, i):
        big[:, 1:Tn + 1, o:o + q.shape[-1]] = q.cuda()
        views.append(big[:, 1:Tn + 1, o:o + q.shape[-1]])
        o += q.shape[-1]
    assert not views[0].is_contiguous()
    out = _run("rk4", m, t, *views, x0, ev, zj, vj, want_recon=False)
    assert out[2] is None and out[3] is None
    assert traj_rel_err(out[0].cpu(), ref[0], bdim=0) <= TOL_GPU and traj_rel_err(out[1].cpu(), ref[1], bdim=0) <= TOL_GPU


def _golden_model(method):
    from py_psnode_amd import models
    from py_psnode_amd import neural_dae as nd
    d = load("g7_grad_dae02_h64.npz")
    m = models.DAE_Model(8, 2, 2, 2, 64, direct_encode=True)
    m.load_state_dict({k[4:].replace("__", "."): T(v) for k, v in d.items() if k.startswith("sd__")})
    m = m.cuda()
    m.solver = {"euler": nd.Euler, "midpoint": nd.Midpoint, "rk4": nd.RK4}[method]()
    m.solver.fused = "require"
    m.one_launch = True              # K3g is opt-in (the row kernels + K3c are 8-12 % faster at hidden 64: DESIGN.md K3g)
    return d, m


@pytest.mark.parametrize("method", METHODS)
def test_model_forward_takes_the_single_launch_route_and_matches_the_reference(method):
    """models.DAE_Model(direct_encode, hidden 64) under no_grad: ONE fused launch behind Init_Func, equal to the reference's golden
    forward (the four outputs of the reference's own DAE_Model, make_goldens_r2/r3) and to the row-kernel + K3c route training uses."""
    d, m = _golden_model(method)
    g = lambda k: T(d[k]).cuda()
    kw = dict(t=g("t"), x=g("x"), z=g("z"), v=g("v"), i=g("i"), event_t=g("event_t"), z_jump=g("z_jump"), v_jump=g("v_jump"))
    calls = []
    orig = fused().dae_encoded_integrate
    try:
        fused().dae_encoded_integrate = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        with torch.no_grad():
            outs = m(**kw)
    finally:
        fused().dae_encoded_integrate = orig
    assert calls == [1], "the no-grad direct_encode DAE forward must be ONE fused launch"
    for k in range(4):
        assert traj_rel_err(outs[k].cpu(), d[f"{method}_out{k}"], bdim=0) <= TOL_GPU, k
    outs2 = m(**kw)                      # autograd on: row kernels + K3c + row kernels
    assert outs2[0].requires_grad
    for k in range(4):
        assert traj_rel_err(outs2[k].detach().cpu(), outs[k].cpu(), bdim=0) <= TOL_GPU, k


def test_single_launch_route_steps_aside_for_hooks_and_the_generic_kernel():
    d, m = _golden_model("rk4")
    g = lambda k: T(d[k]).cuda()
    kw = dict(t=g("t"), x=g("x"), z=g("z"), v=g("v"), i=g("i"), event_t=g("event_t"), z_jump=g("z_jump"), v_jump=g("v_jump"))
    calls = []
    orig = fused().dae_encoded_integrate
    seen = []
    try:
        fused().dae_encoded_integrate = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        h = m.v_encoder.register_forward_hook(lambda mod, inp, out: seen.append(1))
        with torch.no_grad():
            ref = m(**kw)
        h.remove()
        assert calls == [] and seen, "a forward hook on an encoder must fire: module-by-module route"
        m.solver.kernel = "generic"
        with torch.no_grad():
            m(**kw)
        assert calls == []
        m.solver.kernel = "auto"
        with torch.no_grad():
            out = m(**kw)
        assert calls == [1]
    finally:
        fused().dae_encoded_integrate = orig
    for k in range(4):
        assert traj_rel_err(out[k].cpu(), ref[k].cpu(), bdim=0) <= TOL_GPU


@pytest.mark.parametrize("zd", [2, 0])
def test_dae_encoded_forward_full_size_matches_unfused_route(zd):
    """B=4096 x 200 steps (every CU busy): the one launch against the row kernels + K3c on the same inputs, all four outputs."""
    from py_psnode_amd import models
    from py_psnode_amd import neural_dae as nd
    torch.manual_seed(3)
    B, Tn = 4096, 201
    m = models.DAE_Model(8, zd, 2, 2, 64, direct_encode=True, solver=nd.RK4()).cuda()
    m.solver.fused = "require"
    m.one_launch = True
    g = torch.Generator().manual_seed(4)
    r = lambda *s: (0.3 * torch.randn(*s, generator=g)).cuda()
    t = (torch.arange(Tn, dtype=torch.float32) * 0.01).view(1, Tn, 1).repeat(B, 1, 1).cuda()
    kw = dict(t=t, x=r(B, Tn, 8), z=r(B, Tn, zd), v=r(B, Tn, 2), i=r(B, Tn, 2), event_t=t[:, [3, 150], :].contiguous(), z_jump=r(B, 2, zd),
              v_jump=r(B, 2, 2))
    with torch.no_grad():
        one = m(**kw)
        m.one_launch = False
        ref = m(**kw)
    for k in range(4):
        assert torch.isfinite(one[k]).all()
        assert traj_rel_err(one[k].cpu(), ref[k].cpu(), bdim=0) <= TOL_GPU, k


def test_dae_encoded_unsupported_shape_raises():
    """hidden 16 (the other direct_encode width) is not this kernel's: UnsupportedShapeError, and the model takes the old route."""
    from py_psnode_amd import _lib
    global H
    H_keep = H
    try:
        H = 16
        case = _case(5, 6, 8, 2, 2, 2, seed=1)
    finally:
        H = H_keep
    with pytest.raises(_lib.UnsupportedShapeError):
        _run("rk4", *case)


def test_default_route_is_the_row_kernels_and_the_env_knob_selects_k3g(monkeypatch):
    d, m = _golden_model("euler")
    m.one_launch = None
    g = lambda k: T(d[k]).cuda()
    kw = dict(t=g("t"), x=g("x"), z=g("z"), v=g("v"), i=g("i"), event_t=g("event_t"), z_jump=g("z_jump"), v_jump=g("v_jump"))
    calls = []
    orig = fused().dae_encoded_integrate
    try:
        fused().dae_encoded_integrate = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        monkeypatch.delenv("PSNODE_DAE02_ONE_LAUNCH", raising=False)
        with torch.no_grad():
            m(**kw)
        assert calls == []
        monkeypatch.setenv("PSNODE_DAE02_ONE_LAUNCH", "1")
        with torch.no_grad():
            m(**kw)
        assert calls == [1]
    finally:
        fused().dae_encoded_integrate = orig
