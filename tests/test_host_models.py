"""Host-side mirror of the reference interface, checked on CPU against goldens captured from the reference.

These run the solver's callback walk (CPU tensors can never take the HIP route) -- they pin the model glue,
event semantics, keyword conventions and state-dict key names, not the kernels.
"""
import numpy as np
import pytest
import torch

from helpers import TOL_ORACLE, T, load, rel_err
from py_psnode_amd import models
from py_psnode_amd import neural_dae as nd

SOLVERS = {"euler": nd.Euler, "midpoint": nd.Midpoint, "rk4": nd.RK4}


def _load_sd(model, d):
    sd = {k[4:].replace("__", "."): T(v) for k, v in d.items() if k.startswith("sd__")}
    assert set(sd) == set(model.state_dict()), "state-dict keys differ from the reference's"
    model.load_state_dict(sd)


def _build(tag):
    if tag == "ode01":
        return models.ODE_Model(8, 2, 64)
    if tag == "ode02":
        return models.ODE_Model(8, 2, 16, direct_encode=True)
    if tag == "dae01":
        return models.DAE_Model(8, 2, 2, 2, 64)
    if tag == "dae02":
        return models.DAE_Model(8, 2, 2, 2, 16, direct_encode=True)
    return models.DAE_Model(8, 0, 2, 2, 16, direct_encode=True)


@pytest.mark.parametrize("tag", ["ode01", "ode02", "dae01", "dae02", "dae02_z0"])
@pytest.mark.parametrize("method", ["euler", "midpoint", "rk4"])
def test_model_forward_matches_reference(tag, method):
    d = load(f"g4_model_{tag}.npz")
    m = _build(tag)
    _load_sd(m, d)
    m.solver = SOLVERS[method]()
    with torch.no_grad():
        if tag.startswith("ode"):
            out = m(t=T(d["t"]), x=T(d["x"]), z=T(d["z"]), event_t=T(d["event_t"]), z_jump=T(d["z_jump"]))
        else:
            out = m(t=T(d["t"]), x=T(d["x"]), z=T(d["z"]), v=T(d["v"]), i=T(d["i"]), event_t=T(d["event_t"]),
                    z_jump=T(d["z_jump"]), v_jump=T(d["v_jump"]))
    out = out if isinstance(out, tuple) else (out,)
    for k, o in enumerate(out):
        assert rel_err(o, d[f"{method}_out{k}"]) <= TOL_ORACLE, (tag, method, k)


def test_dae01_teacher_forcing():
    d = load("g4_model_dae01.npz")
    m = _build("dae01")
    _load_sd(m, d)
    m.solver = nd.RK4()
    with torch.no_grad():
        xs, is_ = m(t=T(d["t"]), x=T(d["x"]), z=T(d["z"]), v=T(d["v"]), i=T(d["i"]), event_t=T(d["event_t"]),
                    z_jump=T(d["z_jump"]), v_jump=T(d["v_jump"]), input_true_x=True, input_true_i=True)
    assert rel_err(xs, d["rk4_truexi_out0"]) <= TOL_ORACLE
    assert rel_err(is_, d["rk4_truexi_out1"]) <= TOL_ORACLE


def test_output_layout_like_reference():
    """solver returns contiguous [T,B,D]; the model's permuted view has strides (xd, B*xd, 1)."""
    d = load("g4_model_ode01.npz")
    m = _build("ode01")
    _load_sd(m, d)
    with torch.no_grad():
        out = m(t=T(d["t"]), x=T(d["x"]), z=T(d["z"]), event_t=T(d["event_t"]), z_jump=T(d["z_jump"]))
    B, Tn, xd = d["x"].shape
    assert out.shape == (B, Tn, xd) and out.stride() == (xd, B * xd, 1)


def test_training_walk_backpropagates():
    """The scripts call loss.backward() through the integrator (neural_00_ODE_01_no_encode.py:359)."""
    torch.manual_seed(0)
    m = models.ODE_Model(3, 1, 8, solver=nd.RK4())
    B, Tn = 4, 6
    t = (torch.arange(Tn) * 0.01).view(1, Tn, 1).repeat(B, 1, 1)
    x, z = 0.1 * torch.randn(B, Tn, 3), 0.1 * torch.randn(B, Tn, 1)
    out = m(t=t, x=x, z=z, event_t=torch.full((B, 1, 1), -1.0), z_jump=torch.zeros(B, 1, 1))
    out.pow(2).mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


@pytest.mark.parametrize("method", ["euler", "midpoint", "rk4"])
def test_integrate_ode_x_init_extension_equals_the_upstream_call(method):
    """integrate_ODE(..., x_init=s) == integrate_ODE with x[0] = s (my_solvers.py:53-79), values and gradients: the gradient reaches the
    start state through x_init alone, x itself gets none; x_init and teacher forcing exclude each other."""
    torch.manual_seed(3)
    f = models.DE_Func(4, (8, 8, 8), 3).double()
    solver = {"euler": nd.Euler, "midpoint": nd.Midpoint, "rk4": nd.RK4}[method]()
    Tn, B = 7, 5
    t = (torch.arange(Tn, dtype=torch.float64) * 0.01).view(Tn, 1, 1).repeat(1, B, 1)
    x, z = 0.1 * torch.randn(Tn, B, 3, dtype=torch.float64), 0.1 * torch.randn(Tn, B, 1, dtype=torch.float64)
    s0 = 0.1 * torch.randn(B, 3, dtype=torch.float64)
    a0 = torch.cat((s0, z[0]), dim=-1)
    xa = x.clone(); xa[0] = s0
    xa.requires_grad_(True)
    ref = solver.integrate_ODE(x_func=f, t=t, x=xa, z=z, all_initial=a0)
    ref.pow(2).sum().backward()
    xb, sb = x.clone().requires_grad_(True), s0.clone().requires_grad_(True)
    got = solver.integrate_ODE(x_func=f, t=t, x=xb, z=z, all_initial=a0, x_init=sb)
    assert torch.equal(got, ref)
    got.pow(2).sum().backward()
    assert xb.grad is None and torch.allclose(sb.grad, xa.grad[0], rtol=1e-12, atol=0)
    with pytest.raises(ValueError):
        solver.integrate_ODE(x_func=f, t=t, x=xb, z=z, all_initial=a0, x_init=sb, input_true_x=True)


def test_solver_public_surface():
    s = nd.RK4()
    for attr in ("order", "step_size", "interp", "grid_constructor", "enable_cal_time", "assert_time", "cal_time", "total_time"):
        assert hasattr(s, attr)
    assert (nd.Euler.order, nd.Midpoint.order, nd.RK4.order) == (1, 2, 4)
    with pytest.raises(ValueError):
        nd.Euler(step_size=0.1, grid_constructor=lambda f, x, t: t)
    with pytest.raises(TypeError):
        nd.FixedGridODESolver()          # abstract, as upstream


def test_step_integrate_matches_reference():
    d = load("g1_single_step.npz")
    de = models.DE_Func(10, (64, 64, 64), 8)
    de.load_state_dict({k[5:].replace("__", "."): T(v) for k, v in d.items() if k.startswith("ode__")})
    dd = models.DAE_DE_Func(14, (64, 64, 64), 8)
    dd.load_state_dict({k[5:].replace("__", "."): T(v) for k, v in d.items() if k.startswith("dae__")})
    x0, z0, v0, i0, t0, dt, t1 = (T(d[k]) for k in ("x0", "z0", "v0", "i0", "t0", "dt", "t1"))
    with torch.no_grad():
        for name, cls in SOLVERS.items():
            x1, f0 = cls().step_integrate(func=de, t0=t0, dt=dt, t1=t1, x0=x0, z0=z0, all_initial=T(d["a0_ode"]))
            assert rel_err(x1, d[f"ode_{name}_x1"]) <= TOL_ORACLE and rel_err(f0, d[f"ode_{name}_f0"]) <= TOL_ORACLE
            x1, f0 = cls().step_integrate(func=dd, t0=t0, dt=dt, t1=t1, x0=x0, z0=z0, v0=v0, i0=i0, all_initial=T(d["a0_dae"]))
            assert rel_err(x1, d[f"dae_{name}_x1"]) <= TOL_ORACLE and rel_err(f0, d[f"dae_{name}_f0"]) <= TOL_ORACLE


def test_event_objects():
    ev = nd.ODE_Event()
    assert ev.event_fn(torch.zeros(3, 1)) is False
    et = torch.tensor([[[0.2], [0.5]]]).repeat(3, 1, 1)
    zj = torch.arange(12.0).view(3, 2, 2)
    ev.set_event(et, zj)
    assert ev.event_fn(torch.full((3, 1), 0.5)) is True and ev.event_fn(torch.full((3, 1), 0.3)) is False
    z0 = torch.zeros(3, 2, requires_grad=True)
    out = ev.jump_change_fn(torch.full((3, 1), 0.5), z0)
    assert torch.equal(out, zj[:, 1]) and not out.requires_grad
    dup = nd.ODE_Event()
    dup.set_event(torch.tensor([[[0.5], [0.5]]]).repeat(3, 1, 1), zj)
    with pytest.raises(RuntimeError):
        dup.jump_change_fn(torch.full((3, 1), 0.5), torch.zeros(3, 2))


def test_require_mode_raises_on_cpu():
    m = models.ODE_Model(3, 1, 8, solver=nd.Euler())
    m.solver.fused = "require"
    B, Tn = 2, 3
    t = (torch.arange(Tn) * 0.01).view(1, Tn, 1).repeat(B, 1, 1)
    with torch.no_grad(), pytest.raises(nd.NotFusableError):
        m(t=t, x=torch.zeros(B, Tn, 3), z=torch.zeros(B, Tn, 1), event_t=torch.full((B, 1, 1), -1.0), z_jump=torch.zeros(B, 1, 1))


def test_dataset_npz_roundtrip(tmp_path):
    N, Tn = 6, 5
    rng = np.random.default_rng(0)
    p = tmp_path / "s.npz"
    np.savez(p, name=np.array([["a", "u"]]), t=np.tile(np.arange(Tn, dtype=np.float32).reshape(1, Tn, 1), (N, 1, 1)),
             x=rng.standard_normal((N, Tn, 3)).astype(np.float32), z=rng.standard_normal((N, Tn, 1)).astype(np.float32),
             event_t=np.full((N, 1, 1), -1, np.float32), z_jump=np.zeros((N, 1, 1), np.float32))
    ds = nd.ODE_Curves_Sample(str(p), "cpu", num_sample=4, cut_length=3)
    assert len(ds) == 4 and ds[0][1].shape == (3, 3) and ds.mask.shape == ds.x.shape
    idx = np.random.default_rng(42).choice(np.arange(N), 4, replace=False)
    assert np.array_equal(ds.x.numpy(), np.load(p)["x"][idx][:, :3])


# ----------------------------------------------------------------------------- rows a15 / a16 / f4: G6 (tests/golden/make_goldens_r2.py)
def _sd(d, prefix):
    return {k[len(prefix):].replace("__", "."): T(v) for k, v in d.items() if k.startswith(prefix)}


def test_legacy_per_variable_blocks_match_reference():
    """neural_base.DE_Func / AE_Func (neural_base.py:68-115,199-229), standalone with [B,n,1] inputs: golden G6."""
    d = load("g6_legacy_blocks.npz")
    xd, zd, vd, idim, H = (int(v) for v in d["dims"])
    de, ae = nd.DE_Func(xd, zd, H), nd.AE_Func(xd, vd, idim, H)
    sd_de, sd_ae = _sd(d, "de__"), _sd(d, "ae__")
    assert set(sd_de) == set(de.state_dict()) and set(sd_ae) == set(ae.state_dict()), "state-dict keys differ from the reference's"
    de.load_state_dict(sd_de)
    ae.load_state_dict(sd_ae)
    with torch.no_grad():
        assert rel_err(de.set_initial(T(d["x0"]), T(d["z0"])), d["de_set_initial"]) <= TOL_ORACLE
        assert rel_err(de.f_XZh0_H, d["de_f_XZh0_H"]) <= TOL_ORACLE
        assert rel_err(de.get_decode_x(T(d["Xht"])), d["de_decode"]) <= TOL_ORACLE
        assert rel_err(de(torch.zeros(5, 1), T(d["Xht"]), T(d["zt"])), d["de_forward"]) <= TOL_ORACLE
        assert rel_err(ae(T(d["Xht"]), T(d["vt"])), d["ae_forward"]) <= TOL_ORACLE


def _ode_base_case(d, method, dev):
    rhs = models.DE_Func(10, (64, 64, 64), 8)
    rhs.load_state_dict(_sd(d, "base_de__"))
    base = nd.ODE_Base(rhs.to(dev), SOLVERS[method]())
    c = lambda k: T(d[k]).to(dev)
    event = nd.ODE_Event()
    event.set_event(c("base_event_t"), c("base_z_jump"))
    P = lambda a: a.permute(1, 0, 2)
    return base, event, (P(c("base_t")), P(c("base_x")), P(c("base_z")), c("base_all_initial"))


@pytest.mark.parametrize("method", ["euler", "midpoint", "rk4"])
def test_ode_base_forward_matches_reference(method):
    """ODE_Base.forward (neural_base.py:118-133) is a pass-through to integrate_ODE without input_true_x."""
    d = load("g6_legacy_blocks.npz")
    base, event, args = _ode_base_case(d, method, "cpu")
    assert base.de_function is not None and isinstance(base.solver, nd.FixedGridODESolver)
    with torch.no_grad():
        assert rel_err(base(*args, event.event_fn, event.jump_change_fn), d[f"base_{method}"]) <= TOL_ORACLE
        assert rel_err(base(*args), d[f"base_{method}_noev"]) <= TOL_ORACLE
        assert rel_err(base(t=args[0], x=args[1], z=args[2], all_initial=args[3], event_fn=event.event_fn,
                            jump_change_fn=event.jump_change_fn), d[f"base_{method}"]) <= TOL_ORACLE


def _dae_base_case(d, dev, tx, ti):
    de = models.DAE_DE_Func(14, (64, 64, 64), 8)
    ae = models.AE_Func(26, (64, 64, 64), 2)
    de.load_state_dict(_sd(d, "de__"))
    ae.load_state_dict(_sd(d, "ae__"))
    c = lambda k: T(d[k]).to(dev)
    P = lambda a: a.permute(1, 0, 2)
    event = nd.DAE_Event()
    event.set_event(c("event_t"), c("z_jump"), c("v_jump"))
    args = dict(t=P(c("t")), x=P(c("x")), z=P(c("z")), v=P(c("v")), i=P(c("i")))
    return de.to(dev), ae.to(dev), event, args, c("x_init"), c("all_initial")


@pytest.mark.parametrize("tx,ti", [(False, False), (True, False), (False, True), (True, True)])
def test_dae_base_stores_fields_and_forwards_flags(tx, ti):
    """DAE_Base (neural_base.py:232-255): the ctor's fields as upstream; upstream's forward cannot run, this one is
    a superset taking x_init / all_initial -- checked against the reference's integrate_DAE goldens (G3) for every
    teacher-forcing combination the stored flags select."""
    d = load("g3_dae.npz")
    de, ae, event, args, x_init, a0 = _dae_base_case(d, "cpu", tx, ti)
    base = nd.DAE_Base(de, ae, nd.RK4(), flg_encode_x=False, flg_input_true_x=tx, flg_input_true_i=ti)
    assert (base.de_function, base.ae_function, base.flg_encode_x, base.flg_input_true_x, base.flg_input_true_i) == (de, ae, False, tx, ti)
    with pytest.raises(TypeError):
        base(**args)                                # upstream's call shape: TypeError there too
    with torch.no_grad():
        xs, is_ = base(**args, event_fn=event.event_fn, jump_change_fn=event.jump_change_fn, x_init=x_init, all_initial=a0)
    key = f"rk4_tx{int(tx)}_ti{int(ti)}_ev1"
    assert rel_err(xs, d[key + "_x"]) <= TOL_ORACLE and rel_err(is_, d[key + "_i"]) <= TOL_ORACLE


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["euler", "midpoint", "rk4"])
def test_ode_base_takes_the_fused_route_on_gpu(method):
    from helpers import TOL_GPU, traj_rel_err
    d = load("g6_legacy_blocks.npz")
    base, event, args = _ode_base_case(d, method, "cuda")
    base.solver.fused = "require"
    with torch.no_grad():
        assert traj_rel_err(base(*args, event.event_fn, event.jump_change_fn).cpu(), d[f"base_{method}"]) <= TOL_GPU
        assert traj_rel_err(base(*args).cpu(), d[f"base_{method}_noev"]) <= TOL_GPU


@pytest.mark.gpu
@pytest.mark.parametrize("tx,ti", [(False, False), (True, False), (False, True), (True, True)])
def test_dae_base_takes_the_fused_route_on_gpu(tx, ti):
    from helpers import TOL_GPU, traj_rel_err
    d = load("g3_dae.npz")
    de, ae, event, args, x_init, a0 = _dae_base_case(d, "cuda", tx, ti)
    base = nd.DAE_Base(de, ae, nd.RK4(), flg_input_true_x=tx, flg_input_true_i=ti)
    base.solver.fused = "require"
    with torch.no_grad():
        xs, is_ = base(**args, event_fn=event.event_fn, jump_change_fn=event.jump_change_fn, x_init=x_init, all_initial=a0)
    key = f"rk4_tx{int(tx)}_ti{int(ti)}_ev1"
    assert traj_rel_err(xs.cpu(), d[key + "_x"]) <= TOL_GPU and traj_rel_err(is_.cpu(), d[key + "_i"]) <= TOL_GPU


def test_fused_recognition_probes_the_forward_not_just_the_attribute_names():
    """a module with an `x_dot` Sequential of the right shape but ANOTHER forward (scaled output, swapped concat order,
    use of t0) must not be integrated with the hard-coded recipe.  The probe runs the module's own forward on a few rows."""
    import torch.nn as nn
    from py_psnode_amd import fused

    class Scaled(models.DE_Func):
        def forward(self, t0, xt, zt, all_initial):
            return 2.0 * super().forward(t0, xt, zt, all_initial)

    class Swapped(models.DE_Func):
        def forward(self, t0, xt, zt, all_initial):
            s = torch.cat((xt, zt), dim=-1)
            return self.x_dot(torch.cat((s, s - all_initial, all_initial), dim=-1))

    class UsesTime(models.DE_Func):
        def forward(self, t0, xt, zt, all_initial):
            return super().forward(t0, xt, zt, all_initial) * (1.0 + t0)

    class Renamed(nn.Module):            # the reference scripts' own class shape: unknown type, same recipe -> accepted by the probe
        def __init__(self):
            super().__init__()
            self.x_dot = models._elu_mlp(30, 64, 64, 64, 8)

        def forward(self, t0, xt, zt, all_initial):
            xtzt = torch.cat((xt, zt), dim=-1)
            return self.x_dot.forward(torch.cat((all_initial, xtzt - all_initial, xtzt), dim=-1))

    good = models.DE_Func(10, (64, 64, 64), 8)
    layers = fused.de_layers_of(good, 10, 8)
    assert fused._recipe_ok(good, layers, "de_ode", (8, 2))
    for cls in (Scaled, Swapped, UsesTime):
        m = cls(10, (64, 64, 64), 8)
        ls = fused.de_layers_of(m, 10, 8)
        assert ls is not None                      # the structural checks alone accept it ...
        assert not fused._recipe_ok(m, ls, "de_ode", (8, 2)), cls.__name__      # ... the probe does not
        assert m.__dict__["_psnode_probe"][1] is False                           # cached
    r = Renamed()
    assert fused._recipe_ok(r, fused.de_layers_of(r, 10, 8), "de_ode", (8, 2))
    ae = models.AE_Func(26, (64, 64, 64), 2)
    assert fused._recipe_ok(ae, fused.ae_layers_of(ae, 14, 12, 2), "ae", (8, 2, 2, 14))

    class BadAE(models.AE_Func):
        def forward(self, xt, zt, vt, all_initial):
            return self.i_calculator(torch.cat((all_initial, xt, vt, zt), dim=-1))
    bad = BadAE(26, (64, 64, 64), 2)
    assert not fused._recipe_ok(bad, fused.ae_layers_of(bad, 14, 12, 2), "ae", (8, 2, 2, 14))


def test_enc_hidden_extension_builds_the_64_to_16_reading_and_walks_on_cpu():
    """models.ODE_Model(enc_hidden=64): encoders / decoder of hidden 64 around a hidden-16 latent RHS (BASELINE's "enc/dec 64 -> 16
    latent"; an extension, None = upstream's single hidden_dim).  CPU walk == manual composition with the oracle's latent loop."""
    import torch.nn.functional as F
    from oracle import psnode_oracle as O
    from py_psnode_amd import models
    from py_psnode_amd import neural_dae as nd
    torch.manual_seed(5)
    m = models.ODE_Model(8, 2, 16, direct_encode=True, solver=nd.RK4(), enc_hidden=64)
    up = models.ODE_Model(8, 2, 16, direct_encode=True, solver=nd.RK4())
    assert [tuple(p.shape) for p in m.x_encoder.parameters()] == [(64, 8), (64,), (16, 64), (16,)]
    assert [tuple(p.shape) for p in m.x_decoder.parameters()] == [(64, 16), (64,), (8, 64), (8,)]
    assert [tuple(p.shape) for p in up.x_encoder.parameters()] == [(16, 8), (16,), (16, 16), (16,)]      # upstream layout untouched
    assert [tuple(p.shape) for p in m.de_func.x_dot.parameters()] == [tuple(p.shape) for p in up.de_func.x_dot.parameters()]
    B, Tn = 5, 9
    g = torch.Generator().manual_seed(6)
    t = (torch.arange(Tn, dtype=torch.float32) * 0.01).view(1, Tn, 1).repeat(B, 1, 1)
    x, z = 0.3 * torch.randn(B, Tn, 8, generator=g), 0.3 * torch.randn(B, Tn, 2, generator=g)
    ev, zj = t[:, [2, 6], :].contiguous(), 0.3 * torch.randn(B, 2, 2, generator=g)
    with torch.no_grad():
        pred, re = m(t=t, x=x, z=z, event_t=ev, z_jump=zj)
        seq = lambda s, a: F.linear(F.elu(F.linear(a, s[0].weight, s[0].bias)), s[2].weight, s[2].bias)
        P = lambda a: a.permute(1, 0, 2)
        Xh, Zh = seq(m.x_encoder, x), seq(m.z_encoder, z)
        de = [(l.weight, l.bias) for l in m.de_func.x_dot if isinstance(l, torch.nn.Linear)]
        sol = O.integrate_ode("rk4", de, P(t), P(Xh), P(Zh), torch.cat((Xh[:, 0], Zh[:, 0]), -1), ev, seq(m.z_encoder, zj))
        assert torch.allclose(pred, P(seq(m.x_decoder, sol)), atol=1e-6)
        assert torch.allclose(re, seq(m.x_decoder, Xh), atol=1e-6)


def test_padded_hidden_classes():
    from py_psnode_amd import fused
    assert [fused._padded_hidden(h) for h in (1, 16, 32, 33, 64, 65, 100, 128)] == [32, 32, 32, 64, 64, 128, 128, 128]
    m = torch.arange(6.0).view(3, 2)
    p = fused._pad_rows(m, 5)
    assert p.shape == (5, 2) and torch.equal(p[:3], m) and not p[3:].any() and fused._pad_rows(m, 3) is m
