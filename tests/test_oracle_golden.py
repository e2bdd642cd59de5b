"""Pin the CPU oracle against outputs captured from the real reference (tests/golden/make_goldens.py)."""
import pytest
import torch

from oracle import psnode_oracle as O
from helpers import TOL_ORACLE, T, layers, load, rel_err, tm

METHODS = ("euler", "midpoint", "rk4")


@pytest.mark.parametrize("method", METHODS)
def test_g1_single_step(method):
    d = load("g1_single_step.npz")
    x0, z0, v0, i0 = (T(d[k]) for k in ("x0", "z0", "v0", "i0"))
    t0, dt, t1 = T(d["t0"]), T(d["dt"]), T(d["t1"])
    de_o, de_d = layers(d, "ode__x_dot"), layers(d, "dae__x_dot")
    a0o, a0d = T(d["a0_ode"]), T(d["a0_dae"])
    x1, f0 = O.step(method, lambda xx: O.de_rhs(de_o, xx, (z0,), a0o), t0, dt, t1, x0)
    assert rel_err(x1, d[f"ode_{method}_x1"]) <= TOL_ORACLE
    assert rel_err(f0, d[f"ode_{method}_f0"]) <= TOL_ORACLE
    x1, f0 = O.step(method, lambda xx: O.de_rhs(de_d, xx, (z0, v0, i0), a0d), t0, dt, t1, x0)
    assert rel_err(x1, d[f"dae_{method}_x1"]) <= TOL_ORACLE
    assert rel_err(f0, d[f"dae_{method}_f0"]) <= TOL_ORACLE


@pytest.mark.parametrize("method", METHODS)
def test_g2_integrate_ode(method):
    d = load("g2_ode.npz")
    de = layers(d, "de__x_dot")
    t, tr, x, z = tm(d["t"]), tm(d["t_ragged"]), tm(d["x"]), tm(d["z"])
    a0, ev, zj = T(d["all_initial"]), T(d["event_t"]), T(d["z_jump"])
    no_ev = torch.full_like(ev, -1.0)
    assert rel_err(O.integrate_ode(method, de, t, x, z, a0, no_ev, zj), d[f"{method}_plain"]) <= TOL_ORACLE
    assert rel_err(O.integrate_ode(method, de, t, x, z, a0), d[f"{method}_noevfn"]) <= TOL_ORACLE
    assert rel_err(O.integrate_ode(method, de, t, x, z, a0, ev, zj), d[f"{method}_events"]) <= TOL_ORACLE
    assert rel_err(O.integrate_ode(method, de, t, x, z, a0, ev, zj, input_true_x=True), d[f"{method}_events_truex"]) <= TOL_ORACLE
    assert rel_err(O.integrate_ode(method, de, tr, x, z, a0, ev, zj), d[f"{method}_ragged"]) <= TOL_ORACLE


@pytest.mark.parametrize("method", METHODS)
def test_g3_integrate_dae(method):
    d = load("g3_dae.npz")
    de, ae = layers(d, "de__x_dot"), layers(d, "ae__i_calculator")
    t, x, z, v, i = (tm(d[k]) for k in ("t", "x", "z", "v", "i"))
    xi, a0 = T(d["x_init"]), T(d["all_initial"])
    ev, zj, vj = T(d["event_t"]), T(d["z_jump"]), T(d["v_jump"])
    for tx in (False, True):
        for ti in (False, True):
            for use_ev in (False, True):
                kw = dict(event_t=ev, z_jump=zj, v_jump=vj) if use_ev else {}
                xs, is_ = O.integrate_dae(method, de, ae, xi, t, x, z, v, i, a0, input_true_x=tx, input_true_i=ti, **kw)
                key = f"{method}_tx{int(tx)}_ti{int(ti)}_ev{int(use_ev)}"
                assert rel_err(xs, d[key + "_x"]) <= TOL_ORACLE, key
                assert rel_err(is_, d[key + "_i"]) <= TOL_ORACLE, key
    xe = torch.zeros(x.shape[0], x.shape[1], 0)
    xs, is_ = O.integrate_dae(method, de, ae, xi, t, xe, z, v, i, a0)
    assert rel_err(xs, d[f"{method}_xdim0_x"]) <= TOL_ORACLE
    assert rel_err(is_, d[f"{method}_xdim0_i"]) <= TOL_ORACLE


def test_g5_long_run():
    d = load("g5_long.npz")
    de = layers(d, "de__x_dot")
    t, z = tm(d["t"]), tm(d["z"])
    x = torch.zeros(t.shape[0], t.shape[1], 8)
    x[0] = T(d["x0"])[:, 0]
    a0 = T(d["all_initial"])
    assert rel_err(O.integrate_ode("rk4", de, t, x, z, a0), d["rk4"]) <= TOL_ORACLE
    assert rel_err(O.integrate_ode("euler", de, t, x, z, a0), d["euler"]) <= TOL_ORACLE


def test_event_table_matches_reference_semantics():
    d = load("g2_ode.npz")
    tab = O.event_step_table(tm(d["t"]), T(d["event_t"]))
    assert [k for k, e in enumerate(tab) if e >= 0] == [20, 55]
    assert tab[20] == 0 and tab[55] == 1


def test_oracle_losses_are_the_four_scripts_expressions():
    """The oracle's loss restatements against the scripts' expressions written out literally (Loss_func = F.mse_loss):
    neural_00_ODE_01_no_encode.py:353-355, neural_00_ODE_02_direct_encode.py:267-270, neural_01_DAE_01_no_encode.py:414-419,
    neural_01_DAE_02_direct_encode.py:359-365.  The four objectives differ (x0 term, column weight, reconstruction terms)."""
    from oracle import psnode_oracle as O
    Loss_func = torch.nn.functional.mse_loss
    g = torch.Generator().manual_seed(0)
    B, Tn, xd, idim = 6, 9, 8, 2
    r = lambda *s: torch.randn(*s, generator=g)
    x, x_pred, x_re, i, i_pred, i_re = r(B, Tn, xd), r(B, Tn, xd), r(B, Tn, xd), r(B, Tn, idim), r(B, Tn, idim), r(B, Tn, idim)
    mask_x, mask_1 = (torch.rand(B, Tn, xd, generator=g) > 0.3).float(), (torch.rand(B, Tn, 1, generator=g) > 0.3).float()
    x_loss = torch.sum(torch.sum(Loss_func(x_pred, x, reduction='none') * mask_x, dim=1), dim=0) / torch.sum(mask_x)
    assert torch.equal(O.ode_loss(x_pred, x, mask_x)[0], torch.sum(x_loss))
    x0_loss, x_recon_loss = Loss_func(x[:, 0, :], x_pred[:, 0, :]).view(1), Loss_func(x_re, x).view(1)
    assert torch.allclose(O.ode02_loss(x_pred, x_re, x, mask_x), torch.sum(x0_loss) + torch.sum(x_loss) + torch.sum(x_recon_loss), rtol=1e-6)
    mask = mask_1
    xl = (torch.sum(Loss_func(x_pred, x, reduction='none') * mask)
          + torch.sum(Loss_func(x_pred[:, :, 1:2], x[:, :, 1:2], reduction='none') * mask) * 9) / torch.sum(mask)
    il = torch.sum(Loss_func(i_pred, i, reduction='none') * mask) / torch.sum(mask)
    tail = Loss_func(x[:, 0, :], x_pred[:, 0, :]) + Loss_func(i[:, 0, :], i_pred[:, 0, :])
    assert torch.allclose(O.dae_loss(x_pred, x, i_pred, i, mask)[0], xl + il + tail, rtol=1e-6)
    xl2 = torch.sum(Loss_func(x_pred, x, reduction='none') * mask) / torch.sum(mask)
    assert torch.allclose(O.dae02_loss(x_pred, i_pred, x_re, i_re, x, i, mask), xl2 + il + tail + Loss_func(x_re, x) + Loss_func(i_re, i), rtol=1e-6)
