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
