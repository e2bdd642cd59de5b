#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REAL reference in this container.

Run (build container only -- /root/reference does not exist on the GPU box):

    python tests/golden/make_goldens.py

It imports /root/reference/neural_dae and the four neural_0x scripts (with the 3-line `ray` stub of
SURVEY.md App. A), feeds them seeded synthetic inputs and stores inputs, weights (raw arrays, never
pickled reference classes) and outputs as .npz.  Only data is committed; no reference source travels.

Golden sets:
  g1_single_step.npz    step_integrate, 3 methods, ODE + DAE branch                      -> a4-a7
  g2_ode_*.npz          integrate_ODE B=32 T=101: plain / events / teacher forcing / ragged dt -> a2,a14
  g3_dae_*.npz          integrate_DAE: 4 teacher-forcing combos, events, x_dim==0        -> a3
  g4_model_*.npz        ODE_Model / DAE_Model forwards of the four scripts               -> a8-a13
  g5_long.npz           RK4 B=8 T=1001 drift budget
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    ray = types.ModuleType("ray")
    rw = types.ModuleType("ray.worker")
    rw.init = lambda *a, **k: None
    ray.worker = rw
    sys.modules["ray"] = ray
    sys.modules["ray.worker"] = rw
    sys.path.insert(0, REF)
    import warnings
    warnings.filterwarnings("ignore")
    nd = importlib.import_module("neural_dae")
    mods = {n: importlib.import_module(n) for n in (
        "neural_00_ODE_01_no_encode", "neural_00_ODE_02_direct_encode",
        "neural_01_DAE_01_no_encode", "neural_01_DAE_02_direct_encode")}
    return nd, mods


def sd_arrays(module, prefix=""):
    return {prefix + k.replace(".", "__"): v.detach().numpy().copy() for k, v in module.state_dict().items()}


def grid(B, T, h=0.01):
    return (torch.arange(T, dtype=torch.float32) * h).view(1, T, 1).repeat(B, 1, 1).contiguous()


def rnd(gen, *shape, scale=0.1):
    return (scale * torch.randn(*shape, generator=gen)).float()


def make_events(t, gen, zd, steps=(20, 55), vd=None):
    """event_t[B,nE,1] holding the grid times of `steps`; z_jump[B,nE,zd]."""
    B = t.shape[0]
    ev = torch.stack([t[:, s, :] for s in steps], dim=1).contiguous()       # [B,nE,1]
    zj = rnd(gen, B, len(steps), zd)
    if vd is None:
        return ev, zj
    return ev, zj, rnd(gen, B, len(steps), vd)


def solvers(nd):
    return {"euler": nd.Euler(), "midpoint": nd.Midpoint(), "rk4": nd.RK4()}


def save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    print(f"  wrote {name}: {os.path.getsize(path) / 1024:.0f} KiB")


@torch.no_grad()
def main():
    nd, mods = load_reference()
    ode01, ode02, dae01, dae02 = (mods[n] for n in (
        "neural_00_ODE_01_no_encode", "neural_00_ODE_02_direct_encode",
        "neural_01_DAE_01_no_encode", "neural_01_DAE_02_direct_encode"))
    xd, zd, vd, idim, H = 8, 2, 2, 2, 64

    # ---------------------------------------------------------------- G1 single step
    torch.manual_seed(10)
    g = torch.Generator().manual_seed(11)
    de_o = ode01.DE_Func(xd, zd, H)
    de_d = dae01.DE_Func(xd, zd, vd, idim, H)
    B = 4
    x0, z0, v0, i0 = rnd(g, B, xd), rnd(g, B, zd), rnd(g, B, vd), rnd(g, B, idim)
    a0_o = torch.cat((rnd(g, B, xd), rnd(g, B, zd)), -1)
    a0_d = torch.cat((rnd(g, B, xd), rnd(g, B, zd), rnd(g, B, vd), rnd(g, B, idim)), -1)
    t0 = torch.full((B, 1), 0.3)
    dt = torch.tensor([[0.01], [0.02], [0.005], [0.0]])
    t1 = t0 + dt
    out = dict(x0=x0, z0=z0, v0=v0, i0=i0, a0_ode=a0_o, a0_dae=a0_d, t0=t0, dt=dt, t1=t1)
    out.update(sd_arrays(de_o, "ode__"))
    out.update(sd_arrays(de_d, "dae__"))
    for name, s in solvers(nd).items():
        x1, f0 = s.step_integrate(func=de_o, t0=t0, dt=dt, t1=t1, x0=x0, z0=z0, all_initial=a0_o)
        out[f"ode_{name}_x1"], out[f"ode_{name}_f0"] = x1, f0
        x1, f0 = s.step_integrate(func=de_d, t0=t0, dt=dt, t1=t1, x0=x0, z0=z0, v0=v0, i0=i0, all_initial=a0_d)
        out[f"dae_{name}_x1"], out[f"dae_{name}_f0"] = x1, f0
    save("g1_single_step.npz", **out)

    # ---------------------------------------------------------------- G2 integrate_ODE
    torch.manual_seed(20)
    g = torch.Generator().manual_seed(21)
    de_o = ode01.DE_Func(xd, zd, H)
    B, T = 32, 101
    t = grid(B, T)
    # ragged clocks: trajectories 1..3 have non-uniform per-trajectory dt, trajectory 5 is -1 padded from k=60
    tr = t.clone()
    tr[1] = torch.cumsum(torch.cat((torch.zeros(1), 0.005 + 0.01 * torch.rand(T - 1, generator=g))), 0).view(T, 1)
    tr[2] = tr[2] * 2.0
    tr[3, 50:] = tr[3, 50:] + 0.004
    tr[5, 60:] = -1.0
    x, z = rnd(g, B, T, xd), rnd(g, B, T, zd)
    ev, zj = make_events(t, g, zd)
    a0 = torch.cat((x[:, 0], z[:, 0]), -1)
    no_ev = torch.full((B, 2, 1), -1.0)
    out = dict(t=t, t_ragged=tr, x=x, z=z, event_t=ev, z_jump=zj, all_initial=a0)
    out.update(sd_arrays(de_o, "de__"))
    event = nd.ODE_Event()
    P = lambda a: a.permute(1, 0, 2)
    for name, s in solvers(nd).items():
        event.set_event(no_ev, zj)
        out[f"{name}_plain"] = s.integrate_ODE(x_func=de_o, t=P(t), x=P(x), z=P(z), all_initial=a0,
                                               event_fn=event.event_fn, jump_change_fn=event.jump_change_fn)
        out[f"{name}_noevfn"] = s.integrate_ODE(x_func=de_o, t=P(t), x=P(x), z=P(z), all_initial=a0)
        event.set_event(ev, zj)
        out[f"{name}_events"] = s.integrate_ODE(x_func=de_o, t=P(t), x=P(x), z=P(z), all_initial=a0,
                                                event_fn=event.event_fn, jump_change_fn=event.jump_change_fn)
        out[f"{name}_events_truex"] = s.integrate_ODE(x_func=de_o, t=P(t), x=P(x), z=P(z), all_initial=a0,
                                                      event_fn=event.event_fn, jump_change_fn=event.jump_change_fn,
                                                      input_true_x=True)
        out[f"{name}_ragged"] = s.integrate_ODE(x_func=de_o, t=P(tr), x=P(x), z=P(z), all_initial=a0,
                                                event_fn=event.event_fn, jump_change_fn=event.jump_change_fn)
    save("g2_ode.npz", **out)

    # ---------------------------------------------------------------- G3 integrate_DAE
    torch.manual_seed(30)
    g = torch.Generator().manual_seed(31)
    de_d = dae01.DE_Func(xd, zd, vd, idim, H)
    ae_d = dae01.AE_Func(xd, zd, vd, idim, H)
    B, T = 16, 61
    t = grid(B, T)
    x, z, v, i = rnd(g, B, T, xd), rnd(g, B, T, zd), rnd(g, B, T, vd), rnd(g, B, T, idim)
    x_init = rnd(g, B, xd)
    ev, zj, vj = make_events(t, g, zd, steps=(10, 33), vd=vd)
    a0 = torch.cat((x_init, z[:, 0], v[:, 0], i[:, 0]), -1)
    out = dict(t=t, x=x, z=z, v=v, i=i, x_init=x_init, event_t=ev, z_jump=zj, v_jump=vj, all_initial=a0)
    out.update(sd_arrays(de_d, "de__"))
    out.update(sd_arrays(ae_d, "ae__"))
    event = nd.DAE_Event()
    for name, s in solvers(nd).items():
        for tx in (False, True):
            for ti in (False, True):
                for use_ev in (False, True):
                    if use_ev:
                        event.set_event(ev, zj, vj)
                        kw = dict(event_fn=event.event_fn, jump_change_fn=event.jump_change_fn)
                    else:
                        kw = {}
                    xs, is_ = s.integrate_DAE(x_init=x_init, x_func=de_d, i_func=ae_d, t=P(t), x=P(x), z=P(z), v=P(v), i=P(i),
                                              all_initial=a0, input_true_x=tx, input_true_i=ti, **kw)
                    key = f"{name}_tx{int(tx)}_ti{int(ti)}_ev{int(use_ev)}"
                    out[key + "_x"], out[key + "_i"] = xs, is_
        # dataset with x_dim == 0 (my_solvers.py:97): output width comes from x_init
        xe = torch.zeros(B, T, 0)
        xs, is_ = s.integrate_DAE(x_init=x_init, x_func=de_d, i_func=ae_d, t=P(t), x=P(xe), z=P(z), v=P(v), i=P(i), all_initial=a0)
        out[f"{name}_xdim0_x"], out[f"{name}_xdim0_i"] = xs, is_
    save("g3_dae.npz", **out)

    # ---------------------------------------------------------------- G4 model-level forwards
    B, T = 8, 41
    for tag, seed in (("ode01", 40), ("ode02", 41), ("dae01", 42), ("dae02", 43), ("dae02_z0", 44)):
        torch.manual_seed(seed)
        g = torch.Generator().manual_seed(seed + 100)
        t = grid(B, T)
        zdim = 0 if tag == "dae02_z0" else zd
        x, z, v, i = rnd(g, B, T, xd), rnd(g, B, T, zdim), rnd(g, B, T, vd), rnd(g, B, T, idim)
        ev, zj, vj = make_events(t, g, zdim, steps=(7, 19), vd=vd)
        out = dict(t=t, x=x, z=z, v=v, i=i, event_t=ev, z_jump=zj, v_jump=vj)
        if tag == "ode01":
            m = ode01.ODE_Model(xd, zd, H)
        elif tag == "ode02":
            m = ode02.ODE_Model(xd, zd, 16)
        elif tag == "dae01":
            m = dae01.DAE_Model(xd, zd, vd, idim, H)
        else:
            m = dae02.DAE_Model(xd, zdim, vd, idim, 16)
        out.update(sd_arrays(m, "sd__"))
        for name, s in solvers(nd).items():
            m.solver = s
            if tag.startswith("ode"):
                res = m(t=t, x=x, z=z, event_t=ev, z_jump=zj)
            else:
                res = m(t=t, x=x, z=z, v=v, i=i, event_t=ev, z_jump=zj, v_jump=vj)
            res = res if isinstance(res, tuple) else (res,)
            for k, r in enumerate(res):
                out[f"{name}_out{k}"] = r.contiguous()
        if tag == "dae01":
            m.solver = nd.RK4()
            r = m(t=t, x=x, z=z, v=v, i=i, event_t=ev, z_jump=zj, v_jump=vj, input_true_x=True, input_true_i=True)
            out["rk4_truexi_out0"], out["rk4_truexi_out1"] = r[0].contiguous(), r[1].contiguous()
        save(f"g4_model_{tag}.npz", **out)

    # ---------------------------------------------------------------- G5 long run (drift budget)
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(1)
    de_o = ode01.DE_Func(xd, zd, H)
    B, T = 8, 1001
    t = grid(B, T)
    x0 = rnd(g, B, 1, xd)
    x = torch.cat((x0, torch.zeros(B, T - 1, xd)), 1)
    z = rnd(g, B, T, zd)
    a0 = torch.cat((x[:, 0], z[:, 0]), -1)
    out = dict(t=t, x0=x0, z=z, all_initial=a0)
    out.update(sd_arrays(de_o, "de__"))
    out["rk4"] = nd.RK4().integrate_ODE(x_func=de_o, t=P(t), x=P(x), z=P(z), all_initial=a0)
    out["euler"] = nd.Euler().integrate_ODE(x_func=de_o, t=P(t), x=P(x), z=P(z), all_initial=a0)
    save("g5_long.npz", **out)


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference not mounted; goldens can only be regenerated in the build container")
    main()
