"""Round-4 goldens, generated from the REAL reference (/root/reference) in the build container -- data only, never source:

  g8_tf_grad_<tag>.npz: what the reference's `loss.backward()` produces through a TEACHER-FORCED integrate_ODE / integrate_DAE
  (my_solvers.py:72-74, 111-121): the no_encode models at hidden 64 and 128, B=8, T=21, two events, per-trajectory clocks, Euler /
  Midpoint / RK4, loss = sum_k (out_k * G_k).sum().
    ode01[_h128]        integrate_ODE(..., input_true_x=True) called as ODE_Model.forward does (the flag sits commented out there,
                        neural_00_ODE_01_no_encode.py:88)
    dae01[_h128]        DAE_Model.forward(input_true_x=, input_true_i=) for the three teacher-forced combinations
                        (neural_01_DAE_01_no_encode.py:96,112-113), keys <method>_tx<a>_ti<b>_*
  Stored: inputs, weights, outputs, every parameter .grad and the gradients of z, v, z_jump, v_jump (the dataset rows x, i carry no
  requires_grad: the fused route gives them no gradient).

    python tests/golden/make_goldens_r4.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_goldens import REF, grid, load_reference, make_events, rnd, save, sd_arrays, solvers  # noqa: E402

P = lambda a: a.permute(1, 0, 2)


def g8(nd, mods):
    ode01, dae01 = mods["neural_00_ODE_01_no_encode"], mods["neural_01_DAE_01_no_encode"]
    xd, zd, vd, idim = 8, 2, 2, 2
    B, T = 8, 21
    cases = (("ode01", 90, lambda: ode01.ODE_Model(xd, zd, 64)), ("ode01_h128", 91, lambda: ode01.ODE_Model(xd, zd, 128)),
             ("dae01", 92, lambda: dae01.DAE_Model(xd, zd, vd, idim, 64)), ("dae01_h128", 93, lambda: dae01.DAE_Model(xd, zd, vd, idim, 128)))
    for tag, seed, make in cases:
        torch.manual_seed(seed)
        g = torch.Generator().manual_seed(seed + 100)
        m = make()
        t = grid(B, T)
        t[1:] = t[1:] * (0.5 + torch.rand(B - 1, 1, 1, generator=g))          # per-trajectory clocks (trajectory 0 decides events)
        x, z, v, i = rnd(g, B, T, xd), rnd(g, B, T, zd), rnd(g, B, T, vd), rnd(g, B, T, idim)
        ev, zj, vj = make_events(t, g, zd, steps=(5, 13), vd=vd)
        is_dae = tag.startswith("dae")
        out = dict(t=t, x=x, z=z, v=v, i=i, event_t=ev, z_jump=zj, v_jump=vj)
        out.update(sd_arrays(m, "sd__"))
        Gs = None
        combos = ((True, False), (False, True), (True, True)) if is_dae else ((True, False),)
        for name, s in solvers(nd).items():
            for tx, ti in combos:
                m.solver = s
                m.zero_grad()
                leaves = {k: a.clone().requires_grad_(True) for k, a in (("z", z), ("v", v), ("z_jump", zj), ("v_jump", vj))}
                if is_dae:
                    res = m(t=t, x=x, z=leaves["z"], v=leaves["v"], i=i, event_t=ev, z_jump=leaves["z_jump"], v_jump=leaves["v_jump"],
                            input_true_x=tx, input_true_i=ti)
                else:       # ODE_Model.forward with the flag it keeps commented out
                    m.event.set_event(t=ev, z=leaves["z_jump"])
                    a0 = torch.cat((P(x)[0], P(leaves["z"])[0]), dim=-1)
                    res = P(m.solver.integrate_ODE(x_func=m.de_func, t=P(t), x=P(x), z=P(leaves["z"]), all_initial=a0, event_fn=m.event.event_fn,
                                                   jump_change_fn=m.event.jump_change_fn, input_true_x=True))
                res = res if isinstance(res, tuple) else (res,)
                if Gs is None:
                    Gs = [torch.randn(r.shape, generator=g) for r in res]
                    for k, G in enumerate(Gs):
                        out[f"G{k}"] = G
                sum((r * G).sum() for r, G in zip(res, Gs)).backward()
                key = f"{name}_tx{int(tx)}_ti{int(ti)}"
                for k, r in enumerate(res):
                    out[f"{key}_out{k}"] = r.detach().contiguous()
                for k, p in m.named_parameters():
                    out[f"{key}_gp__" + k.replace(".", "__")] = (p.grad if p.grad is not None else torch.zeros_like(p)).clone()
                for k, a in leaves.items():
                    if a.grad is not None:
                        out[f"{key}_g_{k}"] = a.grad.clone()
        save(f"g8_tf_grad_{tag}.npz", **out)


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference not mounted; goldens can only be regenerated in the build container")
    nd_, mods_ = load_reference()
    g8(nd_, mods_)
