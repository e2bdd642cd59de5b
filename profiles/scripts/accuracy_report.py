#!/usr/bin/env python3
"""Accuracy of the loaded libpsnode_hip.so (PSNODE_LIB_PATH) on the goldens: error vs the reference's fp32 output and
vs an fp64 evaluation of the same algorithm, under both metrics of tests/helpers.py."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import T, layers, load, rel_err, tm, traj_rel_err
from oracle import psnode_oracle as O
from py_psnode_amd import fused

d = load("g5_long.npz"); de = layers(d, "de__x_dot"); t, z = tm(d["t"]), tm(d["z"])
x = torch.zeros(t.shape[0], t.shape[1], 8); x[0] = T(d["x0"])[:, 0]; a0 = T(d["all_initial"])
D = lambda a: a.double()
truth = O.integrate_ode("rk4", [(D(w), D(b)) for w, b in de], D(t), D(x), D(z), D(a0))
ref = T(d["rk4"])
out = fused.ode_integrate("rk4", [(w.cuda(), b.cuda()) for w, b in de], t.cuda(), x.cuda(), z.cuda(), a0.cuda()).cpu()
print("K1 G5 rk4 1000 steps: traj_rel_err vs reference %.3e | vs fp64: ours %.3e reference %.3e | elementwise vs fp64: ours %.3e reference %.3e"
      % (traj_rel_err(out, ref), traj_rel_err(out, truth), traj_rel_err(ref, truth), rel_err(out, truth), rel_err(ref, truth)))
d = load("g2_ode.npz"); de = layers(d, "de__x_dot"); t, x, z = tm(d["t"]), tm(d["x"]), tm(d["z"]); a0 = T(d["all_initial"])
worst = 0.0
for m in ("euler", "midpoint", "rk4"):
    out = fused.ode_integrate(m, [(w.cuda(), b.cuda()) for w, b in de], t.cuda(), x.cuda(), z.cuda(), a0.cuda()).cpu()
    worst = max(worst, traj_rel_err(out, d[f"{m}_noevfn"]))
print("K1 G2 (100 steps, 3 methods) worst traj_rel_err vs reference %.3e" % worst)
# ---- K2 (DAE_01, golden G3) and K3f (ODE_02 whole model, golden G4): both metrics vs the reference's fp32 output
d = load("g3_dae.npz"); de, ae = layers(d, "de__x_dot"), layers(d, "ae__i_calculator")
c = lambda a: a.cuda()
t, x, z, v, i = (c(tm(d[k])) for k in ("t", "x", "z", "v", "i"))
dl = lambda ls: [(c(w), c(b)) for w, b in ls]
for m in ("euler", "midpoint", "rk4"):
    xs, is_ = fused.dae_integrate(m, dl(de), dl(ae), c(T(d["x_init"])), t, x, z, v, i, c(T(d["all_initial"])), event_t=c(T(d["event_t"])),
                                  z_jump=c(T(d["z_jump"])), v_jump=c(T(d["v_jump"])))
    kx, ki = d[f"{m}_tx0_ti0_ev1_x"], d[f"{m}_tx0_ti0_ev1_i"]
    print("K2 G3 %-8s x: traj_rel_err %.3e elementwise %.3e | i: traj_rel_err %.3e elementwise %.3e"
          % (m, traj_rel_err(xs.cpu(), kx), rel_err(xs.cpu(), kx), traj_rel_err(is_.cpu(), ki), rel_err(is_.cpu(), ki)))
from py_psnode_amd import models
from py_psnode_amd import neural_dae as nd
d = load("g4_model_ode02.npz")
mdl = models.ODE_Model(8, 2, 16, direct_encode=True)
mdl.load_state_dict({k[4:].replace("__", "."): T(vv) for k, vv in d.items() if k.startswith("sd__")})
mdl = mdl.cuda()
for m, cls in (("euler", nd.Euler), ("midpoint", nd.Midpoint), ("rk4", nd.RK4)):
    mdl.solver = cls(); mdl.solver.fused = "require"
    with torch.no_grad():
        pred, re = mdl(t=c(T(d["t"])), x=c(T(d["x"])), z=c(T(d["z"])), event_t=c(T(d["event_t"])), z_jump=c(T(d["z_jump"])))
    print("K3f G4-ode02 %-8s x_pred: traj_rel_err %.3e elementwise %.3e | x_re: traj_rel_err %.3e elementwise %.3e"
          % (m, traj_rel_err(pred.cpu(), d[f"{m}_out0"], bdim=0), rel_err(pred.cpu(), d[f"{m}_out0"]),
             traj_rel_err(re.cpu(), d[f"{m}_out1"], bdim=0), rel_err(re.cpu(), d[f"{m}_out1"])))
