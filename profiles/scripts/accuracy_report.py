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
print("G5 rk4 1000 steps: traj_rel_err vs reference %.3e | vs fp64: ours %.3e reference %.3e | elementwise vs fp64: ours %.3e reference %.3e"
      % (traj_rel_err(out, ref), traj_rel_err(out, truth), traj_rel_err(ref, truth), rel_err(out, truth), rel_err(ref, truth)))
d = load("g2_ode.npz"); de = layers(d, "de__x_dot"); t, x, z = tm(d["t"]), tm(d["x"]), tm(d["z"]); a0 = T(d["all_initial"])
worst = 0.0
for m in ("euler", "midpoint", "rk4"):
    out = fused.ode_integrate(m, [(w.cuda(), b.cuda()) for w, b in de], t.cuda(), x.cuda(), z.cuda(), a0.cuda()).cpu()
    worst = max(worst, traj_rel_err(out, d[f"{m}_noevfn"]))
print("G2 (100 steps, 3 methods) worst traj_rel_err vs reference %.3e" % worst)
