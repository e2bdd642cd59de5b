"""Whole training step of the four scripts' models as shipped (hidden 64 / 16 / 64 / 64), B=4096 x T=1001 on one MI355X:
model forward (fused integrator inside) + the script's loss (fused K6) + backward + Adam step."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from py_psnode_amd import loss as L, models  # noqa: E402
from py_psnode_amd import neural_dae as nd  # noqa: E402

dev = torch.device("cuda", 0)
B, T = int(os.environ.get("B", 4096)), int(os.environ.get("T", 1001))
H_OVERRIDE = int(os.environ.get("HIDDEN", 0))      # 0 = the widths the scripts ship with
g = torch.Generator().manual_seed(0)
r = lambda *s: (0.1 * torch.randn(*s, generator=g)).to(dev)
t = (torch.arange(T, dtype=torch.float32) * 0.01).view(1, T, 1).repeat(B, 1, 1).to(dev)
x, z, v, i = r(B, T, 8), r(B, T, 2), r(B, T, 2), r(B, T, 2)
ev, zj, vj = -torch.ones(B, 2, 1, device=dev), torch.zeros(B, 2, 2, device=dev), torch.zeros(B, 2, 2, device=dev)
mask8, mask1 = torch.ones(B, T, 8, device=dev), torch.ones(B, T, 1, device=dev)
out = {}
for tag in sys.argv[1:] or ["ode01", "ode02", "dae01", "dae02"]:
    for method in ("rk4", "euler"):
        solver = {"rk4": nd.RK4, "euler": nd.Euler}[method]()
        if tag == "ode01":
            m = models.ODE_Model(8, 2, H_OVERRIDE or 64, solver=solver)
        elif tag == "ode02":
            m = models.ODE_Model(8, 2, H_OVERRIDE or 16, direct_encode=True, solver=solver)
        elif tag == "dae01":
            m = models.DAE_Model(8, 2, 2, 2, H_OVERRIDE or 64, solver=solver)
        else:
            m = models.DAE_Model(8, 2, 2, 2, H_OVERRIDE or 64, direct_encode=True, solver=solver)
        m = m.to(dev)
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        steps_T = min(T, int(os.environ.get("T_CAP", T)))

        def step():
            opt.zero_grad()
            sl = slice(0, steps_T)
            if tag.startswith("ode"):
                o = m(t=t[:, sl], x=x[:, sl], z=z[:, sl], event_t=ev, z_jump=zj)
                if tag == "ode01":
                    loss = L.ode01_loss(o, x[:, sl], mask8[:, sl])[0]
                else:
                    loss = L.ode02_loss(o[0], o[1], x[:, sl], mask8[:, sl])[0]
            else:
                o = m(t=t[:, sl], x=x[:, sl], z=z[:, sl], v=v[:, sl], i=i[:, sl], event_t=ev, z_jump=zj, v_jump=vj)
                if tag == "dae01":
                    loss = L.dae01_loss(o[0], x[:, sl], o[1], i[:, sl], mask1[:, sl])[0]
                else:
                    loss = L.dae02_loss(o[0], o[1], o[2], o[3], x[:, sl], i[:, sl], mask1[:, sl])[0]
            route = type(o[0].grad_fn if isinstance(o, tuple) else o.grad_fn).__name__
            loss.backward()
            opt.step()
            return route

        route = step(); step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 3
        for _ in range(n):
            step()
        host_ms = (time.perf_counter() - t0) / n * 1e3          # enqueue time: a value near ms_per_train_step means host-bound
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        out[f"{tag}:{method}"] = {"ms_per_train_step": round(ms, 2), "host_enqueue_ms": round(host_ms, 2), "grid_points": steps_T, "state_steps_per_s": round(B * (steps_T - 1) / ms * 1e3),
                                  "prediction_grad_fn": route}
        print(tag, method, out[f"{tag}:{method}"], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/train_step_models.json", "w"), indent=1)
