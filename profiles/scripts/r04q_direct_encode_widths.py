"""Which route the direct_encode models take at hidden widths other than the latent kernels' 16 / 64, and what it costs (B=4096 x 200 steps)."""
import sys, time, warnings
import torch
sys.path.insert(0, ".")
from py_psnode_amd import models, neural_dae as nd
B, T = 4096, 201
r = lambda *s: (0.1 * torch.randn(*s)).cuda()
t = (torch.arange(T, dtype=torch.float32) * 0.01).view(1, T, 1).repeat(B, 1, 1).cuda()
x, z, v, i = r(B, T, 8), r(B, T, 2), r(B, T, 2), r(B, T, 2)
ev, zj, vj = torch.full((B, 2, 1), -1.0).cuda(), r(B, 2, 2), r(B, 2, 2)
for kind in ("ode02", "dae02"):
    for H in (16, 32, 64, 128):
        for method in ("euler",):
            torch.manual_seed(0)
            solver = nd.Euler()
            m = (models.ODE_Model(8, 2, H, direct_encode=True, solver=solver) if kind == "ode02" else models.DAE_Model(8, 2, 2, 2, H, direct_encode=True, solver=solver)).cuda()
            call = (lambda: m(t=t, x=x, z=z, event_t=ev, z_jump=zj)) if kind == "ode02" else (lambda: m(t=t, x=x, z=z, v=v, i=i, event_t=ev, z_jump=zj, v_jump=vj))
            with warnings.catch_warnings(record=True) as wlist:
                warnings.simplefilter("always")
                with torch.no_grad():
                    call(); torch.cuda.synchronize()
                    t0 = time.perf_counter(); call(); torch.cuda.synchronize(); fwd = (time.perf_counter() - t0) * 1e3
                try:
                    o = call(); loss = sum(q.float().pow(2).mean() for q in o); loss.backward(); torch.cuda.synchronize()
                    m.zero_grad(); t0 = time.perf_counter(); o = call(); loss = sum(q.float().pow(2).mean() for q in o); loss.backward(); torch.cuda.synchronize()
                    trn = (time.perf_counter() - t0) * 1e3
                except Exception as e:
                    trn = f"ERR {type(e).__name__}"
            walked = any("stepping through the Python callables" in str(w_.message) for w_ in wlist)
            print(f"{kind} H{H:<4d} {method}: forward {fwd:8.2f} ms, train step {trn if isinstance(trn, str) else f'{trn:8.2f} ms'}  walk={walked}  ({B} x {T - 1} steps)")
