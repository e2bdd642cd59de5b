"""Out-of-bounds probe for the round-6 generic integrator K0 (MFMA layers; register / LDS / streamed forms) and the generic backward K5.  Every device tensor the kernel
reads -- clocks, x, z, v, i, the initial rows, the jump rows -- in turn ENDS exactly at the end of its own 32 MB allocation, on ragged
batches, T = 1, 2 and longer grids (the step loop loads the NEXT grid point's clocks and z | v rows a step ahead: the look-ahead of the
last step must stay inside the tensors), with and without events, with and without teacher forcing."""
import os, sys, torch
import torch.nn as nn
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from py_psnode_amd import fused
dev = torch.device("cuda", 0)
def at_end(t):
    big = torch.empty(8 * 1024 * 1024, dtype=t.dtype, device=dev)
    v = big[big.numel() - t.numel():].view(t.shape)
    v.copy_(t)
    return v, big
def lin(dims):
    ls = [nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:])]
    return [(m.weight.detach().to(dev), m.bias.detach().to(dev)) for m in ls]
r = lambda *s: (0.1 * torch.randn(*s)).to(dev)
torch.manual_seed(0)
# ---- ODE: register form (QM 8, QM 4), LDS form (5 layers), streamed form
for (B, T, xd, zd, hidden) in [(37, 9, 20, 2, (64, 64, 64)), (5, 1, 20, 2, (64, 64, 64)), (3, 2, 8, 2, (64, 64)), (18, 6, 8, 2, (64, 64, 64, 64)),
                               (21, 5, 8, 2, (320, 320, 320)), (9, 4, 4, 70, (64, 64)), (19, 6, 20, 2, (128, 128, 128)), (3, 2, 8, 2, (100,))]:
    ls = lin([3 * (xd + zd)] + list(hidden) + [xd])
    t = (torch.arange(T, dtype=torch.float32) * 0.01).view(T, 1, 1).repeat(1, B, 1).to(dev)
    x, z = r(T, B, xd), r(T, B, zd)
    a0 = torch.cat((x[0], z[0]), -1).contiguous()
    ev = zj = None
    if T > 4:
        ev = torch.stack([t[1, :, :], t[T - 2, :, :]], dim=1).contiguous()
        zj = r(B, 2, zd)
    tensors = {"t": t, "x": x, "z": z, "a0": a0}
    if zj is not None:
        tensors["zj"] = zj
    for nme in list(tensors) + ["none"]:
        q = dict(tensors); hold = None
        if nme != "none" and q[nme].numel():
            q[nme], hold = at_end(q[nme])
        for tx in (False, True):
            for method in ("euler", "rk4"):
                fused.ode_integrate(method, ls, q["t"], q["x"], q["z"], q["a0"], event_t=ev, z_jump=q.get("zj"), input_true_x=tx, kernel="generic")
        torch.cuda.synchronize()
    print("ok K0 ode", B, T, xd, zd, hidden, flush=True)
# ---- DAE
for (B, T, xd, zd, vd, idim, hde, hae) in [(37, 9, 8, 4, 6, 6, (64, 64, 64), (64, 64, 64)), (5, 1, 8, 4, 6, 6, (64, 64, 64), (64, 64, 64)),
                                           (3, 2, 8, 2, 2, 2, (64, 64, 64, 64), (64, 64, 64, 64)), (17, 6, 20, 10, 40, 40, (128, 128, 128), (128, 128, 128)),
                                           (9, 5, 4, 40, 45, 3, (64, 64), (64, 64))]:
    n = xd + zd + vd + idim
    de, ae = lin([3 * n] + list(hde) + [xd]), lin([n + xd + zd + vd] + list(hae) + [idim])
    t = (torch.arange(T, dtype=torch.float32) * 0.01).view(T, 1, 1).repeat(1, B, 1).to(dev)
    x, z, v, i, xi = r(T, B, xd), r(T, B, zd), r(T, B, vd), r(T, B, idim), r(B, xd)
    a0 = torch.cat((xi, z[0], v[0], i[0]), -1).contiguous()
    ev = zj = vj = None
    if T > 4:
        ev = torch.stack([t[1, :, :], t[T - 2, :, :]], dim=1).contiguous()
        zj, vj = r(B, 2, zd), r(B, 2, vd)
    tensors = {"xi": xi, "t": t, "x": x, "z": z, "v": v, "i": i, "a0": a0}
    if zj is not None:
        tensors["zj"] = zj; tensors["vj"] = vj
    for nme in list(tensors) + ["none"]:
        q = dict(tensors); hold = None
        if nme != "none" and q[nme].numel():
            q[nme], hold = at_end(q[nme])
        for tx, ti in ((False, False), (True, True)):
            fused.dae_integrate("rk4", de, ae, q["xi"], q["t"], q["x"], q["z"], q["v"], q["i"], q["a0"], event_t=ev, z_jump=q.get("zj"), v_jump=q.get("vj"),
                                input_true_x=tx, input_true_i=ti, kernel="generic")
        torch.cuda.synchronize()
    print("ok K0 dae", B, T, xd, zd, vd, idim, hde, flush=True)
# ---- K5, the generic backward (register path of the DE since round 6, staged path beyond it)
for (B, T, xd, zd, hidden) in [(37, 9, 20, 2, (64, 64, 64)), (3, 2, 8, 2, (64, 64)), (18, 6, 40, 2, (64,)), (9, 5, 44, 0, (40, 40)), (5, 4, 8, 2, (128, 128, 128))]:
    ls = lin([3 * (xd + zd)] + list(hidden) + [xd])
    t = (torch.arange(T, dtype=torch.float32) * 0.01).view(T, 1, 1).repeat(1, B, 1).to(dev)
    x, z = r(T, B, xd), r(T, B, zd)
    a0 = torch.cat((x[0], z[0]), -1).contiguous()
    G = r(T, B, xd)
    ev = zj = tab = None
    if zd and T > 4:
        ev = torch.stack([t[1, :, :], t[T - 2, :, :]], dim=1).contiguous()
        zj = r(B, 2, zd)
        tab = fused.event_table(t, ev)
    for method in ("euler", "rk4"):
        xs = fused.ode_integrate(method, ls, t, x, z, a0, event_t=ev, z_jump=zj, kernel="generic")
        tensors = {"t": t, "z": z, "a0": a0, "xs": xs, "G": G}
        if zj is not None:
            tensors["zj"] = zj
        for nme in list(tensors) + ["none"]:
            q = dict(tensors); hold = None
            if nme != "none" and q[nme].numel():
                q[nme], hold = at_end(q[nme])
            fused.ode_backward(method, ls, q["t"], q["z"], q["a0"], q["xs"], q["G"], event_idx=tab, z_jump=q.get("zj"), kernel="generic")
            torch.cuda.synchronize()
    print("ok K5 ode", B, T, xd, zd, hidden, flush=True)
for (B, T, xd, zd, vd, idim, H) in [(21, 7, 8, 4, 6, 6, 64), (3, 2, 5, 0, 3, 2, 24)]:
    n = xd + zd + vd + idim
    de, ae = lin([3 * n, H, H, H, xd]), lin([n + xd + zd + vd, H, H, H, idim])
    t = (torch.arange(T, dtype=torch.float32) * 0.01).view(T, 1, 1).repeat(1, B, 1).to(dev)
    z, v, xi, i0 = r(T, B, zd), r(T, B, vd), r(B, xd), r(B, idim)
    a0 = torch.cat((xi, z[0], v[0], i0), -1).contiguous()
    xe, ie = torch.zeros(T, B, 0, device=dev), torch.zeros(T, B, idim, device=dev)
    xs, is_ = fused.dae_integrate("rk4", de, ae, xi, t, xe, z, v, ie, a0, kernel="generic")
    Gx, Gi = r(T, B, xd), r(T, B, idim)
    tensors = {"t": t, "z": z, "v": v, "a0": a0, "xs": xs, "is": is_, "Gx": Gx, "Gi": Gi}
    for nme in list(tensors) + ["none"]:
        q = dict(tensors); hold = None
        if nme != "none" and q[nme].numel():
            q[nme], hold = at_end(q[nme])
        fused.dae_backward("rk4", de, ae, q["t"], q["z"], q["v"], q["a0"], q["xs"], q["is"], q["Gx"], q["Gi"], kernel="generic")
        torch.cuda.synchronize()
    print("ok K5 dae", B, T, xd, zd, vd, idim, H, flush=True)
print("probe done", flush=True)
