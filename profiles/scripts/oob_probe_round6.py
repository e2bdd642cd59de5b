"""Out-of-bounds probe for the round-6 kernels: K4x (the one-wave ODE backward) and K2x's saving instances.  Every device tensor a kernel reads
or writes -- inputs, the SAVED ROWS, the jump rows, the incoming gradient -- in turn ENDS exactly at the end of its own 32 MB allocation
(K4x addresses rows as <uniform base> + <32-bit lane offset> with trajectory indices clamped for a ragged last wave: a wrong clamp reads
past the rows), on ragged batches (B % 4 != 0, B < 4), T = 2, with and without events, with and without dL/dz."""
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
# ---- K4x: forward (saving) once, then the backward with each tensor at the end of its allocation
for (B, T, xd, zd, H) in [(37, 23, 8, 2, 64), (5, 11, 5, 3, 40), (130, 9, 8, 8, 64), (3, 70, 7, 0, 33), (1, 2, 1, 1, 64)]:
    ls = lin([3 * (xd + zd), H, H, H, xd])
    t = (torch.arange(T, dtype=torch.float32) * 0.01).view(T, 1, 1).repeat(1, B, 1).to(dev)
    x, z = r(T, B, xd), r(T, B, zd)
    a0 = torch.cat((x[0], z[0]), -1).contiguous()
    G = r(T, B, xd)
    ev = zj = tab = None
    if zd and T > 4:
        ev = torch.stack([t[1, :, :], t[T - 2, :, :]], dim=1).contiguous()
        zj = r(B, 2, zd)
        tab = fused.event_table(t, ev)
    for method in ("euler", "midpoint", "rk4"):
        xs, saved = fused.ode_integrate(method, ls, t, x, z, a0, event_t=ev, z_jump=zj, save=True)
        tensors = {"t": t, "z": z, "a0": a0, "xs": xs, "G": G, "act": saved[0], "xst": saved[1]}
        if zj is not None:
            tensors["zj"] = zj
        for nme in list(tensors) + ["none"]:
            q = dict(tensors); hold = None
            if nme != "none" and q[nme].numel():
                q[nme], hold = at_end(q[nme])
            for need_z in (True, False):
                fused.ode_backward(method, ls, q["t"], q["z"], q["a0"], q["xs"], q["G"], event_idx=tab, z_jump=q.get("zj"), saved=(q["act"], q["xst"]),
                                   need_grad_z=need_z, need_grad_zj=need_z, kernel="wave")
            torch.cuda.synchronize()
        print("ok K4x", B, T, xd, zd, H, method, flush=True)
# ---- K2x saving instances (forced: AUTO keeps K2 for the saving forward)
for (B, T, xd, zd, vd, idim, H) in [(37, 23, 8, 2, 2, 2, 64), (5, 11, 5, 1, 2, 3, 40), (3, 9, 8, 0, 2, 2, 64)]:
    n = xd + zd + vd + idim
    de, ae = lin([3 * n, H, H, H, xd]), lin([n + xd + zd + vd, H, H, H, idim])
    t = (torch.arange(T, dtype=torch.float32) * 0.01).view(T, 1, 1).repeat(1, B, 1).to(dev)
    z, v, xi, i0 = r(T, B, zd), r(T, B, vd), r(B, xd), r(B, idim)
    a0 = torch.cat((xi, z[0], v[0], i0), -1).contiguous()
    xe, ie = torch.zeros(T, B, 0, device=dev), torch.zeros(T, B, idim, device=dev)
    ev = torch.stack([t[1, :, :], t[T - 2, :, :]], dim=1).contiguous()
    zj, vj = r(B, 2, zd), r(B, 2, vd)
    for method in ("euler", "rk4"):
        tensors = {"xi": xi, "t": t, "z": z, "v": v, "a0": a0, "zj": zj, "vj": vj}
        for nme in list(tensors) + ["none"]:
            q = dict(tensors); hold = None
            if nme != "none" and q[nme].numel():
                q[nme], hold = at_end(q[nme])
            fused.dae_integrate(method, de, ae, q["xi"], q["t"], xe, q["z"], q["v"], ie, q["a0"], event_t=ev, z_jump=q["zj"], v_jump=q["vj"], save=True,
                                kernel="wave")
            torch.cuda.synchronize()
        print("ok K2x save", B, T, method, flush=True)
print("probe done", flush=True)
