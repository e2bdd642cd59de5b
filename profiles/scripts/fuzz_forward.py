"""Random-shape fuzz of the MFMA integrators (K1 / K2: any hidden width <= 128, every slot class, events, teacher forcing; round 4: the
streamed widths 129..256 within the classes they carry -- ODE x_dim <= 8 up to 256 / any x_dim <= 16 up to 192, DAE up to 192 with
z+v+i <= 6) against the generic kernel K0.  usage (GPU box, repo root): python profiles/scripts/fuzz_forward.py [seed] [iterations]"""
import random
import sys

import torch
import torch.nn as nn

sys.path.insert(0, ".")
from py_psnode_amd import fused

random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 4321)
bad = 0


def close(a, b, what, tag):
    global bad
    sc = float(b.abs().amax()); er = float((a - b).abs().amax())
    if not er <= 2e-5 * max(sc, 1e-3):
        bad += 1
        print("MISMATCH", tag, what, f"err {er:.2e} scale {sc:.2e}")


mk = lambda dims: [(l.weight.detach().cuda(), l.bias.detach().cuda()) for l in [nn.Linear(dims[k], dims[k + 1]) for k in range(len(dims) - 1)]]
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    torch.manual_seed(it)
    H = random.choice([5, 16, 24, 32, 48, 64, 80, 128, 129, 160, 192, 200, 256])
    method = random.choice(["euler", "midpoint", "rk4"])
    B, Tn = random.randint(1, 70), random.randint(1, 14)
    r = lambda *s: 0.1 * torch.randn(*s, device="cuda")
    t = (torch.arange(Tn, dtype=torch.float32, device="cuda") * 0.02).view(Tn, 1, 1).repeat(1, B, 1)
    if B > 1:
        t[:, 1:] = t[:, 1:] * (0.5 + torch.rand(1, B - 1, 1, device="cuda"))
    events = Tn > 4 and random.random() < 0.6
    ev = torch.stack([t[1, :, :], t[Tn - 2, :, :]], dim=1).contiguous() if events else None
    tx, ti = random.random() < 0.3, random.random() < 0.3
    if random.random() < 0.5:
        xd, zd = random.randint(1, 16 if (H <= 192 and random.random() < 0.3) else 8), random.randint(0, 8)
        tag = ("ode", H, method, B, Tn, xd, zd, events, tx)
        de = mk([3 * (xd + zd), H, H, H, xd])
        x, z = r(Tn, B, xd), r(Tn, B, zd)
        a0 = torch.cat((x[0], z[0]), -1)
        zj = r(B, 2, zd) if events else None
        kw = dict(event_t=ev, z_jump=zj, input_true_x=tx)
        b = fused.ode_integrate(method, de, t, x, z, a0, kernel="generic", **kw)
        # hidden <= 64 with x_dim <= 8: both MFMA integrators (round 5: K1 "tile", K1x "wave"); otherwise the one "mfma" picks
        for kern in (("tile", "wave") if (H <= 64 and xd <= 8) else ("mfma",)):
            a = fused.ode_integrate(method, de, t, x, z, a0, kernel=kern, **kw)
            close(a, b, "xs " + kern, tag)
    else:
        while True:
            xd, zd, vd, idim = random.randint(1, 8), random.randint(0, 4), random.randint(0, 4), random.randint(1, 4)
            if zd + vd >= 1 and zd + vd + idim <= (8 if H <= 128 else 6):
                break
        H = min(H, 192)                                   # the DAE runs streamed up to 192
        tag = ("dae", H, method, B, Tn, xd, zd, vd, idim, events, tx, ti)
        n = xd + zd + vd + idim
        de, ae = mk([3 * n, H, H, H, xd]), mk([n + xd + zd + vd, H, H, H, idim])
        x, z, v, i, xi = r(Tn, B, xd), r(Tn, B, zd), r(Tn, B, vd), r(Tn, B, idim), r(B, xd)
        a0 = torch.cat((xi, z[0], v[0], i[0]), -1)
        kw = dict(event_t=ev, z_jump=r(B, 2, zd) if events else None, v_jump=r(B, 2, vd) if events else None, input_true_x=tx, input_true_i=ti)
        b = fused.dae_integrate(method, de, ae, xi, t, x, z, v, i, a0, kernel="generic", **kw)
        for kern in (("tile", "wave") if (H <= 64 and not tx and not ti) else ("mfma",)):      # K2 and K2x where both take the call
            a = fused.dae_integrate(method, de, ae, xi, t, x, z, v, i, a0, kernel=kern, **kw)
            close(a[0], b[0], "xs " + kern, tag); close(a[1], b[1], "is " + kern, tag)
print("fuzz done, mismatches:", bad)
