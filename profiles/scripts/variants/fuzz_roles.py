"""Random-shape fuzz of the two-role backward kernels (round 4): K4f / K7f on saved activations at hidden <= 64 -- every external-slot class,
events, ragged tiles, odd and even step counts, grad_is = None -- must be BIT-EQUAL to the one-role instances of the same library
(PSNODE_K4F_NO_ROLES / PSNODE_K7F_NO_ROLES = 1: same arithmetic in the same order, only the wave that issues it differs).
usage (GPU box, repo root): python profiles/scripts/fuzz_roles.py [seed] [iterations]"""
import os, random, sys
import torch, torch.nn as nn
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from py_psnode_amd import fused
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 4321)
bad = 0
BMAX, TMAX = int(os.environ.get("FUZZ_BMAX", "70")), int(os.environ.get("FUZZ_TMAX", "14"))
mk = lambda dims: [(l.weight.detach().cuda(), l.bias.detach().cuda()) for l in [nn.Linear(dims[k], dims[k + 1]) for k in range(len(dims) - 1)]]


def both(run, flat, tag):
    global bad
    for v in ("PSNODE_K4F_NO_ROLES", "PSNODE_K7F_NO_ROLES"): os.environ.pop(v, None)
    two = flat(run())
    os.environ["PSNODE_K4F_NO_ROLES"] = "1"; os.environ["PSNODE_K7F_NO_ROLES"] = "1"
    one = flat(run())
    for v in ("PSNODE_K4F_NO_ROLES", "PSNODE_K7F_NO_ROLES"): os.environ.pop(v, None)
    for k, (p, q) in enumerate(zip(two, one)):
        if not torch.equal(p, q):
            bad += 1; print("DIFFERS", tag, "output", k, f"max |diff| {float((p - q).abs().max()):.3e}")


for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    H = random.choice([8, 16, 24, 32, 40, 48, 64])
    method = random.choice(["euler", "midpoint", "rk4"])
    B, Tn = random.randint(1, BMAX), random.randint(2, TMAX)
    events = Tn > 4 and random.random() < 0.6
    g = torch.Generator().manual_seed(it)
    torch.manual_seed(it)
    r = lambda *s_: (0.1 * torch.randn(*s_, generator=g)).cuda()
    t = (torch.arange(Tn, dtype=torch.float32) * 0.02).view(Tn, 1, 1).repeat(1, B, 1)
    if B > 1: t[:, 1:] = t[:, 1:] * (0.5 + torch.rand(1, B - 1, 1, generator=g))
    t = t.cuda()
    evs = sorted(random.sample(range(1, Tn - 1), 2)) if events else None
    ev = torch.stack([t[evs[0]], t[evs[1]]], dim=1).contiguous() if events else None
    if random.random() < 0.5:
        xd, zd = random.randint(1, 8), random.randint(0, 8)
        tag = ("ode", H, method, B, Tn, xd, zd, events)
        layers = mk([3 * (xd + zd), H, H, H, xd])
        x, z = r(Tn, B, xd), r(Tn, B, zd)
        a0 = torch.cat((x[0], z[0]), -1)
        zj = r(B, 2, zd) if events else None
        G = torch.randn(Tn, B, xd, generator=g).cuda()
        xs, saved = fused.ode_integrate(method, layers, t, x, z, a0, event_t=ev, z_jump=zj, save=True)
        tab = fused.event_table(t, ev) if events else None
        run = lambda: fused.ode_backward(method, layers, t, z, a0, xs, G, event_idx=tab, z_jump=zj, saved=saved, kernel="wide")
        flat = lambda o: [q for q in o[:4] if q is not None] + list(o[4])
    else:
        while True:
            xd, zd, vd, idim = random.randint(1, 8), random.randint(0, 4), random.randint(0, 4), random.randint(1, 4)
            if zd + vd + idim <= 8 and zd + vd >= 1: break
        tag = ("dae", H, method, B, Tn, xd, zd, vd, idim, events)
        n = xd + zd + vd + idim
        de, ae = mk([3 * n, H, H, H, xd]), mk([n + xd + zd + vd, H, H, H, idim])
        z, v, xi, i0 = r(Tn, B, zd), r(Tn, B, vd), r(B, xd), r(B, idim)
        a0 = torch.cat((xi, z[0], v[0], i0), -1)
        zj, vj = (r(B, 2, zd), r(B, 2, vd)) if events else (None, None)
        Gx, Gi = torch.randn(Tn, B, xd, generator=g).cuda(), (torch.randn(Tn, B, idim, generator=g).cuda() if random.random() < 0.8 else None)
        xe, ie = torch.zeros(Tn, B, 0, device="cuda"), torch.zeros(Tn, B, idim, device="cuda")
        xs, is_, saved = fused.dae_integrate(method, de, ae, xi, t, xe, z, v, ie, a0, event_t=ev, z_jump=zj, v_jump=vj, save=True)
        tab = fused.event_table(t, ev) if events else None
        run = lambda: fused.dae_backward(method, de, ae, t, z, v, a0, xs, is_, Gx, Gi, event_idx=tab, z_jump=zj, v_jump=vj, kernel="wide", saved=saved)
        flat = lambda o: [o[k_] for k_ in ("x_init", "z", "v", "z_jump", "v_jump", "all_initial") if o[k_] is not None] + list(o["de"]) + list(o["ae"])
    if os.environ.get("FUZZ_VERBOSE") == "1": print(it, tag, flush=True)
    try:
        both(run, flat, tag)
    except Exception as e:      # a shape the wide kernels do not take is not this fuzz's business
        print("skipped", tag, type(e).__name__, str(e)[:80])
torch.cuda.synchronize()
print("roles fuzz done, mismatches:", bad)
