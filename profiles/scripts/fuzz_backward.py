"""Random-shape fuzz of the wide backwards -- round 3: the one-launch kernels K4f / K7f (+ K7h) in their recompute and saved-activation forms,
and round 2's split routes K4w / K7w in time chunks -- any hidden width <= 128, every slot class, events, ragged tiles -- against the generic
backward K5.  usage (GPU box, repo root): python profiles/scripts/fuzz_backward.py [seed] [iterations]"""
import sys, random, torch, torch.nn as nn
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_gpu_backward as tb
from py_psnode_amd import fused
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 1234)
bad = 0
import os
VERBOSE = os.environ.get("FUZZ_VERBOSE", "0") == "1"
FROM = int(os.environ.get("FUZZ_FROM", "0"))
OLD = os.environ.get("FUZZ_OLD", "0") == "1"
BMAX, TMAX = int(os.environ.get("FUZZ_BMAX", "70")), int(os.environ.get("FUZZ_TMAX", "14"))
def close(a, b, what, tag):
    global bad
    if b is None:
        assert a is None, (what, tag); return
    sc = float(b.abs().max()); er = float((a - b).abs().max())
    if not er <= 3e-4 * max(sc, 1e-5):
        bad += 1; print("MISMATCH", tag, what, f"err {er:.2e} scale {sc:.2e}")
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    H = random.choice([8, 24, 32, 40, 64, 96, 128])
    method = random.choice(["euler", "midpoint", "rk4"])
    B, Tn = random.randint(1, BMAX), random.randint(2, TMAX)
    events = Tn > 4 and random.random() < 0.6
    chunk = random.choice([None, 1, 3, 5])
    use_gi = True if OLD else random.random() < 0.8
    if random.random() < 0.5:
        xd, zd = random.randint(1, 8), random.randint(0, 8)
        tag = ("ode", H, method, B, Tn, xd, zd, events, chunk)
        if VERBOSE: print(it, tag, flush=True)
        if it < FROM: continue
        g = torch.Generator().manual_seed(it)
        torch.manual_seed(it)
        lin = [nn.Linear(a_, b_) for a_, b_ in zip([3 * (xd + zd), H, H, H], [H, H, H, xd])]
        layers = [(m.weight.detach().cuda(), m.bias.detach().cuda()) for m in lin]
        t = (torch.arange(Tn, dtype=torch.float32) * 0.02).view(Tn, 1, 1).repeat(1, B, 1)
        if B > 1: t[:, 1:] = t[:, 1:] * (0.5 + torch.rand(1, B - 1, 1, generator=g))
        r = lambda *s_: (0.1 * torch.randn(*s_, generator=g)).cuda()
        x_in, z = torch.zeros(Tn, B, xd, device="cuda"), r(Tn, B, zd)
        x_in[0] = r(B, xd)
        a0 = torch.cat((x_in[0], z[0]), -1)
        ev = zj = tab = None
        if events and zd > 0:
            ev = torch.stack([t[1, :, :], t[Tn - 2, :, :]], dim=1).contiguous().cuda(); zj = r(B, 2, zd)
            tab = fused.event_table(t.cuda(), ev)
        G = torch.randn(Tn, B, xd, generator=g).cuda()
        xs, saved = fused.ode_integrate(method, layers, t.cuda(), x_in, z, a0, event_t=ev, z_jump=zj, save=True)
        if VERBOSE: torch.cuda.synchronize(); print("    fwd ok", flush=True)
        b = fused.ode_backward(method, layers, t.cuda(), z, a0, xs, G, event_idx=tab, z_jump=zj, kernel="generic")
        if VERBOSE: torch.cuda.synchronize(); print("    generic ok", flush=True)
        runs = [("K4f", lambda: fused.ode_backward(method, layers, t.cuda(), z, a0, xs, G, event_idx=tab, z_jump=zj, kernel="wide")),
                ("K4f saved", lambda: fused.ode_backward(method, layers, t.cuda(), z, a0, xs, G, event_idx=tab, z_jump=zj, kernel="wide", saved=saved))]
        for name, fn in runs:
            if VERBOSE: print("   ", name, flush=True)
            a = fn()
            if VERBOSE: torch.cuda.synchronize()
            for n_, p, q in zip(["gx0", "gz", "gzj", "ga0"], a[:4], b[:4]): close(p, q, f"{name} {n_}", tag)
            for k, (p, q) in enumerate(zip(a[4], b[4])): close(p, q, f"{name} param {k}", tag)
    else:
        while True:
            xd, zd, vd, idim = random.randint(1, 8), random.randint(0, 4), random.randint(0, 4), random.randint(1, 4)
            if zd + vd >= 1 and zd + vd + idim <= 8: break
        tag = ("dae", H, method, B, Tn, xd, zd, vd, idim, events, chunk)
        if OLD: use_gi = random.random() < 0.8      # (the draw order of the first round-3 run, kept to replay its seeds)
        if VERBOSE: print(it, tag, flush=True)
        if it < FROM: continue
        de, ae, t, z, v, xi, a0, ev, zj, vj, Gx, Gi = tb._dae_raw_case(B, Tn, xd, zd, vd, idim, 1000 + it, events, H=H)
        xe, ie = torch.zeros(Tn, B, 0, device="cuda"), torch.zeros(Tn, B, idim, device="cuda")
        if not fused.dae_backward_wide_supported(method, de, ae, xd, zd, vd, idim):
            print("unsupported", tag); continue
        xs, is_, saved = fused.dae_integrate(method, de, ae, xi, t, xe, z, v, ie, a0, event_t=ev, z_jump=zj, v_jump=vj, save=True)
        tab = fused.event_table(t, ev) if ev is not None else None
        gi = Gi if use_gi else None
        b = fused.dae_backward(method, de, ae, t, z, v, a0, xs, is_, Gx, gi, event_idx=tab, z_jump=zj, v_jump=vj, kernel="generic")
        kw = dict(event_idx=tab, z_jump=zj, v_jump=vj)
        runs = [("K7f sliced", lambda: fused._dae_backward_wide_sliced(16, method, de, ae, t, z, v, a0, xs, is_, Gx, gi, tab, zj, vj, None, None)),
                ("K7f", lambda: fused.dae_backward_wide(method, de, ae, t, z, v, a0, xs, is_, Gx, gi, **kw)),
                ("K7f saved", lambda: fused.dae_backward_wide(method, de, ae, t, z, v, a0, xs, is_, Gx, gi, saved=saved, **kw))]
        for name, fn in runs:
            if VERBOSE: print("   ", name, flush=True)
            a = fn()
            if VERBOSE: torch.cuda.synchronize()
            for key in ("x_init", "z", "v", "z_jump", "v_jump", "all_initial"): close(a[key], b[key], f"{name} {key}", tag)
            for grp in ("de", "ae"):
                for k, (p, q) in enumerate(zip(a[grp], b[grp])): close(p, q, f"{name} {grp} {k}", tag)
print("fuzz done, mismatches:", bad)
