"""Round 4: chase of the two defects round 3 fenced.  usage (GPU box): python profiles/scripts/r04_defects.py a|b [repeats]
 a: K4f recompute <Midpoint, NZM = 0, 8 waves> against K5 (the library under PSNODE_LIB_PATH decides workaround / DMA wait states)
 b: K7w / K7f with grad_is = NULL (PSNODE_DEBUG_GIS_NULL=1 lifts the host-side fence) against K5"""
import os, sys, torch, torch.nn as nn
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from py_psnode_amd import fused
which = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
e = lambda p, q: (float((p.double().cpu() - q.double().cpu()).abs().max()) / max(float(q.abs().max()), 1e-9)) if q is not None and q.numel() else 0.0

def case_a(H, method, B, Tn, xd, zd, seed):
    g = torch.Generator().manual_seed(seed); torch.manual_seed(seed)
    lin = [nn.Linear(a_, b_) for a_, b_ in zip([3 * (xd + zd), H, H, H], [H, H, H, xd])]
    layers = [(m.weight.detach().cuda(), m.bias.detach().cuda()) for m in lin]
    t = (torch.arange(Tn, dtype=torch.float32) * 0.02).view(Tn, 1, 1).repeat(1, B, 1).cuda()
    r = lambda *s_: (0.1 * torch.randn(*s_, generator=g)).cuda()
    x_in, z = torch.zeros(Tn, B, xd, device="cuda"), r(Tn, B, zd)
    x_in[0] = r(B, xd)
    a0 = torch.cat((x_in[0], z[0]), -1)
    G = torch.randn(Tn, B, xd, generator=g).cuda()
    xs = fused.ode_integrate(method, layers, t, x_in, z, a0)
    b = fused.ode_backward(method, layers, t, z, a0, xs, G, kernel="generic")
    worst = {}
    for rep in range(reps):
        a = fused.ode_backward(method, layers, t, z, a0, xs, G, kernel="wide")
        torch.cuda.synchronize()
        errs = {"gx0": e(a[0], b[0]), "ga0": e(a[3], b[3])}
        for k, (p, q) in enumerate(zip(a[4], b[4])): errs[f"p{k}"] = e(p, q)
        for k_, v_ in errs.items(): worst[k_] = max(worst.get(k_, 0.0), v_)
        if os.environ.get("DEFECT_VERBOSE"): print(f"     rep {rep}: " + " ".join(f"{k_}={v_:.1e}" for k_, v_ in errs.items()), flush=True)
    bad = {k_: f"{v_:.1e}" for k_, v_ in worst.items() if not v_ <= 3e-4}
    print(f"A H{H} {method} B{B} T{Tn} x{xd} z{zd} seed{seed}: {'BAD ' + str(bad) if bad else 'ok'}", flush=True)
    return bool(bad)

def case_b(H, method, B, Tn, xd, zd, vd, idim, seed):
    import test_gpu_backward as tb
    de, ae, t, z, v, xi, a0, ev, zj, vj, Gx, Gi = tb._dae_raw_case(B, Tn, xd, zd, vd, idim, seed, False, H=H)
    xe, ie = torch.zeros(Tn, B, 0, device="cuda"), torch.zeros(Tn, B, idim, device="cuda")
    xs, is_, saved = fused.dae_integrate(method, de, ae, xi, t, xe, z, v, ie, a0, save=True)
    b = fused.dae_backward(method, de, ae, t, z, v, a0, xs, is_, Gx, None, kernel="generic")
    nbad = 0
    for name, kw in (("K7f", {}), ("K7f saved", {"saved": saved}), ("split", {"fuse_de": False})):
        worst = {}
        for rep in range(reps):
            c = fused.dae_backward_wide(method, de, ae, t, z, v, a0, xs, is_, Gx, None, **kw)
            torch.cuda.synchronize()
            for grp in ("de", "ae"):
                for k, (p, q) in enumerate(zip(c[grp], b[grp])): worst[f"{grp}{k}"] = max(worst.get(f"{grp}{k}", 0.0), e(p, q) if torch.isfinite(p).all() else float("inf"))
            for k_ in ("x_init", "all_initial", "z", "v"):
                if b[k_] is not None: worst[k_] = max(worst.get(k_, 0.0), e(c[k_], b[k_]) if torch.isfinite(c[k_]).all() else float("inf"))
        bad = {k_: f"{v_:.1e}" for k_, v_ in worst.items() if not v_ <= 3e-4}
        nbad += bool(bad)
        print(f"B {name:10s} H{H} {method} B{B} T{Tn} dims {xd},{zd},{vd},{idim} seed{seed}: {'BAD ' + str(bad) if bad else 'ok'}", flush=True)
    return nbad

print("lib:", os.environ.get("PSNODE_LIB_PATH", "(in-tree)"), " poison:", os.environ.get("PSNODE_POISON", "0"), " gis-null:", os.environ.get("PSNODE_DEBUG_GIS_NULL", "0"))
n = 0
if which == "a1":      # the same case three times: first launch of the process, or this data?
    for _ in range(3):
        n += case_a(128, "midpoint", 48, 7, 8, 0, 308)
    n += case_a(128, "midpoint", 48, 7, 8, 0, 309)
    n += case_a(128, "midpoint", 96, 5, 8, 0, 308)
elif which == "a":
    for seed in range(300, 306):
        for xd in (8, 4, 5):
            n += case_a(128, "midpoint", 48, 7, xd, 0, seed + xd)
    n += case_a(96, "midpoint", 33, 9, 8, 0, 7)
    n += case_a(128, "rk4", 48, 7, 8, 0, 8)
    n += case_a(128, "euler", 48, 7, 8, 0, 9)
    n += case_a(128, "midpoint", 48, 7, 8, 2, 10)
else:
    os.environ.pop("PSNODE_DEBUG_GIS_NULL", None)
    for dims in ((4, 2, 0, 2), (4, 1, 1, 2), (8, 2, 0, 2), (4, 0, 1, 3), (3, 2, 0, 2), (8, 2, 2, 2), (4, 1, 0, 1)):
        for H, method, B, Tn in ((64, "rk4", 9, 3), (64, "rk4", 24, 5), (64, "euler", 9, 3), (32, "rk4", 9, 3), (128, "rk4", 9, 3)):
            n += case_b(H, method, B, Tn, *dims, seed=1234)
print("TOTAL BAD", n)
