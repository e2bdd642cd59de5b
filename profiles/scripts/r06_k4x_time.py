"""Round 6: the ODE backward alone -- K4x (kernel="wave": one wave per 4 trajectories, no LDS) against K4f (kernel="wide": the two-role 4-wave
tile) from the SAME saved rows, B=4096 x 1000 steps, hidden 64, HIP events around each call.   usage: r06_k4x_time.py [methods] [reps]
PSNODE_LIB_PATH selects a library variant (ablation builds print timings of WRONG results: max_diff is then meaningless)."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torch.nn as nn
from py_psnode_amd import fused
methods = sys.argv[1].split(",") if len(sys.argv) > 1 else ["rk4", "euler"]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B, Tn, H, xd, zd = int(os.environ.get("K4X_B", 4096)), 1001, 64, 8, 2
torch.manual_seed(0)
de = [(l.weight.detach().cuda(), l.bias.detach().cuda()) for l in [nn.Linear(a, b) for a, b in zip([3 * (xd + zd), H, H, H], [H, H, H, xd])]]
r = lambda *s: 0.1 * torch.randn(*s, device="cuda")
t = (torch.arange(Tn, dtype=torch.float32, device="cuda") * 0.01).view(Tn, 1, 1).repeat(1, B, 1)
x = torch.zeros(Tn, B, xd, device="cuda"); x[0] = r(B, xd)
z = r(Tn, B, zd); a0 = torch.cat((x[0], z[0]), -1)
G = torch.randn(Tn, B, xd, device="cuda")
ev = torch.full((B, 2, 1), -1.0, device="cuda"); zj = torch.zeros(B, 2, zd, device="cuda")
tab = fused.event_table(t, ev)
for m in methods:
    xs, saved = fused.ode_integrate(m, de, t, x, z, a0, event_t=ev, z_jump=zj, save=True)
    out = {}
    for kern in ("wave", "wide"):
        for need_z in (False,):
            kw = dict(event_idx=tab, z_jump=zj, saved=saved, need_grad_z=need_z, need_grad_zj=os.environ.get("K4X_GZJ", "0") == "1", kernel=kern)
            for _ in range(3): g = fused.ode_backward(m, de, t, z, a0, xs, G, **kw)
            torch.cuda.synchronize()
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            for a, b in evs:
                a.record(); g = fused.ode_backward(m, de, t, z, a0, xs, G, **kw); b.record()
            torch.cuda.synchronize()
            ms = sorted(a.elapsed_time(b) for a, b in evs)
            out[kern] = g
            print(f"{m:6s} B={B} backward {kern:5s}: median {ms[len(ms)//2]:.3f} ms  min {ms[0]:.3f}  max {ms[-1]:.3f}", flush=True)
    d = max(float((p - q).abs().max() / q.abs().max().clamp_min(1e-30)) for p, q in zip(out["wave"][4], out["wide"][4]))
    print(f"{m:6s} max param-gradient difference wave vs wide (relative to each tensor's max): {d:.2e}", flush=True)
    del xs, saved, out, g
    torch.cuda.empty_cache()
