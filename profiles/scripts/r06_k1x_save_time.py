"""Round 6: the saving forward (K1x SAVE) alone, B=4096 x 1000 steps, hidden 64, HIP events; PSNODE_LIB_PATH selects an ablation build
(timings of WRONG saved rows).   usage: r06_k1x_save_time.py [methods] [reps]"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torch.nn as nn
from py_psnode_amd import fused
methods = sys.argv[1].split(",") if len(sys.argv) > 1 else ["rk4", "euler"]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B, Tn, H, xd, zd = 4096, 1001, 64, 8, 2
torch.manual_seed(0)
de = [(l.weight.detach().cuda(), l.bias.detach().cuda()) for l in [nn.Linear(a, b) for a, b in zip([3 * (xd + zd), H, H, H], [H, H, H, xd])]]
r = lambda *s: 0.1 * torch.randn(*s, device="cuda")
t = (torch.arange(Tn, dtype=torch.float32, device="cuda") * 0.01).view(Tn, 1, 1).repeat(1, B, 1)
x = torch.zeros(Tn, B, xd, device="cuda"); x[0] = r(B, xd)
z = r(Tn, B, zd); a0 = torch.cat((x[0], z[0]), -1)
ev = torch.full((B, 2, 1), -1.0, device="cuda"); zj = torch.zeros(B, 2, zd, device="cuda")
for m in methods:
    for save in (False, True):
        for _ in range(3): o = fused.ode_integrate(m, de, t, x, z, a0, event_t=ev, z_jump=zj, save=save)
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in evs:
            a.record(); o = fused.ode_integrate(m, de, t, x, z, a0, event_t=ev, z_jump=zj, save=save); b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in evs)
        print(f"{m:6s} forward save={save!s:5s}: median {ms[len(ms)//2]:.3f} ms  min {ms[0]:.3f}  max {ms[-1]:.3f}", flush=True)
        del o
        torch.cuda.empty_cache()
