"""Round 6: the DAE training forward alone -- K2x SAVE (kernel="wave") against K2 SAVE ("tile"), and the plain forwards, B=4096 x 1000 steps,
hidden 64, HIP events.   usage: r06_k2x_save_time.py [methods] [reps]"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torch.nn as nn
from py_psnode_amd import fused
methods = sys.argv[1].split(",") if len(sys.argv) > 1 else ["rk4", "euler"]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B, Tn, H, xd, zd, vd, idim = 4096, 1001, 64, 8, 2, 2, 2
n = xd + zd + vd + idim
torch.manual_seed(0)
mk = lambda dims: [(l.weight.detach().cuda(), l.bias.detach().cuda()) for l in [nn.Linear(dims[k], dims[k + 1]) for k in range(len(dims) - 1)]]
de, ae = mk([3 * n, H, H, H, xd]), mk([n + xd + zd + vd, H, H, H, idim])
r = lambda *s: 0.1 * torch.randn(*s, device="cuda")
t = (torch.arange(Tn, dtype=torch.float32, device="cuda") * 0.01).view(Tn, 1, 1).repeat(1, B, 1)
z, v, xi, i0 = r(Tn, B, zd), r(Tn, B, vd), r(B, xd), r(B, idim)
a0 = torch.cat((xi, z[0], v[0], i0), -1)
xe, ie = torch.zeros(Tn, B, 0, device="cuda"), torch.zeros(Tn, B, idim, device="cuda")
ev = torch.full((B, 2, 1), -1.0, device="cuda"); zj = torch.zeros(B, 2, zd, device="cuda"); vj = torch.zeros(B, 2, vd, device="cuda")
for m in methods:
    for kern in ("wave", "tile"):
        for save in (False, True):
            f = lambda: fused.dae_integrate(m, de, ae, xi, t, xe, z, v, ie, a0, event_t=ev, z_jump=zj, v_jump=vj, save=save, kernel=kern)
            for _ in range(3): o = f()
            torch.cuda.synchronize()
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            for a, b in evs:
                a.record(); o = f(); b.record()
            torch.cuda.synchronize()
            ms = sorted(a.elapsed_time(b) for a, b in evs)
            print(f"{m:6s} DAE forward {kern:4s} save={save!s:5s}: median {ms[len(ms)//2]:.3f} ms  min {ms[0]:.3f}  max {ms[-1]:.3f}", flush=True)
            del o
            torch.cuda.empty_cache()
