"""Per-launch time of the first 40 launches of the headline call in a fresh process (is there a start-up transient, how long?).
usage: launch_transient.py [auto|tile]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from py_psnode_amd import fused
kern = (sys.argv + ["auto"])[1]
dev = torch.device("cuda", 0)
B, T, xd, zd, H = 4096, 1001, 8, 2, 64
g = torch.Generator().manual_seed(0)
r = lambda *s: (0.1 * torch.randn(*s, generator=g)).to(dev)
dims = [3 * (xd + zd), H, H, H, xd]
layers = [(r(dims[k + 1], dims[k]), r(dims[k + 1])) for k in range(4)]
t = (torch.arange(T, dtype=torch.float32) * 0.01).view(T, 1, 1).repeat(1, B, 1).to(dev)
x, z = r(T, B, xd), r(T, B, zd)
a0 = torch.cat((x[0], z[0]), -1)
torch.cuda.synchronize()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
for a, b in evs:
    a.record(); fused.ode_integrate("rk4", layers, t, x, z, a0, kernel=kern); b.record()
torch.cuda.synchronize()
print(kern, " ".join("%.2f" % a.elapsed_time(b) for a, b in evs))
