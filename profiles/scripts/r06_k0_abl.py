"""Times the generic kernel K0 on ODE x8 z2 hidden 64 (4096 x 1000) for the ablation builds of psnode_generic.hip (PSNODE_LIB_PATH)."""
import os, sys, torch, torch.nn as nn
sys.path.insert(0, os.getcwd())
from py_psnode_amd import fused
dev = torch.device("cuda", 0)
B, T = 4096, 1001
torch.manual_seed(1)
dims = [30, 64, 64, 64, 8]
ls = [(l.weight.detach().to(dev), l.bias.detach().to(dev)) for l in [nn.Linear(dims[k], dims[k + 1]) for k in range(4)]]
g = torch.Generator().manual_seed(2)
t = (torch.arange(T, dtype=torch.float32) * 0.01).view(T, 1, 1).repeat(1, B, 1).to(dev)
x = torch.zeros(T, B, 8); x[0] = 0.1 * torch.randn(B, 8, generator=g); x = x.to(dev)
z = (0.1 * torch.randn(T, B, 2, generator=g)).to(dev)
a0 = torch.cat((x[0], z[0]), -1).contiguous()
for m in ("rk4", "euler"):
    ts = []
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fused.ode_integrate(m, ls, t, x, z, a0, kernel="generic"); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(os.environ.get("PSNODE_LIB_PATH", "tree"), m, "%.2f ms" % sorted(ts)[1], flush=True)
