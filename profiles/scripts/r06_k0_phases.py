import os, sys, torch, torch.nn as nn
sys.path.insert(0, os.getcwd())
from py_psnode_amd import fused
dev = torch.device("cuda", 0)
B, T = 4096, 1001
def mk(dims, seed):
    torch.manual_seed(seed)
    return [(l.weight.detach().to(dev), l.bias.detach().to(dev)) for l in [nn.Linear(dims[k], dims[k + 1]) for k in range(len(dims) - 1)]]
xd, zd = 8, 2
n = xd + zd
ls = mk([3 * n, 64, 64, 64, xd], 1)
g = torch.Generator().manual_seed(2)
t = (torch.arange(T, dtype=torch.float32) * 0.01).view(T, 1, 1).repeat(1, B, 1).to(dev)
x = torch.zeros(T, B, xd); x[0] = 0.1 * torch.randn(B, xd, generator=g); x = x.to(dev)
z = (0.1 * torch.randn(T, B, zd, generator=g)).to(dev)
a0 = torch.cat((x[0], z[0]), -1).contiguous()
for m in ("rk4", "euler"):
    fused.ode_integrate(m, ls, t, x, z, a0, kernel="generic"); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fused.ode_integrate(m, ls, t, x, z, a0, kernel="generic"); e1.record(); torch.cuda.synchronize()
    print(m, e0.elapsed_time(e1), "ms", flush=True)
