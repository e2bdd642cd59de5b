"""Out-of-bounds probe for the fused ODE_02 forward (K3f): every input in turn is placed so that it ENDS exactly at the end of a 32 MB device
allocation (whatever lies behind is unmapped or foreign), then the kernel runs.  A read or write past the end faults."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from py_psnode_amd import fused
import torch.nn as nn
dev = torch.device("cuda", 0)
def at_end(t):
    big = torch.empty(8 * 1024 * 1024, dtype=t.dtype, device=dev)      # 32 MB: its own hipMalloc block
    v = big[big.numel() - t.numel():].view(t.shape)
    v.copy_(t)
    return v, big
def mlp2(i, o):
    a, b = nn.Linear(i, 16), nn.Linear(16, o)
    return [(a.weight.detach().to(dev), a.bias.detach().to(dev)), (b.weight.detach().to(dev), b.bias.detach().to(dev))]
torch.manual_seed(0)
for (B, T, xd, zd, events) in [(32, 41, 8, 2, False), (32, 41, 8, 2, True), (37, 23, 8, 2, False), (5, 150, 5, 3, False), (64, 16, 16, 16, False), (1, 2, 1, 1, False), (130, 48, 8, 2, True)]:
    xe, ze, xdec, de = mlp2(xd, 16), mlp2(zd, 16), mlp2(16, xd), mlp2(96, 16)
    t = (torch.arange(T, dtype=torch.float32) * 0.01).view(1, T, 1).repeat(B, 1, 1).to(dev)
    x, z = (0.1 * torch.randn(B, T, xd)).to(dev), (0.1 * torch.randn(B, T, zd)).to(dev)
    ev = zj = None
    if events and T > 4:
        ev = t[:, [1, T - 2], :].contiguous(); zj = (0.1 * torch.randn(B, 2, zd)).to(dev)
    else:
        ev = -torch.ones(B, 2, 1, device=dev); zj = torch.zeros(B, 2, zd, device=dev)
    for which in ("t", "x", "z", "ev", "zj", "none"):
        tt, xx, zz, ee, jj = t, x, z, ev, zj
        hold = None
        if which == "t": tt, hold = at_end(t)
        if which == "x": xx, hold = at_end(x)
        if which == "z": zz, hold = at_end(z)
        if which == "ev": ee, hold = at_end(ev)
        if which == "zj": jj, hold = at_end(zj)
        for method in ("euler", "rk4"):
            for kw in ({}, {"want_latent": True}, {"want_recon": False}):
                out = fused.ode_encoded_integrate(method, xe, ze, xdec, de, tt, xx, zz, event_t=ee, z_jump=jj, **kw)
                torch.cuda.synchronize()
        print("ok", (B, T, xd, zd, events), which, flush=True)
print("probe done")
