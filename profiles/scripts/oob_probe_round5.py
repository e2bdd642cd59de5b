"""Out-of-bounds probe for the round-5 kernels: every device input in turn ENDS exactly at the end of its own 32 MB allocation, then the kernel
runs (ragged batches, grids that end inside a block): K1x / K2x forward (+ saving forward and K4f), the hidden-16 latent forward / backward
(FAST, two-role K8f), the row kernels on a time-major view (4 x 4 tiles), K3r forward / backward."""
import os, sys, torch
import torch.nn as nn
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from py_psnode_amd import fused
dev = torch.device("cuda", 0)
def at_end(t):
    big = torch.empty(8 * 1024 * 1024, dtype=t.dtype, device=dev)
    v = big[big.numel() - t.numel():].view(t.shape)
    v.copy_(t)
    return v, big
def lin(dims):
    ls = [nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:])]
    return [(m.weight.detach().to(dev), m.bias.detach().to(dev)) for m in ls]
r = lambda *s: (0.1 * torch.randn(*s)).to(dev)
def each_at_end(names, tensors, fn, tag):
    for k, nme in enumerate(names + ["none"]):
        args = list(tensors); hold = None
        if nme != "none":
            args[k], hold = at_end(tensors[k])
        fn(*args); torch.cuda.synchronize()
    print("ok", tag, flush=True)
torch.manual_seed(0)
# ---- K1x / saving forward / K4f, K2x
for (B, T, xd, zd, H) in [(37, 23, 8, 2, 64), (5, 11, 5, 3, 40), (130, 9, 8, 2, 64), (3, 70, 7, 8, 20)]:
    ls = lin([3 * (xd + zd), H, H, H, xd])
    t = (torch.arange(T, dtype=torch.float32) * 0.01).view(T, 1, 1).repeat(1, B, 1).to(dev)
    x, z = r(T, B, xd), r(T, B, zd)
    a0 = torch.cat((x[0], z[0]), -1).contiguous()
    G = r(T, B, xd)
    for method in ("euler", "rk4"):
        each_at_end(["t", "x", "z", "a0"], [t, x, z, a0], lambda t_, x_, z_, a_: fused.ode_integrate(method, ls, t_, x_, z_, a_), ("K1x", B, T, xd, zd, H, method))
        def train(t_, x_, z_, a_, g_):
            xs, saved = fused.ode_integrate(method, ls, t_, x_, z_, a_, save=True)
            fused.ode_backward(method, ls, t_, z_, a_, xs, g_, saved=saved)
        each_at_end(["t", "x", "z", "a0", "G"], [t, x, z, a0, G], train, ("K1x save + K4f", B, T, method))
for (B, T, xd, zd, vd, idim, H) in [(37, 23, 8, 2, 2, 2, 64), (5, 11, 5, 1, 2, 3, 40)]:
    n = xd + zd + vd + idim
    de, ae = lin([3 * n, H, H, H, xd]), lin([n + xd + zd + vd, H, H, H, idim])
    t = (torch.arange(T, dtype=torch.float32) * 0.01).view(T, 1, 1).repeat(1, B, 1).to(dev)
    x, z, v, i = r(T, B, xd), r(T, B, zd), r(T, B, vd), r(T, B, idim)
    xi = x[0].contiguous()
    a0 = torch.cat((x[0], z[0], v[0], i[0]), -1).contiguous()
    for method in ("euler", "rk4"):
        each_at_end(["xi", "t", "z", "v", "a0"], [xi, t, z, v, a0], lambda xi_, t_, z_, v_, a_: fused.dae_integrate(method, de, ae, xi_, t_, x, z_, v_, i, a_), ("K2x", B, T, method))
# ---- hidden-16 latent forward / backward
for (B, T) in [(37, 23), (5, 11), (130, 64), (3, 150)]:
    H = 16
    ls = lin([6 * H, H, H])
    t = (torch.arange(T, dtype=torch.float32) * 0.01).view(T, 1, 1).repeat(1, B, 1).to(dev)
    x, z = r(T, B, H), r(T, B, H)
    a0 = torch.cat((x[0], z[0]), -1).contiguous()
    G = r(T, B, H)
    for method in ("euler", "rk4"):
        def lat(t_, x_, z_, a_, g_):
            xs = fused.ode_integrate(method, ls, t_, x_, z_, a_)
            fused.ode_backward(method, ls, t_, z_, a_, xs, g_)
        each_at_end(["t", "x", "z", "a0", "G"], [t, x, z, a0, G], lat, ("latent16 fwd + K8f", B, T, method))
# ---- row kernels on a time-major view, K3r
for (B, T, din, dout) in [(37, 23, 8, 8), (5, 11, 2, 2), (130, 9, 5, 3), (3, 70, 16, 16)]:
    enc = nn.Sequential(nn.Linear(din, 16), nn.ELU(), nn.Linear(16, 16)).to(dev)
    dec = nn.Sequential(nn.Linear(16, 16), nn.ELU(), nn.Linear(16, dout)).to(dev)
    le, ld = fused.sequential_layers(enc), fused.sequential_layers(dec)
    base, G16, Gd = r(B, T, din), r(T, B, 16), r(T, B, dout)
    def rows(b_, g16, gd):
        v = b_.permute(1, 0, 2)
        fused.mlp_rows(le, v); fused.mlp_rows_backward(le, v, g16)
        fused.recon_rows(le, ld, v); fused.recon_rows_backward(le, ld, v, gd)
    each_at_end(["x", "G16", "Gd"], [base, G16, Gd], rows, ("rows / K3r", B, T, din, dout))
print("probe done")
