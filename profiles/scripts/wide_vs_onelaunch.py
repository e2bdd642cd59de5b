"""Backward at hidden 64: the one-launch kernels (K4 / K7) against the adjoint sweep + GEMMs (K4w / K7w), same inputs, same call."""
import sys, time, torch, torch.nn as nn
sys.path.insert(0, ".")
from py_psnode_amd import fused

def mk(dims):
    return [(l.weight.detach().cuda(), l.bias.detach().cuda()) for l in [nn.Linear(dims[k], dims[k + 1]) for k in range(len(dims) - 1)]]

def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

torch.manual_seed(0)
B, T, H = 4096, 1001, 64
for method in ("rk4", "euler"):
    xd, zd = 8, 2
    n = xd + zd
    de = mk([3 * n, H, H, H, xd])
    t = (torch.arange(T, dtype=torch.float32) * 0.01).view(T, 1, 1).repeat(1, B, 1).cuda()
    x = torch.zeros(T, B, xd, device="cuda"); x[0] = 0.1 * torch.randn(B, xd, device="cuda")
    z = 0.1 * torch.randn(T, B, zd, device="cuda")
    a0 = torch.cat((x[0], z[0]), -1)
    xs = fused.ode_integrate(method, de, t, x, z, a0)
    G = torch.randn(T, B, xd, device="cuda")
    for kern in ("mfma", "wide"):
        ms = timeit(lambda: fused.ode_backward(method, de, t, z, a0, xs, G, kernel=kern))
        print(f"ode01 {method} hidden 64 backward kernel={kern}: {ms:.2f} ms")
    xd, zd, vd, idim = 8, 2, 2, 2
    n = xd + zd + vd + idim
    de, ae = mk([3 * n, H, H, H, xd]), mk([n + xd + zd + vd, H, H, H, idim])
    z, v = 0.1 * torch.randn(T, B, zd, device="cuda"), 0.1 * torch.randn(T, B, vd, device="cuda")
    xi, i0 = 0.1 * torch.randn(B, xd, device="cuda"), 0.1 * torch.randn(B, idim, device="cuda")
    a0 = torch.cat((xi, z[0], v[0], i0), -1)
    xe, ie = torch.zeros(T, B, 0, device="cuda"), torch.zeros(T, B, idim, device="cuda")
    xs, is_ = fused.dae_integrate(method, de, ae, xi, t, xe, z, v, ie, a0)
    Gx, Gi = torch.randn(T, B, xd, device="cuda"), torch.randn(T, B, idim, device="cuda")
    for kern in ("mfma", "wide"):
        ms = timeit(lambda: fused.dae_backward(method, de, ae, t, z, v, a0, xs, is_, Gx, Gi, kernel=kern))
        print(f"dae01 {method} hidden 64 backward kernel={kern}: {ms:.2f} ms")
