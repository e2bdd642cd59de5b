"""Times the generic backward K5 (and the forward K0 it follows) on shapes outside the specialised classes: B = 4096 x 1000 steps."""
import os, sys, torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from py_psnode_amd import fused
dev = torch.device("cuda", 0)
B, T = 4096, 1001
def mk(dims, seed):
    torch.manual_seed(seed)
    return [(l.weight.detach().to(dev), l.bias.detach().to(dev)) for l in [nn.Linear(dims[k], dims[k + 1]) for k in range(len(dims) - 1)]]
def ev_time(fn, reps=3):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]
for xd, zd, hidden, method in [(8, 2, (64, 64, 64), "rk4"), (20, 2, (64, 64, 64), "rk4"), (20, 2, (64, 64, 64), "euler"), (20, 2, (128, 128, 128), "euler")]:
    n = xd + zd
    ls = mk([3 * n] + list(hidden) + [xd], 1)
    g = torch.Generator().manual_seed(2)
    t = (torch.arange(T, dtype=torch.float32) * 0.01).view(T, 1, 1).repeat(1, B, 1).to(dev)
    x = torch.zeros(T, B, xd); x[0] = 0.1 * torch.randn(B, xd, generator=g); x = x.to(dev)
    z = (0.1 * torch.randn(T, B, zd, generator=g)).to(dev)
    a0 = torch.cat((x[0], z[0]), -1).contiguous()
    xs = fused.ode_integrate(method, ls, t, x, z, a0, kernel="generic")
    G = torch.randn_like(xs)
    f = ev_time(lambda: fused.ode_integrate(method, ls, t, x, z, a0, kernel="generic"))
    b = ev_time(lambda: fused.ode_backward(method, ls, t, z, a0, xs, G, kernel="generic"))
    print(f"ODE x{xd} z{zd} hidden {list(hidden)} {method}: forward K0 {f:.2f} ms, backward K5 {b:.2f} ms", flush=True)
for xd, zd, vd, idim, H, method in [(8, 2, 2, 2, 64, "rk4"), (8, 4, 6, 6, 64, "rk4"), (8, 4, 6, 6, 64, "euler")]:
    n = xd + zd + vd + idim
    de, ae = mk([3 * n, H, H, H, xd], 3), mk([n + xd + zd + vd, H, H, H, idim], 4)
    g = torch.Generator().manual_seed(5)
    t = (torch.arange(T, dtype=torch.float32) * 0.01).view(T, 1, 1).repeat(1, B, 1).to(dev)
    r = lambda *s: (0.1 * torch.randn(*s, generator=g)).to(dev)
    z, v, xi, i0 = r(T, B, zd), r(T, B, vd), r(B, xd), r(B, idim)
    a0 = torch.cat((xi, z[0], v[0], i0), -1).contiguous()
    xe, ie = torch.zeros(T, B, 0, device=dev), torch.zeros(T, B, idim, device=dev)
    xs, is_ = fused.dae_integrate(method, de, ae, xi, t, xe, z, v, ie, a0, kernel="generic")
    Gx, Gi = torch.randn_like(xs), torch.randn_like(is_)
    f = ev_time(lambda: fused.dae_integrate(method, de, ae, xi, t, xe, z, v, ie, a0, kernel="generic"))
    b = ev_time(lambda: fused.dae_backward(method, de, ae, t, z, v, a0, xs, is_, Gx, Gi, kernel="generic"))
    print(f"DAE x{xd} z{zd} v{vd} i{idim} hidden {H} {method}: forward K0 {f:.2f} ms, backward K5 {b:.2f} ms", flush=True)
