"""Time the integrators on shapes outside the specialised MFMA classes (x_dim > 16, z+v+i > 8, depth != 3, mixed widths): the generic
kernel K0.  B = 4096 trajectories x 1000 steps, median of `reps` launches; prints one line per shape and the same model shape on the
specialised kernel where one exists (for scale).  Usage: python profiles/scripts/r06_generic_time.py [reps]"""
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from py_psnode_amd import fused  # noqa: E402

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B, T = int(os.environ.get("K0_BATCH", "4096")), int(os.environ.get("K0_GRID", "1001"))
dev = torch.device("cuda", 0)


def mk(dims, seed):
    torch.manual_seed(seed)
    return [(l.weight.detach().to(dev), l.bias.detach().to(dev)) for l in [nn.Linear(dims[k], dims[k + 1]) for k in range(len(dims) - 1)]]


def med(fn):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(REPS):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


def ode(xd, zd, hidden, method, kernel="auto"):
    n = xd + zd
    ls = mk([3 * n] + list(hidden) + [xd], 1)
    g = torch.Generator().manual_seed(2)
    t = (torch.arange(T, dtype=torch.float32) * 0.01).view(T, 1, 1).repeat(1, B, 1).to(dev)
    x = torch.zeros(T, B, xd); x[0] = 0.1 * torch.randn(B, xd, generator=g); x = x.to(dev)
    z = (0.1 * torch.randn(T, B, zd, generator=g)).to(dev)
    a0 = torch.cat((x[0], z[0]), -1).contiguous()
    ms = med(lambda: fused.ode_integrate(method, ls, t, x, z, a0, kernel=kernel))
    flop = 2 * sum(a * b for a, b in zip([3 * n] + list(hidden), list(hidden) + [xd]))
    S = {"euler": 1, "midpoint": 2, "rk4": 4}[method]
    print(f"ODE x{xd} z{zd} hidden {list(hidden)} {method} kernel={kernel}: {ms:8.2f} ms   {flop * S * B * (T - 1) / ms / 1e9:7.2f} TFLOP/s dense", flush=True)


def dae(xd, zd, vd, idim, H, method, kernel="auto"):
    n = xd + zd + vd + idim
    de, ae = mk([3 * n, H, H, H, xd], 3), mk([n + xd + zd + vd, H, H, H, idim], 4)
    g = torch.Generator().manual_seed(5)
    t = (torch.arange(T, dtype=torch.float32) * 0.01).view(T, 1, 1).repeat(1, B, 1).to(dev)
    r = lambda *s: (0.1 * torch.randn(*s, generator=g)).to(dev)
    x, z, v, i, xi = r(T, B, xd), r(T, B, zd), r(T, B, vd), r(T, B, idim), r(B, xd)
    a0 = torch.cat((xi, z[0], v[0], i[0]), -1).contiguous()
    ms = med(lambda: fused.dae_integrate(method, de, ae, xi, t, x, z, v, i, a0, kernel=kernel))
    S = {"euler": 1, "midpoint": 2, "rk4": 4}[method]
    fde = 2 * (3 * n * H + 2 * H * H + H * xd)
    fae = 2 * ((n + xd + zd + vd) * H + 2 * H * H + H * idim)
    print(f"DAE x{xd} z{zd} v{vd} i{idim} hidden {H} {method} kernel={kernel}: {ms:8.2f} ms   {(fde * S + fae) * B * (T - 1) / ms / 1e9:7.2f} TFLOP/s dense",
          flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "short":
        ode(8, 2, (64, 64, 64), "rk4", "generic")
        ode(8, 2, (64, 64, 64), "euler", "generic")
        ode(20, 2, (128, 128, 128), "rk4")
        ode(8, 2, (320, 320, 320), "euler")
        dae(8, 4, 6, 6, 64, "rk4")
        dae(20, 10, 40, 40, 128, "rk4")
        sys.exit(0)
    ode(8, 2, (64, 64, 64), "rk4", "auto")
    ode(8, 2, (64, 64, 64), "rk4", "generic")
    ode(16, 2, (64, 64, 64), "rk4", "auto")
    ode(20, 2, (64, 64, 64), "rk4")
    ode(32, 4, (64, 64, 64), "rk4")
    ode(8, 2, (64, 64), "rk4")
    ode(8, 2, (64, 64, 64, 64), "rk4")
    ode(8, 2, (64, 64, 64, 64), "euler")
    ode(8, 2, (128, 64, 32), "rk4")
    ode(20, 2, (128, 128, 128), "rk4")
    ode(8, 2, (320, 320, 320), "euler")
    dae(8, 2, 2, 2, 64, "rk4", "auto")
    dae(8, 2, 2, 2, 64, "rk4", "generic")
    dae(8, 4, 6, 6, 64, "rk4")
    dae(20, 10, 40, 40, 64, "euler")
    dae(20, 10, 40, 40, 128, "rk4")
