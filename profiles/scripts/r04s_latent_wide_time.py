"""K3w timing at the headline batch: the latent integrators of ODE_02 / DAE_02 at hidden 128 / 32 (B=4096 x 1000 steps), K3w vs K0."""
import sys, time
import torch, torch.nn as nn
sys.path.insert(0, ".")
from py_psnode_amd import fused
B, T = 4096, 1001
mk = lambda dims: [(l.weight.detach().cuda(), l.bias.detach().cuda()) for l in [nn.Linear(dims[k], dims[k + 1]) for k in range(len(dims) - 1)]]
r = lambda *s: 0.1 * torch.randn(*s, device="cuda")
t = (torch.arange(T, dtype=torch.float32, device="cuda") * 0.01).view(T, 1, 1).repeat(1, B, 1)
def timeit(f, n=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for H in (128, 32):
    for method in ("rk4", "euler"):
        torch.manual_seed(0)
        de = mk([6 * H, H, H]); x, z = r(T, B, H), r(T, B, H); a0 = torch.cat((x[0], z[0]), -1)
        S = 4 if method == "rk4" else 1
        fl = 2 * (S * 2 * H * H + H * H) * B * (T - 1)
        for k in ("mfma", "generic"):
            try:
                ms = timeit(lambda: fused.ode_integrate(method, de, t, x, z, a0, kernel=k), 2 if k == "generic" else 5)
                print(f"latent ODE H{H} {method} {k:8s} {ms:9.2f} ms  executed-flop frac {fl / ms / 1e9 / 157.3:.3f}")
            except Exception as e:
                print(f"latent ODE H{H} {method} {k:8s} {type(e).__name__}")
        de, ae = mk([12 * H, H, H]), mk([7 * H, H, H])
        v, i, xi = r(T, B, H), r(T, B, H), r(B, H); a0 = torch.cat((xi, z[0], v[0], i[0]), -1)
        fl = 2 * (S * 2 * H * H + 3 * H * H + 4 * H * H) * B * (T - 1)
        for k in ("mfma", "generic"):
            try:
                ms = timeit(lambda: fused.dae_integrate(method, de, ae, xi, t, x, z, v, i, a0, kernel=k), 1 if k == "generic" else 5)
                print(f"latent DAE H{H} {method} {k:8s} {ms:9.2f} ms  executed-flop frac {fl / ms / 1e9 / 157.3:.3f}")
            except Exception as e:
                print(f"latent DAE H{H} {method} {k:8s} {type(e).__name__}")
