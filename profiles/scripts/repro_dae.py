import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_gpu_backward as tb
from py_psnode_amd import fused
def run(H, method, B, Tn, xd, zd, vd, idim, seed=1234):
    de, ae, t, z, v, xi, a0, ev, zj, vj, Gx, Gi = tb._dae_raw_case(B, Tn, xd, zd, vd, idim, seed, False, H=H)
    xe, ie = torch.zeros(Tn, B, 0, device="cuda"), torch.zeros(Tn, B, idim, device="cuda")
    xs, is_, saved = fused.dae_integrate(method, de, ae, xi, t, xe, z, v, ie, a0, save=True)
    e = lambda p, q: float((p - q).abs().max()) / max(float(q.abs().max()), 1e-9)
    for name, gi in (("None", None), ("zeros", torch.zeros_like(Gi)), ("given", Gi)):
        b = fused.dae_backward(method, de, ae, t, z, v, a0, xs, is_, Gx, gi, kernel="generic")
        c = fused.dae_backward_wide(method, de, ae, t, z, v, a0, xs, is_, Gx, gi)
        d = fused.dae_backward_wide(method, de, ae, t, z, v, a0, xs, is_, Gx, gi, saved=saved)
        print(f"H{H} {method} B{B} T{Tn} dims {xd},{zd},{vd},{idim} gi={name}: K7f", [f"{e(p, q):.0e}" for p, q in zip(c["ae"], b["ae"])], " saved", [f"{e(p, q):.0e}" for p, q in zip(d["ae"], b["ae"])], flush=True)
run(64, "rk4", 9, 3, 4, 2, 0, 2)
run(64, "rk4", 9, 3, 4, 1, 1, 2)
run(64, "rk4", 24, 5, 8, 2, 2, 2)
run(64, "euler", 9, 3, 4, 2, 0, 2)
run(32, "rk4", 9, 3, 4, 2, 0, 2)
run(128, "rk4", 9, 3, 4, 2, 0, 2)
