"""Soak of the round-5 two-role kernels at the bench size: the ODE_02 model forward (K3f two-role) N times and its training step (K8f two-role,
one autograd node per module) M times on the same inputs -- every repetition must reproduce the first one bit for bit (a race between the
roles' LDS rings and barriers would show as a difference)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from py_psnode_amd import loss as L, models
from py_psnode_amd import neural_dae as nd
N, M = int((sys.argv + ["60"])[1]), int((sys.argv + ["60", "30"])[2])
dev = torch.device("cuda", 0)
for method, Tn, B in (("rk4", 1001, 4096), ("euler", 1001, 4096), ("rk4", 333, 4093), ("midpoint", 50, 130)):
    g = torch.Generator().manual_seed(1)
    r = lambda *s: (0.1 * torch.randn(*s, generator=g)).to(dev)
    t = (torch.arange(Tn, dtype=torch.float32) * 0.01).view(1, Tn, 1).repeat(B, 1, 1).to(dev)
    x, z = r(B, Tn, 8), r(B, Tn, 2)
    ev, zj = -torch.ones(B, 2, 1, device=dev), torch.zeros(B, 2, 2, device=dev)
    mask = torch.ones(B, Tn, 8, device=dev)
    torch.manual_seed(0)
    m = models.ODE_Model(8, 2, 16, direct_encode=True, solver={"rk4": nd.RK4, "euler": nd.Euler, "midpoint": nd.Midpoint}[method]()).to(dev)
    m.solver.fused = "require"
    with torch.no_grad():
        ref = [o.clone() for o in m(t=t, x=x, z=z, event_t=ev, z_jump=zj)]
        bad = 0
        for _ in range(N):
            out = m(t=t, x=x, z=z, event_t=ev, z_jump=zj)
            bad += sum(int(not torch.equal(a, b)) for a, b in zip(out, ref))
    def step():
        m.zero_grad(set_to_none=True)
        o = m(t=t, x=x, z=z, event_t=ev, z_jump=zj)
        L.ode02_loss(o[0], o[1], x, mask)[0].backward()
        return [p.grad.clone() for p in m.parameters()]
    g0 = step()
    badg = 0
    for _ in range(M):
        badg += sum(int(not torch.equal(a, b)) for a, b in zip(step(), g0))
    fin = all(bool(torch.isfinite(q).all()) for q in ref + g0)
    print(f"{method} B={B} T={Tn}: forward repeats differing {bad}/{N}, gradient repeats differing {badg}/{M}, finite {fin}")
