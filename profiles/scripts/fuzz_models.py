"""Random-shape fuzz of the direct_encode models (ODE_02 / DAE_02 at hidden 16 / 64, with and without z, events, ragged tiles) and of the
no_encode models at random hidden widths: gradients of the default training route (forward saves its activations where the kernels can)
against the recompute route (PSNODE_SAVE_ACTIVATIONS=0, pinned to the reference gradients by tests/test_grad_goldens.py), same fp32 inputs.
usage (GPU box, repo root): python profiles/scripts/fuzz_models.py [seed] [iterations]"""
import os
import random
import sys

import torch

sys.path.insert(0, ".")
from py_psnode_amd import autograd as pag, models, neural_dae as nd  # noqa: E402

random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    tag = random.choice(["ode02", "dae02", "ode01", "dae01"])
    H = random.choice([16, 64]) if tag.endswith("02") else random.choice([8, 32, 40, 64, 100, 128])
    zd = random.choice([0, 2]) if tag == "dae02" else random.choice([1, 2, 4])
    method = random.choice(["euler", "midpoint", "rk4"])
    B, T = random.randint(1, int(os.environ.get("FUZZ_BMAX", "40"))), random.randint(1, int(os.environ.get("FUZZ_TMAX", "12")))
    events = T > 7 and random.random() < 0.5
    torch.manual_seed(it)
    g = torch.Generator().manual_seed(1000 + it)
    r = lambda *s: 0.1 * torch.randn(*s, generator=g)
    t = (torch.arange(T, dtype=torch.float32) * 0.01).view(1, T, 1).repeat(B, 1, 1)
    x, z, v, i = r(B, T, 8), r(B, T, zd), r(B, T, 2), r(B, T, 2)
    ev = t[:, [2, 6], :].contiguous() if events else -torch.ones(B, 2, 1)
    zj, vj = r(B, 2, zd), r(B, 2, 2)
    cls = {"euler": nd.Euler, "midpoint": nd.Midpoint, "rk4": nd.RK4}[method]
    if tag == "ode02":
        m = models.ODE_Model(8, zd, H, direct_encode=True, solver=cls())
    elif tag == "dae02":
        m = models.DAE_Model(8, zd, 2, 2, H, direct_encode=True, solver=cls())
    elif tag == "ode01":
        m = models.ODE_Model(8, zd, H, solver=cls())
    else:
        m = models.DAE_Model(8, zd, 2, 2, H, solver=cls())
    m = m.cuda()
    m.solver.fused = "require"
    c = lambda a: a.cuda()

    def run(mode):
        pag.SAVE_ACTIVATIONS = mode
        m.zero_grad()
        if tag.startswith("ode"):
            outs = m(t=c(t), x=c(x), z=c(z), event_t=c(ev), z_jump=c(zj))
        else:
            outs = m(t=c(t), x=c(x), z=c(z), v=c(v), i=c(i), event_t=c(ev), z_jump=c(zj), v_jump=c(vj))
        outs = outs if isinstance(outs, tuple) else (outs,)
        sum(((o - 0.05) ** 2).sum() for o in outs).backward()
        torch.cuda.synchronize()
        return [o.detach().clone() for o in outs], [None if p.grad is None else p.grad.detach().clone() for p in m.parameters()]

    try:
        o1, g1 = run("1")
    except Exception as e:      # noqa: BLE001  (shapes whose forward cannot save raise UnsupportedShapeError under "1": fall back to auto)
        o1, g1 = run("auto")
    o0, g0 = run("0")
    info = (tag, H, zd, method, B, T, events)
    for k, (a, b) in enumerate(zip(o1, o0)):
        # (round 5: a training forward that saves runs its ELUs in the plain domain, one that does not -- like inference -- in the log2e-scaled
        #  domain of K1 / K1x / K2: the same function, different roundings)
        if not float((a - b).abs().max()) <= 1e-5 * max(float(b.abs().max()), 1e-6):
            bad += 1; print("OUTPUT DIFFERS", info, k, float((a - b).abs().max()))
    for (n, _), a, b in zip(m.named_parameters(), g1, g0):
        if (a is None) != (b is None):
            bad += 1; print("GRAD NONE-NESS", info, n); continue
        if a is None:
            continue
        sc, er = float(b.abs().max()), float((a - b).abs().max())
        if not er <= 2e-4 * max(sc, 1e-6):
            bad += 1; print("MISMATCH", info, n, f"err {er:.2e} scale {sc:.2e}")
print("model fuzz done, mismatches:", bad)
