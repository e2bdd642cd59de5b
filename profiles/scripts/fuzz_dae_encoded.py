"""Random-shape fuzz of K3g (psnode_dae_encoded_integrate_f32: the DAE_02 model forward at hidden 64 in one launch) against the
row kernels + K3c on the same inputs: x_dim <= 16, z | v | i <= 8 wide (z absent at random), events, ragged clocks and tiles, with and
without the reconstructions.  usage (GPU box, repo root): python profiles/scripts/fuzz_dae_encoded.py [seed] [iterations]"""
import random
import sys

import torch

sys.path.insert(0, ".")
from py_psnode_amd import models  # noqa: E402
from py_psnode_amd import neural_dae as nd  # noqa: E402

random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 99)
bad = 0


def close(a, b, what, tag):
    global bad
    sc = float(b.abs().amax()); er = float((a - b).abs().amax())
    if not er <= 2e-5 * max(sc, 1e-3):
        bad += 1
        print("MISMATCH", tag, what, f"err {er:.2e} scale {sc:.2e}")


for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    torch.manual_seed(1000 + it)
    method = random.choice(["euler", "midpoint", "rk4"])
    xd, zd, vd, idim = random.randint(1, 16), random.choice([0, 0, 1, 2, 5, 8]), random.randint(1, 8), random.randint(1, 8)
    B, Tn = random.randint(1, 70), random.randint(1, 14)
    m = models.DAE_Model(xd, zd, vd, idim, 64, direct_encode=True, solver={"euler": nd.Euler, "midpoint": nd.Midpoint, "rk4": nd.RK4}[method]()).cuda()
    m.solver.fused = "require"
    r = lambda *s: 0.3 * torch.randn(*s, device="cuda")
    t = (torch.arange(Tn, dtype=torch.float32, device="cuda") * 0.02).view(1, Tn, 1).repeat(B, 1, 1)
    if B > 1:
        t[1:] = t[1:] * (0.5 + torch.rand(B - 1, 1, 1, device="cuda"))
    events = Tn > 4 and random.random() < 0.6
    ev = t[:, [1, Tn - 2], :].contiguous() if events else torch.full((B, 2, 1), -1.0, device="cuda")
    kw = dict(t=t, x=r(B, Tn, xd), z=r(B, Tn, zd), v=r(B, Tn, vd), i=r(B, Tn, idim), event_t=ev, z_jump=r(B, 2, zd), v_jump=r(B, 2, vd))
    tag = (method, B, Tn, xd, zd, vd, idim, events)
    with torch.no_grad():
        m.one_launch = True
        one = m(**kw)
        m.one_launch = False
        ref = m(**kw)
    for k, name in enumerate(("x_pred", "i_pred", "x_re", "i_re")):
        close(one[k], ref[k], name, tag)
print("fuzz done, mismatches:", bad)
