"""Out-of-bounds probe, the OLDER kernels: the tile integrators K1 / K2 at every width class (32 ... 256, x_dim up to 16), teacher forcing, the generic
kernels, the direct_encode models at the other hidden widths (K3w / K9w), the one-launch DAE_02 forward (K3g), the loss kernel -- ragged batches,
every input in turn ending exactly at the end of its own 32 MB allocation."""
import os, sys, torch
import torch.nn as nn
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from py_psnode_amd import fused, loss as L, models
from py_psnode_amd import neural_dae as nd
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
B, T = 37, 9
t = (torch.arange(T, dtype=torch.float32) * 0.01).view(T, 1, 1).repeat(1, B, 1).to(dev)
ev = t[[2, T - 3], :, :].permute(1, 0, 2).contiguous()
# ---- ODE integrators by width class / kernel / teacher forcing
for (H, xd, zd, kern) in [(32, 8, 2, "auto"), (64, 8, 2, "tile"), (64, 16, 2, "auto"), (100, 8, 2, "auto"), (128, 12, 4, "auto"), (192, 8, 2, "auto"), (256, 8, 2, "auto"),
                           (64, 8, 2, "generic"), (300, 8, 2, "auto")]:
    ls = lin([3 * (xd + zd), H, H, H, xd])
    x, z, zj = r(T, B, xd), r(T, B, zd), r(B, 2, zd)
    a0 = torch.cat((x[0], z[0]), -1).contiguous()
    for tf in (False, True):
        def f(t_, x_, z_, a_, e_, j_):
            try:
                fused.ode_integrate("rk4", ls, t_, x_, z_, a_, event_t=e_, z_jump=j_, kernel=kern, input_true_x=tf)
            except ValueError:
                pass
        each_at_end(["t", "x", "z", "a0", "ev", "zj"], [t, x, z, a0, ev, zj], f, ("ode_integrate", H, xd, zd, kern, "tf" if tf else ""))
# ---- DAE integrators
for (H, kern) in [(32, "auto"), (64, "tile"), (128, "auto"), (192, "auto"), (64, "generic")]:
    xd, zd, vd, idim = 8, 2, 2, 2
    n = xd + zd + vd + idim
    de, ae = lin([3 * n, H, H, H, xd]), lin([n + xd + zd + vd, H, H, H, idim])
    x, z, v, i, zj, vj = r(T, B, xd), r(T, B, zd), r(T, B, vd), r(T, B, idim), r(B, 2, zd), r(B, 2, vd)
    xi = x[0].contiguous()
    a0 = torch.cat((x[0], z[0], v[0], i[0]), -1).contiguous()
    for (tx, ti) in ((False, False), (True, True)):
        def f(xi_, t_, x_, z_, v_, i_, a_, e_, zj_, vj_):
            fused.dae_integrate("rk4", de, ae, xi_, t_, x_, z_, v_, i_, a_, event_t=e_, z_jump=zj_, v_jump=vj_, kernel=kern, input_true_x=tx, input_true_i=ti)
        each_at_end(["xi", "t", "x", "z", "v", "i", "a0", "ev", "zj", "vj"], [xi, t, x, z, v, i, a0, ev, zj, vj], f, ("dae_integrate", H, kern, tx, ti))
# ---- direct_encode models at the other hidden widths, DAE_02 in one launch, training steps
Bm, Tm = 21, 13
tb = (torch.arange(Tm, dtype=torch.float32) * 0.01).view(1, Tm, 1).repeat(Bm, 1, 1).to(dev)
xb, zb, vb, ib = r(Bm, Tm, 8), r(Bm, Tm, 2), r(Bm, Tm, 2), r(Bm, Tm, 2)
evb, zjb, vjb = tb[:, [2, Tm - 3], :].contiguous(), r(Bm, 2, 2), r(Bm, 2, 2)
mask = torch.ones(Bm, Tm, 1, device=dev)
for tag, H, one in (("ode02", 32, None), ("ode02", 128, None), ("ode02", 64, None), ("dae02", 32, None), ("dae02", 128, None), ("dae02", 16, None), ("dae02", 64, True)):
    if tag == "ode02":
        m = models.ODE_Model(8, 2, H, direct_encode=True, solver=nd.Euler()).to(dev)
        names, tens = ["t", "x", "z", "ev", "zj"], [tb, xb, zb, evb, zjb]
        call = lambda a: m(t=a[0], x=a[1], z=a[2], event_t=a[3], z_jump=a[4])
    else:
        m = models.DAE_Model(8, 2, 2, 2, H, direct_encode=True, solver=nd.Euler()).to(dev)
        m.one_launch = one
        names, tens = ["t", "x", "z", "v", "i", "ev", "zj", "vj"], [tb, xb, zb, vb, ib, evb, zjb, vjb]
        call = lambda a: m(t=a[0], x=a[1], z=a[2], v=a[3], i=a[4], event_t=a[5], z_jump=a[6], v_jump=a[7])
    for k, nme in enumerate(names + ["none"]):
        a = list(tens); hold = None
        if nme != "none":
            a[k], hold = at_end(tens[k])
        with torch.no_grad():
            call(a)
        torch.cuda.synchronize()
        m.zero_grad(set_to_none=True)
        out = call(a)
        sum((o * o).sum() for o in out).backward()
        torch.cuda.synchronize()
    print("ok model", tag, H, one, flush=True)
# ---- the loss kernel
for D in (8, 2, 5):
    p, tg, mk = r(Tm, Bm, D).permute(1, 0, 2), r(Bm, Tm, D), torch.ones(Bm, Tm, 1, device=dev)
    each_at_end(["pred", "target", "mask"], [p.contiguous(), tg, mk], lambda p_, t_, m_: L.masked_mse_terms(p_, t_, m_, want_grad=True), ("masked_mse", D))
print("probe done")
