"""Out-of-bounds probe at the MODEL level: the four script models on their default fused routes (forward without autograd, and a training step
through the scripts' losses), ragged batch, every input tensor in turn ending exactly at the end of its own 32 MB allocation."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from py_psnode_amd import loss as L, models
from py_psnode_amd import neural_dae as nd
dev = torch.device("cuda", 0)
def at_end(t):
    big = torch.empty(8 * 1024 * 1024, dtype=t.dtype, device=dev)
    v = big[big.numel() - t.numel():].view(t.shape)
    v.copy_(t)
    return v, big
torch.manual_seed(0)
r = lambda *s: (0.1 * torch.randn(*s)).to(dev)
for (B, T) in [(37, 23), (5, 70)]:
    for events in (False, True):
        t = (torch.arange(T, dtype=torch.float32) * 0.01).view(1, T, 1).repeat(B, 1, 1).to(dev)
        x, z, v, i = r(B, T, 8), r(B, T, 2), r(B, T, 2), r(B, T, 2)
        ev = t[:, [2, T - 3], :].contiguous() if events else -torch.ones(B, 2, 1, device=dev)
        zj, vj = r(B, 2, 2), r(B, 2, 2)
        mask = torch.ones(B, T, 1, device=dev)
        for tag, H, method in (("ode01", 64, "rk4"), ("ode01", 128, "euler"), ("dae01", 64, "rk4"), ("dae01", 128, "euler"), ("ode02", 16, "rk4"), ("ode02", 16, "euler"),
                               ("dae02", 64, "rk4"), ("dae02", 64, "euler")):
            solver = {"rk4": nd.RK4, "euler": nd.Euler}[method]()
            if tag.startswith("ode"):
                m = models.ODE_Model(8, 2, H, direct_encode=tag == "ode02", solver=solver).to(dev)
                names, tens = ["t", "x", "z", "ev", "zj"], [t, x, z, ev, zj]
                call = lambda a: m(t=a[0], x=a[1], z=a[2], event_t=a[3], z_jump=a[4])
            else:
                m = models.DAE_Model(8, 2, 2, 2, H, direct_encode=tag == "dae02", solver=solver).to(dev)
                names, tens = ["t", "x", "z", "v", "i", "ev", "zj", "vj"], [t, x, z, v, i, ev, zj, vj]
                call = lambda a: m(t=a[0], x=a[1], z=a[2], v=a[3], i=a[4], event_t=a[5], z_jump=a[6], v_jump=a[7])
            m.solver.fused = "require"
            for k, nme in enumerate(names + ["none"]):
                a = list(tens); hold = None
                if nme != "none":
                    a[k], hold = at_end(tens[k])
                with torch.no_grad():
                    call(a)
                torch.cuda.synchronize()
                m.zero_grad(set_to_none=True)
                out = call(a)
                out = out if isinstance(out, tuple) else (out,)
                xa = a[1]
                if tag == "ode01": loss = L.ode01_loss(out[0], xa, mask)[0]
                elif tag == "ode02": loss = L.ode02_loss(out[0], out[1], xa, mask)[0]
                elif tag == "dae01": loss = L.dae01_loss(out[0], xa, out[1], a[4], mask)[0]
                else: loss = L.dae02_loss(out[0], out[1], out[2], out[3], xa, a[4], mask)[0]
                loss.backward()
                torch.cuda.synchronize()
            print("ok", tag, H, method, (B, T), "events" if events else "no events", flush=True)
print("probe done")
