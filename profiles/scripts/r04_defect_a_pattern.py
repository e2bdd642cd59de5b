"""Round 4, defect (a): WHICH entries of dL/dall_initial and dW1 are wrong on K4f <Midpoint, NZM=0, 8 waves, recompute> (run with
PSNODE_POISON=1 so that unwritten partials show as NaN).  usage: PSNODE_LIB_PATH=build/var_old/lib.so PSNODE_POISON=1 python ..."""
import os, sys, torch, torch.nn as nn
sys.path.insert(0, ".")
from py_psnode_amd import fused
H, method, B, Tn, xd, zd, seed = 128, "midpoint", 48, 7, 8, 0, 308
g = torch.Generator().manual_seed(seed); torch.manual_seed(seed)
lin = [nn.Linear(a_, b_) for a_, b_ in zip([3 * (xd + zd), H, H, H], [H, H, H, xd])]
layers = [(m.weight.detach().cuda(), m.bias.detach().cuda()) for m in lin]
t = (torch.arange(Tn, dtype=torch.float32) * 0.02).view(Tn, 1, 1).repeat(1, B, 1).cuda()
r = lambda *s_: (0.1 * torch.randn(*s_, generator=g)).cuda()
x_in, z = torch.zeros(Tn, B, xd, device="cuda"), r(Tn, B, zd)
x_in[0] = r(B, xd)
a0 = torch.cat((x_in[0], z[0]), -1)
G = torch.randn(Tn, B, xd, generator=g).cuda()
xs = fused.ode_integrate(method, layers, t, x_in, z, a0)
b = fused.ode_backward(method, layers, t, z, a0, xs, G, kernel="generic")
a = fused.ode_backward(method, layers, t, z, a0, xs, G, kernel="wide")
torch.cuda.synchronize()
ga, gb = a[3].cpu(), b[3].cpu()
print("ga0 [B, n]: per-column max |err| / scale:", [f"{float((ga[:, c] - gb[:, c]).abs().max() / gb.abs().max()):.1e}" for c in range(ga.shape[1])])
print("ga0 per-trajectory (first 20):", [f"{float((ga[k] - gb[k]).abs().max() / gb.abs().max()):.1e}" for k in range(20)])
print("ga0 ratio wide/generic, trajectory 0:", [f"{float(ga[0, c] / gb[0, c]):.3f}" for c in range(ga.shape[1])])
W, Wr = a[4][0].cpu(), b[4][0].cpu()
n = xd + zd
bad = ~(((W - Wr).abs() <= 3e-4 * Wr.abs().max()))
print("dW1 [H, 3n]: rows with a bad/NaN entry:", sorted(set(torch.nonzero(bad)[:, 0].tolist())))
print("dW1 bad columns:", sorted(set(torch.nonzero(bad)[:, 1].tolist())), " NaN count", int(torch.isnan(W).sum()), "of", W.numel())
for u in (0, 1, 16, 17, 64, 127):
    print(f"  row {u}: wide", [f"{float(v):+.3e}" for v in W[u, :2 * n:3]], "| generic", [f"{float(v):+.3e}" for v in Wr[u, :2 * n:3]])
