import torch, time, sys
sys.path.insert(0, ".")
from py_psnode_amd import models, neural_dae as nd
torch.manual_seed(0)
B, T = 4096, 1001
m = models.DAE_Model(8, 2, 2, 2, 64, direct_encode=True, solver=nd.RK4()).cuda(); m.solver.fused = "require"
t = (torch.arange(T, dtype=torch.float32) * 0.01).view(1, T, 1).repeat(B, 1, 1).cuda()
r = lambda *s: (0.1 * torch.randn(*s)).cuda()
x, z, v, i = r(B, T, 8), r(B, T, 2), r(B, T, 2), r(B, T, 2)
ev, zj, vj = torch.full((B, 2, 1), -1.0).cuda(), r(B, 2, 2), r(B, 2, 2)
with torch.no_grad():
    m(t=t, x=x, z=z, v=v, i=i, event_t=ev, z_jump=zj, v_jump=vj); torch.cuda.synchronize()
    t0 = time.perf_counter(); m(t=t, x=x, z=z, v=v, i=i, event_t=ev, z_jump=zj, v_jump=vj); torch.cuda.synchronize()
print("DAE_02 hidden 64 model forward B=4096 T=1001: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
