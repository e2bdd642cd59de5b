"""Which ATen ops (and shapes) launch the elementwise glue kernels in one training step of a script model: torch.profiler with shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from py_psnode_amd import loss as L, models  # noqa: E402
from py_psnode_amd import neural_dae as nd  # noqa: E402
dev = torch.device("cuda", 0)
B, T = 4096, 1001
g = torch.Generator().manual_seed(0)
r = lambda *s: (0.1 * torch.randn(*s, generator=g)).to(dev)
t = (torch.arange(T, dtype=torch.float32) * 0.01).view(1, T, 1).repeat(B, 1, 1).to(dev)
x, z, v, i = r(B, T, 8), r(B, T, 2), r(B, T, 2), r(B, T, 2)
ev, zj, vj = -torch.ones(B, 2, 1, device=dev), torch.zeros(B, 2, 2, device=dev), torch.zeros(B, 2, 2, device=dev)
mask1 = torch.ones(B, T, 1, device=dev)
tag = sys.argv[1] if len(sys.argv) > 1 else "dae02"
m = models.DAE_Model(8, 2, 2, 2, 64, direct_encode=True, solver=nd.RK4()).to(dev)
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
def step():
    opt.zero_grad()
    o = m(t=t, x=x, z=z, v=v, i=i, event_t=ev, z_jump=zj, v_jump=vj)
    loss = L.dae02_loss(o[0], o[1], o[2], o[3], x, i, mask1)[0]
    loss.backward()
    opt.step()
step(); step(); torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key in ("aten::add", "aten::add_", "aten::copy_", "aten::clone", "aten::contiguous", "aten::cat", "aten::mul", "aten::zeros_like", "aten::fill_", "aten::sum", "aten::select_backward", "aten::slice_backward", "aten::index_put_"):
        rows.append((e.device_time_total, e.key, e.count, str(e.input_shapes)[:140]))
rows.sort(reverse=True)
for r_ in rows[:40]:
    print(f"{r_[0]:10.0f} us  {r_[1]:22s} x{r_[2]:<4d} {r_[3]}")
