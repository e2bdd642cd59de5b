"""Which ATen ops (shapes, issuing Python line) and which HIP kernels make one training step of a direct_encode MODEL as bench.py's
MODEL TRAIN line runs it (encoders -> fused latent integrator -> decoders -> the script's loss -> backward).
usage: glue_trace_model.py [ode02|dae02] [rk4|euler]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from py_psnode_amd import loss as L, models  # noqa: E402
from py_psnode_amd import neural_dae as nd  # noqa: E402
wl, method = (sys.argv + ["ode02", "rk4"])[1:3]
dev = torch.device("cuda", 0)
w = dict(bench.WORKLOADS[wl]); B, T, H = w["B"], w["T"], w["H"]
g = torch.Generator().manual_seed(0)
r = lambda *s: (0.1 * torch.randn(*s, generator=g)).to(dev)
t = (torch.arange(T, dtype=torch.float32) * 0.01).view(1, T, 1).repeat(B, 1, 1).to(dev)
x, z, v, i = r(B, T, 8), r(B, T, 2), r(B, T, 2), r(B, T, 2)
ev, zj, vj = -torch.ones(B, 2, 1, device=dev), torch.zeros(B, 2, 2, device=dev), torch.zeros(B, 2, 2, device=dev)
mask8, mask1 = torch.ones(B, T, 8, device=dev), torch.ones(B, T, 1, device=dev)
solver = {"rk4": nd.RK4, "euler": nd.Euler, "midpoint": nd.Midpoint}[method]()
torch.manual_seed(0)
m = (models.ODE_Model(8, 2, H, direct_encode=True, solver=solver) if wl == "ode02" else models.DAE_Model(8, 2, 2, 2, H, direct_encode=True, solver=solver)).to(dev)
m.solver.fused = "require"


def step():
    m.zero_grad(set_to_none=True)
    if wl == "ode02":
        o = m(t=t, x=x, z=z, event_t=ev, z_jump=zj)
        loss = L.ode02_loss(o[0], o[1], x, mask8)[0]
    else:
        o = m(t=t, x=x, z=z, v=v, i=i, event_t=ev, z_jump=zj, v_jump=vj)
        loss = L.dae02_loss(o[0], o[1], o[2], o[3], x, i, mask1)[0]
    loss.backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(); torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=8):
    if e.key.startswith("aten::") and e.device_time_total > 0:
        st = [s for s in e.stack if "py_psnode_amd" in s or "bench.py" in s or "glue_trace" in s]
        rows.append((e.device_time_total, e.key, e.count, str(e.input_shapes)[:80], (st[0] if st else "")[-80:]))
rows.sort(reverse=True)
print(f"# {wl} {method} H{H} MODEL TRAIN: aten ops with device time, one step; total {sum(r_[0] for r_ in rows):.0f} us device")
for r_ in rows[:40]:
    print(f"{r_[0]:8.0f} us  {r_[1]:22s} x{r_[2]:<3d} {r_[3]:80s} {r_[4]}")
kern = {}
for e in prof.events():
    if e.device_type is not None and str(e.device_type).endswith("CUDA") and e.device_time_total > 0:
        kern[e.name[:100]] = kern.get(e.name[:100], [0, 0])
        kern[e.name[:100]][0] += e.device_time_total; kern[e.name[:100]][1] += 1
print("# device kernels of the step")
for k, (us, n) in sorted(kern.items(), key=lambda kv: -kv[1][0])[:30]:
    print(f"{us:8.0f} us x{n:<3d} {k}")
