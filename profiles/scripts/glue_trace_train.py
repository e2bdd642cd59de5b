"""Which ATen ops (with shapes and the Python line that issued them) make the elementwise glue of one bench.Trainer step
(fused forward + masked-MSE + fused backward).  usage: glue_trace_train.py [ode01|dae01] [rk4|euler] [hidden]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
wl, method, H = (sys.argv + ["dae01", "rk4", "64"])[1:4]
dev = torch.device("cuda", 0)
w = dict(bench.WORKLOADS[wl]); w["H"] = int(H)
p = bench.to_dev(bench.make_problem(w, w["B"], w["T"]), dev)
tr = bench.Trainer(w, p, method, "auto", "mse-fused", dev)
for _ in range(3):
    tr.step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    tr.step(); torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=6):
    if e.key.startswith("aten::") and e.device_time_total > 0:
        st = [s for s in e.stack if "py_psnode_amd" in s or "bench.py" in s]
        rows.append((e.device_time_total, e.self_cpu_time_total, e.key, e.count, str(e.input_shapes)[:90], (st[0] if st else "")[-70:]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"# {wl} {method} H{H}: aten ops with device time, one step; total {tot:.0f} us device, {sum(r[1] for r in rows):.0f} us self cpu")
for r_ in rows[:45]:
    print(f"{r_[0]:8.0f} us dev {r_[1]:7.0f} us cpu  {r_[2]:20s} x{r_[3]:<3d} {r_[4]:90s} {r_[5]}")
import time
t0 = time.perf_counter(); tr.step(); t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"# host enqueue of one step: {(t1 - t0) * 1e3:.2f} ms")
