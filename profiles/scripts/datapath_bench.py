"""Per-batch cost of forming the north-star batch (B=4096, T=1001; t,x,z,event_t,z_jump,mask) on the device:
the scripts' route (DataLoader default collate on the host + six pageable .to(device), neural_00_ODE_01_no_encode.py:288,343-347)
vs py_psnode_amd.datapath (dataset resident in HBM, one index_select per tensor)."""
import json
import os
import sys
import time

import torch
from torch.utils.data import DataLoader, Dataset

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from py_psnode_amd import datapath  # noqa: E402

N, T, B = 8192, 1001, 4096
g = torch.Generator().manual_seed(0)


class Curves(Dataset):          # same fields / __getitem__ as ODE_Curves_Sample (neural_base.py:10-40)
    def __init__(self):
        self.t = (torch.arange(T, dtype=torch.float32) * 0.01).view(1, T, 1).repeat(N, 1, 1)
        self.x, self.z = 0.1 * torch.randn(N, T, 8, generator=g), 0.1 * torch.randn(N, T, 2, generator=g)
        self.event_t, self.z_jump = -torch.ones(N, 2, 1), torch.zeros(N, 2, 2)
        self.mask = torch.ones(N, T, 8)

    def __len__(self):
        return N

    def __getitem__(self, i):
        return self.t[i], self.x[i], self.z[i], self.event_t[i], self.z_jump[i], self.mask[i]


ds = Curves()
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
out = {}

t0 = time.perf_counter()
nb = 0
for _ in range(2):
    for batch in DataLoader(ds, batch_size=B, shuffle=True):
        batch = [d.to(dev) for d in batch]
        nb += 1
torch.cuda.synchronize()
out["dataloader_collate_plus_to_device_ms_per_batch"] = (time.perf_counter() - t0) / nb * 1e3

t0 = time.perf_counter()
res = datapath.ResidentDataset(ds, dev)
torch.cuda.synchronize()
out["resident_staging_once_ms"] = (time.perf_counter() - t0) * 1e3
out["resident_bytes"] = res.nbytes()
loader = datapath.ResidentLoader(res, batch_size=B, shuffle=True)
for batch in loader:
    pass
torch.cuda.synchronize()
t0 = time.perf_counter()
nb = 0
for _ in range(10):
    for batch in loader:
        nb += 1
torch.cuda.synchronize()
out["resident_loader_ms_per_batch"] = (time.perf_counter() - t0) / nb * 1e3
out["batch_bytes"] = sum(b.numel() * 4 for b in batch)
out["speedup"] = out["dataloader_collate_plus_to_device_ms_per_batch"] / out["resident_loader_ms_per_batch"]
print(json.dumps(out))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/datapath_bench.json", "w"), indent=1)
