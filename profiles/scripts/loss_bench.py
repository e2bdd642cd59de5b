"""K6 (psnode_masked_mse_f32) vs the same loss in PyTorch ops, BASELINE batch (B=4096, T=1001, xd=8), on one GPU.
Algorithmic bytes per (b,t) row, D=8: pred 32 + target 32 + mask 4*mw read, grad 32 written."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from py_psnode_amd import loss as L  # noqa: E402

dev = torch.device("cuda", 0)
B, T, D = 4096, 1001, 8
g = torch.Generator().manual_seed(0)
x = (0.3 * torch.randn(B, T, D, generator=g)).to(dev)
xs = (x.permute(1, 0, 2) + 0.05 * torch.randn(T, B, D, device=dev)).contiguous()      # time-major integrator output
mse = torch.nn.functional.mse_loss


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


rows = []
for mw in (D, 1):
    mask = (torch.rand(B, T, mw, generator=g) > 0.1).float().to(dev)
    inv = L.inv_mask_sum(mask)

    def fused_fwd_bwd():
        p = xs.permute(1, 0, 2).requires_grad_(True)
        L.masked_mse(p, x, mask, inv_norm=inv)[0].backward()

    def fused_kernel_only():
        L.masked_mse_terms(xs.permute(1, 0, 2), x, mask, inv_norm=inv, want_grad=True)

    def fused_value_only():
        L.masked_mse_terms(xs.permute(1, 0, 2), x, mask, inv_norm=inv, want_grad=False)

    def torch_fwd_bwd():
        p = xs.permute(1, 0, 2).requires_grad_(True)
        torch.sum(torch.sum(torch.sum(mse(p, x, reduction="none") * mask, dim=1), dim=0) / torch.sum(mask)).backward()

    algo = B * T * (3 * 4 * D + 4 * mw)
    for name, fn, nbytes in (("K6 loss+grad (kernel call)", fused_kernel_only, algo), ("K6 value only", fused_value_only, algo - B * T * 4 * D),
                             ("K6 via autograd (fwd+bwd)", fused_fwd_bwd, None), ("PyTorch ops fwd+bwd", torch_fwd_bwd, None)):
        ms = timed(fn)
        rows.append({"mask_width": mw, "what": name, "ms": round(ms, 4),
                     "GBs": round(nbytes / ms / 1e6, 1) if nbytes else None, "hbm_frac": round(nbytes / ms / 1e6 / 8000, 3) if nbytes else None})
        print(rows[-1])
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/loss_bench.json", "w"), indent=1)
