#!/usr/bin/env python3
"""Achieved GRADIENT error of the fused training route: for every G7 model x method, every parameter /
input gradient of `loss.backward()` on the fused HIP route (solver.fused = "require") against
  (a) the reference's own fp32 gradients (tests/golden/g7_grad_*.npz, generated from /root/reference), and
  (b) an fp64 walk of the same model through this package's callback route on the CPU (the truth both fp32 runs approximate),
as max |g - ref| / max |ref| per tensor (the metric of tests/test_grad_goldens.py).  Prints the worst tensor per (model, method) and the
global worst, which is what TOL_GPU of the test is set from (<= 5 x worst achieved).  Also the reference's own distance to fp64.
    usage: python profiles/scripts/grad_accuracy_report.py            (GPU box; the goldens travel with the repo)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from helpers import T, load  # noqa: E402
import test_grad_goldens as tg  # noqa: E402


def rel(a, b):
    a, b = a.detach().double().cpu(), torch.as_tensor(b, dtype=torch.float64)
    scale = float(b.abs().max()) if b.numel() else 0.0
    return (float((a - b).abs().max()) / max(scale, 1e-6)) if b.numel() else 0.0


def grads_of(m, leaves, d, method):
    out = {}
    for name, p in m.named_parameters():
        out["p:" + name] = p.grad if p.grad is not None else torch.zeros_like(p)
    for k, a in leaves.items():
        if f"{method}_g_{k}" in d:
            out["in:" + k] = a.grad if a.grad is not None else torch.zeros_like(a)
    return out


def fp64_walk(tag, method):
    d = load(f"g7_grad_{tag}.npz")
    m = tg._build(tag).double()
    m.load_state_dict({k[4:].replace("__", "."): T(v).double() for k, v in d.items() if k.startswith("sd__")})
    m.solver = tg.SOLVERS[method]()
    m.solver.fused = "off"
    c = lambda k: T(d[k]).double()
    leaves = {k: c(k).requires_grad_(True) for k in ("x", "z", "v", "i", "z_jump", "v_jump")}
    if tag.startswith("dae"):
        res = m(t=c("t"), x=leaves["x"], z=leaves["z"], v=leaves["v"], i=leaves["i"], event_t=c("event_t"), z_jump=leaves["z_jump"],
                v_jump=leaves["v_jump"])
    else:
        res = m(t=c("t"), x=leaves["x"], z=leaves["z"], event_t=c("event_t"), z_jump=leaves["z_jump"])
    res = res if isinstance(res, tuple) else (res,)
    sum((r * c(f"G{k}")).sum() for k, r in enumerate(res)).backward()
    return grads_of(m, leaves, d, method)


worst_all = (0.0, "")
worst_all64 = (0.0, "")
print("%-12s %-8s | %-44s %9s | %-44s %9s | %9s" % ("model", "method", "worst tensor vs reference fp32", "rel err", "worst tensor vs fp64 walk", "rel err",
                                                    "ref vs 64"))
for tag in tg.TAGS:
    for method in ("euler", "midpoint", "rk4"):
        d, m, res, leaves = tg._run_model(tag, method, "cuda", "require")
        g = grads_of(m, leaves, d, method)
        t64 = fp64_walk(tag, method)
        w_ref, w_64, r_64 = (0.0, ""), (0.0, ""), 0.0
        for name, a in g.items():
            key = f"{method}_gp__" + name[2:].replace(".", "__") if name.startswith("p:") else f"{method}_g_{name[3:]}"
            e = rel(a, d[key])
            if e > w_ref[0]:
                w_ref = (e, name)
            e64 = rel(a, t64[name])
            if e64 > w_64[0]:
                w_64 = (e64, name)
            r_64 = max(r_64, rel(torch.as_tensor(d[key]), t64[name]))
        print("%-12s %-8s | %-44s %9.2e | %-44s %9.2e | %9.2e" % (tag, method, w_ref[1][:44], w_ref[0], w_64[1][:44], w_64[0], r_64))
        if w_ref[0] > worst_all[0]:
            worst_all = (w_ref[0], f"{tag} {method} {w_ref[1]}")
        if w_64[0] > worst_all64[0]:
            worst_all64 = (w_64[0], f"{tag} {method} {w_64[1]}")
print("worst achieved vs the reference's fp32 gradients: %.3e (%s)" % worst_all)
print("worst achieved vs the fp64 walk:                  %.3e (%s)" % worst_all64)
print("tests/test_grad_goldens.py TOL_GPU = %.1e" % tg.TOL_GPU)
