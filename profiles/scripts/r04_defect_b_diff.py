"""Round 4, defect (b): which stored row differs first between grad_is = NULL and an explicit zero tensor on the split route (K7w)?
Every buffer fused.dae_backward_wide hands the kernel uninitialised is recorded (fused._empty is wrapped) in both runs and compared
pairwise.  usage (GPU box): PSNODE_DEBUG_GIS_NULL=1 python profiles/scripts/r04_defect_b_diff.py"""
import os, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
os.environ["PSNODE_DEBUG_GIS_NULL"] = "1"
import test_gpu_backward as tb
from py_psnode_amd import fused
rec = []
orig = fused._empty
def spy(*a, **k):
    t = orig(*a, **k); rec.append(t); return t
fused._empty = spy
names = ["ws"] + [f"act{q}" for q in range(3)] + [f"delta{q}" for q in range(3)] + [f"aact{q}" for q in range(3)] + [f"adelta{q}" for q in range(3)] + ["agi", "gk", "Xs", "dsum0", "dsum1", "dsum2"]
def run(H, method, B, Tn, xd, zd, vd, idim):
    de, ae, t, z, v, xi, a0, ev, zj, vj, Gx, Gi = tb._dae_raw_case(B, Tn, xd, zd, vd, idim, 1234, False, H=H)
    xe, ie = torch.zeros(Tn, B, 0, device="cuda"), torch.zeros(Tn, B, idim, device="cuda")
    xs, is_ = fused.dae_integrate(method, de, ae, xi, t, xe, z, v, ie, a0)
    out = []
    for gi in (None, torch.zeros_like(Gi)):
        rec.clear()
        g = fused.dae_backward_wide(method, de, ae, t, z, v, a0, xs, is_, Gx, gi, fuse_de=False, chunk_steps=Tn)
        torch.cuda.synchronize()
        out.append(([r.clone() for r in rec], g))
    (ra, ga), (rb, gb) = out
    print(f"== H{H} {method} B{B} T{Tn} dims {xd},{zd},{vd},{idim}: {len(ra)} / {len(rb)} buffers")
    for nm, p, q in zip(names, ra, rb):
        if nm == "ws" or p.shape != q.shape: continue
        d = (p - q).abs()
        if float(d.max()) > 0:
            idx = torch.nonzero(d > 0)
            print(f"   {nm:8s} shape {tuple(p.shape)} differs in {idx.shape[0]} elements, max {float(d.max()):.3e}; first {idx[0].tolist()} last {idx[-1].tolist()}  NULL {float(p[tuple(idx[0])]):.6e} zeros {float(q[tuple(idx[0])]):.6e}")
            if nm == "agi":
                cols = sorted(set(idx[:, -1].tolist())); rows = sorted(set(idx[:, 0].tolist()))
                print("            agi: differing slot columns", cols, "grid rows", rows)
        else:
            print(f"   {nm:8s} identical")
run(64, "euler", 9, 3, 4, 2, 0, 2)
run(64, "euler", 9, 3, 4, 1, 0, 1)
