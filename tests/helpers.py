"""Shared test helpers: golden loading, layer extraction, error metric."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Two metrics.
#  rel_err      elementwise  max |y - ref| / max(|ref|, 1e-3)   (SURVEY.md 8(d)).  Used for oracle-vs-golden, where
#               both sides run the same ATen CPU ops.  It is NOT attainable across implementations in fp32: the
#               reference's own fp32 result differs from an fp64 evaluation of the same algorithm by 1.5e-5 .. 7e-5
#               under it (near-zero elements of O(1e-7) absolute roundoff) -- see DESIGN.md "Accuracy gate".
#  traj_rel_err per trajectory  max_{t,d} |y - ref| / max(max_{t,d} |ref|, 1e-3): "trajectories within 1e-5 rel-err of
#               reference" (north_star).  The reference's own fp32-vs-fp64 noise under it is 5e-7 .. 1.3e-6.
REL_FLOOR = 1e-3
TOL_GPU = 1e-5          # north_star tolerance, applied to traj_rel_err
TOL_ORACLE = 2e-6       # oracle vs goldens (rel_err): same ATen ops; allows a different CPU's GEMM blocking


def load(name):
    d = np.load(os.path.join(GOLDEN, name))
    return {k: d[k] for k in d.files}


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def layers(d, prefix):
    """[(W,b), ...] of an nn.Sequential stored as '<prefix>__<idx>__weight' arrays, in index order."""
    idx = sorted({int(k[len(prefix) + 2:].split("__")[0]) for k in d if k.startswith(prefix + "__") and k.endswith("__weight")})
    return [(T(d[f"{prefix}__{i}__weight"]), T(d[f"{prefix}__{i}__bias"])) for i in idx]


def rel_err(y, ref):
    y = torch.as_tensor(y, dtype=torch.float64)
    ref = torch.as_tensor(ref, dtype=torch.float64)
    assert y.shape == ref.shape, (y.shape, ref.shape)
    if ref.numel() == 0:
        return 0.0
    return float(((y - ref).abs() / ref.abs().clamp_min(REL_FLOOR)).max())


def traj_rel_err(y, ref, bdim=1):
    """max over trajectories of (max abs error of the trajectory / max abs value of the reference trajectory)."""
    y = torch.as_tensor(y, dtype=torch.float64)
    ref = torch.as_tensor(ref, dtype=torch.float64)
    assert y.shape == ref.shape, (y.shape, ref.shape)
    if ref.numel() == 0:
        return 0.0
    other = [k for k in range(ref.dim()) if k != bdim]
    err = (y - ref).abs().amax(other)
    scale = ref.abs().amax(other).clamp_min(REL_FLOOR)
    return float((err / scale).max())


def tm(a):
    """[B,T,D] array -> time-major [T,B,D] tensor VIEW (like the scripts' permute(1,0,2))."""
    return T(a).permute(1, 0, 2)
