"""Shared test helpers: golden loading, layer extraction, error metric."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# accuracy gate of SURVEY.md section 8(d): max |y - y_ref| / max(|y_ref|, 1e-3)
REL_FLOOR = 1e-3
TOL_GPU = 1e-5          # north_star: trajectories within 1e-5 rel-err of the reference
TOL_ORACLE = 2e-6       # oracle vs goldens: same ATen ops; allows a different CPU's GEMM blocking


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


def tm(a):
    """[B,T,D] array -> time-major [T,B,D] tensor VIEW (like the scripts' permute(1,0,2))."""
    return T(a).permute(1, 0, 2)
