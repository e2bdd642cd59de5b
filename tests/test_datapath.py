"""Data path (py_psnode_amd/datapath.py): the device-resident loader yields what the scripts' DataLoader route yields.

CPU tests run the loader with device='cpu' (it is host plumbing); the gpu test checks the pinned staging and that the
batches feed the fused model unchanged."""
import os
import sys

import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from py_psnode_amd import datapath  # noqa: E402
from py_psnode_amd.neural_dae.neural_base import DAE_Curves_Sample, ODE_Curves_Sample  # noqa: E402


def _write_npz(path, n=23, T=12, dae=False, with_mask=True, seed=0):
    """Synthetic dataset in the npz format of SURVEY.md App. C (neural_base.py:14-32,150-158), incl. -1 padded tails."""
    r = np.random.default_rng(seed)
    f32 = lambda *s: r.standard_normal(s).astype(np.float32)
    t = np.tile((np.arange(T, dtype=np.float32) * 0.01).reshape(1, T, 1), (n, 1, 1))
    mask = np.ones((n, T, 1 if dae else 8), dtype=np.float32)
    for s in range(0, n, 5):            # unstable samples: padded with t = -1 and masked out
        t[s, T - 3:] = -1.0
        mask[s, T - 3:] = 0.0
    d = dict(name=np.array([["x", "pu"]] * 8, dtype=object), t=t, x=f32(n, T, 8), z=f32(n, T, 2),
             event_t=np.tile(np.array([0.03, 0.07], dtype=np.float32).reshape(1, 2, 1), (n, 1, 1)), z_jump=f32(n, 2, 2))
    if dae:
        d.update(v=f32(n, T, 2), i=f32(n, T, 2), v_jump=f32(n, 2, 2))
    if with_mask or dae:
        d["mask"] = mask
    np.savez(path, **d)


@pytest.mark.parametrize("dae", [False, True])
@pytest.mark.parametrize("shuffle", [False, True])
@pytest.mark.parametrize("batch,drop_last", [(5, False), (8, True), (64, False)])
def test_resident_loader_yields_the_dataloader_batches(tmp_path, dae, shuffle, batch, drop_last):
    p = str(tmp_path / "d.npz")
    _write_npz(p, dae=dae)
    ds = (DAE_Curves_Sample if dae else ODE_Curves_Sample)(p, "cpu")
    torch.manual_seed(123)
    ref = list(DataLoader(ds, batch_size=batch, shuffle=shuffle, drop_last=drop_last))
    res = datapath.ResidentDataset(ds, "cpu")
    torch.manual_seed(123)
    loader = datapath.ResidentLoader(res, batch_size=batch, shuffle=shuffle, drop_last=drop_last)
    got = list(loader)
    assert len(got) == len(ref) == len(loader)
    assert res.fields == (datapath.DAE_FIELDS if dae else datapath.ODE_FIELDS)
    for gb, rb in zip(got, ref):
        assert len(gb) == len(rb)
        for g, r in zip(gb, rb):
            assert g.shape == r.shape and g.is_contiguous() and torch.equal(g, r)


def test_resident_dataset_defaults_and_errors(tmp_path):
    p = str(tmp_path / "d.npz")
    _write_npz(p, with_mask=False)
    ds = ODE_Curves_Sample(p, "cpu", num_sample=10, cut_length=7)
    res = datapath.ResidentDataset(ds, "cpu")
    assert len(res) == 10 and res.x.shape == (10, 7, 8) and torch.equal(res.mask, torch.ones(10, 7, 8))   # neural_base.py:31
    assert res.nbytes() == sum(getattr(ds, k).numel() * 4 for k in datapath.ODE_FIELDS)
    ds.x = ds.x.double()
    with pytest.raises(TypeError):
        datapath.ResidentDataset(ds, "cpu")
    with pytest.raises(ValueError):
        datapath.ResidentLoader(res, batch_size=0)
    with pytest.raises(AttributeError):
        res.nope


@pytest.mark.gpu
def test_resident_batches_feed_the_fused_model(tmp_path):
    from py_psnode_amd import models, neural_dae as nd
    p = str(tmp_path / "d.npz")
    _write_npz(p, n=40, T=21)
    ds = ODE_Curves_Sample(p, "cuda")
    res = datapath.ResidentDataset(ds, "cuda")
    assert res.t.is_cuda and len(res) == 40
    model = models.ODE_Model(8, 2, 64, solver=nd.RK4()).cuda()
    model.solver.fused = "require"
    torch.manual_seed(5)
    ref_batches = [[d.cuda() for d in b] for b in DataLoader(ds, batch_size=16, shuffle=True)]
    torch.manual_seed(5)
    with torch.no_grad():
        for got, ref in zip(datapath.ResidentLoader(res, batch_size=16, shuffle=True), ref_batches):
            for g, r in zip(got, ref):
                assert g.is_cuda and torch.equal(g, r)
            t, x, z, event_t, z_jump, mask = got
            a = model(t=t, x=x, z=z, event_t=event_t, z_jump=z_jump)
            b = model(t=ref[0], x=ref[1], z=ref[2], event_t=ref[3], z_jump=ref[4])
            assert torch.equal(a, b)


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference not present (GPU box)")
@pytest.mark.parametrize("dae", [False, True])
def test_reference_dataset_class_loads_the_same_tensors(tmp_path, dae):
    """Build container only: the reference's own ODE_/DAE_Curves_Sample (neural_base.py:10-40,136-166) on the same npz gives
    the tensors our dataset mirror and the resident copy hold (incl. the seeded num_sample draw and cut_length)."""
    import types
    p = str(tmp_path / "d.npz")
    _write_npz(p, dae=dae, with_mask=dae)
    saved = {k: v for k, v in sys.modules.items() if k in ("ray", "ray.worker") or k == "neural_dae" or k.startswith("neural_dae.")}
    for k in list(saved):
        del sys.modules[k]
    ray, rw = types.ModuleType("ray"), types.ModuleType("ray.worker")
    rw.init = lambda *a, **k: None
    ray.worker = rw
    sys.modules.update({"ray": ray, "ray.worker": rw})      # neural_base.py:4 imports an unused symbol from ray
    sys.path.insert(0, "/root/reference")
    try:
        import importlib
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref_nb = importlib.import_module("neural_dae.neural_base")
        assert ref_nb.__file__.startswith("/root/reference")
        ref_ds = (ref_nb.DAE_Curves_Sample if dae else ref_nb.ODE_Curves_Sample)(p, "cpu", num_sample=9, cut_length=10)
    finally:
        sys.path.remove("/root/reference")
        for k in [k for k in sys.modules if k in ("ray", "ray.worker") or k == "neural_dae" or k.startswith("neural_dae.")]:
            del sys.modules[k]
        sys.modules.update(saved)
    ours = (DAE_Curves_Sample if dae else ODE_Curves_Sample)(p, "cpu", num_sample=9, cut_length=10)
    res = datapath.ResidentDataset(ref_ds, "cpu")            # duck-typed: accepts the reference's object as it is
    assert len(ref_ds) == len(ours) == len(res) == 9
    for k in res.fields:
        assert torch.equal(getattr(ref_ds, k), getattr(ours, k)) and torch.equal(getattr(res, k), getattr(ours, k)), k
    for a, b in zip(ref_ds[3], ours[3]):
        assert torch.equal(a, b)
