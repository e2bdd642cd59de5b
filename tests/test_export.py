"""TorchScript export compatibility: models.ODE_Model / DAE_Model.save_model write what the
reference's save_model writes for the downstream C++ consumer (neural_00_ODE_01_no_encode.py:93-101,
neural_01_DAE_01_no_encode.py:117-133, neural_0x_..._02_direct_encode.py) -- same file set, same positional forward
signature, same state_dict keys/shapes, and the reloaded modules reproduce the reference modules' outputs bit for bit
on recorded inputs.  Fixtures: tests/golden/export_schema.json + g6_export_io.npz (make_export_fixture.py)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

from py_psnode_amd import models  # noqa: E402

SCHEMA = json.load(open(os.path.join(GOLD, "export_schema.json")))


def _build(tag):
    dims = SCHEMA[tag]["dims"]
    de = tag.endswith("02") or tag.endswith("02_z0")
    return (models.ODE_Model if tag.startswith("ode") else models.DAE_Model)(*dims, direct_encode=de)


@pytest.mark.parametrize("final", [False, True])
@pytest.mark.parametrize("tag", sorted(SCHEMA))
def test_export_matches_reference_files_signatures_and_values(tmp_path, tag, final):
    io = np.load(os.path.join(GOLD, "g6_export_io.npz"))
    entry = SCHEMA[tag]
    model = _build(tag)
    for name, mod in entry["modules"].items():      # the reference's weights into our sub-modules, by the reference's keys
        sd = {k: torch.from_numpy(io[f"{tag}.{name}.state.{k}"]) for k in mod["state"]}
        getattr(model, name).load_state_dict(sd, strict=True)
    out_dir = tmp_path / "m"
    (model.final_save if final else model.save_model)(out_dir)
    assert sorted(os.listdir(out_dir)) == entry["files"]
    if "dim_txt" in entry:
        assert open(out_dir / "dim.txt").read() == entry["dim_txt"]
    for name, mod in entry["modules"].items():
        sm = torch.jit.load(str(out_dir / f"{name}.pt"))
        assert [a.name for a in sm.forward.schema.arguments][1:] == mod["args"], name
        assert {k: list(v.shape) for k, v in sm.state_dict().items()} == mod["state"], name
        ins = [torch.from_numpy(io[f"{tag}.{name}.in.{a}"]) for a in mod["args"]]
        assert torch.equal(sm(*ins), torch.from_numpy(io[f"{tag}.{name}.out"])), name
