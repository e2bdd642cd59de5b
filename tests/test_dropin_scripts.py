"""Drop-in proof: the REFERENCE'S OWN script classes (ODE_Model / DAE_Model of the four neural_0x scripts) running on
top of this build's `neural_dae` package must reproduce the goldens the real reference produced.

Needs /root/reference for the script files, so it only runs in the build container (skipped on the GPU box)."""
import importlib
import os
import sys
import warnings

import pytest
import torch

from helpers import TOL_ORACLE, T, load, rel_err

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference scripts not present (GPU box)")

SCRIPTS = {"ode01": "neural_00_ODE_01_no_encode", "ode02": "neural_00_ODE_02_direct_encode",
           "dae01": "neural_01_DAE_01_no_encode", "dae02": "neural_01_DAE_02_direct_encode"}


@pytest.fixture(scope="module")
def swapped():
    import py_psnode_amd
    saved = {k: v for k, v in sys.modules.items() if k == "neural_dae" or k.startswith("neural_dae.")}
    nd = py_psnode_amd.install_as_neural_dae()
    sys.path.insert(0, REF)          # for the script files and their `utils` import; `neural_dae` is already ours
    warnings.filterwarnings("ignore")
    try:
        yield nd
    finally:
        sys.path.remove(REF)
        for k in [k for k in sys.modules if k == "neural_dae" or k.startswith("neural_dae.")]:
            del sys.modules[k]
        sys.modules.update(saved)
        for name in SCRIPTS.values():
            sys.modules.pop(name, None)
        sys.modules.pop("utils", None)


@pytest.mark.parametrize("tag", ["ode01", "ode02", "dae01", "dae02"])
def test_reference_script_models_on_our_neural_dae(swapped, tag):
    mod = importlib.import_module(SCRIPTS[tag])
    assert mod.Euler.__module__.startswith("py_psnode_amd."), "the script must have imported OUR solvers"
    d = load(f"g4_model_{tag}.npz")
    if tag == "ode01":
        m = mod.ODE_Model(8, 2, 64)
    elif tag == "ode02":
        m = mod.ODE_Model(8, 2, 16)
    elif tag == "dae01":
        m = mod.DAE_Model(8, 2, 2, 2, 64)
    else:
        m = mod.DAE_Model(8, 2, 2, 2, 16)
    m.load_state_dict({k[4:].replace("__", "."): T(v) for k, v in d.items() if k.startswith("sd__")})
    for method, cls in (("euler", swapped.Euler), ("rk4", swapped.RK4)):
        m.solver = cls()
        with torch.no_grad():
            if tag.startswith("ode"):
                out = m(t=T(d["t"]), x=T(d["x"]), z=T(d["z"]), event_t=T(d["event_t"]), z_jump=T(d["z_jump"]))
            else:
                out = m(t=T(d["t"]), x=T(d["x"]), z=T(d["z"]), v=T(d["v"]), i=T(d["i"]), event_t=T(d["event_t"]),
                        z_jump=T(d["z_jump"]), v_jump=T(d["v_jump"]))
        out = out if isinstance(out, tuple) else (out,)
        for k, o in enumerate(out):
            assert rel_err(o, d[f"{method}_out{k}"]) <= TOL_ORACLE, (tag, method, k)


def test_script_classes_are_recognised_for_fusion(swapped):
    """The scripts' in-file DE_Func/AE_Func (not ours) must pass the structural recogniser."""
    from py_psnode_amd import fused
    mod = importlib.import_module(SCRIPTS["dae01"])
    m = mod.DAE_Model(8, 2, 2, 2, 64)
    assert len(fused.de_layers_of(m.de_func, 14, 8)) == 4
    assert len(fused.ae_layers_of(m.ae_func, 14, 12, 2)) == 4
    ok, ev_t, zj, vj = fused._event_tensors(m.event.event_fn, m.event.jump_change_fn, True)
    assert ok and ev_t is None


@pytest.mark.parametrize("tag", ["ode02", "dae02"])
def test_accelerate_keeps_the_reference_models_intact(swapped, tag, tmp_path):
    """py_psnode_amd.accelerate() on the REFERENCE'S OWN direct_encode classes: encoders/decoders become RowsSequential (fused row
    kernels on a HIP device), state-dict keys / parameters / outputs / TorchScript export (`save_model`) unchanged."""
    import py_psnode_amd
    from py_psnode_amd.models import RowsSequential
    mod = importlib.import_module(SCRIPTS[tag])
    d = load(f"g4_model_{tag}.npz")
    m = mod.ODE_Model(8, 2, 16) if tag == "ode02" else mod.DAE_Model(8, 2, 2, 2, 16)
    keys = list(m.state_dict().keys())
    params = [id(p) for p in m.parameters()]
    assert py_psnode_amd.accelerate(m) is m and py_psnode_amd.accelerate(m) is m        # idempotent
    swapped_names = [n for n, c in m.named_children() if isinstance(c, RowsSequential)]
    assert swapped_names == [n for n, c in m.named_children() if n.endswith(("_encoder", "_decoder"))] and len(swapped_names) >= 3
    assert list(m.state_dict().keys()) == keys and [id(p) for p in m.parameters()] == params
    m.load_state_dict({k[4:].replace("__", "."): T(v) for k, v in d.items() if k.startswith("sd__")})
    m.solver = swapped.RK4()
    with torch.no_grad():
        if tag == "ode02":
            out = m(t=T(d["t"]), x=T(d["x"]), z=T(d["z"]), event_t=T(d["event_t"]), z_jump=T(d["z_jump"]))
        else:
            out = m(t=T(d["t"]), x=T(d["x"]), z=T(d["z"]), v=T(d["v"]), i=T(d["i"]), event_t=T(d["event_t"]),
                    z_jump=T(d["z_jump"]), v_jump=T(d["v_jump"]))
    for k, o in enumerate(out):
        assert rel_err(o, d[f"rk4_out{k}"]) <= TOL_ORACLE, (tag, k)
    m.save_model(tmp_path / "saved")                       # the script's own TorchScript export still works
    enc = torch.jit.load(str(tmp_path / "saved" / "x_encoder.pt"))
    x = T(d["x"])
    assert torch.equal(enc(x), torch.nn.Sequential.forward(m.x_encoder, x))
