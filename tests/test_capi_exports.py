"""The C-ABI library loads and exports every symbol include/psnode_hip.h declares (no compute, no GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "psnode_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(psnode_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    from py_psnode_amd import _lib
    assert _declared() == sorted(_lib.EXPORTS)


def test_library_exports_every_declared_symbol():
    from py_psnode_amd import _lib
    lib = _lib.load()
    for name in _declared():
        assert hasattr(lib, name), name
    assert lib.psnode_abi_version() == _lib.ABI_VERSION == 10
    assert b"gfx950" in lib.psnode_build_info()
    assert lib.psnode_status_string(0) == b"ok" and b"NULL" in lib.psnode_status_string(-1)


def test_host_only_entry_points():
    """workspace sizing and argument validation run without touching a device."""
    from py_psnode_amd import _lib
    lib = _lib.load()
    m = _lib.MlpF32()
    m.n_layers, m.in_dim = 4, 30
    for k, o in enumerate((64, 64, 64, 8)):
        m.out_dim[k] = o
    need = lib.psnode_workspace_bytes(ctypes.byref(m), None)
    assert need >= 4 * (30 * 64 + 64 * 64 * 2 + 64 * 8)
    a = _lib.OdeArgsF32()
    a.method, a.x_dim, a.z_dim, a.T, a.B = 7, 8, 2, 10, 4
    assert lib.psnode_ode_integrate_f32(ctypes.byref(a), None, 0, None) == -3      # bad method
    a.method = _lib.RK4_38
    a.de = m
    a.de.in_dim = 31
    assert lib.psnode_ode_integrate_f32(ctypes.byref(a), None, 0, None) == -2      # in_features != 3*(xd+zd)
    a.de.in_dim = 30
    assert lib.psnode_ode_integrate_f32(ctypes.byref(a), None, 0, None) == -1      # NULL weights
    assert lib.psnode_ode_integrate_f32(None, None, 0, None) == -1
    with pytest.raises(ValueError):
        _lib.check(-2, "x")
    with pytest.raises(_lib.PsnodeStatusError):
        _lib.check(-6, "x")


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from py_psnode_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.PsnodeLibraryError):
        _lib.load()


def test_recogniser():
    import torch.nn as nn
    from py_psnode_amd import fused, models
    from py_psnode_amd.neural_dae import neural_base
    de = models.DE_Func(10, (64, 64, 64), 8)
    assert len(fused.de_layers_of(de, 10, 8)) == 4
    assert fused.de_layers_of(de, 11, 8) is None                      # wrong recipe width
    assert fused.de_layers_of(lambda **k: 0, 10, 8) is None           # plain callable
    assert fused.de_layers_of(neural_base.DE_Func(2, 1, 4), 3, 2) is None   # legacy block
    tanh = models.DE_Func(10, (64,), 8)
    tanh.x_dot[1] = nn.Tanh()
    assert fused.de_layers_of(tanh, 10, 8) is None
    extra = models.DE_Func(10, (64,), 8)
    extra.other = nn.Linear(2, 2)
    assert fused.de_layers_of(extra, 10, 8) is None
    ae = models.AE_Func(26, (64, 64, 64), 2)
    assert len(fused.ae_layers_of(ae, 14, 12, 2)) == 4


def test_integration_md_binding_example_matches_the_abi():
    """The ctypes structs a maintainer would copy out of INTEGRATION.md must have the layout of include/psnode_hip.h (here: of the
    binding this package itself uses): a struct that stops short of the trailing fields makes the library read past its end."""
    import ctypes
    import os
    import re
    from py_psnode_amd import _lib
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
    block = text[text.index("# neural_dae/_psnode.py"):]
    block = block[:block.index("lib = ctypes.CDLL")]
    ns = {}
    exec("import ctypes\nfrom ctypes import c_int32, c_int64, c_uint32, c_void_p, c_size_t\n" + re.sub(r"^import ctypes, torch$", "", block, flags=re.M), ns)
    assert ctypes.sizeof(ns["Mlp"]) == ctypes.sizeof(_lib.MlpF32)
    assert ctypes.sizeof(ns["View"]) == ctypes.sizeof(_lib.ViewF32)
    assert ctypes.sizeof(ns["OdeArgs"]) == ctypes.sizeof(_lib.OdeArgsF32)
    assert [f[0] for f in ns["OdeArgs"]._fields_] == [f[0] for f in _lib.OdeArgsF32._fields_]
