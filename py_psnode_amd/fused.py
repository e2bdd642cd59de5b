"""Host side of the fused HIP integrator: recognise the reference's MLP right-hand sides, marshal
tensors into the C ABI (include/psnode_hip.h) and enqueue on the caller's current HIP stream.

Raw-tensor entry points (`ode_integrate`, `dae_integrate`) take MLPs as [(W[out,in], b[out]), ...]
exactly as nn.Linear stores them; `plan_ode` / `plan_dae` do the recognition for the solver classes
in py_psnode_amd.neural_dae.  Nothing here computes on the CPU and nothing here imports oracle/.
"""
import ctypes
import os
import weakref
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import _lib

Layers = Sequence[Tuple[torch.Tensor, torch.Tensor]]

METHOD_ID = {"euler": _lib.EULER, "midpoint": _lib.MIDPOINT, "rk4": _lib.RK4_38}
KERNEL_ID = {"auto": _lib.KERNEL_AUTO, "generic": _lib.KERNEL_GENERIC, "mfma": _lib.KERNEL_MFMA, "wide": _lib.KERNEL_MFMA_WIDE}

# PSNODE_POISON=1 (debug / `pytest -m gpu` leg of tests/test_gpu_fuzz.py): every buffer this module hands a kernel uninitialised --
# outputs, stored rows, workspaces -- is filled with NaN bit patterns first, so that a kernel (or a host-side contraction) that consumes
# memory nobody wrote shows up as NaN instead of as whatever the caching allocator happened to recycle.
_POISON = os.environ.get("PSNODE_POISON", "0") == "1"


def _empty(*size, **kw) -> torch.Tensor:
    t = torch.empty(*size, **kw)
    if _POISON and t.numel():
        if t.dtype.is_floating_point:
            t.fill_(float("nan"))
        elif t.dtype == torch.uint8:
            t.fill_(0xFF)            # 0xFFFFFFFF read as fp32 is a NaN
        else:
            t.fill_(-(1 << 30))
    return t


# ----------------------------------------------------------------------------- recognition
def sequential_layers(seq) -> Optional[List[Tuple[torch.Tensor, torch.Tensor]]]:
    """[(W,b), ...] if `seq` is nn.Sequential(Linear, ELU(alpha=1), Linear, ..., Linear), else None
    (the only MLP shape the reference's live right-hand sides use, neural_00_ODE_01_no_encode.py:61-64)."""
    if not isinstance(seq, nn.Sequential) or len(seq) == 0 or len(seq) % 2 == 0:
        return None
    out = []
    for k, m in enumerate(seq):
        if k % 2 == 0:
            if type(m) is not nn.Linear or m.bias is None:
                return None
            out.append((m.weight, m.bias))
        else:
            if type(m) is not nn.ELU or m.alpha != 1.0:
                return None
    if len(out) > _lib.MAX_LAYERS:
        return None
    for (w, _), (w2, _) in zip(out[:-1], out[1:]):
        if w2.shape[1] != w.shape[0]:
            return None
    return out


def de_layers_of(x_func, n: int, x_dim: int):
    """Layers of a DE_Func (attribute `x_dot`, input recipe cat(a0, s-a0, s))."""
    if not isinstance(x_func, nn.Module) or _overrides_forward_hooks(x_func):
        return None
    layers = sequential_layers(getattr(x_func, "x_dot", None))
    if layers is None or layers[0][0].shape[1] != 3 * n or layers[-1][0].shape[0] != x_dim:
        return None
    if not _only_params_of(x_func, x_func.x_dot):
        return None
    return layers


def ae_layers_of(i_func, n: int, m: int, i_dim: int):
    """Layers of an AE_Func (attribute `i_calculator`, input recipe cat(a0, x, z, v))."""
    if not isinstance(i_func, nn.Module) or _overrides_forward_hooks(i_func):
        return None
    layers = sequential_layers(getattr(i_func, "i_calculator", None))
    if layers is None or layers[0][0].shape[1] != n + m or layers[-1][0].shape[0] != i_dim:
        return None
    if not _only_params_of(i_func, i_func.i_calculator):
        return None
    return layers


def _mlp_eval(layers, u):
    for k, (w, b) in enumerate(layers):
        u = nn.functional.linear(u, w, b)
        if k + 1 < len(layers):
            u = nn.functional.elu(u)
    return u


def _recipe_ok(mod: nn.Module, layers, kind: str, widths) -> bool:
    """Does `mod.forward` really compute the recipe the kernels hard-code?  The structural checks (attribute name, Sequential
    shape, no extra parameters) say nothing about forward(): a user DE_Func that scales its output, uses t0 or concatenates in
    another order would be integrated WRONGLY.  One numeric probe per (module, forward function): a few random rows through the
    module's own forward against MLP(cat(a0, s - a0, s)) (DE) / MLP(cat(a0, x, z, v)) (AE) on the module's device.  The result
    is cached on the module; this package's own classes are known and skip the probe."""
    fwd = type(mod).forward
    if getattr(fwd, "_psnode_recipe", None) == kind:
        return True
    key = (fwd, kind, tuple(widths))
    cached = mod.__dict__.get("_psnode_probe")
    if cached is not None and cached[0] == key:
        return cached[1]
    ok = False
    try:
        w0 = layers[0][0]
        dev, dt = w0.device, w0.dtype
        g = torch.Generator(device="cpu").manual_seed(1234)
        R = 5
        parts = [torch.randn(R, d, generator=g).to(device=dev, dtype=dt) for d in widths]
        a0 = torch.randn(R, sum(widths), generator=g).to(device=dev, dtype=dt)
        t0 = torch.rand(R, 1, generator=g).to(device=dev, dtype=dt)
        with torch.no_grad():
            if kind == "de_ode":
                got = mod(t0=t0, xt=parts[0], zt=parts[1], all_initial=a0)
                s_ = torch.cat(parts, -1)
                want = _mlp_eval(layers, torch.cat((a0, s_ - a0, s_), -1))
            elif kind == "de_dae":
                got = mod(t0=t0, xt=parts[0], zt=parts[1], vt=parts[2], it=parts[3], all_initial=a0)
                s_ = torch.cat(parts, -1)
                want = _mlp_eval(layers, torch.cat((a0, s_ - a0, s_), -1))
            else:   # "ae": all_initial spans x|z|v|i, the inputs x, z, v
                a0 = torch.randn(R, widths[3], generator=g).to(device=dev, dtype=dt)
                got = mod(xt=parts[0], zt=parts[1], vt=parts[2], all_initial=a0)
                want = _mlp_eval(layers, torch.cat((a0, parts[0], parts[1], parts[2]), -1))
            ok = bool(got.shape == want.shape and torch.allclose(got, want, rtol=1e-4, atol=1e-6))
    except Exception:
        ok = False
    mod.__dict__["_psnode_probe"] = (key, ok)
    return ok


def _overrides_forward_hooks(mod: nn.Module) -> bool:
    return bool(mod._forward_hooks) or bool(mod._forward_pre_hooks)


def _only_params_of(mod: nn.Module, seq: nn.Module) -> bool:
    """A module is taken to follow the DE / AE input recipe when its only parameters are those of its
    `x_dot` / `i_calculator` Sequential (true of every live DE_Func / AE_Func in the reference scripts; the
    legacy neural_base.DE_Func has many more sub-modules and neither attribute, so it never gets here)."""
    return {id(p) for p in mod.parameters()} == {id(p) for p in seq.parameters()}


# ----------------------------------------------------------------------------- marshalling
def _f32_dev(t: torch.Tensor, dev, name: str) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: fused integrator is fp32-only, got {t.dtype}")
    if t.device != dev:
        raise ValueError(f"{name}: on {t.device}, expected {dev}")
    return t.detach()


def _view(t: Optional[torch.Tensor], dev, name: str, keep: list) -> _lib.ViewF32:
    """[T,B,D] tensor -> strided view struct; copies only when the last dim is not unit-stride."""
    if t is None or t.shape[-1] == 0:
        return _lib.ViewF32(None, 0, 0)
    t = _f32_dev(t, dev, name)
    if t.shape[-1] > 1 and t.stride(2) != 1:
        t = t.contiguous()
    keep.append(t)
    return _lib.ViewF32(t.data_ptr(), t.stride(0), t.stride(1))


def _aligned16(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """The MFMA backward kernels of the latent shapes read rows as float4: a view whose base or strides are not 16-byte aligned
    would make the library report 'unsupported' AFTER the fused forward has run (the generic backward does not fit those shapes).
    Such a view -- rare: torch allocations are 256-byte aligned, widths there are multiples of 16 -- is copied once."""
    if t is None or t.shape[-1] < 4:
        return t
    if t.data_ptr() % 16 or t.stride(-1) != 1 or any(st % 4 for st in t.stride()[:-1]):
        return t.contiguous()
    return t


def _mlp(layers: Layers, dev, name: str, keep: list) -> _lib.MlpF32:
    m = _lib.MlpF32()
    if not 1 <= len(layers) <= _lib.MAX_LAYERS:
        raise ValueError(f"{name}: {len(layers)} Linear layers, supported 1..{_lib.MAX_LAYERS}")
    m.n_layers = len(layers)
    m.in_dim = layers[0][0].shape[1]
    for k, (w, b) in enumerate(layers):
        w = _f32_dev(w, dev, f"{name}.weight[{k}]").contiguous()
        b = _f32_dev(b, dev, f"{name}.bias[{k}]").contiguous()
        keep += [w, b]
        m.out_dim[k] = w.shape[0]
        m.weight[k] = w.data_ptr()
        m.bias[k] = b.data_ptr()
    return m


def _check_tb(name: str, a: Optional[torch.Tensor], T: int, B: int, min_T: Optional[int] = None):
    """Leading dims of a time-major input against the call's (T, B): a mismatch would be an out-of-bounds device read."""
    if a is None or a.shape[-1] == 0:
        return
    need_T = T if min_T is None else min_T
    if a.dim() != 3 or a.shape[1] != B or a.shape[0] < need_T:
        raise ValueError(f"{name}: shape {tuple(a.shape)} does not cover [T={need_T}, B={B}, D]")


def _check_jump(name: str, j: Optional[torch.Tensor], B: int, width: int, event_idx):
    if event_idx is None or width == 0:
        return
    if j is None:
        raise ValueError(f"{name}: events need the jump values")
    if j.dim() != 3 or j.shape[0] != B or j.shape[2] != width or j.shape[1] < 1:
        raise ValueError(f"{name}: shape {tuple(j.shape)}, expected [B={B}, nE>=1, {width}]")


def _jump(j: Optional[torch.Tensor], dev, name: str, keep: list):
    if j is None or j.shape[-1] == 0:
        return None, 0, 0
    j = _f32_dev(j, dev, name)
    if j.shape[-1] > 1 and j.stride(2) != 1:
        j = j.contiguous()
    keep.append(j)
    return j.data_ptr(), j.stride(0), j.stride(1)


def event_table(t: torch.Tensor, event_t: Optional[torch.Tensor], check_duplicates: bool = False) -> Optional[torch.Tensor]:
    """int32[T-1] device table: index of the event at each step, -1 = none.

    Same decision as ODE_Event.event_fn / jump_change_fn (neural_base.py:52-62): trajectory 0's clock
    against trajectory 0's event list, exact fp32 equality -- resolved by one tiny kernel instead of one
    host sync per step.  `check_duplicates` synchronises and raises where the reference would
    (two events at one time make its `.view(z0.shape)` fail).
    """
    T = t.shape[0]
    if event_t is None or T < 2 or event_t.shape[1] == 0:      # an empty event list is "no events", as in the reference
        return None
    lib = _lib.load()
    dev = t.device
    t_arg, event_arg = t, event_t          # the caller's objects: what the duplicate-check memo is keyed on (detach() makes new ones)
    t = _f32_dev(t, dev, "t")
    event_t = _f32_dev(event_t, dev, "event_t")
    tab = _empty(T - 1, dtype=torch.int32, device=dev)
    dup = torch.zeros(1, dtype=torch.int32, device=dev)
    n_ev = event_t.shape[1]
    st = torch.cuda.current_stream(dev).cuda_stream
    rc = lib.psnode_event_table_f32(T - 1, t.data_ptr(), t.stride(0), event_t.data_ptr(), event_t.stride(1), n_ev,
                                    tab.data_ptr(), dup.data_ptr(), st)
    _lib.check(rc, "psnode_event_table_f32")
    if check_duplicates and not _dup_check_known(t_arg, event_arg):
        if int(dup.item()):
            raise RuntimeError("two events share one time stamp: the reference's jump_change_fn cannot view "
                               "z_jump[:, mask] as z0.shape (neural_base.py:61)")
        _dup_check_remember(t_arg, event_arg)
    return tab


# (clock, event list) pairs already found free of duplicate event times: the same tensor OBJECTS (or views of the same base objects)
# at the same in-place version need no second 4-byte read-back -- which is a device synchronisation per call, 0.9 ms of a 0.94 ms
# ODE_02 forward (profiles/scripts/host_overhead.py); a new batch is a new object and is checked.
_DUP_OK = {}      # id(event tensor's base) -> (weakref to it, key of the event view, weakref to the clock's base, key of the clock view)


def _dup_key(t):
    base = t._base if t._base is not None else t
    return base, (t.data_ptr(), tuple(t.shape), tuple(t.stride()), base._version)


def _dup_check_known(t, event_t) -> bool:
    eb, ek = _dup_key(event_t)
    tb, tk = _dup_key(t)
    hit = _DUP_OK.get(id(eb))
    return hit is not None and hit[0]() is eb and hit[1] == ek and hit[2]() is tb and hit[3] == tk


def _dup_check_remember(t, event_t):
    eb, ek = _dup_key(event_t)
    tb, tk = _dup_key(t)
    ident = id(eb)
    _DUP_OK[ident] = (weakref.ref(eb, lambda _r, ident=ident: _DUP_OK.pop(ident, None)), ek, weakref.ref(tb), tk)


def _workspace(lib, de: _lib.MlpF32, ae, dev) -> torch.Tensor:
    nbytes = lib.psnode_workspace_bytes(ctypes.byref(de), ctypes.byref(ae) if ae is not None else None)
    return _empty(nbytes + 256, dtype=torch.uint8, device=dev)


def _aligned_ptr(ws: torch.Tensor):
    p = (ws.data_ptr() + 255) // 256 * 256
    return p, ws.numel() - (p - ws.data_ptr())


_MFMA_CLASSES = ("MFMA integrators K1 / K2 cover `in -> H -> H -> H -> out` ELU-MLPs with H <= 128 (any x_dim <= 16 for the ODE, "
                 "x_dim <= 8 and z+v+i <= 8 for the DAE), and -- weights streamed from L2 -- the ODE up to H = 192 at any x_dim <= 16 and "
                 "up to H = 256 at x_dim <= 8, the DAE up to H = 192 with z+v+i <= 6")
_k0_warned = False


def _mfma_miss(rc: int, kernel: str, what: str, de_layers: Layers):
    """kernel='mfma' on a shape the MFMA integrators do not carry: say which shapes they do, and what the fallback costs."""
    if rc == -5 and kernel == "mfma":
        widths = [int(w.shape[0]) for w, _ in de_layers[:-1]]
        raise _lib.UnsupportedShapeError(
            f"{what}: no MFMA integrator for hidden widths {widths}.  {_MFMA_CLASSES}.  kernel='auto' runs this shape on the generic "
            "kernel K0 (any width that fits the 160 KB LDS), at roughly 10x the time per state-step (40.5 vs 3.9 ms per 4096 x 1000 "
            "RK4 batch at hidden 64; DESIGN.md 'Shapes without an MFMA specialisation')")


def _note_k0(lib, args, dae: bool, de_layers: Layers):
    """AUTO landing on K0 with a no_encode-style MLP wider than the MFMA classes: never silently ~10x slower."""
    global _k0_warned
    if _k0_warned or len(de_layers) != 4:
        return
    widths = [int(w.shape[0]) for w, _ in de_layers[:-1]]
    if len(set(widths)) != 1 or widths[0] <= 128:
        return
    k = (lib.psnode_dae_kernel_for if dae else lib.psnode_ode_kernel_for)(ctypes.byref(args))
    if k == _lib.KERNEL_GENERIC:
        _k0_warned = True
        import warnings
        warnings.warn(f"hidden width {widths[0]} runs on the generic kernel K0, roughly 10x the time per state-step of the MFMA "
                      f"integrators.  {_MFMA_CLASSES}", RuntimeWarning, stacklevel=3)


def ode_integrate(method: str, de_layers: Layers, t, x, z, all_initial, event_t=None, z_jump=None,
                  input_true_x: bool = False, kernel: str = "auto", event_idx: Optional[torch.Tensor] = None,
                  check_events: bool = False, out: Optional[torch.Tensor] = None, save: bool = False):
    """Fused integrate_ODE (replaces my_solvers.py:52-80 + my_fixed_grid.py + DE_Func.forward).

    save=True (training forward, K1 shapes only -- `ode_save_hidden`): the kernel also writes what autograd would save, the hidden
    activations [T-1,S,3,B,Hp] and the stage inputs [T-1,S,B,xd]; returns (xs, (act, xstage)) for `ode_backward(..., saved=)`.

    t[T,B,1], x[T,B,xd], z[T,B,zd] may be arbitrary strided views with a unit-stride last dim
    (the scripts pass permute(1,0,2) views); returns a fresh contiguous xs[T,B,xd].
    Only x[0] is read unless input_true_x, so x may be a [1,B,xd] view (time-chunked launches restart from the previous
    chunk's last row); `out` (contiguous [T,B,xd]) lets the caller place the result, e.g. in a slice of a larger buffer.
    """
    lib = _lib.load()
    dev = x.device
    if dev.type != "cuda":
        raise ValueError("fused integrator needs tensors on a HIP device")
    T, B, xd = t.shape[0], x.shape[1], x.shape[2]
    if x.shape[0] < (T if input_true_x else 1):
        raise ValueError("x has fewer grid points than t")
    zd = z.shape[-1]
    _check_tb("t", t, T, B)
    _check_tb("z", z, T, B)
    if event_idx is not None and (event_idx.numel() < T - 1 or event_idx.dtype != torch.int32):
        raise ValueError(f"event_idx must be int32[T-1={T - 1}], got {event_idx.dtype}[{event_idx.numel()}]")
    keep: list = []
    a = _lib.OdeArgsF32()
    a.method = METHOD_ID[method]
    a.kernel = KERNEL_ID[kernel]
    a.flags = _lib.FLAG_INPUT_TRUE_X if input_true_x else 0
    a.x_dim, a.z_dim, a.T, a.B = xd, zd, T, B
    a.de = _mlp(de_layers, dev, "de", keep)
    a.t = _view(t, dev, "t", keep)
    a.x = _view(x, dev, "x", keep)
    a.z = _view(z, dev, "z", keep)
    a0 = _f32_dev(all_initial, dev, "all_initial").contiguous()
    if a0.shape != (B, xd + zd):
        raise ValueError(f"all_initial has shape {tuple(a0.shape)}, expected {(B, xd + zd)}")
    keep.append(a0)
    a.all_initial = a0.data_ptr()
    with torch.cuda.device(dev):
        if event_idx is None:
            event_idx = event_table(t, event_t, check_events)
        if event_idx is not None:
            _check_jump("z_jump", z_jump, B, zd, event_idx)
            keep.append(event_idx)
            a.event_idx = event_idx.data_ptr()
            a.z_jump, a.zj_stride_b, a.zj_stride_e = _jump(z_jump, dev, "z_jump", keep)
        if out is None:
            out = _empty((T, B, xd), dtype=torch.float32, device=dev)
        elif out.shape != (T, B, xd) or not out.is_contiguous() or out.dtype != torch.float32 or out.device != dev:
            raise ValueError("out must be a contiguous fp32 [T,B,xd] tensor on the inputs' device")
        a.x_out = out.data_ptr()
        saved = None
        if save:
            Hp = lib.psnode_ode_save_hidden(ctypes.byref(a))
            if Hp <= 0:
                raise _lib.UnsupportedShapeError("ode_integrate(save=True): the MFMA integrator K1 does not take this shape")
            S = {"euler": 1, "midpoint": 2, "rk4": 4}[method]
            L = len(de_layers) - 1       # hidden layers: 3 for the no_encode MLPs (K1), 1 for the latent ones at hidden 64 (K3c)
            saved = (_empty((max(T - 1, 0), S, L, B, Hp), dtype=torch.float32, device=dev),
                     _empty((max(T - 1, 0), S, B, xd), dtype=torch.float32, device=dev))
            if T >= 2:
                a.save_act, a.save_xstage = saved[0].data_ptr(), saved[1].data_ptr()
        ws = _workspace(lib, a.de, None, dev)
        wp, wn = _aligned_ptr(ws)
        rc = lib.psnode_ode_integrate_f32(ctypes.byref(a), wp, wn, torch.cuda.current_stream(dev).cuda_stream)
    _mfma_miss(rc, kernel, "psnode_ode_integrate_f32", de_layers)
    _lib.check(rc, "psnode_ode_integrate_f32")
    if kernel == "auto":
        _note_k0(lib, a, False, de_layers)
    # the stream-ordered caching allocator keeps `keep`/`ws` storage valid until the kernel has run
    return (out, saved) if save else out


def ode_save_hidden(method: str, de_layers: Layers, x_dim: int, z_dim: int, kernel: str = "auto") -> int:
    """Row width of the saved activations if the forward for these dims can save them (K1 proper), else 0."""
    if de_layers[0][0].device.type != "cuda" or len(de_layers) > _lib.MAX_LAYERS:
        return 0
    lib = _lib.load()
    a = _lib.OdeArgsF32()
    a.method, a.kernel, a.x_dim, a.z_dim, a.T, a.B = METHOD_ID[method], KERNEL_ID[kernel], x_dim, z_dim, 2, 1
    a.de = _mlp(de_layers, de_layers[0][0].device, "de", [])
    return int(lib.psnode_ode_save_hidden(ctypes.byref(a)))


def dae_integrate(method: str, de_layers: Layers, ae_layers: Layers, x_init, t, x, z, v, i, all_initial,
                  event_t=None, z_jump=None, v_jump=None, input_true_x: bool = False, input_true_i: bool = False,
                  kernel: str = "auto", event_idx: Optional[torch.Tensor] = None, check_events: bool = False, out=None,
                  save: bool = False):
    """Fused integrate_DAE (replaces my_solvers.py:82-131 + step functions + DE_Func/AE_Func forwards).
    `out` = (xs, is) contiguous [T,B,xd] / [T,B,id] tensors to write into (time-chunked launches).
    save=True (training forward, K2 shapes without teacher forcing -- `dae_save_hidden`): the kernel also writes what autograd would
    save (psnode_dae_args_f32::save_*); returns (xs, is, saved) with saved = (act [T-1,S,3,B,Hp], xstage [T-1,S,B,xd],
    ae_act [3,T,B,Hp], ev_act [nE,3,B,Hp] | None, ev_i [nE,B,16] | None) for `dae_backward(..., saved=)`."""
    lib = _lib.load()
    dev = x_init.device
    if dev.type != "cuda":
        raise ValueError("fused integrator needs tensors on a HIP device")
    T, B = t.shape[0], t.shape[1]
    xd, zd, vd, idim = x_init.shape[-1], z.shape[-1], v.shape[-1], i.shape[-1]
    if x_init.dim() != 2 or x_init.shape[0] != B:
        raise ValueError(f"x_init: shape {tuple(x_init.shape)}, expected [B={B}, x_dim]")
    _check_tb("t", t, T, B)
    _check_tb("z", z, T, B)
    _check_tb("v", v, T, B)
    if input_true_x:
        _check_tb("x", x, T, B)
    if input_true_i:
        _check_tb("i", i, T, B)
    if event_idx is not None and (event_idx.numel() < T - 1 or event_idx.dtype != torch.int32):
        raise ValueError(f"event_idx must be int32[T-1={T - 1}], got {event_idx.dtype}[{event_idx.numel()}]")
    keep: list = []
    a = _lib.DaeArgsF32()
    a.method = METHOD_ID[method]
    a.kernel = KERNEL_ID[kernel]
    a.flags = (_lib.FLAG_INPUT_TRUE_X if input_true_x else 0) | (_lib.FLAG_INPUT_TRUE_I if input_true_i else 0)
    a.x_dim, a.z_dim, a.v_dim, a.i_dim, a.T, a.B = xd, zd, vd, idim, T, B
    a.de = _mlp(de_layers, dev, "de", keep)
    a.ae = _mlp(ae_layers, dev, "ae", keep)
    a.t = _view(t, dev, "t", keep)
    a.x = _view(x if input_true_x else None, dev, "x", keep)
    a.z = _view(z, dev, "z", keep)
    a.v = _view(v, dev, "v", keep)
    a.i = _view(i if input_true_i else None, dev, "i", keep)
    if input_true_x and x.shape[-1] != xd:
        raise ValueError("input_true_x needs dataset x of width x_init.shape[-1]")
    xi = _f32_dev(x_init, dev, "x_init").contiguous()
    a0 = _f32_dev(all_initial, dev, "all_initial").contiguous()
    if a0.shape != (B, xd + zd + vd + idim):
        raise ValueError(f"all_initial has shape {tuple(a0.shape)}, expected {(B, xd + zd + vd + idim)}")
    keep += [xi, a0]
    a.x_init, a.all_initial = xi.data_ptr(), a0.data_ptr()
    with torch.cuda.device(dev):
        if event_idx is None:
            event_idx = event_table(t, event_t, check_events)
        if event_idx is not None:
            _check_jump("z_jump", z_jump, B, zd, event_idx)
            _check_jump("v_jump", v_jump, B, vd, event_idx)
            keep.append(event_idx)
            a.event_idx = event_idx.data_ptr()
            a.z_jump, a.zj_stride_b, a.zj_stride_e = _jump(z_jump, dev, "z_jump", keep)
            a.v_jump, a.vj_stride_b, a.vj_stride_e = _jump(v_jump, dev, "v_jump", keep)
        if out is None:
            xs = _empty((T, B, xd), dtype=torch.float32, device=dev)
            is_ = _empty((T, B, idim), dtype=torch.float32, device=dev)
        else:
            xs, is_ = out
            if xs.shape != (T, B, xd) or is_.shape != (T, B, idim) or not (xs.is_contiguous() and is_.is_contiguous()):
                raise ValueError("out must be contiguous fp32 ([T,B,xd], [T,B,id])")
        a.x_out, a.i_out = xs.data_ptr(), is_.data_ptr()
        saved = None
        if save:
            Hp = lib.psnode_dae_save_hidden(ctypes.byref(a))
            if Hp <= 0:
                raise _lib.UnsupportedShapeError("dae_integrate(save=True): the MFMA integrator K2 does not take this shape")
            S = {"euler": 1, "midpoint": 2, "rk4": 4}[method]
            f32 = dict(dtype=torch.float32, device=dev)
            n_ev = (z_jump if z_jump is not None else v_jump).shape[1] if event_idx is not None else 0
            L = len(de_layers) - 1       # hidden layers: 3 (K2), 1 for the latent shapes at hidden 64 (K3c; i0 rows are then i_dim wide)
            saved = (_empty((max(T - 1, 0), S, L, B, Hp), **f32), _empty((max(T - 1, 0), S, B, xd), **f32),
                     _empty((L, T, B, Hp), **f32),
                     torch.zeros((n_ev, L, B, Hp), **f32) if n_ev else None,
                     torch.zeros((n_ev, B, 16 if L == 3 else idim), **f32) if n_ev else None)
            a.save_act, a.save_xstage, a.save_ae_act = saved[0].data_ptr(), saved[1].data_ptr(), saved[2].data_ptr()
            if T < 2:       # no step: nothing but the head at grid point 0 is written; the struct wants all three or none
                dummy = _empty(16, **f32)
                keep.append(dummy)
                a.save_act = a.save_xstage = dummy.data_ptr()
            if n_ev:
                a.save_ev_act, a.save_ev_i = saved[3].data_ptr(), saved[4].data_ptr()
        ws = _workspace(lib, a.de, a.ae, dev)
        wp, wn = _aligned_ptr(ws)
        rc = lib.psnode_dae_integrate_f32(ctypes.byref(a), wp, wn, torch.cuda.current_stream(dev).cuda_stream)
    _mfma_miss(rc, kernel, "psnode_dae_integrate_f32", de_layers)
    _lib.check(rc, "psnode_dae_integrate_f32")
    if kernel == "auto":
        _note_k0(lib, a, True, de_layers)
    return (xs, is_, saved) if save else (xs, is_)


def dae_save_hidden(method: str, de_layers: Layers, ae_layers: Layers, x_dim: int, z_dim: int, v_dim: int, i_dim: int,
                    kernel: str = "auto") -> int:
    """Row width of the saved activations if the forward for these dims can save them (K2 proper), else 0."""
    if de_layers[0][0].device.type != "cuda" or max(len(de_layers), len(ae_layers)) > _lib.MAX_LAYERS:
        return 0
    lib = _lib.load()
    a = _lib.DaeArgsF32()
    a.method, a.kernel, a.x_dim, a.z_dim, a.v_dim, a.i_dim, a.T, a.B = METHOD_ID[method], KERNEL_ID[kernel], x_dim, z_dim, v_dim, i_dim, 2, 1
    dev = de_layers[0][0].device
    a.de, a.ae = _mlp(de_layers, dev, "de", []), _mlp(ae_layers, dev, "ae", [])
    return int(lib.psnode_dae_save_hidden(ctypes.byref(a)))


def _bwd_args(method, de_layers, x_dim, z_dim, T, B, dev, keep, kernel="auto"):
    a = _lib.OdeBwdArgsF32()
    a.method = METHOD_ID[method]
    a.kernel = KERNEL_ID[kernel]
    a.x_dim, a.z_dim, a.T, a.B = x_dim, z_dim, T, B
    a.de = _mlp(de_layers, dev, "de", keep)
    return a


def ode_backward_supported(method: str, de_layers: Layers, x_dim: int, z_dim: int, kernel: str = "auto") -> bool:
    """True if a fused backward kernel covers this shape: the MFMA class (3n->64->64->64->x, x<=8, z<=4) or any MLP whose
    activations and parameter gradients fit the LDS (generic backward)."""
    if de_layers[0][0].device.type != "cuda" or len(de_layers) > _lib.MAX_LAYERS:
        return False
    if kernel in ("auto", "mfma") and latent_wide_shape(de_layers, None, x_dim, z_dim):
        return True                          # K3w (saving) + K9w + library GEMMs
    lib = _lib.load()
    a = _bwd_args(method, de_layers, x_dim, z_dim, 2, 1, de_layers[0][0].device, [], kernel)
    return bool(lib.psnode_ode_backward_supported(ctypes.byref(a)))


def ode_backward_wide_supported(method: str, de_layers: Layers, x_dim: int, z_dim: int) -> bool:
    """Shapes of the two-part backward (psnode_ode_backward_wide_f32 + GEMMs): 3n -> H -> H -> H -> x, H in {32, 64, 128}."""
    if de_layers[0][0].device.type != "cuda" or len(de_layers) != 4:
        return False
    lib = _lib.load()
    a = _lib.OdeBwdWideArgsF32()
    a.method, a.x_dim, a.z_dim, a.T, a.B = METHOD_ID[method], x_dim, z_dim, 2, 1
    a.de = _mlp(de_layers, de_layers[0][0].device, "de", [])
    return bool(lib.psnode_ode_backward_wide_supported(ctypes.byref(a)))


def latent_wide_shape(de_layers: Layers, ae_layers: Optional[Layers], x_dim: int, z_dim: int, v_dim: int = 0, i_dim: int = 0) -> bool:
    """The latent shapes of the direct_encode models at a hidden width the dedicated latent kernels do not take (every H <= 128 with
    H % 4 == 0 other than 16 / 64 -- e.g. the scripts' argparse default --hidden 128): forward K3w, backward K9w + library GEMMs."""
    H = x_dim
    if len(de_layers) != 2 or H in (16, 64) or H < 4 or H > 128 or H % 4 or de_layers[0][0].device.type != "cuda":
        return False
    if ae_layers is None:
        return z_dim == H and tuple(de_layers[0][0].shape) == (H, 6 * H) and tuple(de_layers[1][0].shape) == (H, H)
    nblk = 4 if z_dim else 3
    return (len(ae_layers) == 2 and z_dim in (0, H) and v_dim == H and i_dim == H and tuple(de_layers[0][0].shape) == (H, 3 * nblk * H)
            and tuple(de_layers[1][0].shape) == (H, H) and tuple(ae_layers[0][0].shape) == (H, (2 * nblk - 1) * H)
            and tuple(ae_layers[1][0].shape) == (H, H))


def latent_backward_wide(method: str, de_layers: Layers, ae_layers: Optional[Layers], t, z, v, all_initial, xs, is_, grad_xs, grad_is,
                         event_idx=None, z_jump=None, v_jump=None, saved=None, need_grad_z: bool = True):
    """Backward of the latent integrate_ODE / integrate_DAE at the hidden widths of `latent_wide_shape` (split form): the sequential
    adjoint sweep K9w (psnode_latent_backward_wide_f32) reads the activations the K3w training forward saved and stores the adjoint rows;
    every parameter / input gradient is then a contraction over those rows as library GEMMs.  Returns the dict of `dae_backward` (for the
    ODE: keys x_init = dL/dx[0], z, z_jump, all_initial, de)."""
    if saved is None:
        raise ValueError("latent_backward_wide reads the activations of a forward call with save=True")
    lib = _lib.load()
    dev = xs.device
    dae = ae_layers is not None
    T, B, H = xs.shape
    zd = z.shape[-1] if z is not None else 0
    S = {"euler": 1, "midpoint": 2, "rk4": 4}[method]
    nblk = (4 if zd else 3) if dae else 2
    n = nblk * H
    f32 = dict(dtype=torch.float32, device=dev)
    keep: list = []
    a = _lib.LatentBwdWideArgsF32()
    a.method, a.hidden, a.z_dim, a.dae, a.T, a.B = METHOD_ID[method], H, zd, int(dae), T, B
    a.de = _mlp(de_layers, dev, "de", keep)
    if dae:
        a.ae = _mlp(ae_layers, dev, "ae", keep)
    a.t = _view(t, dev, "t", keep)
    a0 = _f32_dev(all_initial, dev, "all_initial").contiguous()
    xs_c = _f32_dev(xs, dev, "xs").contiguous()
    gx_c = _f32_dev(grad_xs, dev, "grad_xs").contiguous() if grad_xs is not None else torch.zeros_like(xs_c)
    gi_c = _f32_dev(grad_is, dev, "grad_is").contiguous() if (dae and grad_is is not None) else None
    keep += [a0, xs_c, gx_c, gi_c]
    a.grad_xs = gx_c.data_ptr()
    a.grad_is = gi_c.data_ptr() if gi_c is not None else None
    s_act, s_xst = saved[0], saved[1]
    s_ae = saved[2] if dae else None
    s_ev, s_evi = (saved[3], saved[4]) if dae and len(saved) > 3 else (None, None)
    if tuple(s_act.shape) != (max(T - 1, 0), S, 1, B, H) or tuple(s_xst.shape) != (max(T - 1, 0), S, B, H) or (dae and tuple(s_ae.shape) != (1, T, B, H)):
        raise ValueError("saved activations do not belong to this call (shape)")
    keep += [s_act, s_xst, s_ae, s_ev, s_evi]
    n_ev = 0
    evl = hit = None
    if event_idx is not None:
        keep.append(event_idx)
        a.event_idx = event_idx.data_ptr()
        n_ev = (z_jump if (z_jump is not None and zd) else v_jump).shape[1] if (dae or zd) else 0
        evl = event_idx[:T - 1].long()
        hit = (evl >= 0).view(T - 1, 1, 1)
    with torch.cuda.device(dev):
        gk, d1 = _empty((max(T - 1, 0), S, B, H), **f32), _empty((max(T - 1, 0), S, B, H), **f32)
        d1s = _empty((max(T - 1, 0), B, H), **f32)
        gx0 = _empty((B, H), **f32)
        a.gk, a.d1, a.d1s, a.grad_x0 = gk.data_ptr(), d1.data_ptr(), d1s.data_ptr(), gx0.data_ptr()
        if T >= 2:
            a.saved_act = s_act.data_ptr()
        gi = da1 = gi_ev = da1_ev = None
        if dae:
            gi, da1 = _empty((T, B, H), **f32), _empty((T, B, H), **f32)
            a.gi, a.da1, a.saved_ae_act = gi.data_ptr(), da1.data_ptr(), s_ae.data_ptr()
            if event_idx is not None:
                if s_ev is None or tuple(s_ev.shape) != (n_ev, 1, B, H):
                    raise ValueError("saved event activations do not belong to this call (shape)")
                gi_ev, da1_ev = torch.zeros((n_ev, B, H), **f32), torch.zeros((n_ev, B, H), **f32)
                a.gi_ev, a.da1_ev, a.saved_ev_act = gi_ev.data_ptr(), da1_ev.data_ptr(), s_ev.data_ptr()
        ws = _empty(lib.psnode_latent_backward_wide_workspace_bytes(H) + 256, dtype=torch.uint8, device=dev)
        wp, wn = _aligned_ptr(ws)
        _lib.check(lib.psnode_latent_backward_wide_f32(ctypes.byref(a), wp, wn, torch.cuda.current_stream(dev).cuda_stream),
                   "psnode_latent_backward_wide_f32")
        # ---- contractions over the stored rows
        (W1, _b1), (W2, _b2) = [(w.detach(), b.detach()) for w, b in de_layers]
        G = max(1, (T - 1) * S)
        R = lambda q: q.reshape(-1, H)
        g = {"z_jump": None, "v_jump": None, "z": None, "v": None}
        if T >= 2:
            gW2 = _gemm_tn(R(gk), R(s_act), G)
            gb2 = R(gk).sum(0)
            Px = _gemm_tn(R(d1), R(s_xst), G)
            # the external blocks each step used: the grid point's rows or, at a jump step, the jump rows; (DAE) i_k or the event's i0
            used = []
            if zd:
                zc = z.detach()[:T - 1]
                used.append(torch.where(hit, z_jump.detach()[:, evl.clamp_min(0)].permute(1, 0, 2), zc) if evl is not None else zc)
            if dae:
                vc = v.detach()[:T - 1]
                used.append(torch.where(hit, v_jump.detach()[:, evl.clamp_min(0)].permute(1, 0, 2), vc) if evl is not None else vc)
                ic = _f32_dev(is_, dev, "is").detach()[:T - 1]
                used.append(torch.where(hit, s_evi[evl.clamp_min(0)], ic) if evl is not None else ic)
            Pcat = torch.cat([Px] + [_gemm_tn(R(d1s), R(u.contiguous()), T - 1) for u in used], 1)          # [H, n]: d(Ws + Wd)-side products
            S1 = d1s.sum(0)                                                                                  # [B, H]
        else:
            gW2, gb2, Pcat, S1 = torch.zeros_like(W2), torch.zeros(H, **f32), torch.zeros((H, n), **f32), torch.zeros((B, H), **f32)
        A0cat = S1.t() @ a0                                                                                  # [H, n]
        gW1 = torch.cat((A0cat, Pcat - A0cat, Pcat), 1)
        ga0 = S1 @ (W1[:, 0:n] - W1[:, n:2 * n])
        g["de"] = [gW1, S1.sum(0), gW2, gb2]
        # input gradients through the DE's external blocks (the jump rows at jump steps)
        def route(raw, grid_target, jump_target):
            if evl is not None:
                if jump_target is not None:
                    jump_target.index_add_(1, evl.clamp_min(0), (raw * hit).permute(1, 0, 2))
                raw = torch.where(hit, torch.zeros_like(raw), raw)
            grid_target[:T - 1] += raw
        Fb = lambda blk: W1[:, 2 * n + H * blk:2 * n + H * (blk + 1)] + W1[:, n + H * blk:n + H * (blk + 1)]
        if zd and need_grad_z:
            g["z"] = torch.zeros((T, B, H), **f32)
            g["z_jump"] = torch.zeros((B, n_ev, H), **f32) if evl is not None else None
            if T >= 2:
                route((R(d1s) @ Fb(1)).view(T - 1, B, H), g["z"], g["z_jump"])
        if dae:
            g["v"] = torch.zeros((T, B, H), **f32)
            g["v_jump"] = torch.zeros((B, n_ev, H), **f32) if evl is not None else None
            if T >= 2:
                route((R(d1s) @ Fb(nblk - 2)).view(T - 1, B, H), g["v"], g["v_jump"])
            # ---- the AE head: rows per grid point (un-jumped inputs) and per event taken (x of the jump step, jump rows)
            (A1, _ab1), (A2, _ab2) = [(w.detach(), b.detach()) for w, b in ae_layers]
            ah = s_ae[0]
            gA2 = _gemm_tn(R(gi), R(ah), T)
            gab2 = R(gi).sum(0)
            cols = [_gemm_tn(R(da1), R(xs_c), T)]
            if zd:
                cols.append(_gemm_tn(R(da1), R(z.detach().contiguous()), T))
            cols.append(_gemm_tn(R(da1), R(v.detach().contiguous()), T))
            Sa1 = da1.sum(0)
            if evl is not None and n_ev:
                step_of = torch.zeros(n_ev, dtype=torch.long, device=dev).scatter_reduce_(
                    0, evl.clamp_min(0), torch.arange(T - 1, device=dev) * (evl >= 0), "amax")
                gA2 = gA2 + R(gi_ev).t() @ R(s_ev[:, 0])
                gab2 = gab2 + R(gi_ev).sum(0)
                ev_in = [xs_c[step_of]] + ([z_jump.detach().permute(1, 0, 2)] if zd else []) + [v_jump.detach().permute(1, 0, 2)]
                cols = [c + R(da1_ev).t() @ R(u.contiguous()) for c, u in zip(cols, ev_in)]
                Sa1 = Sa1 + da1_ev.sum(0)
            gA1 = torch.cat([Sa1.t() @ a0] + cols, 1)
            ga0 = ga0 + Sa1 @ A1[:, 0:n]
            g["ae"] = [gA1, Sa1.sum(0), gA2, gab2]
            Ab = lambda q: A1[:, n + H * q:n + H * (q + 1)]          # q: 0 = x, then z (if any), v
            if zd and need_grad_z:
                g["z"] += (R(da1) @ Ab(1)).view(T, B, H)
                if g["z_jump"] is not None:
                    g["z_jump"] += (R(da1_ev) @ Ab(1)).view(n_ev, B, H).permute(1, 0, 2)
            g["v"] += (R(da1) @ Ab(nblk - 2)).view(T, B, H)
            if g["v_jump"] is not None:
                g["v_jump"] += (R(da1_ev) @ Ab(nblk - 2)).view(n_ev, B, H).permute(1, 0, 2)
    g["x_init"] = gx0
    g["all_initial"] = ga0
    return g


def _padded_hidden(h: int) -> int:
    """Width class the MFMA kernels run a hidden width at (csrc/psnode_pack.h: padded_hidden): rows they store have this many columns,
    the ones beyond `h` are exact zeros."""
    return 32 if h <= 32 else (64 if h <= 64 else 128)


def _pad_rows(m: torch.Tensor, rows: int) -> torch.Tensor:
    return m if m.shape[0] == rows else torch.cat((m, m.new_zeros((rows - m.shape[0],) + tuple(m.shape[1:]))), 0)


def _gemm_tn(a2: torch.Tensor, b2: torch.Tensor, groups: int) -> torch.Tensor:
    """a2^T @ b2 for tall-skinny [N, p], [N, q] (N in the millions): `groups` independent partial products + one sum, so the library
    GEMM has parallelism over the contraction (one [p,N]x[N,q] call runs on a handful of workgroups: 21 vs 124 TFLOP/s at p=q=128)."""
    N = a2.shape[0]
    groups = max(1, min(groups, 256))     # the [groups, p, q] partial products are materialised: cap them (small B x long T chunks)
    while groups > 1 and N % groups:
        groups -= 1
    return torch.bmm(a2.view(groups, N // groups, -1).transpose(1, 2), b2.view(groups, N // groups, -1)).sum(0)


def ode_backward_wide(method: str, de_layers: Layers, t, z, all_initial, xs, grad_xs, event_idx=None, z_jump=None,
                      chunk_steps: Optional[int] = None, need_grad_z: bool = True):
    """Backward of `ode_integrate` for hidden widths without a one-launch backward kernel (32, 128; also valid at 64): the sequential
    adjoint sweep on the MFMA kernel K4w in time chunks (psnode_ode_backward_wide_f32), the parameter gradients and the input
    gradients that are plain contractions over its stored rows as library GEMMs.  Same return value as `ode_backward`."""
    lib = _lib.load()
    dev = xs.device
    T, B, xd = xs.shape
    zd = z.shape[-1]
    n = xd + zd
    Hr = de_layers[0][0].shape[0]                       # the MLP's width; H = the width the kernel runs it at (zero-padded rows)
    H = _padded_hidden(Hr)
    S = {"euler": 1, "midpoint": 2, "rk4": 4}[method]
    keep: list = []
    a = _lib.OdeBwdWideArgsF32()
    a.method, a.x_dim, a.z_dim, a.T, a.B = METHOD_ID[method], xd, zd, T, B
    a.de = _mlp(de_layers, dev, "de", keep)
    a.t, a.z = _view(t, dev, "t", keep), _view(z, dev, "z", keep)
    a0 = _f32_dev(all_initial, dev, "all_initial").contiguous()
    xs_c, g_c = _f32_dev(xs, dev, "xs").contiguous(), _f32_dev(grad_xs, dev, "grad_xs").contiguous()
    keep += [a0, xs_c, g_c]
    a.all_initial, a.xs, a.grad_xs = a0.data_ptr(), xs_c.data_ptr(), g_c.data_ptr()
    if event_idx is not None:
        keep.append(event_idx)
        a.event_idx = event_idx.data_ptr()
        a.z_jump, a.zj_stride_b, a.zj_stride_e = _jump(z_jump, dev, "z_jump", keep)
    W1, W2, W3, W4 = (w.detach() for w, _ in de_layers)
    gW = [torch.zeros_like(w) for w in (W1, W2, W3, W4)]
    gb = [torch.zeros(w.shape[0], dtype=torch.float32, device=dev) for w in (W1, W2, W3, W4)]
    gz = torch.zeros((T, B, zd), dtype=torch.float32, device=dev) if (zd > 0 and need_grad_z) else None
    gzj = torch.zeros((B, z_jump.shape[1], zd), dtype=torch.float32, device=dev) if (event_idx is not None and zd > 0) else None
    S1 = torch.zeros((B, H), dtype=torch.float32, device=dev)
    carry = torch.zeros((B, xd), dtype=torch.float32, device=dev)
    a.carry = carry.data_ptr()
    if T >= 2:
        if chunk_steps is None:      # ~3 GB of stored rows per chunk
            chunk_steps = max(1, min(T - 1, int(3e9 // (6 * 4 * S * B * H))))
        Fz = _pad_rows(W1[:, 2 * n + xd:3 * n] + W1[:, n + xd:2 * n], H) if zd > 0 else None   # (Ws + Wd)[:, z columns]
        with torch.cuda.device(dev):
            nbytes = lib.psnode_ode_backward_wide_workspace_bytes(ctypes.byref(a))
            ws = _empty(nbytes + 256, dtype=torch.uint8, device=dev)
            wp, wn = _aligned_ptr(ws)
            st = torch.cuda.current_stream(dev).cuda_stream
            zt = z.detach() if zd > 0 else None
            for k1 in range(T - 1, 0, -chunk_steps):
                k0 = max(0, k1 - chunk_steps)
                Tc = k1 - k0
                rows = [_empty((Tc, S, B, H), dtype=torch.float32, device=dev) for _ in range(6)]
                gk = _empty((Tc, S, B, xd), dtype=torch.float32, device=dev)
                Xs = _empty((Tc, S, B, xd), dtype=torch.float32, device=dev)
                dsum = [_empty((Tc, B, H), dtype=torch.float32, device=dev) for _ in range(3)]
                a.k0, a.k1 = k0, k1
                for q in range(3):
                    a.act[q], a.delta[q], a.dsum[q] = rows[q].data_ptr(), rows[3 + q].data_ptr(), dsum[q].data_ptr()
                a.gk, a.xstage = gk.data_ptr(), Xs.data_ptr()
                _lib.check(lib.psnode_ode_backward_wide_f32(ctypes.byref(a), wp, wn, st), "psnode_ode_backward_wide_f32")
                h1, h2, h3, d1, d2, d3 = (r.view(-1, H) for r in rows)
                G = Tc * S
                gW[3] += _gemm_tn(gk.view(-1, xd), h3, G)[:, :Hr]; gb[3] += gk.view(-1, xd).sum(0)
                gW[2] += _gemm_tn(d3, h2, G)[:Hr, :Hr]; gb[2] += dsum[2].sum((0, 1))[:Hr]
                gW[1] += _gemm_tn(d2, h1, G)[:Hr, :Hr]; gb[1] += dsum[1].sum((0, 1))[:Hr]
                # input of L1 per (step, stage): cat(a0, s - a0, s), s = cat(X_s, z of the step (jump values at event steps))
                if zd > 0:
                    zc = zt[k0:k1]                                                           # [Tc, B, zd] view
                    if event_idx is not None:
                        evc = event_idx[k0:k1].long()
                        hit = (evc >= 0).view(Tc, 1, 1)
                        zc = torch.where(hit, z_jump.detach()[:, evc.clamp_min(0)].permute(1, 0, 2), zc)
                    s_in = torch.cat((Xs, zc.unsqueeze(1).expand(Tc, S, B, zd)), -1)
                else:
                    s_in = Xs
                U = torch.cat((a0.view(1, 1, B, n).expand(Tc, S, B, n), s_in - a0, s_in), -1).reshape(-1, 3 * n)
                gW[0] += _gemm_tn(d1, U, G)[:Hr]
                D1 = dsum[0]                                                                 # [Tc, B, H]: sum over the stages
                D1s = D1.sum(0)
                S1 += D1s; gb[0] += D1s.sum(0)[:Hr]
                if zd > 0 and (gz is not None or gzj is not None):
                    gzc = D1.reshape(-1, H) @ Fz                                             # [Tc*B, zd]
                    gzc = gzc.view(Tc, B, zd)
                    if event_idx is not None:
                        if gzj is not None:
                            gzj.index_add_(1, evc.clamp_min(0), (gzc * hit).permute(1, 0, 2))
                        gzc = torch.where(hit, torch.zeros_like(gzc), gzc)
                    if gz is not None:
                        gz[k0:k1] = gzc
                del rows, gk, Xs, U, D1, dsum
    gx0 = carry + g_c[0]
    ga0 = S1 @ _pad_rows(W1[:, 0:n] - W1[:, n:2 * n], H)                                    # d all_initial = sum_t D1 . (Wa - Wd)
    return gx0, gz, gzj, ga0, [gW[0], gb[0], gW[1], gb[1], gW[2], gb[2], gW[3], gb[3]]


def _check_saved(act, xst, T, B, xd, S, L, dev):
    """The saved stage activations / stage inputs reach the kernels as raw pointers: a tuple from another call (other T, B, method or
    width) would be read out of bounds, so its shape is checked here ([T-1,S,L,B,Hp] / [T-1,S,B,xd], contiguous, on this device)."""
    ok = (act.dim() == 5 and tuple(act.shape[:4]) == (T - 1, S, L, B) and tuple(xst.shape) == (T - 1, S, B, xd)
          and act.is_contiguous() and xst.is_contiguous() and act.device == dev and xst.device == dev
          and act.dtype == torch.float32 and xst.dtype == torch.float32)
    if not ok:
        raise ValueError(f"saved activations do not belong to this call: got {tuple(act.shape)} / {tuple(xst.shape)}, "
                         f"expected [{T - 1},{S},{L},{B},Hp] / [{T - 1},{S},{B},{xd}] contiguous fp32 on {dev}")


def _split_grads(flat, layers):
    out, off = [], 0
    for w, b in layers:
        out.append(flat[off:off + w.numel()].view_as(w)); off += w.numel()
        out.append(flat[off:off + b.numel()].view_as(b)); off += b.numel()
    return out


def dae_backward_supported(method: str, de_layers: Layers, ae_layers: Layers, x_dim, z_dim, v_dim, i_dim) -> bool:
    if de_layers[0][0].device.type != "cuda" or max(len(de_layers), len(ae_layers)) > _lib.MAX_LAYERS:
        return False
    if latent_wide_shape(de_layers, ae_layers, x_dim, z_dim, v_dim, i_dim):
        return True                          # K3w (saving) + K9w + library GEMMs
    lib = _lib.load()
    a = _lib.DaeBwdArgsF32()
    a.method = METHOD_ID[method]
    a.x_dim, a.z_dim, a.v_dim, a.i_dim, a.T, a.B = x_dim, z_dim, v_dim, i_dim, 2, 1
    dev = de_layers[0][0].device
    a.de, a.ae = _mlp(de_layers, dev, "de", []), _mlp(ae_layers, dev, "ae", [])
    return bool(lib.psnode_dae_backward_supported(ctypes.byref(a)))


def dae_backward_wide_supported(method: str, de_layers: Layers, ae_layers: Layers, x_dim, z_dim, v_dim, i_dim) -> bool:
    """Shapes of the two-part DAE backward (psnode_dae_backward_wide_f32 + GEMMs): DE 3n -> H -> H -> H -> x and
    AE n+x+z+v -> H -> H -> H -> i with H in {32, 64, 128}, x <= 8, z+v+i <= 8."""
    if de_layers[0][0].device.type != "cuda" or len(de_layers) != 4 or len(ae_layers) != 4:
        return False
    lib = _lib.load()
    a = _lib.DaeBwdWideArgsF32()
    a.method, a.x_dim, a.z_dim, a.v_dim, a.i_dim, a.T, a.B = METHOD_ID[method], x_dim, z_dim, v_dim, i_dim, 2, 1
    dev = de_layers[0][0].device
    a.de, a.ae = _mlp(de_layers, dev, "de", []), _mlp(ae_layers, dev, "ae", [])
    return bool(lib.psnode_dae_backward_wide_supported(ctypes.byref(a)))


def dae_backward_wide(method: str, de_layers: Layers, ae_layers: Layers, t, z, v, all_initial, xs, is_, grad_xs, grad_is, event_idx=None,
                      z_jump=None, v_jump=None, chunk_steps: Optional[int] = None, fuse_de: bool = True, saved=None, x_true=None, i_true=None):
    """Backward of `dae_integrate` for hidden widths <= 128 other than the one-launch kernel's (K7, hidden 64): the sequential adjoint
    sweep -- DE stages, AE head per grid point, event-time recompute -- on an MFMA kernel (psnode_dae_backward_wide_f32).
    fuse_de (default): ONE launch over the whole grid (K7f) that also forms the DE's parameter gradients and the DE's share of the
    input gradients; only the AE head's rows (one set per grid point) are contracted here.  fuse_de=False: round 2's split (K7w in time
    chunks, every contraction a library GEMM over stored rows).  saved = what `dae_integrate(save=True)` returned for the same call
    (fuse_de only): the kernel evaluates nothing forwards.  x_true / i_true [T,B,.]: backward of a teacher-forced call
    (input_true_x / input_true_i, my_solvers.py:111-121) -- the dataset rows the forward call fed the DE / the heads; K7f recompute form
    only.  Same return value as `dae_backward`."""
    lib = _lib.load()
    dev = xs.device
    T, B, xd = xs.shape
    zd, vd, idim = z.shape[-1], v.shape[-1], is_.shape[-1]
    nzv, ne = zd + vd, zd + vd + idim
    n = xd + ne
    Hr = de_layers[0][0].shape[0]                       # the MLPs' width; H = the width the kernel runs them at (zero-padded rows)
    H = _padded_hidden(Hr)
    S = {"euler": 1, "midpoint": 2, "rk4": 4}[method]
    keep: list = []
    a = _lib.DaeBwdWideArgsF32()
    a.method, a.x_dim, a.z_dim, a.v_dim, a.i_dim, a.T, a.B = METHOD_ID[method], xd, zd, vd, idim, T, B
    a.de, a.ae = _mlp(de_layers, dev, "de", keep), _mlp(ae_layers, dev, "ae", keep)
    a.t, a.z, a.v = _view(t, dev, "t", keep), _view(z, dev, "z", keep), _view(v, dev, "v", keep)
    a0 = _f32_dev(all_initial, dev, "all_initial").contiguous()
    xs_c, is_c = _f32_dev(xs, dev, "xs").contiguous(), _f32_dev(is_, dev, "is").contiguous()
    gx_c = _f32_dev(grad_xs, dev, "grad_xs").contiguous() if grad_xs is not None else torch.zeros_like(xs_c)
    gi_c = _f32_dev(grad_is, dev, "grad_is").contiguous() if grad_is is not None else None
    keep += [a0, xs_c, is_c, gx_c, gi_c]
    a.all_initial, a.xs, a.is_, a.grad_xs = a0.data_ptr(), xs_c.data_ptr(), is_c.data_ptr(), gx_c.data_ptr()
    a.grad_is = gi_c.data_ptr() if gi_c is not None else None
    f32 = dict(dtype=torch.float32, device=dev)
    n_ev = 0
    if event_idx is not None:
        keep.append(event_idx)
        a.event_idx = event_idx.data_ptr()
        a.z_jump, a.zj_stride_b, a.zj_stride_e = _jump(z_jump, dev, "z_jump", keep)
        a.v_jump, a.vj_stride_b, a.vj_stride_e = _jump(v_jump, dev, "v_jump", keep)
        n_ev = (z_jump if z_jump is not None else v_jump).shape[1]
        a.n_events = n_ev
        # rows of the event-time heads: zero-initialised, so that events no step takes contribute nothing
        ev_rows = [torch.zeros((n_ev, B, H), **f32) for _ in range(6)]
        ev_gi, ev_i = torch.zeros((n_ev, B, 16), **f32), torch.zeros((n_ev, B, 16), **f32)
        for q in range(3):
            a.ev_act[q], a.ev_delta[q] = ev_rows[q].data_ptr(), ev_rows[3 + q].data_ptr()
        a.ev_gi, a.ev_i = ev_gi.data_ptr(), ev_i.data_ptr()
    W1, W2, W3, W4 = (w.detach() for w, _ in de_layers)
    A1, A2, A3, A4 = (w.detach() for w, _ in ae_layers)
    # Host-side accumulators of the routes that contract stored rows here (the split form; the AE head's K7h route): made on demand.
    # The one-launch route at hidden <= 64 (every gradient formed in the kernel) needs none of them -- 18 fills, a [T,B,z+v] copy and 2
    # small GEMM operands per call that the step's profile showed as glue (profiles/r04o_glue_dae01.txt).
    gW = gb = S1 = Fe = Ae = zv_all = None
    gA, gab, Sa1 = [None] * 4, [None] * 4, None

    def host_accumulators(de_too: bool):
        nonlocal gW, gb, gA, gab, S1, Sa1, Fe, Ae, zv_all
        if Sa1 is None:
            gA = [torch.zeros_like(w) for w in (A1, A2, A3, A4)]
            gab = [torch.zeros(w.shape[0], **f32) for w in (A1, A2, A3, A4)]
            Sa1 = torch.zeros((B, H), **f32)                    # sum over the heads of the AE's delta_1
            Ae = _pad_rows(A1[:, n + xd:n + xd + nzv], H)
            zv_all = torch.cat((z.detach(), v.detach()), -1)    # [T, B, nzv] (one copy of the two input views)
        if de_too and S1 is None:
            gW = [torch.zeros_like(w) for w in (W1, W2, W3, W4)]
            gb = [torch.zeros(w.shape[0], **f32) for w in (W1, W2, W3, W4)]
            S1 = torch.zeros((B, H), **f32)                     # sum over steps and stages of the DE's delta_1
            Fe = _pad_rows(W1[:, n + xd:n + xd + nzv] + W1[:, 2 * n + xd:2 * n + xd + nzv], H)       # (Ws + Wd)[:, z|v columns]

    gzv = torch.zeros((T, B, nzv), **f32)                       # dL/d(z|v) of the un-jumped inputs
    gjump = torch.zeros((B, n_ev, nzv), **f32) if n_ev else None
    carry_x, carry_i = torch.zeros((B, xd), **f32), torch.zeros((B, 16), **f32)
    a.carry_x, a.carry_i = carry_x.data_ptr(), carry_i.data_ptr()
    if saved is not None and not fuse_de:
        raise ValueError("saved activations are read by the fused-DE form only")
    xt_c = it_c = None
    if x_true is not None or i_true is not None:
        if saved is not None or not fuse_de:
            raise ValueError("teacher forcing: the fused-DE recompute form only")
        xt_c = _f32_dev(x_true, dev, "x_true").contiguous() if x_true is not None else None
        it_c = _f32_dev(i_true, dev, "i_true").contiguous() if i_true is not None else None
        if (xt_c is not None and tuple(xt_c.shape) != (T, B, xd)) or (it_c is not None and tuple(it_c.shape) != (T, B, idim)):
            raise ValueError("x_true / i_true must be [T,B,x_dim] / [T,B,i_dim]")
        keep += [xt_c, it_c]
        a.flags = (_lib.FLAG_INPUT_TRUE_X if xt_c is not None else 0) | (_lib.FLAG_INPUT_TRUE_I if it_c is not None else 0)
        a.x_true = xt_c.data_ptr() if xt_c is not None else None
        a.i_true = it_c.data_ptr() if it_c is not None else None
    if fuse_de and saved is None and x_true is None and i_true is None:
        # one launch over the whole grid stores the AE head's rows of EVERY grid point (6 x [T,B,H] + [T,B,16] + the u rows of K7h): a very
        # long grid on a full card goes through the time-chunked split form instead (bounded at ~3 GB of rows per chunk).  Decided HERE,
        # before the fused-only fields of the argument struct are filled: psnode_dae_backward_wide_f32 picks K7f on grad_params_de != NULL
        # (round 3 cleared the flag after filling them -- every chunk but the first then failed with PSNODE_ERR_DIMS)
        free, _ = torch.cuda.mem_get_info(dev)
        if (6 * H + 40) * 4 * T * B > free // 2:
            fuse_de = False
    if fuse_de:
        gp_de = _empty(sum(w.numel() + w.shape[0] for w in (W1, W2, W3, W4)), **f32)
        ga0_de = _empty((B, n), **f32)
        a.grad_params_de, a.grad_all_initial_de = gp_de.data_ptr(), ga0_de.data_ptr()
        a.grad_zv = gzv.data_ptr()
        a.grad_jump = gjump.data_ptr() if gjump is not None else None
    jump_all = None
    if n_ev:
        parts = ([z_jump.detach()] if zd > 0 else []) + ([v_jump.detach()] if vd > 0 else [])
        jump_all = torch.cat(parts, -1)                         # [B, n_ev, nzv]

    def head_grads(act, delta, gi_slots, x_rows, zv_rows):
        """parameter / input gradients of AE heads from their stored rows; all arguments [R, B, .]"""
        nonlocal Sa1
        h1, h2, h3, d1, d2, d3 = (r.reshape(-1, H) for r in (*act, *delta))
        R = act[0].shape[0]
        gi = (gi_slots[..., nzv:ne] + gi_slots[..., ne + nzv:2 * ne]).reshape(-1, idim)
        gA[3].add_(_gemm_tn(gi, h3, R)[:, :Hr]); gab[3].add_(gi.sum(0))
        gA[2].add_(_gemm_tn(d3, h2, R)[:Hr, :Hr]); gab[2].add_(d3.sum(0)[:Hr])
        gA[1].add_(_gemm_tn(d2, h1, R)[:Hr, :Hr]); gab[1].add_(d2.sum(0)[:Hr])
        U = torch.cat((a0.view(1, B, n).expand(R, B, n), x_rows, zv_rows), -1).reshape(-1, n + xd + nzv)
        gA[0].add_(_gemm_tn(d1, U, R)[:Hr]); gab[0].add_(d1.sum(0)[:Hr])
        Sa1 += delta[0].sum(0)
        return (d1 @ Ae).view(R, B, nzv)

    def head_grads_hip(act, act_row_stride, delta, gi_slots, x_rows, zv_rows):
        """the same contractions on K7h (psnode_dae_head_grads_f32): act = 3 device pointers' tensors whose row r lies r*act_row_stride
        floats behind the first; delta / gi_slots / x_rows / zv_rows [R, B, .] contiguous"""
        nonlocal Sa1
        R = delta[0].shape[0]
        u = torch.zeros((R, B, 16), **f32)
        u[..., :xd] = x_rows
        if nzv > 0:
            u[..., xd:xd + nzv] = zv_rows
        h = _lib.DaeHeadGradsArgsF32()
        h.R, h.B, h.hidden, h.n_zv = R, B, Hr, nzv
        for q in range(3):
            h.act[q], h.delta[q] = act[q].data_ptr(), delta[q].data_ptr()
        h.act_row_stride = act_row_stride
        h.gi, h.u = gi_slots.data_ptr(), u.data_ptr()
        A1c = A1.contiguous()
        h.aw1, h.aw1_cols, h.zv_col0 = A1c.data_ptr(), A1c.shape[1], n + xd
        gza = _empty((R, B, 8), **f32) if nzv > 0 else None
        sa1 = _empty((B, H), **f32)
        out = _empty(lib.psnode_dae_head_grads_out_floats(Hr), **f32)
        h.grad_zv = gza.data_ptr() if gza is not None else None
        h.sa1, h.out = sa1.data_ptr(), out.data_ptr()
        nb = lib.psnode_dae_head_grads_workspace_bytes(ctypes.byref(h))
        hws = _empty(nb + 256, dtype=torch.uint8, device=dev)
        hp_, hn_ = _aligned_ptr(hws)
        _lib.check(lib.psnode_dae_head_grads_f32(ctypes.byref(h), hp_, hn_, torch.cuda.current_stream(dev).cuda_stream),
                   "psnode_dae_head_grads_f32")
        o = 0
        gA[1].add_(out[o:o + Hr * Hr].view(Hr, Hr)); o += Hr * Hr
        gA[2].add_(out[o:o + Hr * Hr].view(Hr, Hr)); o += Hr * Hr
        P3 = out[o:o + 16 * Hr].view(16, Hr); o += 16 * Hr
        gA[3].add_(P3[nzv:ne] + P3[ne + nzv:2 * ne])
        P0 = out[o:o + Hr * 16].view(Hr, 16); o += Hr * 16
        gA[0][:, n:n + xd + nzv].add_(P0[:, :xd + nzv])
        gA[0][:, :n].add_(sa1[:, :Hr].t() @ a0)
        for q in range(3):
            gab[q].add_(out[o:o + Hr]); o += Hr
        sg = out[o:o + 16]
        gab[3].add_(sg[nzv:ne] + sg[ne + nzv:2 * ne])
        Sa1 += sa1
        return gza[..., :nzv] if gza is not None else None

    # (grad_is = None goes to the kernels as NULL: they read the rows of `is` instead and mask them out, branch-free.  Round 3 passed an
    #  explicit zero tensor here because NULL gave wrong AE gradients on one register class; round 4 found the cause -- a uniform
    #  `if (grad_is)` branch scheduled between an MFMA and the consumer of its result, psnode_dae_backward_wide.hip:add_gis -- and removed
    #  the branch, so the C ABI's documented NULL is safe for every caller)
    if fuse_de:
        if saved is not None:
            s_act, s_xst, s_ae, s_ev, s_evi = saved
            if s_ae.shape != (3, T, B, H) or s_act.shape != (T - 1, S, 3, B, H) or (n_ev and (s_ev is None or s_ev.shape != (n_ev, 3, B, H))):
                raise ValueError("saved activations do not belong to this call (shape)")
            keep += [s_act, s_xst, s_ae, s_ev, s_evi]
            a.saved_act, a.saved_xstage, a.saved_ae_act = s_act.data_ptr(), s_xst.data_ptr(), s_ae.data_ptr()
            if n_ev:
                a.saved_ev_act, a.saved_ev_i = s_ev.data_ptr(), s_evi.data_ptr()
                for q in range(3):
                    ev_rows[q] = s_ev[:, q]
            # > 0 (hidden <= 64, saved activations): the AE head's gradients are formed in the kernel too -- no head rows, no K7h
            n_ae_raw = int(lib.psnode_dae_backward_wide_ae_floats(ctypes.byref(a)))
            arows = None if n_ae_raw else [s_ae[0], s_ae[1], s_ae[2]] + [_empty((T, B, H), **f32) for _ in range(3)]
        else:
            n_ae_raw = int(lib.psnode_dae_backward_wide_ae_floats(ctypes.byref(a)))
            arows = None if n_ae_raw else [_empty((T, B, H), **f32) for _ in range(6)]
        a.k0, a.k1 = 0, T - 1
        gae_raw = agi = None
        if n_ae_raw:
            gae_raw = _empty(n_ae_raw, **f32)
            a.grad_params_ae_raw = gae_raw.data_ptr()
        else:
            agi = _empty((T, B, 16), **f32)
            for q in range(3):
                a.ae_act[q], a.ae_delta[q] = arows[q].data_ptr(), arows[3 + q].data_ptr()
            a.ae_gi = agi.data_ptr()
        with torch.cuda.device(dev):
            nbytes = lib.psnode_dae_backward_wide_workspace_bytes(ctypes.byref(a))
            ws = _empty(nbytes + 256, dtype=torch.uint8, device=dev)
            wp, wn = _aligned_ptr(ws)
            st = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(lib.psnode_dae_backward_wide_f32(ctypes.byref(a), wp, wn, st), "psnode_dae_backward_wide_f32")
            if n_ae_raw:
                # one launch did everything: unpack [dAW1 | db1 | dAW2 | db2 | dAW3 | db3 | P3 (16 slots x h) | sg (16 slots)]
                K1a, o = n + xd + nzv, 0
                gA[0] = gae_raw[o:o + Hr * K1a].view(Hr, K1a); o += Hr * K1a
                gab[0] = gae_raw[o:o + Hr]; o += Hr
                gA[1] = gae_raw[o:o + Hr * Hr].view(Hr, Hr); o += Hr * Hr
                gab[1] = gae_raw[o:o + Hr]; o += Hr
                gA[2] = gae_raw[o:o + Hr * Hr].view(Hr, Hr); o += Hr * Hr
                gab[2] = gae_raw[o:o + Hr]; o += Hr
                P3 = gae_raw[o:o + 16 * Hr].view(16, Hr); o += 16 * Hr
                sg = gae_raw[o:o + 16]
                gA[3] = P3[nzv:ne] + P3[ne + nzv:2 * ne]
                gab[3] = sg[nzv:ne] + sg[ne + nzv:2 * ne]
                g = {"z_jump": None, "v_jump": None}
                g["x_init"] = carry_x + gx_c[0]
                g["all_initial"] = ga0_de                      # (the kernel added the AE's share)
                g["z"] = gzv[..., :zd].contiguous() if zd > 0 else None
                g["v"] = gzv[..., zd:].contiguous() if vd > 0 else None
                if n_ev:
                    g["z_jump"] = gjump[..., :zd].contiguous() if zd > 0 else None
                    g["v_jump"] = gjump[..., zd:].contiguous() if vd > 0 else None
                g["de"] = _split_grads(gp_de, de_layers)
                g["ae"] = [gA[0], gab[0], gA[1], gab[1], gA[2], gab[2], gA[3], gab[3]]
                return g
            host_accumulators(False)
            # the AE head's rows -> its parameter gradients and its share of the input gradients (K7h)
            # (the heads at the grid points read the dataset rows under input_true_x; the event heads below always the running state)
            gza = head_grads_hip(arows[:3], B * H, arows[3:], agi, xt_c if x_true is not None else xs_c, zv_all)
            if nzv > 0:
                gzv += gza
            del arows, agi
            if n_ev:
                evl = event_idx.long()
                step_of = torch.zeros(n_ev, dtype=torch.long, device=dev).scatter_reduce_(
                    0, evl.clamp_min(0), torch.arange(T - 1, device=dev) * (evl >= 0), "amax")
                ev_stride = ev_rows[0].stride(0)         # B*H (own buffers) or 3*B*H (layer q of the forward call's [nE,3,B,H])
                gza = head_grads_hip(ev_rows[:3], ev_stride, ev_rows[3:], ev_gi, xs_c[step_of], jump_all.permute(1, 0, 2))
                if nzv > 0:
                    gjump += gza.permute(1, 0, 2)
        g = {"z_jump": None, "v_jump": None}
        g["x_init"] = carry_x + gx_c[0]
        g["all_initial"] = ga0_de + Sa1 @ _pad_rows(A1[:, 0:n], H)
        g["z"] = gzv[..., :zd].contiguous() if zd > 0 else None
        g["v"] = gzv[..., zd:].contiguous() if vd > 0 else None
        if n_ev:
            g["z_jump"] = gjump[..., :zd].contiguous() if zd > 0 else None
            g["v_jump"] = gjump[..., zd:].contiguous() if vd > 0 else None
        g["de"] = _split_grads(gp_de, de_layers)
        g["ae"] = [gA[0], gab[0], gA[1], gab[1], gA[2], gab[2], gA[3], gab[3]]
        return g
    host_accumulators(True)
    if chunk_steps is None:      # ~3 GB of stored rows per chunk
        chunk_steps = max(1, min(T - 1, int(3e9 // ((6 * S + 6) * 4 * B * H))))
    with torch.cuda.device(dev):
        nbytes = lib.psnode_dae_backward_wide_workspace_bytes(ctypes.byref(a))
        ws = _empty(nbytes + 256, dtype=torch.uint8, device=dev)
        wp, wn = _aligned_ptr(ws)
        st = torch.cuda.current_stream(dev).cuda_stream
        for k1 in range(T - 1, 0, -chunk_steps):
            k0 = max(0, k1 - chunk_steps)
            Tc = k1 - k0
            rows = [_empty((Tc, S, B, H), **f32) for _ in range(6)]
            arows = [_empty((Tc + 1, B, H), **f32) for _ in range(6)]
            agi = _empty((Tc + 1, B, 16), **f32)
            gk = _empty((Tc, S, B, xd), **f32)
            Xs = _empty((Tc, S, B, xd), **f32)
            dsum = [_empty((Tc, B, H), **f32) for _ in range(3)]
            a.k0, a.k1 = k0, k1
            for q in range(3):
                a.act[q], a.delta[q], a.dsum[q] = rows[q].data_ptr(), rows[3 + q].data_ptr(), dsum[q].data_ptr()
                a.ae_act[q], a.ae_delta[q] = arows[q].data_ptr(), arows[3 + q].data_ptr()
            a.gk, a.xstage, a.ae_gi = gk.data_ptr(), Xs.data_ptr(), agi.data_ptr()
            _lib.check(lib.psnode_dae_backward_wide_f32(ctypes.byref(a), wp, wn, st), "psnode_dae_backward_wide_f32")
            # ---- DE
            h1, h2, h3, d1, d2, d3 = (r.view(-1, H) for r in rows)
            G = Tc * S
            gW[3] += _gemm_tn(gk.view(-1, xd), h3, G)[:, :Hr]; gb[3] += gk.view(-1, xd).sum(0)
            gW[2] += _gemm_tn(d3, h2, G)[:Hr, :Hr]; gb[2] += dsum[2].sum((0, 1))[:Hr]
            gW[1] += _gemm_tn(d2, h1, G)[:Hr, :Hr]; gb[1] += dsum[1].sum((0, 1))[:Hr]
            # L1 input per (step, stage): cat(a0, s - a0, s), s = cat(X_s, z|v|i of the step -- jump values and recomputed i0 at events)
            ext = torch.cat((zv_all[k0:k1], is_c[k0:k1]), -1)                                  # [Tc, B, ne]
            if n_ev:
                evc = event_idx[k0:k1].long()
                hit = (evc >= 0).view(Tc, 1, 1)
                evi = evc.clamp_min(0)
                ext_ev = torch.cat((jump_all[:, evi].permute(1, 0, 2), ev_i[evi][..., ne + nzv:2 * ne]), -1)
                ext = torch.where(hit, ext_ev, ext)
            s_in = torch.cat((Xs, ext.unsqueeze(1).expand(Tc, S, B, ne)), -1)
            U = torch.cat((a0.view(1, 1, B, n).expand(Tc, S, B, n), s_in - a0, s_in), -1).reshape(-1, 3 * n)
            gW[0] += _gemm_tn(d1, U, G)[:Hr]
            D1 = dsum[0]                                                                         # [Tc, B, H]
            D1s = D1.sum(0)
            S1 += D1s; gb[0] += D1s.sum(0)[:Hr]
            if nzv > 0:
                gc = (D1.reshape(-1, H) @ Fe).view(Tc, B, nzv)
                if n_ev:
                    gjump.index_add_(1, evi, (gc * hit).permute(1, 0, 2))
                    gc = torch.where(hit, torch.zeros_like(gc), gc)
                gzv[k0:k1] += gc
            del rows, gk, Xs, U, D1, s_in, dsum
            # ---- AE heads at grid points k0+lo .. k1 (row r of the chunk = grid point k0 + r)
            lo = 0 if k0 == 0 else 1
            gza = head_grads([r[lo:] for r in arows[:3]], [r[lo:] for r in arows[3:]], agi[lo:], xs_c[k0 + lo:k1 + 1], zv_all[k0 + lo:k1 + 1])
            if nzv > 0:
                gzv[k0 + lo:k1 + 1] += gza
            del arows, agi
        if n_ev:    # event-time heads g(x_k; jumps): x of the step that takes the event (events no step takes have zero rows)
            evl = event_idx.long()
            step_of = torch.zeros(n_ev, dtype=torch.long, device=dev).scatter_reduce_(
                0, evl.clamp_min(0), torch.arange(T - 1, device=dev) * (evl >= 0), "amax")
            gza = head_grads(ev_rows[:3], ev_rows[3:], ev_gi, xs_c[step_of], jump_all.permute(1, 0, 2))
            if nzv > 0:
                gjump += gza.permute(1, 0, 2)
    g = {"z_jump": None, "v_jump": None}
    g["x_init"] = carry_x + gx_c[0]
    ga0 = S1 @ _pad_rows(W1[:, 0:n] - W1[:, n:2 * n], H) + Sa1 @ _pad_rows(A1[:, 0:n], H)
    g["all_initial"] = ga0
    g["z"] = gzv[..., :zd].contiguous() if zd > 0 else None
    g["v"] = gzv[..., zd:].contiguous() if vd > 0 else None
    if n_ev:
        g["z_jump"] = gjump[..., :zd].contiguous() if zd > 0 else None
        g["v_jump"] = gjump[..., zd:].contiguous() if vd > 0 else None
    g["de"] = [gW[0], gb[0], gW[1], gb[1], gW[2], gb[2], gW[3], gb[3]]
    g["ae"] = [gA[0], gab[0], gA[1], gab[1], gA[2], gab[2], gA[3], gab[3]]
    return g


def dae_backward(method: str, de_layers: Layers, ae_layers: Layers, t, z, v, all_initial, xs, is_, grad_xs, grad_is, event_idx=None,
                 z_jump=None, v_jump=None, kernel: str = "auto", saved=None):
    """Backward pass of `dae_integrate` (no teacher forcing): the one-launch MFMA backward (K7) for the DAE_01 shape class at hidden 64,
    the fused-DE sweep K7f (`dae_backward_wide`) at the other hidden widths <= 128, else the generic backward kernel (K5);
    `kernel` = "auto" | "mfma" | "generic" | "wide" (K7f at any width <= 128) | "split" (round 2's K7w + library GEMMs).
    saved = what `dae_integrate(save=True)` returned (read by K7f and, for the latent shapes at hidden 64, K9; K7 / K8 / K5 recompute
    and refuse them).
    Returns dict(x_init, z, v, z_jump, v_jump, all_initial, de=[...], ae=[...]) of gradients."""
    lib = _lib.load()
    dev = xs.device
    T, B, xd = xs.shape
    zd, vd, idim = z.shape[-1], v.shape[-1], is_.shape[-1]
    # hidden 32 / 128 (no one-launch MFMA backward): the adjoint sweep on K7w + library GEMMs instead of the generic K5
    if kernel in ("wide", "split") and T < 2:
        kernel = "generic"       # no step to sweep: the split backward has no head-only form, K5 handles the single grid point
    if kernel == "split":
        return dae_backward_wide(method, de_layers, ae_layers, t, z, v, all_initial, xs, is_, grad_xs, grad_is, event_idx=event_idx,
                                 z_jump=z_jump, v_jump=v_jump, fuse_de=False)
    if saved is not None and latent_wide_shape(de_layers, ae_layers, xd, zd, vd, idim):
        return latent_backward_wide(method, de_layers, ae_layers, t, z, v, all_initial, xs, is_, grad_xs, grad_is, event_idx=event_idx,
                                    z_jump=z_jump, v_jump=v_jump, saved=saved)
    if kernel == "wide" or (kernel == "auto" and len(de_layers) == 4 and (de_layers[0][0].shape[0] != 64 or saved is not None) and T >= 2
                            and dae_backward_wide_supported(method, de_layers, ae_layers, xd, zd, vd, idim)):
        return dae_backward_wide(method, de_layers, ae_layers, t, z, v, all_initial, xs, is_, grad_xs, grad_is, event_idx=event_idx,
                                 z_jump=z_jump, v_jump=v_jump, saved=saved)
    keep: list = []
    a = _lib.DaeBwdArgsF32()
    a.method = METHOD_ID[method]
    a.kernel = KERNEL_ID[kernel]
    a.x_dim, a.z_dim, a.v_dim, a.i_dim, a.T, a.B = xd, zd, vd, idim, T, B
    a.de, a.ae = _mlp(de_layers, dev, "de", keep), _mlp(ae_layers, dev, "ae", keep)
    z, v, z_jump, v_jump = _aligned16(z), _aligned16(v), _aligned16(z_jump), _aligned16(v_jump)
    a.t, a.z, a.v = _view(t, dev, "t", keep), _view(z, dev, "z", keep), _view(v, dev, "v", keep)
    a0 = _f32_dev(all_initial, dev, "all_initial").contiguous()
    xs_c, is_c = _f32_dev(xs, dev, "xs").contiguous(), _f32_dev(is_, dev, "is").contiguous()
    gx_c = _f32_dev(grad_xs, dev, "grad_xs").contiguous() if grad_xs is not None else torch.zeros_like(xs_c)
    gi_c = _f32_dev(grad_is, dev, "grad_is").contiguous() if grad_is is not None else None
    keep += [a0, xs_c, is_c, gx_c, gi_c]
    a.all_initial, a.xs, a.is_, a.grad_xs = a0.data_ptr(), xs_c.data_ptr(), is_c.data_ptr(), gx_c.data_ptr()
    a.grad_is = gi_c.data_ptr() if gi_c is not None else None
    g = {"z_jump": None, "v_jump": None}
    if event_idx is not None:
        keep.append(event_idx)
        a.event_idx = event_idx.data_ptr()
        a.z_jump, a.zj_stride_b, a.zj_stride_e = _jump(z_jump, dev, "z_jump", keep)
        a.v_jump, a.vj_stride_b, a.vj_stride_e = _jump(v_jump, dev, "v_jump", keep)
        n_ev = (z_jump if z_jump is not None else v_jump).shape[1]
        a.n_events = n_ev
        if zd > 0:
            g["z_jump"] = torch.zeros((B, n_ev, zd), dtype=torch.float32, device=dev)
            a.grad_z_jump = g["z_jump"].data_ptr()
        if vd > 0:
            g["v_jump"] = torch.zeros((B, n_ev, vd), dtype=torch.float32, device=dev)
            a.grad_v_jump = g["v_jump"].data_ptr()
    with torch.cuda.device(dev):
        g["x_init"] = _empty((B, xd), dtype=torch.float32, device=dev)
        g["all_initial"] = _empty((B, xd + zd + vd + idim), dtype=torch.float32, device=dev)
        g["z"] = _empty((T, B, zd), dtype=torch.float32, device=dev) if zd > 0 else None
        g["v"] = _empty((T, B, vd), dtype=torch.float32, device=dev) if vd > 0 else None
        npd = sum(w.numel() + b.numel() for w, b in de_layers)
        npa = sum(w.numel() + b.numel() for w, b in ae_layers)
        gde = _empty(npd, dtype=torch.float32, device=dev)
        gae = _empty(npa, dtype=torch.float32, device=dev)
        a.grad_x_init, a.grad_all_initial = g["x_init"].data_ptr(), g["all_initial"].data_ptr()
        a.grad_z = g["z"].data_ptr() if g["z"] is not None else None
        a.grad_v = g["v"].data_ptr() if g["v"] is not None else None
        a.grad_params_de, a.grad_params_ae = gde.data_ptr(), gae.data_ptr()
        if saved is not None and T >= 2:        # (K9 reads them; the C side refuses them for the kernels that recompute)
            s_act, s_xst, s_ae, s_ev, s_evi = saved
            L = len(de_layers) - 1
            _check_saved(s_act, s_xst, T, B, xd, {"euler": 1, "midpoint": 2, "rk4": 4}[method], L, dev)
            if tuple(s_ae.shape[:3]) != (L, T, B) or s_ae.shape[-1] != s_act.shape[-1] or not s_ae.is_contiguous() or s_ae.device != dev:
                raise ValueError("saved AE activations do not belong to this call (shape / device)")
            keep += [s_act, s_xst, s_ae, s_ev, s_evi]
            a.saved_act, a.saved_xstage, a.saved_ae_act = s_act.data_ptr(), s_xst.data_ptr(), s_ae.data_ptr()
            if event_idx is not None:
                n_ev_ = (z_jump if z_jump is not None else v_jump).shape[1]
                if s_ev is None or s_evi is None or s_ev.shape[0] != n_ev_ or s_ev.shape[2] != B or s_evi.shape[:2] != (n_ev_, B):
                    raise ValueError("saved event activations do not belong to this call (shape)")
                a.saved_ev_act, a.saved_ev_i = s_ev.data_ptr(), s_evi.data_ptr()
        nbytes = lib.psnode_dae_backward_workspace_bytes(ctypes.byref(a))
        ws = _empty(nbytes + 256, dtype=torch.uint8, device=dev)
        wp, wn = _aligned_ptr(ws)
        rc = lib.psnode_dae_backward_f32(ctypes.byref(a), wp, wn, torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "psnode_dae_backward_f32")
    g["de"], g["ae"] = _split_grads(gde, de_layers), _split_grads(gae, ae_layers)
    return g


def ode_backward(method: str, de_layers: Layers, t, z, all_initial, xs, grad_xs, event_idx=None, z_jump=None, need_grad_z: bool = True,
                 kernel: str = "auto", saved=None, input_true_x: bool = False):
    """Backward pass of `ode_integrate` in one launch.  `saved` = what `ode_integrate(save=True)` returned next to
    xs: K4f then skips the recompute of the stage evaluations.  input_true_x: backward of a teacher-forced call (my_solvers.py:72-74) --
    `xs` must then be the DATASET x the forward call started every step from; K4f (hidden <= 128, x_dim <= 8) only.
    Returns (grad_x0 [B,xd], grad_z [T,B,zd] | None, grad_z_jump | None, grad_all_initial [B,n], [grad W1, b1, ..., W4, b4])."""
    lib = _lib.load()
    dev = xs.device
    T, B, xd = xs.shape
    zd = z.shape[-1]
    # kernel: "auto" = K4 at hidden 64 (z_dim <= 4), the width-generic one-launch K4f at every other width <= 128 (z_dim <= 8), else the
    # generic K5; "wide" forces K4f; "split" is round 2's route for those widths (adjoint sweep K4w + library GEMMs over stored rows),
    # kept as an A/B arm (profiles/scripts/wide_vs_onelaunch.py)
    if kernel == "split":
        return ode_backward_wide(method, de_layers, t, z, all_initial, xs, grad_xs, event_idx=event_idx, z_jump=z_jump,
                                 need_grad_z=need_grad_z)
    if saved is not None and not input_true_x and latent_wide_shape(de_layers, None, xd, zd):
        g = latent_backward_wide(method, de_layers, None, t, z, None, all_initial, xs, None, grad_xs, None, event_idx=event_idx,
                                 z_jump=z_jump, saved=saved, need_grad_z=need_grad_z)
        return g["x_init"], g["z"], g["z_jump"], g["all_initial"], g["de"]
    keep: list = []
    a = _bwd_args(method, de_layers, xd, zd, T, B, dev, keep, kernel)
    if input_true_x:
        if saved is not None:
            raise ValueError("a teacher-forced forward saves no activations")
        a.flags = _lib.FLAG_INPUT_TRUE_X
    z, z_jump = _aligned16(z), _aligned16(z_jump)
    a.t = _view(t, dev, "t", keep)
    a.z = _view(z, dev, "z", keep)
    a0 = _f32_dev(all_initial, dev, "all_initial").contiguous()
    xs_c = _f32_dev(xs, dev, "xs").contiguous()
    g_c = _f32_dev(grad_xs, dev, "grad_xs").contiguous()
    keep += [a0, xs_c, g_c]
    a.all_initial, a.xs, a.grad_xs = a0.data_ptr(), xs_c.data_ptr(), g_c.data_ptr()
    gzj = None
    if event_idx is not None:
        keep.append(event_idx)
        a.event_idx = event_idx.data_ptr()
        a.z_jump, a.zj_stride_b, a.zj_stride_e = _jump(z_jump, dev, "z_jump", keep)
        if z_jump is not None and zd > 0:
            a.n_events = z_jump.shape[1]
            gzj = torch.zeros((B, z_jump.shape[1], zd), dtype=torch.float32, device=dev)
            a.grad_z_jump = gzj.data_ptr()
    with torch.cuda.device(dev):
        gx0 = _empty((B, xd), dtype=torch.float32, device=dev)
        ga0 = _empty((B, xd + zd), dtype=torch.float32, device=dev)
        gz = _empty((T, B, zd), dtype=torch.float32, device=dev) if (need_grad_z and zd > 0) else None
        npar = lib.psnode_ode_backward_param_count(ctypes.byref(a))
        gpar = _empty(npar, dtype=torch.float32, device=dev)
        a.grad_x0, a.grad_all_initial, a.grad_params = gx0.data_ptr(), ga0.data_ptr(), gpar.data_ptr()
        a.grad_z = gz.data_ptr() if gz is not None else None
        if saved is not None and T >= 2:
            _check_saved(saved[0], saved[1], T, B, xd, {"euler": 1, "midpoint": 2, "rk4": 4}[method], len(de_layers) - 1, dev)
            keep += [saved[0], saved[1]]
            a.saved_act, a.saved_xstage = saved[0].data_ptr(), saved[1].data_ptr()
        nbytes = lib.psnode_ode_backward_workspace_bytes(ctypes.byref(a))
        ws = _empty(nbytes + 256, dtype=torch.uint8, device=dev)
        wp, wn = _aligned_ptr(ws)
        rc = lib.psnode_ode_backward_f32(ctypes.byref(a), wp, wn, torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "psnode_ode_backward_f32")
    return gx0, gz, gzj, ga0, _split_grads(gpar, de_layers)


def mlp_rows(layers: Layers, inp: torch.Tensor) -> torch.Tensor:
    """Fused `nn.Sequential(Linear, ELU, Linear)` over the last dim of `inp` (any leading shape) on the HIP row kernel:
    the encoders / decoders of the direct_encode models (neural_00_ODE_02_direct_encode.py:64-69)."""
    lib = _lib.load()
    dev = inp.device
    keep: list = []
    m = _mlp(layers, dev, "mlp", keep)
    if not lib.psnode_mlp_rows_supported(ctypes.byref(m)):
        raise ValueError("mlp_rows: needs Linear(in, H) ELU Linear(H, out) with H in {16, 64}, in <= 16 (or in = H = 64), out <= 16 (or out = H)")
    x = _f32_dev(inp, dev, "input")
    if x.shape[-1] != m.in_dim:
        raise ValueError(f"mlp_rows: input width {x.shape[-1]}, expected {m.in_dim}")
    x2 = x.reshape(-1, x.shape[-1])
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    out = _empty((*x.shape[:-1], layers[-1][0].shape[0]), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.psnode_mlp_rows_f32(ctypes.byref(m), x2.shape[0], x2.data_ptr(), x2.stride(0), out.data_ptr(), out.shape[-1],
                                     torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "psnode_mlp_rows_f32")
    return out


def ode_encoded_supported(x_encoder: Layers, z_encoder: Layers, x_decoder: Layers, de_layers: Layers) -> bool:
    """Shapes of the fused direct_encode ODE forward (psnode_ode_encoded_integrate_f32): every MLP 2 layers with hidden 16."""
    try:
        shapes = [(l[0][0].shape, l[1][0].shape) for l in (x_encoder, z_encoder, x_decoder, de_layers)]
    except (IndexError, TypeError):
        return False
    if any(len(l) != 2 for l in (x_encoder, z_encoder, x_decoder, de_layers)):
        return False
    (xe1, xe2), (ze1, ze2), (xd1, xd2), (de1, de2) = shapes
    H = 16
    return (xe1[0] == H and tuple(xe2) == (H, H) and ze1[0] == H and tuple(ze2) == (H, H) and tuple(xd1) == (H, H) and xd2[1] == H
            and xd2[0] == xe1[1] and tuple(de1) == (H, 6 * H) and tuple(de2) == (H, H) and 1 <= xe1[1] <= H and 1 <= ze1[1] <= H)


def ode_encoded_integrate(method: str, x_encoder: Layers, z_encoder: Layers, x_decoder: Layers, de_layers: Layers, t, x, z,
                          event_t=None, z_jump=None, event_idx=None, want_recon: bool = True, want_latent: bool = False,
                          check_events: bool = False):
    """The whole ODE_Model.forward of neural_00_ODE_02_direct_encode.py:74-89 in ONE launch (hidden_dim 16): encoders, latent
    integrate_ODE, decoder of the solution and the reconstruction x_decoder(x_encoder(x)).  t, x, z are the scripts' B-major
    tensors [B,T,*] (any strides with a contiguous last dim); z_jump is the RAW [B,nE,z_dim] tensor.  Returns
    (x_pred [B,T,xd] as the permuted view of a time-major buffer -- like the script --, x_re [B,T,xd] or None, Xh_sol [T,B,16] or None)."""
    lib = _lib.load()
    dev = x.device
    keep: list = []
    a = _lib.OdeEncodedArgsF32()
    a.method = METHOD_ID[method]
    B, T, xd = x.shape
    zd = z.shape[-1]
    if t.shape[:2] != (B, T) or z.shape[:2] != (B, T):
        raise ValueError(f"ode_encoded_integrate: t {tuple(t.shape)}, x {tuple(x.shape)}, z {tuple(z.shape)} disagree on [B,T]")
    a.x_dim, a.z_dim, a.T, a.B = xd, zd, T, B
    a.x_encoder = _mlp(x_encoder, dev, "x_encoder", keep)
    a.z_encoder = _mlp(z_encoder, dev, "z_encoder", keep)
    a.x_decoder = _mlp(x_decoder, dev, "x_decoder", keep)
    a.de = _mlp(de_layers, dev, "de", keep)
    if not lib.psnode_ode_encoded_supported(ctypes.byref(a)):
        raise _lib.UnsupportedShapeError("ode_encoded_integrate: needs x_encoder xd->16->16, z_encoder zd->16->16, x_decoder 16->16->xd, de 96->16->16")
    a.t = _view(t.permute(1, 0, 2), dev, "t", keep)
    a.x = _view(x.permute(1, 0, 2), dev, "x", keep)
    a.z = _view(z.permute(1, 0, 2), dev, "z", keep)
    if event_idx is None and event_t is not None and event_t.shape[1] > 0 and T > 1:
        event_idx = event_table(t.permute(1, 0, 2), event_t, check_duplicates=check_events)
    if event_idx is not None:
        if z_jump is None:
            raise ValueError("ode_encoded_integrate: events need z_jump")
        if z_jump.shape[0] != B or z_jump.shape[-1] != zd:
            raise ValueError(f"ode_encoded_integrate: z_jump {tuple(z_jump.shape)} does not match [B={B}, nE, zd={zd}]")
        keep.append(event_idx)
        a.event_idx = event_idx.data_ptr()
        a.z_jump, a.zj_stride_b, a.zj_stride_e = _jump(z_jump, dev, "z_jump", keep)
    x_pred = _empty((T, B, xd), dtype=torch.float32, device=dev)
    a.x_pred = x_pred.data_ptr()
    x_re = xh = None
    if want_recon:
        x_re = _empty((B, T, xd), dtype=torch.float32, device=dev)
        a.x_re, a.xre_stride_t, a.xre_stride_b = x_re.data_ptr(), xd, T * xd
    if want_latent:
        xh = _empty((T, B, 16), dtype=torch.float32, device=dev)
        a.xh_out = xh.data_ptr()
    with torch.cuda.device(dev):
        rc = lib.psnode_ode_encoded_integrate_f32(ctypes.byref(a), torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "psnode_ode_encoded_integrate_f32")
    return x_pred.permute(1, 0, 2), x_re, xh


def _dae_encoded_args(mlps, dev, keep, xd, zd, vd, idim):
    a = _lib.DaeEncodedArgsF32()
    a.x_dim, a.z_dim, a.v_dim, a.i_dim = xd, zd, vd, idim
    names = ("x_encoder", "z_encoder", "v_encoder", "i_encoder", "x_decoder", "i_decoder", "de", "ae")
    for name, m in zip(names, mlps):
        if m is not None:
            setattr(a, name, _mlp(m, dev, name, keep))
    return a


def dae_encoded_supported(x_encoder, z_encoder, v_encoder, i_encoder, x_decoder, i_decoder, de_layers, ae_layers) -> bool:
    """Shapes of the fused direct_encode DAE forward (psnode_dae_encoded_integrate_f32, hidden_dim 64; z_encoder None = z_dim 0)."""
    mlps = (x_encoder, z_encoder, v_encoder, i_encoder, x_decoder, i_decoder, de_layers, ae_layers)
    try:
        if any(m is not None and len(m) != 2 for m in mlps) or any(m is None for k, m in enumerate(mlps) if k != 1):
            return False
        dev = x_encoder[0][0].device
        if dev.type != "cuda":
            return False
        xd, vd, idim = x_encoder[0][0].shape[1], v_encoder[0][0].shape[1], i_encoder[0][0].shape[1]
        zd = z_encoder[0][0].shape[1] if z_encoder is not None else 0
        a = _dae_encoded_args(mlps, dev, [], xd, zd, vd, idim)
    except (IndexError, TypeError, ValueError):
        return False
    return bool(_lib.load().psnode_dae_encoded_supported(ctypes.byref(a)))


def dae_encoded_integrate(method: str, x_encoder, z_encoder, v_encoder, i_encoder, x_decoder, i_decoder, de_layers, ae_layers,
                          x0, t, x, z, v, i, event_t=None, z_jump=None, v_jump=None, event_idx=None, want_recon: bool = True,
                          check_events: bool = False):
    """The whole DAE_Model.forward of neural_01_DAE_02_direct_encode.py:125-153 in ONE launch (hidden_dim 64): the four encoders,
    all_initial, the latent integrate_DAE, both decoders of the solution and the two reconstructions.  x0 [B,xd] is Init_Func's output;
    t, x, z, v, i are the scripts' B-major tensors [B,T,*] (z of width 0 when the model has no z_encoder); z_jump / v_jump the RAW
    [B,nE,*] tensors.  Returns (x_pred, i_pred, x_re, i_re), each [B,T,*] as the permuted view of a time-major buffer for the
    predictions (like the script) and B-major contiguous for the reconstructions (None without want_recon)."""
    lib = _lib.load()
    dev = v.device
    keep: list = []
    B, T, vd = v.shape
    xd, idim, zd = x0.shape[-1], i.shape[-1], (z.shape[-1] if z is not None else 0)
    if z_encoder is None:
        zd = 0
    mlps = (x_encoder, z_encoder, v_encoder, i_encoder, x_decoder, i_decoder, de_layers, ae_layers)
    a = _dae_encoded_args(mlps, dev, keep, xd, zd, vd, idim)
    a.method, a.T, a.B = METHOD_ID[method], T, B
    if not lib.psnode_dae_encoded_supported(ctypes.byref(a)):
        raise _lib.UnsupportedShapeError("dae_encoded_integrate: needs encoders in->64->64 (x <= 16, z | v | i <= 8 wide), decoders "
                                         "64->64->out, de 12H|9H->64->64, ae 7H|5H->64->64")
    for name, q, wdt in (("t", t, 1), ("x", x, xd), ("v", v, vd), ("i", i, idim)) + ((("z", z, zd),) if zd else ()):
        if q is None or q.shape[:2] != (B, T) or q.shape[-1] != wdt:
            raise ValueError(f"dae_encoded_integrate: {name} {None if q is None else tuple(q.shape)} does not match [B={B}, T={T}, {wdt}]")
    a.t = _view(t.permute(1, 0, 2), dev, "t", keep)
    a.x = _view(x.permute(1, 0, 2), dev, "x", keep)
    a.v = _view(v.permute(1, 0, 2), dev, "v", keep)
    a.i = _view(i.permute(1, 0, 2), dev, "i", keep)
    if zd:
        a.z = _view(z.permute(1, 0, 2), dev, "z", keep)
    x0c = _f32_dev(x0, dev, "x0").contiguous()
    if x0c.shape != (B, xd):
        raise ValueError(f"dae_encoded_integrate: x0 {tuple(x0c.shape)} does not match [B={B}, xd={xd}]")
    keep.append(x0c)
    a.x0 = x0c.data_ptr()
    if event_idx is None and event_t is not None and event_t.shape[1] > 0 and T > 1:
        event_idx = event_table(t.permute(1, 0, 2), event_t, check_duplicates=check_events)
    if event_idx is not None:
        _check_jump("v_jump", v_jump, B, vd, event_idx)
        keep.append(event_idx)
        a.event_idx = event_idx.data_ptr()
        a.v_jump, a.vj_stride_b, a.vj_stride_e = _jump(v_jump, dev, "v_jump", keep)
        if zd:
            _check_jump("z_jump", z_jump, B, zd, event_idx)
            a.z_jump, a.zj_stride_b, a.zj_stride_e = _jump(z_jump, dev, "z_jump", keep)
    with torch.cuda.device(dev):
        x_pred = _empty((T, B, xd), dtype=torch.float32, device=dev)
        i_pred = _empty((T, B, idim), dtype=torch.float32, device=dev)
        a.x_pred, a.i_pred = x_pred.data_ptr(), i_pred.data_ptr()
        x_re = i_re = None
        if want_recon:
            x_re = _empty((B, T, xd), dtype=torch.float32, device=dev)
            i_re = _empty((B, T, idim), dtype=torch.float32, device=dev)
            a.x_re, a.xre_stride_t, a.xre_stride_b = x_re.data_ptr(), xd, T * xd
            a.i_re, a.ire_stride_t, a.ire_stride_b = i_re.data_ptr(), idim, T * idim
        ws = _empty(lib.psnode_dae_encoded_workspace_bytes(ctypes.byref(a)) + 256, dtype=torch.uint8, device=dev)
        wp, wn = _aligned_ptr(ws)
        rc = lib.psnode_dae_encoded_integrate_f32(ctypes.byref(a), wp, wn, torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "psnode_dae_encoded_integrate_f32")
    return x_pred.permute(1, 0, 2), i_pred.permute(1, 0, 2), x_re, i_re


def mlp_rows_backward(layers: Layers, inp: torch.Tensor, grad_out: torch.Tensor, need_grad_in: bool = True):
    """Backward of `mlp_rows`: returns (grad_in or None, [dW1, db1, dW2, db2]) from the saved input and grad_out (row kernel)."""
    lib = _lib.load()
    dev = inp.device
    keep: list = []
    m = _mlp(layers, dev, "mlp", keep)
    if not lib.psnode_mlp_rows_supported(ctypes.byref(m)):
        raise ValueError("mlp_rows_backward: unsupported MLP shape")
    x2 = _f32_dev(inp, dev, "input").reshape(-1, inp.shape[-1])
    g2 = _f32_dev(grad_out, dev, "grad_out").reshape(-1, grad_out.shape[-1])
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    if g2.stride(-1) != 1:
        g2 = g2.contiguous()
    if g2.shape[0] != x2.shape[0] or g2.shape[1] != layers[-1][0].shape[0]:
        raise ValueError(f"mlp_rows_backward: grad_out {tuple(grad_out.shape)} does not match input {tuple(inp.shape)}")
    rows = x2.shape[0]
    with torch.cuda.device(dev):
        gin = _empty((*inp.shape[:-1], inp.shape[-1]), dtype=torch.float32, device=dev) if need_grad_in else None
        npar = sum(w.numel() + b.numel() for w, b in layers)
        gp = _empty(npar, dtype=torch.float32, device=dev)
        nbytes = lib.psnode_mlp_rows_backward_workspace_bytes(ctypes.byref(m), rows)
        ws = _empty(nbytes + 256, dtype=torch.uint8, device=dev)
        wp, wn = _aligned_ptr(ws)
        rc = lib.psnode_mlp_rows_backward_f32(ctypes.byref(m), rows, x2.data_ptr(), x2.stride(0), g2.data_ptr(), g2.stride(0),
                                              gin.data_ptr() if gin is not None else None, inp.shape[-1], gp.data_ptr(), wp, wn,
                                              torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "psnode_mlp_rows_backward_f32")
    return gin, _split_grads(gp, layers)


class _RowsMlp(torch.autograd.Function):
    """mlp_rows with a fused backward: forward saves only the input rows (h is recomputed in the backward kernel)."""

    @staticmethod
    def forward(ctx, inp, w1, b1, w2, b2):
        ctx.save_for_backward(inp, w1, b1, w2, b2)
        return mlp_rows([(w1, b1), (w2, b2)], inp)

    @staticmethod
    def backward(ctx, grad_out):
        inp, w1, b1, w2, b2 = ctx.saved_tensors
        gin, gp = mlp_rows_backward([(w1.detach(), b1.detach()), (w2.detach(), b2.detach())], inp.detach(), grad_out,
                                    need_grad_in=ctx.needs_input_grad[0])
        return (gin, *gp)


def mlp_rows_autograd(seq, inp: torch.Tensor) -> torch.Tensor:
    """`seq(inp)` for a recognised Linear-ELU-Linear on the row kernels, differentiable w.r.t. the input and the parameters."""
    lin = [m for m in seq if isinstance(m, nn.Linear)]
    return _RowsMlp.apply(inp, lin[0].weight, lin[0].bias, lin[1].weight, lin[1].bias)


def rows_layers_of(seq, inp: torch.Tensor, allow_grad: bool = False):
    """Layers if `seq(inp)` can run on the row kernel (2-layer ELU-MLP, hidden 16 / 64, fp32 HIP tensor); with autograd in play
    only when `allow_grad` (the caller then goes through mlp_rows_autograd)."""
    if inp.device.type != "cuda" or inp.dtype != torch.float32 or inp.numel() == 0:
        return None
    layers = sequential_layers(seq)
    if layers is None or len(layers) != 2:
        return None
    H, din, dout = layers[0][0].shape[0], layers[0][0].shape[1], layers[1][0].shape[0]
    if H not in (16, 64) or not (din <= 16 or (din == 64 and H == 64)) or not (dout <= 16 or dout == H):
        return None
    if not allow_grad and _needs_autograd([inp] + [p for wb in layers for p in wb]):
        return None
    return layers


# ----------------------------------------------------------------------------- planning for the solver classes
def _event_tensors(event_fn, jump_change_fn, want_v: bool):
    """(ok, event_t, z_jump, v_jump).  Events can be fused when both callbacks are the bound methods of one
    ODE_Event / DAE_Event-like object (attributes event_t, z_jump[, v_jump]) -- neural_base.py:43-65,169-196."""
    if event_fn is None:
        return True, None, None, None       # my_solvers.py:70: `event_fn is not None and ...`
    ev = getattr(event_fn, "__self__", None)
    if ev is None or getattr(jump_change_fn, "__self__", None) is not ev:
        return False, None, None, None
    if getattr(event_fn, "__name__", "") != "event_fn" or getattr(jump_change_fn, "__name__", "") != "jump_change_fn":
        return False, None, None, None
    if not getattr(type(ev), "_psnode_event", False):
        return False, None, None, None
    if ev.event_t is None:
        return True, None, None, None       # neural_base.py:53
    return True, ev.event_t, ev.z_jump, (getattr(ev, "v_jump", None) if want_v else None)


def _needs_autograd(tensors) -> bool:
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def _all_f32_on(dev, *tensors) -> bool:
    return all(a is None or (torch.is_tensor(a) and a.dtype == torch.float32 and a.device == dev) for a in tensors)


def plan_ode(x_func, x, z, all_initial, event_fn, jump_change_fn, t=None):
    """None if this integrate_ODE call cannot run fused, else (de_layers, event_t, z_jump, needs_autograd)."""
    if x.device.type != "cuda" or x.dtype != torch.float32 or x.dim() != 3 or z.dim() != 3:
        return None
    if not _all_f32_on(x.device, z, all_initial, t):      # mixed dtypes / devices: 'auto' promises the walk, not a TypeError
        return None
    xd, zd = x.shape[-1], z.shape[-1]
    if all_initial.dim() != 2 or all_initial.shape[-1] != xd + zd:
        return None
    layers = de_layers_of(x_func, xd + zd, xd)
    if layers is None or not _recipe_ok(x_func, layers, "de_ode", (xd, zd)):
        return None
    ok, event_t, z_jump, _ = _event_tensors(event_fn, jump_change_fn, False)
    if not ok or not _all_f32_on(x.device, event_t, z_jump):
        return None
    needs_grad = _needs_autograd([x, z, all_initial, z_jump] + [p for wb in layers for p in wb])
    return layers, event_t, z_jump, needs_grad


def plan_dae(x_init, x_func, i_func, z, v, i, all_initial, event_fn, jump_change_fn, t=None):
    if x_init.device.type != "cuda" or x_init.dtype != torch.float32 or z.dim() != 3:
        return None
    if not _all_f32_on(x_init.device, z, v, all_initial, t):
        return None
    xd, zd, vd, idim = x_init.shape[-1], z.shape[-1], v.shape[-1], i.shape[-1]
    n = xd + zd + vd + idim
    if all_initial.dim() != 2 or all_initial.shape[-1] != n:
        return None
    de = de_layers_of(x_func, n, xd)
    ae = ae_layers_of(i_func, n, xd + zd + vd, idim)
    if de is None or ae is None:
        return None
    if not _recipe_ok(x_func, de, "de_dae", (xd, zd, vd, idim)) or not _recipe_ok(i_func, ae, "ae", (xd, zd, vd, n)):
        return None
    ok, event_t, z_jump, v_jump = _event_tensors(event_fn, jump_change_fn, True)
    if not ok or not _all_f32_on(x_init.device, event_t, z_jump, v_jump):
        return None
    needs_grad = _needs_autograd([x_init, z, v, all_initial, z_jump, v_jump] + [p for wb in list(de) + list(ae) for p in wb])
    return de, ae, event_t, z_jump, v_jump, needs_grad
