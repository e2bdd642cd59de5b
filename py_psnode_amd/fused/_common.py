"""Shared host-side plumbing of the fused HIP integrator: recognising the reference's MLP right-hand sides, marshalling tensors into the
C ABI structs (include/psnode_hip.h), the device-side event table, workspaces.  Nothing here computes on the CPU and nothing here
imports oracle/."""
import ctypes
import os
import weakref
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from .. import _lib

Layers = Sequence[Tuple[torch.Tensor, torch.Tensor]]


METHOD_ID = {"euler": _lib.EULER, "midpoint": _lib.MIDPOINT, "rk4": _lib.RK4_38}


KERNEL_ID = {"auto": _lib.KERNEL_AUTO, "generic": _lib.KERNEL_GENERIC, "mfma": _lib.KERNEL_MFMA, "wide": _lib.KERNEL_MFMA_WIDE,
             "tile": _lib.KERNEL_MFMA_TILE, "wave": _lib.KERNEL_MFMA_WAVE}      # forward ODE calls: K1 (4-wave tile) / K1x (one wave per 4 trajectories)


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


def _padded_hidden(h: int) -> int:
    """Width class the MFMA kernels run a hidden width at (csrc/psnode_pack.h: padded_hidden): rows they store have this many columns,
    the ones beyond `h` are exact zeros."""
    return 32 if h <= 32 else (64 if h <= 64 else 128)


def _pad_rows(m: torch.Tensor, rows: int) -> torch.Tensor:
    return m if m.shape[0] == rows else torch.cat((m, m.new_zeros((rows - m.shape[0],) + tuple(m.shape[1:]))), 0)


def gemm_tn(a2: torch.Tensor, b2: torch.Tensor, want_colsum: bool = False):
    """K10 (psnode_gemm_tn_f32): a2^T @ b2 for tall-skinny fp32 [R, p], [R, q] row tensors on the hand-written MFMA contraction kernel
    (rows as the contraction index, deterministic partial sums); with want_colsum also sum_r a2[r, :].  None if the shapes / alignment
    are outside the kernel's class (p, q <= 128 and multiples of 4, 16-byte aligned rows) -- the caller then decides."""
    if a2.dim() != 2 or b2.dim() != 2 or a2.shape[0] != b2.shape[0] or a2.device.type != "cuda" or a2.dtype != torch.float32 or b2.dtype != torch.float32:
        return None
    if a2.stride(1) != 1:
        a2 = a2.contiguous()
    if b2.stride(1) != 1:
        b2 = b2.contiguous()
    if a2.shape[0] == 0:      # nothing to contract (an empty tensor has no device pointer to hand over)
        if a2.shape[1] > 128 or b2.shape[1] > 128 or (a2.shape[1] & 3) or (b2.shape[1] & 3):
            return None
        c = torch.zeros((a2.shape[1], b2.shape[1]), dtype=torch.float32, device=a2.device)
        return (c, torch.zeros(a2.shape[1], dtype=torch.float32, device=a2.device)) if want_colsum else c
    lib = _lib.load()
    a = _lib.GemmTnArgsF32()
    a.rows, a.M, a.N = a2.shape[0], a2.shape[1], b2.shape[1]
    a.A, a.lda, a.B, a.ldb = a2.data_ptr(), a2.stride(0) if a2.shape[0] > 1 else a2.shape[1], b2.data_ptr(), b2.stride(0) if b2.shape[0] > 1 else b2.shape[1]
    if not lib.psnode_gemm_tn_supported(ctypes.byref(a)):
        return None
    dev = a2.device
    with torch.cuda.device(dev):
        c = _empty((a.M, a.N), dtype=torch.float32, device=dev)
        cs = _empty((a.M,), dtype=torch.float32, device=dev) if want_colsum else None
        a.C = c.data_ptr()
        a.colsum_a = cs.data_ptr() if cs is not None else None
        ws = _empty(lib.psnode_gemm_tn_workspace_bytes(ctypes.byref(a)) + 256, dtype=torch.uint8, device=dev)
        wp, wn = _aligned_ptr(ws)
        _lib.check(lib.psnode_gemm_tn_f32(ctypes.byref(a), wp, wn, torch.cuda.current_stream(dev).cuda_stream), "psnode_gemm_tn_f32")
    return (c, cs) if want_colsum else c


def _gemm_tn(a2: torch.Tensor, b2: torch.Tensor, groups: int) -> torch.Tensor:
    """a2^T @ b2 for tall-skinny [N, p], [N, q] (N in the millions).  Round 6: on K10 (gemm_tn above) whenever the shapes are in its class
    -- every call of the latent-wide backward at hidden % 4 == 0 --; otherwise (odd widths) `groups` independent library partial products
    + one sum."""
    c = gemm_tn(a2, b2)
    if c is not None:
        return c
    N = a2.shape[0]
    groups = max(1, min(groups, 256))     # the [groups, p, q] partial products are materialised: cap them (small B x long T chunks)
    while groups > 1 and N % groups:
        groups -= 1
    return torch.bmm(a2.view(groups, N // groups, -1).transpose(1, 2), b2.view(groups, N // groups, -1)).sum(0)


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
