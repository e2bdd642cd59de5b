"""Forward entry points on raw tensors: `ode_integrate` / `dae_integrate` (psnode_ode_integrate_f32 / psnode_dae_integrate_f32 -> K1 / K1x /
K2 / K3* / K0) and the queries that tell the autograd bridge which calls can save their activations."""
import ctypes
from typing import Optional

import torch

from .. import _lib
from ._common import (KERNEL_ID, Layers, METHOD_ID, _aligned16, _aligned_ptr, _check_jump, _check_tb, _empty, _f32_dev, _jump, _mlp, _view, _workspace, event_table)

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
            "kernel K0 (any layer count and widths that fit the 160 KB LDS; MFMA layers since round 6), at 2.5-4x the time per "
            "state-step of a specialised integrator (9.5 vs 3.5 ms per 4096 x 1000 RK4 batch at hidden 64; DESIGN.md 'K0')")


def _note_k0(lib, args, dae: bool, de_layers: Layers):
    """AUTO landing on K0 with a no_encode-style MLP wider than the MFMA classes: never silently slower (2.5-4x per flop)."""
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
        warnings.warn(f"hidden width {widths[0]} runs on the generic kernel K0 (weights streamed from L2), at 2-3x the time per flop of the "
                      f"MFMA integrators.  {_MFMA_CLASSES}", RuntimeWarning, stacklevel=3)


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
    if save:      # the latent-wide saving forward (K3w) reads rows as float4: a misaligned view is copied once here instead of failing in
        x, z, z_jump = _aligned16(x), _aligned16(z), _aligned16(z_jump)      # the middle of a training step (the backward does the same)
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
    if save:      # (as ode_integrate)
        x_init, z, v, z_jump, v_jump = _aligned16(x_init), _aligned16(z), _aligned16(v), _aligned16(z_jump), _aligned16(v_jump)
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
