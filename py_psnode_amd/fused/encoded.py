"""Whole direct_encode model forwards in ONE launch: ODE_02 (K3f, psnode_ode_encoded_integrate_f32) and DAE_02 at hidden 64 (K3g,
psnode_dae_encoded_integrate_f32)."""
import ctypes

import torch

from .. import _lib
from ._common import Layers, METHOD_ID, _aligned_ptr, _check_jump, _empty, _f32_dev, _jump, _mlp, _view, event_table

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
