"""Fused masked-MSE loss (kernel K6, psnode_masked_mse_f32): the step right after the integrator in the scripts'
training loops, as one pass over (prediction, target, mask) that returns the loss AND d loss / d prediction.

Reference expressions (Loss_func = nn.functional.mse_loss, neural_00_ODE_01_no_encode.py:49) -- the four scripts differ:
  ODE_01  x_loss = sum(sum(Loss_func(x_pred, x, reduction='none') * mask, dim=1), dim=0) / sum(mask);  loss = sum(x_loss)
          (neural_00_ODE_01_no_encode.py:353-355; the x0_loss of :353 is computed but NOT added)              -> ode01_loss
  ODE_02  loss = Loss_func(x[:,0,:], x_pred[:,0,:]) + sum(x_loss) + Loss_func(x_re, x)
          (neural_00_ODE_02_direct_encode.py:267-270)                                                         -> ode02_loss
  DAE_01  x_loss = (sum(se*mask) + 9*sum(se[:,:,1:2]*mask)) / sum(mask);  i_loss = sum(se_i*mask)/sum(mask)
          loss = x_loss + i_loss + Loss_func(x[:,0,:], x_pred[:,0,:]) + Loss_func(i[:,0,:], i_pred[:,0,:])
          (neural_01_DAE_01_no_encode.py:414-419)                                                             -> dae01_loss
  DAE_02  as DAE_01 WITHOUT the 9x extra weight on x column 1 (commented out upstream), plus the reconstruction terms
          Loss_func(x_re, x) + Loss_func(i_re, i)   (neural_01_DAE_02_direct_encode.py:359-365)                -> dae02_loss
Tensors are in the scripts' [B,T,D] shape with any strides: the integrator's `xs.permute(1,0,2)` and the DataLoader's
B-major batches go in as they are.  There is no non-HIP implementation here."""
import ctypes
from typing import Optional, Sequence

import torch

from . import _lib
from .fused import _empty


def _view_bt(t: torch.Tensor, dev, name: str, keep: list) -> _lib.ViewF32:
    """[B,T,D] tensor -> time-major view struct (stride_t, stride_b); copies only if the last dim is strided."""
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: fp32 only, got {t.dtype}")
    if t.device != dev:
        raise ValueError(f"{name}: on {t.device}, expected {dev}")
    t = t.detach()
    if t.shape[-1] > 1 and t.stride(2) != 1:
        t = t.contiguous()
    keep.append(t)
    return _lib.ViewF32(t.data_ptr(), t.stride(1), t.stride(0))


_COL_WEIGHTS = {}


def masked_mse_terms(pred, target, mask=None, col_weight=None, inv_norm: Optional[torch.Tensor] = None, scale: float = 1.0,
                     t0_coef: float = 0.0, want_grad: bool = False):
    """Raw kernel call.  pred/target [B,T,D]; mask None | [B,T,1] | [B,T,D]; col_weight None | [D] tensor;
    inv_norm None | 0-dim/1-element DEVICE tensor.  Returns (out[D+2], grad_tm) with out = per-column masked terms,
    the t=0 term, the total; grad_tm = d total / d pred as a contiguous TIME-MAJOR [T,B,D] tensor (or None)."""
    lib = _lib.load()
    if pred.dim() != 3 or pred.shape != target.shape:
        raise ValueError(f"pred {tuple(pred.shape)} and target {tuple(target.shape)} must be equal [B,T,D] shapes")
    dev = pred.device
    if dev.type != "cuda":
        raise ValueError("masked_mse: tensors must be on a HIP device (no CPU path)")
    B, T, D = pred.shape
    keep = []
    a = _lib.LossArgsF32()
    a.T, a.B, a.D = T, B, D
    a.pred, a.target = _view_bt(pred, dev, "pred", keep), _view_bt(target, dev, "target", keep)
    if mask is not None:
        if mask.dim() != 3 or mask.shape[:2] != pred.shape[:2] or mask.shape[2] not in (1, D):
            raise ValueError(f"mask {tuple(mask.shape)} must be [B,T,1] or [B,T,D]")
        a.mask_width = mask.shape[2]
        a.mask = _view_bt(mask, dev, "mask", keep)
    if col_weight is not None:
        if isinstance(col_weight, torch.Tensor):
            cw = col_weight.to(device=dev, dtype=torch.float32).contiguous()
        else:       # a python sequence (the DAE scripts' [1, 10, 1, ...]): uploaded once per (values, device) -- a pageable H2D copy
            key = (tuple(float(c) for c in col_weight), str(dev))   # per call is a device synchronisation in the middle of the step
            cw = _COL_WEIGHTS.get(key)
            if cw is None:
                cw = _COL_WEIGHTS[key] = torch.tensor(key[0], dtype=torch.float32, device=dev)
        if cw.numel() != D:
            raise ValueError(f"col_weight has {cw.numel()} entries, expected {D}")
        keep.append(cw)
        a.col_weight = cw.data_ptr()
    if inv_norm is not None:
        if inv_norm.numel() != 1 or inv_norm.dtype != torch.float32 or inv_norm.device != dev:
            raise ValueError("inv_norm must be a one-element fp32 tensor on the same device")
        keep.append(inv_norm)
        a.inv_norm = inv_norm.data_ptr()
    a.scale, a.t0_coef = float(scale), float(t0_coef)
    out = _empty(D + 2, dtype=torch.float32, device=dev)
    grad = _empty((T, B, D), dtype=torch.float32, device=dev) if want_grad else None
    a.out = out.data_ptr()
    a.grad_pred = grad.data_ptr() if want_grad else None
    nbytes = lib.psnode_masked_mse_workspace_bytes(ctypes.byref(a))
    ws = _empty(nbytes + 256, dtype=torch.uint8, device=dev)
    p = (ws.data_ptr() + 255) // 256 * 256
    with torch.cuda.device(dev):
        st = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(lib.psnode_masked_mse_f32(ctypes.byref(a), ctypes.c_void_p(p), ctypes.c_size_t(nbytes), ctypes.c_void_p(st)),
                   "psnode_masked_mse_f32")
    return out, grad


class _MaskedMSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, mask, col_weight, inv_norm, scale, t0_coef):
        out, grad = masked_mse_terms(pred, target, mask, col_weight, inv_norm, scale, t0_coef, want_grad=pred.requires_grad)
        ctx.grad_tm = grad
        ctx.mark_non_differentiable(out)
        return out[pred.shape[2] + 1].clone(), out

    @staticmethod
    def backward(ctx, go, _go_terms):
        g = ctx.grad_tm.permute(1, 0, 2) * go      # [B,T,D] view of the time-major buffer: what PermuteBackward wants
        return g, None, None, None, None, None, None


def masked_mse(pred, target, mask=None, col_weight: Optional[Sequence[float]] = None, inv_norm: Optional[torch.Tensor] = None,
               scale: float = 1.0, t0_coef: float = 0.0):
    """Differentiable (w.r.t. pred) fused loss.  Returns (total, terms) where terms = [per-column..., t0 term, total]."""
    return _MaskedMSE.apply(pred, target, mask, col_weight, inv_norm, scale, t0_coef)


def inv_mask_sum(mask: torch.Tensor) -> torch.Tensor:
    """1 / sum(mask) as a device scalar (no host sync)."""
    return torch.sum(mask).reciprocal().reshape(1)


def ode01_loss(x_pred, x, mask):
    """ODE_01 training loss: neural_00_ODE_01_no_encode.py:353-355 (masked term only)."""
    return masked_mse(x_pred, x, mask, inv_norm=inv_mask_sum(mask))


ode_loss = ode01_loss     # round-1 name


def recon_loss(x_re, x):
    """Unmasked mean squared reconstruction error `Loss_func(x_re, x)`: neural_00_ODE_02_direct_encode.py:269."""
    return masked_mse(x_re, x, None, scale=1.0 / x.numel())


def ode02_loss(x_pred, x_re, x, mask):
    """ODE_02 training loss: x0 term + masked term + reconstruction (neural_00_ODE_02_direct_encode.py:267-270).
    Returns (loss, (terms of the prediction call, terms of the reconstruction call))."""
    B, _, xd = x.shape
    lp, tp = masked_mse(x_pred, x, mask, inv_norm=inv_mask_sum(mask), t0_coef=1.0 / (B * xd))
    lr, tr = recon_loss(x_re, x)
    return lp + lr, (tp, tr)


def dae01_loss(x_pred, x, i_pred, i, mask, x_col_weight: Optional[Sequence[float]] = None):
    """DAE_01 training loss: neural_01_DAE_01_no_encode.py:414-419 (column 1 of x counted 1 + 9 times unless `x_col_weight`
    says otherwise)."""
    B, _, xd = x.shape
    idim = i.shape[2]
    if x_col_weight is None:
        x_col_weight = [10.0 if d == 1 else 1.0 for d in range(xd)]
    inv = inv_mask_sum(mask)
    lx, tx = masked_mse(x_pred, x, mask, col_weight=x_col_weight, inv_norm=inv, t0_coef=1.0 / (B * xd))
    li, ti = masked_mse(i_pred, i, mask, inv_norm=inv, t0_coef=1.0 / (B * idim))
    return lx + li, (tx, ti)


dae_loss = dae01_loss     # round-1 name


def dae02_loss(x_pred, i_pred, x_re, i_re, x, i, mask):
    """DAE_02 training loss: unweighted columns + the two reconstruction terms (neural_01_DAE_02_direct_encode.py:359-365)."""
    base, terms = dae01_loss(x_pred, x, i_pred, i, mask, x_col_weight=[1.0] * x.shape[2])
    lxr, txr = recon_loss(x_re, x)
    lir, tir = recon_loss(i_re, i)
    return base + lxr + lir, (*terms, txr, tir)
