"""Encoder / decoder row MLPs of the direct_encode models (`nn.Sequential(Linear, ELU, Linear)` over the last dim) on the HIP row
kernels K3b, forward and backward, with an autograd bridge."""
import ctypes
from typing import Optional

import torch
import torch.nn as nn

from .. import _lib
from ._common import Layers, _aligned_ptr, _empty, _f32_dev, _mlp, _split_grads, sequential_layers
from .plan import _needs_autograd

def _row_addressing(x: torch.Tensor):
    """(tensor to keep alive, rows, row stride, inner rows, outer stride) of `x`'s last-dim rows for the row kernels.  A 3-D tensor with
    a dense last dim goes in AS IT IS LAID OUT -- row r = i0 * n1 + i1 at i0 * stride(0) + i1 * stride(1) -- so the time-major view
    `x.permute(1, 0, 2)` of a [B,T,D] batch (neural_00_ODE_02_direct_encode.py:76) is read in place (two clones of the dataset per
    training step before round 5); anything else is flattened (a copy only if the flattening needs one)."""
    d = x.shape[-1]
    if x.dim() == 3 and (d == 1 or x.stride(2) == 1) and x.stride(0) >= 0 and x.stride(1) >= d and x.shape[0] * x.shape[1] < 2 ** 32:
        n0, n1 = x.shape[0], x.shape[1]
        if x.stride(0) == n1 * x.stride(1) or n0 == 1:
            return x, n0 * n1, x.stride(1), 0, 0
        return x, n0 * n1, x.stride(1), n1, x.stride(0)
    x2 = x.reshape(-1, d)
    if x2.stride(-1) != 1 and d > 1:
        x2 = x2.contiguous()
    return x2, x2.shape[0], max(x2.stride(0), d), 0, 0


# ----------------------------------------------------------------------------- K11 / K10: the row MLPs at every other width <= 128
def linear_rows(x2: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor] = None, epi: int = 0, hh: Optional[torch.Tensor] = None,
                transposed: bool = False, out_shape=None) -> torch.Tensor:
    """K11 (psnode_linear_rows_f32): Y = epi(x2 @ Wm^T + bias) over the rows of a contiguous-row fp32 [R, K] tensor.  Wm = W ([N, K], an
    nn.Linear weight) or, with transposed=True, W^T (W is [K, N]: `x2 @ W`).  epi 0 identity / 1 ELU / 2 multiply by ELU'(hh), hh = ELU
    outputs [R, N]."""
    lib = _lib.load()
    dev = x2.device
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    R, K = x2.shape
    Wc = W.detach()
    N = Wc.shape[1] if transposed else Wc.shape[0]
    a = _lib.LinearRowsArgsF32()
    a.rows, a.K, a.N = R, K, N
    a.X, a.ldx = x2.data_ptr(), x2.stride(0) if R > 1 else K
    a.W = Wc.data_ptr()
    a.w_stride_n, a.w_stride_k = (Wc.stride(1), Wc.stride(0)) if transposed else (Wc.stride(0), Wc.stride(1))
    bc = None
    if bias is not None:
        bc = bias.detach().contiguous()
        a.bias = bc.data_ptr()
    a.epi = epi
    if epi == 2:
        if hh is None or hh.shape != (R, N) or hh.stride(-1) != 1:
            raise ValueError("linear_rows(epi=2): hh must be a contiguous-row [R, N] tensor of ELU outputs")
        a.Hh, a.ldh = hh.data_ptr(), hh.stride(0) if R > 1 else N
    with torch.cuda.device(dev):
        # (out_shape: the caller's [..., N] shape with prod(...) = R -- allocated in that shape, so that an autograd Function can hand it out
        #  without a view created inside its forward, which autograd refuses to see modified in place: `x_pred[0] = x0` of the DAE script)
        y = _empty(tuple(out_shape) if out_shape is not None else (R, N), dtype=torch.float32, device=dev)
        a.Y, a.ldy = y.data_ptr(), N
        if not lib.psnode_linear_rows_supported(ctypes.byref(a)):
            raise ValueError(f"linear_rows: K = {K}, N = {N} outside the kernel's class (<= 128)")
        if R:
            _lib.check(lib.psnode_linear_rows_f32(ctypes.byref(a), torch.cuda.current_stream(dev).cuda_stream), "psnode_linear_rows_f32")
    return y


def _pad4(t2: torch.Tensor) -> torch.Tensor:
    """[R, n] -> contiguous [R, 4 ceil(n / 4)] (zero columns behind): K10 contracts float4 row segments."""
    n = t2.shape[1]
    if n % 4 == 0:
        return t2 if t2.stride(1) == 1 and t2.stride(0) % 4 == 0 and t2.data_ptr() % 16 == 0 else t2.contiguous()
    out = torch.zeros((t2.shape[0], (n + 3) // 4 * 4), dtype=t2.dtype, device=t2.device)
    out[:, :n] = t2
    return out


def wide_rows_class(layers) -> bool:
    """Linear(in, H) ELU Linear(H, out) with every width <= 128: what K11 / K10 carry (any width; K3b keeps hidden 16 / 64)."""
    if layers is None or len(layers) != 2:
        return False
    H, din, dout = layers[0][0].shape[0], layers[0][0].shape[1], layers[1][0].shape[0]
    return layers[1][0].shape[1] == H and 1 <= H <= 128 and 1 <= din <= 128 and 1 <= dout <= 128


def _k3b_class(layers) -> bool:
    if layers is None or len(layers) != 2:
        return False
    H, din, dout = layers[0][0].shape[0], layers[0][0].shape[1], layers[1][0].shape[0]
    return H in (16, 64) and (din <= 16 or (din == 64 and H == 64)) and (dout <= 16 or dout == H)


def wide_mlp_rows(layers: Layers, inp: torch.Tensor, want_hidden: bool = False):
    """The two-layer row MLP on K11: h = ELU(x W1^T + b1) (one launch, ELU fused), y = h W2^T + b2 (one launch)."""
    (W1, b1), (W2, b2) = layers
    x2 = inp.reshape(-1, inp.shape[-1])
    h = linear_rows(x2, W1, b1, epi=1)
    y = linear_rows(h, W2, b2, out_shape=(*inp.shape[:-1], W2.shape[0]))
    return (y, x2, h) if want_hidden else y


class _WideRowsMlp(torch.autograd.Function):
    """Linear-ELU-Linear over rows at the widths K3b does not carry, hand-written both ways (round 6): forward 2 x K11; backward
    delta = (g W2) * ELU'(h) (K11, epilogue 2), grad_in = delta W1 (K11), dW2 | db2 = g^T h | colsum g and dW1 | db1 = delta^T x | colsum
    delta (K10).  Saves the input rows and the hidden rows."""

    @staticmethod
    def forward(ctx, inp, w1, b1, w2, b2):
        y, x2, h = wide_mlp_rows([(w1.detach(), b1.detach()), (w2.detach(), b2.detach())], inp.detach(), want_hidden=True)
        ctx.save_for_backward(x2, h, w1, w2)
        ctx.in_shape = inp.shape
        return y

    @staticmethod
    def backward(ctx, gy):
        from ._common import gemm_tn
        x2, h, w1, w2 = ctx.saved_tensors
        w1, w2 = w1.detach(), w2.detach()
        dout, H, din = w2.shape[0], w2.shape[1], w1.shape[1]
        g2 = gy.reshape(-1, dout)
        if g2.stride(-1) != 1:
            g2 = g2.contiguous()
        delta = linear_rows(g2, w2, None, epi=2, hh=h, transposed=True)                 # [R, H] = (g2 @ W2) * ELU'(h)
        gin = linear_rows(delta, w1, None, transposed=True, out_shape=ctx.in_shape) if ctx.needs_input_grad[0] else None
        hp = _pad4(h)
        if hp.shape[1] <= 128 and ((dout + 3) // 4 * 4) <= 128:
            c2, s2 = gemm_tn(_pad4(g2), hp, want_colsum=True)
            dW2, db2 = c2[:dout, :H], s2[:dout]
            c1, s1 = gemm_tn(_pad4(delta), _pad4(x2), want_colsum=True)
            dW1, db1 = c1[:H, :din], s1[:H]
        else:
            dW2, db2, dW1, db1 = g2.t() @ h, g2.sum(0), delta.t() @ x2, delta.sum(0)
        return gin, dW1.contiguous(), db1.contiguous(), dW2.contiguous(), db2.contiguous()


def mlp_rows(layers: Layers, inp: torch.Tensor) -> torch.Tensor:
    """Fused `nn.Sequential(Linear, ELU, Linear)` over the last dim of `inp` (any leading shape) on the HIP row kernel:
    the encoders / decoders of the direct_encode models (neural_00_ODE_02_direct_encode.py:64-69)."""
    lib = _lib.load()
    dev = inp.device
    if not _k3b_class(layers) and wide_rows_class(layers):
        return wide_mlp_rows([(w.detach(), b.detach()) for w, b in layers], _f32_dev(inp, dev, "input"))
    keep: list = []
    m = _mlp(layers, dev, "mlp", keep)
    if not lib.psnode_mlp_rows_supported(ctypes.byref(m)):
        raise ValueError("mlp_rows: needs Linear(in, H) ELU Linear(H, out) with H in {16, 64}, in <= 16 (or in = H = 64), out <= 16 (or out = H)")
    x = _f32_dev(inp, dev, "input")
    if x.shape[-1] != m.in_dim:
        raise ValueError(f"mlp_rows: input width {x.shape[-1]}, expected {m.in_dim}")
    x2, rows, rstride, inner, outer = _row_addressing(x)
    out = _empty((*x.shape[:-1], layers[-1][0].shape[0]), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.psnode_mlp_rows_f32(ctypes.byref(m), rows, x2.data_ptr(), rstride, inner, outer, out.data_ptr(), out.shape[-1],
                                     torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "psnode_mlp_rows_f32")
    return out


def mlp_rows_backward(layers: Layers, inp: torch.Tensor, grad_out: torch.Tensor, need_grad_in: bool = True):
    """Backward of `mlp_rows`: returns (grad_in or None, [dW1, db1, dW2, db2]) from the saved input and grad_out (row kernel)."""
    lib = _lib.load()
    dev = inp.device
    keep: list = []
    m = _mlp(layers, dev, "mlp", keep)
    if not lib.psnode_mlp_rows_supported(ctypes.byref(m)):
        raise ValueError("mlp_rows_backward: unsupported MLP shape")
    x2, rows, rstride, inner, outer = _row_addressing(_f32_dev(inp, dev, "input"))
    g2 = _f32_dev(grad_out, dev, "grad_out").reshape(-1, grad_out.shape[-1])
    if g2.stride(-1) != 1:
        g2 = g2.contiguous()
    if g2.shape[0] != rows or g2.shape[1] != layers[-1][0].shape[0]:
        raise ValueError(f"mlp_rows_backward: grad_out {tuple(grad_out.shape)} does not match input {tuple(inp.shape)}")
    with torch.cuda.device(dev):
        gin = _empty((*inp.shape[:-1], inp.shape[-1]), dtype=torch.float32, device=dev) if need_grad_in else None
        npar = sum(w.numel() + b.numel() for w, b in layers)
        gp = _empty(npar, dtype=torch.float32, device=dev)
        nbytes = lib.psnode_mlp_rows_backward_workspace_bytes(ctypes.byref(m), rows)
        ws = _empty(nbytes + 256, dtype=torch.uint8, device=dev)
        wp, wn = _aligned_ptr(ws)
        rc = lib.psnode_mlp_rows_backward_f32(ctypes.byref(m), rows, x2.data_ptr(), rstride, inner, outer, g2.data_ptr(), g2.stride(0),
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


def mlp_rows_backward_multi(layers: Layers, inps, grad_outs, need_grad_in):
    """Backward of ONE module applied to several row sets (`inps[k]`, `grad_outs[k]`; a None grad_out = that output was unused): every
    set's kernel leaves its per-wave partials side by side in one buffer, one fixed-order reduction forms the module's gradient
    (psnode_mlp_rows_reduce_f32) -- no reduction per set, no autograd `add` per parameter tensor and extra use.
    Returns ([grad_in or None per set], [dW1, db1, dW2, db2])."""
    lib = _lib.load()
    live = [k for k, g in enumerate(grad_outs) if g is not None]
    dev = inps[0].device
    keep: list = []
    m = _mlp(layers, dev, "mlp", keep)
    if not lib.psnode_mlp_rows_supported(ctypes.byref(m)):
        raise ValueError("mlp_rows_backward_multi: unsupported MLP shape")
    npar = sum(w.numel() + b.numel() for w, b in layers)
    gins = [None] * len(inps)
    with torch.cuda.device(dev):
        gp = _empty(npar, dtype=torch.float32, device=dev)
        if not live:
            return gins, _split_grads(gp.zero_(), layers)
        sets = []
        for k in live:
            x2, rows, rstride, inner, outer = _row_addressing(_f32_dev(inps[k], dev, "input"))
            g2 = _f32_dev(grad_outs[k], dev, "grad_out").reshape(-1, grad_outs[k].shape[-1])
            if g2.stride(-1) != 1:
                g2 = g2.contiguous()
            if g2.shape[0] != rows or g2.shape[1] != layers[-1][0].shape[0]:
                raise ValueError(f"mlp_rows_backward_multi: grad_out {tuple(grad_outs[k].shape)} does not match input {tuple(inps[k].shape)}")
            sets.append((k, x2, rows, rstride, inner, outer, g2, int(lib.psnode_mlp_rows_backward_parts(ctypes.byref(m), rows))))
        n_parts = sum(s_[-1] for s_ in sets)
        nbytes = lib.psnode_mlp_rows_reduce_workspace_bytes(ctypes.byref(m), n_parts)
        ws = _empty(nbytes + 256, dtype=torch.uint8, device=dev)
        wp, wn = _aligned_ptr(ws)
        st = torch.cuda.current_stream(dev).cuda_stream
        off = 0
        for k, x2, rows, rstride, inner, outer, g2, parts in sets:
            gin = _empty(tuple(inps[k].shape), dtype=torch.float32, device=dev) if need_grad_in[k] else None
            gins[k] = gin
            _lib.check(lib.psnode_mlp_rows_backward_f32(ctypes.byref(m), rows, x2.data_ptr(), rstride, inner, outer, g2.data_ptr(), g2.stride(0),
                                                        gin.data_ptr() if gin is not None else None, inps[k].shape[-1], None, wp + off,
                                                        parts * npar * 4, st), "psnode_mlp_rows_backward_f32")
            off += parts * npar * 4
        _lib.check(lib.psnode_mlp_rows_reduce_f32(ctypes.byref(m), n_parts, wp, wn, gp.data_ptr(), st), "psnode_mlp_rows_reduce_f32")
    return gins, _split_grads(gp, layers)


class _RowsMlpMulti(torch.autograd.Function):
    """One module over several row sets in one autograd node (see mlp_rows_backward_multi)."""

    @staticmethod
    def forward(ctx, w1, b1, w2, b2, *inps):
        ctx.save_for_backward(w1, b1, w2, b2, *inps)
        ctx.set_materialize_grads(False)          # an unused output arrives as None (its set is skipped), not as a tensor of zeros
        return tuple(mlp_rows([(w1, b1), (w2, b2)], a) for a in inps)

    @staticmethod
    def backward(ctx, *grad_outs):
        w1, b1, w2, b2, *inps = ctx.saved_tensors
        gins, gp = mlp_rows_backward_multi([(w1.detach(), b1.detach()), (w2.detach(), b2.detach())], [a.detach() for a in inps], grad_outs,
                                           ctx.needs_input_grad[4:])
        return (*gp, *gins)


def mlp_rows_autograd_multi(seq, *inps):
    """`tuple(seq(a) for a in inps)` for a recognised Linear-ELU-Linear on the row kernels, ONE autograd node for all of them (K3b's
    widths; at the other widths one K11 / K10 node per set)."""
    lin = [m for m in seq if isinstance(m, nn.Linear)]
    if not _k3b_class([(lin[0].weight, lin[0].bias), (lin[1].weight, lin[1].bias)]):
        return tuple(_WideRowsMlp.apply(a, lin[0].weight, lin[0].bias, lin[1].weight, lin[1].bias) for a in inps)
    return _RowsMlpMulti.apply(lin[0].weight, lin[0].bias, lin[1].weight, lin[1].bias, *inps)


# ----------------------------------------------------------------------------- the reconstruction branch in one kernel each way (K3r)
def recon_rows_supported(enc_layers: Layers, dec_layers: Layers, inp: torch.Tensor) -> bool:
    """x_decoder(x_encoder(inp)) on the fused reconstruction kernels: fp32 HIP tensor, encoder in <= 16 -> 16 -> 16, decoder 16 -> 16 -> out <= 16."""
    if inp.device.type != "cuda" or inp.dtype != torch.float32 or inp.numel() == 0 or len(enc_layers) != 2 or len(dec_layers) != 2:
        return False
    keep: list = []
    return bool(_lib.load().psnode_recon_rows_supported(ctypes.byref(_mlp(enc_layers, inp.device, "encoder", keep)),
                                                        ctypes.byref(_mlp(dec_layers, inp.device, "decoder", keep))))


def recon_rows(enc_layers: Layers, dec_layers: Layers, inp: torch.Tensor) -> torch.Tensor:
    """x_re = decoder(encoder(inp)) over the last dim (neural_00_ODE_02_direct_encode.py:87) in ONE launch: the encoded rows never reach memory."""
    lib = _lib.load()
    dev = inp.device
    keep: list = []
    me, md = _mlp(enc_layers, dev, "encoder", keep), _mlp(dec_layers, dev, "decoder", keep)
    x = _f32_dev(inp, dev, "input")
    if x.shape[-1] != me.in_dim:
        raise ValueError(f"recon_rows: input width {x.shape[-1]}, expected {me.in_dim}")
    x2, rows, rstride, inner, outer = _row_addressing(x)
    out = _empty((*x.shape[:-1], dec_layers[-1][0].shape[0]), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.psnode_recon_rows_f32(ctypes.byref(me), ctypes.byref(md), rows, x2.data_ptr(), rstride, inner, outer, out.data_ptr(),
                                             out.shape[-1], torch.cuda.current_stream(dev).cuda_stream), "psnode_recon_rows_f32")
    return out


def recon_rows_backward(enc_layers: Layers, dec_layers: Layers, inp: torch.Tensor, grad_out: torch.Tensor):
    """Parameter gradients of both modules from the raw rows and dL/dx_re: ([dW1e, db1e, dW2e, db2e], [dW1d, db1d, dW2d, db2d])."""
    lib = _lib.load()
    dev = inp.device
    keep: list = []
    me, md = _mlp(enc_layers, dev, "encoder", keep), _mlp(dec_layers, dev, "decoder", keep)
    x2, rows, rstride, inner, outer = _row_addressing(_f32_dev(inp, dev, "input"))
    g2 = _f32_dev(grad_out, dev, "grad_out").reshape(-1, grad_out.shape[-1])
    if g2.stride(-1) != 1:
        g2 = g2.contiguous()
    if g2.shape[0] != rows or g2.shape[1] != dec_layers[-1][0].shape[0]:
        raise ValueError(f"recon_rows_backward: grad_out {tuple(grad_out.shape)} does not match input {tuple(inp.shape)}")
    with torch.cuda.device(dev):
        npar = int(lib.psnode_recon_rows_param_count(ctypes.byref(me), ctypes.byref(md)))
        gp = _empty(npar, dtype=torch.float32, device=dev)
        nbytes = lib.psnode_recon_rows_backward_workspace_bytes(ctypes.byref(me), ctypes.byref(md), rows)
        ws = _empty(nbytes + 256, dtype=torch.uint8, device=dev)
        wp, wn = _aligned_ptr(ws)
        _lib.check(lib.psnode_recon_rows_backward_f32(ctypes.byref(me), ctypes.byref(md), rows, x2.data_ptr(), rstride, inner, outer, g2.data_ptr(),
                                                      g2.stride(0), gp.data_ptr(), wp, wn, torch.cuda.current_stream(dev).cuda_stream),
                   "psnode_recon_rows_backward_f32")
    ne = sum(w.numel() + b.numel() for w, b in enc_layers)
    return _split_grads(gp[:ne], enc_layers), _split_grads(gp[ne:], dec_layers)


class _ReconRows(torch.autograd.Function):
    """decoder(encoder(inp)) for DATA rows (inp gets no gradient): saves the input rows only; the backward recomputes everything in one kernel."""

    @staticmethod
    def forward(ctx, inp, w1e, b1e, w2e, b2e, w1d, b1d, w2d, b2d):
        ctx.save_for_backward(inp, w1e, b1e, w2e, b2e, w1d, b1d, w2d, b2d)
        return recon_rows([(w1e, b1e), (w2e, b2e)], [(w1d, b1d), (w2d, b2d)], inp)

    @staticmethod
    def backward(ctx, grad_out):
        inp, w1e, b1e, w2e, b2e, w1d, b1d, w2d, b2d = (q.detach() for q in ctx.saved_tensors)
        ge, gd = recon_rows_backward([(w1e, b1e), (w2e, b2e)], [(w1d, b1d), (w2d, b2d)], inp, grad_out)
        return (None, *ge, *gd)


def recon_rows_autograd(encoder, decoder, inp: torch.Tensor) -> torch.Tensor:
    """`decoder(encoder(inp))` for two recognised Linear-ELU-Linear modules, differentiable w.r.t. their parameters (inp must not need a gradient)."""
    le = [m for m in encoder if isinstance(m, nn.Linear)]
    ld = [m for m in decoder if isinstance(m, nn.Linear)]
    return _ReconRows.apply(inp, le[0].weight, le[0].bias, le[1].weight, le[1].bias, ld[0].weight, ld[0].bias, ld[1].weight, ld[1].bias)


def mlp_rows_autograd(seq, inp: torch.Tensor) -> torch.Tensor:
    """`seq(inp)` for a recognised Linear-ELU-Linear on the row kernels, differentiable w.r.t. the input and the parameters."""
    lin = [m for m in seq if isinstance(m, nn.Linear)]
    if not _k3b_class([(lin[0].weight, lin[0].bias), (lin[1].weight, lin[1].bias)]):
        return _WideRowsMlp.apply(inp, lin[0].weight, lin[0].bias, lin[1].weight, lin[1].bias)
    return _RowsMlp.apply(inp, lin[0].weight, lin[0].bias, lin[1].weight, lin[1].bias)


def rows_layers_of(seq, inp: torch.Tensor, allow_grad: bool = False):
    """Layers if `seq(inp)` can run on the row kernel (2-layer ELU-MLP, hidden 16 / 64, fp32 HIP tensor); with autograd in play
    only when `allow_grad` (the caller then goes through mlp_rows_autograd)."""
    if inp.device.type != "cuda" or inp.dtype != torch.float32 or inp.numel() == 0:
        return None
    layers = sequential_layers(seq)
    if layers is None or len(layers) != 2:
        return None
    if not _k3b_class(layers) and not wide_rows_class(layers):      # K3b at hidden 16 / 64, K11 / K10 at every other width <= 128
        return None
    if not allow_grad and _needs_autograd([inp] + [p for wb in layers for p in wb]):
        return None
    return layers
