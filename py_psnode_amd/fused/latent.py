"""The latent integrators of the direct_encode models at the hidden widths without a dedicated latent kernel (every H <= 128 with
H % 4 == 0 other than 16 / 64, e.g. the scripts' argparse default --hidden 128): forward K3w saves its rows, the adjoint sweep K9w
(psnode_latent_backward_wide_f32) stores the adjoint rows, the parameter / input gradients are contractions over them here."""
import ctypes
from typing import Optional

import torch

from .. import _lib
from ._common import Layers, METHOD_ID, _aligned_ptr, _empty, _f32_dev, _gemm_tn, _mlp, _view, gemm_tn

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
            both = gemm_tn(R(gk), R(s_act), want_colsum=True)      # K10: the contraction and the bias gradient (column sums) in one pass
            gW2, gb2 = both if both is not None else (_gemm_tn(R(gk), R(s_act), G), R(gk).sum(0))
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
        from .rows import linear_rows
        rows_x_w = lambda rows2, Wkn: linear_rows(rows2, Wkn, None, transposed=True)      # [R, K] @ [K, N] on K11 (no library GEMM over T*B rows)
        if zd and need_grad_z:
            g["z"] = torch.zeros((T, B, H), **f32)
            g["z_jump"] = torch.zeros((B, n_ev, H), **f32) if evl is not None else None
            if T >= 2:
                route(rows_x_w(R(d1s), Fb(1)).view(T - 1, B, H), g["z"], g["z_jump"])
        if dae:
            g["v"] = torch.zeros((T, B, H), **f32)
            g["v_jump"] = torch.zeros((B, n_ev, H), **f32) if evl is not None else None
            if T >= 2:
                route(rows_x_w(R(d1s), Fb(nblk - 2)).view(T - 1, B, H), g["v"], g["v_jump"])
            # ---- the AE head: rows per grid point (un-jumped inputs) and per event taken (x of the jump step, jump rows)
            (A1, _ab1), (A2, _ab2) = [(w.detach(), b.detach()) for w, b in ae_layers]
            ah = s_ae[0]
            both = gemm_tn(R(gi), R(ah), want_colsum=True)
            gA2, gab2 = both if both is not None else (_gemm_tn(R(gi), R(ah), T), R(gi).sum(0))
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
                g["z"] += rows_x_w(R(da1), Ab(1)).view(T, B, H)
                if g["z_jump"] is not None:
                    g["z_jump"] += (R(da1_ev) @ Ab(1)).view(n_ev, B, H).permute(1, 0, 2)
            g["v"] += rows_x_w(R(da1), Ab(nblk - 2)).view(T, B, H)
            if g["v_jump"] is not None:
                g["v_jump"] += (R(da1_ev) @ Ab(nblk - 2)).view(n_ev, B, H).permute(1, 0, 2)
    g["x_init"] = gx0
    g["all_initial"] = ga0
    return g
