"""Backward of `dae_integrate`: K7f in one launch for the DAE_01 shape class at hidden <= 128 (psnode_dae_backward_wide_f32, + K7h for
the AE head's rows where the kernel does not form the head's gradients itself), K9 / K8 / K9w for the latent shapes, the generic K5
otherwise (psnode_dae_backward_f32)."""
import ctypes

import torch

from .. import _lib
from ._common import (KERNEL_ID, Layers, METHOD_ID, _aligned16, _aligned_ptr, _check_saved, _empty, _f32_dev, _jump, _mlp, _pad_rows, _padded_hidden, _split_grads, _view)
from .latent import latent_backward_wide, latent_wide_shape

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
    if bool(lib.psnode_dae_backward_supported(ctypes.byref(a))):      # K9 / K8 (latent shapes) or K5
        return True
    return dae_backward_wide_supported(method, de_layers, ae_layers, x_dim, z_dim, v_dim, i_dim)      # K7f: the DAE_01 class at hidden <= 128


def dae_backward_wide_supported(method: str, de_layers: Layers, ae_layers: Layers, x_dim, z_dim, v_dim, i_dim) -> bool:
    """Shapes of K7f (psnode_dae_backward_wide_f32): DE 3n -> h -> h -> h -> x and AE n+x+z+v -> h -> h -> h -> i with h <= 128,
    x <= 8, z+v+i <= 8."""
    if de_layers[0][0].device.type != "cuda" or len(de_layers) != 4 or len(ae_layers) != 4:
        return False
    lib = _lib.load()
    a = _lib.DaeBwdWideArgsF32()
    a.method, a.x_dim, a.z_dim, a.v_dim, a.i_dim, a.T, a.B = METHOD_ID[method], x_dim, z_dim, v_dim, i_dim, 2, 1
    dev = de_layers[0][0].device
    a.de, a.ae = _mlp(de_layers, dev, "de", []), _mlp(ae_layers, dev, "ae", [])
    return bool(lib.psnode_dae_backward_wide_supported(ctypes.byref(a)))


def dae_backward_wide(method: str, de_layers: Layers, ae_layers: Layers, t, z, v, all_initial, xs, is_, grad_xs, grad_is, event_idx=None,
                      z_jump=None, v_jump=None, saved=None, x_true=None, i_true=None):
    """Backward of `dae_integrate` at hidden <= 128 (the DAE_01 shape class): K7f (psnode_dae_backward_wide_f32) -- ONE launch over the
    whole grid that sweeps the adjoint through the DE stages, the AE head per grid point and the event-time recomputes and forms the DE's
    parameter gradients and the DE's share of the input gradients in the kernel.  With saved activations at hidden <= 64 the head's
    gradients are formed in the kernel too; otherwise its rows (one set per grid point) are contracted by K7h (`head_grads_hip`).
    saved = what `dae_integrate(save=True)` returned for the same call: the kernel evaluates nothing forwards.  x_true / i_true [T,B,.]:
    backward of a teacher-forced call (input_true_x / input_true_i, my_solvers.py:111-121) -- the dataset rows the forward call fed the
    DE / the heads; recompute form only.  A call whose head rows (6 x [T,B,H] + [T,B,40]) would not fit half of the free HBM is run over
    BATCH slices (trajectories are independent: parameter gradients add, per-trajectory gradients concatenate).
    Same return value as `dae_backward`."""
    lib = _lib.load()
    dev = xs.device
    T, B, xd = xs.shape
    zd, vd, idim = z.shape[-1], v.shape[-1], is_.shape[-1]
    nzv, ne = zd + vd, zd + vd + idim
    n = xd + ne
    Hr = de_layers[0][0].shape[0]                       # the MLPs' width; H = the width the kernel runs them at (zero-padded rows)
    H = _padded_hidden(Hr)
    S = {"euler": 1, "midpoint": 2, "rk4": 4}[method]
    if saved is None and B > 16:
        # the recompute form stores the AE head's rows of EVERY grid point (6 x [T,B,H] + [T,B,16] + the u rows of K7h): a very long grid on
        # a full card goes through in batch slices (a saved-activation call is not sliced: its forward already held ~S times as much)
        free, _ = torch.cuda.mem_get_info(dev)
        need = (6 * H + 40) * 4 * T * B
        if need > free // 2:
            nsl = min((B + 15) // 16, int(-(-need // max(free // 2, 1))))
            step = -(-((B + nsl - 1) // nsl) // 16) * 16
            return _dae_backward_wide_sliced(step, method, de_layers, ae_layers, t, z, v, all_initial, xs, is_, grad_xs, grad_is,
                                             event_idx, z_jump, v_jump, x_true, i_true)
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
    # Host-side accumulators of the AE head's K7h route: made on demand.  The one-launch route at hidden <= 64 (every gradient formed in
    # the kernel) needs none of them -- 18 fills, a [T,B,z+v] copy and 2 small GEMM operands per call that the step's profile showed as
    # glue (profiles/r04o_glue_dae01.txt).
    zv_all = None
    gA, gab, Sa1 = [None] * 4, [None] * 4, None

    def host_accumulators():
        nonlocal gA, gab, Sa1, zv_all
        gA = [torch.zeros_like(w) for w in (A1, A2, A3, A4)]
        gab = [torch.zeros(w.shape[0], **f32) for w in (A1, A2, A3, A4)]
        Sa1 = torch.zeros((B, H), **f32)                    # sum over the heads of the AE's delta_1
        zv_all = torch.cat((z.detach(), v.detach()), -1)    # [T, B, nzv] (one copy of the two input views)

    gzv = torch.zeros((T, B, nzv), **f32)                       # dL/d(z|v) of the un-jumped inputs
    gjump = torch.zeros((B, n_ev, nzv), **f32) if n_ev else None
    carry_x = torch.zeros((B, xd), **f32)
    a.carry_x = carry_x.data_ptr()
    xt_c = it_c = None
    if x_true is not None or i_true is not None:
        if saved is not None:
            raise ValueError("teacher forcing: the recompute form only (a teacher-forced forward saves nothing)")
        xt_c = _f32_dev(x_true, dev, "x_true").contiguous() if x_true is not None else None
        it_c = _f32_dev(i_true, dev, "i_true").contiguous() if i_true is not None else None
        if (xt_c is not None and tuple(xt_c.shape) != (T, B, xd)) or (it_c is not None and tuple(it_c.shape) != (T, B, idim)):
            raise ValueError("x_true / i_true must be [T,B,x_dim] / [T,B,i_dim]")
        keep += [xt_c, it_c]
        a.flags = (_lib.FLAG_INPUT_TRUE_X if xt_c is not None else 0) | (_lib.FLAG_INPUT_TRUE_I if it_c is not None else 0)
        a.x_true = xt_c.data_ptr() if xt_c is not None else None
        a.i_true = it_c.data_ptr() if it_c is not None else None
    gp_de = _empty(sum(w.numel() + w.shape[0] for w in (W1, W2, W3, W4)), **f32)
    ga0_de = _empty((B, n), **f32)
    a.grad_params_de, a.grad_all_initial_de = gp_de.data_ptr(), ga0_de.data_ptr()
    a.grad_zv = gzv.data_ptr()
    a.grad_jump = gjump.data_ptr() if gjump is not None else None
    jump_all = None
    if n_ev:
        parts = ([z_jump.detach()] if zd > 0 else []) + ([v_jump.detach()] if vd > 0 else [])
        jump_all = torch.cat(parts, -1)                         # [B, n_ev, nzv]

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
        host_accumulators()
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


def _dae_backward_wide_sliced(step, method, de_layers, ae_layers, t, z, v, all_initial, xs, is_, grad_xs, grad_is, event_idx, z_jump, v_jump,
                              x_true, i_true):
    """`dae_backward_wide` over batch slices of `step` trajectories: parameter gradients add, per-trajectory gradients concatenate."""
    B = xs.shape[1]
    tb = lambda q, b0, b1: None if q is None else q[:, b0:b1]          # [T, B, .] views
    bb = lambda q, b0, b1: None if q is None else q[b0:b1]             # [B, ...] tensors
    out = None
    for b0 in range(0, B, step):
        b1 = min(B, b0 + step)
        g = dae_backward_wide(method, de_layers, ae_layers, tb(t, b0, b1), tb(z, b0, b1), tb(v, b0, b1), bb(all_initial, b0, b1),
                              tb(xs, b0, b1), tb(is_, b0, b1), tb(grad_xs, b0, b1), tb(grad_is, b0, b1), event_idx=event_idx,
                              z_jump=bb(z_jump, b0, b1), v_jump=bb(v_jump, b0, b1), x_true=tb(x_true, b0, b1), i_true=tb(i_true, b0, b1))
        if out is None:
            out = {k: ([q.clone() for q in val] if k in ("de", "ae") else ([val] if val is not None else None)) for k, val in g.items()}
        else:
            for k, val in g.items():
                if k in ("de", "ae"):
                    for acc, q in zip(out[k], val):
                        acc.add_(q)
                elif val is not None:
                    out[k].append(val)
    cat_dim = {"x_init": 0, "all_initial": 0, "z": 1, "v": 1, "z_jump": 0, "v_jump": 0}
    return {k: (val if k in ("de", "ae") else (torch.cat(val, cat_dim[k]) if val is not None else None)) for k, val in out.items()}


def dae_backward(method: str, de_layers: Layers, ae_layers: Layers, t, z, v, all_initial, xs, is_, grad_xs, grad_is, event_idx=None,
                 z_jump=None, v_jump=None, kernel: str = "auto", saved=None):
    """Backward pass of `dae_integrate` (no teacher forcing): the one-launch K7f (`dae_backward_wide`) for the DAE_01 shape class at
    hidden <= 128, K9 / K8 / K9w for the latent shapes of the direct_encode models, else the generic backward kernel (K5);
    `kernel` = "auto" | "mfma" | "generic" | "wide" (K7f or an error).
    saved = what `dae_integrate(save=True)` returned (read by K7f, K9 and K9w; K8 / K5 recompute and refuse them).
    Returns dict(x_init, z, v, z_jump, v_jump, all_initial, de=[...], ae=[...]) of gradients."""
    lib = _lib.load()
    dev = xs.device
    T, B, xd = xs.shape
    zd, vd, idim = z.shape[-1], v.shape[-1], is_.shape[-1]
    if kernel in ("wide", "mfma") and T < 2 and len(de_layers) == 4:
        kernel = "generic"       # no step to sweep: K7f has no head-only form, K5 handles the single grid point
    if saved is not None and latent_wide_shape(de_layers, ae_layers, xd, zd, vd, idim):
        return latent_backward_wide(method, de_layers, ae_layers, t, z, v, all_initial, xs, is_, grad_xs, grad_is, event_idx=event_idx,
                                    z_jump=z_jump, v_jump=v_jump, saved=saved)
    if kernel == "wide" or (kernel in ("auto", "mfma") and len(de_layers) == 4 and T >= 2
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
