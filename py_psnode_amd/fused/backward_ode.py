"""Backward of `ode_integrate` (psnode_ode_backward_f32): K4f in one launch at hidden <= 128, K8f / K9 / K9w for the latent shapes of the
direct_encode models, the generic K5 otherwise."""
import ctypes

import torch

from .. import _lib
from ._common import (KERNEL_ID, Layers, METHOD_ID, _aligned16, _aligned_ptr, _check_saved, _empty, _f32_dev, _jump, _mlp, _split_grads, _view)
from .latent import latent_backward_wide, latent_wide_shape

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


def ode_backward(method: str, de_layers: Layers, t, z, all_initial, xs, grad_xs, event_idx=None, z_jump=None, need_grad_z: bool = True,
                 kernel: str = "auto", saved=None, input_true_x: bool = False, need_grad_zj: bool = True):
    """Backward pass of `ode_integrate` in one launch.  `saved` = what `ode_integrate(save=True)` returned next to
    xs: K4f then skips the recompute of the stage evaluations.  input_true_x: backward of a teacher-forced call (my_solvers.py:72-74) --
    `xs` must then be the DATASET x the forward call started every step from; K4f (hidden <= 128, x_dim <= 8) only.
    need_grad_z / need_grad_zj = False: dL/dz / dL/dz_jump are not formed (the scripts' z and z_jump are dataset tensors: K4x then runs
    without its per-step dL/dz layer, and the [B,nE,zd] zero fill is not made).
    kernel: "wave" = K4x (one wave per 4 trajectories; hidden 33..64, saved rows), "wide" / "tile" = K4f; "auto" picks between them.
    Returns (grad_x0 [B,xd], grad_z [T,B,zd] | None, grad_z_jump | None, grad_all_initial [B,n], [grad W1, b1, ..., W4, b4])."""
    lib = _lib.load()
    dev = xs.device
    T, B, xd = xs.shape
    zd = z.shape[-1]
    # kernel: "auto" / "mfma" = the one-launch K4f at every hidden width <= 128 (z_dim <= 8), K8f / K9 / K9w for the latent shapes, else the
    # generic K5 ("auto" only); "wide" forces K4f
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
        if z_jump is not None and zd > 0 and need_grad_zj:
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
