"""The model glue of the four reference training scripts, as one parametrised pair of classes.

    ODE_Model(x_dim, z_dim, hidden_dim, direct_encode=False)   neural_00_ODE_01_no_encode.py:71-91
                                                               neural_00_ODE_02_direct_encode.py:60-89
    DAE_Model(x_dim, z_dim, v_dim, i_dim, hidden_dim, direct_encode=False)
                                                               neural_01_DAE_01_no_encode.py:86-115
                                                               neural_01_DAE_02_direct_encode.py:103-153

Sub-module and parameter names equal the scripts' (`de_func.x_dot.0.weight`, `x_encoder.2.bias`,
`init_func.init_fun.4.weight`, ...), so their checkpoints load with load_state_dict.  Inputs are B-major
[B,T,D]; the solver gets permute(1,0,2) views and the result is permuted back, as upstream.
The scripts hard-code Euler(); pass `solver=` or assign `model.solver = RK4()`.
"""
import os

import torch
import torch.nn as nn

from .neural_dae import DAE_Event, Euler, ODE_Event


def _elu_mlp(*dims):
    mods = []
    for k in range(len(dims) - 1):
        mods.append(nn.Linear(int(dims[k]), int(dims[k + 1])))
        if k + 2 < len(dims):
            mods.append(nn.ELU())
    return nn.Sequential(*mods)


def _tm(a):
    return a.permute(1, 0, 2)


def _time_major_route(x) -> bool:
    """fp32 HIP tensors: the direct_encode models lay their latent tensors out time-major (the fused kernels' layout)."""
    return x.device.type == "cuda" and x.dtype == torch.float32 and x.dim() == 3


def _rows(seq, inp):
    """Encoder / decoder over every (b,t) row on the fused HIP row kernels when the call is fusable (fp32 HIP tensor,
    Linear-ELU-Linear with hidden 16 / 64): forward kernel alone without autograd, forward + fused backward kernel under it;
    otherwise the module itself (CPU, other widths)."""
    from . import fused
    if fused._overrides_forward_hooks(seq) or any(fused._overrides_forward_hooks(m) for m in seq):
        return seq(inp)          # user hooks on the module or its layers must fire: the module itself, layer by layer
    layers = fused.rows_layers_of(seq, inp, allow_grad=True)
    if layers is None:
        return seq(inp)
    needs_grad = torch.is_grad_enabled() and (inp.requires_grad or any(p.requires_grad for p in seq.parameters()))
    return fused.mlp_rows_autograd(seq, inp) if needs_grad else fused.mlp_rows(layers, inp)


def _rows_multi(seq, *inps):
    """`tuple(seq(a) for a in inps)` -- ONE autograd node on the row kernels when every call is fusable and something needs a gradient
    (the module's gradient is then one fixed-order reduction over all the sets' partials instead of a reduction per set and an autograd
    `add` per parameter tensor and extra use); otherwise call by call (`_rows`)."""
    from . import fused
    if len(inps) > 1 and torch.is_grad_enabled() and not (fused._overrides_forward_hooks(seq) or any(fused._overrides_forward_hooks(m) for m in seq)):
        if all(fused.rows_layers_of(seq, a, allow_grad=True) is not None for a in inps) and \
                (any(a.requires_grad for a in inps) or any(p.requires_grad for p in seq.parameters())):
            return fused.mlp_rows_autograd_multi(seq, *inps)
    return tuple(_rows(seq, a) for a in inps)


def _recon(encoder, decoder, inp):
    """`decoder(encoder(inp))` of DATA rows on the fused reconstruction kernels (K3r), or None when the pair is not fusable there (hooks, other
    widths, an input that needs a gradient itself, CPU)."""
    from . import fused
    if inp.requires_grad or inp.device.type != "cuda" or os.environ.get("PSNODE_NO_RECON") == "1":
        return None
    if any(fused._overrides_forward_hooks(m) for seq in (encoder, decoder) for m in (seq, *seq)):
        return None
    le, ld = fused.sequential_layers(encoder), fused.sequential_layers(decoder)
    if le is None or ld is None or not fused.recon_rows_supported(le, ld, inp):
        return None
    needs_grad = torch.is_grad_enabled() and any(p.requires_grad for seq in (encoder, decoder) for p in seq.parameters())
    return fused.recon_rows_autograd(encoder, decoder, inp) if needs_grad else fused.recon_rows(le, ld, inp)


class RowsSequential(nn.Sequential):
    """An `nn.Sequential(Linear, ELU, Linear)` -- same children, same parameters, same state-dict keys -- whose forward runs on the
    fused HIP row kernels (psnode_mlp_rows_f32 / _backward_f32) whenever the call is fusable (fp32 HIP tensor, hidden 16 / 64), and
    is the plain Sequential otherwise (CPU, other widths, TorchScript export: `torch.jit.script` compiles only the plain loop)."""

    def forward(self, input):
        if not torch.jit.is_scripting():
            return _rows_of_accelerated(self, input)
        for module in self:
            input = module(input)
        return input


@torch.jit.unused
def _rows_of_accelerated(seq, inp):
    from . import fused
    layers = fused.rows_layers_of(seq, inp, allow_grad=True)
    if layers is None:
        return nn.Sequential.forward(seq, inp)
    needs_grad = torch.is_grad_enabled() and (inp.requires_grad or any(p.requires_grad for p in seq.parameters()))
    return fused.mlp_rows_autograd(seq, inp) if needs_grad else fused.mlp_rows(layers, inp)


def accelerate(model: nn.Module, names=("_encoder", "_decoder")) -> nn.Module:
    """Put the encoders / decoders of a direct_encode model -- the REFERENCE'S OWN ODE_Model / DAE_Model of
    neural_00_ODE_02_direct_encode.py:64-69 / neural_01_DAE_02_direct_encode.py:107-118 included -- on the fused row kernels:
    every direct child whose name ends in `_encoder` / `_decoder` and that is exactly `nn.Sequential(Linear, ELU(1), Linear)` has
    its class swapped in place to `RowsSequential`.  Parameters, buffers, state-dict keys, optimizer references and the
    TorchScript export (`save_model`) are unchanged; `accelerate` is idempotent and returns the model."""
    from . import fused
    for name, child in model.named_children():
        if not name.endswith(tuple(names)) or type(child) is not nn.Sequential:
            continue
        layers = fused.sequential_layers(child)
        if layers is None or len(layers) != 2:
            continue
        child.__class__ = RowsSequential
    return model


class DE_Func(nn.Module):
    """ODE right-hand side: x_dot MLP over cat(a0, s - a0, s), s = cat(xt, zt).  Positional order of forward() as in
    neural_00_ODE_01_no_encode.py:66 -- the TorchScript export (`save_model`) is called positionally downstream."""

    def __init__(self, state_width, hidden_dims, out_dim):
        super().__init__()
        self.x_dot = _elu_mlp(3 * state_width, *hidden_dims, out_dim)

    def forward(self, t0: torch.Tensor, xt: torch.Tensor, zt: torch.Tensor, all_initial: torch.Tensor):
        s = torch.cat((xt, zt), dim=-1)
        return self.x_dot(torch.cat((all_initial, s - all_initial, s), dim=-1))

    forward._psnode_recipe = "de_ode"     # known to follow the kernels' input recipe (fused._recipe_ok skips the numeric probe)


class DAE_DE_Func(nn.Module):
    """DAE right-hand side, s = cat(xt, zt, vt, it); positional order of neural_01_DAE_01_no_encode.py:69."""

    def __init__(self, state_width, hidden_dims, out_dim):
        super().__init__()
        self.x_dot = _elu_mlp(3 * state_width, *hidden_dims, out_dim)

    def forward(self, t0: torch.Tensor, xt: torch.Tensor, zt: torch.Tensor, vt: torch.Tensor, it: torch.Tensor, all_initial: torch.Tensor):
        s = torch.cat((xt, zt, vt, it), dim=-1)
        return self.x_dot(torch.cat((all_initial, s - all_initial, s), dim=-1))

    forward._psnode_recipe = "de_dae"


def _export(model, path, names, on_cpu):
    """torch.jit.script each named sub-module into `path`/<name>.pt -- the files the scripts' save_model / final_save
    write for the downstream C++ consumer (neural_00_ODE_01_no_encode.py:93-101, neural_01_DAE_02_direct_encode.py:155-203).
    direct_encode models also write hidden_dim into dim.txt."""
    import pathlib
    path = pathlib.Path(path)
    if not path.exists():
        path.mkdir()
    if getattr(model, "direct_encode", False):
        with open(str(path / "dim.txt"), "w") as f:
            f.write(str(model.hidden_dim))
    for name in names:
        mod = getattr(model, name, None)
        if mod is None:            # DAE_02 with z_dim == 0 has no z_encoder
            continue
        torch.jit.script(mod.to("cpu") if on_cpu else mod).save(str(path / f"{name}.pt"))


class AE_Func(nn.Module):
    """i_calculator MLP over cat(a0, xt, zt, vt)."""

    def __init__(self, in_width, hidden_dims, out_dim):
        super().__init__()
        self.i_calculator = _elu_mlp(in_width, *hidden_dims, out_dim)

    def forward(self, xt: torch.Tensor, zt: torch.Tensor, vt: torch.Tensor, all_initial: torch.Tensor):
        return self.i_calculator(torch.cat((all_initial, xt, zt, vt), dim=-1))

    forward._psnode_recipe = "ae"


class Init_Func(nn.Module):
    def __init__(self, x_dim, z_dim, v_dim, i_dim, hidden_dim):
        super().__init__()
        self.init_fun = _elu_mlp(z_dim + v_dim + i_dim, hidden_dim, hidden_dim, x_dim)

    def forward(self, z0: torch.Tensor, v0: torch.Tensor, i0: torch.Tensor):
        return self.init_fun(torch.cat([z0, v0, i0], dim=-1))


class ODE_Model(nn.Module):
    def __init__(self, x_dim, z_dim, hidden_dim, direct_encode=False, solver=None, enc_hidden=None):
        """`enc_hidden` is an EXTENSION to the reference (whose one `hidden_dim` drives encoder width, latent width and RHS width
        alike, neural_00_ODE_02_direct_encode.py:52-53,64-70): encoders / decoder with their own hidden width around a
        `hidden_dim`-wide latent space -- BASELINE's "enc/dec 64 -> 16 latent" reading of the direct_encode config.  None = upstream."""
        super().__init__()
        H = hidden_dim
        E = H if enc_hidden is None else int(enc_hidden)
        self.hidden_dim = H
        self.enc_hidden = E
        self.direct_encode = direct_encode
        if direct_encode:
            self.x_encoder = _elu_mlp(x_dim, E, H)
            self.x_decoder = _elu_mlp(H, E, x_dim)
            self.z_encoder = _elu_mlp(z_dim, E, H)
            self.de_func = DE_Func(2 * H, (H,), H)                 # Linear(6H,H) ELU Linear(H,H)
        else:
            self.de_func = DE_Func(x_dim + z_dim, (H, H, H), x_dim)
        self.solver = solver if solver is not None else Euler()
        self.event = ODE_Event()

    def forward(self, t, x, z, event_t, z_jump):
        if not self.direct_encode:
            self.event.set_event(t=event_t, z=z_jump)
            a0 = torch.cat((_tm(x)[0], _tm(z)[0]), dim=-1)
            xs = self.solver.integrate_ODE(x_func=self.de_func, t=_tm(t), x=_tm(x), z=_tm(z), all_initial=a0,
                                           event_fn=self.event.event_fn, jump_change_fn=self.event.jump_change_fn)
            return _tm(xs)
        fused_out = self._forward_encoded(t, x, z, event_t, z_jump)
        if fused_out is not None:
            return fused_out
        if _time_major_route(x):
            # HIP route: the encoders run over the TIME-MAJOR views of the raw inputs (a few floats per row to gather) and write the
            # latent tensors in the layout the integrator and its backward use, so no [T,B,H] tensor is ever permuted, copied or summed
            # on the way there or back; the first row is encoded once more for all_initial (B rows) instead of being selected out of the
            # big tensor, whose gradient would otherwise be a zero-filled [T,B,H] tensor added to the integrator's.  Row-wise functions:
            # the values are the ones of the B-major evaluation.
            from .neural_dae.my_solvers import FixedGridODESolver
            own = isinstance(self.solver, FixedGridODESolver)
            Zh, z0h, zjh = _rows_multi(self.z_encoder, _tm(z), z[:, 0], z_jump)
            # this package's solvers take the start state on its own (x_init): Xh then reaches the integrator only for its shape, and its
            # gradient is the decoder's alone instead of that plus a [T,B,H] tensor of zeros with one row set.  With that, the encoded rows
            # Xh = x_encoder(x) have ONE consumer left -- the reconstruction x_decoder(Xh) -- and at hidden 16 that whole branch is one
            # kernel each way (fused.recon_rows_autograd, K3r): Xh never reaches memory.
            x_re = _recon(self.x_encoder, self.x_decoder, _tm(x)) if own else None
            if x_re is not None:
                x0h = _rows(self.x_encoder, x[:, 0])
                Xh_shape = x0h.detach().unsqueeze(0).expand(x.shape[1], -1, -1)       # [T,B,H] for the solver's shape checks only
            else:
                Xh, x0h = _rows_multi(self.x_encoder, _tm(x), x[:, 0])
                Xh_shape = Xh.detach() if own else Xh
            a0 = torch.cat((x0h, z0h), dim=-1)
            self.event.set_event(t=event_t, z=zjh)
            Xh_sol = self.solver.integrate_ODE(x_func=self.de_func, t=_tm(t), x=Xh_shape, z=Zh, all_initial=a0,
                                               event_fn=self.event.event_fn, jump_change_fn=self.event.jump_change_fn,
                                               **({"x_init": x0h} if own else {}))
            if x_re is not None:
                return _tm(_rows(self.x_decoder, Xh_sol)), _tm(x_re)
            x_pred, x_re = _rows_multi(self.x_decoder, Xh_sol, Xh)
            return _tm(x_pred), _tm(x_re)
        Xh_bt = _rows(self.x_encoder, x)                          # [B,T,H]; the solver gets the usual permuted view
        Xh = _tm(Xh_bt)
        Zh = _tm(_rows(self.z_encoder, z))
        a0 = torch.cat((Xh[0], Zh[0]), dim=-1)
        self.event.set_event(t=event_t, z=_rows(self.z_encoder, z_jump))
        Xh_sol = self.solver.integrate_ODE(x_func=self.de_func, t=_tm(t), x=Xh, z=Zh, all_initial=a0,
                                           event_fn=self.event.event_fn, jump_change_fn=self.event.jump_change_fn)
        # reconstruction: decoding the B-major tensor gives the same [B,T,xd] values as upstream's
        # x_decoder(Xh).permute(1,0,2) without first materialising the permuted view
        return _tm(_rows(self.x_decoder, Xh_sol)), _rows(self.x_decoder, Xh_bt)


    def _forward_encoded(self, t, x, z, event_t, z_jump):
        """The whole direct_encode forward in ONE HIP launch (psnode_ode_encoded_integrate_f32: encoders, latent integration,
        decoder, reconstruction -- Xh / Zh / Xh_solution never reach memory) when nothing needs autograd, the tensors are fp32 on a
        HIP device, hidden_dim is 16 and the solver is one of this package's with fusing allowed; None otherwise (the caller then
        takes the row kernels + solver route, which is also the training route)."""
        from . import fused
        from .neural_dae.my_solvers import FixedGridODESolver
        solver = self.solver
        if not isinstance(solver, FixedGridODESolver) or getattr(solver, "fused", "off") == "off" or not solver.method:
            return None
        if getattr(solver, "kernel", "auto") == "generic":        # PSNODE_KERNEL=generic asks for K0: row kernels + solver route
            return None
        if any(fused._overrides_forward_hooks(m) for m in (self, self.x_encoder, self.z_encoder, self.x_decoder, self.de_func,
                                                          self.de_func.x_dot)):
            return None                                            # user hooks must fire: module-by-module route
        if x.device.type != "cuda" or any(a.dtype != torch.float32 for a in (t, x, z)) or x.dim() != 3 or x.shape[1] < 1:
            return None
        if not getattr(type(self.event), "_psnode_event", False) or type(self.de_func) is not DE_Func:
            return None
        mlps = [fused.sequential_layers(m) for m in (self.x_encoder, self.z_encoder, self.x_decoder, self.de_func.x_dot)]
        if any(m is None for m in mlps) or not fused.ode_encoded_supported(*mlps):
            return None
        if fused._needs_autograd([x, z, z_jump] + [p for m in mlps for wb in m for p in wb]):
            return None
        if event_t is not None and (event_t.dtype != torch.float32 or z_jump is None or z_jump.dtype != torch.float32):
            return None
        # NOTE: upstream (and the module-by-module route) leave event.z_jump = z_encoder(z_jump) [B,nE,H]; this route never forms
        # the encoded jumps, so the event object holds the RAW [B,nE,z_dim] tensor afterwards (nothing reads it after forward)
        self.event.set_event(t=event_t, z=z_jump)
        x_pred, x_re, _ = fused.ode_encoded_integrate(solver.method, *mlps, t, x, z, event_t=event_t, z_jump=z_jump,
                                                      check_events=solver._check_events_now(event_t))
        return x_pred, x_re

    _EXPORTS = ("x_encoder", "x_decoder", "z_encoder", "de_func")

    def save_model(self, path):
        """neural_00_ODE_01_no_encode.py:93-96 / neural_00_ODE_02_direct_encode.py:91-102"""
        _export(self, path, self._EXPORTS if self.direct_encode else ("de_func",), on_cpu=False)

    def final_save(self, path):
        """as save_model, after moving the sub-modules to the CPU (neural_00_ODE_01_no_encode.py:98-101)"""
        _export(self, path, self._EXPORTS if self.direct_encode else ("de_func",), on_cpu=True)


class DAE_Model(nn.Module):
    def __init__(self, x_dim, z_dim, v_dim, i_dim, hidden_dim, direct_encode=False, solver=None):
        super().__init__()
        H = hidden_dim
        self.hidden_dim = H
        self.direct_encode = direct_encode
        if direct_encode:
            self.x_encoder = _elu_mlp(x_dim, H, H)
            self.x_decoder = _elu_mlp(H, H, x_dim)
            self.z_encoder = _elu_mlp(z_dim, H, H) if z_dim != 0 else None
            self.v_encoder = _elu_mlp(v_dim, H, H)
            self.i_encoder = _elu_mlp(i_dim, H, H)
            self.i_decoder = _elu_mlp(H, H, i_dim)
        self.init_func = Init_Func(x_dim, z_dim, v_dim, i_dim, H)
        if direct_encode:
            parts = 3 if z_dim == 0 else 4                         # latent blocks in all_initial (x, [z,] v, i)
            self.de_func = DAE_DE_Func(parts * H, (H,), H)         # Linear(12H|9H, H) ELU Linear(H,H)
            self.ae_func = AE_Func((2 * parts - 1) * H, (H,), H)   # Linear(7H|5H, H) ELU Linear(H,H)
        else:
            n = x_dim + z_dim + v_dim + i_dim
            self.de_func = DAE_DE_Func(n, (H, H, H), x_dim)
            self.ae_func = AE_Func(n + x_dim + z_dim + v_dim, (H, H, H), i_dim)
        self.solver = solver if solver is not None else Euler()
        self.event = DAE_Event()

    def forward(self, t, x, z, v, i, event_t, z_jump, v_jump, input_true_x=False, input_true_i=False):
        x0 = self.init_func(z0=_tm(z)[0], v0=_tm(v)[0], i0=_tm(i)[0])
        if not self.direct_encode:
            self.event.set_event(t=event_t, z=z_jump, v=v_jump)
            a0 = torch.cat((x0, _tm(z)[0], _tm(v)[0], _tm(i)[0]), dim=-1)
            xs, is_ = self.solver.integrate_DAE(x_init=x0, x_func=self.de_func, i_func=self.ae_func, t=_tm(t), x=_tm(x),
                                                z=_tm(z), v=_tm(v), i=_tm(i), all_initial=a0,
                                                event_fn=self.event.event_fn, jump_change_fn=self.event.jump_change_fn,
                                                input_true_x=input_true_x, input_true_i=input_true_i)
            return _tm(xs), _tm(is_)
        fused_out = self._forward_encoded(x0, t, x, z, v, i, event_t, z_jump, v_jump)
        if fused_out is not None:
            return fused_out
        enc_z = (lambda a: a) if self.z_encoder is None else (lambda a: _rows(self.z_encoder, a))
        Xh0 = _rows(self.x_encoder, x0)
        if _time_major_route(x):      # HIP route: time-major latent tensors, first row re-encoded for all_initial (see ODE_Model.forward)
            Xh = _rows(self.x_encoder, _tm(x))
            Ih, i0h = _rows_multi(self.i_encoder, _tm(i), i[:, 0])
            Vh, v0h, vjh = _rows_multi(self.v_encoder, _tm(v), v[:, 0], v_jump)
            Zh, z0h, zjh = (_tm(z), z[:, 0], z_jump) if self.z_encoder is None else _rows_multi(self.z_encoder, _tm(z), z[:, 0], z_jump)
            a0 = torch.cat((Xh0, z0h, v0h, i0h), dim=-1)
            self.event.set_event(t=event_t, z=zjh, v=vjh)
            Xh_sol, Ih_sol = self.solver.integrate_DAE(x_init=Xh0, x_func=self.de_func, i_func=self.ae_func, t=_tm(t), x=Xh,
                                                       z=Zh, v=Vh, i=Ih, all_initial=a0, event_fn=self.event.event_fn,
                                                       jump_change_fn=self.event.jump_change_fn)
            x_pred, x_re = _rows_multi(self.x_decoder, Xh_sol, Xh)
            i_pred, i_re = _rows_multi(self.i_decoder, Ih_sol, Ih)
            x_pred[0] = x0                                         # neural_01_DAE_02_direct_encode.py:150
            return _tm(x_pred), _tm(i_pred), _tm(x_re), _tm(i_re)
        Xh_bt, Ih_bt = _rows(self.x_encoder, x), _rows(self.i_encoder, i)
        Xh, Zh, Vh, Ih = _tm(Xh_bt), _tm(enc_z(z)), _tm(_rows(self.v_encoder, v)), _tm(Ih_bt)
        a0 = torch.cat((Xh0, Zh[0], Vh[0], Ih[0]), dim=-1)
        self.event.set_event(t=event_t, z=enc_z(z_jump), v=_rows(self.v_encoder, v_jump))
        Xh_sol, Ih_sol = self.solver.integrate_DAE(x_init=Xh0, x_func=self.de_func, i_func=self.ae_func, t=_tm(t), x=Xh,
                                                   z=Zh, v=Vh, i=Ih, all_initial=a0, event_fn=self.event.event_fn,
                                                   jump_change_fn=self.event.jump_change_fn)
        x_pred = _rows(self.x_decoder, Xh_sol)
        x_pred[0] = x0                                             # neural_01_DAE_02_direct_encode.py:150
        return _tm(x_pred), _tm(_rows(self.i_decoder, Ih_sol)), _rows(self.x_decoder, Xh_bt), _rows(self.i_decoder, Ih_bt)

    one_launch = None      # True / False: take / never take the one-launch forward K3g; None: PSNODE_DAE02_ONE_LAUNCH (default off, see below)

    def _forward_encoded(self, x0, t, x, z, v, i, event_t, z_jump, v_jump):
        """The whole direct_encode forward behind Init_Func in ONE HIP launch (psnode_dae_encoded_integrate_f32, K3g: four encoders,
        all_initial, latent integrate_DAE, both decoders, both reconstructions -- none of the six latent [T,B,64] tensors reaches
        memory) when nothing needs autograd, the tensors are fp32 on a HIP device, hidden_dim is 64 and the solver is one of this
        package's with fusing allowed; None otherwise (row kernels + solver route: also the training route)."""
        from . import fused
        from .neural_dae.my_solvers import FixedGridODESolver
        solver = self.solver
        if not isinstance(solver, FixedGridODESolver) or getattr(solver, "fused", "off") == "off" or not solver.method:
            return None
        if getattr(solver, "kernel", "auto") == "generic":
            return None
        # Measured (profiles/r04m_dae02_routes.txt, 4096 x 1000 steps): the one launch moves 140 B per state-step instead of 3.4 KB (13.7 GB vs 0.586 GB per batch, measured) and needs
        # none of the six [T,B,64] latent tensors (6.3 GB at that size), but at hidden 64 the encoders / decoders are MFMA work that the
        # row kernels run at full occupancy, and it is 8 % (RK4) / 12 % (Euler) SLOWER than the row kernels + K3c: opt-in
        # (model.one_launch = True or PSNODE_DAE02_ONE_LAUNCH=1), the row-kernel route stays the default.
        want = self.one_launch if self.one_launch is not None else os.environ.get("PSNODE_DAE02_ONE_LAUNCH", "0") == "1"
        if not want:
            return None
        mods = [self.x_encoder, self.z_encoder, self.v_encoder, self.i_encoder, self.x_decoder, self.i_decoder, self.de_func.x_dot,
                self.ae_func.i_calculator]
        if any(m is not None and fused._overrides_forward_hooks(m) for m in [self, self.de_func, self.ae_func] + mods):
            return None
        if v.device.type != "cuda" or any(q.dtype != torch.float32 for q in (t, x, z, v, i, x0)) or v.dim() != 3 or v.shape[1] < 1:
            return None
        if not getattr(type(self.event), "_psnode_event", False) or type(self.de_func) is not DAE_DE_Func or type(self.ae_func) is not AE_Func:
            return None
        if (self.z_encoder is None) != (z.shape[-1] == 0) or x.shape[-1] != x0.shape[-1]:
            return None
        mlps = [None if m is None else fused.sequential_layers(m) for m in mods]
        if any(m is None for k, m in enumerate(mlps) if k != 1) or (self.z_encoder is not None and mlps[1] is None):
            return None
        if not fused.dae_encoded_supported(*mlps):
            return None
        if fused._needs_autograd([x0, x, z, v, i, z_jump, v_jump] + [p for m in mlps if m is not None for wb in m for p in wb]):
            return None
        if event_t is not None and (event_t.dtype != torch.float32 or v_jump is None or v_jump.dtype != torch.float32
                                    or (self.z_encoder is not None and (z_jump is None or z_jump.dtype != torch.float32))):
            return None
        # as ODE_Model._forward_encoded: the event object holds the RAW jump tensors afterwards (nothing reads them after forward)
        self.event.set_event(t=event_t, z=z_jump, v=v_jump)
        return fused.dae_encoded_integrate(solver.method, *mlps, x0, t, x, z, v, i, event_t=event_t, z_jump=z_jump, v_jump=v_jump,
                                           check_events=solver._check_events_now(event_t))

    _EXPORTS = ("x_encoder", "x_decoder", "z_encoder", "v_encoder", "i_encoder", "i_decoder", "init_func", "de_func", "ae_func")

    def save_model(self, path):
        """neural_01_DAE_01_no_encode.py:117-124 / neural_01_DAE_02_direct_encode.py:155-177"""
        _export(self, path, self._EXPORTS if self.direct_encode else self._EXPORTS[6:], on_cpu=False)

    def final_save(self, path):
        _export(self, path, self._EXPORTS if self.direct_encode else self._EXPORTS[6:], on_cpu=True)
