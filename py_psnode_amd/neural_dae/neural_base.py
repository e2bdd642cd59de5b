"""Names the reference exports from neural_dae/neural_base.py, for drop-in imports by the training scripts.

On the hot path: ODE_Event / DAE_Event (their tensors feed the fused kernel's event table) and the thin
ODE_Base / DAE_Base wrappers.  The npz datasets and the legacy per-variable DE_Func / AE_Func blocks are
host-side / dead code upstream; they are provided so `from neural_dae import ...`
keeps working and state dicts keep their key names.
"""
import numpy as np
import torch
import torch.nn as nn
from torch.utils.data import Dataset

from .my_solvers import FixedGridODESolver


# ----------------------------------------------------------------------------- events
class ODE_Event:
    """Step jumps of the external input z at fault times (neural_base.py:43-65).

    event_fn: trajectory 0's clock against trajectory 0's event list, exact float equality, decides for the
    whole batch.  jump_change_fn: the whole batch's z is replaced by z_jump[:, e] for that one step.
    """
    _psnode_event = True   # lets the fused path read event_t / z_jump instead of calling back per step

    def __init__(self):
        self.event_t = None
        self.z_jump = None

    def set_event(self, t: torch.Tensor, z: torch.Tensor):
        self.event_t = t
        self.z_jump = z

    def event_fn(self, t0: torch.Tensor):
        if self.event_t is None:
            return False
        return bool((self.event_t[0] == t0[0]).any())

    def _column(self, jump: torch.Tensor, t0: torch.Tensor, like: torch.Tensor):
        hit = (self.event_t[0] == t0[0][0]).view(-1)
        out = like.clone().detach()
        out[:] = jump[:, hit].view(out.shape)      # exactly one event may match, as upstream
        return out

    def jump_change_fn(self, t0: torch.Tensor, z0: torch.Tensor):
        return self._column(self.z_jump, t0, z0)


class DAE_Event(ODE_Event):
    """z and v jump together (neural_base.py:169-196)."""

    def __init__(self):
        super().__init__()
        self.v_jump = None

    def set_event(self, t: torch.Tensor, z: torch.Tensor, v: torch.Tensor):
        self.event_t = t
        self.z_jump = z
        self.v_jump = v

    def jump_change_fn(self, t0, z0, v0):
        return self._column(self.z_jump, t0, z0), self._column(self.v_jump, t0, v0)


# ----------------------------------------------------------------------------- datasets (npz format: SURVEY.md App. C)
class _Curves(Dataset):
    _series = ()
    _jumps = ()
    _mask_required = False

    def __init__(self, data_path, device, num_sample=None, cut_length=None, contain_larger_than=None):
        super().__init__()
        self.rng = np.random.default_rng(42)
        f = np.load(data_path, allow_pickle=True)
        total = f["t"].shape[0]
        while True:   # redraw the subsample until it contains a large excursion (neural_base.py:16-21)
            index = np.arange(total)
            if num_sample is not None:
                index = self.rng.choice(index, num_sample, replace=False)
            if contain_larger_than is None or np.any(f["x"][index] > contain_larger_than):
                break
        n_grid = f["t"].shape[1] if cut_length is None else min(cut_length, f["t"].shape[1])
        self.data_name = f["name"]
        for key in ("t",) + self._series:
            setattr(self, key, torch.from_numpy(f[key][index][:, 0:n_grid]))
        self.event_t = torch.from_numpy(f["event_t"][index])
        for key in self._jumps:
            setattr(self, key, torch.from_numpy(f[key][index]))
        if self._mask_required or "mask" in f.files:
            self.mask = torch.from_numpy(f["mask"][index][:, 0:n_grid])
        else:
            self.mask = torch.ones(self.x.shape).to(torch.float32)
        lengths = {getattr(self, key).shape[1] for key in ("t",) + self._series}
        assert len(lengths) == 1, "Sample shapes are wrong!"

    def __len__(self):
        return self.t.shape[0]


class ODE_Curves_Sample(_Curves):
    _series = ("x", "z")
    _jumps = ("z_jump",)

    def __getitem__(self, idx):
        return self.t[idx], self.x[idx], self.z[idx], self.event_t[idx], self.z_jump[idx], self.mask[idx]


class DAE_Curves_Sample(_Curves):
    _series = ("x", "z", "v", "i")
    _jumps = ("z_jump", "v_jump")
    _mask_required = True

    def __getitem__(self, idx):
        return (self.t[idx], self.x[idx], self.z[idx], self.v[idx], self.i[idx], self.event_t[idx],
                self.z_jump[idx], self.v_jump[idx], self.mask[idx])


# ----------------------------------------------------------------------------- legacy per-variable blocks
def _mlp2(n_in, hidden, n_out, act):
    return nn.Sequential(nn.Linear(n_in, hidden), act(), nn.Linear(hidden, n_out))


def _per_var(mods, h):
    """Apply module k to variable k of h[B, n_var, H] and restack along the variable axis."""
    return torch.cat([m(h[:, k:k + 1]) for k, m in enumerate(mods)], dim=-2)


class DE_Func(nn.Module):
    """Older architecture with one encoder per scalar variable, state [B, n_var, H] (neural_base.py:68-115).
    Not callable by the solvers' keyword convention -- standalone use only."""

    def __init__(self, x_dim, z_dim, hidden_dim):
        super().__init__()
        H = hidden_dim
        self.x_encoder, self.x_decoder = nn.ModuleList(), nn.ModuleList()
        self.Xh_Ext_H, self.Xh_dot_H = nn.ModuleList(), nn.ModuleList()
        for _ in range(x_dim):
            self.x_encoder.append(_mlp2(1, H, H, nn.Tanh))
            self.x_decoder.append(_mlp2(H, H, 1, nn.Tanh))
            self.Xh_Ext_H.append(_mlp2(H, H, H, nn.ELU))
            self.Xh_dot_H.append(_mlp2(H, H, H, nn.ELU))
        self.z_encoder, self.Zh_Ext_H = nn.ModuleList(), nn.ModuleList()
        for _ in range(z_dim):
            self.z_encoder.append(_mlp2(1, H, H, nn.Tanh))
            self.Zh_Ext_H.append(_mlp2(H, H, H, nn.ELU))
        self.Xh_dot_V = nn.Sequential(nn.Linear(int((x_dim + z_dim) * 3), H), nn.ELU(), nn.Linear(H, H), nn.ELU(),
                                      nn.Linear(H, H), nn.ELU(), nn.Linear(H, x_dim))

    def _features(self, Xh, z):
        fz = torch.cat([ext(enc(z[:, k:k + 1])) for k, (enc, ext) in enumerate(zip(self.z_encoder, self.Zh_Ext_H))], dim=-2)
        return torch.cat((_per_var(self.Xh_Ext_H, Xh), fz), dim=-2)

    def set_initial(self, x0, z0):
        self.Xh0 = self.get_encode_Xh(x0, z0)
        return self.Xh0

    def get_encode_Xh(self, x, z):
        Xh0 = _per_var(self.x_encoder, x)
        self.f_XZh0_H = self._features(Xh0, z)
        return Xh0

    def get_decode_x(self, Xh):
        return _per_var(self.x_decoder, Xh)

    def forward(self, t0, Xht, zt):
        f = self._features(Xht, zt)
        mixed = self.Xh_dot_V(torch.cat((f, self.f_XZh0_H, f - self.f_XZh0_H), dim=-2).permute(0, 2, 1)).permute(0, 2, 1)
        return _per_var(self.Xh_dot_H, mixed)


class AE_Func(nn.Module):
    """Legacy algebraic block (neural_base.py:199-229); standalone use only."""

    def __init__(self, x_dim, v_dim, i_dim, hidden_dim):
        super().__init__()
        H = hidden_dim
        self.Xh_Ext_H = nn.ModuleList([_mlp2(H, H, H, nn.ELU) for _ in range(x_dim)])
        self.z2_encoder, self.Z2h_Ext_H = nn.ModuleList(), nn.ModuleList()
        for _ in range(i_dim):
            self.z2_encoder.append(_mlp2(1, H, H, nn.Tanh))
            self.Z2h_Ext_H.append(_mlp2(H, H, H, nn.ELU))
        self.Yh_func_V = nn.Sequential(nn.Linear(int(x_dim + v_dim), H), nn.ELU(), nn.Linear(H, H), nn.ELU(),
                                       nn.Linear(H, H), nn.ELU(), nn.Linear(H, i_dim))
        self.y_decoder, self.Yh_Ext_H = nn.ModuleList(), nn.ModuleList()
        for _ in range(i_dim):
            self.y_decoder.append(_mlp2(H, H, 1, nn.Tanh))
            self.Yh_Ext_H.append(_mlp2(H, H, H, nn.ELU))

    def forward(self, Xht, vt):
        fv = torch.cat([ext(enc(vt[:, k:k + 1])) for k, (enc, ext) in enumerate(zip(self.z2_encoder, self.Z2h_Ext_H))], dim=-2)
        Yh = self.Yh_func_V(torch.cat((_per_var(self.Xh_Ext_H, Xht), fv), dim=-2).permute(0, 2, 1)).permute(0, 2, 1)
        return torch.cat([dec(ext(Yh[:, k:k + 1])) for k, (dec, ext) in enumerate(zip(self.y_decoder, self.Yh_Ext_H))], dim=-2)


# ----------------------------------------------------------------------------- thin wrappers
class ODE_Base(nn.Module):
    """Pass-through to solver.integrate_ODE (neural_base.py:118-133)."""

    def __init__(self, de_func, solver: FixedGridODESolver, flg_encode_x=False):
        super().__init__()
        self.de_function = de_func
        self.solver = solver

    def forward(self, t, x, z, all_initial, event_fn=None, jump_change_fn=None):
        return self.solver.integrate_ODE(x_func=self.de_function, t=t, x=x, z=z, all_initial=all_initial,
                                         event_fn=event_fn, jump_change_fn=jump_change_fn)


class DAE_Base(nn.Module):
    """Stores the reference's fields (neural_base.py:232-242).  Upstream's forward cannot run (it passes a
    non-existent `encode_x` kwarg and omits x_init / all_initial); this one takes them
    explicitly and forwards the teacher-forcing flags."""

    def __init__(self, de_func, ae_func, solver: FixedGridODESolver, flg_encode_x=False, flg_input_true_x=False, flg_input_true_i=False):
        super().__init__()
        self.de_function = de_func
        self.ae_function = ae_func
        self.solver = solver
        self.flg_encode_x = flg_encode_x
        self.flg_input_true_x = flg_input_true_x
        self.flg_input_true_i = flg_input_true_i

    def forward(self, t, x, z, v, i, event_fn=None, jump_change_fn=None, x_init=None, all_initial=None):
        if x_init is None or all_initial is None:
            raise TypeError("DAE_Base.forward needs x_init and all_initial (integrate_DAE has no defaults for them)")
        return self.solver.integrate_DAE(x_init=x_init, x_func=self.de_function, i_func=self.ae_function, t=t, x=x, z=z,
                                         v=v, i=i, all_initial=all_initial, event_fn=event_fn, jump_change_fn=jump_change_fn,
                                         input_true_x=self.flg_input_true_x, input_true_i=self.flg_input_true_i)
