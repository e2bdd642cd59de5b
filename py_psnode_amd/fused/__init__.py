"""Host side of the fused HIP integrator: recognise the reference's MLP right-hand sides, marshal
tensors into the C ABI (include/psnode_hip.h) and enqueue on the caller's current HIP stream.

Raw-tensor entry points (`ode_integrate`, `dae_integrate`, `ode_backward`, `dae_backward`, ...) take MLPs as [(W[out,in], b[out]), ...]
exactly as nn.Linear stores them; `plan_ode` / `plan_dae` do the recognition for the solver classes in py_psnode_amd.neural_dae.
Nothing here computes on the CPU and nothing here imports oracle/.  One module per kernel family:
    _common       marshalling, MLP recognition, event table        forward       ode_integrate / dae_integrate
    backward_ode  K4f / K8f / K9 / K5                              backward_dae  K7f (+ K7h) / K9 / K8 / K5
    latent        K9w + the contractions over its rows             rows          encoder / decoder row MLPs (K3b)
    encoded       the one-launch direct_encode model forwards      plan          fusability of a solver call
"""
from .. import _lib  # noqa: F401
from . import _common, backward_dae, backward_ode, encoded, forward, latent, plan, rows  # noqa: F401
from ._common import (Layers, METHOD_ID, KERNEL_ID, _POISON, _empty, sequential_layers, de_layers_of, ae_layers_of, _mlp_eval, _recipe_ok, _overrides_forward_hooks, _only_params_of, _f32_dev, _view, _aligned16, _mlp, _check_tb, _check_jump, _jump, event_table, _DUP_OK, _dup_key, _dup_check_known, _dup_check_remember, _workspace, _aligned_ptr, _padded_hidden, _pad_rows, _gemm_tn, _check_saved, _split_grads)  # noqa: F401
from .forward import (_MFMA_CLASSES, _k0_warned, _mfma_miss, _note_k0, ode_integrate, ode_save_hidden, dae_integrate, dae_save_hidden)  # noqa: F401
from .backward_ode import (_bwd_args, ode_backward_supported, ode_backward)  # noqa: F401
from .backward_dae import (dae_backward_supported, dae_backward_wide_supported, dae_backward_wide, _dae_backward_wide_sliced, dae_backward)  # noqa: F401
from .latent import (latent_wide_shape, latent_backward_wide)  # noqa: F401
from .rows import (mlp_rows, mlp_rows_backward, mlp_rows_backward_multi, _RowsMlp, _RowsMlpMulti, mlp_rows_autograd, mlp_rows_autograd_multi, rows_layers_of, recon_rows_supported, recon_rows, recon_rows_backward, _ReconRows, linear_rows, wide_mlp_rows, wide_rows_class, _WideRowsMlp,
                   recon_rows_autograd)  # noqa: F401
from .encoded import (ode_encoded_supported, ode_encoded_integrate, _dae_encoded_args, dae_encoded_supported, dae_encoded_integrate)  # noqa: F401
from .plan import (_event_tensors, _needs_autograd, _all_f32_on, plan_ode, plan_dae)  # noqa: F401
