"""ctypes binding of libpsnode_hip.so (the C ABI declared in include/psnode_hip.h).

The library is the product's only compute path.  There is no fallback: if it cannot be loaded,
`load()` raises `PsnodeLibraryError` and every fused call fails loudly.
"""
import ctypes
import os
from ctypes import c_char_p, c_int32, c_int64, c_size_t, c_uint32, c_void_p

MAX_LAYERS = 8
MAX_WIDTH = 1024

OK = 0
EULER, MIDPOINT, RK4_38 = 0, 1, 2
KERNEL_AUTO, KERNEL_GENERIC, KERNEL_MFMA, KERNEL_MFMA_WIDE, KERNEL_MFMA_TILE, KERNEL_MFMA_WAVE = 0, 1, 2, 3, 4, 5
FLAG_INPUT_TRUE_X, FLAG_INPUT_TRUE_I = 1, 2

ABI_VERSION = 10         # == PSNODE_ABI_VERSION of include/psnode_hip.h (2: round-2 exports + arg structs, folded forward image;
                         #  3: save_act / save_xstage in the ODE forward args, saved_* in the backward args, psnode_ode_save_hidden;
                         #  4: the DAE's save_* / saved_* / fused-DE outputs in psnode_dae_args_f32 / psnode_dae_bwd_wide_args_f32,
                         #     psnode_dae_save_hidden;
                         #  5: flags (+ x_true / i_true) in psnode_ode_bwd_args_f32 / psnode_dae_bwd_wide_args_f32: teacher-forced backward;
                         #  6: psnode_dae_encoded_*: the DAE_02 model forward in one launch;
                         #  7: psnode_latent_backward_wide_*: the adjoint sweep of the latent integrators at hidden widths other than 16 / 64;
                         #  8: the split backward forms are gone -- no psnode_ode_backward_wide_*, no k0 / k1 / stored DE rows in
                         #     psnode_dae_bwd_wide_args_f32 -- and PSNODE_KERNEL_MFMA_TILE / _WAVE select K1 / K1x for forward ODE calls;
                         #  9: row addressing (inner rows / outer stride) in psnode_mlp_rows_*, psnode_recon_rows_*, psnode_mlp_rows_reduce_f32 /
                         #     _backward_parts: the row kernels read [B,T,D] batches as time-major rows in place;
                         # 10: psnode_gemm_tn_* / psnode_linear_rows_*: the contraction over rows (K10) and the row-linear layer (K11) that replace library GEMMs; the `kernel` field of
                         #     psnode_ode_bwd_args_f32 selects K4x (_MFMA_WAVE) / K4f (_MFMA_TILE, _MFMA_WIDE))
LIB_NAME = "libpsnode_hip.so"
# PSNODE_LIB_PATH lets kernel experiments (profiles/scripts/*) load an alternative build of the same ABI
LIB_PATH = os.environ.get("PSNODE_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)

EXPORTS = (
    "psnode_abi_version", "psnode_build_info", "psnode_status_string", "psnode_workspace_bytes",
    "psnode_event_table_f32", "psnode_ode_integrate_f32", "psnode_dae_integrate_f32",
    "psnode_ode_kernel_for", "psnode_dae_kernel_for", "psnode_ode_save_hidden", "psnode_dae_save_hidden", "psnode_dae_head_grads_out_floats",
    "psnode_dae_head_grads_workspace_bytes", "psnode_dae_head_grads_f32", "psnode_mlp_rows_supported", "psnode_mlp_rows_f32",
    "psnode_ode_backward_supported", "psnode_ode_backward_param_count", "psnode_ode_backward_workspace_bytes",
    "psnode_ode_backward_f32", "psnode_dae_backward_supported", "psnode_dae_backward_workspace_bytes", "psnode_dae_backward_f32",
    "psnode_masked_mse_workspace_bytes", "psnode_masked_mse_f32",
    "psnode_mlp_rows_backward_workspace_bytes", "psnode_mlp_rows_backward_f32",
    "psnode_mlp_rows_backward_parts", "psnode_mlp_rows_reduce_workspace_bytes", "psnode_mlp_rows_reduce_f32",
    "psnode_recon_rows_supported", "psnode_recon_rows_f32", "psnode_recon_rows_param_count", "psnode_recon_rows_backward_workspace_bytes",
    "psnode_recon_rows_backward_f32",
    "psnode_ode_encoded_supported", "psnode_ode_encoded_integrate_f32",
    "psnode_dae_encoded_supported", "psnode_dae_encoded_workspace_bytes", "psnode_dae_encoded_integrate_f32",
    "psnode_latent_backward_wide_supported", "psnode_latent_backward_wide_workspace_bytes", "psnode_latent_backward_wide_f32",
    "psnode_dae_backward_wide_supported", "psnode_dae_backward_wide_workspace_bytes", "psnode_dae_backward_wide_f32",
    "psnode_dae_backward_wide_ae_floats",
    "psnode_gemm_tn_supported", "psnode_gemm_tn_workspace_bytes", "psnode_gemm_tn_f32",
    "psnode_linear_rows_supported", "psnode_linear_rows_f32",
)


class PsnodeLibraryError(RuntimeError):
    pass


class PsnodeStatusError(RuntimeError):
    pass


class UnsupportedShapeError(ValueError):
    """PSNODE_ERR_UNSUPPORTED: no kernel covers this shape (e.g. an MLP too wide for the generic kernel's LDS budget)."""


class MlpF32(ctypes.Structure):
    _fields_ = [("n_layers", c_int32), ("in_dim", c_int32), ("out_dim", c_int32 * MAX_LAYERS),
                ("weight", c_void_p * MAX_LAYERS), ("bias", c_void_p * MAX_LAYERS)]


class ViewF32(ctypes.Structure):
    _fields_ = [("ptr", c_void_p), ("stride_t", c_int64), ("stride_b", c_int64)]


class OdeArgsF32(ctypes.Structure):
    _fields_ = [("method", c_int32), ("kernel", c_int32), ("flags", c_uint32), ("x_dim", c_int32), ("z_dim", c_int32),
                ("T", c_int64), ("B", c_int64), ("de", MlpF32), ("t", ViewF32), ("x", ViewF32), ("z", ViewF32),
                ("all_initial", c_void_p), ("event_idx", c_void_p), ("z_jump", c_void_p),
                ("zj_stride_b", c_int64), ("zj_stride_e", c_int64), ("x_out", c_void_p),
                ("save_act", c_void_p), ("save_xstage", c_void_p)]


class DaeArgsF32(ctypes.Structure):
    _fields_ = [("method", c_int32), ("kernel", c_int32), ("flags", c_uint32),
                ("x_dim", c_int32), ("z_dim", c_int32), ("v_dim", c_int32), ("i_dim", c_int32),
                ("T", c_int64), ("B", c_int64), ("de", MlpF32), ("ae", MlpF32),
                ("t", ViewF32), ("x", ViewF32), ("z", ViewF32), ("v", ViewF32), ("i", ViewF32),
                ("x_init", c_void_p), ("all_initial", c_void_p), ("event_idx", c_void_p),
                ("z_jump", c_void_p), ("zj_stride_b", c_int64), ("zj_stride_e", c_int64),
                ("v_jump", c_void_p), ("vj_stride_b", c_int64), ("vj_stride_e", c_int64),
                ("x_out", c_void_p), ("i_out", c_void_p),
                ("save_act", c_void_p), ("save_xstage", c_void_p), ("save_ae_act", c_void_p), ("save_ev_act", c_void_p),
                ("save_ev_i", c_void_p)]


class OdeBwdArgsF32(ctypes.Structure):
    _fields_ = [("method", c_int32), ("kernel", c_int32), ("x_dim", c_int32), ("z_dim", c_int32), ("T", c_int64), ("B", c_int64),
                ("de", MlpF32), ("t", ViewF32), ("z", ViewF32), ("all_initial", c_void_p), ("event_idx", c_void_p), ("z_jump", c_void_p),
                ("zj_stride_b", c_int64), ("zj_stride_e", c_int64), ("n_events", c_int32), ("xs", c_void_p), ("grad_xs", c_void_p),
                ("grad_x0", c_void_p), ("grad_z", c_void_p), ("grad_z_jump", c_void_p), ("grad_all_initial", c_void_p),
                ("grad_params", c_void_p), ("saved_act", c_void_p), ("saved_xstage", c_void_p), ("flags", ctypes.c_uint32)]


class DaeBwdArgsF32(ctypes.Structure):
    _fields_ = [("method", c_int32), ("kernel", c_int32), ("x_dim", c_int32), ("z_dim", c_int32), ("v_dim", c_int32), ("i_dim", c_int32),
                ("T", c_int64), ("B", c_int64), ("de", MlpF32), ("ae", MlpF32), ("t", ViewF32), ("z", ViewF32), ("v", ViewF32),
                ("all_initial", c_void_p), ("event_idx", c_void_p),
                ("z_jump", c_void_p), ("zj_stride_b", c_int64), ("zj_stride_e", c_int64),
                ("v_jump", c_void_p), ("vj_stride_b", c_int64), ("vj_stride_e", c_int64), ("n_events", c_int32),
                ("xs", c_void_p), ("is_", c_void_p), ("grad_xs", c_void_p), ("grad_is", c_void_p),
                ("grad_x_init", c_void_p), ("grad_z", c_void_p), ("grad_v", c_void_p), ("grad_z_jump", c_void_p),
                ("grad_v_jump", c_void_p), ("grad_all_initial", c_void_p), ("grad_params_de", c_void_p), ("grad_params_ae", c_void_p),
                ("saved_act", c_void_p), ("saved_xstage", c_void_p), ("saved_ae_act", c_void_p), ("saved_ev_act", c_void_p),
                ("saved_ev_i", c_void_p)]


class OdeEncodedArgsF32(ctypes.Structure):
    _fields_ = [("method", c_int32), ("x_dim", c_int32), ("z_dim", c_int32), ("T", c_int64), ("B", c_int64),
                ("x_encoder", MlpF32), ("z_encoder", MlpF32), ("x_decoder", MlpF32), ("de", MlpF32),
                ("t", ViewF32), ("x", ViewF32), ("z", ViewF32), ("event_idx", c_void_p), ("z_jump", c_void_p),
                ("zj_stride_b", c_int64), ("zj_stride_e", c_int64), ("x_pred", c_void_p), ("x_re", c_void_p),
                ("xre_stride_t", c_int64), ("xre_stride_b", c_int64), ("xh_out", c_void_p)]


class DaeEncodedArgsF32(ctypes.Structure):
    _fields_ = [("method", c_int32), ("x_dim", c_int32), ("z_dim", c_int32), ("v_dim", c_int32), ("i_dim", c_int32),
                ("T", c_int64), ("B", c_int64),
                ("x_encoder", MlpF32), ("z_encoder", MlpF32), ("v_encoder", MlpF32), ("i_encoder", MlpF32),
                ("x_decoder", MlpF32), ("i_decoder", MlpF32), ("de", MlpF32), ("ae", MlpF32),
                ("t", ViewF32), ("x", ViewF32), ("z", ViewF32), ("v", ViewF32), ("i", ViewF32), ("x0", c_void_p),
                ("event_idx", c_void_p), ("z_jump", c_void_p), ("zj_stride_b", c_int64), ("zj_stride_e", c_int64),
                ("v_jump", c_void_p), ("vj_stride_b", c_int64), ("vj_stride_e", c_int64),
                ("x_pred", c_void_p), ("i_pred", c_void_p),
                ("x_re", c_void_p), ("xre_stride_t", c_int64), ("xre_stride_b", c_int64),
                ("i_re", c_void_p), ("ire_stride_t", c_int64), ("ire_stride_b", c_int64)]


class LatentBwdWideArgsF32(ctypes.Structure):
    _fields_ = [("method", c_int32), ("hidden", c_int32), ("z_dim", c_int32), ("dae", c_int32), ("T", c_int64), ("B", c_int64),
                ("de", MlpF32), ("ae", MlpF32), ("t", ViewF32), ("event_idx", c_void_p), ("grad_xs", c_void_p), ("grad_is", c_void_p),
                ("saved_act", c_void_p), ("saved_ae_act", c_void_p), ("saved_ev_act", c_void_p),
                ("gk", c_void_p), ("d1", c_void_p), ("d1s", c_void_p), ("gi", c_void_p), ("da1", c_void_p), ("gi_ev", c_void_p),
                ("da1_ev", c_void_p), ("grad_x0", c_void_p)]


class DaeBwdWideArgsF32(ctypes.Structure):
    _fields_ = [("method", c_int32), ("x_dim", c_int32), ("z_dim", c_int32), ("v_dim", c_int32), ("i_dim", c_int32),
                ("T", c_int64), ("B", c_int64), ("de", MlpF32), ("ae", MlpF32),
                ("t", ViewF32), ("z", ViewF32), ("v", ViewF32), ("all_initial", c_void_p), ("event_idx", c_void_p),
                ("z_jump", c_void_p), ("zj_stride_b", c_int64), ("zj_stride_e", c_int64),
                ("v_jump", c_void_p), ("vj_stride_b", c_int64), ("vj_stride_e", c_int64), ("n_events", c_int32),
                ("xs", c_void_p), ("is_", c_void_p), ("grad_xs", c_void_p), ("grad_is", c_void_p),
                ("carry_x", c_void_p),
                ("ae_act", c_void_p * 3), ("ae_delta", c_void_p * 3), ("ae_gi", c_void_p),
                ("ev_act", c_void_p * 3), ("ev_delta", c_void_p * 3), ("ev_gi", c_void_p), ("ev_i", c_void_p),
                ("grad_params_de", c_void_p), ("grad_zv", c_void_p), ("grad_jump", c_void_p), ("grad_all_initial_de", c_void_p),
                ("saved_act", c_void_p), ("saved_xstage", c_void_p), ("saved_ae_act", c_void_p), ("saved_ev_act", c_void_p),
                ("saved_ev_i", c_void_p), ("flags", ctypes.c_uint32), ("x_true", c_void_p), ("i_true", c_void_p),
                ("grad_params_ae_raw", c_void_p)]


class DaeHeadGradsArgsF32(ctypes.Structure):
    _fields_ = [("R", c_int64), ("B", c_int64), ("hidden", c_int32), ("n_zv", c_int32), ("act", c_void_p * 3), ("act_row_stride", c_int64),
                ("delta", c_void_p * 3), ("gi", c_void_p), ("u", c_void_p), ("aw1", c_void_p), ("aw1_cols", c_int32), ("zv_col0", c_int32),
                ("grad_zv", c_void_p), ("sa1", c_void_p), ("out", c_void_p)]


class GemmTnArgsF32(ctypes.Structure):
    _fields_ = [("rows", c_int64), ("M", c_int32), ("N", c_int32), ("A", c_void_p), ("lda", c_int64), ("B", c_void_p), ("ldb", c_int64),
                ("C", c_void_p), ("colsum_a", c_void_p)]


class LinearRowsArgsF32(ctypes.Structure):
    _fields_ = [("rows", c_int64), ("K", c_int32), ("N", c_int32), ("X", c_void_p), ("ldx", c_int64), ("W", c_void_p),
                ("w_stride_n", c_int64), ("w_stride_k", c_int64), ("bias", c_void_p), ("epi", c_int32), ("Hh", c_void_p), ("ldh", c_int64),
                ("Y", c_void_p), ("ldy", c_int64)]


class LossArgsF32(ctypes.Structure):
    _fields_ = [("T", c_int64), ("B", c_int64), ("D", c_int32), ("mask_width", c_int32),
                ("pred", ViewF32), ("target", ViewF32), ("mask", ViewF32),
                ("col_weight", c_void_p), ("inv_norm", c_void_p), ("scale", ctypes.c_float), ("t0_coef", ctypes.c_float),
                ("out", c_void_p), ("grad_pred", c_void_p)]


_lib = None


def load():
    """Load libpsnode_hip.so once.  `import torch` must come first so that the HIP runtime torch bundles
    (libamdhip64.so.7) is the one instance both sides share."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (ordering requirement above)
    if not os.path.exists(LIB_PATH):
        raise PsnodeLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C py_psnode_amd/csrc`.  There is no non-HIP fallback for the fused integrator.")
    try:
        lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    except OSError as e:
        raise PsnodeLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    try:
        lib.psnode_abi_version.restype = c_int32
        got = lib.psnode_abi_version()
    except AttributeError as e:
        raise PsnodeLibraryError(f"{LIB_PATH} is not a psnode library (no psnode_abi_version): rebuild it with `make -C py_psnode_amd/csrc`") from e
    if got != ABI_VERSION:       # checked BEFORE the symbol lookups: a stale build fails here with a rebuild hint, not with a bare AttributeError
        raise PsnodeLibraryError(f"ABI version mismatch: {LIB_PATH} is version {got}, this binding is version {ABI_VERSION} -- "
                                 "rebuild it with `make -C py_psnode_amd/csrc` (or `python -c 'import __graft_entry__ as g; g.build()'`)")
    missing = [n for n in EXPORTS if not hasattr(lib, n)]
    if missing:
        raise PsnodeLibraryError(f"{LIB_PATH} lacks {missing}: stale build, rebuild it with `make -C py_psnode_amd/csrc`")
    lib.psnode_build_info.restype = c_char_p
    lib.psnode_ode_save_hidden.restype = c_int32
    lib.psnode_ode_save_hidden.argtypes = [ctypes.POINTER(OdeArgsF32)]
    lib.psnode_dae_head_grads_out_floats.restype = c_int32
    lib.psnode_dae_head_grads_out_floats.argtypes = [c_int32]
    lib.psnode_dae_head_grads_workspace_bytes.restype = c_size_t
    lib.psnode_dae_head_grads_workspace_bytes.argtypes = [ctypes.POINTER(DaeHeadGradsArgsF32)]
    lib.psnode_dae_head_grads_f32.restype = c_int32
    lib.psnode_dae_head_grads_f32.argtypes = [ctypes.POINTER(DaeHeadGradsArgsF32), c_void_p, c_size_t, c_void_p]
    lib.psnode_dae_save_hidden.restype = c_int32
    lib.psnode_dae_save_hidden.argtypes = [ctypes.POINTER(DaeArgsF32)]
    lib.psnode_status_string.restype = c_char_p
    lib.psnode_status_string.argtypes = [c_int32]
    lib.psnode_workspace_bytes.restype = c_size_t
    lib.psnode_workspace_bytes.argtypes = [ctypes.POINTER(MlpF32), ctypes.POINTER(MlpF32)]
    lib.psnode_event_table_f32.restype = c_int32
    lib.psnode_event_table_f32.argtypes = [c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p]
    lib.psnode_ode_integrate_f32.restype = c_int32
    lib.psnode_ode_integrate_f32.argtypes = [ctypes.POINTER(OdeArgsF32), c_void_p, c_size_t, c_void_p]
    lib.psnode_dae_integrate_f32.restype = c_int32
    lib.psnode_dae_integrate_f32.argtypes = [ctypes.POINTER(DaeArgsF32), c_void_p, c_size_t, c_void_p]
    lib.psnode_ode_kernel_for.restype = c_int32
    lib.psnode_ode_kernel_for.argtypes = [ctypes.POINTER(OdeArgsF32)]
    lib.psnode_dae_kernel_for.restype = c_int32
    lib.psnode_dae_kernel_for.argtypes = [ctypes.POINTER(DaeArgsF32)]
    lib.psnode_mlp_rows_supported.restype = c_int32
    lib.psnode_mlp_rows_supported.argtypes = [ctypes.POINTER(MlpF32)]
    lib.psnode_mlp_rows_f32.restype = c_int32
    lib.psnode_mlp_rows_f32.argtypes = [ctypes.POINTER(MlpF32), c_int64, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p]
    lib.psnode_ode_backward_supported.restype = c_int32
    lib.psnode_ode_backward_supported.argtypes = [ctypes.POINTER(OdeBwdArgsF32)]
    lib.psnode_ode_backward_param_count.restype = c_int64
    lib.psnode_ode_backward_param_count.argtypes = [ctypes.POINTER(OdeBwdArgsF32)]
    lib.psnode_ode_backward_workspace_bytes.restype = c_size_t
    lib.psnode_ode_backward_workspace_bytes.argtypes = [ctypes.POINTER(OdeBwdArgsF32)]
    lib.psnode_ode_backward_f32.restype = c_int32
    lib.psnode_ode_backward_f32.argtypes = [ctypes.POINTER(OdeBwdArgsF32), c_void_p, c_size_t, c_void_p]
    lib.psnode_dae_backward_supported.restype = c_int32
    lib.psnode_dae_backward_supported.argtypes = [ctypes.POINTER(DaeBwdArgsF32)]
    lib.psnode_dae_backward_workspace_bytes.restype = c_size_t
    lib.psnode_dae_backward_workspace_bytes.argtypes = [ctypes.POINTER(DaeBwdArgsF32)]
    lib.psnode_dae_backward_f32.restype = c_int32
    lib.psnode_dae_backward_f32.argtypes = [ctypes.POINTER(DaeBwdArgsF32), c_void_p, c_size_t, c_void_p]
    lib.psnode_mlp_rows_backward_workspace_bytes.restype = c_size_t
    lib.psnode_mlp_rows_backward_workspace_bytes.argtypes = [ctypes.POINTER(MlpF32), c_int64]
    lib.psnode_mlp_rows_backward_f32.restype = c_int32
    lib.psnode_mlp_rows_backward_f32.argtypes = [ctypes.POINTER(MlpF32), c_int64, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64,
                                                 c_void_p, c_void_p, c_size_t, c_void_p]
    lib.psnode_mlp_rows_backward_parts.restype = c_int64
    lib.psnode_mlp_rows_backward_parts.argtypes = [ctypes.POINTER(MlpF32), c_int64]
    lib.psnode_mlp_rows_reduce_workspace_bytes.restype = c_size_t
    lib.psnode_mlp_rows_reduce_workspace_bytes.argtypes = [ctypes.POINTER(MlpF32), c_int64]
    lib.psnode_mlp_rows_reduce_f32.restype = c_int32
    lib.psnode_mlp_rows_reduce_f32.argtypes = [ctypes.POINTER(MlpF32), c_int64, c_void_p, c_size_t, c_void_p, c_void_p]
    P = ctypes.POINTER(MlpF32)
    lib.psnode_recon_rows_supported.restype = c_int32
    lib.psnode_recon_rows_supported.argtypes = [P, P]
    lib.psnode_recon_rows_f32.restype = c_int32
    lib.psnode_recon_rows_f32.argtypes = [P, P, c_int64, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p]
    lib.psnode_recon_rows_param_count.restype = c_int64
    lib.psnode_recon_rows_param_count.argtypes = [P, P]
    lib.psnode_recon_rows_backward_workspace_bytes.restype = c_size_t
    lib.psnode_recon_rows_backward_workspace_bytes.argtypes = [P, P, c_int64]
    lib.psnode_recon_rows_backward_f32.restype = c_int32
    lib.psnode_recon_rows_backward_f32.argtypes = [P, P, c_int64, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_size_t,
                                                   c_void_p]
    lib.psnode_masked_mse_workspace_bytes.restype = c_size_t
    lib.psnode_masked_mse_workspace_bytes.argtypes = [ctypes.POINTER(LossArgsF32)]
    lib.psnode_masked_mse_f32.restype = c_int32
    lib.psnode_masked_mse_f32.argtypes = [ctypes.POINTER(LossArgsF32), c_void_p, c_size_t, c_void_p]
    lib.psnode_ode_encoded_supported.restype = c_int32
    lib.psnode_ode_encoded_supported.argtypes = [ctypes.POINTER(OdeEncodedArgsF32)]
    lib.psnode_ode_encoded_integrate_f32.restype = c_int32
    lib.psnode_ode_encoded_integrate_f32.argtypes = [ctypes.POINTER(OdeEncodedArgsF32), c_void_p]
    lib.psnode_latent_backward_wide_supported.restype = c_int32
    lib.psnode_latent_backward_wide_supported.argtypes = [c_int32, c_int32, c_int32]
    lib.psnode_latent_backward_wide_workspace_bytes.restype = c_size_t
    lib.psnode_latent_backward_wide_workspace_bytes.argtypes = [c_int32]
    lib.psnode_latent_backward_wide_f32.restype = c_int32
    lib.psnode_latent_backward_wide_f32.argtypes = [ctypes.POINTER(LatentBwdWideArgsF32), c_void_p, c_size_t, c_void_p]
    lib.psnode_dae_encoded_supported.restype = c_int32
    lib.psnode_dae_encoded_supported.argtypes = [ctypes.POINTER(DaeEncodedArgsF32)]
    lib.psnode_dae_encoded_workspace_bytes.restype = c_size_t
    lib.psnode_dae_encoded_workspace_bytes.argtypes = [ctypes.POINTER(DaeEncodedArgsF32)]
    lib.psnode_dae_encoded_integrate_f32.restype = c_int32
    lib.psnode_dae_encoded_integrate_f32.argtypes = [ctypes.POINTER(DaeEncodedArgsF32), c_void_p, c_size_t, c_void_p]
    lib.psnode_dae_backward_wide_supported.restype = c_int32
    lib.psnode_dae_backward_wide_supported.argtypes = [ctypes.POINTER(DaeBwdWideArgsF32)]
    lib.psnode_dae_backward_wide_workspace_bytes.restype = c_size_t
    lib.psnode_dae_backward_wide_workspace_bytes.argtypes = [ctypes.POINTER(DaeBwdWideArgsF32)]
    lib.psnode_dae_backward_wide_f32.restype = c_int32
    lib.psnode_dae_backward_wide_f32.argtypes = [ctypes.POINTER(DaeBwdWideArgsF32), c_void_p, c_size_t, c_void_p]
    lib.psnode_dae_backward_wide_ae_floats.restype = c_size_t
    lib.psnode_dae_backward_wide_ae_floats.argtypes = [ctypes.POINTER(DaeBwdWideArgsF32)]
    lib.psnode_linear_rows_supported.restype = c_int32
    lib.psnode_linear_rows_supported.argtypes = [ctypes.POINTER(LinearRowsArgsF32)]
    lib.psnode_linear_rows_f32.restype = c_int32
    lib.psnode_linear_rows_f32.argtypes = [ctypes.POINTER(LinearRowsArgsF32), c_void_p]
    lib.psnode_gemm_tn_supported.restype = c_int32
    lib.psnode_gemm_tn_supported.argtypes = [ctypes.POINTER(GemmTnArgsF32)]
    lib.psnode_gemm_tn_workspace_bytes.restype = c_size_t
    lib.psnode_gemm_tn_workspace_bytes.argtypes = [ctypes.POINTER(GemmTnArgsF32)]
    lib.psnode_gemm_tn_f32.restype = c_int32
    lib.psnode_gemm_tn_f32.argtypes = [ctypes.POINTER(GemmTnArgsF32), c_void_p, c_size_t, c_void_p]
    _lib = lib
    return lib


def check(status, what):
    if status != OK:
        msg = load().psnode_status_string(status).decode()
        if status == -5:
            raise UnsupportedShapeError(f"{what}: {msg} (status {status})")
        if status in (-2, -3):
            raise ValueError(f"{what}: {msg} (status {status})")
        raise PsnodeStatusError(f"{what}: {msg} (status {status})")
