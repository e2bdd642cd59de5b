/*
 * psnode_hip.h -- C ABI of the MI355X (gfx950) fixed-grid neural-ODE/DAE integrator.
 *
 * This is the drop-in boundary for the hot path of xxh0523/Py_PSNODE.  The reference has no FFI of
 * its own (it is pure Python/PyTorch); each entry point below replaces the
 * Python interface named in its comment (file:line under /root/reference), and is what a ctypes
 * binding inside the reference's `neural_dae` package would bind (INTEGRATION.md shows that stub).
 *
 * Conventions
 *   - plain C: pointers, sizes, POD structs.  No torch types.  All float data is fp32.
 *   - every pointer is a DEVICE pointer unless the comment says "host".
 *   - all work is enqueued asynchronously on `stream` (a hipStream_t passed as void*); no call
 *     synchronises the device, allocates user-visible memory or keeps a reference to its arguments
 *     beyond the enqueue.  Re-entrant across streams/devices; no global mutable state.
 *   - scratch memory is caller-provided (`workspace`); its size comes from psnode_workspace_bytes().
 *   - return value: PSNODE_OK (0) or a negative psnode_status; never throws/aborts.
 *     psnode_status_string() maps a status to text.
 *   - tensors are TIME-MAJOR logical [T,B,D] views described by (ptr, stride_t, stride_b) in ELEMENTS
 *     with the last dimension contiguous -- exactly what the scripts hand the solver:
 *     `x.permute(1,0,2)` of B-major [B,T,D] memory (neural_00_ODE_01_no_encode.py:82-84).
 */
#ifndef PSNODE_HIP_H
#define PSNODE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSNODE_ABI_VERSION 10
#define PSNODE_MAX_LAYERS 8      /* Linear layers per MLP */
#define PSNODE_MAX_WIDTH 1024    /* widest layer OUTPUT the kernels accept */
#define PSNODE_MAX_IN_WIDTH 2048 /* widest first-layer INPUT (the latent DE of DAE_02 at --hidden 128 is 12 x 128 = 1536 wide) */

typedef enum {
    PSNODE_OK = 0,
    PSNODE_ERR_NULL = -1,        /* required pointer is NULL */
    PSNODE_ERR_DIMS = -2,        /* bad / inconsistent dimensions (e.g. first Linear in_features != recipe) */
    PSNODE_ERR_METHOD = -3,      /* unknown method id */
    PSNODE_ERR_WORKSPACE = -4,   /* workspace too small or misaligned */
    PSNODE_ERR_UNSUPPORTED = -5, /* shape outside kernel limits, or the requested kernel is not available for it */
    PSNODE_ERR_HIP = -6          /* a HIP runtime call failed (launch error, bad stream, ...) */
} psnode_status;

/* my_fixed_grid.py:12 (Euler), :20 (Midpoint), :35 (RK4 = 3/8-rule rk4_alt_step_func) */
typedef enum { PSNODE_EULER = 0, PSNODE_MIDPOINT = 1, PSNODE_RK4_38 = 2 } psnode_method;

/* kernel selection: AUTO picks the MFMA kernel when the shape has one, else the generic kernel */
typedef enum {
    PSNODE_KERNEL_AUTO = 0, PSNODE_KERNEL_GENERIC = 1, PSNODE_KERNEL_MFMA = 2,
    PSNODE_KERNEL_MFMA_TILE = 4,     /* K1 / K2, the 4-waves-per-16-trajectory-tile MFMA integrators, where AUTO / MFMA would pick K1x / K2x; psnode_ode_backward_f32: K4f */
    PSNODE_KERNEL_MFMA_WAVE = 5,     /* K1x / K2x, the one-wave-per-4-trajectories (exchange-free) MFMA integrators (incl. their saving instances);
                                        psnode_ode_backward_f32: K4x, the backward of the same form (hidden 33..64, saved rows); UNSUPPORTED outside their shapes */
    PSNODE_KERNEL_MFMA_WIDE = 3      /* backward calls only: the one-launch MFMA backward K4f (hidden <= 128 zero-padded to 32 / 64 / 128); since ABI 8
                                        (K4 removed) the same kernel AUTO / MFMA pick for these shapes */
} psnode_kernel;

/* flags: teacher forcing of my_solvers.py:52 (input_true_x) and :82 (input_true_x, input_true_i) */
#define PSNODE_FLAG_INPUT_TRUE_X 1u
#define PSNODE_FLAG_INPUT_TRUE_I 2u

/* An nn.Sequential(Linear, ELU, Linear, ..., Linear) exactly as nn.Linear stores it:
 * weight[l] row-major [out_dim[l], in] (in = in_dim for l = 0, else out_dim[l-1]), bias[l] [out_dim[l]].
 * ELU(alpha=1) follows every layer but the last.
 * Replaces the forward of DE_Func.x_dot / AE_Func.i_calculator
 * (neural_00_ODE_01_no_encode.py:61-68, neural_01_DAE_01_no_encode.py:64-83). */
typedef struct {
    int32_t n_layers;
    int32_t in_dim;
    int32_t out_dim[PSNODE_MAX_LAYERS];
    const float* weight[PSNODE_MAX_LAYERS];
    const float* bias[PSNODE_MAX_LAYERS];
} psnode_mlp_f32;

/* strided time-major view [T,B,D], element strides, last dim contiguous */
typedef struct {
    const float* ptr;
    int64_t stride_t;
    int64_t stride_b;
} psnode_view_f32;

/* Arguments of FixedGridODESolver.integrate_ODE (my_solvers.py:52-80) with a recognised DE_Func
 * (neural_00_ODE_01_no_encode.py:58-68; latent variant neural_00_ODE_02_direct_encode.py:49-57).
 *
 *   xs[0] = x[0];  for k in 0..T-2:  dt = t[k+1]-t[k];
 *     zk  = event_idx[k] >= 0 ? z_jump[:, event_idx[k]] : z[k]
 *     src = INPUT_TRUE_X ? x[k] : xs[k];   xs[k+1] = src + step(method, f(. ; zk), dt, src)
 *   f(x; z) = MLP_de( cat(a0, cat(x,z) - a0, cat(x,z)) ),  a0 = all_initial
 */
typedef struct {
    int32_t method;            /* psnode_method */
    int32_t kernel;            /* psnode_kernel */
    uint32_t flags;            /* PSNODE_FLAG_INPUT_TRUE_X */
    int32_t x_dim, z_dim;
    int64_t T, B;
    psnode_mlp_f32 de;         /* in_dim must equal 3*(x_dim+z_dim), last out_dim must equal x_dim */
    psnode_view_f32 t;         /* [T,B,1] */
    psnode_view_f32 x;         /* [T,B,x_dim]; only x[0] is read unless INPUT_TRUE_X */
    psnode_view_f32 z;         /* [T,B,z_dim] */
    const float* all_initial;  /* [B, x_dim+z_dim] contiguous */
    const int32_t* event_idx;  /* int32[T-1], -1 = no event at that step; NULL = no events */
    const float* z_jump;       /* [B,nE,z_dim] : z_jump[b*zj_stride_b + e*zj_stride_e + d] */
    int64_t zj_stride_b, zj_stride_e;
    float* x_out;              /* [T,B,x_dim] contiguous (my_solvers.py:62) */
    /* Training forward (ABI 3, optional, both NULL = off): what autograd would save for loss.backward() -- the hidden activations of the
     * three ELU layers per (step, stage, trajectory) and the stage inputs -- written by the integrator itself so that the backward
     * (psnode_ode_backward_f32 with saved_act / saved_xstage) does not recompute the stage evaluations (a third of its MFMA work):
     *   save_act    [T-1, S, 3, B, Hp]   Hp = hidden rounded up to 32 / 64 / 128 (psnode_ode_save_hidden), S = stages of the method
     *   save_xstage [T-1, S, B, x_dim]
     * 6 KB per state-step at hidden 128: sized for the 288 GB of an MI355X, meant for the widths where the recompute is what bounds the
     * training step.  Only the MFMA integrator K1 writes them (psnode_ode_save_hidden() > 0, no teacher forcing): else UNSUPPORTED. */
    float* save_act;
    float* save_xstage;
} psnode_ode_args_f32;

/* Arguments of FixedGridODESolver.integrate_DAE (my_solvers.py:82-131) with recognised DE_Func/AE_Func
 * (neural_01_DAE_01_no_encode.py:61-83; latent variant neural_01_DAE_02_direct_encode.py:70-100).
 *
 *   x0 = x_init; i0 = g(TRUE_X ? x[0] : x0; z[0], v[0]);  xs[0]=x0; is[0]=i0
 *   for k: [event: zk,vk <- jumps; i0 = g(x0; zk, vk)]
 *          x1 = step(f(. ; zk, vk, TRUE_I ? i[k] : i0), start = TRUE_X ? x[k] : x0)
 *          i1 = g(TRUE_X ? x[k+1] : x1; z[k+1], v[k+1]);  xs[k+1]=x1; is[k+1]=i1; x0=x1; i0=i1
 *   f = MLP_de(cat(a0, s-a0, s)), s = cat(x,z,v,i);   g = MLP_ae(cat(a0, x, z, v))
 */
typedef struct {
    int32_t method, kernel;
    uint32_t flags;            /* PSNODE_FLAG_INPUT_TRUE_X | PSNODE_FLAG_INPUT_TRUE_I */
    int32_t x_dim, z_dim, v_dim, i_dim;
    int64_t T, B;
    psnode_mlp_f32 de;         /* in_dim = 3*(x+z+v+i), out = x_dim */
    psnode_mlp_f32 ae;         /* in_dim = (x+z+v+i) + (x+z+v), out = i_dim */
    psnode_view_f32 t, x, z, v, i;   /* x, i only read under teacher forcing (may be NULL otherwise) */
    const float* x_init;       /* [B,x_dim] contiguous */
    const float* all_initial;  /* [B, x+z+v+i] contiguous */
    const int32_t* event_idx;  /* as above */
    const float* z_jump;       /* [B,nE,z_dim] */
    int64_t zj_stride_b, zj_stride_e;
    const float* v_jump;       /* [B,nE,v_dim] */
    int64_t vj_stride_b, vj_stride_e;
    float* x_out;              /* [T,B,x_dim] contiguous */
    float* i_out;              /* [T,B,i_dim] contiguous */
    /* Optional training side outputs, all of them or none (as psnode_ode_args_f32::save_act): what loss.backward() would have autograd
     * keep, so that the backward call (psnode_dae_bwd_wide_args_f32::saved_*) does not recompute the forward:
     *   save_act    [T-1, S, L, B, Hp]   the DE's ELU outputs per (step, stage); Hp = psnode_dae_save_hidden(), L = hidden layers of the MLPs
     *   save_xstage [T-1, S, B, x_dim]   the DE's stage inputs
     *   save_ae_act [L, T, B, Hp]        the AE head's ELU outputs per grid point (my_solvers.py:95, :121)
     *   save_ev_act [nE, L, B, Hp]       the same for the event-time heads i0 = g(x_k; jumps) (my_solvers.py:108-110); rows of events no
     *   save_ev_i   [nE, B, W]           step takes are not written.  save_ev_i: i0 -- K2 shapes (L = 3): W = 16, the slot layout of
     *                                    psnode_dae_bwd_wide_args_f32; latent shapes at hidden 64 (L = 1, K3c): W = i_dim = 64.
     * The two event buffers are required only with event_idx.  Only the MFMA integrators K2 and K3c write them, without teacher forcing
     * (psnode_dae_save_hidden() > 0): else UNSUPPORTED. */
    float* save_act;
    float* save_xstage;
    float* save_ae_act;
    float* save_ev_act;
    float* save_ev_i;
} psnode_dae_args_f32;

/* ABI / build identification. */
int32_t psnode_abi_version(void);
const char* psnode_build_info(void);               /* e.g. "psnode_hip gfx950 ..." (static storage) */
const char* psnode_status_string(int32_t status);  /* static storage */

/* Scratch bytes needed by an integrate call with these MLPs (ae may be NULL).  Host-only, no HIP calls. */
size_t psnode_workspace_bytes(const psnode_mlp_f32* de, const psnode_mlp_f32* ae);

/* Replaces ODE_Event.event_fn + jump_change_fn's index search (neural_base.py:52-62, 178-196), resolved
 * once per call on the device instead of one host sync per step:
 *   event_idx[k] = e such that event_times[e*stride_e] == clock[k*stride_k] (exact fp32 equality), else -1
 * for k in [0, n_steps).  `clock` is trajectory 0's time column, `event_times` trajectory 0's event list.
 * If several e match, *dup_flag (optional int32, device) is set to 1 and the first match is used
 * (the reference raises on that input). */
int32_t psnode_event_table_f32(int64_t n_steps, const float* clock, int64_t stride_k,
                               const float* event_times, int64_t stride_e, int32_t n_events,
                               int32_t* event_idx, int32_t* dup_flag, void* stream);

/* Replaces FixedGridODESolver.integrate_ODE (my_solvers.py:52-80) + Euler/Midpoint/RK4._step_func
 * (my_fixed_grid.py:12-59) + DE_Func.forward for the whole batch and all T-1 steps. */
int32_t psnode_ode_integrate_f32(const psnode_ode_args_f32* args, void* workspace, size_t workspace_bytes, void* stream);

/* Replaces FixedGridODESolver.integrate_DAE (my_solvers.py:82-131) + step functions + DE_Func/AE_Func forwards. */
int32_t psnode_dae_integrate_f32(const psnode_dae_args_f32* args, void* workspace, size_t workspace_bytes, void* stream);

/* Row-wise 2-layer ELU-MLP  out[r,:] = W2.ELU(W1.in[r,:] + b1) + b2  over `rows` rows (row strides in elements).
 * Replaces the forward of the direct_encode encoders/decoders nn.Sequential(Linear, ELU, Linear) applied to every
 * (b,t) row: x_encoder / z_encoder / x_decoder (neural_00_ODE_02_direct_encode.py:64-69,74-88) and
 * v_encoder / i_encoder / i_decoder (neural_01_DAE_02_direct_encode.py:107-118,126-152).
 * Supported: hidden width 16 (the scripts' default hidden_dim) or 64 (their debug override), in/out width up to the
 * hidden width -- see psnode_mlp_rows_supported. */
int32_t psnode_mlp_rows_supported(const psnode_mlp_f32* mlp);
/* Input row r sits at in + r * in_row_stride (in_inner_rows == 0), or -- two-level -- at
 *   in + (r / in_inner_rows) * in_outer_stride + (r % in_inner_rows) * in_row_stride      (rows, in_inner_rows < 2^32):
 * the DataLoader's [B,T,D] batch read as the time-major rows r = t * B + b the latent tensors are laid out in (in_inner_rows = B,
 * in_row_stride = T * D, in_outer_stride = D) -- the x.permute(1, 0, 2) of neural_00_ODE_02_direct_encode.py:76 without a copy. */
int32_t psnode_mlp_rows_f32(const psnode_mlp_f32* mlp, int64_t rows, const float* in, int64_t in_row_stride, int64_t in_inner_rows,
                            int64_t in_outer_stride, float* out, int64_t out_row_stride, void* stream);

/* Backward of psnode_mlp_rows_f32: grad_in[r,:] (optional) and the parameter gradients as ONE flat vector in nn.Linear order
 * [W1 (H x in), b1 (H), W2 (out x H), b2 (out)], from the saved input rows and grad_out.  What loss.backward() does for the
 * encoders/decoders of the direct_encode models (neural_00_ODE_02_direct_encode.py:267-275 through :74-88).
 * Deterministic (per-wave partials in `workspace`, summed in a fixed order).  grad_params == NULL: the partials are left unreduced in
 * `workspace` (psnode_mlp_rows_backward_parts(mlp, rows) * param_count floats suffice) for psnode_mlp_rows_reduce_f32 below. */
size_t psnode_mlp_rows_backward_workspace_bytes(const psnode_mlp_f32* mlp, int64_t rows);
int32_t psnode_mlp_rows_backward_f32(const psnode_mlp_f32* mlp, int64_t rows, const float* in, int64_t in_row_stride,
                                     int64_t in_inner_rows, int64_t in_outer_stride, const float* grad_out, int64_t gout_row_stride, float* grad_in, int64_t gin_row_stride,
                                     float* grad_params, void* workspace, size_t workspace_bytes, void* stream);

/* A module applied to SEVERAL row sets in one step (x_encoder over the grid rows and the first row, z_encoder over grid rows, first row and
 * jump rows: neural_00_ODE_02_direct_encode.py:76-82; the decoder over the solution and the reconstruction, :86-88): call
 * psnode_mlp_rows_backward_f32 once per set with grad_params == NULL and `workspace` = that set's slice of ONE partials buffer
 * (psnode_mlp_rows_backward_parts(mlp, rows) vectors of the module's parameter count each, sets side by side), then
 * psnode_mlp_rows_reduce_f32 over all n_parts vectors: the module's gradient, summed in a fixed order (deterministic).  Replaces one
 * two-launch reduction per set and autograd's `add` per parameter tensor and extra use.  (ABI 9) */
int64_t psnode_mlp_rows_backward_parts(const psnode_mlp_f32* mlp, int64_t rows);
size_t psnode_mlp_rows_reduce_workspace_bytes(const psnode_mlp_f32* mlp, int64_t n_parts);   /* (n_parts + 32) * param_count floats */
int32_t psnode_mlp_rows_reduce_f32(const psnode_mlp_f32* mlp, int64_t n_parts, void* workspace, size_t workspace_bytes, float* grad_params,
                                   void* stream);

/* K10 (ABI 10): tall-skinny contraction over rows,  C[m][n] = sum_r A[r][m] * B[r][n]  (+ colsum_a[m] = sum_r A[r][m], optional), on MFMA
 * with the rows as the contraction index: the weight gradients that are not accumulated inside a sweep kernel -- K9w's blocks over its
 * stored rows at the direct_encode models' hidden widths other than 16 / 64 (the scripts' argparse default --hidden 128:
 * neural_00_ODE_02_direct_encode.py:160-162, what loss.backward() forms for de_func there) -- instead of a library GEMM.
 * M, N <= 128 and multiples of 4; lda / ldb (elements) multiples of 4; A, B 16-byte aligned; C row-major [M, N].  Deterministic
 * (per-workgroup partials in `workspace`, summed in a fixed order). */
typedef struct {
    int64_t rows;
    int32_t M, N;
    const float* A;                  /* [rows, M], row stride lda */
    int64_t lda;
    const float* B;                  /* [rows, N], row stride ldb */
    int64_t ldb;
    float* C;                        /* [M, N] */
    float* colsum_a;                 /* [M] or NULL */
} psnode_gemm_tn_args_f32;
int32_t psnode_gemm_tn_supported(const psnode_gemm_tn_args_f32* args);
size_t psnode_gemm_tn_workspace_bytes(const psnode_gemm_tn_args_f32* args);
int32_t psnode_gemm_tn_f32(const psnode_gemm_tn_args_f32* args, void* workspace, size_t workspace_bytes, void* stream);

/* K11 (ABI 10): one linear layer over rows with a fused epilogue,  Y[r][n] = epi(sum_k X[r][k] * Wm[n][k] + bias[n]),  K, N <= 128:
 * the forward AND backward of the direct_encode encoders / decoders (nn.Sequential(Linear, ELU, Linear) over every (b, t) row,
 * neural_00_ODE_02_direct_encode.py:64-69, 74-88) at the hidden widths psnode_mlp_rows_f32 does not carry (the scripts' argparse default
 * --hidden 128), and the row-wise products of the latent-wide backward.  Wm[n][k] = W[n * w_stride_n + k * w_stride_k]: an nn.Linear
 * weight [N, K] as it is (w_stride_n = K, w_stride_k = 1) or read transposed (a [K, N] tensor: w_stride_n = 1, w_stride_k = N).
 * epi: 0 identity, 1 ELU(alpha = 1), 2 multiply by ELU'(Hh[r][n]) where Hh holds ELU OUTPUTS (the delta of a hidden layer).
 * bias may be NULL.  Row strides in elements; no workspace. */
typedef struct {
    int64_t rows;
    int32_t K, N;
    const float* X;                  /* [rows, K], row stride ldx */
    int64_t ldx;
    const float* W;
    int64_t w_stride_n, w_stride_k;
    const float* bias;               /* [N] or NULL */
    int32_t epi;
    const float* Hh;                 /* epi == 2: [rows, N], row stride ldh */
    int64_t ldh;
    float* Y;                        /* [rows, N], row stride ldy */
    int64_t ldy;
} psnode_linear_rows_args_f32;
int32_t psnode_linear_rows_supported(const psnode_linear_rows_args_f32* args);
int32_t psnode_linear_rows_f32(const psnode_linear_rows_args_f32* args, void* stream);

/* The RECONSTRUCTION branch of the direct_encode ODE model at hidden 16 as ONE row kernel each way (ABI 9):
 *   x_re = x_decoder(x_encoder(x))        neural_00_ODE_02_direct_encode.py:87     (x_encoder in <= 16 -> 16 -> 16, x_decoder 16 -> 16 -> out <= 16)
 * and its share of loss.backward() (:267-275).  The encoded rows never reach memory: the forward reads x and writes x_re, the backward reads x
 * and dL/dx_re and returns the parameter gradients of BOTH modules as one flat vector
 *   [W1e (16 x in), b1e (16), W2e (16 x 16), b2e (16) | W1d (16 x 16), b1d (16), W2d (out x 16), b2d (out)]        (nn.Linear order)
 * (deterministic: per-wave partials in `workspace`, summed in a fixed order).  Input rows are addressed as in psnode_mlp_rows_f32. */
int32_t psnode_recon_rows_supported(const psnode_mlp_f32* encoder, const psnode_mlp_f32* decoder);
int32_t psnode_recon_rows_f32(const psnode_mlp_f32* encoder, const psnode_mlp_f32* decoder, int64_t rows, const float* in, int64_t in_row_stride,
                              int64_t in_inner_rows, int64_t in_outer_stride, float* out, int64_t out_row_stride, void* stream);
int64_t psnode_recon_rows_param_count(const psnode_mlp_f32* encoder, const psnode_mlp_f32* decoder);
size_t psnode_recon_rows_backward_workspace_bytes(const psnode_mlp_f32* encoder, const psnode_mlp_f32* decoder, int64_t rows);
int32_t psnode_recon_rows_backward_f32(const psnode_mlp_f32* encoder, const psnode_mlp_f32* decoder, int64_t rows, const float* in,
                                       int64_t in_row_stride, int64_t in_inner_rows, int64_t in_outer_stride, const float* grad_out,
                                       int64_t gout_row_stride, float* grad_params, void* workspace, size_t workspace_bytes, void* stream);

/* Backward (discretise-then-optimise) pass through psnode_ode_integrate_f32: what loss.backward() computes when it
 * walks the unrolled T-step autograd graph of integrate_ODE (neural_00_ODE_01_no_encode.py:358-360 through
 * my_solvers.py:66-78), in one launch.  Inputs: the forward arguments, the forward result xs and dL/dxs.
 * Outputs: dL/dx[0], dL/dz (per grid point; steps that took a jump put their gradient into grad_z_jump instead),
 * dL/dall_initial and dL/d(parameters) as ONE flat vector in nn.Linear order
 * [W1 (64 x 3n), b1, W2, b2, W3, b3, W4 (x_dim x 64), b4] (psnode_ode_backward_param_count floats).
 * Kernels: MFMA backwards for the shape classes 3n -> h -> h -> h -> x_dim (h <= 128, x_dim <= 8, z_dim <= 8; K4x / K4f) and the latent
 * 6H -> H -> H with x_dim = z_dim = H in {16, 64} (16-byte aligned rows); the generic backward for any MLP whose activations
 * fit the 160 KB LDS (its parameter-gradient accumulators move to the workspace when they do not).  No teacher forcing;
 * t carries no gradient.  Deterministic (per-workgroup partials summed in a fixed order). */
typedef struct {
    int32_t method;
    int32_t kernel;                  /* psnode_kernel: AUTO = MFMA backward when the shape has one (K4x with saved rows at hidden 33..64 up to 4608
                                        trajectories, else K4f), else generic; _MFMA_WAVE / _MFMA_TILE / _MFMA_WIDE force K4x / K4f / K4f */
    int32_t x_dim, z_dim;
    int64_t T, B;
    psnode_mlp_f32 de;
    psnode_view_f32 t, z;
    const float* all_initial;        /* [B, x+z] contiguous */
    const int32_t* event_idx;        /* as in the forward call, or NULL */
    const float* z_jump;
    int64_t zj_stride_b, zj_stride_e;
    int32_t n_events;
    const float* xs;                 /* forward result [T,B,x_dim] contiguous */
    const float* grad_xs;            /* dL/dxs         [T,B,x_dim] contiguous */
    float* grad_x0;                  /* [B,x_dim] */
    float* grad_z;                   /* [T,B,z_dim] contiguous, or NULL */
    float* grad_z_jump;              /* [B,n_events,z_dim] contiguous, zero-initialised by the caller, or NULL */
    float* grad_all_initial;         /* [B, x+z] */
    float* grad_params;              /* flat, psnode_ode_backward_param_count() floats */
    const float* saved_act;          /* ABI 3, optional: what the forward call wrote to save_act / save_xstage (same method, T, B, MLP).  With them the */
    const float* saved_xstage;       /* one-launch backward K4f skips the recompute of the stage evaluations; both NULL = recompute */
    uint32_t flags;                  /* ABI 5: PSNODE_FLAG_INPUT_TRUE_X = backward of a teacher-forced call (my_solvers.py:72-74: every step starts
                                        from the dataset row x[k]): `xs` then holds the DATASET x [T,B,x_dim] contiguous (the forward result is not
                                        needed), no adjoint is carried from step to step, grad_x0 = dL/dxs[0] + the start adjoint of step 0.
                                        K4f only (hidden <= 128, x_dim <= 8, recompute form: saved_* must be NULL) */
} psnode_ode_bwd_args_f32;

int32_t psnode_ode_backward_supported(const psnode_ode_bwd_args_f32* args);
int64_t psnode_ode_backward_param_count(const psnode_ode_bwd_args_f32* args);
size_t psnode_ode_backward_workspace_bytes(const psnode_ode_bwd_args_f32* args);
int32_t psnode_ode_backward_f32(const psnode_ode_bwd_args_f32* args, void* workspace, size_t workspace_bytes, void* stream);

/* Backward pass through psnode_dae_integrate_f32 (no teacher forcing): loss.backward() through integrate_DAE
 * (neural_01_DAE_01_no_encode.py:422-424 over my_solvers.py:94-129), including the AE head, the feedback of the
 * algebraic variable into the DE input and the event-time recomputation i0 = g(x0; jumps).
 * grad_params_de / grad_params_ae: flat nn.Linear-order vectors (psnode_dae_backward_param_counts). */
typedef struct {
    int32_t method;
    int32_t kernel;                  /* PSNODE_KERNEL_AUTO: an MFMA backward when the shape has one (3n-64-64-64 DE + AE with
                                        x <= 8, z+v+i <= 8; the latent blocks-of-H shapes, H in {16, 64}), else the generic one */
    int32_t x_dim, z_dim, v_dim, i_dim;
    int64_t T, B;
    psnode_mlp_f32 de, ae;
    psnode_view_f32 t, z, v;
    const float* all_initial;        /* [B, x+z+v+i] */
    const int32_t* event_idx;
    const float* z_jump; int64_t zj_stride_b, zj_stride_e;
    const float* v_jump; int64_t vj_stride_b, vj_stride_e;
    int32_t n_events;
    const float* xs;                 /* forward results [T,B,x_dim], [T,B,i_dim] contiguous */
    const float* is;
    const float* grad_xs;            /* dL/dxs, dL/dis (grad_is may be NULL = zeros) */
    const float* grad_is;
    float* grad_x_init;              /* [B,x_dim] */
    float* grad_z;                   /* [T,B,z_dim] or NULL */
    float* grad_v;                   /* [T,B,v_dim] or NULL */
    float* grad_z_jump;              /* [B,n_events,z_dim], zero-initialised by the caller, or NULL */
    float* grad_v_jump;              /* [B,n_events,v_dim], zero-initialised by the caller, or NULL */
    float* grad_all_initial;         /* [B, x+z+v+i] */
    float* grad_params_de;
    float* grad_params_ae;
    /* ABI 4, optional, all or none (the two event buffers only with event_idx): what the forward call wrote to
     * psnode_dae_args_f32::save_* (same method, T, B, MLPs).  Read by K9 (the latent shapes at hidden 64), which then evaluates nothing
     * forwards; UNSUPPORTED for the shapes whose kernel behind this entry point recomputes (K7, K8, K5). */
    const float* saved_act;
    const float* saved_xstage;
    const float* saved_ae_act;
    const float* saved_ev_act;
    const float* saved_ev_i;
} psnode_dae_bwd_args_f32;

int32_t psnode_dae_backward_supported(const psnode_dae_bwd_args_f32* args);
size_t psnode_dae_backward_workspace_bytes(const psnode_dae_bwd_args_f32* args);
int32_t psnode_dae_backward_f32(const psnode_dae_bwd_args_f32* args, void* workspace, size_t workspace_bytes, void* stream);

/* The DAE backward at hidden <= 128 (K7f, csrc/psnode_dae_backward_fused.hip): loss.backward() through integrate_DAE
 * (my_solvers.py:94-129) in ONE launch over the whole grid -- per grid point the AE head's adjoint (its output adjoint = dL/dis of that
 * grid point + what the DE of the step starting there returned through its algebraic inputs), per step the DE stages, at event steps the
 * recompute i0 = g(x0; jumps) and its adjoint.  The DE's parameter gradients are accumulated in the kernel (grad_params_de: flat,
 * nn.Linear order W1,b1,..,W4,b4) and the DE's share of the input gradients is written in its final layout:
 *     grad_zv             [T, B, z+v]          dL/d(z | v) through the DE (zero at event steps and at grid point T-1); every entry written
 *     grad_jump           [B, n_events, z+v]   the same for the jump values of the events taken; zero-initialised by the caller
 *     grad_all_initial_de [B, x+z+v+i]         the DE's share of dL/dall_initial
 * carry_x [B,x_dim] (output) = dL/dx_init without dL/dxs[0].
 * "Slot" layout of an algebraic-variable row [.., 16]: slot q < 2(z+v+i) is the DE's external input q of the `s - a0` block (q < z+v+i) or of
 * the `s` block; the adjoint / value of i-dim d sits in slots z+v+d and (z+v+i)+z+v+d (values: equal; adjoints: to be summed by the caller).
 * Unless psnode_dae_backward_wide_ae_floats(args) > 0 (below) the AE head's rows are written for the caller to contract (K7h,
 * psnode_dae_head_grads_f32: AE parameter gradients, the AE's share of the input gradients), row r = grid point r:
 *     ae_act[l], ae_delta[l]    : [T, B, H]  AE head at grid point r (ELU outputs / pre-activation adjoints of hidden layer l = 1..3)
 *     ae_gi                     : [T, B, 16] adjoint of the head's output, slot layout
 *     ev_act[l], ev_delta[l]    : [n_events, B, H] the event-time recompute of event e (written for the events taken)
 *     ev_gi, ev_i               : [n_events, B, 16] its output adjoint and its value i0, slot layout
 * Shape class: de = 3n -> h -> h -> h -> x_dim, ae = n+x+z+v -> h -> h -> h -> i_dim (the same h <= 128), x_dim <= 8, z+v+i <= 8;
 * H = h rounded up to the kernels' 32 / 64 / 128 (columns h..H-1 of the rows are exact zeros: zero-padded units).
 * (ABI <= 7 also had a split form behind this entry point -- adjoint sweep in time chunks + library GEMMs on the host side -- and its
 * ODE counterpart psnode_ode_backward_wide_f32; removed in ABI 8, K4f / K7f cover their shapes.) */
typedef struct {
    int32_t method;
    int32_t x_dim, z_dim, v_dim, i_dim;
    int64_t T, B;
    psnode_mlp_f32 de, ae;
    psnode_view_f32 t, z, v;
    const float* all_initial;        /* [B, x+z+v+i] */
    const int32_t* event_idx;        /* int32[T-1] or NULL */
    const float* z_jump; int64_t zj_stride_b, zj_stride_e;
    const float* v_jump; int64_t vj_stride_b, vj_stride_e;
    int32_t n_events;
    const float* xs;                 /* forward results [T,B,x_dim], [T,B,i_dim] contiguous */
    const float* is;
    const float* grad_xs;            /* dL/dxs [T,B,x_dim] */
    const float* grad_is;            /* dL/dis [T,B,i_dim] or NULL (= zeros) */
    float* carry_x;                  /* [B,x_dim] out */
    float* ae_act[3];
    float* ae_delta[3];
    float* ae_gi;
    float* ev_act[3];
    float* ev_delta[3];
    float* ev_gi;
    float* ev_i;
    float* grad_params_de;           /* required */
    float* grad_zv;
    float* grad_jump;
    float* grad_all_initial_de;
    /* Optional, all of them or none: what the forward call wrote to psnode_dae_args_f32::save_* (same method, T, B,
     * MLPs; the two event buffers with event_idx).  The kernel then evaluates nothing forwards; ae_act / ev_act / ev_i are NOT written
     * (contract over saved_ae_act / saved_ev_act instead). */
    const float* saved_act;
    const float* saved_xstage;
    const float* saved_ae_act;
    const float* saved_ev_act;
    const float* saved_ev_i;
    /* ABI 5, recompute form only (saved_* NULL): backward of a teacher-forced integrate_DAE (my_solvers.py:111-121).
     * PSNODE_FLAG_INPUT_TRUE_X: the DE of step k starts from x_true[k] and the head at grid point j reads x_true[j] (dataset rows
     * [T,B,x_dim] contiguous) -- no adjoint flows from step to step through x except through an event's recomputed i0, whose head reads
     * the RUNNING state xs[k]; PSNODE_FLAG_INPUT_TRUE_I: the DE reads i_true[k] ([T,B,i_dim] contiguous) instead of the head's value --
     * the DE's algebraic adjoint is dropped (the AE -> DE link is cut).  Gradients w.r.t. the dataset rows themselves are not formed. */
    uint32_t flags;
    const float* x_true;
    const float* i_true;
    /* ABI 5, at hidden <= 64 WITH saved activations (psnode_dae_backward_wide_ae_floats(args) > 0; then REQUIRED): the AE head's gradients are
     * formed in the kernel as well -- nothing is left to contract, no head row is written (ae_act / ae_delta / ae_gi / ev_delta / ev_gi may
     * be NULL; ev_act / ev_i are still scratch of the recompute form).  Flat output, h = the MLPs' hidden width, K1a = n + x + z + v:
     *     [ dAW1 (h x K1a) | db1 (h) | dAW2 (h x h) | db2 | dAW3 (h x h) | db3 | P3 (16 x h) | sg (16) ]
     * P3 / sg are per ext SLOT (slot q < ne: the `s - a0` block, ne <= q < 2 ne: the `s` block; an algebraic variable d owns slots
     * nzv + d and ne + nzv + d): dAW4[d] = P3[nzv + d] + P3[ne + nzv + d], db4[d] likewise from sg.  grad_all_initial_de then holds the WHOLE
     * dL/dall_initial and grad_zv / grad_jump the whole dL/d(z|v) (DE + head). */
    float* grad_params_ae_raw;
} psnode_dae_bwd_wide_args_f32;

int32_t psnode_dae_backward_wide_supported(const psnode_dae_bwd_wide_args_f32* args);   /* dims only */
size_t psnode_dae_backward_wide_ae_floats(const psnode_dae_bwd_wide_args_f32* args);     /* floats of grad_params_ae_raw; 0 = head rows + K7h */
size_t psnode_dae_backward_wide_workspace_bytes(const psnode_dae_bwd_wide_args_f32* args);
int32_t psnode_dae_backward_wide_f32(const psnode_dae_bwd_wide_args_f32* args, void* workspace, size_t workspace_bytes, void* stream);

/* K7h: every contraction over the AE head's rows that the fused-DE form of psnode_dae_backward_wide_f32 leaves to the caller, in one
 * launch (no library GEMM): for rows r < R (grid points, or events) and trajectories b, with h_l = act[l] row r (the head's ELU outputs:
 * saved by the forward call or written by the backward call) and delta_l its layer adjoints,
 *     out = [ dAW2 (h x h) | dAW3 (h x h) | P3 (16 x h): gi[slot] (x) h3 | P0 (h x 16): delta1 (x) u | db1 db2 db3 (3 x h) | sum gi (16) ]
 *     sa1[b] = sum_r delta1[r, b]                      (for dL/dall_initial and dAW1's all_initial columns)
 *     grad_zv[r, b, c] = sum_unit AW1[unit][zv_col0 + c] delta1[r, b][unit],  c < n_zv   (row stride 8)
 * with h = hidden (the MLP's real width; rows are Hp = 32 / 64 / 128 floats wide, zero padded), gi / u rows of 16 floats (gi: slot layout
 * of psnode_dae_bwd_wide_args_f32; u: the head's first-layer input columns behind all_initial, x | z | v, zero padded).  Deterministic
 * (per-workgroup partials summed in a fixed order). */
typedef struct {
    int64_t R, B;
    int32_t hidden;
    int32_t n_zv;
    const float* act[3];             /* row r of layer l: act[l] + r * act_row_stride, [B, Hp] */
    int64_t act_row_stride;
    const float* delta[3];           /* [R, B, Hp] */
    const float* gi;                 /* [R, B, 16] */
    const float* u;                  /* [R, B, 16] */
    const float* aw1;                /* the AE's first nn.Linear weight [hidden, aw1_cols] (read for grad_zv only) */
    int32_t aw1_cols, zv_col0;
    float* grad_zv;                  /* [R, B, 8] or NULL */
    float* sa1;                      /* [B, Hp] */
    float* out;                      /* [psnode_dae_head_grads_out_floats(hidden)] */
} psnode_dae_head_grads_args_f32;

int32_t psnode_dae_head_grads_out_floats(int32_t hidden);
size_t psnode_dae_head_grads_workspace_bytes(const psnode_dae_head_grads_args_f32* args);
int32_t psnode_dae_head_grads_f32(const psnode_dae_head_grads_args_f32* args, void* workspace, size_t workspace_bytes, void* stream);

/* Masked, column-weighted squared-error loss of one (prediction, target) pair and its gradient, in one pass:
 *
 *   out[d]   = scale * inv_norm * col_weight[d] * sum_{t,b} mask[t,b,(d)] * (pred[t,b,d] - target[t,b,d])^2     d < D
 *   out[D]   = t0_coef * sum_{b,d} (pred[0,b,d] - target[0,b,d])^2
 *   out[D+1] = sum_d out[d] + out[D]                                                  (the scalar the scripts call backward on)
 *   grad_pred[t,b,d] = d out[D+1] / d pred[t,b,d]                                     (contiguous [T,B,D], optional)
 *
 * This is the step right after the path in the training loops:
 *   `torch.sum(torch.sum(Loss_func(x_pred, x, reduction='none') * mask, dim=1), dim=0) / torch.sum(mask)` summed over d
 *   (neural_00_ODE_01_no_encode.py:353-355) -> mask = the batch's mask, inv_norm = 1/sum(mask), scale = 1;
 *   the DAE's weighted form `(sum(se*mask) + 9*sum(se[:,:,1:2]*mask)) / sum(mask)` and `Loss_func(x[:,0,:], x_pred[:,0,:])`
 *   (neural_01_DAE_01_no_encode.py:414-419) -> col_weight = [1,10,1,...], t0_coef = 1/(B*D);
 *   the direct_encode reconstruction term `Loss_func(x_re, x)` (neural_00_ODE_02_direct_encode.py:269) -> no mask,
 *   scale = 1/(T*B*D).
 * pred/target/mask are logical [T,B,*] views (element strides, last dim contiguous): pred is normally the integrator's
 * time-major output, target/mask the scripts' B-major tensors.  mask_width: 0 = no mask, 1 = [T,B,1], D = [T,B,D].
 * inv_norm is a DEVICE scalar (so that sum(mask) -- or its all-reduce over ranks -- never has to visit the host);
 * NULL = 1.  Deterministic: per-workgroup partial sums are combined in a fixed order (in double). */
typedef struct {
    int64_t T, B;
    int32_t D;
    int32_t mask_width;
    psnode_view_f32 pred, target, mask;
    const float* col_weight;         /* device [D] or NULL (= ones) */
    const float* inv_norm;           /* device scalar or NULL (= 1) */
    float scale;
    float t0_coef;
    float* out;                      /* device [D+2] */
    float* grad_pred;                /* device [T,B,D] contiguous, or NULL */
} psnode_loss_args_f32;

size_t psnode_masked_mse_workspace_bytes(const psnode_loss_args_f32* args);
int32_t psnode_masked_mse_f32(const psnode_loss_args_f32* args, void* workspace, size_t workspace_bytes, void* stream);

/* The WHOLE forward of the direct_encode ODE model in one launch -- replaces ODE_Model.forward of
 * neural_00_ODE_02_direct_encode.py:74-89 (hidden_dim = 16, the script's default):
 *
 *   Xh = x_encoder(x); Zh = z_encoder(z); a0 = cat(Xh[0], Zh[0]); Zh_jump = z_encoder(z_jump)
 *   Xh_sol = integrate_ODE(de, t, Xh, Zh, a0, events on Zh_jump)                (my_solvers.py:52-80)
 *   x_pred = x_decoder(Xh_sol);  x_re = x_decoder(Xh)
 *
 * t, x, z are the scripts' RAW tensors as time-major views (permute(1,0,2) of [B,T,*]); z_jump is the RAW [B,nE,z_dim] tensor
 * (encoded in the kernel at the steps that take a jump); event_idx as produced by psnode_event_table_f32.
 * Every MLP is Linear(in,16) ELU Linear(16,out): x_encoder x_dim->16->16, z_encoder z_dim->16->16, x_decoder 16->16->x_dim,
 * de 96->16->16 (x_dim, z_dim <= 16).  Xh, Zh and Xh_sol never reach memory (108 B of HBM traffic per state-step instead of ~490);
 * xh_out (optional) receives the latent trajectory [T,B,16] for callers that want it.  No workspace. */
typedef struct {
    int32_t method;                  /* psnode_method */
    int32_t x_dim, z_dim;
    int64_t T, B;
    psnode_mlp_f32 x_encoder, z_encoder, x_decoder, de;
    psnode_view_f32 t, x, z;         /* [T,B,1], [T,B,x_dim], [T,B,z_dim] */
    const int32_t* event_idx;        /* int32[T-1] or NULL */
    const float* z_jump;             /* raw [B,nE,z_dim] */
    int64_t zj_stride_b, zj_stride_e;
    float* x_pred;                   /* [T,B,x_dim] contiguous */
    float* x_re;                     /* x_re[t*xre_stride_t + b*xre_stride_b + d], or NULL to skip the reconstruction */
    int64_t xre_stride_t, xre_stride_b;
    float* xh_out;                   /* [T,B,16] contiguous or NULL */
} psnode_ode_encoded_args_f32;

int32_t psnode_ode_encoded_supported(const psnode_ode_encoded_args_f32* args);   /* 1 / 0, dims only */
int32_t psnode_ode_encoded_integrate_f32(const psnode_ode_encoded_args_f32* args, void* stream);

/* The whole DAE_Model.forward of neural_01_DAE_02_direct_encode.py:125-153 at its shipped hidden_dim 64 in ONE launch (K3g; ABI 6):
 *   Xh0 = x_encoder(x0)                                   x0 = Init_Func(z[0], v[0], i[0]), computed by the caller (once per batch)
 *   Zh = z_encoder(z) (absent when z_dim == 0), Vh = v_encoder(v), Xh = x_encoder(x), Ih = i_encoder(i)
 *   all_initial = cat(Xh0, Zh[0], Vh[0], Ih[0]);  jumps = z_encoder(z_jump) | v_encoder(v_jump)
 *   (Xh_sol, Ih_sol) = integrate_DAE(x_init = Xh0, latent DE 12H|9H -> H -> H, latent AE 7H|5H -> H -> H)      (my_solvers.py:82-131)
 *   x_pred = x_decoder(Xh_sol), x_pred[0] = x0;  i_pred = i_decoder(Ih_sol);  x_re = x_decoder(Xh);  i_re = i_decoder(Ih)
 * t, x, z, v, i are the scripts' RAW tensors as time-major views; z_jump / v_jump the RAW [B,nE,*] tensors; event_idx as produced by
 * psnode_event_table_f32.  Every encoder is Linear(in,64) ELU Linear(64,64) (x_dim <= 16; z_dim, v_dim, i_dim <= 8), every decoder
 * Linear(64,64) ELU Linear(64,out).  None of the six latent [T,B,64] tensors reaches memory.  x_re / i_re: both or neither (NULL: the
 * reconstruction and the x / i encoders' per-row work are skipped; x may then be NULL too).  Workspace: the packed weight images. */
typedef struct {
    int32_t method;                  /* psnode_method */
    int32_t x_dim, z_dim, v_dim, i_dim;
    int64_t T, B;
    psnode_mlp_f32 x_encoder, z_encoder, v_encoder, i_encoder, x_decoder, i_decoder, de, ae;
    psnode_view_f32 t, x, z, v, i;   /* [T,B,1], [T,B,x_dim], [T,B,z_dim], [T,B,v_dim], [T,B,i_dim] */
    const float* x0;                 /* [B, x_dim] contiguous */
    const int32_t* event_idx;        /* int32[T-1] or NULL */
    const float* z_jump;             /* raw [B,nE,z_dim] */
    int64_t zj_stride_b, zj_stride_e;
    const float* v_jump;             /* raw [B,nE,v_dim] */
    int64_t vj_stride_b, vj_stride_e;
    float* x_pred;                   /* [T,B,x_dim] contiguous */
    float* i_pred;                   /* [T,B,i_dim] contiguous */
    float* x_re;                     /* x_re[t*xre_stride_t + b*xre_stride_b + d], or NULL */
    int64_t xre_stride_t, xre_stride_b;
    float* i_re;                     /* i_re[t*ire_stride_t + b*ire_stride_b + d], or NULL */
    int64_t ire_stride_t, ire_stride_b;
} psnode_dae_encoded_args_f32;

int32_t psnode_dae_encoded_supported(const psnode_dae_encoded_args_f32* args);   /* 1 / 0, dims only */
size_t psnode_dae_encoded_workspace_bytes(const psnode_dae_encoded_args_f32* args);
int32_t psnode_dae_encoded_integrate_f32(const psnode_dae_encoded_args_f32* args, void* workspace, size_t workspace_bytes, void* stream);

/* Adjoint sweep through the latent integrators of the direct_encode models at the hidden widths K9 / K8 do not take (every
 * hidden_dim <= 128 with hidden_dim % 4 == 0 other than 16 / 64 -- the scripts' argparse default --hidden 128; K9w, ABI 7): the
 * sequential half of loss.backward() through integrate_ODE / integrate_DAE (my_solvers.py:66-78, 94-129).  Reads what the forward call
 * saved (psnode_*_integrate_f32 with save_act ...: [T-1,S,B,H] hidden activations, the AE head's [T,B,H] and event [nE,B,H] rows) and
 * writes the rows every parameter / input gradient is a plain contraction over (library GEMMs on the caller's side,
 * py_psnode_amd/fused.py: latent_backward_wide):
 *   gk, d1 [T-1,S,B,H]  adjoint of each stage's RHS value / of its hidden pre-activation        d1s [T-1,B,H]  sum of d1 over the stages
 *   gi, da1 [T,B,H]     (DAE) adjoint of i_k / of the AE head's hidden pre-activation           gi_ev, da1_ev [nE,B,H]  of the event heads
 *   grad_x0 [B,H]       dL/dx_0 (x_init of the DAE, x[0] of the ODE)
 * Every array contiguous and 16-byte aligned; the event rows zero-initialised by the caller (events no step takes stay zero). */
typedef struct {
    int32_t method, hidden, z_dim, dae;      /* z_dim = hidden, or 0 for the DAE without z */
    int64_t T, B;
    psnode_mlp_f32 de, ae;                   /* the latent MLPs (ae unused for the ODE) */
    psnode_view_f32 t;
    const int32_t* event_idx;                /* int32[T-1] or NULL */
    const float *grad_xs, *grad_is;          /* [T,B,H]; grad_is NULL = zeros (or the ODE) */
    const float *saved_act, *saved_ae_act, *saved_ev_act;
    float *gk, *d1, *d1s, *gi, *da1, *gi_ev, *da1_ev, *grad_x0;
} psnode_latent_bwd_wide_args_f32;

int32_t psnode_latent_backward_wide_supported(int32_t hidden, int32_t z_dim, int32_t dae);
size_t psnode_latent_backward_wide_workspace_bytes(int32_t hidden);
int32_t psnode_latent_backward_wide_f32(const psnode_latent_bwd_wide_args_f32* args, void* workspace, size_t workspace_bytes, void* stream);

/* Row width Hp of save_act for these dims (32 / 64 / 128) if an AUTO / MFMA call can save its activations, else 0 (dims only). */
int32_t psnode_ode_save_hidden(const psnode_ode_args_f32* args);
int32_t psnode_dae_save_hidden(const psnode_dae_args_f32* args);

/* Which kernel an AUTO call with these dims (and batch size B) would run: PSNODE_KERNEL_GENERIC, PSNODE_KERNEL_MFMA or -- the
 * one-wave-per-4-trajectories integrators K1x / K2x, hidden <= 64 at up to one wave per SIMD -- PSNODE_KERNEL_MFMA_WAVE. */
int32_t psnode_ode_kernel_for(const psnode_ode_args_f32* args);
int32_t psnode_dae_kernel_for(const psnode_dae_args_f32* args);

#ifdef __cplusplus
}
#endif
#endif /* PSNODE_HIP_H */
