// C ABI of the backward passes (include/psnode_hip.h: psnode_ode_backward_*, psnode_dae_backward_*): argument validation and the
// dispatch among the backward kernel families --
//   K4x (psnode_backward_x.hip)          ODE, hidden 33..64 with saved rows, up to one wave per SIMD: one wave = 4 trajectories, no LDS (round 6)
//   K4f (psnode_backward_fused.hip)      ODE, in -> H -> H -> H -> x at hidden <= 128: one launch, saved-activation and recompute forms
//   K8f / K9 (psnode_latent_dpp.hip / psnode_latent64_bwd*.hip)   the latent integrators of the direct_encode models at hidden 16 / 64
//   K8 (psnode_latent_bwd.hip)           the latent DAE at hidden 16
//   K5 (psnode_generic_bwd.hip)          anything else that fits the LDS
// (the DAE's no_encode shapes go through psnode_dae_backward_wide_f32 -> K7f, psnode_dae_backward_fused.hip).
// Rounds 1-4 also carried K4 / K7, hidden-64 specialisations of the recompute form; K4f / K7f cover their shapes (round 5:
// profiles/scripts/variants/ keeps the sources).
#include <string.h>

#include "psnode_pack.h"

using namespace psnode;

namespace {
int generic_np(const psnode_mlp_f32& m) {
    int np = 0, k = m.in_dim;
    for (int l = 0; l < m.n_layers; ++l) { np += m.out_dim[l] * (k + 1); k = m.out_dim[l]; }
    return np;
}
bool ode_generic_ok(const psnode_ode_bwd_args_f32* a) {
    const psnode_mlp_f32& m = a->de;
    if (a->x_dim < 1 || a->z_dim < 0 || m.n_layers < 1 || m.n_layers > kMaxLayers) return false;
    if (m.in_dim != 3 * (a->x_dim + a->z_dim) || m.out_dim[m.n_layers - 1] != a->x_dim) return false;
    return generic_bwd_fits(&a->de, nullptr, a->x_dim, a->z_dim, 0, 0) != 0;
}
// K4f: every width <= 128 (z_dim up to 8), saved-activation and recompute forms
bool use_fused_bwd(const psnode_ode_bwd_args_f32* a) { return a->kernel != PSNODE_KERNEL_GENERIC && fused_bwd_shape_ok(a); }
// K8 needs 16-byte aligned rows; with pointers not yet known (dims-only queries) the shape decides
bool use_latent_bwd(const psnode_ode_bwd_args_f32* a) {
    return a->kernel != PSNODE_KERNEL_GENERIC && latent_bwd_shape_ok(a) && (!a->xs || latent_bwd_ptrs_ok(a));
}
bool use_latent64_bwd(const psnode_ode_bwd_args_f32* a) {   // K9: the only fused backward for this shape (K5's LDS budget is too small)
    return a->kernel != PSNODE_KERNEL_GENERIC && latent64_ode_bwd_shape_ok(a) && (!a->xs || latent64_ode_bwd_ptrs_ok(a));
}
}  // namespace

extern "C" int32_t psnode_ode_backward_supported(const psnode_ode_bwd_args_f32* a) {
    if (!a || a->method < PSNODE_EULER || a->method > PSNODE_RK4_38) return 0;
    if (a->kernel == PSNODE_KERNEL_MFMA_WIDE || a->kernel == PSNODE_KERNEL_MFMA_TILE) return fused_bwd_shape_ok(a);
    if (a->kernel == PSNODE_KERNEL_MFMA_WAVE) return bwd_x_shape_ok(a);       // K4x (needs the saved rows at launch)
    if (a->kernel == PSNODE_KERNEL_MFMA) return fused_bwd_shape_ok(a) || use_latent_bwd(a) || use_latent64_bwd(a);
    return use_fused_bwd(a) || use_latent_bwd(a) || use_latent64_bwd(a) || ode_generic_ok(a);
}

extern "C" int64_t psnode_ode_backward_param_count(const psnode_ode_bwd_args_f32* a) {
    return a && a->de.n_layers >= 1 && a->de.n_layers <= kMaxLayers ? generic_np(a->de) : 0;
}

extern "C" size_t psnode_ode_backward_workspace_bytes(const psnode_ode_bwd_args_f32* a) {
    if (!a || !psnode_ode_backward_supported(a)) return 0;
    size_t floats = ode_generic_ok(a) ? generic_bwd_workspace_floats(&a->de, nullptr, a->B) : 0;
    if (latent_bwd_shape_ok(a)) floats = latent_bwd_workspace_floats(a->B) > floats ? latent_bwd_workspace_floats(a->B) : floats;
    if (latent64_ode_bwd_shape_ok(a)) floats = latent64_ode_bwd_workspace_floats(a->B) > floats ? latent64_ode_bwd_workspace_floats(a->B) : floats;
    if (fused_bwd_shape_ok(a)) { const size_t f3 = fused_bwd_workspace_floats(a); floats = f3 > floats ? f3 : floats; }
    if (bwd_x_shape_ok(a)) { const size_t f4 = bwd_x_workspace_floats(a); floats = f4 > floats ? f4 : floats; }
    return floats * sizeof(float);
}

extern "C" int32_t psnode_ode_backward_f32(const psnode_ode_bwd_args_f32* a, void* workspace, size_t workspace_bytes, void* stream) {
    if (!a) return PSNODE_ERR_NULL;
    if (a->method < PSNODE_EULER || a->method > PSNODE_RK4_38) return PSNODE_ERR_METHOD;
    if (a->T < 1 || a->B < 1) return PSNODE_ERR_DIMS;
    if (!psnode_ode_backward_supported(a)) return PSNODE_ERR_UNSUPPORTED;
    for (int l = 0; l < a->de.n_layers; ++l) if (!a->de.weight[l] || !a->de.bias[l]) return PSNODE_ERR_NULL;
    if (!a->t.ptr || !a->all_initial || !a->xs || !a->grad_xs || !a->grad_x0 || !a->grad_all_initial || !a->grad_params) return PSNODE_ERR_NULL;
    if (a->z_dim > 0 && !a->z.ptr) return PSNODE_ERR_NULL;
    if (a->event_idx && a->z_dim > 0 && !a->z_jump) return PSNODE_ERR_NULL;
    if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 255u) || workspace_bytes < psnode_ode_backward_workspace_bytes(a))
        return PSNODE_ERR_WORKSPACE;
    if ((a->saved_act != nullptr) != (a->saved_xstage != nullptr)) return PSNODE_ERR_NULL;
    if (a->kernel == PSNODE_KERNEL_MFMA_WAVE && !bwd_x_preferred(a)) return PSNODE_ERR_UNSUPPORTED;   // K4x: saved rows, no teacher forcing
    if (a->kernel == PSNODE_KERNEL_MFMA_TILE && !fused_bwd_shape_ok(a)) return PSNODE_ERR_UNSUPPORTED;
    if (a->saved_act && !use_fused_bwd(a) && !use_latent64_bwd(a)) return PSNODE_ERR_UNSUPPORTED;      // only K4x, K4f and K9 read them
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (a->flags & ~PSNODE_FLAG_INPUT_TRUE_X) return PSNODE_ERR_UNSUPPORTED;
    if (a->flags & PSNODE_FLAG_INPUT_TRUE_X) {      // teacher-forced backward: K4f (recompute form) is the kernel that has it
        if (a->kernel == PSNODE_KERNEL_GENERIC || a->saved_act || !fused_bwd_shape_ok(a)) return PSNODE_ERR_UNSUPPORTED;
        return fused_bwd_launch(a, static_cast<float*>(workspace), s);
    }
    if (bwd_x_preferred(a)) return bwd_x_launch(a, static_cast<float*>(workspace), s);      // K4x: up to one wave per SIMD at hidden 33..64
    if (use_latent_bwd(a)) return latent_bwd_launch(a, static_cast<float*>(workspace), s);
    if (use_latent64_bwd(a)) return latent64_ode_bwd_launch(a, static_cast<float*>(workspace), s);
    if (use_fused_bwd(a)) return fused_bwd_launch(a, static_cast<float*>(workspace), s);
    if (a->kernel == PSNODE_KERNEL_MFMA) return PSNODE_ERR_UNSUPPORTED;   // latent shape, unaligned views
    return generic_backward_launch(a->method, a->x_dim, a->z_dim, 0, 0, a->T, a->B, &a->de, nullptr,
                                   ViewDev{a->t.ptr, a->t.stride_t, a->t.stride_b}, ViewDev{a->z.ptr, a->z.stride_t, a->z.stride_b},
                                   ViewDev{nullptr, 0, 0}, a->all_initial, a->event_idx, a->z_jump, a->zj_stride_b, a->zj_stride_e, nullptr,
                                   0, 0, a->n_events, a->xs, nullptr, a->grad_xs, nullptr, a->grad_x0, a->grad_z, nullptr, a->grad_z_jump,
                                   nullptr, a->grad_all_initial, a->grad_params, nullptr, static_cast<float*>(workspace), s);
}

namespace {
bool use_latent16_dae_bwd(const psnode_dae_bwd_args_f32* a) {
    return a->kernel != PSNODE_KERNEL_GENERIC && latent16_dae_bwd_shape_ok(a) && (!a->xs || latent16_dae_bwd_ptrs_ok(a));
}
bool use_latent64_dae_bwd(const psnode_dae_bwd_args_f32* a) {
    return a->kernel != PSNODE_KERNEL_GENERIC && latent64_dae_bwd_shape_ok(a) && (!a->xs || latent64_dae_bwd_ptrs_ok(a));
}
}  // namespace

extern "C" int32_t psnode_dae_backward_supported(const psnode_dae_bwd_args_f32* a) {
    if (!a || a->method < PSNODE_EULER || a->method > PSNODE_RK4_38) return 0;
    if (a->x_dim < 1 || a->z_dim < 0 || a->v_dim < 0 || a->i_dim < 1) return 0;
    if (a->kernel == PSNODE_KERNEL_MFMA) return use_latent64_dae_bwd(a) || use_latent16_dae_bwd(a);
    if (use_latent64_dae_bwd(a) || use_latent16_dae_bwd(a)) return 1;
    const int n = a->x_dim + a->z_dim + a->v_dim + a->i_dim;
    const psnode_mlp_f32 &d = a->de, &g = a->ae;
    if (d.n_layers < 1 || d.n_layers > kMaxLayers || g.n_layers < 1 || g.n_layers > kMaxLayers) return 0;
    if (d.in_dim != 3 * n || d.out_dim[d.n_layers - 1] != a->x_dim) return 0;
    if (g.in_dim != n + a->x_dim + a->z_dim + a->v_dim || g.out_dim[g.n_layers - 1] != a->i_dim) return 0;
    return generic_bwd_fits(&a->de, &a->ae, a->x_dim, a->z_dim, a->v_dim, a->i_dim);
}

extern "C" size_t psnode_dae_backward_workspace_bytes(const psnode_dae_bwd_args_f32* a) {
    if (!a || !psnode_dae_backward_supported(a)) return 0;
    if (latent64_dae_bwd_shape_ok(a) && a->kernel != PSNODE_KERNEL_GENERIC) {
        // sized for every kernel this launch can end up on: K9, or K5 when the pointers turn out unaligned and K5 fits the shape
        size_t f = latent64_dae_bwd_workspace_floats(a);
        if (generic_bwd_fits(&a->de, &a->ae, a->x_dim, a->z_dim, a->v_dim, a->i_dim)) {
            const size_t k5 = generic_bwd_workspace_floats(&a->de, &a->ae, a->B);
            f = k5 > f ? k5 : f;
        }
        return f * sizeof(float);
    }
    if (latent16_dae_bwd_shape_ok(a)) {      // K8 (DAE) or, for unaligned views / kernel = generic, K5: the larger of the two
        const size_t k8 = latent16_dae_bwd_workspace_floats(a), k5 = generic_bwd_workspace_floats(&a->de, &a->ae, a->B);
        return (k8 > k5 ? k8 : k5) * sizeof(float);
    }
    return generic_bwd_workspace_floats(&a->de, &a->ae, a->B) * sizeof(float);
}

extern "C" int32_t psnode_dae_backward_f32(const psnode_dae_bwd_args_f32* a, void* workspace, size_t workspace_bytes, void* stream) {
    if (!a) return PSNODE_ERR_NULL;
    if (a->method < PSNODE_EULER || a->method > PSNODE_RK4_38) return PSNODE_ERR_METHOD;
    if (a->T < 1 || a->B < 1) return PSNODE_ERR_DIMS;
    if (!psnode_dae_backward_supported(a)) return PSNODE_ERR_UNSUPPORTED;
    for (int l = 0; l < a->de.n_layers; ++l) if (!a->de.weight[l] || !a->de.bias[l]) return PSNODE_ERR_NULL;
    for (int l = 0; l < a->ae.n_layers; ++l) if (!a->ae.weight[l] || !a->ae.bias[l]) return PSNODE_ERR_NULL;
    if (!a->t.ptr || !a->all_initial || !a->xs || !a->is || !a->grad_xs || !a->grad_x_init || !a->grad_all_initial || !a->grad_params_de ||
        !a->grad_params_ae)
        return PSNODE_ERR_NULL;
    if ((a->z_dim > 0 && !a->z.ptr) || (a->v_dim > 0 && !a->v.ptr)) return PSNODE_ERR_NULL;
    if (a->event_idx && ((a->z_dim > 0 && !a->z_jump) || (a->v_dim > 0 && !a->v_jump))) return PSNODE_ERR_NULL;
    if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 255u) || workspace_bytes < psnode_dae_backward_workspace_bytes(a))
        return PSNODE_ERR_WORKSPACE;
    {
        const bool sv = a->saved_act != nullptr;
        if ((a->saved_xstage != nullptr) != sv || (a->saved_ae_act != nullptr) != sv) return PSNODE_ERR_NULL;
        if (sv && a->event_idx && (!a->saved_ev_act || !a->saved_ev_i)) return PSNODE_ERR_NULL;
        if (sv && !use_latent64_dae_bwd(a)) return PSNODE_ERR_UNSUPPORTED;      // only K9 reads them here
    }
    if (use_latent64_dae_bwd(a)) return latent64_dae_bwd_launch(a, static_cast<float*>(workspace), static_cast<hipStream_t>(stream));
    if (use_latent16_dae_bwd(a)) return latent16_dae_bwd_launch(a, static_cast<float*>(workspace), static_cast<hipStream_t>(stream));
    if (a->kernel == PSNODE_KERNEL_MFMA) return PSNODE_ERR_UNSUPPORTED;
    return generic_backward_launch(a->method, a->x_dim, a->z_dim, a->v_dim, a->i_dim, a->T, a->B, &a->de, &a->ae,
                                   ViewDev{a->t.ptr, a->t.stride_t, a->t.stride_b}, ViewDev{a->z.ptr, a->z.stride_t, a->z.stride_b},
                                   ViewDev{a->v.ptr, a->v.stride_t, a->v.stride_b}, a->all_initial, a->event_idx, a->z_jump, a->zj_stride_b,
                                   a->zj_stride_e, a->v_jump, a->vj_stride_b, a->vj_stride_e, a->n_events, a->xs, a->is, a->grad_xs, a->grad_is,
                                   a->grad_x_init, a->grad_z, a->grad_v, a->grad_z_jump, a->grad_v_jump, a->grad_all_initial,
                                   a->grad_params_de, a->grad_params_ae, static_cast<float*>(workspace), static_cast<hipStream_t>(stream));
}
