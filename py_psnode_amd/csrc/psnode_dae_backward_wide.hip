// C ABI front door of the DAE backward at hidden <= 128 (include/psnode_hip.h: psnode_dae_backward_wide_*): argument validation, then K7f
// (psnode_dae_backward_fused.hip) -- ONE launch over the whole grid that sweeps the adjoint through the DE stages, the AE head per grid
// point and the event-time recomputes and forms the DE's parameter gradients in the kernel (at hidden <= 64 with saved activations the
// head's too; otherwise the head's rows go to K7h, psnode_head_grads.hip).
// Rounds 2-4 also carried the split form behind this entry point (K7w: stored rows in time chunks + library GEMMs on the host side);
// profiles/scripts/variants/k7w_dae_backward_split.hip.txt keeps its source.
#include <string.h>

#include "psnode_pack.h"

using namespace psnode;

extern "C" {

int32_t psnode_dae_backward_wide_supported(const psnode_dae_bwd_wide_args_f32* a) {
    if (!a || a->method < PSNODE_EULER || a->method > PSNODE_RK4_38) return 0;
    const int xd = a->x_dim, zd = a->z_dim, vd = a->v_dim, id = a->i_dim;
    if (xd < 1 || xd > 4 * kNXc || zd < 0 || vd < 0 || id < 1) return 0;
    const int nzv = zd + vd, ne = nzv + id, n = xd + ne;
    if (ne > 8) return 0;
    const int NZM = (2 * ne + 3) / 4, NZA = (nzv + 3) / 4;
    if (NZA < 1 || NZA > 2 || NZM > kMaxNZM || (NZA == 2 && NZM < 3)) return 0;
    const int h = wide_hidden(a->de);
    if (!h || wide_hidden(a->ae) != h || a->ae.out_dim[0] != a->de.out_dim[0]) return 0;
    return a->de.in_dim == 3 * n && a->de.out_dim[3] == xd && a->ae.in_dim == n + xd + nzv && a->ae.out_dim[3] == id;
}

size_t psnode_dae_backward_wide_ae_floats(const psnode_dae_bwd_wide_args_f32* a) {
    if (!psnode_dae_backward_wide_supported(a)) return 0;
    return dae_fused_bwd_ae_floats(a);
}

size_t psnode_dae_backward_wide_workspace_bytes(const psnode_dae_bwd_wide_args_f32* a) {
    if (!psnode_dae_backward_wide_supported(a)) return 0;
    return dae_fused_bwd_workspace_floats(a) * sizeof(float);
}

int32_t psnode_dae_backward_wide_f32(const psnode_dae_bwd_wide_args_f32* p, void* workspace, size_t workspace_bytes, void* stream) {
    if (!p) return PSNODE_ERR_NULL;
    if (p->method < PSNODE_EULER || p->method > PSNODE_RK4_38) return PSNODE_ERR_METHOD;
    if (!psnode_dae_backward_wide_supported(p)) return PSNODE_ERR_UNSUPPORTED;
    if (p->T < 2 || p->B < 1) return PSNODE_ERR_DIMS;
    for (int l = 0; l < 4; ++l) if (!p->de.weight[l] || !p->de.bias[l] || !p->ae.weight[l] || !p->ae.bias[l]) return PSNODE_ERR_NULL;
    if (!p->grad_params_de) return PSNODE_ERR_NULL;
    if (p->flags & ~(PSNODE_FLAG_INPUT_TRUE_X | PSNODE_FLAG_INPUT_TRUE_I)) return PSNODE_ERR_UNSUPPORTED;
    const bool ae_in_kernel = dae_fused_bwd_ae_floats(p) > 0;      // hidden <= 64 with saved activations: no head rows are written at all
    if (!p->t.ptr || !p->all_initial || !p->xs || !p->is || !p->grad_xs || !p->carry_x || (!ae_in_kernel && !p->ae_gi)) return PSNODE_ERR_NULL;
    {
        const bool sv = p->saved_act != nullptr;
        if ((p->saved_xstage != nullptr) != sv || (p->saved_ae_act != nullptr) != sv) return PSNODE_ERR_NULL;
        if (sv && p->event_idx && (!p->saved_ev_act || !p->saved_ev_i)) return PSNODE_ERR_NULL;
        if (!p->grad_all_initial_de || (p->z_dim + p->v_dim > 0 && (!p->grad_zv || (p->event_idx && !p->grad_jump)))) return PSNODE_ERR_NULL;
    }
    for (int l = 0; l < 3; ++l) if (!ae_in_kernel && (!p->ae_act[l] || !p->ae_delta[l])) return PSNODE_ERR_NULL;
    if ((p->z_dim > 0 && !p->z.ptr) || (p->v_dim > 0 && !p->v.ptr)) return PSNODE_ERR_NULL;
    if (p->event_idx) {
        if (p->n_events < 1 || (p->z_dim > 0 && !p->z_jump) || (p->v_dim > 0 && !p->v_jump) || !p->ev_i) return PSNODE_ERR_NULL;
        if (!ae_in_kernel && !p->ev_gi) return PSNODE_ERR_NULL;
        for (int l = 0; l < 3; ++l) if (!p->ev_act[l] || (!ae_in_kernel && !p->ev_delta[l])) return PSNODE_ERR_NULL;
    }
    if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 255u) || workspace_bytes < psnode_dae_backward_wide_workspace_bytes(p))
        return PSNODE_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int H = wide_hidden(p->de), zd = p->z_dim, vd = p->v_dim;
    {   // per-lane offsets inside a row are 32-bit byte offsets next to a scalar row base
        const int64_t lim = (int64_t)1 << 30, Bm = p->B;
        const int64_t sb[] = {H, p->t.stride_b, zd > 0 ? p->z.stride_b : 0, vd > 0 ? p->v.stride_b : 0,
                              p->event_idx && zd > 0 ? p->zj_stride_b : 0, p->event_idx && vd > 0 ? p->vj_stride_b : 0};
        for (int64_t q : sb) if (q < 0 || Bm * q + 64 >= lim) return PSNODE_ERR_DIMS;
    }
    return dae_fused_bwd_launch(p, static_cast<float*>(workspace), s);
}

}  // extern "C"
