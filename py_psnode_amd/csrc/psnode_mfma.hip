// K1/K2 dispatch: which shapes run on the MFMA integrators (psnode_mfma_impl.h) and the hidden-64 instantiation.
#include "psnode_mfma_impl.h"

namespace psnode {
namespace {

// width class (32 / 64 / 128, forward also 192 / 256) the MFMA kernels run a 4-layer in -> H -> H -> H -> out MLP at: H itself or,
// zero-padded, the next one up (psnode_pack.h: PackMfma::hreal); 0 if H > 256 or the shape is another one
int mfma_hidden(const MlpDev& m, int in_dim, int out_dim) {
    if (m.n_layers != 4 || m.in_dim != in_dim || m.out_dim[3] != out_dim) return 0;
    const int h = m.out_dim[0];
    if (m.out_dim[1] != h || m.out_dim[2] != h) return 0;
    return padded_hidden_fwd(h);
}
// ... for the training forwards (saved activations): the widths a fused backward exists at
int mfma_hidden_saving(const MlpDev& m, int in_dim, int out_dim) {
    const int h = mfma_hidden(m, in_dim, out_dim);
    return h <= 128 ? h : 0;
}

}  // namespace

bool mfma_ode_supported(const IntegrateDev& a) {
    if (latent_shape_ok(a, false)) return a.a0 == nullptr || latent_ptrs_ok(a, false);   // a0 == NULL: dims-only query
    if (latent64_shape_ok(a, false)) return a.a0 == nullptr || latent64_ptrs_ok(a, false);
    if (latentw_shape_ok(a, false)) return a.a0 == nullptr || latentw_ptrs_ok(a, false);
    const int h = mfma_hidden(a.de, 3 * (a.xd + a.zd), a.xd);
    if (a.xd < 1 || a.xd > 4 * kNXw || !h) return false;
    if (!streamed_class_ok(h / 16, false, a.xd > 4 * kNXc, nzm_of(a, false))) return false;     // hidden 193..256: x_dim <= 8
    return nzm_of(a, false) <= kMaxNZM;       // z_dim <= 8
}

int mfma_ode_save_hidden(const IntegrateDev& a) {
    // K3c: one hidden layer per MLP.  With pointers present (a0 != NULL: a launch, not a dims-only query) the call only takes K3c when its
    // alignment requirements hold -- otherwise AUTO would fall back to K0, which writes no saved rows, and return PSNODE_OK all the same
    if (latent64_shape_ok(a, false)) return (a.a0 == nullptr || latent64_ptrs_ok(a, false)) ? 64 : 0;
    if (latent_shape_ok(a, false)) return 0;
    if (latentw_shape_ok(a, false)) return (a.a0 == nullptr || latentw_ptrs_ok(a, false)) ? a.xd : 0;     // K3w saves rows of the real width
    if ((a.flags & PSNODE_FLAG_INPUT_TRUE_X) || a.xd < 1 || a.xd > 4 * kNXc || nzm_of(a, false) > kMaxNZM) return 0;
    return mfma_hidden_saving(a.de, 3 * (a.xd + a.zd), a.xd);
}

bool mfma_dae_supported(const IntegrateDev& a) {
    if (latent_shape_ok(a, true)) return a.a0 == nullptr || latent_ptrs_ok(a, true);
    if (latent64_shape_ok(a, true)) return a.a0 == nullptr || latent64_ptrs_ok(a, true);
    if (latentw_shape_ok(a, true)) return a.a0 == nullptr || latentw_ptrs_ok(a, true);
    const int n = a.xd + a.zd + a.vd + a.id;
    if (a.xd < 1 || a.xd > 4 * kNXc || a.id < 1) return false;
    const int h = mfma_hidden(a.de, 3 * n, a.xd);
    if (!h || mfma_hidden(a.ae, n + a.xd + a.zd + a.vd, a.id) != h || a.ae.out_dim[0] != a.de.out_dim[0]) return false;   // DE and AE share --hidden
    const int NZM = nzm_of(a, true), NZA = nza_of(a);
    if (!streamed_class_ok(h / 16, true, false, NZM)) return false;       // DAE: hidden <= 192, and z+v+i <= 6 above 128
    switch (NZM * 10 + NZA) {
        case 11: case 21: case 31: case 41: case 32: case 42: return true;
        default: return false;
    }
}

int mfma_dae_save_hidden(const IntegrateDev& a) {
    if (latent64_shape_ok(a, true)) return (a.a0 == nullptr || latent64_ptrs_ok(a, true)) ? 64 : 0;
    if (latent_shape_ok(a, true)) return 0;
    if (latentw_shape_ok(a, true)) return (a.a0 == nullptr || latentw_ptrs_ok(a, true)) ? a.xd : 0;
    if ((a.flags & (PSNODE_FLAG_INPUT_TRUE_X | PSNODE_FLAG_INPUT_TRUE_I)) || !mfma_dae_supported(a)) return 0;
    return mfma_hidden_saving(a.de, 3 * (a.xd + a.zd + a.vd + a.id), a.xd);
}

size_t mfma_pack_floats(const psnode_mlp_f32* de, const psnode_mlp_f32* ae) {
    if (de && de->n_layers == 2) {
        const size_t m = latent_pack_floats() > latent64_pack_floats() ? latent_pack_floats() : latent64_pack_floats();
        return m > latentw_pack_floats() ? m : latentw_pack_floats();
    }
    if (!de || de->n_layers != 4) return 0;
    const int n = de->in_dim / 3, nw = (padded_hidden_fwd(de->out_dim[0]) ? padded_hidden_fwd(de->out_dim[0]) : de->out_dim[0] + 15) / 16;
    const size_t one = (size_t)nw * (max_regs(nw) + (n + 3) / 4) * 64 + stream_image_floats(nw);
    const size_t both = ae ? 2 * one : one;
    const size_t wave = ae ? mfma_xd_pack_floats() : mfma_x_pack_floats();
    return both > wave ? both : wave;
}

hipError_t launch_mfma(const IntegrateDev& a, bool dae, float* pack, hipStream_t stream) {
    if (latent_shape_ok(a, dae)) return launch_latent(a, dae, pack, stream);
    if (latent64_shape_ok(a, dae)) return launch_latent64(a, dae, pack, stream);
    if (latentw_shape_ok(a, dae)) return launch_latent_wide(a, dae, pack, stream);
    // K1x (one wave per 4 trajectories, no LDS exchange) where it takes the shape, up to ONE wave per SIMD (1024 SIMDs x 4 trajectories, + 1/8
    // of slack): beyond, two or more tiles per CU hide K1's exchanges and K1 is the faster one (profiles/r05f_tile_vs_wave.txt: B = 8192
    // 0.752 vs 0.750, 12288 0.779 vs 0.757).  PSNODE_KERNEL_MFMA_TILE / _WAVE force either.
    if (!dae && mfma_x_ode_preferred(a)) return launch_mfma_x(a, pack, stream);
    if (dae && mfma_x_dae_preferred(a)) return launch_mfma_xd(a, pack, stream);
    switch (padded_hidden_fwd(a.de.out_dim[0])) {
        case 32: return launch_mfma_h32(a, dae, pack, stream);
        case 128: return launch_mfma_h128(a, dae, pack, stream);
        case 192: return launch_mfma_h192(a, dae, pack, stream);
        case 256: return launch_mfma_h256(a, dae, pack, stream);
        default: return launch_mfma_nw<NW>(a, dae, pack, stream);
    }
}

}  // namespace psnode
