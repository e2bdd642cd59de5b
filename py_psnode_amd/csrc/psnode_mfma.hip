// MFMA (v_mfma_f32_16x16x4_f32) fused integrator kernels for gfx950 -- see DESIGN.md "K1".
#include "psnode_common.h"

namespace psnode {

bool mfma_ode_supported(const IntegrateDev&) { return false; }
bool mfma_dae_supported(const IntegrateDev&) { return false; }
size_t mfma_pack_floats(const psnode_mlp_f32*, const psnode_mlp_f32*) { return 0; }
hipError_t launch_mfma(const IntegrateDev&, bool, float*, hipStream_t) { return hipErrorNotSupported; }

}  // namespace psnode
