// K1/K2 for --hidden 193..256: 16 waves per 16-trajectory tile, the H->H weights streamed from the L2-resident stream image every
// layer (kernel template in psnode_mfma_impl.h, `weights_streamed`).  Forward only: training at these widths takes K5 or the walk.
#define PSNODE_ELU_LITERALS   // register-bound kernels: ELU coefficients as literals, not as 8 resident VGPRs (psnode_common.h)
#include "psnode_mfma_impl.h"

namespace psnode {

hipError_t launch_mfma_h256(const IntegrateDev& a, bool dae, float* pack, hipStream_t stream) {
    return launch_mfma_nw<16>(a, dae, pack, stream);
}

}  // namespace psnode
