// K1/K2 for --hidden 128: 8 waves per 16-trajectory tile (kernel template in psnode_mfma_impl.h).
#define PSNODE_ELU_LITERALS   // register-bound kernels: ELU coefficients as literals, not as 8 resident VGPRs (psnode_common.h)
#include "psnode_mfma_impl.h"

namespace psnode {

hipError_t launch_mfma_h128(const IntegrateDev& a, bool dae, float* pack, hipStream_t stream) {
    return launch_mfma_nw<8>(a, dae, pack, stream);
}

}  // namespace psnode
