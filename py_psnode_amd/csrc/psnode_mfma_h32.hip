// K1/K2 for --hidden 32: 2 waves per 16-trajectory tile (kernel template in psnode_mfma_impl.h).
#include "psnode_mfma_impl.h"

namespace psnode {

hipError_t launch_mfma_h32(const IntegrateDev& a, bool dae, float* pack, hipStream_t stream) {
    return launch_mfma_nw<2>(a, dae, pack, stream);
}

}  // namespace psnode
