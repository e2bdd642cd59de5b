// Internal declarations shared by the HIP translation units of libpsnode_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "psnode_hip.h"

namespace psnode {

constexpr int kMaxLayers = PSNODE_MAX_LAYERS;

// Device-side MLP descriptor. `wt[l]` points into the workspace (layout depends on the kernel family),
// `bias[l]` is the caller's nn.Linear bias.
struct MlpDev {
    int n_layers;
    int in_dim;
    int out_dim[kMaxLayers];
    const float* w[kMaxLayers];      // caller's row-major [out,in]
    const float* wt[kMaxLayers];     // workspace: transposed [in,out] (generic kernel)
    const float* bias[kMaxLayers];
};

struct ViewDev {
    const float* p;
    long long st, sb;
};

// One struct for ODE and DAE: the ODE path is the DAE path with v_dim = i_dim = 0 and no AE head.
struct IntegrateDev {
    int method;
    unsigned flags;
    int xd, zd, vd, id;
    long long T, B;
    MlpDev de, ae;
    ViewDev t, x, z, v, i;
    const float* x_init;   // DAE only
    const float* a0;
    const int* ev;
    const float* zj;
    long long zjb, zje;
    const float* vj;
    long long vjb, vje;
    float* xo;
    float* io;
    int maxw;              // widest activation vector (generic kernel LDS sizing)
};

// ELU(alpha=1) with the negative branch at expm1 quality: ATen's CPU kernel (what the reference runs)
// returns expm1(x) for x <= 0 -- checked bitwise in the build container ("ELU").
__device__ __forceinline__ float elu1(float x) { return x > 0.0f ? x : expm1f(x); }

// fp32(1/3): the reference multiplies fp32 tensors by the python double 1/3, which ATen rounds to fp32
// (my_fixed_grid.py:8,43-44).
constexpr float kOneThird = 0.333333343267440796f;

// psnode_generic.hip
hipError_t launch_generic(const IntegrateDev& a, bool dae, hipStream_t stream);
size_t generic_lds_bytes(const IntegrateDev& a, bool dae);

// psnode_mfma.hip
bool mfma_ode_supported(const IntegrateDev& a);
bool mfma_dae_supported(const IntegrateDev& a);
size_t mfma_pack_floats(const psnode_mlp_f32* de, const psnode_mlp_f32* ae);
hipError_t launch_mfma(const IntegrateDev& a, bool dae, float* pack, hipStream_t stream);

// psnode_latent.hip (direct_encode latent shapes, hidden_dim 16)
bool latent_shape_ok(const IntegrateDev& a, bool dae);
bool latent_ptrs_ok(const IntegrateDev& a, bool dae);
size_t latent_pack_floats();
hipError_t launch_latent(const IntegrateDev& a, bool dae, float* pack, hipStream_t stream);

}  // namespace psnode
