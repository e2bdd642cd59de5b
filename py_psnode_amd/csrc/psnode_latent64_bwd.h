// Shared between psnode_latent64_bwd.hip (K9, one-role kernels + host side) and psnode_latent64_bwd_roles.hip (the two-role form of the
// saved-activation DAE instance, compiled with the VGPR form of the MFMA accumulators: 2 waves per SIMD leave it 256 registers).
#pragma once
#include "psnode_common.h"

namespace psnode {

typedef float f4_9 __attribute__((ext_vector_type(4)));
struct Bwd9Dev {
    IntegrateDev a;          // t, z, v, a0, ev, zj, vj (+strides), T, B, zd, method
    const float *xs, *is_, *gxs, *gis;
    float *gx0, *gz, *gv, *gzj, *gvj, *ga0, *wpart;
    int n_events, NP_de, NP_ae;
};
hipError_t launch9_roles(int method, int nbe, const Bwd9Dev& d, const float* pde, const float* pae, hipStream_t s);

namespace {
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int H9 = 64, NW9 = 4, SCR9 = 64 * 4 + 4 * 8;
__device__ __forceinline__ f4 m9(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f4 dact9(f4 h) { return elu_grad_quad(h); }
__device__ __forceinline__ f4 z9() { return f4{0.f, 0.f, 0.f, 0.f}; }
struct A9 { f4 c[4]; };      // gradient of one 64x64 block, own 16 rows: c[chunk] rows 16w+4g+r, columns 16((w+chunk)&3) + j
}  // namespace

}  // namespace psnode
