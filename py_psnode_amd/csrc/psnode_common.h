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
    int maxw;              // widest activation vector incl. the MLP inputs (generic kernel buffer A)
    int maxo;              // widest layer OUTPUT (generic kernel buffer B: it only ever holds layer outputs)
};

// ELU(alpha=1) with the negative branch at expm1 quality: ATen's CPU kernel (what the reference runs) returns
// expm1(x) for x <= 0 -- checked bitwise in the build container ("ELU").
// libm flavour, used by the generic kernel:
__device__ __forceinline__ float elu1(float x) { return x > 0.0f ? x : expm1f(x); }

// Inline flavour for the MFMA kernels (no libm call, branch free):
//   xn = min(x, 0);  q(xn) = degree-7 Taylor of expm1 for xn > -0.25 (truncation 1.5e-9 relative),
//   exp2(xn*log2e) - 1 below (result in (-1, -0.22]: absolute error ~1 ulp of exp, <= 1.4e-7 relative);
//   ELU(x) = max(x, 0) + q(xn)   (one of the two terms is exactly 0).
// This min / max+add form measured 2 % faster than a v_med3 form and 6 % faster than a sign select on K1
// (profiles/scripts/k1_elu_ab.sh, interleaved A/B on one MI355X: 4.998 / 5.110 / 5.298 ms).
// PSNODE_ELU=1 (experiments only) is the exp2-only form: same trajectory-level error on the goldens but only ABSOLUTE
// 2^-24 accuracy near 0-, i.e. no relative accuracy for tiny activations -- not used.
#ifndef PSNODE_ELU
#define PSNODE_ELU 0
#endif
__device__ __forceinline__ float elu_fast(float x) {
#if defined(PSNODE_ABLATE) && (PSNODE_ABLATE & 1)   // timing experiment: ELU -> one max (WRONG results)
    return fmaxf(x, -0.5f);
#elif PSNODE_ELU == 1
    return fmaxf(x, __builtin_amdgcn_exp2f(fminf(x, 0.0f) * 1.44269504088896340736f) - 1.0f);
#else
    const float xn = fminf(x, 0.0f);
#if PSNODE_ELU == 5     // A/B: Estrin evaluation of the same polynomial (shorter dependency chain, 2 more ops)
    const float x2 = xn * xn, x4 = x2 * x2;
    const float qa = fmaf(xn, 0.5f, 1.0f), qb = fmaf(xn, 1.0f / 24.0f, 1.0f / 6.0f), qc = fmaf(xn, 1.0f / 720.0f, 1.0f / 120.0f);
    const float lo = fmaf(qb, x2, qa), hi = fmaf(x2, 1.0f / 5040.0f, qc);
    float p = xn * fmaf(hi, x4, lo);
#else
    float p = fmaf(xn, 1.0f / 5040.0f, 1.0f / 720.0f);
    p = fmaf(xn, p, 1.0f / 120.0f);
    p = fmaf(xn, p, 1.0f / 24.0f);
    p = fmaf(xn, p, 1.0f / 6.0f);
    p = fmaf(xn, p, 0.5f);
    p = fmaf(xn, p, 1.0f);
    p = xn * p;
#endif
    const float e = __builtin_amdgcn_exp2f(xn * 1.44269504088896340736f) - 1.0f;
    return fmaxf(x, 0.0f) + (xn > -0.25f ? p : e);
#endif
}

// Cooperative copy of `count` floats global -> LDS by a 256-thread workgroup (float4 when both sides are 16-B aligned),
// followed by a barrier.  Used by the generic kernels to stage one layer's weights (or a chunk of it) per use.
constexpr int kWBuf = 4096;   // floats (16 KB)
__device__ __forceinline__ void stage_weights(const float* __restrict__ src, float* dst, int count) {
    const int tid = threadIdx.x, nt = blockDim.x;
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const int n4 = count >> 2;
        for (int i = tid; i < n4; i += nt) reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
        for (int i = (n4 << 2) + tid; i < count; i += nt) dst[i] = src[i];
    } else {
        for (int i = tid; i < count; i += nt) dst[i] = src[i];
    }
    __syncthreads();
}

// LDS-only workgroup barrier: waits for this wave's LDS traffic (lgkmcnt), not for its outstanding global prefetches --
// a plain __syncthreads() would also drain vmcnt and serialise the one-step-ahead loads of the MFMA kernels.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// fp32(1/3): the reference multiplies fp32 tensors by the python double 1/3, which ATen rounds to fp32
// (my_fixed_grid.py:8,43-44).
constexpr float kOneThird = 0.333333343267440796f;

// Butcher tableau of the three fixed-grid methods (my_fixed_grid.py:12-59): a(s, j) = coefficient of k_j in the input of
// stage s, b(s) = weight of k_s in the update.  RK4 is the 3/8 rule.  Used by the backward kernels (adjoint recursion).
__host__ __device__ __forceinline__ constexpr int rk_stages(int method) { return method == PSNODE_EULER ? 1 : (method == PSNODE_MIDPOINT ? 2 : 4); }
__host__ __device__ __forceinline__ constexpr float rk_a(int method, int s, int j) {
    if (method == PSNODE_MIDPOINT) return 0.5f;
    if (method == PSNODE_RK4_38) {
        if (s == 1) return kOneThird;
        if (s == 2) return j == 0 ? -kOneThird : 1.0f;
        if (s == 3) return j == 1 ? -1.0f : 1.0f;
    }
    return 0.0f;
}
__host__ __device__ __forceinline__ constexpr float rk_b(int method, int s) {
    if (method == PSNODE_EULER) return 1.0f;
    if (method == PSNODE_MIDPOINT) return s == 1 ? 1.0f : 0.0f;
    return (s == 0 || s == 3) ? 0.125f : 0.375f;
}

// psnode_generic.hip
hipError_t launch_generic(const IntegrateDev& a, bool dae, hipStream_t stream);
size_t generic_lds_bytes(const IntegrateDev& a, bool dae);
hipError_t launch_pack_transpose(const MlpDev& de, const MlpDev* ae, hipStream_t stream);

// psnode_mfma.hip
bool mfma_ode_supported(const IntegrateDev& a);
bool mfma_dae_supported(const IntegrateDev& a);
size_t mfma_pack_floats(const psnode_mlp_f32* de, const psnode_mlp_f32* ae);
hipError_t launch_mfma(const IntegrateDev& a, bool dae, float* pack, hipStream_t stream);
// psnode_capi.hip: fixed-order sum of per-workgroup partial vectors (parameter gradients of every backward kernel)
hipError_t launch_reduce_partials(const float* part, float* out_a, float* out_b, int np_a, int np_b, int nparts, hipStream_t s);
// K8 (psnode_latent_bwd.hip): backward of the latent ODE integrator at hidden 16
bool latent16_dae_bwd_shape_ok(const psnode_dae_bwd_args_f32* a);
bool latent16_dae_bwd_ptrs_ok(const psnode_dae_bwd_args_f32* a);
size_t latent16_dae_bwd_workspace_floats(const psnode_dae_bwd_args_f32* a);
int latent16_dae_bwd_launch(const psnode_dae_bwd_args_f32* a, float* workspace, hipStream_t s);
bool latent_bwd_shape_ok(const psnode_ode_bwd_args_f32* a);
bool latent_bwd_ptrs_ok(const psnode_ode_bwd_args_f32* a);
size_t latent_bwd_workspace_floats(long long B);
int latent_bwd_launch(const psnode_ode_bwd_args_f32* a, float* workspace, hipStream_t s);
// K9 (psnode_latent64_bwd.hip): backward of the latent ODE / DAE integrators at hidden 64
bool latent64_ode_bwd_shape_ok(const psnode_ode_bwd_args_f32* a);
bool latent64_ode_bwd_ptrs_ok(const psnode_ode_bwd_args_f32* a);
size_t latent64_ode_bwd_workspace_floats(long long B);
int latent64_ode_bwd_launch(const psnode_ode_bwd_args_f32* a, float* workspace, hipStream_t s);
bool latent64_dae_bwd_shape_ok(const psnode_dae_bwd_args_f32* a);
bool latent64_dae_bwd_ptrs_ok(const psnode_dae_bwd_args_f32* a);
size_t latent64_dae_bwd_workspace_floats(const psnode_dae_bwd_args_f32* a);
int latent64_dae_bwd_launch(const psnode_dae_bwd_args_f32* a, float* workspace, hipStream_t s);
// K7 (psnode_dae_backward.hip): MFMA backward of the DAE integrator at hidden 64
bool dae_mfma_bwd_shape_ok(const psnode_dae_bwd_args_f32* a);
size_t dae_mfma_bwd_workspace_floats(const psnode_dae_bwd_args_f32* a);
int dae_mfma_bwd_launch(const psnode_dae_bwd_args_f32* a, float* workspace, hipStream_t s);
hipError_t launch_mfma_h32(const IntegrateDev& a, bool dae, float* pack, hipStream_t stream);    // psnode_mfma_h32.hip
hipError_t launch_mfma_h128(const IntegrateDev& a, bool dae, float* pack, hipStream_t stream);   // psnode_mfma_h128.hip

// psnode_generic_bwd.hip (K5: generic fused backward, ODE and DAE)
size_t generic_bwd_workspace_floats(const psnode_mlp_f32* de, const psnode_mlp_f32* ae, long long B);
int generic_bwd_fits(const psnode_mlp_f32* de, const psnode_mlp_f32* ae, int xd, int zd, int vd, int id);
int generic_backward_launch(int method, int xd, int zd, int vd, int id, long long T, long long B, const psnode_mlp_f32* de,
                            const psnode_mlp_f32* ae, ViewDev t, ViewDev z, ViewDev v, const float* a0, const int* ev, const float* zj,
                            long long zjb, long long zje, const float* vj, long long vjb, long long vje, int n_events, const float* xs,
                            const float* is_, const float* gxs, const float* gis, float* gx0, float* gz, float* gv, float* gzj, float* gvj,
                            float* ga0, float* gparams_de, float* gparams_ae, float* workspace, hipStream_t stream);

// psnode_latent.hip (direct_encode latent shapes, hidden_dim 16)
bool latent_shape_ok(const IntegrateDev& a, bool dae);
bool latent_ptrs_ok(const IntegrateDev& a, bool dae);
size_t latent_pack_floats();
hipError_t launch_latent(const IntegrateDev& a, bool dae, float* pack, hipStream_t stream);

// psnode_latent64.hip (direct_encode latent shapes, hidden_dim 64)
bool latent64_shape_ok(const IntegrateDev& a, bool dae);
bool latent64_ptrs_ok(const IntegrateDev& a, bool dae);
size_t latent64_pack_floats();
hipError_t launch_latent64(const IntegrateDev& a, bool dae, float* pack, hipStream_t stream);

}  // namespace psnode
