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
    const float* wt[kMaxLayers];     // workspace: MFMA image + padded bias (generic forward) / transposed [in,out] (generic backward)
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
    float* sact;           // ODE training forward: saved activations [T-1,S,3,B,Hp] / stage inputs [T-1,S,B,xd], or null
    float* sxst;
    float* saeact;         // DAE training forward: AE head activations per grid point [3,T,B,Hp], per event [nE,3,B,Hp] and the event's i0 [nE,B,16]
    float* sevact;
    float* sevi;
    int maxw;              // widest activation vector incl. the MLP inputs (generic kernel buffer A)
    int maxo;              // widest layer OUTPUT (generic kernel buffer B: it only ever holds layer outputs)
    int kern;              // the caller's psnode_*_args_f32::kernel (PSNODE_KERNEL_MFMA_TILE / _WAVE pick between K1 and K1x)
    unsigned k0_res;       // generic kernel: which layers' weight images are resident in LDS (bit l: DE layer l, bit 8 + l: AE layer l)
};

// ELU(alpha=1) with the negative branch at expm1 quality: ATen's CPU kernel (what the reference runs) returns
// expm1(x) for x <= 0 -- checked bitwise in the build container ("ELU").
// libm flavour, used by the generic kernel:
__device__ __forceinline__ float elu1(float x) { return x > 0.0f ? x : expm1f(x); }

// Inline flavour for the MFMA / DPP kernels (no libm call, branch free, no select), written on float PAIRS so that the arithmetic
// lowers to v_pk_{mul,add}_f32.  On gfx950 the fp32 MFMA shares the fp32 datapath with the VALU: VALU work next to
// v_mfma_f32_16x16x4_f32 is ADDITIVE (32 cycles per MFMA + 5 per VALU instruction + 12 per v_exp, one wave per SIMD;
// profiles/r02a_ubench_mfma.txt), so the ELU's instruction count is wall time of every kernel in this library.
//
// Round 3 (default):   ELU(x) = max(x, exp2(min(x, 0) * log2e) - 1)        [= max(x, 0) + (exp2(min(x, 0) log2e) - 1), one add less]
//   x > 0: the exponential side is exp2(0) - 1 = 0 exactly, ELU(x) = x bit for bit.  x <= 0: t = exp2(..) carries <= 1 ulp(t) <= 6e-8
//   of ABSOLUTE error, t - 1 is exact for t >= 0.5 (Sterbenz) and rounds once below: |ELU - expm1(x)| <= 1.2e-7 everywhere.  What it
//   gives up against the round-1/2 form below is RELATIVE accuracy for x -> 0-: the result is a multiple of 6e-8 there.  The next
//   thing that happens to an ELU output is a dot product with a weight row, where an absolute 6e-8 on a value that is itself tiny
//   is what an fp32 sum loses anyway: measured on the reference's goldens the trajectories are as close to the reference as with the
//   expm1-quality form (per-trajectory error 1.05e-7 vs 2.1e-7 over 1000 RK4 steps, K2 1.5e-7 vs 1.0e-7; profiles/r03b_elu_exp2_ab.txt)
//   -- two orders inside the 1e-5 gate -- for 4.5 instead of 8 issue slots per value: K1 4.21 -> 3.87 ms, K2 5.60 -> 5.20 ms.
//
// -DPSNODE_ELU_EXPM1 (rounds 1-2): negative branch at expm1 quality, as ATen's CPU ELU (relative accuracy down to denormal x):
//   xc = med3(x, knee, 0)  in [knee, 0]      knee = -0.25
//   xe = min(x, knee)      in (-inf, knee]
//   ELU(x) = max(x, 0) + [ xc * q(xc) + (exp2(xe * log2e) - t0) ],   t0 = exp2(knee * log2e) evaluated by the same instructions
//   * x >= knee: xe = knee, the exp term is EXACTLY 0 and xc * q(xc) is expm1 with q the degree-4 near-minimax of expm1(x)/x
//     on [-0.25, 0] (max relative error 9.1e-8 in fp32 evaluation);
//   * x <  knee: xc = knee, result = expm1(knee)[poly] + (e^x - e^knee): absolute error <= 2 ulp of exp on a value <= -0.22.
//   The knee is made opaque to the compiler so that t0 is produced by the hardware v_exp_f32 (not constant-folded by a host
//   libm whose last bit may differ): exp2(xe * log2e) - t0 must cancel EXACTLY for x >= knee.
typedef float elu_f2 __attribute__((ext_vector_type(2)));
typedef float elu_f4 __attribute__((ext_vector_type(4)));
constexpr float kLog2e = 1.44269504088896340736f;
__device__ __forceinline__ float elu_knee() {
    float c = -0.25f;
    asm("" : "+v"(c));          // not volatile: loop-invariant, the compiler hoists it and the exp2 below out of the time loop
    return c;
}
// Polynomial coefficients as register PAIRS: with literal operands the compiler splits about a fifth of the packed FMAs back
// into two scalar v_fmaak/v_fmamk (literal forms exist only for the scalar opcode).  Translation units whose kernels are
// register-bound (the backward kernels, hidden 128) define PSNODE_ELU_LITERALS before including this header and keep the literals.
__device__ __forceinline__ elu_f2 elu_splat(float c) {
    elu_f2 r = elu_f2{c, c};
#ifndef PSNODE_ELU_LITERALS
    asm("" : "+v"(r));
#endif
    return r;
}
__device__ __forceinline__ elu_f2 elu_pair(const elu_f2 x, const float knee, const float neg_t0, const elu_f2 c4, const elu_f2 c3,
                                           const elu_f2 c2, const elu_f2 c1) {
#ifndef PSNODE_ELU_EXPM1
    // = max(x, exp2(min(x,0) log2e) - 1): e^x - 1 >= x everywhere, so the max picks x for x > 0 (where the other side is exactly 0)
    // and the exponential side for x <= 0 -- one packed add less than max(x,0) + (..)
    // Round 5: exp2(min(x, 0) log2e) == clamp(exp2(x log2e), 0, 1) bit for bit (for x <= 0 the same product goes through the same
    // v_exp_f32; for x > 0 both are exactly 1), and the clamp is the VOP3 OUTPUT MODIFIER of v_exp_f32 (`v_exp_f32_e64 v, v clamp`:
    // the compiler folds fmed3(e, 0, 1) into it under the default DX10_CLAMP mode) -- the two v_min_f32 per pair are gone: 3 issue
    // slots per value instead of 4 (profiles/r05_ubench_4x4.txt checks the two forms bitwise over 3 x 2^24 inputs).
    const elu_f2 yq = x * kLog2e;
    const elu_f2 tq = elu_f2{__builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(yq[0]), 0.0f, 1.0f),
                             __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(yq[1]), 0.0f, 1.0f)};
    const elu_f2 uq = tq - 1.0f;
    return elu_f2{fmaxf(x[0], uq[0]), fmaxf(x[1], uq[1])};
#endif
    const elu_f2 xc = elu_f2{__builtin_amdgcn_fmed3f(x[0], knee, 0.0f), __builtin_amdgcn_fmed3f(x[1], knee, 0.0f)};
    const elu_f2 xe = elu_f2{fminf(x[0], knee), fminf(x[1], knee)};
    const elu_f2 xp = elu_f2{fmaxf(x[0], 0.0f), fmaxf(x[1], 0.0f)};
    const elu_f2 y = xe * kLog2e;
    const elu_f2 t = elu_f2{__builtin_amdgcn_exp2f(y[0]), __builtin_amdgcn_exp2f(y[1])};
    const elu_f2 u = t + neg_t0;
    elu_f2 q = __builtin_elementwise_fma(xc, c4, c3);
    q = __builtin_elementwise_fma(xc, q, c2);
    q = __builtin_elementwise_fma(xc, q, c1);
    q = __builtin_elementwise_fma(xc, q, elu_f2{1.0f, 1.0f});
    return xp + __builtin_elementwise_fma(xc, q, u);
}
// everything up to the first elu_pair is loop-invariant: the compiler hoists it out of the time loop
#define PSNODE_ELU_CONSTS                                                                   \
    const float knee = elu_knee();                                                          \
    const float neg_t0 = -__builtin_amdgcn_exp2f(knee * kLog2e);                            \
    const elu_f2 c4 = elu_splat(0.007513605989515781f), c3 = elu_splat(0.04149065539240837f), \
                 c2 = elu_splat(0.16665108501911163f), c1 = elu_splat(0.4999995231628418f);
__device__ __forceinline__ elu_f4 elu_quad(const elu_f4 v) {
    PSNODE_ELU_CONSTS
    const elu_f2 a = elu_pair(elu_f2{v[0], v[1]}, knee, neg_t0, c4, c3, c2, c1);
    const elu_f2 b = elu_pair(elu_f2{v[2], v[3]}, knee, neg_t0, c4, c3, c2, c1);
    return elu_f4{a[0], a[1], b[0], b[1]};
}
// Round 5, the inference forwards K1 / K2: ELU in the log2(e)-scaled domain.  With p' = log2e * p delivered by the MFMAs themselves
// (PackMfma::scaled), g = log2e * ELU(p) = max(p', log2e * clamp(exp2(p')) - log2e): v_exp (clamp) -> v_pk_fma -> v_max, a dependent
// chain of THREE instructions and 2.5 issue slots per value; the unscaled form's leading v_pk_mul is gone (at one wave per SIMD every
// instruction of this chain is exposed latency: removing the v_min alone took K1 from 3.89 to 3.68 ms).  p' > 0: the fma is exactly 0,
// g = p' bit for bit; p' <= 0: fma(log2e, e, -log2e) rounds the exact log2e (e - 1) once.
__device__ __forceinline__ elu_f4 elu_quad_scaled(const elu_f4 v) {
    const elu_f2 c = elu_f2{kLog2e, kLog2e}, nc = elu_f2{-kLog2e, -kLog2e};
    const elu_f2 ea = elu_f2{__builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(v[0]), 0.0f, 1.0f), __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(v[1]), 0.0f, 1.0f)};
    const elu_f2 eb = elu_f2{__builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(v[2]), 0.0f, 1.0f), __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(v[3]), 0.0f, 1.0f)};
    const elu_f2 ua = __builtin_elementwise_fma(c, ea, nc), ub = __builtin_elementwise_fma(c, eb, nc);
    return elu_f4{fmaxf(v[0], ua[0]), fmaxf(v[1], ua[1]), fmaxf(v[2], ub[0]), fmaxf(v[3], ub[1])};
}
__device__ __forceinline__ float elu_fast(const float x) {   // scalar form of the same function (bit-identical to elu_quad)
    PSNODE_ELU_CONSTS
    return elu_pair(elu_f2{x, x}, knee, neg_t0, c4, c3, c2, c1)[0];
}
#undef PSNODE_ELU_CONSTS

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
#ifdef PSNODE_LDS_BARRIER_SYNC      // discriminator builds only (round 4 defect chase): the full barrier, draining vmcnt too
    __syncthreads();
    return;
#endif
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// LDS-DMA of 64 lanes x 16 B (K4f / K7f image swaps): `global_load_lds_dwordx4 <lane*16>, <uniform 64-bit base>` with the LDS
// destination in M0 (saved / restored around it).  The instruction sits in inline asm, where the compiler's hazard recognizer does not
// look, so the asm has to carry its own wait states:
//   * SALU write of M0 -> LDS-DMA reads M0: 1 wait state;
//   * VALU write of an SGPR (v_readlane / v_readfirstlane: how the spilled, loop-invariant slot bases come back) -> VMEM reads that SGPR
//     as its scalar address: FIVE wait states (gfx9 `VmemSgprWaitStates`), or the load may go out with the OLD register contents.
// Rounds 2-3 had `s_nop 0` here: s_waitcnt + 2 s_mov + s_nop 0 = 4 wait states when the compiler schedules the v_readlane of the
// base's high half directly in front of the asm -- 237 such sites in the 15 recompute 8-wave instances of K4f alone
// (profiles/scripts/isa_asm_hazards.py, profiles/r04_dma_hazard_*.txt).  `s_nop 2` makes the asm self-sufficient (2 s_mov + 3 = 5
// without counting the s_waitcnt).  PSNODE_DMA_NOP=0 rebuilds the old sequence.
#ifndef PSNODE_DMA_NOP
#define PSNODE_DMA_NOP 2
#endif
#define PSNODE_STR2(x) #x
#define PSNODE_STR(x) PSNODE_STR2(x)
#define PSNODE_LDS_DMA_ASM                                                                                            \
    "s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop " PSNODE_STR(PSNODE_DMA_NOP)            \
    "\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"

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
hipError_t launch_pack_transpose(const MlpDev& de, const MlpDev* ae, hipStream_t stream);   // generic backward: transposed weights
hipError_t launch_pack_image(const MlpDev& de, const MlpDev* ae, int xd, int n, int nzv, hipStream_t stream);   // generic forward: MFMA images
size_t generic_image_floats(int K, int N);
hipError_t launch_pack_plain_images(const MlpDev& m, float* const* img, float* const* imgT, hipStream_t stream);   // generic backward, register path

// psnode_mfma.hip
bool mfma_ode_supported(const IntegrateDev& a);
int mfma_ode_save_hidden(const IntegrateDev& a);     // row width of saved activations if K1 proper takes the shape (no latent / teacher forcing), else 0
bool mfma_dae_supported(const IntegrateDev& a);
int mfma_dae_save_hidden(const IntegrateDev& a);     // likewise for K2 proper
size_t mfma_pack_floats(const psnode_mlp_f32* de, const psnode_mlp_f32* ae);
hipError_t launch_mfma(const IntegrateDev& a, bool dae, float* pack, hipStream_t stream);
// psnode_mfma_x.hip (K1x: one wave per 4 trajectories, no LDS exchange; ODE inference at hidden <= 64)
bool mfma_x_ode_supported(const IntegrateDev& a);
bool mfma_x_ode_preferred(const IntegrateDev& a);       // ... and AUTO / MFMA (or the forced _WAVE) would run it on this call (batch size)
size_t mfma_x_pack_floats();
hipError_t launch_mfma_x(const IntegrateDev& a, float* pack, hipStream_t stream);
// psnode_mfma_xd.hip (K2x: the same decomposition for the DAE; inference at hidden <= 64, i_dim <= 4, no teacher forcing)
bool mfma_x_dae_supported(const IntegrateDev& a);
bool mfma_x_dae_preferred(const IntegrateDev& a);
size_t mfma_xd_pack_floats();
hipError_t launch_mfma_xd(const IntegrateDev& a, float* pack, hipStream_t stream);
// psnode_capi.hip: fixed-order sum of per-workgroup partial vectors (parameter gradients of every backward kernel).  `part` is SCRATCH and
// is DESTROYED: with 128 or more vectors the first pass sums slices in place (the first vector of each slice then holds the slice's sum),
// so a buffer cannot be reduced twice.  The summation order is fixed for a given nparts (it changes at the 128-vector threshold).
hipError_t launch_reduce_partials(float* part, float* out_a, float* out_b, int np_a, int np_b, int nparts, hipStream_t s);
// K4f (psnode_backward_fused.hip): one-launch MFMA backward of the ODE integrator at hidden <= 128 (zero-padded to 32 / 64 / 128), x_dim <= 8, z_dim <= 8
bool fused_bwd_shape_ok(const psnode_ode_bwd_args_f32* a);
size_t fused_bwd_workspace_floats(const psnode_ode_bwd_args_f32* a);
int fused_bwd_launch(const psnode_ode_bwd_args_f32* a, float* workspace, hipStream_t s);
// K4x (psnode_backward_x.hip): the exchange-free backward of the ODE integrator at hidden 33..64 -- one wave = 4 trajectories, saved rows only
bool bwd_x_shape_ok(const psnode_ode_bwd_args_f32* a);
bool bwd_x_preferred(const psnode_ode_bwd_args_f32* a);
size_t bwd_x_workspace_floats(const psnode_ode_bwd_args_f32* a);
int bwd_x_launch(const psnode_ode_bwd_args_f32* a, float* workspace, hipStream_t s);
// K7f (psnode_dae_backward_fused.hip): the DAE backward at hidden <= 128 with the DE's parameter gradients formed in the kernel
size_t dae_fused_bwd_workspace_floats(const psnode_dae_bwd_wide_args_f32* a);
int dae_fused_bwd_launch(const psnode_dae_bwd_wide_args_f32* a, float* workspace, hipStream_t s);
size_t dae_fused_bwd_ae_floats(const psnode_dae_bwd_wide_args_f32* a);
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
hipError_t launch_mfma_h32(const IntegrateDev& a, bool dae, float* pack, hipStream_t stream);    // psnode_mfma_h32.hip
hipError_t launch_mfma_h128(const IntegrateDev& a, bool dae, float* pack, hipStream_t stream);   // psnode_mfma_h128.hip
hipError_t launch_mfma_h192(const IntegrateDev& a, bool dae, float* pack, hipStream_t stream);   // psnode_mfma_h192.hip (forward only)
hipError_t launch_mfma_h256(const IntegrateDev& a, bool dae, float* pack, hipStream_t stream);   // psnode_mfma_h256.hip (forward only)

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
// K3f (psnode_latent_dpp.hip): the latent ODE at hidden 16 on VALU + DPP row broadcasts (any alignment)
hipError_t launch_latent_dpp(const IntegrateDev& a, hipStream_t stream);
// K8f: backward through the hidden-16 latent ODE, same mapping (any alignment); workspace = per-wave parameter-gradient partials
size_t latent_bwd_dpp_workspace_floats(long long B);
int latent_bwd_dpp_launch(const psnode_ode_bwd_args_f32* a, float* workspace, hipStream_t s);

// psnode_latent64.hip (direct_encode latent shapes, hidden_dim 64)
bool latent64_shape_ok(const IntegrateDev& a, bool dae);
bool latent64_ptrs_ok(const IntegrateDev& a, bool dae);
size_t latent64_pack_floats();
hipError_t launch_latent64(const IntegrateDev& a, bool dae, float* pack, hipStream_t stream);
// psnode_latent_wide.hip (K3w: the latent shapes at every other hidden_dim <= 128, weights streamed from L2)
bool latentw_shape_ok(const IntegrateDev& a, bool dae);
bool latentw_ptrs_ok(const IntegrateDev& a, bool dae);
size_t latentw_pack_floats();
hipError_t launch_latent_wide(const IntegrateDev& a, bool dae, float* pack, hipStream_t stream);

// ELU'(pre) from h = ELU(pre): 1 for pre > 0 (h > 0), exp(pre) = h + 1 otherwise -- written min(h, 0) + 1 (identical values; one clamp
// + one packed add per pair instead of add + compare + select per value: VALU instructions are wall time next to fp32 MFMAs)
typedef float psnode_f4_ __attribute__((ext_vector_type(4)));
// (round 6: min(h, 0) written med3(h, -2, 0) -- the same value for every ELU output (h > -1) in ONE VOP3 instruction; fminf costs a
//  canonicalising v_max in front of the v_min for a value that comes from memory: 8 -> 4 instructions per tile, found in K4f's ISA)
__device__ __forceinline__ float elu_grad(float h) { return __builtin_amdgcn_fmed3f(h, -2.0f, 0.0f) + 1.0f; }
__device__ __forceinline__ psnode_f4_ elu_grad_quad(psnode_f4_ h) {
    return psnode_f4_{__builtin_amdgcn_fmed3f(h[0], -2.0f, 0.0f), __builtin_amdgcn_fmed3f(h[1], -2.0f, 0.0f),
                      __builtin_amdgcn_fmed3f(h[2], -2.0f, 0.0f), __builtin_amdgcn_fmed3f(h[3], -2.0f, 0.0f)} + 1.0f;
}

// Addressing idiom of the time-loop kernels: <uniform row base in SGPRs> + <32-bit per-lane BYTE offset> = the hardware's
// `global_* v, voffset, s[base:base+1]` form.  sbase() makes the row base opaque at each use: left visible, `base + lane offset` is
// loop-invariant per array (or a strength-reduced induction pointer), and the compiler keeps one precomputed 64-bit per-lane pointer for
// every array the loop touches -- dozens of VGPR pairs, which the register-bound kernels then spill to scratch (K7w: 1652 B -> 0; K9: 200 B
// -> 0).  The rebuilt pointer is typed GLOBAL (address space 1): a generic pointer made from integers makes every access a flat_* one with
// a 64-bit VGPR address.  ldg / stg read / write an element at a byte offset (one VGPR; an element index would be widened to 64 bits).
template <typename T>
using gptr = __attribute__((address_space(1))) T*;
template <typename T>
__device__ __forceinline__ gptr<T> sbase(T* p) {
    unsigned lo = (unsigned)reinterpret_cast<uintptr_t>(p), hi = (unsigned)(reinterpret_cast<uintptr_t>(p) >> 32);
    lo = __builtin_amdgcn_readfirstlane(lo);
    hi = __builtin_amdgcn_readfirstlane(hi);
    asm volatile("" : "+s"(lo), "+s"(hi));
    return (gptr<T>)(((uintptr_t)hi << 32) | lo);
}
// (the offset is made opaque too: instruction selection works per basic block, and a zero-extension hoisted out of the loop arrives as
//  a 64-bit register pair, which selects `v_lshl_add_u64` + a 64-bit vaddr instead of saddr + 32-bit voffset)
template <typename V, typename T>
__device__ __forceinline__ V ldg(gptr<T> base, unsigned byte_off) {
    asm volatile("" : "+v"(byte_off));
    return *(gptr<const V>)((gptr<const char>)base + byte_off);
}
// read-once / write-once rows (the activations a training forward saves for its backward): nontemporal at 8 waves (NT = true), so that
// they stream past L2 instead of evicting the packed weight images those kernels re-read every step (PSNODE_SAVED_NT=0: plain accesses).
// Same-box A/B (profiles/r03y_saved_nt_ab.txt), training step with / without: hidden 128 ODE 34.55 / 34.69 ms, DAE 51.7 / 52.2; at hidden 64
// (weights in registers, nothing to evict) 14.45 / 14.27 and 21.8 / 21.6 -- hence only at 8 waves.
#ifndef PSNODE_SAVED_NT
#define PSNODE_SAVED_NT 1
#endif
template <typename V, bool NT, typename T>
__device__ __forceinline__ V ldg_nt(gptr<T> base, unsigned byte_off) {
    asm volatile("" : "+v"(byte_off));
    if constexpr (NT && PSNODE_SAVED_NT) return __builtin_nontemporal_load((gptr<const V>)((gptr<const char>)base + byte_off));
    else return *(gptr<const V>)((gptr<const char>)base + byte_off);
}
template <bool NT, typename V>
__device__ __forceinline__ void store_nt(V* p, const V v) {
    if constexpr (NT && PSNODE_SAVED_NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
// Predicated row element WITHOUT control flow: the load is unconditional from an offset the caller has clamped into the row, the
// predicate only selects the value (v_cndmask).  Written `cond ? ldg(..) : 0.0f` the load sits in one arm of a branch, and every such arm
// in a time-loop kernel is a divergent EXEC region in which the register allocator may place a spill: the spill then saves only the
// lanes active THERE (none at all when the predicate is false for the whole wave) and a reload outside of the region hands the other
// lanes garbage -- round 3's defect (a), DESIGN.md "Round 4: the two fenced defects".
template <typename T>
__device__ __forceinline__ float ldg_sel(gptr<const T> base, unsigned safe_byte_off, bool on) {
    const float v = ldg<float>(base, safe_byte_off);
    return on ? v : 0.0f;
}
template <typename V, typename T>
__device__ __forceinline__ void stg(gptr<T> base, unsigned byte_off, V v) {
    asm volatile("" : "+v"(byte_off));
    *(gptr<V>)((gptr<char>)base + byte_off) = v;
}

}  // namespace psnode
