// K11 -- one linear layer over rows with a fused epilogue, on MFMA (round 6):
//     Y[r][n] = epi( sum_k X[r][k] * Wm[n][k] + bias[n] ),   r < rows,  K, N <= 128
//     epi: 0 = identity, 1 = ELU(alpha 1), 2 = multiply by ELU'(Hh[r][n]) with Hh = a saved ELU OUTPUT (h > -1:  ELU' = min(h, 0) + 1)
// The encoders / decoders of the direct_encode models (nn.Sequential(Linear, ELU, Linear) over every (b, t) row:
// neural_00_ODE_02_direct_encode.py:64-69, 74-88; neural_01_DAE_02_direct_encode.py:107-118, 126-152) at the hidden widths the two-layer
// row kernels K3b do not carry -- the scripts' argparse default --hidden 128 -- forward AND backward (delta = (g W2) * ELU'(h), grad_in =
// delta W1: the same kernel with the weight matrix read transposed), and the row-wise products of the latent-wide backward (fused/latent.py).
// Before this kernel those were ATen: addmm + a separate ELU pass + a separate ELU-backward pass per layer.
//
// v_mfma_f32_16x16x4_f32, a wave owns 16 RT rows: D[i = output n][j = row] += A[i][k] * B[k][j], A = weights (from an LDS image of the whole
// matrix, <= 64 KB, built once per workgroup), B = the rows.  VEC path (K % 16 == 0, 16-byte aligned rows): lane (k, j) loads float4
// X[row j][16 q + 4 k .. + 3]; component c of that register is the B operand of MFMA (q, c), whose contraction slots are the columns
// 16 q + 4 k + c -- any bijection of the columns onto (MFMA, slot) is valid as long as the weight image uses the same one, and this one makes
// a row's loads 64 contiguous bytes per instruction.  Scalar path (K < 16 or unaligned: the encoders' first layer, the decoders' gradient):
// column 4 m + k.  Output: lane (g, j), register r = Y[row j][16 nt + 4 g + r]: one float4 store per lane and output tile.
#include "psnode_common.h"

namespace psnode {
namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

struct LinRowsDev {
    const float *X, *W, *bias, *Hh;
    float* Y;
    long long rows, ldx, ldy, ldh, w_sn, w_sk;
    int K, N, epi;
};

__device__ __forceinline__ f4 lr_mfma(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

constexpr int kLrRT = 2;        // row tiles per wave and pass: every weight operand read from LDS feeds RT MFMAs

// KQ = ceil(K / 16) (VEC: float4 loads) or ceil(K / 4) MFMA steps (scalar: KQ counts steps of 4 columns)
template <int KS, bool VEC>
__global__ __launch_bounds__(256) void linear_rows_kernel(const LinRowsDev a) {
    extern __shared__ __attribute__((aligned(16))) float wimg[];      // [nt][step][lane]: the A operand of MFMA `step` of output tile nt
    const int l = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int k = l >> 4, j = l & 15, g = k;
    const int NT = (a.N + 15) / 16;
    constexpr int STEPS = VEC ? 4 * KS : KS;                          // MFMAs per output tile
    // ---- weight image: A operand lane (k, i) of step s of tile nt = Wm[16 nt + i][col(s, k)]
    for (int idx = threadIdx.x; idx < NT * STEPS * 64; idx += 256) {
        const int lane = idx & 63, s = (idx >> 6) % STEPS, nt = (idx >> 6) / STEPS;
        const int kk = lane >> 4, i = lane & 15, n = 16 * nt + i;
        const int col = VEC ? 16 * (s >> 2) + 4 * kk + (s & 3) : 4 * s + kk;
        wimg[idx] = (n < a.N && col < a.K) ? a.W[(long long)n * a.w_sn + (long long)col * a.w_sk] : 0.0f;
    }
    __syncthreads();
    const long long tiles = (a.rows + 16 * kLrRT - 1) / (16 * kLrRT);
    for (long long tile = (long long)blockIdx.x * 4 + w; tile < tiles; tile += (long long)gridDim.x * 4) {
        const long long r0 = tile * 16 * kLrRT;
        // ---- the rows: B operands of every step, for both row tiles
        float xb[kLrRT][STEPS];
#pragma unroll
        for (int rt = 0; rt < kLrRT; ++rt) {
            const long long row = r0 + 16 * rt + j;
            const long long rc = row < a.rows ? row : a.rows - 1;
            const float* px = a.X + rc * a.ldx;
            if constexpr (VEC) {
#pragma unroll
                for (int q = 0; q < KS; ++q) {
                    const f4 v = *reinterpret_cast<const f4*>(px + 16 * q + 4 * k);
#pragma unroll
                    for (int c = 0; c < 4; ++c) xb[rt][4 * q + c] = v[c];
                }
            } else {
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const int col = 4 * s + k;
                    const float v = px[col < a.K ? col : 0];
                    xb[rt][s] = col < a.K ? v : 0.0f;
                }
            }
        }
        for (int nt = 0; nt < NT; ++nt) {
            const int n0 = 16 * nt + 4 * g;
            f4 acc[kLrRT];
            {
                f4 b4 = f4{0.f, 0.f, 0.f, 0.f};
                if (a.bias) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) b4[r] = n0 + r < a.N ? a.bias[n0 + r] : 0.0f;
                }
#pragma unroll
                for (int rt = 0; rt < kLrRT; ++rt) acc[rt] = b4;
            }
            const float* wt = wimg + (size_t)nt * STEPS * 64 + l;
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                const float wa = wt[s * 64];
#pragma unroll
                for (int rt = 0; rt < kLrRT; ++rt) acc[rt] = lr_mfma(wa, xb[rt][s], acc[rt]);
            }
#pragma unroll
            for (int rt = 0; rt < kLrRT; ++rt) {
                const long long row = r0 + 16 * rt + j;
                if (row >= a.rows || n0 >= a.N) continue;
                f4 y = acc[rt];
                if (a.epi == 1) y = elu_quad(y);
                const bool full = n0 + 3 < a.N;
                if (a.epi == 2) {
                    const float* ph = a.Hh + row * a.ldh + n0;
                    f4 h = f4{0.f, 0.f, 0.f, 0.f};
                    if (full && !(a.ldh & 3)) h = *reinterpret_cast<const f4*>(ph);
                    else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) h[r] = n0 + r < a.N ? ph[r] : 0.0f;
                    }
                    y = y * elu_grad_quad(h);
                }
                float* py = a.Y + row * a.ldy + n0;
                if (full && !(a.ldy & 3)) *reinterpret_cast<f4*>(py) = y;
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n0 + r < a.N) py[r] = y[r];
                }
            }
        }
    }
}

template <int KS, bool VEC>
hipError_t lr_launch(const LinRowsDev& d, hipStream_t s) {
    const int NT = (d.N + 15) / 16, steps = VEC ? 4 * KS : KS;
    const size_t lds = (size_t)NT * steps * 64 * sizeof(float);
    auto kern = &linear_rows_kernel<KS, VEC>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const long long tiles = (d.rows + 16 * kLrRT - 1) / (16 * kLrRT);
    long long nwg = (tiles + 3) / 4;
    if (nwg > 2048) nwg = 2048;                 // grid-stride over the row tiles: the weight image is built once per workgroup
    if (nwg < 1) nwg = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), lds, s, d);
    return hipGetLastError();
}

}  // namespace
}  // namespace psnode

using namespace psnode;

extern "C" int32_t psnode_linear_rows_supported(const psnode_linear_rows_args_f32* a) {
    if (!a || a->rows < 0 || a->K < 1 || a->N < 1 || a->K > 128 || a->N > 128) return 0;
    if (a->epi < 0 || a->epi > 2 || a->ldx < a->K || a->ldy < a->N) return 0;
    if (a->epi == 2 && a->ldh < a->N) return 0;
    return 1;
}

extern "C" int32_t psnode_linear_rows_f32(const psnode_linear_rows_args_f32* a, void* stream) {
    if (!a) return PSNODE_ERR_NULL;
    if (!psnode_linear_rows_supported(a)) return PSNODE_ERR_UNSUPPORTED;
    if (a->rows == 0) return PSNODE_OK;
    if (!a->X || !a->W || !a->Y || (a->epi == 2 && !a->Hh)) return PSNODE_ERR_NULL;
    LinRowsDev d;
    d.X = a->X; d.W = a->W; d.bias = a->bias; d.Hh = a->Hh; d.Y = a->Y;
    d.rows = a->rows; d.ldx = a->ldx; d.ldy = a->ldy; d.ldh = a->ldh; d.w_sn = a->w_stride_n; d.w_sk = a->w_stride_k;
    d.K = a->K; d.N = a->N; d.epi = a->epi;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool vec = (a->K % 16 == 0) && !(a->ldx & 3) && !(reinterpret_cast<uintptr_t>(a->X) & 15);
    hipError_t e;
    if (vec) {
        switch (a->K / 16) {
            case 1: e = lr_launch<1, true>(d, s); break;
            case 2: e = lr_launch<2, true>(d, s); break;
            case 3: e = lr_launch<3, true>(d, s); break;
            case 4: e = lr_launch<4, true>(d, s); break;
            case 5: e = lr_launch<5, true>(d, s); break;
            case 6: e = lr_launch<6, true>(d, s); break;
            case 7: e = lr_launch<7, true>(d, s); break;
            default: e = lr_launch<8, true>(d, s); break;
        }
    } else {
        const int ks = (a->K + 3) / 4;
        if (ks <= 1) e = lr_launch<1, false>(d, s);
        else if (ks <= 2) e = lr_launch<2, false>(d, s);
        else if (ks <= 4) e = lr_launch<4, false>(d, s);
        else if (ks <= 8) e = lr_launch<8, false>(d, s);
        else if (ks <= 16) e = lr_launch<16, false>(d, s);
        else e = lr_launch<32, false>(d, s);
    }
    return e == hipSuccess ? PSNODE_OK : PSNODE_ERR_HIP;
}
