// K3b -- row-wise ELU-MLP for the direct_encode encoders / decoders: out[r] = W2 . ELU(W1 . in[r] + b1) + b2 for every
// (b,t) row (neural_00_ODE_02_direct_encode.py:64-69,74-88; neural_01_DAE_02_direct_encode.py:107-118,126-152).
//
// M = B*T ~ 4.1 M rows with K, N <= 16: a skinny GEMM pair that is HBM-bound (enc x: 32 B in + 64 B out per row vs
// 384 flop), so the kernel is a stream: one wave takes 16 rows (= the N of v_mfma_f32_16x16x4_f32), loads them with
// coalesced 8/16-byte accesses (lane (g, row) reads columns NM*g .. NM*g+NM-1), keeps the hidden tile in registers
// (D rows of L1 are the B operands of L2, as in psnode_latent.hip) and writes one float4 per lane.  Weights (< 1 KB)
// sit in VGPRs; nothing is staged through LDS.
#include "psnode_common.h"

namespace psnode {
namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int RH = 16;

__device__ __forceinline__ f4 rmfma(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

struct RowsArgs {
    const float *w1, *b1, *w2, *b2, *in;
    float* out;
    long long rows, in_stride, out_stride;
    int in_dim, out_dim;
};

// NM = MFMAs of layer 1 = ceil(in_dim / 4); lane group g supplies columns NM*g + m.
template <int NM>
__global__ __launch_bounds__(256) void rows_kernel(const RowsArgs a) {
    const int l = threadIdx.x & 63, g = l >> 4, j = l & 15, i = l & 15;
    // weights -> registers
    float w1[NM], w2[4];
    f4 b1r, b2r;
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        const int c = NM * g + m;
        w1[m] = c < a.in_dim ? a.w1[i * a.in_dim + c] : 0.0f;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        w2[r] = i < a.out_dim ? a.w2[i * RH + 4 * g + r] : 0.0f;
        b1r[r] = a.b1[4 * g + r];
        b2r[r] = 4 * g + r < a.out_dim ? a.b2[4 * g + r] : 0.0f;
    }
    const long long tiles = (a.rows + 15) / 16;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    const bool vec_in = (NM == 2 || NM == 4) && a.in_dim == 4 * NM && a.in_stride % NM == 0 && (reinterpret_cast<uintptr_t>(a.in) % (4 * NM)) == 0;
    const bool vec_out = a.out_dim % 4 == 0 && a.out_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0;
    for (long long t = wave; t < tiles; t += nwaves) {
        const long long row = t * 16 + j;
        const bool valid = row < a.rows;
        const float* src = a.in + (valid ? row : a.rows - 1) * a.in_stride + NM * g;
        float v[NM];
        if (vec_in) {
            if constexpr (NM == 4) {
                const f4 q = *reinterpret_cast<const f4*>(src);
                v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
            } else if constexpr (NM == 2) {
                const float2 q = *reinterpret_cast<const float2*>(src);
                v[0] = q.x; v[1] = q.y;
            }
        } else {
#pragma unroll
            for (int m = 0; m < NM; ++m) v[m] = NM * g + m < a.in_dim ? src[m] : 0.0f;
        }
        f4 acc = b1r;
#pragma unroll
        for (int m = 0; m < NM; ++m) acc = rmfma(w1[m], v[m], acc);
        const f4 h = f4{elu_fast(acc[0]), elu_fast(acc[1]), elu_fast(acc[2]), elu_fast(acc[3])};
        f4 oA = rmfma(w2[0], h[0], b2r), oB = rmfma(w2[1], h[1], f4{0.f, 0.f, 0.f, 0.f});
        oA = rmfma(w2[2], h[2], oA);
        oB = rmfma(w2[3], h[3], oB);
        const f4 o = oA + oB;
        if (valid && 4 * g < a.out_dim) {
            float* dst = a.out + row * a.out_stride + 4 * g;
            if (vec_out) {
                *reinterpret_cast<f4*>(dst) = o;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (4 * g + r < a.out_dim) dst[r] = o[r];
            }
        }
    }
}

}  // namespace
}  // namespace psnode

using namespace psnode;

extern "C" int32_t psnode_mlp_rows_supported(const psnode_mlp_f32* m) {
    return m && m->n_layers == 2 && m->in_dim >= 1 && m->in_dim <= 16 && m->out_dim[0] == RH && m->out_dim[1] >= 1 && m->out_dim[1] <= 16;
}

extern "C" int32_t psnode_mlp_rows_f32(const psnode_mlp_f32* m, int64_t rows, const float* in, int64_t in_row_stride, float* out,
                                       int64_t out_row_stride, void* stream) {
    if (!m || !in || !out) return PSNODE_ERR_NULL;
    if (!psnode_mlp_rows_supported(m)) return PSNODE_ERR_UNSUPPORTED;
    if (!m->weight[0] || !m->weight[1] || !m->bias[0] || !m->bias[1]) return PSNODE_ERR_NULL;
    if (rows < 0 || in_row_stride < m->in_dim || out_row_stride < m->out_dim[1]) return PSNODE_ERR_DIMS;
    if (rows == 0) return PSNODE_OK;
    RowsArgs a{m->weight[0], m->bias[0], m->weight[1], m->bias[1], in, out, rows, in_row_stride, out_row_stride, m->in_dim, m->out_dim[1]};
    const long long tiles = (rows + 15) / 16;
    long long blocks = (tiles + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;   // 8 workgroups per CU, grid-stride over the rest
    const dim3 grid((unsigned)blocks), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int NM = (m->in_dim + 3) / 4;
    switch (NM) {
        case 1: hipLaunchKernelGGL(rows_kernel<1>, grid, block, 0, s, a); break;
        case 2: hipLaunchKernelGGL(rows_kernel<2>, grid, block, 0, s, a); break;
        case 3: hipLaunchKernelGGL(rows_kernel<3>, grid, block, 0, s, a); break;
        default: hipLaunchKernelGGL(rows_kernel<4>, grid, block, 0, s, a); break;
    }
    return hipGetLastError() == hipSuccess ? PSNODE_OK : PSNODE_ERR_HIP;
}
