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

__device__ __forceinline__ f4 rmfma(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

struct RowsArgs {
    const float *w1, *b1, *w2, *b2, *in;
    float* out;
    long long rows, in_stride, out_stride;
    int in_dim, out_dim;
};

// NM = MFMAs of layer 1 = ceil(in_dim / 4); lane group g supplies columns NM*g + m.
// HT = hidden tiles (hidden width 16*HT: 1 or 4), OT = output tiles (ceil(out_dim / 16): 1 or 4).  One wave owns its 16 rows
// through both layers: hidden unit 16*ht + 4g + r of L1's D tile ht is k-slot g of L2's MFMA (ht, r) -- no exchange.
template <int NM, int HT, int OT>
__global__ __launch_bounds__(256) void rows_kernel(const RowsArgs a) {
    constexpr int HID = 16 * HT;
    const int l = threadIdx.x & 63, g = l >> 4, j = l & 15, i = l & 15;
    // weights -> registers
    float w1[HT][NM], w2[OT][HT * 4];
    f4 b1r[HT], b2r[OT];
#pragma unroll
    for (int ht = 0; ht < HT; ++ht) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const int c = NM * g + m;
            w1[ht][m] = c < a.in_dim ? a.w1[(16 * ht + i) * a.in_dim + c] : 0.0f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) b1r[ht][r] = a.b1[16 * ht + 4 * g + r];
    }
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
        const int o = 16 * ot + i;
#pragma unroll
        for (int q = 0; q < HT * 4; ++q) w2[ot][q] = o < a.out_dim ? a.w2[o * HID + 16 * (q >> 2) + 4 * g + (q & 3)] : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) b2r[ot][r] = 16 * ot + 4 * g + r < a.out_dim ? a.b2[16 * ot + 4 * g + r] : 0.0f;
    }
    const long long tiles = (a.rows + 15) / 16;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    constexpr int VB = NM == 2 ? 8 : 16;   // bytes of one vector load
    const bool vec_in = (NM == 2 || NM % 4 == 0) && a.in_dim == 4 * NM && a.in_stride % (VB / 4) == 0 && (reinterpret_cast<uintptr_t>(a.in) % VB) == 0;
    const bool vec_out = a.out_dim % 4 == 0 && a.out_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0;
    for (long long t = wave; t < tiles; t += nwaves) {
        const long long row = t * 16 + j;
        const bool valid = row < a.rows;
        const float* src = a.in + (valid ? row : a.rows - 1) * a.in_stride + NM * g;
        float v[NM];
        if (vec_in) {
            if constexpr (NM % 4 == 0) {
#pragma unroll
                for (int c4 = 0; c4 < NM / 4; ++c4) {
                    const f4 q = reinterpret_cast<const f4*>(src)[c4];
                    v[4 * c4] = q[0]; v[4 * c4 + 1] = q[1]; v[4 * c4 + 2] = q[2]; v[4 * c4 + 3] = q[3];
                }
            } else if constexpr (NM == 2) {
                const float2 q = *reinterpret_cast<const float2*>(src);
                v[0] = q.x; v[1] = q.y;
            }
        } else {
#pragma unroll
            for (int m = 0; m < NM; ++m) v[m] = NM * g + m < a.in_dim ? src[m] : 0.0f;
        }
        f4 h[HT];
#pragma unroll
        for (int ht = 0; ht < HT; ++ht) {
            f4 acc = b1r[ht];
#pragma unroll
            for (int m = 0; m < NM; ++m) acc = rmfma(w1[ht][m], v[m], acc);
            h[ht] = f4{elu_fast(acc[0]), elu_fast(acc[1]), elu_fast(acc[2]), elu_fast(acc[3])};
        }
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
            f4 oA = b2r[ot], oB = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ht = 0; ht < HT; ++ht) {
                oA = rmfma(w2[ot][4 * ht + 0], h[ht][0], oA); oB = rmfma(w2[ot][4 * ht + 1], h[ht][1], oB);
                oA = rmfma(w2[ot][4 * ht + 2], h[ht][2], oA); oB = rmfma(w2[ot][4 * ht + 3], h[ht][3], oB);
            }
            const f4 o = oA + oB;
            if (valid && 16 * ot + 4 * g < a.out_dim) {
                float* dst = a.out + row * a.out_stride + 16 * ot + 4 * g;
                if (vec_out) {
                    *reinterpret_cast<f4*>(dst) = o;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (16 * ot + 4 * g + r < a.out_dim) dst[r] = o[r];
                }
            }
        }
    }
}

template <int HT, int OT>
void launch_rows(int NM, dim3 grid, dim3 block, hipStream_t s, const RowsArgs& a) {
    switch (NM) {
        case 1: hipLaunchKernelGGL((rows_kernel<1, HT, OT>), grid, block, 0, s, a); break;
        case 2: hipLaunchKernelGGL((rows_kernel<2, HT, OT>), grid, block, 0, s, a); break;
        case 3: hipLaunchKernelGGL((rows_kernel<3, HT, OT>), grid, block, 0, s, a); break;
        case 4: hipLaunchKernelGGL((rows_kernel<4, HT, OT>), grid, block, 0, s, a); break;
        default: hipLaunchKernelGGL((rows_kernel<16, HT, OT>), grid, block, 0, s, a); break;   // in_dim = 64 (decoders at hidden 64)
    }
}

}  // namespace
}  // namespace psnode

using namespace psnode;

extern "C" int32_t psnode_mlp_rows_supported(const psnode_mlp_f32* m) {
    // encoders: in <= 16 -> H -> H ; decoders: H -> H -> out <= 16 ; H in {16, 64}.  (A 64-wide input is the decoder's case:
    // there the first layer is H -> H, i.e. in_dim == H.)
    if (!m || m->n_layers != 2 || m->in_dim < 1 || m->out_dim[1] < 1) return 0;
    const int H = m->out_dim[0];
    if (H != 16 && H != 64) return 0;
    const bool in_ok = m->in_dim <= 16 || (m->in_dim == 64 && H == 64), out_ok = m->out_dim[1] <= 16 || m->out_dim[1] == H;
    return in_ok && out_ok;
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
    const int NM = (m->in_dim + 3) / 4, H = m->out_dim[0], OT = (m->out_dim[1] + 15) / 16;
    if (H == 16) launch_rows<1, 1>(NM, grid, block, s, a);
    else if (OT == 1) launch_rows<4, 1>(NM, grid, block, s, a);
    else launch_rows<4, 4>(NM, grid, block, s, a);
    return hipGetLastError() == hipSuccess ? PSNODE_OK : PSNODE_ERR_HIP;
}
