// K10 -- tall-skinny contraction over rows on MFMA (round 6):
//     C[m][n] = sum_r A[r][m] * B[r][n],     colsum[m] = sum_r A[r][m]        rows ~ 10^6..10^7, M, N <= 128
// This is every weight gradient that is NOT accumulated inside a sweep kernel: K9w's dW2 / dW1 blocks over its stored rows (the direct_encode
// models at hidden widths other than 16 / 64 -- the scripts' argparse default --hidden 128, neural_00_ODE_02_direct_encode.py:160-162 --
// fused/latent.py contracted them with torch.bmm over 256 row groups: a 32x32x256-macro-tile library GEMM at 4.2 ms per 134 GFLOP
// contraction) and the parameter gradients of the wide row MLPs.  Deterministic: per-workgroup partials, summed in a fixed order.
//
// v_mfma_f32_16x16x4_f32 with the ROWS as the contraction index: one MFMA adds 4 rows to a 16 x 16 tile of C.  Lane (k = l >> 4, j = l & 15)
// loads ONE float4 per 64 columns of a row: A[row 4s + k][64 q + 4 j .. + 3] -- a row's 64 columns are 256 contiguous bytes over the 16
// lanes of a group.  The four components of that register ARE four A (or B) operands: tile (q, c) holds the columns 64 q + 4 i + c, i < 16
// (a 16-element column set with stride 4 -- any partition of the columns into sets of 16 is a valid tiling; this one needs no shuffle).
// A workgroup = 4 waves over the same rows: every wave loads all of A (<= 2 float4) and its own quarter of B's tiles, and owns the
// (<= 8) x (<= 2) tiles of C between them: <= 16 accumulators of 4 registers.
#include "psnode_common.h"

namespace psnode {
namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

struct GemmTnDev {
    const float *A, *B;
    long long rows, lda, ldb, rows_per_wg;
    int M, N;
    float* part;            // [workgroups][M * N + M]
};

__device__ __forceinline__ f4 tn_mfma(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// MQ = ceil(M / 64), NQ = ceil(N / 64) in {1, 2}.  B tiles of wave w: NQ == 1: (q 0, component w); NQ == 2: q = w >> 1, components 2 (w & 1), + 1.
template <int MQ, int NQ>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const GemmTnDev a) {
    const int l = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int k = l >> 4, j = l & 15;
    const long long r0 = (long long)blockIdx.x * a.rows_per_wg;
    const long long r1 = r0 + a.rows_per_wg < a.rows ? r0 + a.rows_per_wg : a.rows;
    const int qb = NQ == 2 ? (w >> 1) : 0, cb0 = NQ == 2 ? 2 * (w & 1) : w;
    // column quads this lane loads; a quad beyond the matrix is read from column 0 and zeroed (branch-free)
    bool a_on[MQ];
    unsigned a_off[MQ];
#pragma unroll
    for (int q = 0; q < MQ; ++q) { a_on[q] = 64 * q + 4 * j < a.M; a_off[q] = a_on[q] ? 64 * q + 4 * j : 0; }
    const bool b_on = 64 * qb + 4 * j < a.N;
    const unsigned b_off = b_on ? 64 * qb + 4 * j : 0;
    const f4 zero4 = f4{0.f, 0.f, 0.f, 0.f};
    f4 acc[4 * MQ][NQ];
#pragma unroll
    for (int t = 0; t < 4 * MQ; ++t)
#pragma unroll
        for (int u = 0; u < NQ; ++u) acc[t][u] = zero4;
    f4 csum[MQ];
#pragma unroll
    for (int q = 0; q < MQ; ++q) csum[q] = zero4;

    // Column quads beyond the matrix are read from column 0 and NOT zeroed: C[m][n] depends on column m of A and column n of B only, and
    // the entries of invalid columns are never stored.  Rows: the main loop takes full quads of rows without any predicate (the first
    // version selected zeros behind every load: the selects sat right behind the loads, the wave waited out each round trip on the spot
    // and the look-ahead was gone -- found in the ISA); the last, partial quad is one masked step.
    // (B: exactly the wave's NQ components of its quad -- 4 or 8 bytes at b_off + cb0 -- so that no register is indexed at run time)
    const int nrows = (int)(r1 > r0 ? r1 - r0 : 0), nfull = nrows >> 2, rem = nrows & 3;
    const float* pa = a.A + (r0 + k) * a.lda;
    const float* pb = a.B + (r0 + k) * a.ldb + b_off + cb0;
    const long long sa = 4 * a.lda, sb = 4 * a.ldb;
    auto load = [&](f4 (&av)[MQ], float (&bv)[NQ]) {
#pragma unroll
        for (int q = 0; q < MQ; ++q) av[q] = *reinterpret_cast<const f4*>(pa + a_off[q]);
        if constexpr (NQ == 2) {
            const float2 v = *reinterpret_cast<const float2*>(pb);
            bv[0] = v.x; bv[1] = v.y;
        } else {
            bv[0] = *pb;
        }
        pa += sa; pb += sb;
    };
    auto mfmas = [&](const f4 (&av)[MQ], const float (&bv)[NQ]) {
#pragma unroll
        for (int q = 0; q < MQ; ++q) {
            csum[q] += av[q];                           // (every wave: 2 packed adds next to 16 MFMAs; wave 0 writes them)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int u = 0; u < NQ; ++u) acc[4 * q + c][u] = tn_mfma(av[q][c], bv[u], acc[4 * q + c][u]);
        }
    };
    f4 av[MQ], an[MQ];
    float bv[NQ], bn[NQ];
#pragma unroll
    for (int q = 0; q < MQ; ++q) { av[q] = zero4; an[q] = zero4; }
#pragma unroll
    for (int u = 0; u < NQ; ++u) { bv[u] = 0.0f; bn[u] = 0.0f; }
    if (nfull > 0) {
        // two register sets, the loop written out twice: a set is requested a whole 16-MFMA batch before it is used and never copied
        load(av, bv);
        int s = 0;
        // (sched_barrier: left to the scheduler, the first consumers of a freshly requested set -- its column-sum adds -- are hoisted to right
        //  behind the loads and the wave waits out the round trip there)
        for (; s + 2 < nfull; s += 2) {
            load(an, bn);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(av, bv);
            __builtin_amdgcn_sched_barrier(0);
            load(av, bv);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(an, bn);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (s + 1 < nfull) {                            // two steps left: the set in hand and one more
            load(an, bn);
            mfmas(av, bv);
            mfmas(an, bn);
        } else {
            mfmas(av, bv);                              // one step left
        }
    }
    if (rem > 0) {                                      // the last 1..3 rows: lane groups k >= rem contribute zeros
        const bool on = k < rem;
        const long long row = r0 + 4ll * nfull + (on ? k : 0);
        const float* qa = a.A + row * a.lda;
        const float* qb_ = a.B + row * a.ldb + b_off + cb0;
#pragma unroll
        for (int q = 0; q < MQ; ++q) { const f4 v = *reinterpret_cast<const f4*>(qa + a_off[q]); av[q] = on ? v : zero4; }
#pragma unroll
        for (int u = 0; u < NQ; ++u) { const float v = qb_[u]; bv[u] = on ? v : 0.0f; }
        mfmas(av, bv);
    }
    // tile (q, c) x (qb, cb): lane (g = k, j), register r  ->  C[64 q + 4 (4 g + r) + c][64 qb + 4 j + cb]
    float* wp = a.part + (size_t)blockIdx.x * ((size_t)a.M * a.N + a.M);
#pragma unroll
    for (int q = 0; q < MQ; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                const int n = 64 * qb + 4 * j + (NQ == 2 ? u + cb0 : cb0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = 64 * q + 4 * (4 * k + r) + c;
                    if (m < a.M && n < a.N) wp[(size_t)m * a.N + n] = acc[4 * q + c][u][r];
                }
            }
    if (w == 0) {       // column sums of A: over the four row slots (lane groups), then out from group 0
#pragma unroll
        for (int q = 0; q < MQ; ++q) {
            f4 v = csum[q];
#pragma unroll
            for (int c = 0; c < 4; ++c) { v[c] += __shfl_xor(v[c], 16, 64); v[c] += __shfl_xor(v[c], 32, 64); }
            if (k == 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c) { const int m = 64 * q + 4 * j + c; if (m < a.M) wp[(size_t)a.M * a.N + m] = v[c]; }
            }
        }
    }
}

long long tn_workgroups(long long rows) {
    // ~4 workgroups per CU, at least 256 rows each (a partial vector is up to 64 KB: a few hundred of them stay cheap to reduce)
    long long n = (rows + 255) / 256;
    return n < 1 ? 1 : (n > 1024 ? 1024 : n);
}

}  // namespace
}  // namespace psnode

using namespace psnode;

extern "C" int32_t psnode_gemm_tn_supported(const psnode_gemm_tn_args_f32* a) {
    if (!a || a->rows < 0 || a->M < 1 || a->N < 1 || a->M > 128 || a->N > 128) return 0;
    if ((a->M & 3) || (a->N & 3) || a->lda < a->M || a->ldb < a->N || (a->lda & 3) || (a->ldb & 3)) return 0;     // float4 row segments
    if (a->A && ((reinterpret_cast<uintptr_t>(a->A) & 15) || (reinterpret_cast<uintptr_t>(a->B) & 15))) return 0;
    return 1;
}

extern "C" size_t psnode_gemm_tn_workspace_bytes(const psnode_gemm_tn_args_f32* a) {
    if (!psnode_gemm_tn_supported(a)) return 0;
    return ((size_t)tn_workgroups(a->rows) * ((size_t)a->M * a->N + a->M) + 256) * sizeof(float);
}

extern "C" int32_t psnode_gemm_tn_f32(const psnode_gemm_tn_args_f32* a, void* workspace, size_t workspace_bytes, void* stream) {
    if (!a) return PSNODE_ERR_NULL;
    if (!psnode_gemm_tn_supported(a)) return PSNODE_ERR_UNSUPPORTED;
    if (!a->A || !a->B || !a->C) return PSNODE_ERR_NULL;
    if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 255u) || workspace_bytes < psnode_gemm_tn_workspace_bytes(a)) return PSNODE_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long nwg = tn_workgroups(a->rows);
    GemmTnDev d;
    d.A = a->A; d.B = a->B; d.rows = a->rows; d.lda = a->lda; d.ldb = a->ldb; d.M = a->M; d.N = a->N;
    d.rows_per_wg = ((a->rows + nwg - 1) / nwg + 3) / 4 * 4;
    d.part = static_cast<float*>(workspace);
    const dim3 grid((unsigned)nwg), block(256);
    const int mq = (a->M + 63) / 64, nq = (a->N + 63) / 64;
    if (mq == 1 && nq == 1) hipLaunchKernelGGL((gemm_tn_kernel<1, 1>), grid, block, 0, s, d);
    else if (mq == 1) hipLaunchKernelGGL((gemm_tn_kernel<1, 2>), grid, block, 0, s, d);
    else if (nq == 1) hipLaunchKernelGGL((gemm_tn_kernel<2, 1>), grid, block, 0, s, d);
    else hipLaunchKernelGGL((gemm_tn_kernel<2, 2>), grid, block, 0, s, d);
    if (hipGetLastError() != hipSuccess) return PSNODE_ERR_HIP;
    // the column sums ride behind C in every partial vector; a caller that does not want them gets them into the tail of the workspace
    float* cs = a->colsum_a ? a->colsum_a : d.part + (size_t)nwg * ((size_t)a->M * a->N + a->M);
    return launch_reduce_partials(d.part, a->C, cs, a->M * a->N, a->M, (int)nwg, s) == hipSuccess ? PSNODE_OK : PSNODE_ERR_HIP;
}
