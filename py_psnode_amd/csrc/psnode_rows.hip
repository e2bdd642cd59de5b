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
    unsigned in_inner;          // 0: row r at in + r * in_stride; else at in + (r / in_inner) * in_outer + (r % in_inner) * in_stride
    long long in_outer;
};

// Offset of input row r: flat, or two-level -- the [B,T,D] batch of the DataLoader read as the TIME-MAJOR rows r = t * B + b
// (x.permute(1, 0, 2) of neural_00_ODE_02_direct_encode.py:76 without materialising it: outer = t, stride D; inner = b, stride T * D).
__device__ __forceinline__ long long row_offset(const long long r, const long long stride, const unsigned inner, const long long outer) {
    if (inner == 0) return r * stride;
    const unsigned o = (unsigned)r / inner;
    return (long long)o * outer + (long long)((unsigned)r - o * inner) * stride;
}
// Which rows a 16-row tile holds.  Flat input: rows 16 t .. 16 t + 15.  Two-level input whose OUTER index is the contiguous one (the time-major
// view of a [B,T,D] batch: outer = t, pitch D; inner = b, pitch T D): a tile is 4 outer x 4 inner indices -- 4 grid points of 4 trajectories:
// the input comes in runs of 4 D floats and the output / gradient rows in runs of 4 rows -- instead of 16 trajectories of one grid point
// (16 reads of D floats, T D floats apart: every 32-byte sector from its own DRAM page).
struct TileWalk {
    long long rows, stride, outer_stride;
    unsigned inner, outer_n, tiles_o, tiles_i;
    bool square;
    long long tiles;
    __device__ __forceinline__ TileWalk(long long rows_, long long stride_, unsigned inner_, long long outer_stride_)
        : rows(rows_), stride(stride_), outer_stride(outer_stride_), inner(inner_) {
        square = inner > 0 && outer_stride < stride;
        outer_n = inner > 0 ? (unsigned)(rows / inner) : 0u;
        tiles_o = (outer_n + 3u) / 4u;
        tiles_i = (inner + 3u) / 4u;
        tiles = square ? (long long)tiles_i * tiles_o : (rows + 15) / 16;
    }
    // row index (clamped into the set when the tile runs over an edge), validity and input offset of tile row j
    __device__ __forceinline__ void at(const long long t, const int j, long long& row, bool& valid, long long& in_off) const {
        if (square) {
            const unsigned ti = (unsigned)t / tiles_o, to = (unsigned)t - ti * tiles_o;       // consecutive tiles: consecutive outer blocks of one inner block
            const unsigned i = 4u * ti + ((unsigned)j & 3u), o = 4u * to + ((unsigned)j >> 2);
            valid = i < inner && o < outer_n;
            const unsigned ic = i < inner ? i : inner - 1u, oc = o < outer_n ? o : outer_n - 1u;
            row = (long long)oc * inner + ic;
            in_off = (long long)oc * outer_stride + (long long)ic * stride;
        } else {
            const long long r = t * 16 + j;
            valid = r < rows;
            row = valid ? r : rows - 1;
            in_off = row_offset(row, stride, inner, outer_stride);
        }
    }
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
    const TileWalk walk(a.rows, a.in_stride, a.in_inner, a.in_outer);
    const long long tiles = walk.tiles;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    constexpr int VB = NM == 2 ? 8 : 16;   // bytes of one vector load
    const bool vec_in = (NM == 2 || NM % 4 == 0) && a.in_dim == 4 * NM && a.in_stride % (VB / 4) == 0 && a.in_outer % (VB / 4) == 0 && (reinterpret_cast<uintptr_t>(a.in) % VB) == 0;
    const bool vec_out = a.out_dim % 4 == 0 && a.out_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0;
    for (long long t = wave; t < tiles; t += nwaves) {
        long long row, in_off;
        bool valid;
        walk.at(t, j, row, valid, in_off);
        const float* src = a.in + in_off + NM * g;
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
            h[ht] = elu_quad(acc);
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

// ---------------------------------------------------------------------------------------------------------------
// K3b backward: d in, dW1, db1, dW2, db2 of the same row MLP, one pass over (in, grad_out).  Training the direct_encode
// models runs the encoders/decoders over all B*T rows under autograd (neural_00_ODE_02_direct_encode.py:74-88,267-275):
// skinny GEMM pairs with M = 4.1 M that rocBLAS handles poorly (28 of the 34 ms of an ODE_02 training step).  Same
// stream structure as the forward: one wave per 16-row tile, h recomputed (L2's forward is not needed), delta1 =
// (W2^T g) * ELU'(pre1) and d in = W1^T delta1 on MFMA with register-resident transposed weights; the weight gradients
// contract over the tile's 16 rows (operands transposed through private padded LDS tiles) and accumulate in registers
// over all of the wave's tiles; per-wave partials are summed by a second kernel in a fixed order (deterministic).
struct RowsBwdArgs {
    const float *w1, *b1, *w2, *in, *gout;
    float *gin, *wpart;
    long long rows, in_stride, gout_stride, gin_stride;
    int in_dim, out_dim, np;
    unsigned in_inner;          // as RowsArgs
    long long in_outer;
};

constexpr int RSCR = 64 * 4 + 4 * 8;   // padded transpose tile (floats)

// WLDS (round 4; the decoders at hidden 64: in = hidden = 64): the two 64 x 64 operand sets of the first layer -- W1 for the recomputed hidden
// layer, W1^T for d in -- live in LDS (8 KB per workgroup, every wave reads the values its lanes would have held: lane-linear, conflict-free)
// instead of 128 registers per lane.  373 registers meant ONE wave per SIMD for a kernel whose tile is 224 MFMAs behind 13 in-wave
// transposes and 64 ELUs; with 245 two waves share a SIMD (__launch_bounds__(256, 2)).
template <int NM, int HT, int OT>
__global__ __launch_bounds__(256, ((NM == 16 && HT == 4 && OT == 1) || (NM <= 4 && HT == 4 && OT == 4)) ? 2 : 1) void rows_bwd_kernel(const RowsBwdArgs a) {
    constexpr int HID = 16 * HT;
    constexpr int IT = NM > 4 ? NM / 4 : 1;      // input column tiles
    constexpr int NC = NM > 4 ? 4 : NM;          // columns per lane per tile
    constexpr bool WLDS = NM == 16 && HT == 4 && OT == 1;
    __shared__ __attribute__((aligned(16))) float scr_all[4][2][RSCR];
    __shared__ __attribute__((aligned(16))) f4 lw1[WLDS ? HT * (NM / 4) * 64 : 1], lw1t[WLDS ? IT * HT * 64 : 1];
    // ... and the encoders at hidden 64 (in <= 16, out = hidden = 64): W2^T (64 registers of 313) in LDS, 249 registers, two waves per SIMD
    constexpr bool W2LDS = NM <= 4 && HT == 4 && OT == 4;
    __shared__ __attribute__((aligned(16))) f4 lw2t[W2LDS ? HT * OT * 64 : 1];
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6, g = l >> 4, j = l & 15, i = l & 15;
    float* scrA = scr_all[wv][0];
    float* scrB = scr_all[wv][1];
    // column of the input a tile row (gr, rr) of tile q stands for, or -1
    auto col_of = [&](const int gr, const int rr, const int q) -> int {
        const int c = NM > 4 ? NM * gr + 4 * q + rr : (rr < NC ? NM * gr + rr : -1);
        return (c >= 0 && c < a.in_dim) ? c : -1;
    };
    // ---- weights -> registers
    float w1[WLDS ? 1 : HT][WLDS ? 1 : NM], w2t[W2LDS ? 1 : HT][W2LDS ? 1 : OT * 4], w1t[WLDS ? 1 : IT][WLDS ? 1 : HT * 4];
    f4 b1r[HT];
#pragma unroll
    for (int ht = 0; ht < HT; ++ht) {
        if constexpr (WLDS) {
#pragma unroll
            for (int m4 = 0; m4 < NM / 4; ++m4) {      // (every wave writes the same values: the weights depend on the lane only)
                f4 q;
#pragma unroll
                for (int r = 0; r < 4; ++r) { const int c = NM * g + 4 * m4 + r; q[r] = c < a.in_dim ? a.w1[(16 * ht + i) * a.in_dim + c] : 0.0f; }
                lw1[(ht * (NM / 4) + m4) * 64 + l] = q;
            }
        } else {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const int c = NM * g + m;
            w1[ht][m] = c < a.in_dim ? a.w1[(16 * ht + i) * a.in_dim + c] : 0.0f;
        }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) b1r[ht][r] = a.b1[16 * ht + 4 * g + r];
        if constexpr (W2LDS) {
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) {
                f4 q_;
#pragma unroll
                for (int r = 0; r < 4; ++r) { const int o = 16 * ot + 4 * g + r; q_[r] = o < a.out_dim ? a.w2[o * HID + 16 * ht + i] : 0.0f; }
                lw2t[(ht * OT + ot) * 64 + l] = q_;
            }
        } else {
#pragma unroll
        for (int q = 0; q < OT * 4; ++q) {            // delta = W2^T g: rows = units of tile ht, k-slot g <-> out dim 16*ot + 4g + r
            const int o = 16 * (q >> 2) + 4 * g + (q & 3);
            w2t[ht][q] = o < a.out_dim ? a.w2[o * HID + 16 * ht + i] : 0.0f;
        }
        }
    }
#pragma unroll
    for (int q = 0; q < IT; ++q) {
        const int c = col_of(i >> 2, i & 3, q);
        if constexpr (WLDS) {
#pragma unroll
            for (int ht = 0; ht < HT; ++ht) {
                f4 t_;
#pragma unroll
                for (int r = 0; r < 4; ++r) t_[r] = c >= 0 ? a.w1[(16 * ht + 4 * g + r) * a.in_dim + c] : 0.0f;
                lw1t[(q * HT + ht) * 64 + l] = t_;
            }
        } else {
#pragma unroll
        for (int k = 0; k < HT * 4; ++k)               // d in = W1^T delta1: rows = input columns, k-slot g <-> unit 16*ht + 4g + r
            w1t[q][k] = c >= 0 ? a.w1[(16 * (k >> 2) + 4 * g + (k & 3)) * a.in_dim + c] : 0.0f;
        }
    }
    if constexpr (WLDS || W2LDS) __syncthreads();
    f4 accW1[HT][IT], accW2[OT][HT], sb1[HT], sb2[OT];
#pragma unroll
    for (int ht = 0; ht < HT; ++ht) {
        sb1[ht] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < IT; ++q) accW1[ht][q] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) accW2[ot][ht] = f4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) sb2[ot] = f4{0.f, 0.f, 0.f, 0.f};

    // D-layout tile (rows 4g+r, col j) -> o[kk] = M[row i][col 4kk+g], through a private padded LDS tile
    auto transpose = [&](float* scr, const f4 v) -> f4 {
        *reinterpret_cast<f4*>(scr + 4 * l + 8 * g) = v;
        const float* s = scr + 4 * (16 * (i >> 2) + g) + 8 * (i >> 2) + (i & 3);
        return f4{s[0], s[16], s[32], s[48]};
    };

    const TileWalk walk(a.rows, a.in_stride, a.in_inner, a.in_outer);
    const long long tiles = walk.tiles;
    const long long wave = (long long)blockIdx.x * 4 + wv, nwaves = (long long)gridDim.x * 4;
    constexpr int VB = NM == 2 ? 8 : 16;
    const bool vec_in = (NM == 2 || NM % 4 == 0) && a.in_dim == 4 * NM && a.in_stride % (VB / 4) == 0 && a.in_outer % (VB / 4) == 0 && (reinterpret_cast<uintptr_t>(a.in) % VB) == 0;
    const bool vec_gin = a.gin && (NM == 2 || NM % 4 == 0) && a.in_dim == 4 * NM && a.gin_stride % (VB / 4) == 0 && (reinterpret_cast<uintptr_t>(a.gin) % VB) == 0;
    const bool vec_go = a.out_dim % 4 == 0 && a.gout_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(a.gout) & 15) == 0;
    for (long long t = wave; t < tiles; t += nwaves) {
        long long row, in_off;
        bool valid;
        walk.at(t, j, row, valid, in_off);
        const float* src = a.in + in_off + NM * g;
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
        f4 go[OT];
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
            go[ot] = f4{0.f, 0.f, 0.f, 0.f};
            if (valid && 16 * ot + 4 * g < a.out_dim) {
                const float* gs = a.gout + row * a.gout_stride + 16 * ot + 4 * g;
                if (vec_go) go[ot] = *reinterpret_cast<const f4*>(gs);
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (16 * ot + 4 * g + r < a.out_dim) go[ot][r] = gs[r];
                }
            }
            sb2[ot] += go[ot];
        }
        // hidden activations and delta1
        f4 h[HT], d1[HT];
#pragma unroll
        for (int ht = 0; ht < HT; ++ht) {
            f4 acc = b1r[ht];
            if constexpr (WLDS) {
#pragma unroll
                for (int m4 = 0; m4 < NM / 4; ++m4) {
                    const f4 wq = lw1[(ht * (NM / 4) + m4) * 64 + l];
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc = rmfma(wq[r], v[4 * m4 + r], acc);
                }
            } else {
#pragma unroll
            for (int m = 0; m < NM; ++m) acc = rmfma(w1[ht][m], v[m], acc);
            }
            h[ht] = elu_quad(acc);
            f4 tA = {0.f, 0.f, 0.f, 0.f}, tB = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) {
                if constexpr (W2LDS) {
                    const f4 wq = lw2t[(ht * OT + ot) * 64 + l];
                    tA = rmfma(wq[0], go[ot][0], tA); tB = rmfma(wq[1], go[ot][1], tB);
                    tA = rmfma(wq[2], go[ot][2], tA); tB = rmfma(wq[3], go[ot][3], tB);
                } else {
                tA = rmfma(w2t[ht][4 * ot + 0], go[ot][0], tA); tB = rmfma(w2t[ht][4 * ot + 1], go[ot][1], tB);
                tA = rmfma(w2t[ht][4 * ot + 2], go[ot][2], tA); tB = rmfma(w2t[ht][4 * ot + 3], go[ot][3], tB);
                }
            }
            const f4 tt = tA + tB;
#pragma unroll
            for (int r = 0; r < 4; ++r) d1[ht][r] = valid ? tt[r] * elu_grad(h[ht][r]) : 0.0f;
            sb1[ht] += d1[ht];
        }
        // d in
        if (a.gin) {
#pragma unroll
            for (int q = 0; q < IT; ++q) {
                f4 gA = {0.f, 0.f, 0.f, 0.f}, gB = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ht = 0; ht < HT; ++ht) {
                    if constexpr (WLDS) {
                        const f4 wt = lw1t[(q * HT + ht) * 64 + l];
                        gA = rmfma(wt[0], d1[ht][0], gA); gB = rmfma(wt[1], d1[ht][1], gB);
                        gA = rmfma(wt[2], d1[ht][2], gA); gB = rmfma(wt[3], d1[ht][3], gB);
                    } else {
                    gA = rmfma(w1t[q][4 * ht + 0], d1[ht][0], gA); gB = rmfma(w1t[q][4 * ht + 1], d1[ht][1], gB);
                    gA = rmfma(w1t[q][4 * ht + 2], d1[ht][2], gA); gB = rmfma(w1t[q][4 * ht + 3], d1[ht][3], gB);
                    }
                }
                const f4 gi = gA + gB;                      // lane (g, j): columns NM*g + 4q + r of row j
                if (valid) {
                    float* dst = a.gin + row * a.gin_stride + NM * g + 4 * q;
                    if (vec_gin && NC == 4) *reinterpret_cast<f4*>(dst) = gi;
                    else if (vec_gin && NM == 2) *reinterpret_cast<float2*>(dst) = float2{gi[0], gi[1]};
                    else {
#pragma unroll
                        for (int r = 0; r < NC; ++r) if (NM * g + 4 * q + r < a.in_dim) dst[r] = gi[r];
                    }
                }
            }
        }
        // weight gradients: contraction over the tile's 16 rows
        f4 hT[HT], vT[IT];
#pragma unroll
        for (int ht = 0; ht < HT; ++ht) hT[ht] = transpose((ht & 1) ? scrB : scrA, h[ht]);
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
            const f4 gT = transpose((ot & 1) ? scrA : scrB, go[ot]);
#pragma unroll
            for (int ht = 0; ht < HT; ++ht) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) accW2[ot][ht] = rmfma(gT[kk], hT[ht][kk], accW2[ot][ht]);
            }
        }
#pragma unroll
        for (int q = 0; q < IT; ++q) {
            f4 vt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < NC; ++r) vt[r] = valid ? v[4 * q + r] : 0.0f;
            vT[q] = transpose((q & 1) ? scrB : scrA, vt);
        }
#pragma unroll
        for (int ht = 0; ht < HT; ++ht) {
            const f4 dT = transpose((ht & 1) ? scrA : scrB, d1[ht]);
#pragma unroll
            for (int q = 0; q < IT; ++q) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) accW1[ht][q] = rmfma(dT[kk], vT[q][kk], accW1[ht][q]);
            }
        }
    }
    // ---- per-wave partial, nn.Linear order [W1 (H x in), b1, W2 (out x H), b2]
    float* wp = a.wpart + (size_t)wave * a.np;
    const int oB1 = HID * a.in_dim, oW2 = oB1 + HID, oB2 = oW2 + a.out_dim * HID;
#pragma unroll
    for (int ht = 0; ht < HT; ++ht) {
#pragma unroll
        for (int q = 0; q < IT; ++q) {
            const int c = col_of(j >> 2, j & 3, q);
#pragma unroll
            for (int r = 0; r < 4; ++r) if (c >= 0) wp[(16 * ht + 4 * g + r) * a.in_dim + c] = accW1[ht][q][r];
        }
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = 16 * ot + 4 * g + r;
                if (o < a.out_dim) wp[oW2 + o * HID + 16 * ht + j] = accW2[ot][ht][r];
            }
        }
    }
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int ht = 0; ht < HT; ++ht) sb1[ht][r] += __shfl_xor(sb1[ht][r], m, 64);
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) sb2[ot][r] += __shfl_xor(sb2[ot][r], m, 64);
        }
    }
    if (j == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int ht = 0; ht < HT; ++ht) wp[oB1 + 16 * ht + 4 * g + r] = sb1[ht][r];
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) if (16 * ot + 4 * g + r < a.out_dim) wp[oB2 + 16 * ot + 4 * g + r] = sb2[ot][r];
        }
    }
}

// Two-stage fixed-order sum of the per-wave partials: stage 1 sums slices of the partial sets (grid.y slices), stage 2 the slices.
constexpr int kRedSlices = 32;
__global__ void rows_reduce_stage1(const float* __restrict__ part, float* __restrict__ mid, int np, int nparts) {
    const int pidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (pidx >= np) return;
    const int per = (nparts + kRedSlices - 1) / kRedSlices, q0 = blockIdx.y * per, q1 = q0 + per < nparts ? q0 + per : nparts;
    // eight independent chains (one chain = `per` DEPENDENT memory round trips: 31 us for 72 partials per slice); the order stays fixed
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int q = q0;
    for (; q + 8 <= q1; q += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += part[(size_t)(q + j) * np + pidx];
    }
    for (int j = 0; q < q1; ++q, ++j) acc[j] += part[(size_t)q * np + pidx];
    mid[(size_t)blockIdx.y * np + pidx] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
}
__global__ void rows_reduce_stage2(const float* __restrict__ mid, float* __restrict__ out, int np) {
    const int pidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (pidx >= np) return;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < kRedSlices; ++q) acc[q & 7] += mid[(size_t)q * np + pidx];
    out[pidx] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
}

template <int HT, int OT>
void launch_rows_bwd(int NM, dim3 grid, dim3 block, hipStream_t s, const RowsBwdArgs& a) {
    switch (NM) {
        case 1: hipLaunchKernelGGL((rows_bwd_kernel<1, HT, OT>), grid, block, 0, s, a); break;
        case 2: hipLaunchKernelGGL((rows_bwd_kernel<2, HT, OT>), grid, block, 0, s, a); break;
        case 3: hipLaunchKernelGGL((rows_bwd_kernel<3, HT, OT>), grid, block, 0, s, a); break;
        case 4: hipLaunchKernelGGL((rows_bwd_kernel<4, HT, OT>), grid, block, 0, s, a); break;
        default: hipLaunchKernelGGL((rows_bwd_kernel<16, HT, OT>), grid, block, 0, s, a); break;
    }
}

int rows_np(const psnode_mlp_f32* m) { return m->out_dim[0] * (m->in_dim + 1) + m->out_dim[1] * (m->out_dim[0] + 1); }
long long rows_bwd_blocks(long long rows) {
    const long long tiles = (rows + 15) / 16;
    long long blocks = (tiles + 3) / 4;
    return blocks > 512 ? 512 : (blocks < 1 ? 1 : blocks);
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

extern "C" int32_t psnode_mlp_rows_f32(const psnode_mlp_f32* m, int64_t rows, const float* in, int64_t in_row_stride, int64_t in_inner_rows,
                                       int64_t in_outer_stride, float* out, int64_t out_row_stride, void* stream) {
    if (!m || !in || !out) return PSNODE_ERR_NULL;
    if (!psnode_mlp_rows_supported(m)) return PSNODE_ERR_UNSUPPORTED;
    if (!m->weight[0] || !m->weight[1] || !m->bias[0] || !m->bias[1]) return PSNODE_ERR_NULL;
    if (rows < 0 || in_row_stride < m->in_dim || out_row_stride < m->out_dim[1]) return PSNODE_ERR_DIMS;
    if (in_inner_rows < 0 || in_inner_rows > 0xffffffffll || (in_inner_rows > 0 && (rows > 0xffffffffll || in_outer_stride < 0))) return PSNODE_ERR_DIMS;
    if (rows == 0) return PSNODE_OK;
    RowsArgs a{m->weight[0], m->bias[0], m->weight[1], m->bias[1], in, out, rows, in_row_stride, out_row_stride, m->in_dim, m->out_dim[1],
               (unsigned)in_inner_rows, in_inner_rows > 0 ? in_outer_stride : 0};
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

extern "C" size_t psnode_mlp_rows_backward_workspace_bytes(const psnode_mlp_f32* m, int64_t rows) {
    if (!m || !psnode_mlp_rows_supported(m) || rows < 0) return 0;
    return ((size_t)rows_bwd_blocks(rows) * 4 + kRedSlices) * rows_np(m) * sizeof(float);
}

extern "C" int32_t psnode_mlp_rows_backward_f32(const psnode_mlp_f32* m, int64_t rows, const float* in, int64_t in_row_stride,
                                                int64_t in_inner_rows, int64_t in_outer_stride, const float* grad_out, int64_t gout_row_stride, float* grad_in, int64_t gin_row_stride,
                                                float* grad_params, void* workspace, size_t workspace_bytes, void* stream) {
    if (!m || !in || !grad_out) return PSNODE_ERR_NULL;      // grad_params == NULL: leave the per-wave partials in `workspace`, unreduced
    if (!psnode_mlp_rows_supported(m)) return PSNODE_ERR_UNSUPPORTED;
    if (!m->weight[0] || !m->weight[1] || !m->bias[0]) return PSNODE_ERR_NULL;
    if (rows < 0 || in_row_stride < m->in_dim || gout_row_stride < m->out_dim[1] || (grad_in && gin_row_stride < m->in_dim)) return PSNODE_ERR_DIMS;
    if (in_inner_rows < 0 || in_inner_rows > 0xffffffffll || (in_inner_rows > 0 && (rows > 0xffffffffll || in_outer_stride < 0))) return PSNODE_ERR_DIMS;
    const size_t need = grad_params ? psnode_mlp_rows_backward_workspace_bytes(m, rows)
                                    : (size_t)rows_bwd_blocks(rows) * 4 * rows_np(m) * sizeof(float);
    if (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15)) return PSNODE_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int np = rows_np(m);
    const long long blocks = rows_bwd_blocks(rows);
    RowsBwdArgs a{m->weight[0], m->bias[0], m->weight[1], in, grad_out, grad_in, static_cast<float*>(workspace), rows, in_row_stride,
                  gout_row_stride, gin_row_stride, m->in_dim, m->out_dim[1], np, (unsigned)in_inner_rows, in_inner_rows > 0 ? in_outer_stride : 0};
    const dim3 grid((unsigned)blocks), block(256);
    const int NM = (m->in_dim + 3) / 4, H = m->out_dim[0], OT = (m->out_dim[1] + 15) / 16;
    if (H == 16) launch_rows_bwd<1, 1>(NM, grid, block, s, a);
    else if (OT == 1) launch_rows_bwd<4, 1>(NM, grid, block, s, a);
    else launch_rows_bwd<4, 4>(NM, grid, block, s, a);
    if (hipGetLastError() != hipSuccess) return PSNODE_ERR_HIP;
    if (!grad_params) return PSNODE_OK;
    float* mid = static_cast<float*>(workspace) + (size_t)blocks * 4 * np;
    hipLaunchKernelGGL(rows_reduce_stage1, dim3((np + 255) / 256, kRedSlices), dim3(256), 0, s, static_cast<const float*>(workspace), mid, np,
                       (int)(blocks * 4));
    hipLaunchKernelGGL(rows_reduce_stage2, dim3((np + 255) / 256), dim3(256), 0, s, mid, grad_params, np);
    return hipGetLastError() == hipSuccess ? PSNODE_OK : PSNODE_ERR_HIP;
}

// One module applied to several row sets in a step (x_encoder over the grid rows AND the first row, z_encoder over grid rows, first row and
// jump rows: neural_00_ODE_02_direct_encode.py:76-82): every set's backward leaves its per-wave partials side by side in ONE buffer
// (psnode_mlp_rows_backward_f32 with grad_params == NULL), and one fixed-order reduction over all of them forms the module's gradient --
// instead of one two-launch reduction per set and an autograd `add` per parameter tensor and extra use.
extern "C" int64_t psnode_mlp_rows_backward_parts(const psnode_mlp_f32* m, int64_t rows) {
    if (!m || !psnode_mlp_rows_supported(m) || rows < 0) return 0;
    return rows_bwd_blocks(rows) * 4;
}
extern "C" size_t psnode_mlp_rows_reduce_workspace_bytes(const psnode_mlp_f32* m, int64_t n_parts) {
    if (!m || !psnode_mlp_rows_supported(m) || n_parts < 0) return 0;
    return ((size_t)n_parts + kRedSlices) * rows_np(m) * sizeof(float);
}
extern "C" int32_t psnode_mlp_rows_reduce_f32(const psnode_mlp_f32* m, int64_t n_parts, void* workspace, size_t workspace_bytes,
                                              float* grad_params, void* stream) {
    if (!m || !workspace || !grad_params) return PSNODE_ERR_NULL;
    if (!psnode_mlp_rows_supported(m)) return PSNODE_ERR_UNSUPPORTED;
    if (n_parts < 1 || n_parts > 0x7fffffffll) return PSNODE_ERR_DIMS;
    if (workspace_bytes < psnode_mlp_rows_reduce_workspace_bytes(m, n_parts) || (reinterpret_cast<uintptr_t>(workspace) & 15)) return PSNODE_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int np = rows_np(m);
    float* mid = static_cast<float*>(workspace) + (size_t)n_parts * np;
    hipLaunchKernelGGL(rows_reduce_stage1, dim3((np + 255) / 256, kRedSlices), dim3(256), 0, s, static_cast<const float*>(workspace), mid, np, (int)n_parts);
    hipLaunchKernelGGL(rows_reduce_stage2, dim3((np + 255) / 256), dim3(256), 0, s, mid, grad_params, np);
    return hipGetLastError() == hipSuccess ? PSNODE_OK : PSNODE_ERR_HIP;
}
