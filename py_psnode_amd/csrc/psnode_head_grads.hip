// K7h -- every contraction over the AE head's stored rows, in one launch (C ABI psnode_dae_head_grads_f32): what is left of
// loss.backward() through i_j = g(x_j; z_j, v_j) (my_solvers.py:95, :121, :108-110; neural_base.py AE_Func) once the sequential sweep
// K7f (psnode_dae_backward_fused.hip) has written the head's adjoint rows.  Replaces the library GEMMs, column sums and concatenations
// of round 2's host side (fused.dae_backward_wide:head_grads).  Per row (grid point or event) r and trajectory b, with h_l the head's ELU
// outputs (saved by the forward call or stored by K7f) and delta_l its layer adjoints:
//     dAW2 += delta2 (x) h1      dAW3 += delta3 (x) h2      dAW4[slot] += gi[slot] (x) h3      dAW1[:, u-columns] += delta1 (x) u
//     db_l += delta_l            sa1[b] += delta1  (per trajectory: the caller contracts it with all_initial)
//     gza[r, b] = AW1[:, z|v columns]^T delta1                                     (dL/dz, dL/dv through the head)
// Plan as K4f's weight gradients: one workgroup = 16 trajectories x all rows, NWV = H/16 waves, wave w owns units 16w..16w+15 and
// accumulates dAW_l[all units][own units] in registers for the whole launch; delta_l of every wave is published in padded LDS tiles
// that are read TRANSPOSED (the contraction runs over the tile's trajectories on MFMA); per-workgroup partials, summed in a fixed
// order by reduce_partials.  No weights are resident: ~110 registers, the loads of the next row are in flight during the MFMAs of this one.
#include <string.h>

#include "psnode_wide_pack.h"

namespace psnode {
namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 hm4(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

struct HeadDev {
    long long R, B;
    int hreal, nzv, NP;
    const float* h[3];
    long long h_rs;                 // floats between consecutive rows of h[l]
    const float* d[3];              // [R, B, H]
    const float *gi, *u;            // [R, B, 16]
    const float* aw1;               // AW1 [hreal, k1a]; columns zv0 .. zv0+nzv-1 multiply z | v
    int k1a, zv0;
    float* gza;                     // [R, B, 8] or null
    float* sa1;                     // [gridDim.y][B, H]: per row-split partial (summed in a fixed order afterwards when gridDim.y > 1)
    float* wpart;                   // [gridDim.y * gridDim.x][NP]
};

// rows are independent: at <= 4 waves (one wave per SIMD and workgroup, LDS and registers for three or four workgroups per CU) the rows are
// split over blockIdx.y so that a CU holds several workgroups whose exchanges overlap
__global__ void sum_splits_kernel(const float* __restrict__ part, float* __restrict__ out, const long long n, const int splits) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float acc = part[i];
    for (int q = 1; q < splits; ++q) acc += part[(long long)q * n + i];
    out[i] = acc;
}

constexpr int HTILE = 64 * 4 + 4 * 8;     // padded 16x16 tile (K4f: FTILE)

template <int NWV>
__global__ __launch_bounds__(64 * NWV) void head_grads_kernel(const HeadDev a) {
    constexpr int H = 16 * NWV;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int l = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = l >> 4, j = l & 15, i = j;
    // tiles: [parity][0: delta2 | 1: delta3 | 2: gza partials][wave], then one private transpose tile per wave
    auto tile = [&](const int par, const int which, const int wv) -> float* { return lds + ((par * 3 + which) * NWV + wv) * HTILE; };
    float* scr = lds + 6 * NWV * HTILE + w * HTILE;
    const long long b0 = (long long)blockIdx.x * TBM;
    const bool valid = b0 + j < a.B;
    const long long b = valid ? b0 + j : a.B - 1;
    const int HR = a.hreal, nzv = a.nzv;

    const int toff = 4 * l + 8 * g;
    const int roff = 72 * (i >> 2) + 4 * g + (i & 3);
    auto put = [&](float* t_, const f4 v) { *reinterpret_cast<f4*>(t_ + toff) = v; };
    auto getl = [&](const float* t_) -> f4 { return *reinterpret_cast<const f4*>(t_ + toff); };
    auto get_row = [&](const float* t_) -> f4 { const float* s_ = t_ + roff; return f4{s_[0], s_[16], s_[32], s_[48]}; };
    auto transpose = [&](const f4 v) -> f4 { put(scr, v); return get_row(scr); };

    // gza operand: row i <-> z|v column i, k-slot g: AW1[16w+4g+r][zv0 + i]
    float aeT[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int uu = 16 * w + 4 * g + r;
        aeT[r] = (i < nzv && uu < HR) ? a.aw1[(size_t)uu * a.k1a + a.zv0 + i] : 0.0f;
    }
    const bool want_gza = a.gza != nullptr && nzv > 0;

    const f4 zero4 = f4{0.f, 0.f, 0.f, 0.f};
    f4 acc2[NWV], acc3[NWV], accP3 = zero4, accP0 = zero4, S1 = zero4, S2 = zero4, S3 = zero4, SG = zero4;
#pragma unroll
    for (int c = 0; c < NWV; ++c) { acc2[c] = zero4; acc3[c] = zero4; }

    const unsigned offH = 4u * ((unsigned)(b * H) + 16 * w + 4 * g);
    const unsigned offS = 4u * ((unsigned)(b * 16) + 4 * g);       // this lane's four u columns 4g..4g+3
    const unsigned offG = 4u * ((unsigned)(b * 16) + g);           // slot 4m + g at + 16 m
    const long long drs = a.B * H;
    struct Row { f4 h1, h2, h3, d1, d2, d3, u4, gs; };
    auto load_row = [&](const long long r) -> Row {
        Row q;
        q.h1 = ldg<f4>(sbase(a.h[0] + r * a.h_rs), offH);
        q.h2 = ldg<f4>(sbase(a.h[1] + r * a.h_rs), offH);
        q.h3 = ldg<f4>(sbase(a.h[2] + r * a.h_rs), offH);
        q.d1 = ldg<f4>(sbase(a.d[0] + r * drs), offH);
        q.d2 = ldg<f4>(sbase(a.d[1] + r * drs), offH);
        q.d3 = ldg<f4>(sbase(a.d[2] + r * drs), offH);
        q.u4 = ldg<f4>(sbase(a.u + r * a.B * 16), offS);
        const gptr<const float> gr = sbase(a.gi + r * a.B * 16);
#pragma unroll
        for (int m = 0; m < 4; ++m) q.gs[m] = ldg<float>(gr, offG + 16u * m);
        return q;
    };
    int p = 0;
    const long long rlo = a.R * blockIdx.y / gridDim.y, rhi = a.R * (blockIdx.y + 1) / gridDim.y;      // this workgroup's rows
    Row nx = load_row(rlo < a.R ? rlo : a.R - 1);
    for (long long r = rlo; r < rhi; ++r) {
        __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0): the row requested a whole iteration ago
        Row q = nx;
        nx = load_row(r + 1 < rhi ? r + 1 : r);       // unconditional (clamped): no phis behind the loads
        if (!valid) { q.d1 = zero4; q.d2 = zero4; q.d3 = zero4; q.gs = zero4; }     // padding trajectories duplicate the last one
        put(tile(p, 0, w), q.d2);
        put(tile(p, 1, w), q.d3);
        if (want_gza) {
            f4 pa = hm4(aeT[0], q.d1[0], zero4), pb = hm4(aeT[1], q.d1[1], zero4);
            pa = hm4(aeT[2], q.d1[2], pa); pb = hm4(aeT[3], q.d1[3], pb);
            put(tile(p, 2, w), pa + pb);
        }
        S1 += q.d1; S2 += q.d2; S3 += q.d3; SG += q.gs;
        {   // the two skinny products: dAW4[slot (g, r) = 4r + g][own unit], dAW1[own unit (g, r)][u column]
            const f4 gT = transpose(q.gs);
            const f4 h3T = transpose(q.h3);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) accP3 = hm4(gT[kk], h3T[kk], accP3);
            const f4 dT = transpose(q.d1);
            const f4 uT = transpose(q.u4);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) accP0 = hm4(dT[kk], uT[kk], accP0);
        }
        const f4 h1T = transpose(q.h1);
        const f4 h2T = transpose(q.h2);
        lds_barrier();
#ifndef PSNODE_K7H_AHEAD
#define PSNODE_K7H_AHEAD 8      // transposed tiles read this many chunks ahead of the MFMAs that consume them (0: read where used; 0 / 2 / 4 / 8 at hidden 128: 4.67 / 4.66 / 4.58 / 4.25 ms, profiles/r03y_k7h_ahead_ab.txt)
#endif
        if constexpr (PSNODE_K7H_AHEAD > 0) {
            constexpr int G = PSNODE_K7H_AHEAD < NWV ? PSNODE_K7H_AHEAD : NWV;
#pragma unroll
            for (int c0 = 0; c0 < NWV; c0 += G) {
                f4 d2T[G], d3T[G];
#pragma unroll
                for (int q = 0; q < G; ++q) {
                    const int ws = (w + c0 + q) & (NWV - 1);
                    d2T[q] = get_row(tile(p, 0, ws));
                    d3T[q] = get_row(tile(p, 1, ws));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int q = 0; q < G; ++q) {      // 2 G independent accumulator chains
                        acc2[c0 + q] = hm4(d2T[q][kk], h1T[kk], acc2[c0 + q]);
                        acc3[c0 + q] = hm4(d3T[q][kk], h2T[kk], acc3[c0 + q]);
                    }
            }
        } else {
#pragma unroll
        for (int c = 0; c < NWV; ++c) {
            const int ws = (w + c) & (NWV - 1);
            const f4 d2T = get_row(tile(p, 0, ws));
            const f4 d3T = get_row(tile(p, 1, ws));
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) { acc2[c] = hm4(d2T[kk], h1T[kk], acc2[c]); acc3[c] = hm4(d3T[kk], h2T[kk], acc3[c]); }
        }
        }
        if (want_gza) {
            if (w == 0) {
                f4 out = zero4;
#pragma unroll
                for (int c = 0; c < NWV; ++c) out += getl(tile(p, 2, c));
                if (valid && g < 2) stg<f4>(sbase(a.gza + r * a.B * 8), 4u * ((unsigned)(b * 8) + 4 * g), out);
            }
        }
        p ^= 1;
    }

    // ---- epilogue: sa1 (per trajectory), per-workgroup partials [dAW2 | dAW3 | dAW4 slots (16 x HR) | dAW1 u-columns (HR x 16) | db1 db2 db3 | sum gi (16)]
    if (valid) *reinterpret_cast<f4*>(a.sa1 + ((size_t)blockIdx.y * a.B + b) * H + 16 * w + 4 * g) = S1;
    float* wp = a.wpart + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * a.NP;
    const int oW3 = HR * HR, oP3 = 2 * HR * HR, oP0 = oP3 + 16 * HR, oB = oP0 + HR * 16, oG = oB + 3 * HR;
    const int v = 16 * w + j;        // own column
#pragma unroll
    for (int c = 0; c < NWV; ++c) {
        const int ub = 16 * ((w + c) & (NWV - 1)) + 4 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (ub + r < HR && v < HR) {
                wp[(size_t)(ub + r) * HR + v] = acc2[c][r];
                wp[oW3 + (size_t)(ub + r) * HR + v] = acc3[c][r];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (v < HR) wp[oP3 + (size_t)(4 * r + g) * HR + v] = accP3[r];
        const int u = 16 * w + 4 * g + r;
        if (u < HR) wp[oP0 + (size_t)u * 16 + j] = accP0[r];
    }
    f4 sb1 = S1, sb2 = S2, sb3 = S3, sg = SG;
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sb1[r] += __shfl_xor(sb1[r], m, 64); sb2[r] += __shfl_xor(sb2[r], m, 64); sb3[r] += __shfl_xor(sb3[r], m, 64);
            sg[r] += __shfl_xor(sg[r], m, 64);
        }
    }
    if (j == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int u = 16 * w + 4 * g + r;
            if (u < HR) { wp[oB + u] = sb1[r]; wp[oB + HR + u] = sb2[r]; wp[oB + 2 * HR + u] = sb3[r]; }
            if (w == 0) wp[oG + 4 * r + g] = sg[r];
        }
    }
}

int head_np(int hr) { return 2 * hr * hr + 16 * hr + hr * 16 + 3 * hr + 16; }

}  // namespace
}  // namespace psnode

using namespace psnode;

extern "C" {

int32_t psnode_dae_head_grads_out_floats(int32_t hidden) { return hidden >= 1 && hidden <= 128 ? head_np(hidden) : 0; }

#ifndef PSNODE_K7H_SPLITS
#define PSNODE_K7H_SPLITS 4
#endif
static int head_splits(const psnode_dae_head_grads_args_f32* a) {
    const int nw = padded_hidden(a->hidden) / 16;
    if (nw > 4 || a->R < 64) return 1;
    return PSNODE_K7H_SPLITS;
}

size_t psnode_dae_head_grads_workspace_bytes(const psnode_dae_head_grads_args_f32* a) {
    if (!a || a->hidden < 1 || a->hidden > 128 || a->B < 1 || !padded_hidden(a->hidden)) return 0;
    const size_t nwg = (size_t)((a->B + TBM - 1) / TBM), rs = (size_t)head_splits(a);
    return (rs * nwg * head_np(a->hidden) + (rs > 1 ? rs * (size_t)a->B * padded_hidden(a->hidden) : 0) + 128) * sizeof(float);
}

int32_t psnode_dae_head_grads_f32(const psnode_dae_head_grads_args_f32* p, void* workspace, size_t workspace_bytes, void* stream) {
    if (!p) return PSNODE_ERR_NULL;
    if (p->hidden < 1 || p->hidden > 128 || p->B < 1 || p->R < 1 || p->n_zv < 0 || p->n_zv > 8) return PSNODE_ERR_DIMS;
    for (int l = 0; l < 3; ++l) if (!p->act[l] || !p->delta[l]) return PSNODE_ERR_NULL;
    if (!p->gi || !p->u || !p->out || !p->sa1 || (p->n_zv > 0 && p->grad_zv && !p->aw1)) return PSNODE_ERR_NULL;
    if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 255u) || workspace_bytes < psnode_dae_head_grads_workspace_bytes(p))
        return PSNODE_ERR_WORKSPACE;
    const int H = padded_hidden(p->hidden), nw = H / 16;
    if (!H) return PSNODE_ERR_UNSUPPORTED;
    if (p->B * (int64_t)H + 64 >= ((int64_t)1 << 30)) return PSNODE_ERR_DIMS;      // 32-bit per-lane byte offsets inside a row
    hipStream_t s = static_cast<hipStream_t>(stream);
    HeadDev a;
    memset(&a, 0, sizeof(a));
    a.R = p->R; a.B = p->B; a.hreal = p->hidden; a.nzv = p->n_zv; a.NP = head_np(p->hidden);
    for (int l = 0; l < 3; ++l) { a.h[l] = p->act[l]; a.d[l] = p->delta[l]; }
    a.h_rs = p->act_row_stride;
    a.gi = p->gi; a.u = p->u; a.aw1 = p->aw1; a.k1a = p->aw1_cols; a.zv0 = p->zv_col0;
    const size_t nwg = (size_t)((p->B + TBM - 1) / TBM);
    const int rs = head_splits(p);
    a.gza = p->grad_zv; a.wpart = static_cast<float*>(workspace);
    float* sa1_part = a.wpart + (((size_t)rs * nwg * a.NP + 63) / 64) * 64;
    a.sa1 = rs > 1 ? sa1_part : p->sa1;
    const dim3 grid((unsigned)nwg, (unsigned)rs), block(64 * nw);
    const size_t lds = (size_t)7 * nw * HTILE * sizeof(float);
    hipError_t e = hipSuccess;
    switch (nw) {
        case 2: hipLaunchKernelGGL(head_grads_kernel<2>, grid, block, lds, s, a); break;
        case 4: hipLaunchKernelGGL(head_grads_kernel<4>, grid, block, lds, s, a); break;
        default: {
            auto kern = &head_grads_kernel<8>;
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return PSNODE_ERR_HIP;
            hipLaunchKernelGGL(kern, grid, block, lds, s, a);
        }
    }
    if (hipGetLastError() != hipSuccess) return PSNODE_ERR_HIP;
    if (rs > 1) {
        const long long n = p->B * (long long)H;
        hipLaunchKernelGGL(sum_splits_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, sa1_part, p->sa1, n, rs);
        if (hipGetLastError() != hipSuccess) return PSNODE_ERR_HIP;
    }
    return launch_reduce_partials(a.wpart, p->out, nullptr, a.NP, 0, (int)(nwg * rs), s) == hipSuccess ? PSNODE_OK : PSNODE_ERR_HIP;
}

}  // extern "C"
