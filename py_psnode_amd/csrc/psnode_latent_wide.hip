// K3w -- latent-space integrator of the direct_encode variants at EVERY hidden_dim <= 128 that the dedicated kernels (K3f / K3a: 16,
// K3c: 64) do not take -- in particular the argparse default --hidden 128 of both direct_encode scripts
// (neural_00_ODE_02_direct_encode.py:160-162, neural_01_DAE_02_direct_encode.py:246-248), and 32 / 48 / 100 ... zero-padded:
//   ODE:  DE = Linear(6H,H) ELU Linear(H,H)                        state Xh[H], external Zh[H]
//   DAE:  DE = Linear(12H|9H,H) ELU Linear(H,H), AE = Linear(7H|5H,H) ELU Linear(H,H), blocks x | [z] | v | i of width H
// Before round 4 these widths ran on the generic kernel K0 (79 / 245 ms per 4096 x 1000 Euler batch at hidden 128).
//
// Decomposition as K3c: NWV = Hp / 16 waves (Hp = 64 or 128: the width H is padded to) per tile of 16 trajectories, wave w owns hidden
// units AND latent dims 16w..16w+15, every matrix is a set of Hp x Hp blocks, L1's `s - a0` and `s` column groups folded
// (F_b = Ws_b + Wd_b, A0_b = Wa0_b - Wd_b), the a0 group a per-trajectory constant, external blocks a per-step constant.
// What differs: at Hp = 128 the DAE's nine blocks are 576 KB -- no CU holds them -- so EVERY block is streamed from an L2-resident
// image ([block][wave][chunk][lane] f4, one coalesced 1 KB load per chunk and wave) through a ring of four chunk slots, exactly as the
// H->H layers of K1 / K2 at hidden 129..256 (psnode_mfma_impl.h: mid_glb); and the gathered vectors are not held in registers but read
// back from named LDS slots (x_k, stage input, hidden, i_k, AE hidden) or, for the external blocks, straight from the caller's rows.
// Rows are H floats wide (H % 4 == 0): the columns beyond H of a padded block are zero in the image and the loads of a chunk that lies
// beyond H are redirected to column 0 and masked.
#include <cstring>

#include "psnode_common.h"

namespace psnode {
namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 mfw(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f4 eluw(f4 v) { return elu_quad(v); }

// Image: blocks in the order  DE: F[nblk] | W2 | A0[nblk]   AE: W[nbe] | W2 | A0[nblk]   (DE first), each nw*nw*64 f4:
//   f4 component r of [blk][w][c][lane] = Blk[16w + i][16((w+c) % nw) + 4g + r]  (0 outside the real H x H);
// then the bias vectors in D layout, [vec][w][lane] f4 (component r = b[16w + 4g + r]): DE b1, DE b2, AE b1, AE b2.
struct PackW {
    int nw, H, nblk, nbe, dae;
    const float *dw1, *db1, *dw2, *db2, *aw1, *ab1, *aw2, *ab2;
    f4* out;
};
__host__ __device__ inline int packw_blocks(int nblk, int nbe, bool dae) { return (2 * nblk + 1) + (dae ? nbe + 1 + nblk : 0); }
__host__ __device__ inline size_t packw_f4(int nw, int nblk, int nbe, bool dae) {
    return (size_t)packw_blocks(nblk, nbe, dae) * nw * nw * 64 + (size_t)4 * nw * 64;
}

__global__ void packw_kernel(const PackW p) {
    const int nb = packw_blocks(p.nblk, p.nbe, p.dae), per = p.nw * p.nw * 64, n = p.nblk * p.H;
    const long long total = (long long)nb * per + 4 * p.nw * 64;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        f4 v = {0.f, 0.f, 0.f, 0.f};
        if (idx < (long long)nb * per) {
            const int blk = (int)(idx / per), rem = (int)(idx % per), lane = rem & 63, c = (rem >> 6) % p.nw, w = (rem >> 6) / p.nw;
            const int i = lane & 15, g = lane >> 4, u = 16 * w + i;
            const int nde = 2 * p.nblk + 1;
            const bool ae = blk >= nde;
            const int q = ae ? blk - nde : blk;                 // block index inside its MLP
            const int nfront = ae ? p.nbe : p.nblk;
            const int k1 = ae ? (n + p.nbe * p.H) : 3 * n;      // in_features
            const float* w1 = ae ? p.aw1 : p.dw1;
            const float* w2 = ae ? p.aw2 : p.dw2;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = 16 * ((w + c) % p.nw) + 4 * g + r;
                float x = 0.0f;
                if (u < p.H && col < p.H) {
                    const float* row = w1 + (size_t)u * k1;
                    if (q < nfront) x = ae ? row[n + p.H * q + col] : row[2 * n + p.H * q + col] + row[n + p.H * q + col];
                    else if (q == nfront) x = w2[(size_t)u * p.H + col];
                    else { const int b = q - nfront - 1; x = ae ? row[p.H * b + col] : row[p.H * b + col] - row[n + p.H * b + col]; }
                }
                v[r] = x;
            }
        } else {
            const int rem = (int)(idx - (long long)nb * per), lane = rem & 63, w = (rem >> 6) % p.nw, vec = (rem >> 6) / p.nw, g = lane >> 4;
            const float* src = vec == 0 ? p.db1 : (vec == 1 ? p.db2 : (vec == 2 ? p.ab1 : p.ab2));
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int u = 16 * w + 4 * g + r; v[r] = (src && u < p.H) ? src[u] : 0.0f; }
        }
        p.out[idx] = v;
    }
}

#ifndef PSNODE_K3W_DEPTH
#define PSNODE_K3W_DEPTH 4
#endif

// SAVE (training forward): what autograd would keep -- the DE's hidden layer and stage input per (step, stage) (a.sact / a.sxst
// [T-1,S,B,H]), the AE head's hidden layer per grid point (a.saeact [T,B,H]) and, for events taken, of the event-time head and its
// value i0 (a.sevact / a.sevi [nE,B,H]): the sweep kernel below reads them instead of recomputing.
template <int METHOD, int NBE, bool DAE, int NWV, bool SAVE = false>
__global__ __launch_bounds__(64 * NWV) void latent_wide_kernel(const IntegrateDev a, const f4* __restrict__ img) {
    constexpr int NBLK = 1 + NBE, NZV = DAE ? NBE - 1 : NBE, PER = NWV * NWV * 64;
    constexpr int NDE = 2 * NBLK + 1;
    // block ids in the image
    constexpr int B_F = 0, B_W2 = NBLK, B_A0 = NBLK + 1, B_AF = NDE, B_AW2 = NDE + NBE, B_AA0 = NDE + NBE + 1;
    constexpr int NBLOCKS = NDE + (DAE ? NBE + 1 + NBLK : 0);
    typedef f4 Vec[NWV][64];
    __shared__ Vec sX, sS, sH, sI, sAH;     // gathered-vector slots: x_k, stage input, DE hidden, i_k, AE hidden (own dims per wave)
    const int l = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = l >> 4, j = l & 15;
    const long long b0 = (long long)blockIdx.x * 16;
    const bool valid = b0 + j < a.B;
    const long long b = valid ? b0 + j : a.B - 1;
    const int H = a.xd, n = H * NBLK;
    const bool own_in = 16 * w + 4 * g < H;             // this lane's four own dims exist (H % 4 == 0)

    auto wrap = [&](const int x) -> int { return x >= NWV ? x - NWV : x; };
    // biases (D layout)
    const f4* bias = img + (size_t)NBLOCKS * PER;
    const f4 b1r = bias[(0 * NWV + w) * 64 + l], b2r = bias[(1 * NWV + w) * 64 + l];
    const f4 ab1r = DAE ? bias[(2 * NWV + w) * 64 + l] : f4{0.f, 0.f, 0.f, 0.f}, ab2r = DAE ? bias[(3 * NWV + w) * 64 + l] : f4{0.f, 0.f, 0.f, 0.f};

    // ---- one Hp x Hp block against a vector: A operands streamed (ring of WD chunk slots), B operands chunk by chunk from `src`
    //      src(c) -> f4 = the vector's dims 16((w+c)%NWV) + 4g + (0..3) of trajectory j
    auto mm_stream = [&](const int blk, auto&& src, f4& accA, f4& accB) {
        int loff = 4 * l;
        asm volatile("" : "+v"(loff));       // opaque lane offset: the weight loads must not be hoisted out of the time loop (K1 at 16 waves)
        const f4* wl = reinterpret_cast<const f4*>(reinterpret_cast<const float*>(img + ((size_t)blk * NWV + w) * NWV * 64) + loff);
        constexpr int WD = PSNODE_K3W_DEPTH < NWV ? PSNODE_K3W_DEPTH : NWV;
        f4 wq[WD];
#pragma unroll
        for (int c = 0; c < WD; ++c) wq[c] = wl[c * 64];
        f4 vn = src(0);
#pragma unroll
        for (int c = 0; c < NWV; ++c) {
            const f4 v = vn;
            if (c + 1 < NWV) vn = src(c + 1);
            const f4 wc = wq[c % WD];
            accA = mfw(wc[0], v[0], accA); accB = mfw(wc[1], v[1], accB);
            accA = mfw(wc[2], v[2], accA); accB = mfw(wc[3], v[3], accB);
            if (c + WD < NWV) wq[c % WD] = wl[(c + WD) * 64];
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto from_slot = [&](const Vec& s) { return [&s, w, l, &wrap](const int c) -> f4 { return s[wrap(w + c)][l]; }; };
    // a caller's row of H floats (chunk beyond H: redirected to column 0, masked)
    auto from_row = [&](const float* row) {
        return [row, w, g, H, &wrap](const int c) -> f4 {
            const int col = 16 * wrap(w + c) + 4 * g;
            const f4 v = *reinterpret_cast<const f4*>(row + (col < H ? col : 0));
            return col < H ? v : f4{0.f, 0.f, 0.f, 0.f};
        };
    };
    auto publish = [&](Vec& s, const f4 own) { s[w][l] = own; lds_barrier(); };

    // ---- per-trajectory constants: c0 = b1 + sum_blk A0_blk . a0_blk   (DE and AE)
    f4 c0A = b1r, c0B = {0.f, 0.f, 0.f, 0.f}, caA = ab1r, caB = c0B;
    for (int blk = 0; blk < NBLK; ++blk) {
        const float* a0row = a.a0 + b * n + (long long)H * blk;
        mm_stream(B_A0 + blk, from_row(a0row), c0A, c0B);
        if constexpr (DAE) mm_stream(B_AA0 + blk, from_row(a0row), caA, caB);
    }
    const f4 c0 = c0A + c0B, c0a = caA + caB;

    const long long tst = a.t.st, nT = a.T;
    const float* tp = a.t.p + b * a.t.sb;
    const bool has_z = a.zd > 0;
    const float* vbase = DAE ? a.v.p + b * a.v.sb : nullptr;
    const float* vjbase = DAE ? a.vj + b * a.vjb : nullptr;
    const float* sp[2] = {has_z ? a.z.p + b * a.z.sb : vbase, vbase};
    const long long sst[2] = {has_z ? a.z.st : a.v.st, a.v.st};
    const float* jp[2] = {has_z ? a.zj + b * a.zjb : vjbase, vjbase};
    const long long jse[2] = {has_z ? a.zje : a.vje, a.vje};
    auto ext_row = [&](const int s, const long long k, const int ev) -> const float* { return ev >= 0 ? jp[s] + ev * jse[s] : sp[s] + k * sst[s]; };

    const int ocol = 16 * w + 4 * g;
    f4 x = {0.f, 0.f, 0.f, 0.f};
    if (own_in) x = *reinterpret_cast<const f4*>((DAE ? a.x_init + b * H : a.x.p + b * a.x.sb) + ocol);
    auto store_own = [&](float* base, const long long k, const f4 v) {
        if (valid && own_in) *reinterpret_cast<f4*>(base + (k * a.B + b) * H + ocol) = v;
    };
    // SAVE: running row pointers of (step, stage) -- this lane's own four dims
    const long long srow = SAVE ? a.B * (long long)H : 0;
    float* sa_run = SAVE ? a.sact + b * H + ocol : nullptr;
    float* sx_run = SAVE ? a.sxst + b * H + ocol : nullptr;
    auto head_row = [&](float* base, const long long r) -> float* { return SAVE ? base + (r * a.B + b) * H + ocol : nullptr; };
    // the DE's second layer on the hidden vector in sH; the first layer's x block on the vector in `s` (whose own dims are xs_own)
    auto rhs_from = [&](const Vec& s, const f4 cz, const f4 xs_own) -> f4 {
        f4 accA = cz, accB = {0.f, 0.f, 0.f, 0.f};
        mm_stream(B_F, from_slot(s), accA, accB);
        const f4 h1 = eluw(accA + accB);
        if constexpr (SAVE) {
            if (valid && own_in) { *reinterpret_cast<f4*>(sx_run) = xs_own; *reinterpret_cast<f4*>(sa_run) = h1; }
            sx_run += srow; sa_run += srow;
        }
        publish(sH, h1);
        f4 oA = b2r, oB = {0.f, 0.f, 0.f, 0.f};
        mm_stream(B_W2, from_slot(sH), oA, oB);
        return oA + oB;
    };
    auto rhs = [&](const f4 xs_own, const f4 cz) -> f4 { publish(sS, xs_own); return rhs_from(sS, cz, xs_own); };
    // AE head on x in sX and the external rows of (k, ev); hrow (SAVE): where this lane's four units of the head's hidden layer go
    auto ae_eval = [&](const long long k, const int ev, float* hrow) -> f4 {
        f4 accA = c0a, accB = {0.f, 0.f, 0.f, 0.f};
        if constexpr (DAE) {
            mm_stream(B_AF, from_slot(sX), accA, accB);
#pragma unroll
            for (int s = 0; s < NZV; ++s) mm_stream(B_AF + 1 + s, from_row(ext_row(s, k, ev)), accA, accB);
            const f4 ah1 = eluw(accA + accB);
            if constexpr (SAVE) { if (valid && own_in) *reinterpret_cast<f4*>(hrow) = ah1; }
            publish(sAH, ah1);
            f4 oA = ab2r, oB = {0.f, 0.f, 0.f, 0.f};
            mm_stream(B_AW2, from_slot(sAH), oA, oB);
            return oA + oB;
        }
        return accA;
    };

    store_own(a.xo, 0, x);
    publish(sX, x);
    if constexpr (DAE) {
        const f4 ic = ae_eval(0, -1, head_row(a.saeact, 0));
        store_own(a.io, 0, ic);
        publish(sI, ic);
    }
    if (nT < 2) return;

    float t_cur = tp[0], t_nxt = tp[tst];
    int lane_zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));
    const bool has_ev = a.ev != nullptr;
    const int* evp = (has_ev ? a.ev : reinterpret_cast<const int*>(a.t.p)) + lane_zero;
    int ev_cur = has_ev ? a.ev[0] : -1;
    int ev_raw = evp[nT > 2 ? 1 : 0];

    for (long long k = 0; k + 1 < nT; ++k) {
        const float h_ = t_nxt - t_cur;
        t_cur = t_nxt;
        const int ev_now = __builtin_amdgcn_readfirstlane(ev_cur);
        const bool more = k + 2 < nT;
        t_nxt = tp[(more ? k + 2 : k + 1) * tst];
        ev_cur = (has_ev && more) ? ev_raw : -1;
        ev_raw = evp[k + 3 < nT ? k + 2 : 0];
        if constexpr (DAE) {
            if (ev_now >= 0) {       // i0 = g(x0; jumped z, v)  (my_solvers.py:108-110)
                const f4 ic = ae_eval(k, ev_now, head_row(a.sevact, ev_now));
                if constexpr (SAVE) { if (valid && own_in) *reinterpret_cast<f4*>(head_row(a.sevi, ev_now)) = ic; }
                publish(sI, ic);
            }
        }
        // per-step constant: c0 + sum over external blocks F_blk . block
        f4 czA = c0, czB = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NZV; ++s) mm_stream(B_F + 1 + s, from_row(ext_row(s, k, ev_now)), czA, czB);
        if constexpr (DAE) mm_stream(B_F + NBLK - 1, from_slot(sI), czA, czB);
        const f4 cz = czA + czB;

        const f4 k1 = rhs_from(sX, cz, x);
        if constexpr (METHOD == PSNODE_EULER) {
            x = x + h_ * k1;
        } else if constexpr (METHOD == PSNODE_MIDPOINT) {
            const f4 k2 = rhs(x + k1 * (0.5f * h_), cz);
            x = x + h_ * k2;
        } else {
            const f4 k2 = rhs(x + h_ * k1 * kOneThird, cz);
            const f4 k3 = rhs(x + h_ * (k2 - k1 * kOneThird), cz);
            const f4 k4 = rhs(x + h_ * (k1 - k2 + k3), cz);
            x = x + (k1 + 3.0f * (k2 + k3) + k4) * h_ * 0.125f;
        }
        store_own(a.xo, k + 1, x);
        publish(sX, x);
        if constexpr (DAE) {   // i1 = g(x1; z[k+1], v[k+1]) with the un-jumped rows (my_solvers.py:121)
            const f4 ic = ae_eval(k + 1, -1, head_row(a.saeact, k + 1));
            store_own(a.io, k + 1, ic);
            publish(sI, ic);
        }
    }
}

// =====================================================================================================================================
// K9w -- the adjoint sweep through the latent integrator at these widths (training: loss.backward() through integrate_ODE /
// integrate_DAE between the encoders and the decoders, neural_00_ODE_02_direct_encode.py:267-275 / neural_01_DAE_02_direct_encode.py:359-370
// over my_solvers.py:66-78 / :94-129).  The split form of round 2 (K4w): ONE sequential kernel carries the adjoint from the last step to
// the first with the TRANSPOSED blocks streamed exactly as the forward streams the blocks (a transposed block is just another Hp x Hp
// matrix in the forward format, so `gather, then multiply the own rows` needs no reduce-scatter), reads the activations the training
// forward saved, and stores the rows every parameter / input gradient is a plain contraction over:
//   per (step, stage):  gk  = adjoint of the stage's RHS value            [T-1,S,B,H]      dW2 = gk^T h1,  db2 = sum gk
//                       d1  = (W2^T gk) * ELU'(h1)                        [T-1,S,B,H]      dF_x = d1^T xstage
//   per step:           d1s = sum over the stages of d1                   [T-1,B,H]        dF_z|v|i = d1s^T ext,  d ext = d1s F_ext,  S1 = sum_t d1s
//   per grid point (DAE): gi = adjoint of i_k, da1 = (aw2^T gi) * ELU'(ah1)   [T,B,H] each    AE head: daw2 = gi^T ah1, dA_x = da1^T x_k, ...
//   per event taken (DAE): the same two rows of the event-time head       [nE,B,H]
// Those contractions are library GEMMs over millions of rows on the host side (fused.latent_backward_wide).
struct BwdWDev {
    int method, H, zd, dae;
    long long T, B;
    ViewDev t;
    const int* ev;
    const float *gxs, *gis;                 // [T,B,H] contiguous (gis may be null: zeros)
    const float *sact, *saeact, *sevact;    // saved by the forward: [T-1,S,B,H], [T,B,H], [nE,B,H]
    float *gk, *d1, *d1s;                   // outputs [T-1,S,B,H] x 2, [T-1,B,H]
    float *gi, *da1, *gi_ev, *da1_ev;       // DAE outputs [T,B,H] x 2, [nE,B,H] x 2 (event rows must be zero-initialised by the caller)
    float* gx0;                             // [B,H]: dL/dx_0 (the carried adjoint, the loss gradient of grid point 0 included)
};

// Transposed image: blocks  F_x^T | W2^T | (DAE:) F_i^T | A_x^T | aw2^T , each [wave][chunk][lane] f4 in the FORWARD format of M = Blk^T
struct PackWT {
    int nw, H, nblk, dae;
    const float *dw1, *dw2, *aw1, *aw2;
    f4* out;
};
__global__ void packwt_kernel(const PackWT p) {
    const int nb = p.dae ? 5 : 2, per = p.nw * p.nw * 64, n = p.nblk * p.H, nbe = p.nblk - 1;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < nb * per; idx += gridDim.x * blockDim.x) {
        const int blk = idx / per, rem = idx % per, lane = rem & 63, c = (rem >> 6) % p.nw, w = (rem >> 6) / p.nw;
        const int i = lane & 15, g = lane >> 4, u = 16 * w + i;       // row u of M = column u of Blk
        f4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int col = 16 * ((w + c) % p.nw) + 4 * g + r;        // column of M = row (hidden unit) of Blk
            if (u < p.H && col < p.H) {
                const float* drow = p.dw1 + (size_t)col * 3 * n;
                switch (blk) {
                    case 0: v[r] = drow[2 * n + u] + drow[n + u]; break;                                           // F_x[col][u]
                    case 1: v[r] = p.dw2[(size_t)col * p.H + u]; break;                                            // W2[col][u]
                    case 2: v[r] = drow[2 * n + p.H * (p.nblk - 1) + u] + drow[n + p.H * (p.nblk - 1) + u]; break; // F_i[col][u]
                    case 3: v[r] = p.aw1[(size_t)col * (n + nbe * p.H) + n + u]; break;                            // A_x[col][u]
                    default: v[r] = p.aw2[(size_t)col * p.H + u]; break;                                           // aw2[col][u]
                }
            }
        }
        p.out[idx] = v;
    }
}

template <int METHOD, bool DAE, int NWV>
__global__ __launch_bounds__(64 * NWV) void latent_wide_bwd_kernel(const BwdWDev a, const f4* __restrict__ img) {
    constexpr int S = rk_stages(METHOD);
    constexpr int T_FX = 0, T_W2 = 1, T_FI = 2, T_AX = 3, T_AW2 = 4;
    typedef f4 Vec[NWV][64];
    __shared__ Vec sA, sB, sC;              // gather slots: sA / sB alternate along every chain of two transposed layers, sC takes the lone
                                            // F_i^T product between two such chains (a slot is rewritten only after a barrier of another slot)
    const int l = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = l >> 4, j = l & 15;
    const long long b0 = (long long)blockIdx.x * 16;
    const bool valid = b0 + j < a.B;
    const long long b = valid ? b0 + j : a.B - 1;
    const int H = a.H;
    const bool own_in = 16 * w + 4 * g < H;
    const int ocol = 16 * w + 4 * g;
    auto wrap = [&](const int x) -> int { return x >= NWV ? x - NWV : x; };
    auto mm_stream = [&](const int blk, const Vec& s, f4& accA, f4& accB) {
        int loff = 4 * l;
        asm volatile("" : "+v"(loff));
        const f4* wl = reinterpret_cast<const f4*>(reinterpret_cast<const float*>(img + ((size_t)blk * NWV + w) * NWV * 64) + loff);
        constexpr int WD = PSNODE_K3W_DEPTH < NWV ? PSNODE_K3W_DEPTH : NWV;
        f4 wq[WD];
#pragma unroll
        for (int c = 0; c < WD; ++c) wq[c] = wl[c * 64];
        f4 vn = s[w][l];
#pragma unroll
        for (int c = 0; c < NWV; ++c) {
            const f4 v = vn;
            if (c + 1 < NWV) vn = s[wrap(w + c + 1)][l];
            const f4 wc = wq[c % WD];
            accA = mfw(wc[0], v[0], accA); accB = mfw(wc[1], v[1], accB);
            accA = mfw(wc[2], v[2], accA); accB = mfw(wc[3], v[3], accB);
            if (c + WD < NWV) wq[c % WD] = wl[(c + WD) * 64];
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // y_own = M . gather(own) with M = block `blk` of the transposed image
    auto apply = [&](Vec& slot, const int blk, const f4 own) -> f4 {
        slot[w][l] = own;
        lds_barrier();
        f4 accA = {0.f, 0.f, 0.f, 0.f}, accB = accA;
        mm_stream(blk, slot, accA, accB);
        return accA + accB;
    };
    const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const long long rowB = a.B * (long long)H;                          // floats per [B,H] slab
    const long long lane_off = b * H + ocol;
    auto ld = [&](const float* base, const long long slab) -> f4 {
        return own_in ? *reinterpret_cast<const f4*>(base + slab * rowB + lane_off) : zero4;
    };
    auto st = [&](float* base, const long long slab, const f4 v) {
        if (valid && own_in) *reinterpret_cast<f4*>(base + slab * rowB + lane_off) = v;
    };
    // AE head VJP: rows (gi, da1) of slab `slab` in (gi_out, da1_out), hidden activations from ah; returns A_x^T da1
    auto head_vjp = [&](const f4 gi, const float* ah, float* gi_out, float* da1_out, const long long slab) -> f4 {
        st(gi_out, slab, gi);
        const f4 dh = apply(sA, T_AW2, gi);
        const f4 da1 = dh * elu_grad_quad(ld(ah, slab));
        st(da1_out, slab, da1);
        return apply(sB, T_AX, da1);
    };

    const long long nT = a.T, tst = a.t.st;
    const float* tp = a.t.p + b * a.t.sb;
    f4 lam = ld(a.gxs, nT - 1);                                         // dL/dx_{k+1}, loss gradient included
    f4 gi = (DAE && a.gis) ? ld(a.gis, nT - 1) : zero4;                 // dL/di_{k+1}
    for (long long k = nT - 2; k >= 0; --k) {
        if constexpr (DAE) lam += head_vjp(gi, a.saeact, a.gi, a.da1, k + 1);      // i_{k+1} = g(x_{k+1}; z, v of grid point k+1)
        const float h_ = tp[(k + 1) * tst] - tp[k * tst];
        // DE stages backwards.  gxs[s] = dL/d(stage input s) = F_x^T d1_s
        f4 gx[S], d1sum = zero4, lam_in = lam;
#pragma unroll
        for (int s = S - 1; s >= 0; --s) {
            f4 gk = lam * (h_ * rk_b(METHOD, s));
#pragma unroll
            for (int s2 = s + 1; s2 < S; ++s2) gk += gx[s2] * (h_ * rk_a(METHOD, s2, s));
            const long long slab = k * S + s;
            st(a.gk, slab, gk);
            const f4 dh = apply(sA, T_W2, gk);
            const f4 d1 = dh * elu_grad_quad(ld(a.sact, slab));
            st(a.d1, slab, d1);
            d1sum += d1;
            gx[s] = apply(sB, T_FX, d1);
            lam_in += gx[s];
        }
        st(a.d1s, k, d1sum);
        lam = lam_in + ld(a.gxs, k);
        if constexpr (DAE) {
            f4 gi_de = apply(sC, T_FI, d1sum);                          // the DE of step k consumed i through its per-step constant
            // pin the sum of the two accumulator chains in front of the uniform branch below: scheduled between the last MFMA and the add
            // that reads its result, the taken edge of a branch carries no wait states (round 4's defect (b); isa_lint check B caught it here)
            asm volatile("" : "+v"(gi_de));
            const int ev = a.ev ? __builtin_amdgcn_readfirstlane(a.ev[k]) : -1;
            gi = a.gis ? ld(a.gis, k) : zero4;
            if (ev >= 0) {
                // a jump step evaluated its own head i0 = g(x_k; jumped z, v) (my_solvers.py:108-110): the DE's adjoint of i goes there,
                // the head of grid point k only sees the loss
                lam += head_vjp(gi_de, a.sevact, a.gi_ev, a.da1_ev, ev);
            } else {
                gi += gi_de;
            }
        }
    }
    if constexpr (DAE) lam += head_vjp(gi, a.saeact, a.gi, a.da1, 0);   // i_0 = g(x_0; z[0], v[0])  (my_solvers.py:95)
    if (valid && own_in) *reinterpret_cast<f4*>(a.gx0 + lane_off) = lam;
}

bool two_h(const MlpDev& m, int in_dim, int H) { return m.n_layers == 2 && m.in_dim == in_dim && m.out_dim[0] == H && m.out_dim[1] == H; }
bool al4w(const ViewDev& v) { return v.p && (reinterpret_cast<uintptr_t>(v.p) & 15) == 0 && v.st % 4 == 0 && v.sb % 4 == 0; }

template <int METHOD, int NWV>
hipError_t launchw_method(const IntegrateDev& a, bool dae, const f4* img, hipStream_t s) {
    const dim3 grid((unsigned)((a.B + 15) / 16)), block(64 * NWV);
    if (a.sact) {       // training forward
        if (!dae) hipLaunchKernelGGL((latent_wide_kernel<METHOD, 1, false, NWV, true>), grid, block, 0, s, a, img);
        else if (a.zd) hipLaunchKernelGGL((latent_wide_kernel<METHOD, 3, true, NWV, true>), grid, block, 0, s, a, img);
        else hipLaunchKernelGGL((latent_wide_kernel<METHOD, 2, true, NWV, true>), grid, block, 0, s, a, img);
        return hipGetLastError();
    }
    if (!dae) hipLaunchKernelGGL((latent_wide_kernel<METHOD, 1, false, NWV>), grid, block, 0, s, a, img);
    else if (a.zd) hipLaunchKernelGGL((latent_wide_kernel<METHOD, 3, true, NWV>), grid, block, 0, s, a, img);
    else hipLaunchKernelGGL((latent_wide_kernel<METHOD, 2, true, NWV>), grid, block, 0, s, a, img);
    return hipGetLastError();
}
template <int NWV>
hipError_t launchw_nw(const IntegrateDev& a, bool dae, const f4* img, hipStream_t s) {
    switch (a.method) {
        case PSNODE_EULER: return launchw_method<PSNODE_EULER, NWV>(a, dae, img, s);
        case PSNODE_MIDPOINT: return launchw_method<PSNODE_MIDPOINT, NWV>(a, dae, img, s);
        default: return launchw_method<PSNODE_RK4_38, NWV>(a, dae, img, s);
    }
}

}  // namespace

// the latent shapes at a hidden width H <= 128, H % 4 == 0 (16 and 64 have their own kernels and are asked first)
bool latentw_shape_ok(const IntegrateDev& a, bool dae) {
    if (a.flags) return false;
    const int H = a.xd;
    if (H < 4 || H > 128 || (H & 3)) return false;
    if (!dae) return a.zd == H && two_h(a.de, 6 * H, H);
    if (a.vd != H || a.id != H || (a.zd != H && a.zd != 0)) return false;
    const int nblk = a.zd ? 4 : 3;
    return two_h(a.de, 3 * nblk * H, H) && two_h(a.ae, (2 * nblk - 1) * H, H);
}

bool latentw_ptrs_ok(const IntegrateDev& a, bool dae) {
    auto mis = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
    if (mis(a.a0) || mis(a.xo)) return false;
    if (!dae) return al4w(a.x) && al4w(a.z) && (!a.ev || (!mis(a.zj) && a.zjb % 4 == 0 && a.zje % 4 == 0));
    if (mis(a.x_init) || mis(a.io) || !al4w(a.v) || (a.zd && !al4w(a.z))) return false;
    if (a.ev) {
        if (a.zd && (mis(a.zj) || a.zjb % 4 || a.zje % 4)) return false;
        if (mis(a.vj) || a.vjb % 4 || a.vje % 4) return false;
    }
    return true;
}

size_t latentw_pack_floats() { return 4 * packw_f4(8, 4, 3, true) + 64; }

hipError_t launch_latent_wide(const IntegrateDev& a, bool dae, float* pack, hipStream_t stream) {
    const int H = a.xd, nw = H <= 64 ? 4 : 8;
    PackW p;
    memset(&p, 0, sizeof(p));
    p.nw = nw; p.H = H; p.dae = dae ? 1 : 0;
    p.nblk = dae ? (a.zd ? 4 : 3) : 2;
    p.nbe = p.nblk - 1;
    p.dw1 = a.de.w[0]; p.db1 = a.de.bias[0]; p.dw2 = a.de.w[1]; p.db2 = a.de.bias[1];
    if (dae) { p.aw1 = a.ae.w[0]; p.ab1 = a.ae.bias[0]; p.aw2 = a.ae.w[1]; p.ab2 = a.ae.bias[1]; }
    p.out = reinterpret_cast<f4*>(pack);
    hipLaunchKernelGGL(packw_kernel, dim3(256), dim3(256), 0, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return nw == 4 ? launchw_nw<4>(a, dae, p.out, stream) : launchw_nw<8>(a, dae, p.out, stream);
}


}  // namespace psnode

using namespace psnode;

extern "C" {

int32_t psnode_latent_backward_wide_supported(int32_t hidden, int32_t z_dim, int32_t dae) {
    return hidden >= 4 && hidden <= 128 && (hidden & 3) == 0 && hidden != 16 && hidden != 64 && (z_dim == hidden || (dae && z_dim == 0));
}

size_t psnode_latent_backward_wide_workspace_bytes(int32_t hidden) {
    const int nw = hidden <= 64 ? 4 : 8;
    return ((size_t)5 * nw * nw * 64 * 4 + 64) * sizeof(float);
}

int32_t psnode_latent_backward_wide_f32(const psnode_latent_bwd_wide_args_f32* p, void* workspace, size_t workspace_bytes, void* stream) {
    if (!p) return PSNODE_ERR_NULL;
    if (p->method < PSNODE_EULER || p->method > PSNODE_RK4_38) return PSNODE_ERR_METHOD;
    if (p->T < 1 || p->B < 1) return PSNODE_ERR_DIMS;
    if (!psnode_latent_backward_wide_supported(p->hidden, p->z_dim, p->dae)) return PSNODE_ERR_UNSUPPORTED;
    const int H = p->hidden, nw = H <= 64 ? 4 : 8, nblk = p->dae ? (p->z_dim ? 4 : 3) : 2;
    if (p->de.n_layers != 2 || p->de.in_dim != 3 * nblk * H || p->de.out_dim[0] != H || p->de.out_dim[1] != H) return PSNODE_ERR_DIMS;
    if (p->dae && (p->ae.n_layers != 2 || p->ae.in_dim != (2 * nblk - 1) * H || p->ae.out_dim[0] != H || p->ae.out_dim[1] != H)) return PSNODE_ERR_DIMS;
    if (!p->t.ptr || !p->grad_xs || !p->grad_x0 || !p->de.weight[0] || !p->de.weight[1]) return PSNODE_ERR_NULL;
    if (p->T > 1 && (!p->saved_act || !p->gk || !p->d1 || !p->d1s)) return PSNODE_ERR_NULL;
    if (p->dae && (!p->saved_ae_act || !p->gi || !p->da1 || !p->ae.weight[0] || !p->ae.weight[1])) return PSNODE_ERR_NULL;
    if (p->dae && p->event_idx && (!p->saved_ev_act || !p->gi_ev || !p->da1_ev)) return PSNODE_ERR_NULL;
    auto mis = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) != 0; };
    if (mis(p->grad_xs) || mis(p->grad_is) || mis(p->saved_act) || mis(p->saved_ae_act) || mis(p->saved_ev_act) || mis(p->gk) || mis(p->d1) ||
        mis(p->d1s) || mis(p->gi) || mis(p->da1) || mis(p->gi_ev) || mis(p->da1_ev) || mis(p->grad_x0)) return PSNODE_ERR_DIMS;
    if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 255) || workspace_bytes < psnode_latent_backward_wide_workspace_bytes(H)) return PSNODE_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    PackWT pk;
    pk.nw = nw; pk.H = H; pk.nblk = nblk; pk.dae = p->dae ? 1 : 0;
    pk.dw1 = p->de.weight[0]; pk.dw2 = p->de.weight[1];
    pk.aw1 = p->dae ? p->ae.weight[0] : nullptr; pk.aw2 = p->dae ? p->ae.weight[1] : nullptr;
    pk.out = static_cast<f4*>(workspace);
    hipLaunchKernelGGL(packwt_kernel, dim3(128), dim3(256), 0, s, pk);
    BwdWDev a;
    memset(&a, 0, sizeof(a));
    a.method = p->method; a.H = H; a.zd = p->z_dim; a.dae = p->dae; a.T = p->T; a.B = p->B;
    a.t = ViewDev{p->t.ptr, p->t.stride_t, p->t.stride_b};
    a.ev = p->event_idx; a.gxs = p->grad_xs; a.gis = p->grad_is;
    a.sact = p->saved_act; a.saeact = p->saved_ae_act; a.sevact = p->saved_ev_act;
    a.gk = p->gk; a.d1 = p->d1; a.d1s = p->d1s; a.gi = p->gi; a.da1 = p->da1; a.gi_ev = p->gi_ev; a.da1_ev = p->da1_ev; a.gx0 = p->grad_x0;
    const dim3 grid((unsigned)((a.B + 15) / 16));
    const f4* img = pk.out;
#define PSNODE_LW(METHOD_, DAE_)                                                                                                         \
    {                                                                                                                                    \
        if (nw == 4) hipLaunchKernelGGL((latent_wide_bwd_kernel<METHOD_, DAE_, 4>), grid, dim3(256), 0, s, a, img);                      \
        else hipLaunchKernelGGL((latent_wide_bwd_kernel<METHOD_, DAE_, 8>), grid, dim3(512), 0, s, a, img);                              \
    }
    if (p->dae) {
        switch (p->method) {
            case PSNODE_EULER: PSNODE_LW(PSNODE_EULER, true) break;
            case PSNODE_MIDPOINT: PSNODE_LW(PSNODE_MIDPOINT, true) break;
            default: PSNODE_LW(PSNODE_RK4_38, true) break;
        }
    } else {
        switch (p->method) {
            case PSNODE_EULER: PSNODE_LW(PSNODE_EULER, false) break;
            case PSNODE_MIDPOINT: PSNODE_LW(PSNODE_MIDPOINT, false) break;
            default: PSNODE_LW(PSNODE_RK4_38, false) break;
        }
    }
#undef PSNODE_LW
    return hipGetLastError() == hipSuccess ? PSNODE_OK : PSNODE_ERR_HIP;
}

}  // extern "C"
