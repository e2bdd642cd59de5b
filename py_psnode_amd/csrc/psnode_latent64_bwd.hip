// K9 -- backward (discretise-then-optimise) pass through the latent integrators of the direct_encode variants at
// hidden_dim 64 (K3c's shapes; what neural_01_DAE_02_direct_encode.py ships with, :267):
//   ODE:  DE = Linear(6H,H) ELU Linear(H,H)                          state Xh[64], external Zh[64]
//   DAE:  DE = Linear(12H|9H,H) ELU Linear(H,H), AE = Linear(7H|5H,H) ELU Linear(H,H),  blocks x | [z] | v | i of width 64
// Replaces loss.backward() on the unrolled integrate_ODE / integrate_DAE graph between the encoders and the decoders
// (neural_01_DAE_02_direct_encode.py:359-370 through my_solvers.py:94-129).
//
// Decomposition as K3c: one workgroup = 4 waves = one tile of 16 trajectories; wave w owns hidden units AND state dims
// 16w..16w+15; every matrix is a set of 64x64 blocks (F_b = Ws_b + Wd_b folded as in the forward).  Per block three
// operations exist, all on v_mfma_f32_16x16x4_f32:
//   forward     y_own = Blk . gather(v)            "mid-layer" format, 16 registers per lane (K3c)
//   transposed  g_own = reduce_scatter(Blk^T d)    each wave multiplies its own 16 units (split-K), partial sums are
//                                                  reduce-scattered through LDS (K4's machinery)
//   gradient    dBlk[own units][:] += d (x) v      contraction over the tile's 16 trajectories: A = d^T (in-wave transpose),
//                                                  B = v^T tiles that every wave publishes for its own 16 dims
// Control flow as K7: per grid point the AE head's VJP (DAE), then the step's DE stages forwards / backwards with the
// external blocks frozen over the stages; event steps recompute i = g(x_k; jumps) and chain its VJP in.  The a0 / (s-a0)
// column groups of dW1 and d all_initial are reconstructed once at the end from sum_t(delta1) (a0 is constant over time).
// Register budget (DAE): DE forward blocks + W2 (80), the x / i transposed blocks + W2^T (48), AE forward blocks (48) and
// 160 accumulators stay in VGPR/AGPRs; the AE's transposed blocks and the DE's transposed z|v blocks (up to 96 KB) live
// in LDS, each lane reading back the A-operand values it wrote; W2 of the AE (event steps only) and the a0 blocks (epilogue
// only) are streamed from the packed image.
#define PSNODE_ELU_LITERALS   // register-bound kernels: ELU coefficients as literals, not as 8 resident VGPRs (psnode_common.h)
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "psnode_latent64_bwd.h"

namespace psnode {
namespace {


__device__ __forceinline__ f4 elu9(f4 v) { return elu_quad(v); }

// pack[wave][reg][lane]; forward section identical to K3c's image, then the transposed section:
//   fwd:  F[nfront] (16 each) | B1 (4) | W2 (16) | B2 (4) | A0[nblk] (16 each)
//   T:    FT[nfront] (16 each) | W2T (16) | A0T[nblk] (16 each)
//   forward format     reg 4c+r = Blk[16w + i][16((w+c)&3) + 4g + r]
//   transposed format  reg 4c+r = Blk[16w + 4g + r][16((w+c)&3) + i]
//   DE: F_b = Ws_b + Wd_b, A0_b = Wa0_b - Wd_b.   AE: F_b = the block right after the a0 group, A0_b = Wa0_b.
struct Pack9 {
    int ae, nblk, nfront, k1;
    const float *w1, *b1, *w2, *b2;
    float* out;
};
__host__ __device__ inline int p9_regs(int nfront, int nblk) { return 2 * (16 * nfront + 16 + 16 * nblk) + 8; }

__global__ void pack9_kernel(const Pack9 p) {
    const int n = p.nblk * H9;
    const int B1 = 16 * p.nfront, W2 = B1 + 4, B2 = W2 + 16, A0 = B2 + 4, FT = A0 + 16 * p.nblk, W2T = FT + 16 * p.nfront,
              A0T = W2T + 16, R = A0T + 16 * p.nblk;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < NW9 * R * 64; idx += gridDim.x * blockDim.x) {
        const int lane = idx & 63, reg = (idx >> 6) % R, w = (idx >> 6) / R, i = lane & 15, g = lane >> 4;
        // element Blk[u][c] of block `blk` of kind 0 (F) / 1 (A0) / 2 (W2)
        auto elem = [&](int kind, int blk, int u, int c) -> float {
            const float* row = p.w1 + (size_t)u * p.k1;
            const int cc = H9 * blk + c;
            if (kind == 2) return p.w2[(size_t)u * H9 + c];
            if (kind == 0) return p.ae ? row[n + cc] : row[2 * n + cc] + row[n + cc];
            return p.ae ? row[cc] : row[cc] - row[n + cc];
        };
        auto fwd = [&](int kind, int blk, int kk) { return elem(kind, blk, 16 * w + i, 16 * ((w + (kk >> 2)) & 3) + 4 * g + (kk & 3)); };
        auto tr = [&](int kind, int blk, int kk) { return elem(kind, blk, 16 * w + 4 * g + (kk & 3), 16 * ((w + (kk >> 2)) & 3) + i); };
        float v;
        if (reg < B1) v = fwd(0, reg >> 4, reg & 15);
        else if (reg < W2) v = p.b1[16 * w + 4 * g + (reg - B1)];
        else if (reg < B2) v = fwd(2, 0, reg - W2);
        else if (reg < A0) v = p.b2[16 * w + 4 * g + (reg - B2)];
        else if (reg < FT) v = fwd(1, (reg - A0) >> 4, (reg - A0) & 15);
        else if (reg < W2T) v = tr(0, (reg - FT) >> 4, (reg - FT) & 15);
        else if (reg < A0T) v = tr(2, 0, reg - W2T);
        else v = tr(1, (reg - A0T) >> 4, (reg - A0T) & 15);
        p.out[idx] = v;
    }
}

struct V9 { f4 v[4]; };      // a 64-wide vector in chunk layout: v[c] = dims 16((w+c)&3) + 4g + (0..3) of trajectory j

// REC = false: the training forward (K3c SAVE instances) saved the hidden ELU outputs and stage inputs of every DE stage, the AE head's
// hidden layer per grid point and per event taken, and the event-time i0 (IntegrateDev::sact / sxst / saeact / sevact / sevi): nothing is
// evaluated forwards in here -- no per-step constant, no phase A, no head recompute --, the forward blocks are not even loaded, and the
// rows of step k are requested in front of the head of grid point k+1, whose work hides their latency.
template <int METHOD, int NBE, bool DAE, bool REC = true>
__global__ __launch_bounds__(256) void latent64_backward_kernel(const Bwd9Dev d, const float* __restrict__ pack_de,
                                                                 const float* __restrict__ pack_ae) {
    constexpr int S = rk_stages(METHOD);
    constexpr int NBLK = 1 + NBE, NZV = DAE ? NBE - 1 : NBE, n = H9 * NBLK, NAE = DAE ? NBE : 0;
    constexpr bool TREG = DAE && !REC;                      // REC = false: the forward blocks are gone, every transposed block fits the registers
    constexpr int NLB = (DAE && REC) ? 2 * NBE : 0;         // 64x64 blocks kept in LDS: AE FT[NAE], AE W2T, DE FT of the z|v blocks
    constexpr int LQ_AFT = 0, LQ_AW2T = NAE, LQ_DFT = NAE + 1;
    // register indices of the packed images
    constexpr int D_B1 = 16 * NBLK, D_W2 = D_B1 + 4, D_B2 = D_W2 + 16, D_A0 = D_B2 + 4, D_FT = D_A0 + 16 * NBLK, D_W2T = D_FT + 16 * NBLK,
                  D_A0T = D_W2T + 16, D_R = D_A0T + 16 * NBLK;
    constexpr int A_B1 = 16 * NAE, A_W2 = A_B1 + 4, A_B2 = A_W2 + 16, A_A0 = A_B2 + 4, A_FT = A_A0 + 16 * NBLK, A_W2T = A_FT + 16 * NAE,
                  A_A0T = A_W2T + 16, A_R = A_A0T + 16 * NBLK;
    const IntegrateDev& a = d.a;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    f4* xbuf = reinterpret_cast<f4*>(lds);                 // [2][4][64]        all-gather
    f4* rsbuf = xbuf + 2 * NW9 * 64;                       // [2][4][4][64]     reduce-scatter
    f4* pub = rsbuf + 2 * NW9 * NW9 * 64;                  // [4 slots][4][64]  published transposed own tiles
    f4* wl = pub + 4 * NW9 * 64;                           // [NLB][4 chunks][4 waves][64]
    float* scr_all = reinterpret_cast<float*>(wl + NLB * 4 * NW9 * 64);

    const int l = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = l >> 4, j = l & 15, i = l & 15;
    float* scr = scr_all + w * SCR9;
    const long long b0 = (long long)blockIdx.x * 16;
    const bool valid = b0 + j < a.B;
    const long long b = valid ? b0 + j : a.B - 1;
    const int own = 16 * w + 4 * g;                        // first of this lane's four own dims

    // ---- weights
    const float* pw = pack_de + (size_t)w * D_R * 64 + l;
    const float* pwa = pack_ae + (size_t)w * A_R * 64 + l;
    float wf[REC ? NBLK : 1][16], w2[16], w2t[16], wftx[16], wfti[DAE ? 16 : 1], wftz[DAE ? 1 : 16];
    f4 b1r, b2r;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if constexpr (REC) {
#pragma unroll
            for (int blk = 0; blk < NBLK; ++blk) wf[blk][k] = pw[(16 * blk + k) * 64];
            w2[k] = pw[(D_W2 + k) * 64];
        } else { wf[0][k] = 0.0f; w2[k] = 0.0f; }
        w2t[k] = pw[(D_W2T + k) * 64];
        wftx[k] = pw[(D_FT + k) * 64];
        if constexpr (DAE) wfti[k] = pw[(D_FT + 16 * (NBLK - 1) + k) * 64];
        else wftz[k] = pw[(D_FT + 16 + k) * 64];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { b1r[r] = pw[(D_B1 + r) * 64]; b2r[r] = pw[(D_B2 + r) * 64]; }
    float af[(DAE && REC) ? NAE : 1][16];
    f4 ab1r = z9(), ab2r = z9();
    f4* wlp = wl + w * 64 + l;                              // block q, chunk c at wlp[(q*4 + c) * 256]
    if constexpr (DAE) {
        if constexpr (REC) {
#pragma unroll
            for (int k = 0; k < 16; ++k)
#pragma unroll
                for (int bb = 0; bb < NAE; ++bb) af[bb][k] = pwa[(16 * bb + k) * 64];
#pragma unroll
            for (int r = 0; r < 4; ++r) { ab1r[r] = pwa[(A_B1 + r) * 64]; ab2r[r] = pwa[(A_B2 + r) * 64]; }
        }
        auto stage_blk = [&](const int q, const float* src) {   // 16 packed registers -> LDS block q
#pragma unroll
            for (int c = 0; c < 4; ++c) wlp[(q * 4 + c) * 256] = f4{src[(4 * c) * 64], src[(4 * c + 1) * 64], src[(4 * c + 2) * 64], src[(4 * c + 3) * 64]};
        };
        if constexpr (REC) {
#pragma unroll
            for (int bb = 0; bb < NAE; ++bb) stage_blk(LQ_AFT + bb, pwa + (A_FT + 16 * bb) * 64);
            stage_blk(LQ_AW2T, pwa + A_W2T * 64);
#pragma unroll
            for (int s = 0; s < NZV; ++s) stage_blk(LQ_DFT + s, pw + (D_FT + 16 * (1 + s)) * 64);
        }
    }
    // TREG: the same six blocks in registers (LQ order: AE FT[NAE], AE W2T, DE FT of the z|v blocks)
    float tq[TREG ? 2 * NBE : 1][16];
    if constexpr (TREG) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
#pragma unroll
            for (int bb = 0; bb < NAE; ++bb) tq[LQ_AFT + bb][k] = pwa[(A_FT + 16 * bb + k) * 64];
            tq[LQ_AW2T][k] = pwa[(A_W2T + k) * 64];
#pragma unroll
            for (int s = 0; s < NZV; ++s) tq[LQ_DFT + s][k] = pw[(D_FT + 16 * (1 + s) + k) * 64];
        }
    }

    int coff[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) coff[c] = 16 * ((w + c) & 3) + 4 * g;
    auto load_chunks = [&](const float* rowptr) -> V9 {
        V9 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) o.v[c] = *reinterpret_cast<const f4*>(rowptr + coff[c]);
        return o;
    };
    auto mm = [&](const float (&wr)[16], const V9& x, f4& accA, f4& accB) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            accA = m9(wr[4 * c + 0], x.v[c][0], accA); accB = m9(wr[4 * c + 1], x.v[c][1], accB);
            accA = m9(wr[4 * c + 2], x.v[c][2], accA); accB = m9(wr[4 * c + 3], x.v[c][3], accB);
        }
    };
    int p = 0, q = 0;
    auto gather = [&](const f4 ownv) -> V9 {
        xbuf[(p * NW9 + w) * 64 + l] = ownv;
        lds_barrier();
        V9 o;
        o.v[0] = ownv;
#pragma unroll
        for (int c = 1; c < 4; ++c) o.v[c] = xbuf[(p * NW9 + ((w + c) & 3)) * 64 + l];
        __builtin_amdgcn_sched_barrier(0);      // all three reads in flight before the first dependent MFMA (as K1)
        p ^= 1;
        return o;
    };
    auto layer = [&](const float (&wr)[16], const f4 init, const f4 ownv) -> f4 {   // y_own = init + Blk . gather(own)
        const V9 x = gather(ownv);
        f4 accA = init, accB = z9();
        mm(wr, x, accA, accB);
        return accA + accB;
    };
#ifndef PSNODE_K9_HOOK
#define PSNODE_K9_HOOK 1      // the outer products (weight gradients) that precede a reduce-scatter are issued between its LDS writes and its barrier:
                              // one wave per SIMD has nothing else to keep the MFMA pipe busy while the partial sums travel (K4f: PSNODE_K4F_DEFER_DW)
#endif
    auto no_hook = [] {};
    auto reduce_scatter = [&](const f4 (&part)[4], auto&& hook) -> f4 {
#pragma unroll
        for (int c = 1; c < 4; ++c) rsbuf[((q * NW9 + ((w + c) & 3)) * NW9 + w) * 64 + l] = part[c];
        hook();
        lds_barrier();
        f4 out = part[0];
#pragma unroll
        for (int c = 1; c < 4; ++c) out += rsbuf[((q * NW9 + w) * NW9 + ((w + c) & 3)) * 64 + l];
        q ^= 1;
        return out;
    };
    // g_own = (Blk^T d)_own from a transposed block in registers / in LDS / streamed from the packed image
    auto mulT = [&](const f4 w4, const f4 dl) -> f4 {
        f4 acc = m9(w4[0], dl[0], z9());
        acc = m9(w4[1], dl[1], acc);
        acc = m9(w4[2], dl[2], acc);
        return m9(w4[3], dl[3], acc);
    };
    auto blkT_reg = [&](const float (&wt)[16], const f4 dl, auto&& hook) -> f4 {
        f4 part[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) part[c] = mulT(f4{wt[4 * c], wt[4 * c + 1], wt[4 * c + 2], wt[4 * c + 3]}, dl);
        return reduce_scatter(part, hook);
    };
    auto blkT_lds = [&](const int qb, const f4 dl, auto&& hook) -> f4 {
        f4 part[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if constexpr (TREG) part[c] = mulT(f4{tq[qb][4 * c], tq[qb][4 * c + 1], tq[qb][4 * c + 2], tq[qb][4 * c + 3]}, dl);
            else part[c] = mulT(wlp[(qb * 4 + c) * 256], dl);
        }
        return reduce_scatter(part, hook);
    };
    auto partT_mem = [&](const float* src, const f4 dl, f4 (&part)[4], const bool accumulate) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f4 w4 = f4{src[(4 * c) * 64], src[(4 * c + 1) * 64], src[(4 * c + 2) * 64], src[(4 * c + 3) * 64]};
            const f4 r = mulT(w4, dl);
            part[c] = accumulate ? part[c] + r : r;
        }
    };
    // D-layout tile (rows 4g+r, col j) -> o[kk] = M[row i][col 4kk+g], via the private padded LDS tile
    auto transpose = [&](const f4 v) -> f4 {
        *reinterpret_cast<f4*>(scr + 4 * l + 8 * g) = v;
        const float* s = scr + 4 * (16 * (i >> 2) + g) + 8 * (i >> 2) + (i & 3);
        return f4{s[0], s[16], s[32], s[48]};
    };
    auto publish = [&](const int slot, const f4 ownv) { pub[(slot * NW9 + w) * 64 + l] = transpose(ownv); };
    // dBlk[own units][:] += d (x) v  with dT = transpose(d_own) and v's tiles published in `slot`
#ifndef PSNODE_K9_OUTER_AHEAD
#define PSNODE_K9_OUTER_AHEAD 1      // the four published tiles read first, then four independent accumulator chains (as K4f / K7f / K7h)
#endif
    auto outer = [&](A9& acc, const f4 dT, const int slot) {
        if constexpr (PSNODE_K9_OUTER_AHEAD) {
            f4 vT[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) vT[c] = pub[(slot * NW9 + ((w + c) & 3)) * 64 + l];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc.c[c] = m9(dT[kk], vT[c][kk], acc.c[c]);
            return;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f4 vT = pub[(slot * NW9 + ((w + c) & 3)) * 64 + l];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc.c[c] = m9(dT[kk], vT[kk], acc.c[c]);
        }
    };

    // ---- per-trajectory constants: c0 = b1 + sum_blk A0_blk . a0_blk   (DE and AE)
    f4 c0A = b1r, c0B = z9(), caA = ab1r, caB = z9();
    for (int blk = 0; blk < (REC ? NBLK : 0); ++blk) {
        const V9 a0v = load_chunks(a.a0 + b * n + H9 * blk);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kq = 4 * c + r;
                if (kq & 1) c0B = m9(pw[(D_A0 + 16 * blk + kq) * 64], a0v.v[c][r], c0B);
                else c0A = m9(pw[(D_A0 + 16 * blk + kq) * 64], a0v.v[c][r], c0A);
                if constexpr (DAE) {
                    if (kq & 1) caB = m9(pwa[(A_A0 + 16 * blk + kq) * 64], a0v.v[c][r], caB);
                    else caA = m9(pwa[(A_A0 + 16 * blk + kq) * 64], a0v.v[c][r], caA);
                }
            }
    }
    const f4 c0 = c0A + c0B, c0a = caA + caB;

    const long long tst = a.t.st, nT = a.T;
    // addressing: sbase(uniform row base) + 32-bit per-lane offset (psnode_common.h)
    const unsigned offR = (unsigned)(b * H9) + own;                 // rows of [*, B, 64] tensors
    const unsigned offT = (unsigned)(b * a.t.sb);
    // streamed block s (0 = z or, when the model has no z, v; 1 = v): sources, jump tables, gradient destinations
    const bool has_z = a.zd > 0;
    const float* spb[2] = {has_z ? a.z.p : (DAE ? a.v.p : nullptr), DAE ? a.v.p : nullptr};
    const long long sst[2] = {has_z ? a.z.st : a.v.st, a.v.st};
    const unsigned spo[2] = {(unsigned)(b * (has_z ? a.z.sb : a.v.sb)) + own, (unsigned)(b * a.v.sb) + own};
    const float* jpb[2] = {has_z ? a.zj : (DAE ? a.vj : nullptr), DAE ? a.vj : nullptr};
    const long long jse[2] = {has_z ? a.zje : a.vje, a.vje};
    const unsigned jpo[2] = {(unsigned)(b * (has_z ? a.zjb : a.vjb)) + own, (unsigned)(b * a.vjb) + own};
    float* gdst[2] = {has_z ? d.gz : d.gv, d.gv};
    float* gjdst[2] = {has_z ? d.gzj : d.gvj, d.gvj};
    const unsigned offJ = (unsigned)(b * d.n_events * H9) + own;
    auto load_zv = [&](const int s, const long long k, const int ev) -> f4 {
        if (ev >= 0) return ldg<f4>(sbase(jpb[s] + ev * jse[s]), 4u * jpo[s]);
        return ldg<f4>(sbase(spb[s] + k * sst[s]), 4u * spo[s]);
    };
    auto row_of = [&](const float* base, const long long k) -> f4 {
        return ldg<f4>(sbase(base + k * a.B * H9), 4u * offR);
    };
    auto store_zv = [&](const int s, const long long grid, const int ev, const f4 val) {
        if (!valid) return;
        if (ev >= 0) { if (gjdst[s]) stg<f4>(sbase(gjdst[s] + (long long)ev * H9), 4u * offJ, val); }
        else if (gdst[s]) stg<f4>(sbase(gdst[s] + grid * a.B * H9), 4u * offR, val);
    };

    // ---- accumulators (whole launch)
    A9 accF[NBLK], accW2, accAF[DAE ? NAE : 1], accW2a;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        accW2.c[c] = z9(); accW2a.c[c] = z9();
#pragma unroll
        for (int blk = 0; blk < NBLK; ++blk) accF[blk].c[c] = z9();
#pragma unroll
        for (int bb = 0; bb < (DAE ? NAE : 1); ++bb) accAF[bb].c[c] = z9();
    }
    f4 S1 = z9(), SB2 = z9(), AS1 = z9(), ASB2 = z9();

    // ---- AE head: hidden layer at (x; z|v) with the transposed inputs published in slots 0..NZV, h1a in slot 3
    f4 ah1 = z9();
    struct ZV { f4 b[NZV > 0 ? NZV : 1]; };
    auto ae_hidden = [&](const f4 xo, const ZV& zv) {
        publish(0, xo);
#pragma unroll
        for (int s = 0; s < NZV; ++s) publish(1 + s, zv.b[s]);
        if constexpr (REC) {
            f4 accA = c0a, accB = z9();
            if constexpr (DAE) {
                { const V9 xg = gather(xo); mm(af[0], xg, accA, accB); }
#pragma unroll
                for (int s = 0; s < NZV; ++s) { const V9 zg = gather(zv.b[s]); mm(af[1 + s], zg, accA, accB); }
            }
            ah1 = elu9(accA + accB);
        }                                       // REC = false: the caller has put the saved hidden layer into ah1
    };
    // output of the AE head from ah1 (event steps only): W2 streamed from the packed image
    auto ae_output = [&]() -> f4 {
        const V9 hg = gather(ah1);
        f4 accA = ab2r, accB = z9();
        // (opaque pointer: the addresses of these 16 loads are loop-invariant, and hoisted out of the time loop they are 13 VGPR pairs
        //  that live -- spilled -- across every step for the sake of the event steps)
        const float* pqo = pwa + A_W2 * 64;
        asm volatile("" : "+v"(pqo));
        const gptr<const float> pq = (gptr<const float>)pqo;      // (global, not generic: an opaque generic pointer loads flat_*)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            accA = m9(pq[(4 * c + 0) * 64], hg.v[c][0], accA); accB = m9(pq[(4 * c + 1) * 64], hg.v[c][1], accB);
            accA = m9(pq[(4 * c + 2) * 64], hg.v[c][2], accA); accB = m9(pq[(4 * c + 3) * 64], hg.v[c][3], accB);
        }
        return accA + accB;
    };
    // VJP of the AE head at (x; z|v) with output gradient gi (own i dims): accumulates the AE parameter gradients,
    // returns the gradient w.r.t. x (own dims) and w.r.t. the streamed blocks
    auto ae_vjp = [&](const f4 xo, const ZV& zv, const f4 gi, ZV& gzv) -> f4 {
        ae_hidden(xo, zv);
        publish(3, ah1);
        ASB2 += gi;
        const f4 d1 = blkT_lds(LQ_AW2T, gi, no_hook) * dact9(ah1);          // the barrier inside publishes slots 0..3
        AS1 += d1;
        const f4 giT = transpose(gi);
        const f4 dT = transpose(d1);
        f4 gx = z9();
#pragma unroll
        for (int bb = 0; bb < NAE; ++bb) {
            auto grads = [&] { if (bb == 0) outer(accW2a, giT, 3); outer(accAF[bb], dT, bb); };      // every read of the slots precedes the last barrier
            if constexpr (!PSNODE_K9_HOOK) grads();
            const f4 gb = PSNODE_K9_HOOK ? blkT_lds(LQ_AFT + bb, d1, grads) : blkT_lds(LQ_AFT + bb, d1, no_hook);
            if (bb == 0) gx = gb;
            else gzv.b[bb - 1] = gb;
        }
        return gx;
    };

    // ---- state of the sweep
    f4 gcarry = z9(), gicarry = z9();
    ZV dezv;
#pragma unroll
    for (int s = 0; s < (NZV > 0 ? NZV : 1); ++s) dezv.b[s] = z9();

    for (long long jg = nT - 1; jg >= 0; --jg) {
        f4 g1 = gcarry + (valid ? row_of(d.gxs, jg) : z9());
        // REC = false: the saved rows of step jg-1 (clamped at the last iteration: no branch around the loads), consumed behind the head
        f4 xst[S], h1[S];
        if constexpr (!REC) {
            const long long ks_ = (jg > 0 ? jg - 1 : 0) * S;
#pragma unroll
            for (int s = 0; s < S; ++s) { xst[s] = row_of(a.sxst, ks_ + s); h1[s] = row_of(a.sact, ks_ + s); }
        }
        if constexpr (DAE) {
            // ================= (1) AE head at grid point jg (raw z|v)
            const f4 xj = row_of(d.xs, jg);
            ZV zvj, gzv;
#pragma unroll
            for (int s = 0; s < NZV; ++s) zvj.b[s] = load_zv(s, jg, -1);
            const f4 gi = gicarry + ((d.gis && valid) ? row_of(d.gis, jg) : z9());
            if constexpr (!REC) ah1 = row_of(a.saeact, jg);
            g1 += ae_vjp(xj, zvj, gi, gzv);
#pragma unroll
            for (int s = 0; s < NZV; ++s) store_zv(s, jg, -1, dezv.b[s] + gzv.b[s]);
        } else {
            if (jg == nT - 1) store_zv(0, jg, -1, z9());           // z[T-1] is never read by the ODE loop
        }
        if (jg == 0) { gcarry = g1; break; }

        // ================= (2) step k = jg-1
        const long long k = jg - 1;
        const int ev = a.ev ? __builtin_amdgcn_readfirstlane(a.ev[k]) : -1;
        const float h_ = ldg<float>(sbase(a.t.p + jg * tst), 4u * offT) - ldg<float>(sbase(a.t.p + k * tst), 4u * offT);
        const f4 x0 = row_of(d.xs, k);
        f4 ext[NBE];
#pragma unroll
        for (int s = 0; s < NZV; ++s) ext[s] = load_zv(s, k, ev);
        if constexpr (DAE) {
            if constexpr (REC) {
                if (ev >= 0) {       // i_in = g(x_k; z_jump, v_jump)  (my_solvers.py:108-110)
                    ZV zq;
#pragma unroll
                    for (int s = 0; s < NZV; ++s) zq.b[s] = ext[s];
                    ae_hidden(x0, zq);
                    ext[NBE - 1] = ae_output();
                } else {
                    ext[NBE - 1] = row_of(d.is_, k);
                }
            } else {                 // (one load: the row base is selected, not the loaded values)
                ext[NBE - 1] = row_of(ev >= 0 ? a.sevi : d.is_, ev >= 0 ? (long long)ev : k);
            }
        }
        if constexpr (REC) {
            f4 czA = c0, czB = z9();
#pragma unroll
            for (int e = 0; e < NBE; ++e) { const V9 eg = gather(ext[e]); mm(wf[1 + e], eg, czA, czB); }
            const f4 cz = czA + czB;

            // ---- phase A: stage evaluations
            f4 ks[S];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                f4 acc = z9();
#pragma unroll
                for (int jj = 0; jj < s; ++jj) acc += rk_a(METHOD, s, jj) * ks[jj];
                xst[s] = s == 0 ? x0 : x0 + h_ * acc;
                h1[s] = elu9(layer(wf[0], cz, xst[s]));
                ks[s] = layer(w2, b2r, h1[s]);
            }
        }
        // ---- phase B: stages backwards
        f4 gks[S], gx0 = g1, D1 = z9();
#pragma unroll
        for (int s = 0; s < S; ++s) gks[s] = (h_ * rk_b(METHOD, s)) * g1;
#pragma unroll
        for (int s = S - 1; s >= 0; --s) {
            const f4 gk = gks[s];
            SB2 += gk;
            publish(0, h1[s]);
            publish(1, xst[s]);
            const f4 d1 = blkT_reg(w2t, gk, no_hook) * dact9(h1[s]);        // barrier inside: slots 0, 1 visible afterwards
            D1 += d1;
            const f4 gkT = transpose(gk), d1T = transpose(d1);
            auto grads = [&] { outer(accW2, gkT, 0); outer(accF[0], d1T, 1); };
            if constexpr (!PSNODE_K9_HOOK) grads();
            const f4 gx = PSNODE_K9_HOOK ? blkT_reg(wftx, d1, grads) : blkT_reg(wftx, d1, no_hook);      // barrier after every read of slots 0, 1
            gx0 += gx;
#pragma unroll
            for (int jj = 0; jj < s; ++jj) gks[jj] += (h_ * rk_a(METHOD, s, jj)) * gx;
        }
        S1 += D1;
        // ---- external blocks: frozen over the stages
#pragma unroll
        for (int e = 0; e < NBE; ++e) publish(e, ext[e]);
        lds_barrier();
        const f4 DT = transpose(D1);
        f4 gext[NBE];
#pragma unroll
        for (int e = 0; e < NBE; ++e) {
            auto grads = [&] { outer(accF[1 + e], DT, e); };
            if constexpr (!PSNODE_K9_HOOK) grads();
            if constexpr (PSNODE_K9_HOOK) {
                if constexpr (DAE) gext[e] = e < NZV ? blkT_lds(LQ_DFT + e, D1, grads) : blkT_reg(wfti, D1, grads);
                else gext[e] = blkT_reg(wftz, D1, grads);
            } else {
                if constexpr (DAE) gext[e] = e < NZV ? blkT_lds(LQ_DFT + e, D1, no_hook) : blkT_reg(wfti, D1, no_hook);
                else gext[e] = blkT_reg(wftz, D1, no_hook);
            }
        }
        if constexpr (DAE) {
            if (ev >= 0) {
                // the algebraic input was g(x_k; jumps): chain its VJP in; z|v gradients (DE + AE part) go to the jump arrays
                ZV zq, gq;
#pragma unroll
                for (int s = 0; s < NZV; ++s) zq.b[s] = ext[s];
                if constexpr (!REC) ah1 = row_of(a.sevact, (long long)ev);
                gx0 += ae_vjp(x0, zq, gext[NBE - 1], gq);
#pragma unroll
                for (int s = 0; s < NZV; ++s) { store_zv(s, k, ev, gext[s] + gq.b[s]); dezv.b[s] = z9(); }
                gicarry = z9();
            } else {
#pragma unroll
                for (int s = 0; s < NZV; ++s) dezv.b[s] = gext[s];
                gicarry = gext[NBE - 1];
            }
        } else {
            if (ev >= 0) { store_zv(0, k, ev, gext[0]); store_zv(0, k, -1, z9()); }
            else store_zv(0, k, -1, gext[0]);
        }
        gcarry = gx0;
    }

    // ---- epilogue.  Lane coordinates are re-derived from an opaque copy of the thread index: computed from the prologue's values, the
    //      epilogue's addresses are live (spilled) across the whole time loop.
    {
    int tid_e = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));   // lane id without keeping v0 (threadIdx.x) alive
    asm volatile("" : "+v"(tid_e));
    const int l_e = tid_e & 63, g = l_e >> 4, j = l_e & 15, own = 16 * w + 4 * g;
    const bool valid = b0 + j < a.B;
    const long long b = valid ? b0 + j : a.B - 1;
    float* scr = scr_all + w * SCR9;
    const int l = l_e, i = j;
    auto transpose = [&](const f4 v) -> f4 {
        *reinterpret_cast<f4*>(scr + 4 * l + 8 * g) = v;
        const float* s = scr + 4 * (16 * (i >> 2) + g) + 8 * (i >> 2) + (i & 3);
        return f4{s[0], s[16], s[32], s[48]};
    };
    if (valid) *reinterpret_cast<f4*>(d.gx0 + b * H9 + own) = gcarry;
    for (int blk = 0; blk < NBLK; ++blk) {       // d all_initial block = A0_blk^T sum_t(delta1)  (DE + AE)
        f4 part[4];
        partT_mem(pw + (D_A0T + 16 * blk) * 64, S1, part, false);
        if constexpr (DAE) partT_mem(pwa + (A_A0T + 16 * blk) * 64, AS1, part, true);
        const f4 ga = reduce_scatter(part, no_hook);
        if (valid) *reinterpret_cast<f4*>(d.ga0 + b * n + H9 * blk + own) = ga;
    }
    // ---- parameter-gradient partials of this workgroup: [DE | AE], nn.Linear order [W1, b1, W2, b2]
    float* wp = d.wpart + (size_t)blockIdx.x * (d.NP_de + d.NP_ae);
    auto write_mlp = [&](float* o, const int K1, auto is_ae_c, auto nfront_c, const f4 s1v, const f4 sb2v, const A9* accB_,
                         const A9& accW2_) __attribute__((always_inline)) {
        constexpr bool is_ae = decltype(is_ae_c)::value;
        constexpr int nfront = decltype(nfront_c)::value;
        const int oB1 = H9 * K1, oW2 = oB1 + H9, oB2 = oW2 + H9 * H9;
        const f4 sT = transpose(s1v);
#pragma unroll
        for (int blk = 0; blk < NBLK; ++blk) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int col = H9 * blk + 16 * ((w + c) & 3) + j;
                f4 ca0 = z9();       // sum_t(delta1)^T (x) a0^T for these 16 columns
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const long long tb = b0 + 4 * kk + g;
                    ca0 = m9(sT[kk], tb < a.B ? a.a0[tb * n + col] : 0.0f, ca0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float* row = o + (size_t)(16 * w + 4 * g + r) * K1;
                    row[col] = ca0[r];
                    if constexpr (!is_ae) {
                        const float ws_ = accB_[blk].c[c][r];
                        row[n + col] = ws_ - ca0[r];
                        row[2 * n + col] = ws_;
                    }
                }
            }
        }
        if constexpr (is_ae) {
#pragma unroll
            for (int bb = 0; bb < nfront; ++bb) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int col = n + H9 * bb + 16 * ((w + c) & 3) + j;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[(size_t)(16 * w + 4 * g + r) * K1 + col] = accB_[bb].c[c][r];
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int r = 0; r < 4; ++r) o[oW2 + (16 * w + 4 * g + r) * H9 + 16 * ((w + c) & 3) + j] = accW2_.c[c][r];
        }
        f4 sb1 = s1v, sb2 = sb2v;
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { sb1[r] += __shfl_xor(sb1[r], m, 64); sb2[r] += __shfl_xor(sb2[r], m, 64); }
        }
        if (j == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { o[oB1 + 16 * w + 4 * g + r] = sb1[r]; o[oB2 + 16 * w + 4 * g + r] = sb2[r]; }
        }
    };
    write_mlp(wp, 3 * n, std::false_type{}, std::integral_constant<int, NBLK>{}, S1, SB2, accF, accW2);
    if constexpr (DAE) write_mlp(wp + d.NP_de, n + H9 * NAE, std::true_type{}, std::integral_constant<int, NAE>{}, AS1, ASB2, accAF, accW2a);
    }
}


int np9(int k1) { return H9 * k1 + H9 + H9 * H9 + H9; }
bool two9(const psnode_mlp_f32& m, int in_dim) { return m.n_layers == 2 && m.in_dim == in_dim && m.out_dim[0] == H9 && m.out_dim[1] == H9; }
bool mis9(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; }
bool view9(const psnode_view_f32& v) { return v.ptr && !mis9(v.ptr) && v.stride_t % 4 == 0 && v.stride_b % 4 == 0; }
// per-lane offsets inside a row are 32-bit element offsets next to a scalar row base (psnode_common.h: sbase)
bool fits9(long long B, long long stride_b) { return stride_b >= 0 && (unsigned long long)B * (unsigned long long)stride_b + 64 < (1ull << 30); }   // byte offsets fit 32 bits
size_t pack9_floats(int nblk) { return (size_t)2 * NW9 * p9_regs(nblk, nblk) * 64; }
size_t lds9_bytes(int nlb) { return (size_t)(2 * NW9 * 64 + 2 * NW9 * NW9 * 64 + 4 * NW9 * 64 + nlb * 4 * NW9 * 64) * sizeof(f4) + (size_t)NW9 * SCR9 * sizeof(float); }

template <int METHOD, int NBE, bool DAE>
hipError_t launch9(const Bwd9Dev& d, const float* pde, const float* pae, hipStream_t s) {
    if constexpr (DAE) {
        if (d.a.sact) return launch9_roles(METHOD, NBE, d, pde, pae, s);      // saved activations: the two-role form (psnode_latent64_bwd_roles.hip)
    }
    // (the DAE's saved-activation instance IS the two-role kernel: its one-role twin is not instantiated)
    constexpr bool REC_SAVED = DAE;      // 4th parameter of the saved-activation instance: false (reads the saved rows) for the ODE only
    auto kern = (!DAE && d.a.sact) ? &latent64_backward_kernel<METHOD, NBE, DAE, REC_SAVED> : &latent64_backward_kernel<METHOD, NBE, DAE, true>;
    const size_t lds = lds9_bytes((DAE && !d.a.sact) ? 2 * NBE : 0);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)((d.a.B + 15) / 16)), dim3(256), lds, s, d, pde, pae);
    return hipGetLastError();
}

template <int METHOD>
hipError_t launch9_method(const Bwd9Dev& d, bool dae, const float* pde, const float* pae, hipStream_t s) {
    if (!dae) return launch9<METHOD, 1, false>(d, pde, pae, s);
    if (d.a.zd) return launch9<METHOD, 3, true>(d, pde, pae, s);
    return launch9<METHOD, 2, true>(d, pde, pae, s);
}

int run9(Bwd9Dev& d, bool dae, int nblk, const psnode_mlp_f32& de, const psnode_mlp_f32* ae, float* workspace, float* gp_de, float* gp_ae,
         hipStream_t s) {
    float* pack_de = workspace;
    float* pack_ae = workspace + pack9_floats(nblk) / 2;
    d.wpart = workspace + pack9_floats(nblk);
    d.NP_de = np9(3 * nblk * H9);
    d.NP_ae = dae ? np9((2 * nblk - 1) * H9) : 0;
    Pack9 p;
    p.ae = 0; p.nblk = nblk; p.nfront = nblk; p.k1 = 3 * nblk * H9;
    p.w1 = de.weight[0]; p.b1 = de.bias[0]; p.w2 = de.weight[1]; p.b2 = de.bias[1];
    p.out = pack_de;
    hipLaunchKernelGGL(pack9_kernel, dim3(64), dim3(256), 0, s, p);
    if (dae) {
        Pack9 pq = p;
        pq.ae = 1; pq.nfront = nblk - 1; pq.k1 = (2 * nblk - 1) * H9;
        pq.w1 = ae->weight[0]; pq.b1 = ae->bias[0]; pq.w2 = ae->weight[1]; pq.b2 = ae->bias[1];
        pq.out = pack_ae;
        hipLaunchKernelGGL(pack9_kernel, dim3(64), dim3(256), 0, s, pq);
    }
    if (hipGetLastError() != hipSuccess) return PSNODE_ERR_HIP;
    hipError_t e;
    switch (d.a.method) {
        case PSNODE_EULER: e = launch9_method<PSNODE_EULER>(d, dae, pack_de, pack_ae, s); break;
        case PSNODE_MIDPOINT: e = launch9_method<PSNODE_MIDPOINT>(d, dae, pack_de, pack_ae, s); break;
        default: e = launch9_method<PSNODE_RK4_38>(d, dae, pack_de, pack_ae, s); break;
    }
    if (e != hipSuccess) return PSNODE_ERR_HIP;
    const int nwg = (int)((d.a.B + 15) / 16);
    return launch_reduce_partials(d.wpart, gp_de, gp_ae, d.NP_de, d.NP_ae, nwg, s) == hipSuccess ? PSNODE_OK : PSNODE_ERR_HIP;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------- ODE
bool latent64_ode_bwd_shape_ok(const psnode_ode_bwd_args_f32* a) { return a->x_dim == H9 && a->z_dim == H9 && two9(a->de, 6 * H9); }
bool latent64_ode_bwd_ptrs_ok(const psnode_ode_bwd_args_f32* a) {
    if (mis9(a->all_initial) || mis9(a->xs) || mis9(a->grad_xs) || mis9(a->grad_x0) || mis9(a->grad_all_initial) || !view9(a->z)) return false;
    if (a->grad_z && mis9(a->grad_z)) return false;
    if (!fits9(a->B, a->t.stride_b) || !fits9(a->B, a->z.stride_b) || !fits9(a->B, (long long)H9 * (a->n_events > 0 ? a->n_events : 1))) return false;
    if (a->event_idx && !fits9(a->B, a->zj_stride_b)) return false;
    if (a->event_idx && (mis9(a->z_jump) || a->zj_stride_b % 4 || a->zj_stride_e % 4 || (a->grad_z_jump && mis9(a->grad_z_jump)))) return false;
    return true;
}
size_t latent64_ode_bwd_workspace_floats(long long B) { return pack9_floats(2) + (size_t)((B + 15) / 16) * np9(6 * H9) + 64; }
int latent64_ode_bwd_launch(const psnode_ode_bwd_args_f32* a, float* workspace, hipStream_t s) {
    Bwd9Dev d;
    memset(&d, 0, sizeof(d));
    d.a.method = a->method; d.a.xd = H9; d.a.zd = H9; d.a.T = a->T; d.a.B = a->B;
    d.a.t = ViewDev{a->t.ptr, a->t.stride_t, a->t.stride_b};
    d.a.z = ViewDev{a->z.ptr, a->z.stride_t, a->z.stride_b};
    d.a.a0 = a->all_initial; d.a.ev = a->event_idx; d.a.zj = a->z_jump; d.a.zjb = a->zj_stride_b; d.a.zje = a->zj_stride_e;
    d.xs = a->xs; d.gxs = a->grad_xs; d.gx0 = a->grad_x0; d.gz = a->grad_z; d.gzj = a->grad_z_jump; d.ga0 = a->grad_all_initial;
    d.n_events = a->n_events;
    d.a.sact = const_cast<float*>(a->saved_act); d.a.sxst = const_cast<float*>(a->saved_xstage);
    return run9(d, false, 2, a->de, nullptr, workspace, a->grad_params, nullptr, s);
}

// ---------------------------------------------------------------------------------------------------------------- DAE
bool latent64_dae_bwd_shape_ok(const psnode_dae_bwd_args_f32* a) {
    if (a->x_dim != H9 || a->v_dim != H9 || a->i_dim != H9 || (a->z_dim != H9 && a->z_dim != 0)) return false;
    const int nblk = a->z_dim ? 4 : 3;
    return two9(a->de, 3 * nblk * H9) && two9(a->ae, (2 * nblk - 1) * H9);
}
bool latent64_dae_bwd_ptrs_ok(const psnode_dae_bwd_args_f32* a) {
    if (mis9(a->all_initial) || mis9(a->xs) || mis9(a->is) || mis9(a->grad_xs) || mis9(a->grad_x_init) || mis9(a->grad_all_initial)) return false;
    if ((a->grad_is && mis9(a->grad_is)) || !view9(a->v) || (a->z_dim && !view9(a->z))) return false;
    if ((a->grad_z && mis9(a->grad_z)) || (a->grad_v && mis9(a->grad_v))) return false;
    if (!fits9(a->B, a->t.stride_b) || !fits9(a->B, a->v.stride_b) || (a->z_dim && !fits9(a->B, a->z.stride_b))
        || !fits9(a->B, (long long)H9 * (a->n_events > 0 ? a->n_events : 1))) return false;
    if (a->event_idx && (!fits9(a->B, a->vj_stride_b) || (a->z_dim && !fits9(a->B, a->zj_stride_b)))) return false;
    if (a->event_idx) {
        if (a->z_dim && (mis9(a->z_jump) || a->zj_stride_b % 4 || a->zj_stride_e % 4 || (a->grad_z_jump && mis9(a->grad_z_jump)))) return false;
        if (mis9(a->v_jump) || a->vj_stride_b % 4 || a->vj_stride_e % 4 || (a->grad_v_jump && mis9(a->grad_v_jump))) return false;
    }
    return true;
}
size_t latent64_dae_bwd_workspace_floats(const psnode_dae_bwd_args_f32* a) {
    const int nblk = a->z_dim ? 4 : 3;
    return pack9_floats(nblk) + (size_t)((a->B + 15) / 16) * (np9(3 * nblk * H9) + np9((2 * nblk - 1) * H9)) + 64;
}
int latent64_dae_bwd_launch(const psnode_dae_bwd_args_f32* a, float* workspace, hipStream_t s) {
    Bwd9Dev d;
    memset(&d, 0, sizeof(d));
    d.a.method = a->method; d.a.xd = H9; d.a.zd = a->z_dim; d.a.vd = H9; d.a.id = H9; d.a.T = a->T; d.a.B = a->B;
    d.a.t = ViewDev{a->t.ptr, a->t.stride_t, a->t.stride_b};
    d.a.z = ViewDev{a->z.ptr, a->z.stride_t, a->z.stride_b};
    d.a.v = ViewDev{a->v.ptr, a->v.stride_t, a->v.stride_b};
    d.a.a0 = a->all_initial; d.a.ev = a->event_idx;
    d.a.zj = a->z_jump; d.a.zjb = a->zj_stride_b; d.a.zje = a->zj_stride_e;
    d.a.vj = a->v_jump; d.a.vjb = a->vj_stride_b; d.a.vje = a->vj_stride_e;
    d.xs = a->xs; d.is_ = a->is; d.gxs = a->grad_xs; d.gis = a->grad_is;
    d.gx0 = a->grad_x_init; d.gz = a->grad_z; d.gv = a->grad_v; d.gzj = a->grad_z_jump; d.gvj = a->grad_v_jump; d.ga0 = a->grad_all_initial;
    d.n_events = a->n_events;
    d.a.sact = const_cast<float*>(a->saved_act); d.a.sxst = const_cast<float*>(a->saved_xstage);
    d.a.saeact = const_cast<float*>(a->saved_ae_act); d.a.sevact = const_cast<float*>(a->saved_ev_act); d.a.sevi = const_cast<float*>(a->saved_ev_i);
    return run9(d, true, a->z_dim ? 4 : 3, a->de, &a->ae, workspace, a->grad_params_de, a->grad_params_ae, s);
}

}  // namespace psnode
