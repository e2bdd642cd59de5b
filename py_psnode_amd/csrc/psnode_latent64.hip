// K3c -- latent-space integrator of the direct_encode variants with hidden_dim = 64 (what neural_01_DAE_02_direct_encode.py
// ships with, :267; also ODE_02 run with --hidden 64):
//   ODE:  DE = Linear(6H,H) ELU Linear(H,H)                        state Xh[64], external Zh[64]
//   DAE:  DE = Linear(12H|9H,H) ELU Linear(H,H), AE = Linear(7H|5H,H) ELU Linear(H,H), blocks x | [z] | v | i of width 64
//
// One workgroup = 4 waves = one tile of 16 trajectories (as K1/K2).  Every matrix is a set of 64x64 blocks; each block
// lives in 16 VGPRs per lane in the "mid-layer" format of psnode_mfma.hip (chunk c <-> the 16 columns owned by wave
// (w+c)&3, k = 16w' + 4g + r), so a 64-wide vector that is distributed "16 dims per wave" (D layout) is consumed after one
// lane-linear all-gather through LDS.  Wave w owns hidden units AND state dims 16w..16w+15: the RK update is local, one
// all-gather per stage input and one per hidden vector.
// L1's `s - a0` and `s` column groups are folded:  Ws.s + Wd.(s - a0) = (Ws+Wd).s - Wd.a0, the a0 terms (+ bias, + the a0
// column group) become a per-trajectory constant; external blocks (z, v, and the algebraic i) a per-step constant.
// This halves L1's MFMAs; (Ws+Wd) is rounded once (relative 6e-8), far inside the 1e-5 gate (tests compare with the oracle).
#include <cstdlib>
#include <cstring>

#include "psnode_common.h"

namespace psnode {
namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int H64 = 64;
constexpr int NW64 = 4;

__device__ __forceinline__ f4 mf(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f4 elu4l(f4 v) { return elu_quad(v); }
// pack[wave][reg][lane]; every 64x64 block takes 16 registers: reg 4c+r = Blk[16w + i][16((w+c)&3) + 4g + r]
//   DE: F[NBLK] | B1(4) | W2(16) | B2(4) | A0[NBLK]            F_b = Ws_b + Wd_b,  A0_b = Wa0_b - Wd_b
//   AE: W[NBE]  | B1(4) | W2(16) | B2(4) | A0[NBLK]            blocks x | [z] | v right after the a0 group
struct Pack64 {
    int ae, nblk, nfront;    // nfront = blocks kept in registers in front (DE: NBLK, AE: NBE)
    int k1;                  // in_features
    const float *w1, *b1, *w2, *b2;
    float* out;
};

__global__ void pack64_kernel(const Pack64 p) {
    const int n = p.nblk * H64;
    const int B1 = 16 * p.nfront, W2 = B1 + 4, B2 = W2 + 16, A0 = B2 + 4, R = A0 + 16 * p.nblk;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < NW64 * R * 64; idx += gridDim.x * blockDim.x) {
        const int lane = idx & 63, reg = (idx >> 6) % R, w = (idx >> 6) / R, i = lane & 15, g = lane >> 4, u = 16 * w + i;
        const float* row = p.w1 + (size_t)u * p.k1;
        float v;
        auto col = [&](int kk) { return 16 * ((w + (kk >> 2)) & 3) + 4 * g + (kk & 3); };
        if (reg < B1) {
            const int blk = reg >> 4, c = H64 * blk + col(reg & 15);
            v = p.ae ? row[n + c] : row[2 * n + c] + row[n + c];
        } else if (reg < W2) {
            v = p.b1[16 * w + 4 * g + (reg - B1)];
        } else if (reg < B2) {
            v = p.w2[(size_t)u * H64 + col(reg - W2)];
        } else if (reg < A0) {
            v = p.b2[16 * w + 4 * g + (reg - B2)];
        } else {
            const int q = reg - A0, blk = q >> 4, c = H64 * blk + col(q & 15);
            v = p.ae ? row[c] : row[c] - row[n + c];
        }
        p.out[idx] = v;
    }
}

struct V4 { f4 v[4]; };   // a 64-wide vector in chunk layout: v[c] = dims 16((w+c)&3) + 4g + (0..3) of trajectory j

// SAVE (training forward): what autograd would keep for loss.backward() -- the hidden ELU outputs and stage inputs of the DE per
// (step, stage) (a.sact [T-1,S,B,64], a.sxst [T-1,S,B,64]), of the AE head per grid point (a.saeact [T,B,64]) and, for events taken,
// of the event-time head and its value i0 (a.sevact / a.sevi [nE,B,64]): K9's REC = false instance reads them instead of recomputing.
template <int METHOD, int NBE, bool DAE, bool SAVE = false>
__global__ __launch_bounds__(256) void latent64_kernel(const IntegrateDev a, const float* __restrict__ pack_de,
                                                        const float* __restrict__ pack_ae) {
    constexpr int NBLK = 1 + NBE, NZV = DAE ? NBE - 1 : NBE, n = H64 * NBLK;
    constexpr int RDE = 16 * NBLK + 24 + 16 * NBLK, RAE = 16 * NBE + 24 + 16 * NBLK;
    __shared__ f4 xbuf[2][NW64][64];
    const int l = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = l >> 4, j = l & 15;
    const long long b0 = (long long)blockIdx.x * 16;
    const bool valid = b0 + j < a.B;
    const long long b = valid ? b0 + j : a.B - 1;
    int coff[4];   // column offset of chunk c inside a 64-wide block
#pragma unroll
    for (int c = 0; c < 4; ++c) coff[c] = 16 * ((w + c) & 3) + 4 * g;

    // ---- weights -> registers
    const float* pw = pack_de + (size_t)w * RDE * 64 + l;
    float wf[NBLK][16], w2[16];
    f4 b1r, b2r;
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk)
#pragma unroll
        for (int k = 0; k < 16; ++k) wf[blk][k] = pw[(16 * blk + k) * 64];
#pragma unroll
    for (int k = 0; k < 16; ++k) w2[k] = pw[(16 * NBLK + 4 + k) * 64];
#pragma unroll
    for (int r = 0; r < 4; ++r) { b1r[r] = pw[(16 * NBLK + r) * 64]; b2r[r] = pw[(16 * NBLK + 20 + r) * 64]; }
    const float* pwa = pack_ae + (size_t)w * RAE * 64 + l;
    float af[DAE ? NBE : 1][16], aw2[16];
    f4 ab1r = {0.f, 0.f, 0.f, 0.f}, ab2r = ab1r;
    if constexpr (DAE) {
#pragma unroll
        for (int blk = 0; blk < NBE; ++blk)
#pragma unroll
            for (int k = 0; k < 16; ++k) af[blk][k] = pwa[(16 * blk + k) * 64];
#pragma unroll
        for (int k = 0; k < 16; ++k) aw2[k] = pwa[(16 * NBE + 4 + k) * 64];
#pragma unroll
        for (int r = 0; r < 4; ++r) { ab1r[r] = pwa[(16 * NBE + r) * 64]; ab2r[r] = pwa[(16 * NBE + 20 + r) * 64]; }
    }

    auto load_chunks = [&](const float* rowptr) -> V4 {   // 64 contiguous floats of one trajectory -> chunk layout
        V4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) o.v[c] = *reinterpret_cast<const f4*>(rowptr + coff[c]);
        return o;
    };
    // 16 MFMAs of one 64x64 block against a chunk-layout vector, two accumulator chains
    auto mm = [&](const float (&wr)[16], const V4& x, f4& accA, f4& accB) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            accA = mf(wr[4 * c + 0], x.v[c][0], accA); accB = mf(wr[4 * c + 1], x.v[c][1], accB);
            accA = mf(wr[4 * c + 2], x.v[c][2], accA); accB = mf(wr[4 * c + 3], x.v[c][3], accB);
        }
    };
    int p = 0;
    // all-gather of a vector distributed 16 dims per wave (D layout f4) -> chunk layout in every wave
    auto gather = [&](const f4 own) -> V4 {
        xbuf[p][w][l] = own;
        lds_barrier();
        V4 o;
        o.v[0] = own;
#pragma unroll
        for (int c = 1; c < 4; ++c) o.v[c] = xbuf[p][(w + c) & 3][l];
        __builtin_amdgcn_sched_barrier(0);
        p ^= 1;
        return o;
    };

    // ---- per-trajectory constants: c0 = b1 + sum_blk A0_blk . a0_blk   (DE and AE)
    f4 c0A = b1r, c0B = {0.f, 0.f, 0.f, 0.f}, caA = ab1r, caB = c0B;
    for (int blk = 0; blk < NBLK; ++blk) {
        const V4 a0v = load_chunks(a.a0 + b * n + H64 * blk);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = 4 * c + r;
                if (q & 1) c0B = mf(pw[(16 * NBLK + 24 + 16 * blk + q) * 64], a0v.v[c][r], c0B);
                else c0A = mf(pw[(16 * NBLK + 24 + 16 * blk + q) * 64], a0v.v[c][r], c0A);
                if constexpr (DAE) {
                    if (q & 1) caB = mf(pwa[(16 * NBE + 24 + 16 * blk + q) * 64], a0v.v[c][r], caB);
                    else caA = mf(pwa[(16 * NBE + 24 + 16 * blk + q) * 64], a0v.v[c][r], caA);
                }
            }
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
    struct Ext { V4 b[NZV > 0 ? NZV : 1]; };
    auto load_ext = [&](long long k, int ev, Ext& dst) {
#pragma unroll
        for (int s = 0; s < NZV; ++s) dst.b[s] = load_chunks((ev >= 0 ? jp[s] + ev * jse[s] : sp[s] + k * sst[s]));
    };

    // state of this wave: x dims 16w + 4g + (0..3)
    f4 x = *reinterpret_cast<const f4*>((DAE ? a.x_init + b * H64 : a.x.p + b * a.x.sb) + 16 * w + 4 * g);
    auto store_own = [&](float* base, long long k, const f4 v) {
        if (valid) *reinterpret_cast<f4*>(base + (k * a.B + b) * H64 + 16 * w + 4 * g) = v;
    };
    // one 64x64 layer on a gathered vector: own chunk before the barrier
    auto layer = [&](const float (&wr)[16], const f4 init, const f4 own) -> f4 {
        xbuf[p][w][l] = own;
        f4 accA = init, accB = {0.f, 0.f, 0.f, 0.f};
        accA = mf(wr[0], own[0], accA); accB = mf(wr[1], own[1], accB);
        accA = mf(wr[2], own[2], accA); accB = mf(wr[3], own[3], accB);
        __builtin_amdgcn_sched_barrier(0);
        lds_barrier();
        f4 vq[4];      // all three reads in flight before the first dependent MFMA
#pragma unroll
        for (int c = 1; c < 4; ++c) vq[c] = xbuf[p][(w + c) & 3][l];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 1; c < 4; ++c) {
            const f4 v = vq[c];
            accA = mf(wr[4 * c + 0], v[0], accA); accB = mf(wr[4 * c + 1], v[1], accB);
            accA = mf(wr[4 * c + 2], v[2], accA); accB = mf(wr[4 * c + 3], v[3], accB);
        }
        p ^= 1;
        return accA + accB;
    };
    // SAVE: running row pointers of (step, stage) -- this lane's own four dims
    const long long srow = SAVE ? a.B * H64 : 0;
    float* sa_run = SAVE ? a.sact + b * H64 + 16 * w + 4 * g : nullptr;
    float* sx_run = SAVE ? a.sxst + b * H64 + 16 * w + 4 * g : nullptr;
    auto keep_stage = [&](const f4 xs_own, const f4 h1) {
        if constexpr (SAVE) {
            if (valid) { *reinterpret_cast<f4*>(sx_run) = xs_own; *reinterpret_cast<f4*>(sa_run) = h1; }
            sx_run += srow; sa_run += srow;
        }
    };
    auto rhs_from_gathered = [&](const V4& xg, const f4 cz) -> f4 {      // stage whose input is already gathered (xg.v[0] = own dims)
        f4 accA = cz, accB = {0.f, 0.f, 0.f, 0.f};
        mm(wf[0], xg, accA, accB);
        const f4 h1 = elu4l(accA + accB);
        keep_stage(xg.v[0], h1);
        return layer(w2, b2r, h1);
    };
    auto rhs = [&](const f4 xs_own, const f4 cz) -> f4 {
        const f4 h1 = elu4l(layer(wf[0], cz, xs_own));
        keep_stage(xs_own, h1);
        return layer(w2, b2r, h1);
    };
    // `hrow` (SAVE): where this lane's four units of the head's hidden layer go, or null
    auto ae_eval = [&](const V4& xg, const Ext& zv, float* hrow) -> f4 {
        f4 accA = c0a, accB = {0.f, 0.f, 0.f, 0.f};
        if constexpr (DAE) {
            mm(af[0], xg, accA, accB);
#pragma unroll
            for (int s = 0; s < NZV; ++s) mm(af[1 + s], zv.b[s], accA, accB);
            const f4 h1 = elu4l(accA + accB);
            if constexpr (SAVE) { if (valid) *reinterpret_cast<f4*>(hrow) = h1; }
            return layer(aw2, ab2r, h1);
        }
        return accA;
    };
    auto head_row = [&](float* base, const long long r) -> float* { return SAVE ? base + (r * a.B + b) * H64 + 16 * w + 4 * g : nullptr; };

    store_own(a.xo, 0, x);
    V4 xg = gather(x);                      // gathered x_k: stage 1 of the step and the AE head both consume it
    f4 icur = {0.f, 0.f, 0.f, 0.f};
    if constexpr (DAE) {
        Ext zv0;
        load_ext(0, -1, zv0);
        icur = ae_eval(xg, zv0, head_row(a.saeact, 0));
        store_own(a.io, 0, icur);
    }
    if (nT < 2) return;

    float t_cur = tp[0], t_nxt = tp[tst];
    int lane_zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));
    // The prefetch of step k+1 is UNCONDITIONAL (indices clamped at the last step, the event index a raw table entry masked where it
    // is used): inside `if (more)` the loaded event index was a phi with -1, its copy into the loop-carried register -- and with it an
    // s_waitcnt vmcnt(0) on the nine loads just issued, a full HBM round trip -- sat in every step (round 3, found in the ISA).
    const bool has_ev = a.ev != nullptr;
    const int* evp = (has_ev ? a.ev : reinterpret_cast<const int*>(a.t.p)) + lane_zero;
    int ev_cur = has_ev ? a.ev[0] : -1;
    int ev_raw = evp[nT > 2 ? 1 : 0];
    Ext ext_nxt = {};
    load_ext(0, ev_cur, ext_nxt);

    for (long long k = 0; k + 1 < nT; ++k) {
        const float h_ = t_nxt - t_cur;
        t_cur = t_nxt;
        const Ext extv = ext_nxt;
        const int ev_now = ev_cur;
        const bool more = k + 2 < nT;
        {
            const long long kn = more ? k + 1 : k;                    // the last step re-reads its own inputs (unused)
            t_nxt = tp[(kn + 1) * tst];
            ev_cur = (has_ev && more) ? ev_raw : -1;
            load_ext(kn, ev_cur, ext_nxt);
            ev_raw = evp[k + 3 < nT ? k + 2 : 0];
        }
        if constexpr (DAE) {
            if (__builtin_amdgcn_readfirstlane(ev_now) >= 0) {   // i0 = g(x0; jumped z, v)  (my_solvers.py:108-110)
                Ext zvj;
                load_ext(k, ev_now, zvj);
                icur = ae_eval(xg, zvj, head_row(a.sevact, ev_now));
                if constexpr (SAVE) { if (valid) *reinterpret_cast<f4*>(head_row(a.sevi, ev_now)) = icur; }
            }
        }
        // per-step constant: c0 + sum over external blocks F_blk . block
        f4 czA = c0, czB = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NZV; ++s) mm(wf[1 + s], extv.b[s], czA, czB);
        if constexpr (DAE) {
            const V4 ig = gather(icur);
            mm(wf[NBLK - 1], ig, czA, czB);
        }
        const f4 cz = czA + czB;

        const f4 k1 = rhs_from_gathered(xg, cz);
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
        xg = gather(x);
        if constexpr (DAE) {   // i1 = g(x1; z[k+1], v[k+1]) with the RAW inputs: the prefetch of step k+1 unless that step jumps
            if (more && __builtin_amdgcn_readfirstlane(ev_cur) < 0) {
                icur = ae_eval(xg, ext_nxt, head_row(a.saeact, k + 1));
            } else {
                Ext zva;
                load_ext(k + 1, -1, zva);
                icur = ae_eval(xg, zva, head_row(a.saeact, k + 1));
            }
            store_own(a.io, k + 1, icur);
        }
    }
}

// =====================================================================================================================================
// K3g -- the whole DAE_Model.forward of neural_01_DAE_02_direct_encode.py:125-153 at its shipped hidden_dim 64 in ONE launch:
//   Xh0 = x_encoder(x0);  Zh = z_encoder(z), Vh = v_encoder(v), (Xh, Ih) = (x_encoder(x), i_encoder(i));  a0 = cat(Xh0, Zh[0], Vh[0], Ih[0])
//   (Xh_sol, Ih_sol) = integrate_DAE(latent DE / AE, jumps = encoders of the RAW z_jump / v_jump)
//   x_pred = x_decoder(Xh_sol), x_pred[0] = x0;  i_pred = i_decoder(Ih_sol);  x_re = x_decoder(Xh);  i_re = i_decoder(Ih)
// The latent tensors (six [T,B,64] arrays, ~1 KB per state-step through HBM on the K3b + K3c route) never reach memory: a step reads the
// RAW rows (t, z, v and -- for the reconstructions -- x, i) and writes the four decoded rows.  Same decomposition as K3c (4 waves per
// 16-trajectory tile, wave w owns hidden units and latent dims 16w..16w+15, 64x64 blocks in the mid-layer register format); per step the
// tile additionally runs
//   * the four encoders on the rows of grid point k+1: L1 (in <= 16 -> 64: 2..4 MFMAs), ELU, ONE exchange for the four hidden vectors,
//     L2 (one 64x64 block each), ONE exchange for the four outputs -- Zh | Vh feed the AE head now and the DE's per-step constant next
//     step, Xh | Ih go to the decoders (the reconstruction);
//   * the two decoders on four vectors (solution x, solution i, reconstruction x, reconstruction i): L1 = one block on the gathered
//     vector, ELU, L2 split-K over the waves' own units (4 MFMAs), ONE exchange in which wave d collects output d from the four waves,
//     adds the bias and stores it -- x_pred, i_pred, x_re, i_re are written by waves 0..3.
// => 13 exchanges and ~390 MFMAs per wave and RK4 step (K3c: 10 and 240; the K3b row kernels it replaces ran the same 150 MFMAs per
// 16 rows plus ~1 KB per row of HBM traffic).  Event steps encode the RAW jump rows on the spot.
struct EncW { float w1[4]; f4 b1; float w2[16]; f4 b2; };        // encoder in -> 64 -> 64 (in <= 16)
struct DecW { float w1[16]; f4 b1; float w2s[4]; f4 b2; };       // decoder 64 -> 64 -> out (out <= 16): L2 = this wave's K slice
constexpr int kEncRegs = 28, kDecRegs = 28, kEncDecRegs = 4 * kEncRegs + 2 * kDecRegs;

struct PackEncDec {
    const float *w1[6], *b1[6], *w2[6], *b2[6];    // x_enc, z_enc, v_enc, i_enc, x_dec, i_dec  (z_enc may be all-null: z_dim == 0)
    int in_dim[6], out_dim[6];
    float* out;                                     // [wave][reg][lane]
};

__global__ void pack_encdec_kernel(const PackEncDec p) {
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < NW64 * kEncDecRegs * 64; idx += gridDim.x * blockDim.x) {
        const int lane = idx & 63, reg = (idx >> 6) % kEncDecRegs, w = (idx >> 6) / kEncDecRegs, i = lane & 15, g = lane >> 4, u = 16 * w + i;
        const int m = reg / 28, q = reg % 28;
        auto col = [&](int kk) { return 16 * ((w + (kk >> 2)) & 3) + 4 * g + (kk & 3); };
        float v = 0.0f;
        if (p.w1[m]) {
            if (m < 4) {            // encoder
                if (q < 4) { const int c = 4 * q + g; if (c < p.in_dim[m]) v = p.w1[m][(size_t)u * p.in_dim[m] + c]; }
                else if (q < 8) v = p.b1[m][16 * w + 4 * g + (q - 4)];
                else if (q < 24) v = p.w2[m][(size_t)u * H64 + col(q - 8)];
                else v = p.b2[m][16 * w + 4 * g + (q - 24)];
            } else {                // decoder
                if (q < 16) v = p.w1[m][(size_t)u * H64 + col(q)];
                else if (q < 20) v = p.b1[m][16 * w + 4 * g + (q - 16)];
                else if (q < 24) { if (i < p.out_dim[m]) v = p.w2[m][(size_t)i * H64 + 16 * w + 4 * g + (q - 20)]; }
                else { const int o = 4 * g + (q - 24); if (o < p.out_dim[m]) v = p.b2[m][o]; }
            }
        }
        p.out[idx] = v;
    }
}

struct ModelDev {
    IntegrateDev a;          // method, T, B, raw dims xd | zd | vd | id, raw views t | x | z | v | i, ev, zj, vj, xo = x_pred, io = i_pred
    const float* x0;         // [B, xd]: Init_Func's output
    float* xre; long long xre_st, xre_sb;       // x_re[t * st + b * sb + d] or null
    float* ire; long long ire_st, ire_sb;
};

constexpr size_t kModel2Lds = 18 * sizeof(f4) * NW64 * 64;       // xbuf[2] + slotA[4] + slotB[4] + slotC[4] + slotJ[2] vectors of 4 KB
// K3g, two-role form (the default): 8 waves per 16-trajectory tile, 2 per SIMD, ONE barrier sequence.
//   waves 0..3  "chain":  exactly K3c's work -- per-step constant, RK stages, AE head -- 240 MFMAs and 10 exchanges per RK4 step;
//   waves 4..7  "rows":   the four encoders of grid point k+1 and the two decoders on the four vectors of grid point k -- 150 MFMAs per
//                         step, cut into ten pieces that sit between the chain's ten barriers.
// A barrier is a workgroup barrier, so both roles execute the SAME number of them per step; what differs is the work between them.
// Data crosses roles through LDS at those barriers: the rows waves publish Zh | Vh of grid point k+1 (slot B) before the chain's AE
// head needs them; the chain's own all-gathers of x_{k+1} and i_{k+1} are read by the rows waves as well (they decode them a step later).
// Why two roles: (1) the straight fusion above holds 500 registers per lane, and everything beyond 256 is an AGPR that costs a
// v_accvgpr_read per MFMA operand (318 per step); split in two, each role fits 256.  (2) K3c runs one wave per SIMD and idles through
// every exchange (42 % of its cycles): the rows wave on the same SIMD issues its independent MFMAs exactly there.
template <int METHOD, bool HASZ>
__global__ __launch_bounds__(512) void latent64_model2_kernel(const ModelDev md, const float* __restrict__ pack_de,
                                                               const float* __restrict__ pack_ae, const float* __restrict__ pack_ed) {
    constexpr int NBE = HASZ ? 3 : 2, NBLK = 1 + NBE;
    constexpr int RDE = 16 * NBLK + 24 + 16 * NBLK, RAE = 16 * NBE + 24 + 16 * NBLK;
    constexpr int S = METHOD == PSNODE_EULER ? 1 : (METHOD == PSNODE_MIDPOINT ? 2 : 4);
    constexpr int NBAR = 2 * S + 2;            // exchanges of the chain per step: (2S - 1) stage exchanges, gather x, AE hidden, gather i
    const IntegrateDev& a = md.a;
    typedef f4 Vec[NW64][64];                  // one 64-wide vector per trajectory, 16 dims per wave of a role
    extern __shared__ f4 smem2[];              // 72 KB (kModel2Lds)
    Vec* const xbuf = reinterpret_cast<Vec*>(smem2);        // [2]  the chain's exchanges (parity: an even number per step)
    Vec* const slotA = xbuf + 2;               // [4]  rows: encoder hidden vectors x | z | v | i (jump steps: z | v of the jump rows first)
    Vec* const slotB = slotA + 4;              // [4]  rows: encoder outputs Xh | Zh | Vh | Ih (own dims per rows-wave)
    Vec* const slotC = slotB + 4;              // [4]  rows: decoder partials [output][wave]
    Vec* const slotJ = slotC + 4;              // [2]  rows: encoder outputs of the jump rows z | v (read by the chain until barrier 1 of the step)
    const int l = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool chain = wave < NW64;
    const int w = wave & 3;                    // unit / dim slice 16w..16w+15 inside the role
    const int g = l >> 4, j = l & 15;
    const long long b0 = (long long)blockIdx.x * 16;
    const bool valid = b0 + j < a.B;
    const long long b = valid ? b0 + j : a.B - 1;
    const bool recon = md.xre != nullptr;
    const long long nT = a.T, tst = a.t.st;
    const int xd = a.xd, zd = a.zd, vd = a.vd, idm = a.id;
    const float* tp = a.t.p + b * a.t.sb;
    const bool has_ev = a.ev != nullptr;

    auto mm = [&](const float (&wr)[16], const V4& x, f4& accA, f4& accB) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            accA = mf(wr[4 * c + 0], x.v[c][0], accA); accB = mf(wr[4 * c + 1], x.v[c][1], accB);
            accA = mf(wr[4 * c + 2], x.v[c][2], accA); accB = mf(wr[4 * c + 3], x.v[c][3], accB);
        }
    };
    auto read_gathered = [&](const Vec& buf) -> V4 {      // chunk layout of a vector the four waves of a role published
        V4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) o.v[c] = buf[(w + c) & 3][l];
        return o;
    };
    int lane_zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));
    const int* evp = (has_ev ? a.ev : reinterpret_cast<const int*>(a.t.p)) + lane_zero;
    auto colv = [&](const float* row, const int m, const int width) -> float {
        const int c = 4 * m + g;
        const float v = row[c < width ? c : 0];
        return c < width ? v : 0.0f;
    };

    if (chain) {
        // ================================================================ chain role (K3c's work)
        const float* pw = pack_de + (size_t)w * RDE * 64 + l;
        float wf[NBLK][16], w2[16];
        f4 b1r, b2r;
#pragma unroll
        for (int blk = 0; blk < NBLK; ++blk)
#pragma unroll
            for (int k = 0; k < 16; ++k) wf[blk][k] = pw[(16 * blk + k) * 64];
#pragma unroll
        for (int k = 0; k < 16; ++k) w2[k] = pw[(16 * NBLK + 4 + k) * 64];
#pragma unroll
        for (int r = 0; r < 4; ++r) { b1r[r] = pw[(16 * NBLK + r) * 64]; b2r[r] = pw[(16 * NBLK + 20 + r) * 64]; }
        const float* pwa = pack_ae + (size_t)w * RAE * 64 + l;
        float af[NBE][16], aw2[16];
        f4 ab1r, ab2r;
#pragma unroll
        for (int blk = 0; blk < NBE; ++blk)
#pragma unroll
            for (int k = 0; k < 16; ++k) af[blk][k] = pwa[(16 * blk + k) * 64];
#pragma unroll
        for (int k = 0; k < 16; ++k) aw2[k] = pwa[(16 * NBE + 4 + k) * 64];
#pragma unroll
        for (int r = 0; r < 4; ++r) { ab1r[r] = pwa[(16 * NBE + r) * 64]; ab2r[r] = pwa[(16 * NBE + 20 + r) * 64]; }

        int p = 0;
        auto gather = [&](const f4 own) -> V4 {
            xbuf[p][w][l] = own;
            lds_barrier();
            V4 o;
            o.v[0] = own;
#pragma unroll
            for (int c = 1; c < 4; ++c) o.v[c] = xbuf[p][(w + c) & 3][l];
            __builtin_amdgcn_sched_barrier(0);
            p ^= 1;
            return o;
        };
        auto layer = [&](const float (&wr)[16], const f4 init, const f4 own) -> f4 {
            xbuf[p][w][l] = own;
            f4 accA = init, accB = {0.f, 0.f, 0.f, 0.f};
            accA = mf(wr[0], own[0], accA); accB = mf(wr[1], own[1], accB);
            accA = mf(wr[2], own[2], accA); accB = mf(wr[3], own[3], accB);
            __builtin_amdgcn_sched_barrier(0);
            lds_barrier();
            f4 vq[4];
#pragma unroll
            for (int c = 1; c < 4; ++c) vq[c] = xbuf[p][(w + c) & 3][l];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 1; c < 4; ++c) {
                const f4 v = vq[c];
                accA = mf(wr[4 * c + 0], v[0], accA); accB = mf(wr[4 * c + 1], v[1], accB);
                accA = mf(wr[4 * c + 2], v[2], accA); accB = mf(wr[4 * c + 3], v[3], accB);
            }
            p ^= 1;
            return accA + accB;
        };
        // ---- prologue (5 barriers, mirrored by the rows role): P1 hidden of enc(x0) | P2 Xh0 | P3 encoder hiddens of row 0 | P4 their outputs
        //      | then the AE head of grid point 0 (2 exchanges)
        lds_barrier();                                     // P1 (rows: hidden of x_encoder(x0) in slotA[0])
        lds_barrier();                                     // P2 (rows: Xh0 own dims in slotB[0])
        f4 x = slotB[0][w][l];
        V4 xg = read_gathered(slotB[0]);
        lds_barrier();                                     // P3 (rows: hiddens of row 0)
        lds_barrier();                                     // P4 (rows: outputs of row 0 in slotB[0..3])
        f4 c0A = b1r, c0B = {0.f, 0.f, 0.f, 0.f}, caA = ab1r, caB = c0B;
        auto a0_block = [&](const int blk, const V4& a0v) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = 4 * c + r;
                    if (q & 1) { c0B = mf(pw[(16 * NBLK + 24 + 16 * blk + q) * 64], a0v.v[c][r], c0B); caB = mf(pwa[(16 * NBE + 24 + 16 * blk + q) * 64], a0v.v[c][r], caB); }
                    else { c0A = mf(pw[(16 * NBLK + 24 + 16 * blk + q) * 64], a0v.v[c][r], c0A); caA = mf(pwa[(16 * NBE + 24 + 16 * blk + q) * 64], a0v.v[c][r], caA); }
                }
        };
        a0_block(0, xg);
        if constexpr (HASZ) a0_block(1, read_gathered(slotB[1]));
        a0_block(NBLK - 2, read_gathered(slotB[2]));
        a0_block(NBLK - 1, read_gathered(slotB[3]));
        const f4 c0 = c0A + c0B, c0a = caA + caB;
        // AE head on the encoder outputs of the grid point the rows waves published last (slotB[1] = Zh, slotB[2] = Vh)
        auto ae_eval = [&](const V4& xgv) -> f4 {
            f4 accA = c0a, accB = {0.f, 0.f, 0.f, 0.f};
            mm(af[0], xgv, accA, accB);
            if constexpr (HASZ) { const V4 zg = read_gathered(slotB[1]); mm(af[1], zg, accA, accB); }
            { const V4 vg = read_gathered(slotB[2]); mm(af[NBE - 1], vg, accA, accB); }
            return layer(aw2, ab2r, elu4l(accA + accB));
        };
        // the DE's i block of the per-step constant is formed right behind the gather of i (F_i . Ih: 4 registers carried instead of the
        // gathered vector's 16 -- the chain role lives at 256 registers per lane)
        auto fi_of = [&](const V4& igv) -> f4 {
            f4 accA = {0.f, 0.f, 0.f, 0.f}, accB = accA;
            mm(wf[NBLK - 1], igv, accA, accB);
            return accA + accB;
        };
        f4 czi = fi_of(gather(ae_eval(xg)));               // 2 exchanges: i_0 = g(x_0; z[0], v[0])  (my_solvers.py:95)
        if (nT < 2) { lds_barrier(); return; }             // (rows: partials of grid point 0)

        auto rhs_from_gathered = [&](const V4& xgv, const f4 cz) -> f4 {
            f4 accA = cz, accB = {0.f, 0.f, 0.f, 0.f};
            mm(wf[0], xgv, accA, accB);
            return layer(w2, b2r, elu4l(accA + accB));
        };
        auto rhs = [&](const f4 xs_own, const f4 cz) -> f4 { return layer(w2, b2r, elu4l(layer(wf[0], cz, xs_own))); };
        float t_cur = tp[0], t_nxt = tp[tst];
        int ev_cur = has_ev ? a.ev[0] : -1;
        int ev_raw = evp[nT > 2 ? 1 : 0];
        for (long long k = 0; k + 1 < nT; ++k) {
            const float h_ = t_nxt - t_cur;
            t_cur = t_nxt;
            const int ev_now = ev_cur;
            const bool more = k + 2 < nT;
            t_nxt = tp[(more ? k + 2 : k + 1) * tst];
            ev_cur = (has_ev && more) ? ev_raw : -1;
            ev_raw = evp[k + 3 < nT ? k + 2 : 0];
            // per-step constant from Zh | Vh of grid point k (slot B holds them since the previous step) or, at a jump, of the jump rows
            const bool jump = __builtin_amdgcn_readfirstlane(ev_now) >= 0;
            if (jump) {          // 4 barriers, mirrored: E1 hiddens of the jump rows | E2 their outputs (slotJ: own dims) | AE head (2)
                lds_barrier();
                lds_barrier();
                f4 accA = c0a, accB = {0.f, 0.f, 0.f, 0.f};
                mm(af[0], xg, accA, accB);
                if constexpr (HASZ) { const V4 zg = read_gathered(slotJ[0]); mm(af[1], zg, accA, accB); }
                { const V4 vg = read_gathered(slotJ[1]); mm(af[NBE - 1], vg, accA, accB); }
                czi = fi_of(gather(layer(aw2, ab2r, elu4l(accA + accB))));        // i0 = g(x0; jumped z, v)  (my_solvers.py:108-110)
            }
            f4 czA = c0, czB = {0.f, 0.f, 0.f, 0.f};
            if constexpr (HASZ) { const V4 zg = read_gathered(jump ? slotJ[0] : slotB[1]); mm(wf[1], zg, czA, czB); }
            { const V4 vg = read_gathered(jump ? slotJ[1] : slotB[2]); mm(wf[NBLK - 2], vg, czA, czB); }
            const f4 cz = (czA + czB) + czi;
            const f4 k1 = rhs_from_gathered(xg, cz);                       // 1 exchange
            if constexpr (METHOD == PSNODE_EULER) {
                x = x + h_ * k1;
            } else if constexpr (METHOD == PSNODE_MIDPOINT) {
                const f4 k2 = rhs(x + k1 * (0.5f * h_), cz);              // 2 exchanges per further stage
                x = x + h_ * k2;
            } else {
                const f4 k2 = rhs(x + h_ * k1 * kOneThird, cz);
                const f4 k3 = rhs(x + h_ * (k2 - k1 * kOneThird), cz);
                const f4 k4 = rhs(x + h_ * (k1 - k2 + k3), cz);
                x = x + (k1 + 3.0f * (k2 + k3) + k4) * h_ * 0.125f;
            }
            xg = gather(x);                                                // exchange 2S: the rows waves read it too
            czi = fi_of(gather(ae_eval(xg)));                              // exchanges 2S + 1, 2S + 2 (Zh | Vh of grid point k+1 from slot B)
        }
        lds_barrier();                                     // (rows: partials of the last grid point)
        return;
    }

    // ==================================================================== rows role: encoders one grid point ahead, decoders one behind
    const float* pe = pack_ed + (size_t)w * kEncDecRegs * 64 + l;
    auto load_enc = [&](const int m, EncW& e) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { e.w1[q] = (m == 0 || q < 2) ? pe[(28 * m + q) * 64] : 0.0f; e.b1[q] = pe[(28 * m + 4 + q) * 64]; e.b2[q] = pe[(28 * m + 24 + q) * 64]; }
#pragma unroll
        for (int q = 0; q < 16; ++q) e.w2[q] = pe[(28 * m + 8 + q) * 64];
    };
    auto load_dec = [&](const int m, DecW& d) {
#pragma unroll
        for (int q = 0; q < 16; ++q) d.w1[q] = pe[(28 * m + q) * 64];
#pragma unroll
        for (int q = 0; q < 4; ++q) { d.b1[q] = pe[(28 * m + 16 + q) * 64]; d.w2s[q] = pe[(28 * m + 20 + q) * 64]; d.b2[q] = pe[(28 * m + 24 + q) * 64]; }
    };
    EncW ex, ez, evv, ei;
    DecW dx, di;
    load_enc(0, ex); load_enc(1, ez); load_enc(2, evv); load_enc(3, ei);
    load_dec(4, dx); load_dec(5, di);
    struct Raw { float x[4], z[2], v[2], i[2]; };
    const float* xp = a.x.p ? a.x.p + b * a.x.sb : tp;
    const float* zp = HASZ ? a.z.p + b * a.z.sb : tp;
    const float* vp = a.v.p + b * a.v.sb;
    const float* ip = a.i.p + b * a.i.sb;
    const long long xst = a.x.p ? a.x.st : 0, zst = HASZ ? a.z.st : 0, vst = a.v.st, ist = a.i.st;
    auto load_raw = [&](const long long k, Raw& r) {
#pragma unroll
        for (int m = 0; m < 4; ++m) r.x[m] = recon ? colv(xp + k * xst, m, xd) : 0.0f;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            r.z[m] = HASZ ? colv(zp + k * zst, m, zd) : 0.0f;
            r.v[m] = colv(vp + k * vst, m, vd);
            r.i[m] = colv(ip + k * ist, m, idm);
        }
    };
    auto enc_l1x = [&](const EncW& e, const float* raw, const int nm) -> f4 {       // x: up to 16 columns
        f4 acc = e.b1;
#pragma unroll
        for (int m = 0; m < 4; ++m) if (m < nm) acc = mf(e.w1[m], raw[m], acc);
        return elu4l(acc);
    };
    auto enc_l1 = [&](const EncW& e, const float* raw, const int nm) -> f4 {        // z | v | i: up to 8 columns (two L1 registers held)
        f4 acc = mf(e.w1[0], raw[0], e.b1);
        if (nm > 1) acc = mf(e.w1[1], raw[1], acc);
        return elu4l(acc);
    };
    auto enc_l2 = [&](const EncW& e, const V4& h) -> f4 {
        f4 accA = e.b2, accB = {0.f, 0.f, 0.f, 0.f};
        mm(e.w2, h, accA, accB);
        return accA + accB;
    };
    const int nmx = (xd + 3) >> 2, nmz = (zd + 3) >> 2, nmv = (vd + 3) >> 2, nmi = (idm + 3) >> 2;
    // output d of this tile is written by rows-wave d: 0 x_pred, 1 i_pred, 2 x_re, 3 i_re
    const bool is_i = (w & 1) != 0, is_re = w >= 2;
    float* const obase = is_re ? (is_i ? md.ire : md.xre) : (is_i ? a.io : a.xo);
    const int odim = is_i ? idm : xd;
    const long long ost = is_re ? (is_i ? md.ire_st : md.xre_st) : a.B * odim, osb = is_re ? (is_i ? md.ire_sb : md.xre_sb) : odim;
    const f4 ob2 = is_i ? di.b2 : dx.b2;
    const bool storing = valid && (recon || !is_re);
    float* orow = obase + (storing ? b * osb + 4 * g : 0);
    auto dec_part = [&](const DecW& d, const V4& vin) -> f4 {
        f4 accA = d.b1, accB = {0.f, 0.f, 0.f, 0.f};
        mm(d.w1, vin, accA, accB);
        const f4 h = elu4l(accA + accB);
        f4 pa = mf(d.w2s[0], h[0], f4{0.f, 0.f, 0.f, 0.f}), pb = mf(d.w2s[1], h[1], f4{0.f, 0.f, 0.f, 0.f});
        pa = mf(d.w2s[2], h[2], pa); pb = mf(d.w2s[3], h[3], pb);
        return pa + pb;
    };
    auto reduce_store = [&](const bool first) {       // after the barrier behind the slotC writes
        f4 o = ob2;
#pragma unroll
        for (int c = 0; c < 4; ++c) o += slotC[w][c][l];
        if (first && w == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int d = 4 * g + r; o[r] = md.x0[b * xd + (d < xd ? d : 0)]; }     // x_pred[0] = x0 (:150)
        }
        if (storing) {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (4 * g + r < odim) orow[r] = o[r];
        }
        orow += ost;
    };

    // ---- prologue (mirrors the chain's): Xh0, the encoders of row 0, then row 0's decoders inside the chain's two AE exchanges
    Raw r0;
    load_raw(0, r0);
    {
        float x0raw[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) x0raw[m] = colv(md.x0 + b * xd, m, xd);
        slotA[0][w][l] = enc_l1x(ex, x0raw, nmx);
    }
    lds_barrier();                                         // P1
    slotB[0][w][l] = enc_l2(ex, read_gathered(slotA[0]));
    lds_barrier();                                         // P2: Xh0
    const V4 xh0g = read_gathered(slotB[0]);
    // hiddens of row 0
    slotA[0][w][l] = recon ? enc_l1x(ex, r0.x, nmx) : f4{0.f, 0.f, 0.f, 0.f};
    if constexpr (HASZ) slotA[1][w][l] = enc_l1(ez, r0.z, nmz);
    slotA[2][w][l] = enc_l1(evv, r0.v, nmv);
    slotA[3][w][l] = enc_l1(ei, r0.i, nmi);
    lds_barrier();                                         // P3
    {
        const f4 ox = recon ? enc_l2(ex, read_gathered(slotA[0])) : f4{0.f, 0.f, 0.f, 0.f};
        f4 oz = {0.f, 0.f, 0.f, 0.f};
        if constexpr (HASZ) oz = enc_l2(ez, read_gathered(slotA[1]));
        const f4 ov = enc_l2(evv, read_gathered(slotA[2]));
        const f4 oi = enc_l2(ei, read_gathered(slotA[3]));
        slotB[0][w][l] = ox; slotB[1][w][l] = oz; slotB[2][w][l] = ov; slotB[3][w][l] = oi;
    }
    lds_barrier();                                         // P4: outputs of row 0
    // decode grid point 0 (its solution row is x0 itself: output 0 is overwritten by reduce_store(first))
    slotC[0][w][l] = dec_part(dx, xh0g);
    if (recon) { slotC[2][w][l] = dec_part(dx, read_gathered(slotB[0])); slotC[3][w][l] = dec_part(di, read_gathered(slotB[3])); }
    lds_barrier();                                         // chain: AE hidden of grid point 0
    lds_barrier();                                         // chain: gather i_0 (the chain's second exchange of the prologue: parity 1)
    slotC[1][w][l] = dec_part(di, read_gathered(xbuf[1]));
    if (nT < 2) { lds_barrier(); reduce_store(true); return; }

    int ev_cur = has_ev ? a.ev[0] : -1;
    int ev_raw = evp[nT > 2 ? 1 : 0];
    Raw rn;
    load_raw(1, rn);
    bool first = true;
    // Per step k the rows waves (a) finish the decode of grid point k (partials were written last step: barrier 1 of this step is
    // the one behind them), (b) encode grid point k+1, (c) start the decode of grid point k+1 behind the chain's gathers.
    for (long long k = 0; k + 1 < nT; ++k) {
        __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): the rows requested a step ago
        const Raw rk1 = rn;
        const int ev_now = ev_cur;
        const bool more = k + 2 < nT;
        ev_cur = (has_ev && more) ? ev_raw : -1;
        ev_raw = evp[k + 3 < nT ? k + 2 : 0];
        if (__builtin_amdgcn_readfirstlane(ev_now) >= 0) {   // jump rows: encoded on the spot (hiddens through slotA, free here) into slotJ, 4 barriers
            float rz[2], rv[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                rz[m] = HASZ ? colv(a.zj + b * a.zjb + (long long)ev_now * a.zje, m, zd) : 0.0f;
                rv[m] = colv(a.vj + b * a.vjb + (long long)ev_now * a.vje, m, vd);
            }
            if constexpr (HASZ) slotA[1][w][l] = enc_l1(ez, rz, nmz);
            slotA[2][w][l] = enc_l1(evv, rv, nmv);
            lds_barrier();                                 // E1
            f4 oz = {0.f, 0.f, 0.f, 0.f};
            if constexpr (HASZ) oz = enc_l2(ez, read_gathered(slotA[1]));
            const f4 ov = enc_l2(evv, read_gathered(slotA[2]));
            slotJ[0][w][l] = oz; slotJ[1][w][l] = ov;
            lds_barrier();                                 // E2
            lds_barrier();                                 // chain: AE hidden
            lds_barrier();                                 // chain: gather i0
        }
        // barrier 1 (chain: stage-1 hidden): encoder hiddens of grid point k+1 out, decode of grid point k completes behind it
        load_raw(more ? k + 2 : k + 1, rn);
        slotA[0][w][l] = recon ? enc_l1x(ex, rk1.x, nmx) : f4{0.f, 0.f, 0.f, 0.f};
        if constexpr (HASZ) slotA[1][w][l] = enc_l1(ez, rk1.z, nmz);
        slotA[2][w][l] = enc_l1(evv, rk1.v, nmv);
        slotA[3][w][l] = recon ? enc_l1(ei, rk1.i, nmi) : f4{0.f, 0.f, 0.f, 0.f};
        lds_barrier();                                     // 1
        reduce_store(first);
        first = false;
        f4 oz = {0.f, 0.f, 0.f, 0.f}, ov, ox = oz, oi = oz;
        if constexpr (S == 1) {
            // Euler: 4 barriers per step -- everything between 1 and 2
            if constexpr (HASZ) oz = enc_l2(ez, read_gathered(slotA[1]));
            ov = enc_l2(evv, read_gathered(slotA[2]));
            if (recon) { ox = enc_l2(ex, read_gathered(slotA[0])); oi = enc_l2(ei, read_gathered(slotA[3])); }
        } else {
            if constexpr (HASZ) oz = enc_l2(ez, read_gathered(slotA[1]));
            lds_barrier();                                 // 2
            ov = enc_l2(evv, read_gathered(slotA[2]));
            lds_barrier();                                 // 3
            if (recon) ox = enc_l2(ex, read_gathered(slotA[0]));
            if constexpr (S == 4) lds_barrier();           // 4
            if (recon) oi = enc_l2(ei, read_gathered(slotA[3]));
            if constexpr (S == 4) { lds_barrier(); lds_barrier(); lds_barrier(); }      // 5, 6, 7
        }
        // the chain last read Zh | Vh of grid point k (slot B) for its per-step constant, in front of barrier 1: free to overwrite
        slotB[0][w][l] = ox; slotB[1][w][l] = oz; slotB[2][w][l] = ov; slotB[3][w][l] = oi;
        lds_barrier();                                     // 2S: chain gathers x_{k+1} -- exchange 2S of an even count: parity 1
        slotC[0][w][l] = dec_part(dx, read_gathered(xbuf[1]));
        if (recon) slotC[2][w][l] = dec_part(dx, read_gathered(slotB[0]));
        lds_barrier();                                     // 2S + 1: chain's AE hidden
        if (recon) slotC[3][w][l] = dec_part(di, read_gathered(slotB[3]));
        lds_barrier();                                     // 2S + 2: chain gathers i_{k+1} (parity 1 again)
        slotC[1][w][l] = dec_part(di, read_gathered(xbuf[1]));
    }
    lds_barrier();                                         // behind the last partials
    reduce_store(first);
}

bool two64(const MlpDev& m, int in_dim) { return m.n_layers == 2 && m.in_dim == in_dim && m.out_dim[0] == H64 && m.out_dim[1] == H64; }
bool al4(const ViewDev& v) { return v.p && (reinterpret_cast<uintptr_t>(v.p) & 15) == 0 && v.st % 4 == 0 && v.sb % 4 == 0; }

template <int METHOD>
hipError_t launch64_method(const IntegrateDev& a, bool dae, const float* pde, const float* pae, hipStream_t s) {
    const dim3 grid((unsigned)((a.B + 15) / 16)), block(256);
    if (a.sact) {       // training forward
        if (!dae) hipLaunchKernelGGL((latent64_kernel<METHOD, 1, false, true>), grid, block, 0, s, a, pde, pae);
        else if (a.zd) hipLaunchKernelGGL((latent64_kernel<METHOD, 3, true, true>), grid, block, 0, s, a, pde, pae);
        else hipLaunchKernelGGL((latent64_kernel<METHOD, 2, true, true>), grid, block, 0, s, a, pde, pae);
        return hipGetLastError();
    }
    if (!dae) hipLaunchKernelGGL((latent64_kernel<METHOD, 1, false>), grid, block, 0, s, a, pde, pae);
    else if (a.zd) hipLaunchKernelGGL((latent64_kernel<METHOD, 3, true>), grid, block, 0, s, a, pde, pae);
    else hipLaunchKernelGGL((latent64_kernel<METHOD, 2, true>), grid, block, 0, s, a, pde, pae);
    return hipGetLastError();
}

}  // namespace

bool latent64_shape_ok(const IntegrateDev& a, bool dae) {
    if (a.flags) return false;
    if (!dae) return a.xd == H64 && a.zd == H64 && two64(a.de, 6 * H64);
    if (a.xd != H64 || a.vd != H64 || a.id != H64 || (a.zd != H64 && a.zd != 0)) return false;
    const int nblk = a.zd ? 4 : 3;
    return two64(a.de, 3 * nblk * H64) && two64(a.ae, (2 * nblk - 1) * H64);
}

bool latent64_ptrs_ok(const IntegrateDev& a, bool dae) {
    auto mis = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
    if (mis(a.a0) || mis(a.xo)) return false;
    if (!dae) return al4(a.x) && al4(a.z) && (!a.ev || (!mis(a.zj) && a.zjb % 4 == 0 && a.zje % 4 == 0));
    if (mis(a.x_init) || mis(a.io) || !al4(a.v) || (a.zd && !al4(a.z))) return false;
    if (a.ev) {
        if (a.zd && (mis(a.zj) || a.zjb % 4 || a.zje % 4)) return false;
        if (mis(a.vj) || a.vjb % 4 || a.vje % 4) return false;
    }
    return true;
}

size_t latent64_pack_floats() { return 2 * (size_t)NW64 * (16 * 4 + 24 + 16 * 4) * 64; }

hipError_t launch_latent64(const IntegrateDev& a, bool dae, float* pack, hipStream_t stream) {
    const int nblk = dae ? (a.zd ? 4 : 3) : 2;
    Pack64 p;
    p.ae = 0; p.nblk = nblk; p.nfront = nblk; p.k1 = 3 * nblk * H64;
    p.w1 = a.de.w[0]; p.b1 = a.de.bias[0]; p.w2 = a.de.w[1]; p.b2 = a.de.bias[1];
    p.out = pack;
    hipLaunchKernelGGL(pack64_kernel, dim3(32), dim3(256), 0, stream, p);
    float* pack_ae = pack + latent64_pack_floats() / 2;
    if (dae) {
        Pack64 q = p;
        q.ae = 1; q.nfront = nblk - 1; q.k1 = (2 * nblk - 1) * H64;
        q.w1 = a.ae.w[0]; q.b1 = a.ae.bias[0]; q.w2 = a.ae.w[1]; q.b2 = a.ae.bias[1];
        q.out = pack_ae;
        hipLaunchKernelGGL(pack64_kernel, dim3(32), dim3(256), 0, stream, q);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    switch (a.method) {
        case PSNODE_EULER: return launch64_method<PSNODE_EULER>(a, dae, pack, pack_ae, stream);
        case PSNODE_MIDPOINT: return launch64_method<PSNODE_MIDPOINT>(a, dae, pack, pack_ae, stream);
        default: return launch64_method<PSNODE_RK4_38>(a, dae, pack, pack_ae, stream);
    }
}


namespace {
template <int METHOD>
hipError_t launch_model_method(const ModelDev& md, const float* pde, const float* pae, const float* ped, hipStream_t s) {
    const dim3 grid((unsigned)((md.a.B + 15) / 16));
    auto go = [&](auto kern) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kModel2Lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, grid, dim3(512), kModel2Lds, s, md, pde, pae, ped);
        return hipGetLastError();
    };
    return md.a.zd ? go(&latent64_model2_kernel<METHOD, true>) : go(&latent64_model2_kernel<METHOD, false>);
}
bool enc_ok(const psnode_mlp_f32& m, int in, int maxin) {
    return m.n_layers == 2 && m.in_dim == in && in >= 1 && in <= maxin && m.out_dim[0] == H64 && m.out_dim[1] == H64;
}
bool dec_ok(const psnode_mlp_f32& m, int out) { return m.n_layers == 2 && m.in_dim == H64 && m.out_dim[0] == H64 && m.out_dim[1] == out && out >= 1 && out <= 16; }
bool lat_ok(const psnode_mlp_f32& m, int in) { return m.n_layers == 2 && m.in_dim == in && m.out_dim[0] == H64 && m.out_dim[1] == H64; }
}  // namespace
}  // namespace psnode

using namespace psnode;

extern "C" {

int32_t psnode_dae_encoded_supported(const psnode_dae_encoded_args_f32* p) {
    if (!p) return 0;
    const int nblk = p->z_dim ? 4 : 3;
    if (p->z_dim < 0 || p->z_dim > 8) return 0;
    return enc_ok(p->x_encoder, p->x_dim, 16) && (p->z_dim == 0 || enc_ok(p->z_encoder, p->z_dim, 8)) && enc_ok(p->v_encoder, p->v_dim, 8) &&
           enc_ok(p->i_encoder, p->i_dim, 8) && dec_ok(p->x_decoder, p->x_dim) && dec_ok(p->i_decoder, p->i_dim) &&
           lat_ok(p->de, 3 * nblk * H64) && lat_ok(p->ae, (2 * nblk - 1) * H64);
}

size_t psnode_dae_encoded_workspace_bytes(const psnode_dae_encoded_args_f32* p) {
    if (!p) return 0;
    return (latent64_pack_floats() + (size_t)NW64 * kEncDecRegs * 64 + 64) * sizeof(float);
}

int32_t psnode_dae_encoded_integrate_f32(const psnode_dae_encoded_args_f32* p, void* workspace, size_t workspace_bytes, void* stream) {
    if (!p) return PSNODE_ERR_NULL;
    if (p->method < PSNODE_EULER || p->method > PSNODE_RK4_38) return PSNODE_ERR_METHOD;
    if (p->T < 1 || p->B < 1 || p->x_dim < 1 || p->v_dim < 1 || p->i_dim < 1 || p->z_dim < 0) return PSNODE_ERR_DIMS;
    if (!p->t.ptr || !p->v.ptr || !p->i.ptr || (p->z_dim && !p->z.ptr) || !p->x0 || !p->x_pred || !p->i_pred) return PSNODE_ERR_NULL;
    if ((p->x_re == nullptr) != (p->i_re == nullptr)) return PSNODE_ERR_NULL;       // both reconstructions or neither
    if (p->x_re && !p->x.ptr) return PSNODE_ERR_NULL;
    if (p->event_idx && (!p->v_jump || (p->z_dim && !p->z_jump))) return PSNODE_ERR_NULL;
    if (!psnode_dae_encoded_supported(p)) return PSNODE_ERR_UNSUPPORTED;
    if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 255) || workspace_bytes < psnode_dae_encoded_workspace_bytes(p)) return PSNODE_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* pack = static_cast<float*>(workspace);
    ModelDev md;
    memset(&md, 0, sizeof(md));
    IntegrateDev& a = md.a;
    a.method = p->method; a.xd = p->x_dim; a.zd = p->z_dim; a.vd = p->v_dim; a.id = p->i_dim; a.T = p->T; a.B = p->B;
    auto bind = [](const psnode_mlp_f32& m, MlpDev& d) {
        d.n_layers = m.n_layers; d.in_dim = m.in_dim;
        for (int l = 0; l < m.n_layers; ++l) { d.out_dim[l] = m.out_dim[l]; d.w[l] = m.weight[l]; d.bias[l] = m.bias[l]; }
    };
    bind(p->de, a.de); bind(p->ae, a.ae);
    a.t = ViewDev{p->t.ptr, p->t.stride_t, p->t.stride_b};
    a.x = ViewDev{p->x.ptr, p->x.stride_t, p->x.stride_b};
    a.z = ViewDev{p->z.ptr, p->z.stride_t, p->z.stride_b};
    a.v = ViewDev{p->v.ptr, p->v.stride_t, p->v.stride_b};
    a.i = ViewDev{p->i.ptr, p->i.stride_t, p->i.stride_b};
    a.ev = p->event_idx; a.zj = p->z_jump; a.zjb = p->zj_stride_b; a.zje = p->zj_stride_e;
    a.vj = p->v_jump; a.vjb = p->vj_stride_b; a.vje = p->vj_stride_e;
    a.xo = p->x_pred; a.io = p->i_pred;
    md.x0 = p->x0;
    md.xre = p->x_re; md.xre_st = p->xre_stride_t; md.xre_sb = p->xre_stride_b;
    md.ire = p->i_re; md.ire_st = p->ire_stride_t; md.ire_sb = p->ire_stride_b;
    // packed images: latent DE | AE (as K3c), then the encoders / decoders
    const int nblk = a.zd ? 4 : 3;
    Pack64 pk;
    pk.ae = 0; pk.nblk = nblk; pk.nfront = nblk; pk.k1 = 3 * nblk * H64;
    pk.w1 = a.de.w[0]; pk.b1 = a.de.bias[0]; pk.w2 = a.de.w[1]; pk.b2 = a.de.bias[1];
    pk.out = pack;
    hipLaunchKernelGGL(pack64_kernel, dim3(32), dim3(256), 0, s, pk);
    float* pack_ae = pack + latent64_pack_floats() / 2;
    Pack64 qk = pk;
    qk.ae = 1; qk.nfront = nblk - 1; qk.k1 = (2 * nblk - 1) * H64;
    qk.w1 = a.ae.w[0]; qk.b1 = a.ae.bias[0]; qk.w2 = a.ae.w[1]; qk.b2 = a.ae.bias[1];
    qk.out = pack_ae;
    hipLaunchKernelGGL(pack64_kernel, dim3(32), dim3(256), 0, s, qk);
    float* pack_ed = pack + latent64_pack_floats();
    PackEncDec pe;
    memset(&pe, 0, sizeof(pe));
    const psnode_mlp_f32* ms[6] = {&p->x_encoder, p->z_dim ? &p->z_encoder : nullptr, &p->v_encoder, &p->i_encoder, &p->x_decoder, &p->i_decoder};
    for (int m = 0; m < 6; ++m) {
        if (!ms[m]) continue;
        pe.w1[m] = ms[m]->weight[0]; pe.b1[m] = ms[m]->bias[0]; pe.w2[m] = ms[m]->weight[1]; pe.b2[m] = ms[m]->bias[1];
        pe.in_dim[m] = ms[m]->in_dim; pe.out_dim[m] = ms[m]->out_dim[1];
        if (!pe.w1[m] || !pe.b1[m] || !pe.w2[m] || !pe.b2[m]) return PSNODE_ERR_NULL;
    }
    pe.out = pack_ed;
    hipLaunchKernelGGL(pack_encdec_kernel, dim3(32), dim3(256), 0, s, pe);
    if (hipGetLastError() != hipSuccess) return PSNODE_ERR_HIP;
    hipError_t e;
    switch (a.method) {
        case PSNODE_EULER: e = launch_model_method<PSNODE_EULER>(md, pack, pack_ae, pack_ed, s); break;
        case PSNODE_MIDPOINT: e = launch_model_method<PSNODE_MIDPOINT>(md, pack, pack_ae, pack_ed, s); break;
        default: e = launch_model_method<PSNODE_RK4_38>(md, pack, pack_ae, pack_ed, s); break;
    }
    return e == hipSuccess ? PSNODE_OK : PSNODE_ERR_HIP;
}

}  // extern "C"
