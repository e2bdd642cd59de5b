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

}  // namespace psnode
