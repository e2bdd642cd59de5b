// K9, two-role form -- see the comment at the kernel.  Its own translation unit: built with -amdgpu-mfma-vgpr-form (the one-role K9 / K7
// need the AGPR form for their ~450 live values; this kernel has 256 registers per wave and wants its accumulators in VGPRs).
#define PSNODE_ELU_LITERALS
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "psnode_latent64_bwd.h"

namespace psnode {
namespace {

// =====================================================================================================================================
// K9 two-role form (round 4; DAE, saved activations): 4 CHAIN waves + 4 GRADIENT waves per tile of 16 trajectories.
// Half of K9's MFMAs are the weight-gradient outer products dBlk[own units][:] += d (x) v, and every `v` of them -- the saved hidden
// layers and stage inputs, the x / z|v / i rows, the saved AE hidden layers -- is DATA FROM MEMORY, independent of the adjoint.  So the
// chain waves only sweep the adjoint (the transposed blocks + their reduce-scatters, 240 MFMAs per RK4 step) and drop each `d` (gk, delta1,
// their per-step sum, the head's gi and delta1) into a padded tile; the gradient wave with the same 16 units loads the `v` rows itself,
// transposes them in a private tile and publishes them for the other gradient waves (two parities of four slots, one phase ahead), reads
// the chain's `d` tile TRANSPOSED and accumulates the 160 registers of block gradients.  Both roles run the same barrier sequence (the
// chain's reduce-scatters): per grid point 1 + NBE for the head, 2 per stage, NBE for the external blocks, 1 + NBE per event taken.
// The chain keeps no accumulator, publishes nothing, transposes nothing; its per-iteration inputs are requested an iteration ahead.
template <int METHOD, int NBE>
__global__ __launch_bounds__(512) void latent64_backward_roles_kernel(const Bwd9Dev d, const float* __restrict__ pack_de,
                                                                       const float* __restrict__ pack_ae) {
    constexpr int S = rk_stages(METHOD);
    constexpr int NBLK = 1 + NBE, NZV = NBE - 1, n = H9 * NBLK, NAE = NBE;
    constexpr int LQ_AFT = 0, LQ_AW2T = NAE, LQ_DFT = NAE + 1;
    constexpr int D_B1 = 16 * NBLK, D_W2 = D_B1 + 4, D_B2 = D_W2 + 16, D_A0 = D_B2 + 4, D_FT = D_A0 + 16 * NBLK, D_W2T = D_FT + 16 * NBLK,
                  D_A0T = D_W2T + 16, D_R = D_A0T + 16 * NBLK;
    constexpr int A_B1 = 16 * NAE, A_W2 = A_B1 + 4, A_B2 = A_W2 + 16, A_A0 = A_B2 + 4, A_FT = A_A0 + 16 * NBLK, A_W2T = A_FT + 16 * NAE,
                  A_A0T = A_W2T + 16, A_R = A_A0T + 16 * NBLK;
    (void)D_W2; (void)A_W2; (void)A_B1; (void)A_B2; (void)D_B1; (void)D_B2; (void)A_A0; (void)D_A0;
    const IntegrateDev& a = d.a;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    f4* rsbuf = reinterpret_cast<f4*>(lds);                 // [2][4][4][64]        reduce-scatter (chain)
    f4* pub = rsbuf + 2 * NW9 * NW9 * 64;                   // [3 parities][4 slots][4][64]   transposed `v` tiles (gradient waves); parity 2: event heads
    float* dt = reinterpret_cast<float*>(pub + 3 * 4 * NW9 * 64);     // [2 kinds][4] padded tiles: the chain's `d` vectors
    float* gscr = dt + 2 * NW9 * SCR9;                      // [4] private transpose tiles of the gradient waves
    f4* wl = reinterpret_cast<f4*>(gscr + NW9 * SCR9);      // [NAE][4 chunks][4 waves][64]: the AE's transposed first-layer blocks (chain; each lane reads back what it wrote)
    f4* mbox = wl + NBE * 4 * NW9 * 64;                     // [2 parities][4][64] the stage's saved hidden layer, [4][64] the head's: raw rows for the chain wave with the same units
    f4* mbox_h = mbox + 2 * NW9 * 64;

    const int l = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int w = wv & 3;                                   // units / dims 16w..16w+15 in either role
    const int g = l >> 4, j = l & 15, i = l & 15;
    const long long b0 = (long long)blockIdx.x * 16;
    const bool valid = b0 + j < a.B;
    const long long b = valid ? b0 + j : a.B - 1;
    const int own = 16 * w + 4 * g;
    const long long tst = a.t.st, nT = a.T;
    const unsigned offR = (unsigned)(b * H9) + own;
    const unsigned offT = (unsigned)(b * a.t.sb);
    const bool has_z = a.zd > 0;
    const float* spb[2] = {has_z ? a.z.p : a.v.p, a.v.p};
    const long long sst[2] = {has_z ? a.z.st : a.v.st, a.v.st};
    const unsigned spo[2] = {(unsigned)(b * (has_z ? a.z.sb : a.v.sb)) + own, (unsigned)(b * a.v.sb) + own};
    const float* jpb[2] = {has_z ? a.zj : a.vj, a.vj};
    const long long jse[2] = {has_z ? a.zje : a.vje, a.vje};
    const unsigned jpo[2] = {(unsigned)(b * (has_z ? a.zjb : a.vjb)) + own, (unsigned)(b * a.vjb) + own};
    auto load_zv = [&](const int s, const long long k, const int ev) -> f4 {      // (one load: base and offset are selected, not the values)
        const bool e_ = ev >= 0;
        return ldg<f4>(sbase(e_ ? jpb[s] + (e_ ? ev : 0) * jse[s] : spb[s] + k * sst[s]), 4u * (e_ ? jpo[s] : spo[s]));
    };
    auto row_of = [&](const float* base, const long long k) -> f4 { return ldg<f4>(sbase(base + k * a.B * H9), 4u * offR); };
    const bool has_ev = a.ev != nullptr;
    int lane_zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));
    const int* evp = (has_ev ? a.ev : reinterpret_cast<const int*>(a.t.p)) + lane_zero;      // per-lane load: the entry stays raw until it is used
    const int troff = 4 * (16 * (i >> 2) + g) + 8 * (i >> 2) + (i & 3);                     // transposed read of a padded tile
    auto get_t = [&](const float* t_) -> f4 { const float* s_ = t_ + troff; return f4{s_[0], s_[16], s_[32], s_[48]}; };

    if (wv >= NW9) {
        // ================================================================ gradient wave
        float* scr = gscr + w * SCR9;
        auto transpose = [&](const f4 v) -> f4 { *reinterpret_cast<f4*>(scr + 4 * l + 8 * g) = v; return get_t(scr); };
        auto tr_pub = [&](const int par, const int slot, const f4 v) { pub[((par * 4 + slot) * NW9 + w) * 64 + l] = transpose(v); };
        auto outer = [&](A9& acc, const f4 dT, const int par, const int slot) {
#pragma unroll
            for (int c2 = 0; c2 < 4; c2 += 2) {      // two tiles at a time (registers: this wave holds 144 accumulators)
                f4 vT[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) vT[c] = pub[((par * 4 + slot) * NW9 + ((w + c2 + c) & 3)) * 64 + l];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int c = 0; c < 2; ++c) acc.c[c2 + c] = m9(dT[kk], vT[c][kk], acc.c[c2 + c]);
            }
        };
        auto d_tile = [&](const int kind) -> f4 { return get_t(dt + (kind * NW9 + w) * SCR9); };
        A9 accF[NBLK], accW2, accAF[NAE], accW2a;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            accW2.c[c] = z9(); accW2a.c[c] = z9();
#pragma unroll
            for (int blk = 0; blk < NBLK; ++blk) accF[blk].c[c] = z9();
#pragma unroll
            for (int bb = 0; bb < NAE; ++bb) accAF[bb].c[c] = z9();
        }
        struct Rows { f4 v[4]; };
        // (requests are pinned between two sched_barriers: a global load is not ordered by the LDS-only fences of lds_barrier(), and the
        //  scheduler hoisted all twenty of an iteration to its top -- every row live at once, each spilled the moment it arrived)
        auto req_head = [&](const long long jq, Rows& r) {
            __builtin_amdgcn_sched_barrier(0);
            r.v[0] = row_of(d.xs, jq);
#pragma unroll
            for (int s = 0; s < NZV; ++s) r.v[1 + s] = load_zv(s, jq, -1);
            r.v[3] = row_of(a.saeact, jq);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto pub_head = [&](const int par, const Rows& r) {
            mbox_h[w * 64 + l] = r.v[3];
            tr_pub(par, 0, r.v[0]);
#pragma unroll
            for (int s = 0; s < NZV; ++s) tr_pub(par, 1 + s, r.v[1 + s]);
            tr_pub(par, 3, r.v[3]);
        };
        auto req_stage = [&](const long long idx, Rows& r) {
            __builtin_amdgcn_sched_barrier(0);
            r.v[0] = row_of(a.sact, idx); r.v[1] = row_of(a.sxst, idx);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto pub_stage = [&](const int par, const Rows& r) { mbox[(par * NW9 + w) * 64 + l] = r.v[0]; tr_pub(par, 0, r.v[0]); tr_pub(par, 1, r.v[1]); };
        auto req_ext = [&](const long long k, const int ev, Rows& r) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < NZV; ++s) r.v[s] = load_zv(s, k, ev);
            r.v[NZV] = row_of(ev >= 0 ? a.sevi : d.is_, ev >= 0 ? (long long)ev : k);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto pub_ext = [&](const int par, const Rows& r) {
#pragma unroll
            for (int e = 0; e < NBE; ++e) tr_pub(par, e, r.v[e]);
        };
        // the head's four exchanges (grid point or event): gi in d tile 0 behind the first, delta1 in d tile 1 behind the second
        auto head_phase = [&](const int par, auto&& after_first) {
            lds_barrier();
            { const f4 giT = d_tile(0); outer(accW2a, giT, par, 3); }
            after_first();
            lds_barrier();
            const f4 dT = d_tile(1);
            outer(accAF[0], dT, par, 0);
#pragma unroll
            for (int bb = 1; bb < NAE; ++bb) { lds_barrier(); outer(accAF[bb], dT, par, bb); }
        };
        Rows rs[2] = {};                                    // rows in flight: set (phase parity)
        int ev_n = -1, evr = 0;
        {   // prologue: the first head's tiles (parity 0), the first stage's rows on their way (set 1)
            req_head(nT - 1, rs[0]);
            pub_head(0, rs[0]);
            const long long kk = nT >= 2 ? nT - 2 : 0;
            req_stage(kk * S + (S - 1), rs[1]);
            if (nT >= 2) ev_n = has_ev ? __builtin_amdgcn_readfirstlane(a.ev[nT - 2]) : -1;
            evr = evp[nT >= 3 ? nT - 3 : 0];
        }
        constexpr int NP = S + 2;                           // phases per iteration: head, S stages, external blocks
        // one full iteration, jg >= 1: the head at grid point jg and the step jg-1.  (The head at grid point 0 follows the loop: an exit in
        // the middle of the body made the allocator shuffle and spill the 144 accumulators around it.)
        auto iter = [&](const long long jg, auto kp_tag) {
            constexpr int Q0 = (NP & 1) ? decltype(kp_tag)::value : 0;      // parity of this iteration's first phase
            const long long k = jg - 1, kn = k > 0 ? k - 1 : 0;
            const int ev = ev_n;
            // ---- phase 0: head at grid point jg.  publish phase 1 (stage S-1) from set Q0+1, request phase 2 into set Q0
            head_phase(Q0 & 1, [&] {
                pub_stage((Q0 + 1) & 1, rs[(Q0 + 1) & 1]);
                if constexpr (S >= 2) req_stage(k * S + (S - 2), rs[Q0 & 1]);
                else req_ext(k, ev, rs[Q0 & 1]);
            });
            ev_n = has_ev ? __builtin_amdgcn_readfirstlane(evr) : -1;      // event index of the NEXT iteration's step (requested an iteration ago)
            evr = evp[k >= 2 ? k - 2 : 0];
            // ---- phases 1..S: stages S-1..0 (phase n: tiles in parity n, rows of phase n+1 in set n+1, set n free)
#pragma unroll
            for (int s = S - 1; s >= 0; --s) {
                const int nph = Q0 + S - s, par = nph & 1;
                lds_barrier();                                   // W2^T reduce-scatter: gk is in d tile 0
                { const f4 gkT = d_tile(0); outer(accW2, gkT, par, 0); }
                if (s >= 1) pub_stage(par ^ 1, rs[(nph + 1) & 1]); else pub_ext(par ^ 1, rs[(nph + 1) & 1]);
                if (s >= 2) req_stage(k * S + (s - 2), rs[nph & 1]);
                else if (s == 1) req_ext(k, ev, rs[nph & 1]);
                else req_head(k, rs[nph & 1]);
                lds_barrier();                                   // F_x^T reduce-scatter: delta1 is in d tile 1
                { const f4 d1T = d_tile(1); outer(accF[0], d1T, par, 1); }
            }
            // ---- phase S+1: external blocks
            {
                constexpr int nph = Q0 + S + 1, par = nph & 1;
                lds_barrier();
                const f4 DT = d_tile(0);
                outer(accF[1], DT, par, 0);
                pub_head(par ^ 1, rs[(nph + 1) & 1]);            // the next iteration's head
                req_stage(kn * S + (S - 1), rs[nph & 1]);        // ... and its first stage
                if (ev >= 0) {                                   // the event's head: tiles in parity 2 (rare; loaded here)
                    tr_pub(2, 0, row_of(d.xs, k));
#pragma unroll
                    for (int s = 0; s < NZV; ++s) tr_pub(2, 1 + s, load_zv(s, k, ev));
                    tr_pub(2, 3, row_of(a.sevact, (long long)ev));
                }
#pragma unroll
                for (int e = 1; e < NBE; ++e) { lds_barrier(); outer(accF[1 + e], DT, par, e); }
                if (ev >= 0) head_phase(2, [] {});
            }
        };
        if constexpr (NP & 1) {
            long long jg = nT - 1;
            for (; jg >= 2; jg -= 2) { iter(jg, std::integral_constant<int, 0>{}); iter(jg - 1, std::integral_constant<int, 1>{}); }
            if (jg == 1) { iter(1, std::integral_constant<int, 0>{}); head_phase(1, [] {}); }
            else head_phase(0, [] {});
        } else {
            for (long long jg = nT - 1; jg >= 1; --jg) iter(jg, std::integral_constant<int, 0>{});
            head_phase(0, [] {});
        }
        // ---- epilogue: sum(delta1) of the DE / AE in the d tiles behind the first of the all_initial reduce-scatters
        lds_barrier();
        const f4 s1T = d_tile(0), as1T = d_tile(1);
#pragma unroll
        for (int blk = 1; blk < NBLK; ++blk) lds_barrier();
        float* wp = d.wpart + (size_t)blockIdx.x * (d.NP_de + d.NP_ae);
        auto write_blocks = [&](float* o, const int K1, auto is_ae_c, auto nfront_c, const f4 sT, const A9* accB_, const A9& accW2_)
                                __attribute__((always_inline)) {
            constexpr bool is_ae = decltype(is_ae_c)::value;
            constexpr int nfront = decltype(nfront_c)::value;
            const int oW2 = H9 * K1 + H9;
#pragma unroll
            for (int blk = 0; blk < NBLK; ++blk) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int col = H9 * blk + 16 * ((w + c) & 3) + j;
                    f4 ca0 = z9();
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
                for (int bb = 0; bb < nfront; ++bb)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int col = n + H9 * bb + 16 * ((w + c) & 3) + j;
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[(size_t)(16 * w + 4 * g + r) * K1 + col] = accB_[bb].c[c][r];
                    }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[oW2 + (16 * w + 4 * g + r) * H9 + 16 * ((w + c) & 3) + j] = accW2_.c[c][r];
        };
        write_blocks(wp, 3 * n, std::false_type{}, std::integral_constant<int, NBLK>{}, s1T, accF, accW2);
        write_blocks(wp + d.NP_de, n + H9 * NAE, std::true_type{}, std::integral_constant<int, NAE>{}, as1T, accAF, accW2a);
        return;
    }

    // ==================================================================== chain wave
    const float* pw = pack_de + (size_t)w * D_R * 64 + l;
    const float* pwa = pack_ae + (size_t)w * A_R * 64 + l;
    float w2t[16], wftx[16], wfti[16], tq[2 * NBE][16];
    f4* wlp = wl + w * 64 + l;                              // block q, chunk c at wlp[(q*4 + c) * 256]
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        w2t[k] = pw[(D_W2T + k) * 64];
        wftx[k] = pw[(D_FT + k) * 64];
        wfti[k] = pw[(D_FT + 16 * (NBLK - 1) + k) * 64];
#pragma unroll
        for (int bb = 0; bb < NAE; ++bb) tq[LQ_AFT + bb][k] = 0.0f;      // (in LDS, below)
        tq[LQ_AW2T][k] = pwa[(A_W2T + k) * 64];
#pragma unroll
        for (int s = 0; s < NZV; ++s) tq[LQ_DFT + s][k] = pw[(D_FT + 16 * (1 + s) + k) * 64];
    }
#pragma unroll
    for (int bb = 0; bb < NAE; ++bb) {
        const float* src = pwa + (A_FT + 16 * bb) * 64;
#pragma unroll
        for (int c = 0; c < 4; ++c) wlp[(bb * 4 + c) * 256] = f4{src[(4 * c) * 64], src[(4 * c + 1) * 64], src[(4 * c + 2) * 64], src[(4 * c + 3) * 64]};
    }
    int q = 0;
    auto reduce_scatter = [&](const f4 (&part)[4]) -> f4 {
#pragma unroll
        for (int c = 1; c < 4; ++c) rsbuf[((q * NW9 + ((w + c) & 3)) * NW9 + w) * 64 + l] = part[c];
        lds_barrier();
        f4 out = part[0];
#pragma unroll
        for (int c = 1; c < 4; ++c) out += rsbuf[((q * NW9 + w) * NW9 + ((w + c) & 3)) * 64 + l];
        q ^= 1;
        return out;
    };
    auto mulT = [&](const f4 w4, const f4 dl) -> f4 {
        f4 acc = m9(w4[0], dl[0], z9());
        acc = m9(w4[1], dl[1], acc);
        acc = m9(w4[2], dl[2], acc);
        return m9(w4[3], dl[3], acc);
    };
    auto blkT = [&](const float (&wt)[16], const f4 dl) -> f4 {
        f4 part[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) part[c] = mulT(f4{wt[4 * c], wt[4 * c + 1], wt[4 * c + 2], wt[4 * c + 3]}, dl);
        return reduce_scatter(part);
    };
    auto blkT_l = [&](const int qb, const f4 dl) -> f4 {
        f4 part[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) part[c] = mulT(wlp[(qb * 4 + c) * 256], dl);
        return reduce_scatter(part);
    };
    auto put_d = [&](const int kind, const f4 v) { *reinterpret_cast<f4*>(dt + (kind * NW9 + w) * SCR9 + 4 * l + 8 * g) = v; };
    float* gdst[2] = {has_z ? d.gz : d.gv, d.gv};
    float* gjdst[2] = {has_z ? d.gzj : d.gvj, d.gvj};
    const unsigned offJ = (unsigned)(b * d.n_events * H9) + own;
    auto store_zv = [&](const int s, const long long grid, const int ev, const f4 val) {
        if (!valid) return;
        if (ev >= 0) { if (gjdst[s]) stg<f4>(sbase(gjdst[s] + (long long)ev * H9), 4u * offJ, val); }
        else if (gdst[s]) stg<f4>(sbase(gdst[s] + grid * a.B * H9), 4u * offR, val);
    };
    struct ZV { f4 b[NZV > 0 ? NZV : 1]; };
    f4 S1 = z9(), SB2 = z9(), AS1 = z9(), ASB2 = z9();
    // VJP of the AE head (hidden layer ah saved by the forward) with output adjoint gi: returns dL/dx, fills dL/d(z|v)
    // (ah_src: null = the head mailbox -- read behind the first exchange, which also makes it visible; else the event's saved row)
    auto ae_vjp = [&](const float* ah_src, const f4 gi, ZV& gzv) -> f4 {
        ASB2 += gi;
        put_d(0, gi);
        f4 ahe = z9();
        if (ah_src) ahe = ldg<f4>(sbase(ah_src), 4u * offR);
        const f4 pre = blkT(tq[LQ_AW2T], gi);
        const f4 ah = ah_src ? ahe : mbox_h[w * 64 + l];
        const f4 d1 = pre * dact9(ah);
        AS1 += d1;
        put_d(1, d1);
        f4 gx = z9();
#pragma unroll
        for (int bb = 0; bb < NAE; ++bb) {
            const f4 gb = blkT_l(LQ_AFT + bb, d1);
            if (bb == 0) gx = gb;
            else gzv.b[bb - 1] = gb;
        }
        return gx;
    };
    const bool has_gis = d.gis != nullptr;
    auto load_t = [&](const long long kk) -> float { return ldg<float>(sbase(a.t.p + kk * tst), 4u * offT); };
    f4 gcarry = z9(), gicarry = z9();
    ZV dezv;
#pragma unroll
    for (int s = 0; s < (NZV > 0 ? NZV : 1); ++s) dezv.b[s] = z9();
    // inputs of iteration jg, requested an iteration ahead and kept RAW until they are consumed
    // (the saved hidden layers come through the mailboxes the gradient wave with the same units fills a phase ahead: no register holds them)
    f4 grow = row_of(d.gxs, nT - 1), girow = row_of(has_gis ? d.gis : d.gxs, nT - 1);
    float t_hi = load_t(nT - 1), t_lo = load_t(nT >= 2 ? nT - 2 : 0);
    int ev_n = -1, evr = 0;
    if (nT >= 2) ev_n = has_ev ? __builtin_amdgcn_readfirstlane(a.ev[nT - 2]) : -1;
    evr = evp[nT >= 3 ? nT - 3 : 0];
    int pp = 0;                                             // parity of the phase in flight (head, stages, external blocks: one flip each)
    for (long long jg = nT - 1; jg >= 0; --jg) {
        f4 g1 = gcarry + (valid ? grow : z9());
        const f4 gi = gicarry + ((has_gis && valid) ? girow : z9());
        {
            const long long jn = jg > 0 ? jg - 1 : 0;
            grow = row_of(d.gxs, jn); girow = row_of(has_gis ? d.gis : d.gxs, jn);
        }
        {   // (1) AE head at grid point jg
            ZV gzv;
            g1 += ae_vjp(nullptr, gi, gzv);
#pragma unroll
            for (int s = 0; s < NZV; ++s) store_zv(s, jg, -1, dezv.b[s] + gzv.b[s]);
        }
        if (jg == 0) { gcarry = g1; break; }
        // (2) step k = jg - 1
        const long long k = jg - 1, kn = k > 0 ? k - 1 : 0;
        const int ev = ev_n;
        const float h_ = t_hi - t_lo;
        ev_n = has_ev ? __builtin_amdgcn_readfirstlane(evr) : -1;
        evr = evp[k >= 2 ? k - 2 : 0];
        t_hi = t_lo;
        t_lo = load_t(kn);
        f4 gks[S], gx0 = g1, D1 = z9();
#pragma unroll
        for (int s = 0; s < S; ++s) gks[s] = (h_ * rk_b(METHOD, s)) * g1;
#pragma unroll
        for (int s = S - 1; s >= 0; --s) {
            const f4 gk = gks[s];
            SB2 += gk;
            put_d(0, gk);
            pp ^= 1;
            const f4 pre = blkT(w2t, gk);
            const f4 d1 = pre * dact9(mbox[(pp * NW9 + w) * 64 + l]);
            D1 += d1;
            put_d(1, d1);
            const f4 gx = blkT(wftx, d1);
            gx0 += gx;
#pragma unroll
            for (int jj = 0; jj < s; ++jj) gks[jj] += (h_ * rk_a(METHOD, s, jj)) * gx;
        }
        S1 += D1;
        put_d(0, D1);
        pp ^= 1;                                            // (external blocks; the next head flips again)
        f4 gext[NBE];
#pragma unroll
        for (int e = 0; e < NBE; ++e) gext[e] = e < NZV ? blkT(tq[LQ_DFT + e], D1) : blkT(wfti, D1);
        if (ev >= 0) {
            // the algebraic input was g(x_k; jumps): chain its VJP in; z|v gradients (DE + AE part) go to the jump arrays
            ZV gq;
            gx0 += ae_vjp(a.sevact + (long long)ev * a.B * H9, gext[NBE - 1], gq);
#pragma unroll
            for (int s = 0; s < NZV; ++s) { store_zv(s, k, ev, gext[s] + gq.b[s]); dezv.b[s] = z9(); }
            gicarry = z9();
        } else {
#pragma unroll
            for (int s = 0; s < NZV; ++s) dezv.b[s] = gext[s];
            gicarry = gext[NBE - 1];
        }
        gcarry = gx0;
        pp ^= 1;                                            // the next iteration's head
    }
    // ---- epilogue
    if (valid) *reinterpret_cast<f4*>(d.gx0 + b * H9 + own) = gcarry;
    put_d(0, S1);
    put_d(1, AS1);
    for (int blk = 0; blk < NBLK; ++blk) {       // d all_initial block = A0_blk^T sum_t(delta1)  (DE + AE)
        f4 part[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float* sd = pw + (D_A0T + 16 * blk) * 64;
            const float* sa = pwa + (A_A0T + 16 * blk) * 64;
            part[c] = mulT(f4{sd[(4 * c) * 64], sd[(4 * c + 1) * 64], sd[(4 * c + 2) * 64], sd[(4 * c + 3) * 64]}, S1) +
                      mulT(f4{sa[(4 * c) * 64], sa[(4 * c + 1) * 64], sa[(4 * c + 2) * 64], sa[(4 * c + 3) * 64]}, AS1);
        }
        const f4 ga = reduce_scatter(part);
        if (valid) *reinterpret_cast<f4*>(d.ga0 + b * n + H9 * blk + own) = ga;
    }
    {   // biases: row sums over the 16 trajectories of a lane group
        float* wp = d.wpart + (size_t)blockIdx.x * (d.NP_de + d.NP_ae);
        auto write_bias = [&](float* o, const int K1, const f4 s1v, const f4 sb2v) {
            const int oB1 = H9 * K1, oB2 = oB1 + H9 + H9 * H9;
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
        write_bias(wp, 3 * n, S1, SB2);
        write_bias(wp + d.NP_de, n + H9 * NAE, AS1, ASB2);
    }
}
size_t lds9_roles_bytes(int nae) { return (size_t)(2 * NW9 * NW9 * 64 + 3 * 4 * NW9 * 64 + nae * 4 * NW9 * 64 + 3 * NW9 * 64) * sizeof(f4) + (size_t)(2 * NW9 + NW9) * SCR9 * sizeof(float); }


template <int METHOD>
hipError_t launch9_roles_m(int nbe, const Bwd9Dev& d, const float* pde, const float* pae, hipStream_t s) {
    const size_t ldsr = lds9_roles_bytes(nbe);
#define PSNODE_K9R(NBE_)                                                                                                          \
    {                                                                                                                             \
        auto kr = &latent64_backward_roles_kernel<METHOD, NBE_>;                                                                  \
        hipError_t er = hipFuncSetAttribute(reinterpret_cast<const void*>(kr), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsr); \
        if (er != hipSuccess) return er;                                                                                          \
        hipLaunchKernelGGL(kr, dim3((unsigned)((d.a.B + 15) / 16)), dim3(512), ldsr, s, d, pde, pae);                             \
        return hipGetLastError();                                                                                                 \
    }
    if (nbe == 3) PSNODE_K9R(3)
    PSNODE_K9R(2)
#undef PSNODE_K9R
}
}  // namespace

hipError_t launch9_roles(int method, int nbe, const Bwd9Dev& d, const float* pde, const float* pae, hipStream_t s) {
    switch (method) {
        case PSNODE_EULER: return launch9_roles_m<PSNODE_EULER>(nbe, d, pde, pae, s);
        case PSNODE_MIDPOINT: return launch9_roles_m<PSNODE_MIDPOINT>(nbe, d, pde, pae, s);
        default: return launch9_roles_m<PSNODE_RK4_38>(nbe, d, pde, pae, s);
    }
}

}  // namespace psnode
