// K7f -- backward pass through the DAE integrator for hidden widths 32 / 64 / 128 with the DE's parameter gradients formed IN the kernel
// (C ABI psnode_dae_backward_wide_f32 with psnode_dae_bwd_wide_args_f32::grad_params_de set; neural_01_DAE_01_no_encode.py:419-421 --
// loss.backward() through the loop of my_solvers.py:95-123 -- at the scripts' argparse default --hidden 128).  K7w's sweep
// (psnode_dae_backward_wide.hip: AE head per grid point, DE stages per step, event-time recompute) with K4f's weight-gradient scheme
// (psnode_backward_fused.hip) for the DE: the exchange tiles are padded so that the all-gathered delta_l can be read TRANSPOSED, wave w
// accumulates dW_l[all units][own 16 units] in registers for the whole launch, per-workgroup partials are summed in a fixed order.
// Nothing of the DE's stages is written to HBM any more (round 2's split: 6 rows of H floats per state-step and stage = 50 GB per pass
// at hidden 128, contracted by library GEMMs); dL/dz, dL/dv, the jump gradients and the DE part of dL/dall_initial leave the kernel in
// their final layout.  The AE head runs once per GRID POINT (a quarter of the DE's work at RK4): its six rows per grid point are still
// stored and contracted outside (its accumulators would not fit next to the DE's: 2 x 64 registers per lane at hidden 128).
// 8 waves: the 128 KB weight region of the LDS holds the DE's FORWARD images during phase A and its TRANSPOSED images during phase B
// (LDS-DMA swap as K4f); the AE's H->H images -- forward and transposed -- are read from the packed tensors (L2) next to the MFMAs
// that consume them; stage activations through a per-workgroup ring in the workspace.
#define PSNODE_ELU_LITERALS
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "psnode_wide_pack.h"

namespace psnode {
namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f4 fm4(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

struct FusedDaeDev {
    int method, xd, zd, vd, id, hreal, n_events, NP;
    long long T, B;
    const float *w1, *w4, *aw1, *aw4;     // raw nn.Linear tensors for the small transposed operands
    ViewDev t, z, v;
    const float* a0;
    const int* ev;
    const float *zj, *vj;
    long long zjb, zje, vjb, vje;
    const float *xs, *is, *gxs, *gis;
    const float *xtrue, *itrue;           // teacher forcing (recompute form only; my_solvers.py:111-121): dataset rows [T,B,xd] / [T,B,id]
    int tx, ti;                           //   tx: DE starts and grid heads read xtrue, no x adjoint from step to step; ti: DE reads itrue, AE -> DE link cut
    float* carry_x;                       // [B, xd]: adjoint of x at grid point 0 (without dL/dxs[0])
    float *gzv, *gjump, *ga0;             // [T, B, nzv] (DE part), [B, n_events, nzv] (DE part), [B, n] (DE part)
    float* wpart;                         // [workgroups][NP]
    float* wpart_ae;                      // <= 4 waves: [workgroups][NPA] partials of the AE head's gradients (formed in the kernel: round 4)
    int NPA;
    float* ring;                          // NWV >= 8: [S][3][B][H] stage activations of the step in flight
    float *aact[3], *adelta[3], *agi;     // AE head rows per grid point [T, B, H] / [T, B, 16]
    float *eact[3], *edelta[3], *egi, *ei;
    // saved by the forward call (psnode_dae_args_f32::save_*) or null: recompute
    const float *sact, *sxst, *saeact, *sevact, *sevi;
};

#ifndef PSNODE_K7F_EVERY
#define PSNODE_K7F_EVERY 2      // 8 waves, RK4 / Midpoint: a sched_barrier behind every EVERY-th chunk of the layer loops (K4f: PSNODE_K4F_EVERY)
#endif
constexpr int FTILE = 64 * 4 + 4 * 8;     // padded 16x16 tile (floats): lane l's four rows 4g..4g+3 of column j at 4l + 8g

__host__ __device__ constexpr bool aet_in_lds(int nwv) { return nwv <= 4; }

#ifndef PSNODE_K7F_AE_GRADS
#define PSNODE_K7F_AE_GRADS 1
#endif
#define PSNODE_K7F_AE_GRADS_DEFAULT PSNODE_K7F_AE_GRADS
#ifndef PSNODE_K7F_ROLES
#define PSNODE_K7F_ROLES 1      // <= 4 waves, saved activations: NWV gradient waves per tile own the four H->H weight gradients (K4f: PSNODE_K4F_ROLES)
#endif
#ifndef PSNODE_K7F_ROLES_EARLY
#define PSNODE_K7F_ROLES_EARLY 1
#endif
// Two-role form (round 4; psnode_backward_fused.hip: fused_gradient_wave): waves NWV..2NWV-1 of the workgroup follow the chain's barrier
// sequence -- per step: the head at grid point k+1 (3 exchanges), 3 per stage, the step's external-input all-reduce, the event's head
// (3) -- and contract the delta tiles the chain's all-gathers publish with their own 16 units of the SAVED activations (DE stage rows,
// grid heads, event heads), which they load themselves: dW2 / dW3 of the DE, dAW2 / dAW3 of the AE head.
// LDS tiles of the two-role form behind the 2 NWV exchange parities (FTILE floats each): chain-private NWV | gradient-private NWV |
// mailboxes [kind 0: DE stage rows, 1: head rows][wave][3 layers] | x boxes NWV
template <int NWV> __device__ __forceinline__ float* k7f_roles_mailbox(float* xb, const int kind, const int wv) {
    return xb + (size_t)(4 * NWV + (kind * NWV + wv) * 3) * FTILE;
}
template <int NWV> __device__ __forceinline__ float* k7f_roles_xbox(float* xb, const int wv) { return xb + (size_t)(10 * NWV + wv) * FTILE; }
// ... | delta1 NWV | g (output adjoint of the MLP being swept) | u (its first-layer input rows): handed to the gradient waves (padded tiles)
template <int NWV> __device__ __forceinline__ float* k7f_roles_d1_tile(float* xb, const int wv) { return xb + (size_t)(11 * NWV + wv) * FTILE; }
template <int NWV> __device__ __forceinline__ float* k7f_roles_g_tile(float* xb) { return xb + (size_t)(12 * NWV) * FTILE; }
template <int NWV> __device__ __forceinline__ float* k7f_roles_u_tile(float* xb) { return xb + (size_t)(12 * NWV + 1) * FTILE; }
#ifndef PSNODE_K7F_ROLES_SMALL
#define PSNODE_K7F_ROLES_SMALL 1  // the gradient waves also own dW4 / dW1 (s columns) / dAW4 (P3) / dAW1 (u columns): no in-wave transpose left on the chain
#endif

template <int METHOD, int NZM, int NWV>
__device__ __forceinline__ void dae_fused_gradient_wave(const FusedDaeDev& a, float* __restrict__ xb, const int l, const int wg) {
    constexpr int S = rk_stages(METHOD), H = 16 * NWV;
    const int g = l >> 4, j = l & 15, HR = a.hreal;
    float* scr = xb + (3 * NWV + wg) * FTILE;                          // private transpose tile (behind the chain waves')
    const long long b0 = (long long)blockIdx.x * TBM;
    const long long b = b0 + j < a.B ? b0 + j : a.B - 1;               // padding trajectories: their deltas are zero
    const int toff = 4 * l + 8 * g, roff = 72 * (j >> 2) + 4 * g + (j & 3);
    auto tile = [&](const int par, const int wv) -> const float* { return xb + (par * NWV + wv) * FTILE; };
    auto get_row = [&](const float* t_) -> f4 { const float* s_ = t_ + roff; return f4{s_[0], s_[16], s_[32], s_[48]}; };
    auto transpose = [&](const f4 v) -> f4 { *reinterpret_cast<f4*>(scr + toff) = v; return get_row(scr); };
    const unsigned offH = 4u * ((unsigned)(b * H) + 16 * wg + 4 * g);
    const long long act_layer = a.B * H, nT = a.T;
    constexpr int NX = kNXc;
    unsigned offXc[NX];
#pragma unroll
    for (int r = 0; r < NX; ++r) offXc[r] = 4u * ((unsigned)(b * a.xd) + (4 * r + g < a.xd ? 4 * r + g : 0));
    struct Rows { f4 v1, v2, v3; float x[NX]; };
    auto load_saved = [&](const long long idx, Rows& q) {
        const float* rb = a.sact + (size_t)idx * 3 * act_layer;
        q.v1 = ldg<f4>(sbase(rb), offH);
        q.v2 = ldg<f4>(sbase(rb + act_layer), offH);
        q.v3 = ldg<f4>(sbase(rb + 2 * act_layer), offH);
        const gptr<const float> xrow = sbase(a.sxst + idx * a.B * a.xd);
#pragma unroll
        for (int r = 0; r < NX; ++r) q.x[r] = ldg<float>(xrow, offXc[r]);
    };
    auto load_head = [&](const long long kk, Rows& q) {
        const float* rb = a.saeact + (size_t)kk * act_layer;
        const size_t lay = (size_t)a.T * act_layer;
        q.v1 = ldg<f4>(sbase(rb), offH);
        q.v2 = ldg<f4>(sbase(rb + lay), offH);
        q.v3 = ldg<f4>(sbase(rb + 2 * lay), offH);
    };
    // mailboxes of the chain wave with these units (lane-linear): the chain has no global load of a saved row
    float* mb_act = k7f_roles_mailbox<NWV>(xb, 0, wg);
    float* mb_head = k7f_roles_mailbox<NWV>(xb, 1, wg);
    float* mb_x = k7f_roles_xbox<NWV>(xb, wg);
    auto publish = [&](float* mb, const Rows& q) {
        f4* m4 = reinterpret_cast<f4*>(mb) + l;
        m4[0] = q.v1; m4[64] = q.v2; m4[128] = q.v3;
    };
    auto publish_x = [&](const Rows& q) { reinterpret_cast<f2*>(mb_x)[l] = f2{q.x[0], NX > 1 ? q.x[NX > 1 ? 1 : 0] : 0.0f}; };
    const f4 zero4 = f4{0.f, 0.f, 0.f, 0.f};
    f4 accW2[NWV], accW3[NWV], accA2[NWV], accA3[NWV], accW4 = zero4, accW1s = zero4, accAP3 = zero4, accA1u = zero4;
#pragma unroll
    for (int c = 0; c < NWV; ++c) { accW2[c] = zero4; accW3[c] = zero4; accA2[c] = zero4; accA3[c] = zero4; }
    constexpr bool RSM = PSNODE_K7F_ROLES_SMALL;
    // rows of the u tile that hold column i of the first-layer input: the DE's stage input s = (x | ext), the head's u = (x | z | v)
    const int xd_ = a.xd, nzv_ = a.zd + a.vd, n_ = a.xd + nzv_ + a.id, i = j;
    const int rowx_ = 4 * (i & 3) + (i >> 2), rowe_ = 4 * ((i - xd_) & 3) + 2 + ((i - xd_) >> 2);
    const int srow = i < xd_ ? rowx_ : (i < n_ ? rowe_ : -1), arow = i < xd_ ? rowx_ : (i < xd_ + nzv_ ? rowe_ : -1);
    const int sroff = srow >= 0 ? 72 * (srow >> 2) + 4 * g + (srow & 3) : 0, aroff = arow >= 0 ? 72 * (arow >> 2) + 4 * g + (arow & 3) : 0;
    auto get_at = [&](const float* t_, const int ro) -> f4 { const float* s_ = t_ + ro; return f4{s_[0], s_[16], s_[32], s_[48]}; };
    constexpr int E = PSNODE_K7F_ROLES_EARLY < NWV ? PSNODE_K7F_ROLES_EARLY : NWV;
    auto read_tiles = [&](const int par, f4 (&dT)[NWV]) {
#pragma unroll
        for (int c = 0; c < NWV; ++c) dT[c] = get_row(tile(par, (wg + c) & (NWV - 1)));
    };
    auto chunks = [&](const f4 (&dT)[NWV], const f4 hT, f4 (&acc)[NWV], const int c0, const int c1) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int c = 0; c < NWV; ++c) if (c >= c0 && c < c1) acc[c] = fm4(dT[c][kk], hT[kk], acc[c]);
    };
    int p = 0;
    // one MLP swept backwards by the chain: two all-gathers (delta3, delta2) and an all-reduce; `mid()` runs between the second
    // all-gather's barrier and the all-reduce's (where the mailboxes may be rewritten)
    // (RSM: h3T = the MLP's third ELU output, own units, operand layout; acc4 += g (x) h3, acc1 += delta1 (x) u; uoff / uon: this lane's row of the u tile)
    auto sweep = [&](const f4 h1T, const f4 h2T, const f4 h3T, f4 (&acc3)[NWV], f4 (&acc2)[NWV], f4& acc4, f4& acc1, const int uoff, const bool uon,
                     auto&& mid) {
        f4 dT3[NWV], dT2[NWV], uT = zero4;
        lds_barrier();
        read_tiles(p, dT3);
        if constexpr (RSM) {
            const f4 gT = get_row(k7f_roles_g_tile<NWV>(xb));
            uT = get_at(k7f_roles_u_tile<NWV>(xb), uoff);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc4 = fm4(gT[kk], h3T[kk], acc4);
        }
        chunks(dT3, h2T, acc3, 0, E);
        __builtin_amdgcn_sched_barrier(0);
        p ^= 1;
        lds_barrier();
        read_tiles(p, dT2);
        mid();
        chunks(dT3, h2T, acc3, E, 2 * E);
        chunks(dT2, h1T, acc2, 0, E);
        __builtin_amdgcn_sched_barrier(0);
        p ^= 1;
        lds_barrier();
        if constexpr (RSM) {
            const f4 dT = get_row(k7f_roles_d1_tile<NWV>(xb, wg));
            if (!uon) uT = zero4;
            chunks(dT3, h2T, acc3, 2 * E, NWV);
            chunks(dT2, h1T, acc2, E, NWV);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc1 = fm4(dT[kk], uT[kk], acc1);
        } else {
            chunks(dT3, h2T, acc3, 2 * E, NWV);
            chunks(dT2, h1T, acc2, E, NWV);
        }
        p ^= 1;
    };
    auto nothing = [] {};
    auto tr3 = [&](const f4 v3) -> f4 { return RSM ? transpose(v3) : zero4; };
    // the DE rows of linear index idx live in set (idx parity): `s & 1` when the stage count is even, the parity KP of the step body at
    // Euler, whose time loop is written out twice (psnode_backward_fused.hip)
    Rows r0 = {}, r1 = {}, hn = {};
    constexpr int QLAST = S == 1 ? 0 : ((S - 1) & 1);
    if (nT >= 2) {
        const long long last = (nT - 1) * S - 1;
        if constexpr (QLAST == 0) { load_saved(last, r0); load_saved(last > 0 ? last - 1 : 0, r1); publish(mb_act, r0); publish_x(r0); }
        else { load_saved(last, r1); load_saved(last > 0 ? last - 1 : 0, r0); publish(mb_act, r1); publish_x(r1); }
    }
    load_head(nT - 1, hn);
    publish(mb_head, hn);
    lds_barrier();                                                 // the first rows are in the mailboxes
    auto step = [&](const long long k, auto kp_tag) {
        constexpr int KP = decltype(kp_tag)::value;
        (void)KP;
        const int ev = a.ev ? __builtin_amdgcn_readfirstlane(a.ev[k]) : -1;
        {   // the head at grid point k+1
            const f4 h3T = tr3(hn.v3), h2T = transpose(hn.v2), h1T = transpose(hn.v1);
            load_head(k, hn);                                      // the next head: published in front of this step's last exchange
            sweep(h1T, h2T, h3T, accA3, accA2, accAP3, accA1u, aroff, arow >= 0, nothing);
        }
#pragma unroll
        for (int s = S - 1; s >= 0; --s) {
            const long long idx = k * S + s;
            const int Q = S == 1 ? KP : (s & 1);                   // (compile-time once the stage loop is unrolled)
            if (Q == 0) {
                const f4 h3T = tr3(r0.v3), h2T = transpose(r0.v2), h1T = transpose(r0.v1);
                load_saved(idx > 1 ? idx - 2 : 0, r0);             // two stages ahead
                sweep(h1T, h2T, h3T, accW3, accW2, accW4, accW1s, sroff, srow >= 0, [&] { publish(mb_act, r1); publish_x(r1); });      // the chain's next stage
            } else {
                const f4 h3T = tr3(r1.v3), h2T = transpose(r1.v2), h1T = transpose(r1.v1);
                load_saved(idx > 1 ? idx - 2 : 0, r1);
                sweep(h1T, h2T, h3T, accW3, accW2, accW4, accW1s, sroff, srow >= 0, [&] { publish(mb_act, r0); publish_x(r0); });
            }
        }
        publish(mb_head, hn);
        lds_barrier();                                             // the step's external-input all-reduce
        p ^= 1;
        if (ev >= 0) {                                             // the event's head (rare)
            f4 e1, e2, e3 = zero4;
            const float* rb = a.sevact + (size_t)ev * 3 * act_layer;
            e1 = ldg<f4>(sbase(rb), offH);
            e2 = ldg<f4>(sbase(rb + act_layer), offH);
            if constexpr (RSM) e3 = ldg<f4>(sbase(rb + 2 * act_layer), offH);
            const f4 h3T = tr3(e3), h2T = transpose(e2), h1T = transpose(e1);
            sweep(h1T, h2T, h3T, accA3, accA2, accAP3, accA1u, aroff, arow >= 0, nothing);
        }
    };
    if constexpr (S == 1) {
        long long k = nT - 2;
        for (; k >= 1; k -= 2) { step(k, std::integral_constant<int, 0>{}); step(k - 1, std::integral_constant<int, 1>{}); }
        if (k == 0) step(0, std::integral_constant<int, 0>{});
    } else {
        for (long long k = nT - 2; k >= 0; --k) step(k, std::integral_constant<int, 0>{});
    }
    {   // the head at grid point 0
        const f4 h3T = tr3(hn.v3), h2T = transpose(hn.v2), h1T = transpose(hn.v1);
        sweep(h1T, h2T, h3T, accA3, accA2, accAP3, accA1u, aroff, arow >= 0, nothing);
    }
    lds_barrier();                                                 // epilogue: the two all-reduces of dL/dall_initial
    lds_barrier();
    const int ne = a.zd + a.vd + a.id, n = a.xd + ne, nzv = a.zd + a.vd, K1 = 3 * n, K1a = n + a.xd + nzv;
    float* wp = a.wpart + (size_t)blockIdx.x * a.NP;
    float* wa = a.wpart_ae + (size_t)blockIdx.x * a.NPA;
    const int oW2 = HR * K1 + HR, oW3 = oW2 + HR * HR + HR;
    const int aW2 = HR * K1a + HR, aW3 = aW2 + HR * HR + HR;
    const int v = 16 * wg + j;
#pragma unroll
    for (int c = 0; c < NWV; ++c) {
        const int ub = 16 * ((wg + c) & (NWV - 1)) + 4 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (ub + r < HR && v < HR) {
                wp[oW2 + (size_t)(ub + r) * HR + v] = accW2[c][r];
                wp[oW3 + (size_t)(ub + r) * HR + v] = accW3[c][r];
                wa[aW2 + (size_t)(ub + r) * HR + v] = accA2[c][r];
                wa[aW3 + (size_t)(ub + r) * HR + v] = accA3[c][r];
            }
        }
    }
    if constexpr (RSM) {
        const int oW4 = oW3 + HR * HR + HR, aP3 = aW3 + HR * HR + HR;
#pragma unroll
        for (int r = 0; r < 4; ++r) {   // dW4 rows (g, r) <-> x-dim 4r+g, columns = own units; P3 rows (g, r) <-> slot 4r+g
            const int dd = 4 * r + g;
            if (r < NX && dd < a.xd && v < HR) wp[oW4 + (size_t)dd * HR + v] = accW4[r];
            if (v < HR) wa[aP3 + (size_t)(4 * r + g) * HR + v] = accAP3[r];
        }
        // first layers: ca0 = sum(delta1) (x) a0 over the tile's trajectories; the chain waves left sum(delta1) of the DE in their private
        // tiles and of the AE head in their stage-row mailboxes (padded layout)
        const f4 s1T = get_row(xb + (2 * NWV + wg) * FTILE), sa1T = get_row(k7f_roles_mailbox<NWV>(xb, 0, wg));
        f4 ca0 = zero4, caa = zero4;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const long long tb = b0 + 4 * kk + g;
            const float av = (i < n_ && tb < a.B) ? a.a0[tb * n_ + i] : 0.0f;
            ca0 = fm4(s1T[kk], av, ca0);
            caa = fm4(sa1T[kk], av, caa);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int u = 16 * wg + 4 * g + r;
            if (u < HR) {
                float* row = wp + (size_t)u * K1;
                if (j < n_) { row[j] = ca0[r]; row[n_ + j] = accW1s[r] - ca0[r]; row[2 * n_ + j] = accW1s[r]; }
                float* rowa = wa + (size_t)u * K1a;
                if (j < n_) rowa[j] = caa[r];
                if (j < xd_ + nzv_) rowa[n_ + j] = accA1u[r];
            }
        }
    }
}

// REC = false: the forward call saved every ELU output and stage input: no forward evaluation in here at all (no phase A, no head
// recompute), the DE's transposed images stay in LDS for the whole launch, rows of (step, stage) are requested late in the previous stage.
// ROLES: 2 NWV waves per tile, see dae_fused_gradient_wave.
template <int METHOD, int NZM, int NZA, int NWV, bool REC = true, bool ROLES = false>
__global__ __launch_bounds__(64 * NWV * (ROLES ? 2 : 1)) void dae_backward_fused_kernel(const FusedDaeDev a, const float* __restrict__ pack_de,
                                                                        const float* __restrict__ pack_ae, const f4* __restrict__ pack_t,
                                                                        const f4* __restrict__ pack_f, const f4* __restrict__ pack_ta,
                                                                        const f4* __restrict__ pack_fa, const int NA) {
    constexpr int NX = kNXc, S = rk_stages(METHOD), H = 16 * NWV;
    using RD = Regs<NX, 0, NZM, NWV>;
    using RA = Regs<NX, 0, NZA, NWV>;
    constexpr bool STREAM = NWV >= 8 && REC; // DE forward / transposed images swapped in LDS per phase; activations through the ring
    constexpr bool AEG = NWV >= 8;           // AE images read from L2
    constexpr bool AET_LDS = aet_in_lds(NWV);
#ifndef PSNODE_K7F_BOUND
#define PSNODE_K7F_BOUND 3      // as PSNODE_K4F_BOUND: 0 never, 1 always at 8 waves, 2 RK4 only, 3 RK4 + Midpoint
#endif
    constexpr bool BOUND = NWV >= 8 && (PSNODE_K7F_BOUND == 1 || (PSNODE_K7F_BOUND == 2 && S >= 4) || (PSNODE_K7F_BOUND == 3 && S >= 2));
    constexpr int EVERY = PSNODE_K7F_EVERY;
    constexpr int TSZ = 2 * NWV * NWV * 64;  // f4 per pair of H->H images
    extern __shared__ __attribute__((aligned(16))) float lds[];
    f4* wT = reinterpret_cast<f4*>(lds);                                  // DE: [layer 0: W2 | 1: W3][chunk][wave][lane]; AET_LDS: the AE's transposes behind it
    float* xb = lds + (size_t)TSZ * 4 * (AET_LDS ? 2 : 1);               // [2][NWV] exchange tiles (padded)

    static_assert(!ROLES || (!REC && NWV <= 4), "two-role form: saved activations, <= 4 waves per tile");
    const int l = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if constexpr (ROLES) {
        if (w >= NWV) {
            dae_fused_gradient_wave<METHOD, NZM, NWV>(a, xb, l, w - NWV);
            return;
        }
    }
    const int g = l >> 4, j = l & 15, i = j;
    float* scr = xb + 2 * NWV * FTILE + w * FTILE;                       // this wave's private transpose tile
    const long long b0 = (long long)blockIdx.x * TBM;
    const bool valid = b0 + j < a.B;
    const long long b = valid ? b0 + j : a.B - 1;
    const int xd = a.xd, zd = a.zd, vd = a.vd, idim = a.id, HR = a.hreal;
    const int nzv = zd + vd, ne = nzv + idim, n = xd + ne;

    // ---- forward images -> registers, H->H images -> LDS
    const float* pw = pack_de + (size_t)w * (RD::COUNT + NA) * 64 + l;
    const float* pwa = pack_ae + (size_t)w * (RA::COUNT + NA) * 64 + l;
    constexpr bool WREG = NWV <= 4 && REC;   // H->H forward weights in registers
    float w1xs[NX], w1z[NZM], w2r[WREG ? 4 * NWV : 1], w3r[WREG ? 4 * NWV : 1], w4[4];
    float aw1x[NX], aw1e[NZA], aw2r[WREG ? 4 * NWV : 1], aw3r[WREG ? 4 * NWV : 1], aw4[4];
    f4 b1r, b2, b3, b4, ab1r, ab2, ab3, ab4;
#pragma unroll
    for (int r = 0; r < NX; ++r) { w1xs[r] = pw[(RD::W1A + r) * 64]; aw1x[r] = pwa[(RA::W1A + r) * 64]; }
#pragma unroll
    for (int m = 0; m < NZM; ++m) w1z[m] = pw[(RD::W1E + m) * 64];
#pragma unroll
    for (int m = 0; m < NZA; ++m) aw1e[m] = pwa[(RA::W1E + m) * 64];
#pragma unroll
    for (int k = 0; k < (WREG ? 4 * NWV : 0); ++k) {
        w2r[k] = pw[(RD::W2 + k) * 64]; w3r[k] = pw[(RD::W3 + k) * 64];
        aw2r[k] = pwa[(RA::W2 + k) * 64]; aw3r[k] = pwa[(RA::W3 + k) * 64];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        w4[r] = pw[(RD::W4 + r) * 64]; aw4[r] = AEG ? 0.0f : pwa[(RA::W4 + r) * 64];
        b1r[r] = pw[(RD::B1 + r) * 64]; ab1r[r] = pwa[(RA::B1 + r) * 64];
        b2[r] = pw[(RD::B2 + r) * 64]; b3[r] = pw[(RD::B3 + r) * 64]; b4[r] = pw[(RD::B4 + r) * 64];
        ab2[r] = pwa[(RA::B2 + r) * 64]; ab3[r] = pwa[(RA::B3 + r) * 64];
        ab4[r] = AEG ? 0.0f : pwa[(RA::B4 + r) * 64];
    }
    // LDS-DMA of one DE layer's image (0: W2, 1: W3) into its region (K4f: psnode_backward_fused.hip:dma_layer)
    const unsigned wT_lds = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<uintptr_t>(wT));
    const unsigned lane16 = 16u * (unsigned)l;
    auto dma_layer = [&](const f4* __restrict__ img, const int layer) {
#pragma unroll
        for (int c = 0; c < NWV; ++c) {
            const int slot = (layer * NWV + c) * NWV + w;
            const uintptr_t sb = reinterpret_cast<uintptr_t>(img + (size_t)slot * 64);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)sb), hi = __builtin_amdgcn_readfirstlane((unsigned)(sb >> 32));
            const unsigned long long base = ((unsigned long long)hi << 32) | lo;
            const unsigned dst = __builtin_amdgcn_readfirstlane(wT_lds + (unsigned)slot * 1024u);
            unsigned keep;
            asm volatile(PSNODE_LDS_DMA_ASM
                         : "=&s"(keep) : "v"(lane16), "s"(dst), "s"(base) : "memory");
        }
    };
    auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
    if constexpr (STREAM) {
        dma_layer(pack_f, 0);
        dma_layer(pack_f, 1);
    } else if constexpr (AEG) {       // saved activations: the transposed images, once
        dma_layer(pack_t, 0);
        dma_layer(pack_t, 1);
        dma_wait();
    } else {
#pragma unroll
        for (int c = 0; c < 2 * NWV; ++c) {
            wT[(c * NWV + w) * 64 + l] = pack_t[(c * NWV + w) * 64 + l];
            if constexpr (AET_LDS) wT[TSZ + (c * NWV + w) * 64 + l] = pack_ta[(c * NWV + w) * 64 + l];
        }
    }

    // ---- ext slots of this lane (as K2): kind 0 = z column, 1 = v column, 2 = algebraic variable, 3 = padding
    int ekind[NZM], ecol[NZM], akind[NZA], acol[NZA];
    float a0e[NZM];
#pragma unroll
    for (int m = 0; m < NZM; ++m) {
        const int q = 4 * m + g, e = slot_ext(q, ne);
        ekind[m] = e < 0 ? 3 : (e < zd ? 0 : (e < nzv ? 1 : 2));
        ecol[m] = e < 0 ? 0 : (e < zd ? e : (e < nzv ? e - zd : e - nzv));
        a0e[m] = q < ne ? a.a0[b * n + xd + q] : 0.0f;
    }
#pragma unroll
    for (int m = 0; m < NZA; ++m) {
        const int q = 4 * m + g;
        akind[m] = q < zd ? 0 : (q < nzv ? 1 : 3);
        acol[m] = q < zd ? q : (q < nzv ? q - zd : 0);
    }
    // ---- small transposed operands straight from the nn.Linear tensors (A operand: row i = l & 15, k-slot g)
    //   w4T[r]  = W4[4r+g][16w+i]                         g3  = W4^T gk
    //   fT[r]   = (Ws+Wd)[16w+4g+r][x-dim of row i]       gX  = F_x^T delta1            (output rows (g, r) = x-dim 4r+g)
    //   fE[r]   = W1[16w+4g+r][column(s) of the ext slot of row i]: adjoint of the DE's external inputs, slot layout.  An algebraic
    //             variable keeps its two slots apart (the AE's W4^T sums them); a z | v column gets both of its W1 columns on its
    //             `s - a0` slot and nothing on the other, so that slot q < nzv IS dL/d(z|v)[q]
    //   aw4T[m] = AW4[i-dim of slot 4m+g][16w+i]          g3a = AW4^T (slot adjoints): both slots of an i-dim carry its row
    //   afT[r]  = AW1[16w+4g+r][n + x-dim of row i]
    //   afZ[r]  = AW1[16w+4g+r][n + x + ext slot of row i], slots < nzv: the head's dL/d(z|v) in the slot layout gE has (<= 4 waves)
    float w4T[NX], fT[4], fE[4], aw4T[NZM], afT[4], afZ[4];
    {
        const int u = 16 * w + i, K1 = 3 * n, K1a = n + xd + nzv;
#pragma unroll
        for (int r = 0; r < NX; ++r) { const int d = 4 * r + g; w4T[r] = (d < xd && u < HR) ? a.w4[(size_t)d * HR + u] : 0.0f; }
#pragma unroll
        for (int m = 0; m < NZM; ++m) aw4T[m] = (ekind[m] == 2 && u < HR) ? a.aw4[(size_t)ecol[m] * HR + u] : 0.0f;
        const int o = 4 * (i & 3) + (i >> 2);           // x-dim / ext slot carried by output row i
        const int eo = slot_ext(o, ne);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int uu = 16 * w + 4 * g + r;
            const bool on = uu < HR;
            const float* row1 = a.w1 + (size_t)(on ? uu : 0) * K1;
            fT[r] = (o < xd && on) ? row1[2 * n + o] + row1[n + o] : 0.0f;
            float fe = 0.0f;
            if (on && eo >= 0) {
                if (eo >= nzv) fe = row1[o < ne ? n + xd + o : 2 * n + xd + (o - ne)];
                else if (o < ne) fe = row1[n + xd + o] + row1[2 * n + xd + o];
            }
            fE[r] = fe;
            afT[r] = (o < xd && on) ? a.aw1[(size_t)uu * K1a + n + o] : 0.0f;
            afZ[r] = (o < nzv && on) ? a.aw1[(size_t)uu * K1a + n + xd + o] : 0.0f;
        }
    }
    // bias + W1[:, a0 columns] . a0
    f4 c0 = b1r, c0a = ab1r;
    for (int m = 0; m < NA; ++m) {
        const int q = 4 * m + g;
        const float av = q < n ? a.a0[b * n + q] : 0.0f;
        c0 = fm4(pw[(RD::COUNT + m) * 64], av, c0);
        c0a = fm4(pwa[(RA::COUNT + m) * 64], av, c0a);
    }
    // row of a padded tile that holds column i of the DE's stage input s = (x dims | ext): x-dim d in row 4(d&3) + (d>>2) (registers 0..1
    // of lane group d&3), ext e in row 4(e&3) + 2 + (e>>2) (registers 2..3 of lane group e&3)
    // (mask arithmetic, not ?: -- the nested conditional became two divergent branches, and the register allocator parked the result in
    //  an AGPR from inside one of them: the spill-under-EXEC pattern of round 3's defect (a), caught by isa_lint.py on the first build)
    const int rowx_ = 4 * (i & 3) + (i >> 2), rowe_ = 4 * ((i - xd) & 3) + 2 + ((i - xd) >> 2);
    const int inx_ = -(int)(i < xd), ins_ = -(int)(i < n), ina_ = -(int)(i < xd + nzv);
    const int srow = (inx_ & rowx_) | (~inx_ & ((ins_ & rowe_) | ~ins_));          // -1 outside of (x | ext)
    // ... and the same for the AE head's first-layer input behind all_initial, u = (x dims | z | v)
    const int arow = (inx_ & rowx_) | (~inx_ & ((ina_ & rowe_) | ~ina_));

    // ---- LDS tiles
    const int toff = 4 * l + 8 * g;                                   // own slot of a tile
    const int roff = 72 * (i >> 2) + 4 * g + (i & 3);                 // row i of a tile, columns g, g+4, g+8, g+12
    auto tile = [&](const int par, const int wv) -> float* { return xb + (par * NWV + wv) * FTILE; };
    auto put = [&](float* t_, const f4 v) { *reinterpret_cast<f4*>(t_ + toff) = v; };
    auto getl = [&](const float* t_) -> f4 { return *reinterpret_cast<const f4*>(t_ + toff); };
    auto get_row = [&](const float* t_, const int ro) -> f4 { const float* s_ = t_ + ro; return f4{s_[0], s_[16], s_[32], s_[48]}; };
    auto transpose = [&](const f4 v) -> f4 { put(scr, v); return get_row(scr, roff); };   // own D tile -> operand layout (trajectory g + 4kk)

    int p = 0;
    constexpr bool PREFETCH_ALL = NWV <= 4;
    const f4 zero4 = f4{0.f, 0.f, 0.f, 0.f};
    // forward H->H layer, weights in registers (K1's `mid`); returns the pre-activation
    auto mid = [&](const float (&wm)[4 * NWV], const f4 bias, const f4 h) -> f4 {
        put(tile(p, w), h);
        f4 accA = bias, accB = zero4;
        accA = fm4(wm[0], h[0], accA); accB = fm4(wm[1], h[1], accB);
        accA = fm4(wm[2], h[2], accA); accB = fm4(wm[3], h[3], accB);
        __builtin_amdgcn_sched_barrier(0);
        lds_barrier();
        f4 vq[NWV];
        if constexpr (PREFETCH_ALL) {
#pragma unroll
            for (int c = 1; c < NWV; ++c) vq[c] = getl(tile(p, (w + c) & (NWV - 1)));
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int c = 1; c < NWV; ++c) {
            const f4 v = PREFETCH_ALL ? vq[c] : getl(tile(p, (w + c) & (NWV - 1)));
            accA = fm4(wm[4 * c + 0], v[0], accA); accB = fm4(wm[4 * c + 1], v[1], accB);
            accA = fm4(wm[4 * c + 2], v[2], accA); accB = fm4(wm[4 * c + 3], v[3], accB);
        }
        p ^= 1;
        return accA + accB;
    };
    // H->H layer with the image at f4 offset `base` of wT (forward image: pre-activation from `bias`; transposed image: sum_k W[k][own] d[k])
    auto mid_lds = [&](const int base, const f4 bias, const f4 h) -> f4 {
        put(tile(p, w), h);
        const f4* wl = wT + base + w * 64 + l;
        f4 wq = wl[0];
        f4 accA = bias, accB = zero4;
        accA = fm4(wq[0], h[0], accA); accB = fm4(wq[1], h[1], accB);
        accA = fm4(wq[2], h[2], accA); accB = fm4(wq[3], h[3], accB);
        __builtin_amdgcn_sched_barrier(0);
        lds_barrier();
#pragma unroll
        for (int c = 1; c < NWV; ++c) {
            const f4 v = getl(tile(p, (w + c) & (NWV - 1)));
            wq = wl[c * NWV * 64];
            accA = fm4(wq[0], v[0], accA); accB = fm4(wq[1], v[1], accB);
            accA = fm4(wq[2], v[2], accA); accB = fm4(wq[3], v[3], accB);
            if constexpr (BOUND) { if (c % EVERY == EVERY - 1) __builtin_amdgcn_sched_barrier(0); }
        }
        p ^= 1;
        return accA + accB;
    };
    // the same with the image read from a packed tensor in global memory (8 waves: the AE's images; the LDS holds the DE's): all chunks
    // are requested before the exchange so that their latency overlaps it
    auto mid_g = [&](const f4* __restrict__ img, const f4 bias, const f4 h) -> f4 {
        const f4* wlo = img + w * 64 + l;
        asm volatile("" : "+v"(wlo));      // opaque: the loads are loop-invariant and would be hoisted out of the time loop (and spilled)
        const gptr<const f4> wl = (gptr<const f4>)wlo;
        f4 wq[NWV];
#pragma unroll
        for (int c = 0; c < NWV; ++c) wq[c] = wl[c * NWV * 64];
        put(tile(p, w), h);
        f4 accA = bias, accB = zero4;
        accA = fm4(wq[0][0], h[0], accA); accB = fm4(wq[0][1], h[1], accB);
        accA = fm4(wq[0][2], h[2], accA); accB = fm4(wq[0][3], h[3], accB);
        lds_barrier();
#pragma unroll
        for (int c = 1; c < NWV; ++c) {
            const f4 v = getl(tile(p, (w + c) & (NWV - 1)));
            accA = fm4(wq[c][0], v[0], accA); accB = fm4(wq[c][1], v[1], accB);
            accA = fm4(wq[c][2], v[2], accA); accB = fm4(wq[c][3], v[3], accB);
        }
        p ^= 1;
        return accA + accB;
    };
    // transposed DE layer (image in LDS) + the weight gradient of that layer from the very tiles the all-gather published:
    // acc[c] += delta(block (w+c) % NWV)^T (x) hT, hT = this wave's own input activations in operand layout (K4f: midT)
#ifndef PSNODE_K7F_TREAD_AHEAD
#define PSNODE_K7F_TREAD_AHEAD 1     // <= 4 waves: every LDS read of the layer in flight before the first MFMA that needs one (K4f: PSNODE_K4F_TREAD_AHEAD)
#endif
#ifndef PSNODE_K7F_DEFER_DW
#define PSNODE_K7F_DEFER_DW 1        // <= 4 waves: a layer's weight-gradient MFMAs run behind the NEXT exchange's LDS write (K4f: PSNODE_K4F_DEFER_DW)
#endif
    constexpr bool RSM = ROLES && PSNODE_K7F_ROLES_SMALL;
    constexpr bool DEFER = PREFETCH_ALL && PSNODE_K7F_TREAD_AHEAD && PSNODE_K7F_DEFER_DW && !ROLES;
    f4 pendT[DEFER ? NWV : 1], pend_h = zero4;
#ifndef PSNODE_K7F_DEFER8
#define PSNODE_K7F_DEFER8 1         // 8 waves: the same deferral without holding the transposed tiles: they are re-read from the previous exchange's parity
#endif
#ifndef PSNODE_K7F_DW_PAIR
#define PSNODE_K7F_DW_PAIR 4
#endif
    constexpr bool DEFER8 = NWV >= 8 && PSNODE_K7F_DEFER8 && PSNODE_K7F_DW_PAIR;
    int pend_par = 0;
    auto dw_groups = [&](const int par, const f4 hT, f4 (&acc)[NWV]) {
        constexpr int DG = PSNODE_K7F_DW_PAIR <= 1 ? 2 : PSNODE_K7F_DW_PAIR;      // chunks per group
#pragma unroll
        for (int c = 0; c < NWV; c += DG) {
            f4 dTg[DG];
#pragma unroll
            for (int q = 0; q < DG; ++q) dTg[q] = get_row(tile(par, (w + c + q) & (NWV - 1)), roff);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int q = 0; q < DG; ++q) acc[c + q] = fm4(dTg[q][kk], hT[kk], acc[c + q]);
            if constexpr (BOUND) __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto flush = [&](f4 (&pacc)[NWV]) {
        if constexpr (DEFER8) dw_groups(pend_par, pend_h, pacc);
        if constexpr (DEFER) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int c = 0; c < NWV; ++c) pacc[c] = fm4(pendT[c][kk], pend_h[kk], pacc[c]);
        }
    };
    auto midT = [&](const int layer, const f4 d, const f4 hT, f4 (&acc)[NWV], f4 (*pacc)[NWV] = nullptr) -> f4 {
        put(tile(p, w), d);
        const f4* wl = wT + ((size_t)layer * NWV * NWV + w) * 64 + l;
        f4 wq = wl[0];
#ifndef PSNODE_K7F_W_AHEAD
#define PSNODE_K7F_W_AHEAD 1      // two-role chain: the layer's weight chunks are read from LDS BEFORE the exchange's barrier (they do not depend on it): behind it
                                  // only the three tiles of the other waves are left to read -- half of the LDS traffic on the critical path
#endif
        constexpr bool WAH = ROLES && PSNODE_K7F_W_AHEAD;
        f4 wah[WAH ? NWV : 1];
        if constexpr (WAH) {
#pragma unroll
            for (int c = 1; c < NWV; ++c) wah[c] = wl[c * NWV * 64];
        }
        f4 accA = fm4(wq[0], d[0], zero4), accB = fm4(wq[1], d[1], zero4);
        accA = fm4(wq[2], d[2], accA); accB = fm4(wq[3], d[3], accB);
        if constexpr (DEFER || DEFER8) { if (pacc) flush(*pacc); }
        __builtin_amdgcn_sched_barrier(0);
        lds_barrier();
        if constexpr (PREFETCH_ALL && PSNODE_K7F_TREAD_AHEAD) {
            f4 vq[NWV], wqq[NWV], dTq[ROLES ? 1 : NWV];
#pragma unroll
            for (int c = 1; c < NWV; ++c) { vq[c] = getl(tile(p, (w + c) & (NWV - 1))); wqq[c] = WAH ? wah[WAH ? c : 0] : wl[c * NWV * 64]; }
            if constexpr (!ROLES) {
#pragma unroll
                for (int c = 0; c < NWV; ++c) dTq[c] = get_row(tile(p, (w + c) & (NWV - 1)), roff);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 1; c < NWV; ++c) {
                accA = fm4(wqq[c][0], vq[c][0], accA); accB = fm4(wqq[c][1], vq[c][1], accB);
                accA = fm4(wqq[c][2], vq[c][2], accA); accB = fm4(wqq[c][3], vq[c][3], accB);
            }
            if constexpr (ROLES) {
                (void)dTq; (void)hT; (void)acc;       // the gradient waves' work
            } else if constexpr (DEFER) {
#pragma unroll
                for (int c = 0; c < NWV; ++c) pendT[c] = dTq[c];
                pend_h = hT;
            } else {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int c = 0; c < NWV; ++c) acc[c] = fm4(dTq[c][kk], hT[kk], acc[c]);      // NWV independent chains
            }
            p ^= 1;
            return accA + accB;
        }
#ifndef PSNODE_K7F_DP_GROUP
#define PSNODE_K7F_DP_GROUP 0      // 8 waves, transposed layers: tiles and weight chunks of this many chunks read ahead of their MFMAs (0: as the forward layers).
                                     // hidden-128 training step RK4, 0 / 3 / 4 / 7: ODE (K4f) 35.9 / 34.6 / 36.0 / 34.5 ms, DAE (K7f) 52.1 / 51.9 / 51.9 / 52.0 (profiles/r03y_dp_group_ab.txt)
#endif
        if constexpr (PSNODE_K7F_DP_GROUP > 0 && NWV >= 8) {
            constexpr int PG = PSNODE_K7F_DP_GROUP > 0 ? PSNODE_K7F_DP_GROUP : 1;
#pragma unroll
            for (int c0 = 1; c0 < NWV; c0 += PG) {
                f4 vg[PG], wg[PG];
#pragma unroll
                for (int q = 0; q < PG; ++q) if (c0 + q < NWV) { vg[q] = getl(tile(p, (w + c0 + q) & (NWV - 1))); wg[q] = wl[(c0 + q) * NWV * 64]; }
#pragma unroll
                for (int q = 0; q < PG; ++q) if (c0 + q < NWV) {
                    accA = fm4(wg[q][0], vg[q][0], accA); accB = fm4(wg[q][1], vg[q][1], accB);
                    accA = fm4(wg[q][2], vg[q][2], accA); accB = fm4(wg[q][3], vg[q][3], accB);
                }
                if constexpr (BOUND) __builtin_amdgcn_sched_barrier(0);
            }
        } else
#pragma unroll
        for (int c = 1; c < NWV; ++c) {
            const f4 v = getl(tile(p, (w + c) & (NWV - 1)));
            wq = wl[c * NWV * 64];
            accA = fm4(wq[0], v[0], accA); accB = fm4(wq[1], v[1], accB);
            accA = fm4(wq[2], v[2], accA); accB = fm4(wq[3], v[3], accB);
            if constexpr (BOUND) { if (c % EVERY == EVERY - 1) __builtin_amdgcn_sched_barrier(0); }
        }
#ifndef PSNODE_K7F_DW_PAIR
#define PSNODE_K7F_DW_PAIR 4     // 8 waves: the weight-gradient MFMAs of two chunks interleaved (K4f: PSNODE_K4F_DW_PAIR)
#endif
        if constexpr (PSNODE_K7F_DW_PAIR && NWV >= 8) {
            if constexpr (DEFER8) { pend_par = p; pend_h = hT; }      // (the tiles of this parity stay intact until the exchange after next)
            else dw_groups(p, hT, acc);
        } else {
#pragma unroll
        for (int c = 0; c < NWV; ++c) {
            const f4 dT = get_row(tile(p, (w + c) & (NWV - 1)), roff);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc[c] = fm4(dT[kk], hT[kk], acc[c]);
            if constexpr (BOUND) { if (c % EVERY == EVERY - 1) __builtin_amdgcn_sched_barrier(0); }
        }
        }
        p ^= 1;
        return accA + accB;
    };
    // split-K over the waves' own units (4 MFMAs)
    auto own4 = [&](const float (&wq)[4], const f4 h) -> f4 {
        f4 accA = fm4(wq[0], h[0], zero4), accB = fm4(wq[1], h[1], zero4);
        accA = fm4(wq[2], h[2], accA); accB = fm4(wq[3], h[3], accB);
        return accA + accB;
    };
    // all-reduce of the output rows over the waves (fixed order): 2 rows (x-layout) or 4 rows (slot layout)
    auto allreduce2 = [&](const f2 part, const f2 init, f4 (*pacc)[NWV] = nullptr) -> f2 {
        f2* xb2 = reinterpret_cast<f2*>(tile(p, 0));
        xb2[w * 64 + l] = part;
        if constexpr (DEFER || DEFER8) { if (pacc) flush(*pacc); }
        lds_barrier();
        f2 out = init;
#pragma unroll
        for (int c = 0; c < NWV; ++c) out += xb2[c * 64 + l];
        p ^= 1;
        return out;
    };
    auto allreduce4 = [&](const f4 part, const f4 init, f4 (*pacc)[NWV] = nullptr) -> f4 {
        put(tile(p, w), part);
        if constexpr (DEFER || DEFER8) { if (pacc) flush(*pacc); }
        lds_barrier();
        f4 out = init;
#pragma unroll
        for (int c = 0; c < NWV; ++c) out += getl(tile(p, c));
        p ^= 1;
        return out;
    };

    // per-lane BYTE offsets next to a scalar row base (psnode_common.h: sbase / ldg / stg)
    const unsigned offH = 4u * ((unsigned)(b * H) + 16 * w + 4 * g);      // rows of H floats: this lane's 4 units of its trajectory
    const unsigned offX = 4u * ((unsigned)(b * xd) + g);                  // rows of x_dim floats (+ 16 r)
    unsigned offXc[NX];                                                    // the same with the column clamped into the row (ldg_sel)
#pragma unroll
    for (int r = 0; r < NX; ++r) offXc[r] = 4u * ((unsigned)(b * xd) + (4 * r + g < xd ? 4 * r + g : 0));
    const unsigned offI = 4u * (unsigned)(b * idim);                      // rows of i_dim floats (+ 4 column)
    const unsigned offS = 4u * ((unsigned)(b * 16) + g);                  // slot rows (+ 16 m)
    const unsigned offT = 4u * (unsigned)(b * a.t.sb), offZ = 4u * (unsigned)(b * a.z.sb), offV = 4u * (unsigned)(b * a.v.sb);
    const unsigned offZJ = 4u * (unsigned)(b * a.zjb), offVJ = 4u * (unsigned)(b * a.vjb);
    // z | v rows of grid point k (ev >= 0: the jump values of event ev); both sources are read with a clamped column
    struct RowZV { gptr<const float> z, v; unsigned zo, vo; };
    auto zv_rows = [&, offZ, offV, offZJ, offVJ](const long long k, const int ev) -> RowZV {
        RowZV r;
        // (without z / v inputs the row is the clock's: zv_val loads unconditionally -- no branch, not even a uniform one, in front of
        //  the head's MFMAs: psnode_common.h, ldg_sel)
        r.z = sbase(zd > 0 ? (ev >= 0 ? a.zj + (long long)ev * a.zje : a.z.p + k * a.z.st) : a.t.p);
        r.v = sbase(vd > 0 ? (ev >= 0 ? a.vj + (long long)ev * a.vje : a.v.p + k * a.v.st) : a.t.p);
        const unsigned m = ev >= 0 ? ~0u : 0u;     // (bit select: a ?: between the two captured offsets becomes a select between their ADDRESSES)
        r.zo = zd > 0 ? ((offZJ & m) | (offZ & ~m)) : 0u;
        r.vo = vd > 0 ? ((offVJ & m) | (offV & ~m)) : 0u;
        return r;
    };
    auto zv_val = [&](const RowZV& r, const int kind, const int col) -> float {
        const float zr = ldg<float>(r.z, r.zo + 4u * (kind == 0 ? col : 0));
        const float vr = ldg<float>(r.v, r.vo + 4u * (kind == 1 ? col : 0));
        return kind == 0 ? zr : (kind == 1 ? vr : 0.0f);
    };
    auto load_x2 = [&](const float* base, const long long k, float (&dst)[NX]) {
        const gptr<const float> row = sbase(base + k * a.B * xd);
#pragma unroll
        for (int r = 0; r < NX; ++r) dst[r] = ldg_sel(row, offXc[r], 4 * r + g < xd);      // branch-free (psnode_common.h: ldg_sel)
    };
    // AE head, hidden activations of g(xa; z|v of grid point k or of event ev)
    auto ae_hidden = [&](const float (&xa)[NX], const long long k, const int ev, f4& a1, f4& a2, f4& a3) {
        f4 acc = c0a;
        const RowZV zr = zv_rows(k, ev);
#pragma unroll
        for (int r = 0; r < NX; ++r) acc = fm4(aw1x[r], xa[r], acc);
#pragma unroll
        for (int m = 0; m < NZA; ++m) acc = fm4(aw1e[m], zv_val(zr, akind[m], acol[m]), acc);
        a1 = elu_quad(acc);
        if constexpr (AEG) {
            a2 = elu_quad(mid_g(pack_fa, ab2, a1));
            a3 = elu_quad(mid_g(pack_fa + NWV * NWV * 64, ab3, a2));
        } else if constexpr (WREG) {
            a2 = elu_quad(mid(aw2r, ab2, a1));
            a3 = elu_quad(mid(aw3r, ab3, a2));
        }
    };
#ifndef PSNODE_K7F_AE_GRADS
#define PSNODE_K7F_AE_GRADS 1
#endif
    // Round 4, <= 4 waves (hidden <= 64: 512 registers per lane, the AE's transposed images already in LDS): everything that rounds 2-3
    // contracted over the head's STORED rows -- K7h, one more launch, 6 x [T,B,H] rows written here and read back there, ~1.6 ms + ~0.8 ms
    // of host glue per 4096 x 1000 batch -- is formed in this kernel, by the means the DE's gradients are: the transposed layers run
    // through midT (weight gradient of the layer from the tiles the all-gather published), dAW4 / dAW1 from one in-wave transpose each,
    // the head's dL/d(z|v) rides on the all-reduce of its dL/dx, sums for the biases / all_initial stay in registers.  No head row is stored.
    // (saved-activation form only: in the recompute instances -- forward weights of both MLPs in registers on top -- hipcc 7.2's
    //  `AMDGPU Rewrite AGPR-Copy-MFMA` pass segfaults on <RK4, NZM = 4, NZA = 2, 4 waves>; they keep the head rows + K7h)
    constexpr bool AEW = NWV <= 4 && !REC && PSNODE_K7F_AE_GRADS;
    f4 accA3[AEW ? NWV : 1], accA2[AEW ? NWV : 1], accAP3 = zero4, accA1u = zero4, SA1 = zero4, SA2 = zero4, SA3 = zero4, SGi = zero4;
#pragma unroll
    for (int c = 0; c < (AEW ? NWV : 1); ++c) { accA3[c] = zero4; accA2[c] = zero4; }
    // AE head backwards: output adjoint gs (slot layout); xa / zva = the head's inputs (x rows, z | v in the AE's slot layout) for dAW1.
    // Returns dL/dxa and (AEW) the head's dL/d(z|v), slot layout.  Without AEW the rows are written at row index `row` of (ract, rdelta, rgi).
    struct HeadRows { float *a1, *a2, *a3, *d1, *d2, *d3, *gi; };
    struct HeadOut { f2 gx, gz; };
    auto ae_adjoint = [&](const f4 a1, const f4 a2, const f4 a3, const float (&gs)[NZM], const HeadRows hr, const size_t row,
                          const float (&xa)[NX], const float (&zva)[NZA]) -> HeadOut {
        if constexpr (RSM) {      // the head's output adjoint and first-layer input rows for the gradient waves (the same in every chain wave);
                                  // in FRONT of the MFMAs: a uniform branch behind them puts a VALU write on their destination across its taken edge (ISA lint B)
            if (w == 0) {
                put(k7f_roles_g_tile<NWV>(xb), f4{gs[0], NZM > 1 ? gs[NZM > 1 ? 1 : 0] : 0.0f, NZM > 2 ? gs[NZM > 2 ? 2 : 0] : 0.0f, NZM > 3 ? gs[NZM > 3 ? 3 : 0] : 0.0f});
                put(k7f_roles_u_tile<NWV>(xb), f4{xa[0], NX > 1 ? xa[NX > 1 ? 1 : 0] : 0.0f, zva[0], NZA > 1 ? zva[NZA > 1 ? 1 : 0] : 0.0f});
            }
        }
        f4 g3 = zero4;
#pragma unroll
        for (int m = 0; m < NZM; ++m) g3 = fm4(aw4T[m], gs[m], g3);
        const f4 d3 = g3 * elu_grad_quad(a3);
        if constexpr (AEW) {
            SA3 += d3;
            const f4 gs4 = f4{gs[0], NZM > 1 ? gs[NZM > 1 ? 1 : 0] : 0.0f, NZM > 2 ? gs[NZM > 2 ? 2 : 0] : 0.0f, NZM > 3 ? gs[NZM > 3 ? 3 : 0] : 0.0f};
            SGi += gs4;
            if constexpr (!RSM) {   // dAW4[slot of row][own unit] += gs (x) h3, contracted over the tile's trajectories (rows (g, r) <-> slot 4r+g)
                const f4 gT = transpose(gs4);
                const f4 hT = transpose(a3);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) accAP3 = fm4(gT[kk], hT[kk], accAP3);
            }
            const f4 h2T = ROLES ? zero4 : transpose(a2);
            const f4 d2 = midT(3, d3, h2T, accA3) * elu_grad_quad(a2);           // the AE's W3^T (layer 3 of wT), dAW3 from the published tiles
            SA2 += d2;
            const f4 h1T = ROLES ? zero4 : transpose(a1);
            const f4 d1 = midT(2, d2, h1T, accA2, &accA3) * elu_grad_quad(a1);
            SA1 += d1;
            if constexpr (RSM) put(k7f_roles_d1_tile<NWV>(xb, w), d1);
            const f4 ft = own4(afT, d1), fz = own4(afZ, d1);
            const f4 red = allreduce4(f4{ft[0], ft[1], fz[0], fz[1]}, zero4, &accA2);
            if constexpr (!RSM) {   // dAW1 (u columns) += delta1 (x) u, u = (x | z | v) of the head
                const f4 dT = transpose(d1);
                put(scr, f4{xa[0], NX > 1 ? xa[NX > 1 ? 1 : 0] : 0.0f, zva[0], NZA > 1 ? zva[NZA > 1 ? 1 : 0] : 0.0f});
                const f4 sT = arow >= 0 ? get_row(scr, 72 * (arow >> 2) + 4 * g + (arow & 3)) : zero4;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) accA1u = fm4(dT[kk], sT[kk], accA1u);
            }
            return HeadOut{f2{red[0], red[1]}, f2{red[2], red[3]}};
        }
        f4 d2, d1;
        if constexpr (AET_LDS) {
            d2 = mid_lds(TSZ + NWV * NWV * 64, zero4, d3) * elu_grad_quad(a2);
            d1 = mid_lds(TSZ, zero4, d2) * elu_grad_quad(a1);
        } else {
            d2 = mid_g(pack_ta + NWV * NWV * 64, zero4, d3) * elu_grad_quad(a2);
            d1 = mid_g(pack_ta, zero4, d2) * elu_grad_quad(a1);
        }
        const f4 ft = own4(afT, d1);
        const f2 gx = allreduce2(f2{ft[0], ft[1]}, f2{0.f, 0.f});
        if (valid) {
            const size_t rb = row * a.B * H;
            if constexpr (REC) {      // (saved activations: the caller contracts over the forward call's buffers)
                stg<f4>(sbase(hr.a1 + rb), offH, a1);
                stg<f4>(sbase(hr.a2 + rb), offH, a2);
                stg<f4>(sbase(hr.a3 + rb), offH, a3);
            }
            stg<f4>(sbase(hr.d1 + rb), offH, d1);
            stg<f4>(sbase(hr.d2 + rb), offH, d2);
            stg<f4>(sbase(hr.d3 + rb), offH, d3);
            if (w == 0) {
                const gptr<float> gr = sbase(hr.gi + row * a.B * 16);
#pragma unroll
                for (int m = 0; m < NZM; ++m) stg<float>(gr, offS + 16u * m, gs[m]);
            }
        }
        return HeadOut{gx, f2{0.f, 0.f}};
    };
    // the head's z | v inputs at grid point kk (un-jumped) or at event ev, AE slot layout
    auto head_zv = [&](const long long kk, const int ev_, float (&zva)[NZA]) {
        const RowZV zr = zv_rows(kk, ev_);
#pragma unroll
        for (int m = 0; m < NZA; ++m) zva[m] = zv_val(zr, akind[m], acol[m]);
    };
    // dL/dis[k] enters through the `s`-block slot of its i-dim (one slot per i-dim); padding trajectories carry no adjoint at all
    auto add_gis = [&](const long long k, float (&gs)[NZM]) {
        // grad_is == NULL reads the rows of `is` instead (same shape) and masks them out: no branch (psnode_dae_backward_wide.hip: add_gis)
        const bool has = a.gis != nullptr;
        const gptr<const float> row = sbase((has ? a.gis : a.is) + k * a.B * idim);
#pragma unroll
        for (int m = 0; m < NZM; ++m) {
            const float q = ldg<float>(row, offI + 4u * (ekind[m] == 2 ? ecol[m] : 0));
            gs[m] += (has && valid && ekind[m] == 2 && 4 * m + g >= ne) ? q : 0.0f;
        }
    };

    // saved head activations of grid point kk ([3,T,B,H])
    auto load_head = [&](const long long kk, f4& a1, f4& a2, f4& a3) {
        const size_t lay = (size_t)a.T * a.B * H;
        const float* rb = a.saeact + (size_t)kk * a.B * H;
        a1 = ldg_nt<f4, (NWV >= 8)>(sbase(rb), offH);
        a2 = ldg_nt<f4, (NWV >= 8)>(sbase(rb + lay), offH);
        a3 = ldg_nt<f4, (NWV >= 8)>(sbase(rb + 2 * lay), offH);
    };
    const HeadRows grid_rows{a.aact[0], a.aact[1], a.aact[2], a.adelta[0], a.adelta[1], a.adelta[2], a.agi};
    const HeadRows event_rows{a.eact[0], a.eact[1], a.eact[2], a.edelta[0], a.edelta[1], a.edelta[2], a.egi};

    // ---- accumulators (whole launch)
    f4 accW3[NWV], accW2[NWV], accW4 = zero4, accW1s = zero4, S1 = zero4, S2 = zero4, S3 = zero4;
#pragma unroll
    for (int c = 0; c < NWV; ++c) { accW3[c] = zero4; accW2[c] = zero4; }
    f2 db4 = f2{0.f, 0.f};
    float gcar[NX], gsl[NZM];
#pragma unroll
    for (int r = 0; r < NX; ++r) gcar[r] = 0.0f;
#pragma unroll
    for (int m = 0; m < NZM; ++m) gsl[m] = 0.0f;

    const long long nrow = a.B, nT = a.T;
    // REC = false: rows of (step, stage), linear index idx = k S + s, walked downwards
    const long long act_layer = a.B * H;
    auto load_saved = [&](const long long idx, f4& q1, f4& q2, f4& q3, float (&xq)[NX]) {
        const float* rb = a.sact + (size_t)idx * 3 * act_layer;
        q1 = ldg_nt<f4, (NWV >= 8)>(sbase(rb), offH);
        q2 = ldg_nt<f4, (NWV >= 8)>(sbase(rb + act_layer), offH);
        q3 = ldg_nt<f4, (NWV >= 8)>(sbase(rb + 2 * act_layer), offH);
        const gptr<const float> xrow = sbase(a.sxst + idx * a.B * xd);       // RAW (requested a stage ahead): masked where X[s] is formed
#pragma unroll
        for (int r = 0; r < NX; ++r) xq[r] = ldg<float>(xrow, offXc[r]);
    };
    f4 sv1 = zero4, sv2 = zero4, sv3 = zero4;
    float svx[NX] = {};
#ifndef PSNODE_K7F_HEAD_AHEAD
#define PSNODE_K7F_HEAD_AHEAD 1      // REC = false: the saved rows of the next grid point's head are requested late in the step's last stage (next to
                                     // the next stage's rows) instead of at the top of the next step, where the head needs them at once
#endif
    // (<= 4 waves only: hidden-64 training step 20.9 -> 20.6 ms; at 8 waves the twelve registers cost more in spills than the latency they
    //  hide -- RK4 49.9 -> 56.3 ms, profiles/r03y_head_ahead_ab.txt)
    constexpr bool HEAD_AHEAD = !REC && NWV <= 4 && PSNODE_K7F_HEAD_AHEAD && !ROLES;
    f4 hn1 = zero4, hn2 = zero4, hn3 = zero4;
    if constexpr (!REC && !ROLES) {
        if (nT >= 2) load_saved((nT - 2) * S + (S - 1), sv1, sv2, sv3, svx);
        if constexpr (HEAD_AHEAD) load_head(nT - 1, hn1, hn2, hn3);
    }
    // ROLES: the saved rows arrive through the mailboxes the gradient wave with the same units fills (it requests them two stages / a
    // whole step ahead): this wave issues no global load for them, and holds no register set of rows in flight
    auto mailbox = [&](const int kind, f4& q1, f4& q2, f4& q3) {
        const f4* m4 = reinterpret_cast<const f4*>(k7f_roles_mailbox<NWV>(xb, kind, w)) + l;
        q1 = m4[0]; q2 = m4[64]; q3 = m4[128];
    };
#ifndef PSNODE_K7F_INPUTS_AHEAD
#define PSNODE_K7F_INPUTS_AHEAD 1
#endif
    // Round 4, <= 4 waves: the per-step inputs -- event index, dL/dis and dL/dxs rows, the z | v | i rows, the clock, (recompute form) the
    // state row -- are requested one step AHEAD, in the middle of the previous step, and stay raw until the step consumes them.  Rounds
    // 2-3 loaded each of them where it was used: six `global_load ..; s_waitcnt vmcnt(0)` round trips per step, each of which also waited
    // for the head's row stores issued just before (vmcnt counts stores) -- SQ_WAIT_ANY 48 % of the saved-activation Euler instance
    // (profiles/r04_train_h64_sq_table.txt).  The z | v gradient store of a step is deferred to the next step (in front of the requests)
    // so that the wait at the top of a step only sees loads.  (8 waves: the ~14 registers would be spills; unchanged there.)
    constexpr bool AHEAD = NWV <= 4 && PSNODE_K7F_INPUTS_AHEAD;
    const bool has_gis = a.gis != nullptr;
    auto load_t = [&](const long long kk) -> float { return ldg<float>(sbase(a.t.p + kk * a.t.st), offT); };
    // RAW inputs of step k: dL/dis row k+1 (row of `is` when grad_is is NULL: masked out at the consumer), dL/dxs row k+1, the z / v / i
    // candidates of every ext slot of step k.  No select, mask or subtraction in here: an operation on a freshly loaded value is scheduled
    // next to the load, with the wait for it (DESIGN.md "what the ISA said about waits"); finish_step() applies them a step later.
    struct StepRaw { float gq[NZM], gin[NX], zr[NZM], vr[NZM], ir[NZM], xr[NX], hx[NX], hz[NZA], hv[NZA]; };
    auto load_x2_raw = [&](const float* base, const long long k_, float (&dst)[NX]) {
        const gptr<const float> row = sbase(base + k_ * a.B * xd);
#pragma unroll
        for (int r = 0; r < NX; ++r) dst[r] = ldg<float>(row, offXc[r]);
    };
    auto fetch_step = [&](const long long k_, const int ev_, StepRaw& q) {
        const gptr<const float> grow = sbase((has_gis ? a.gis : a.is) + (k_ + 1) * a.B * idim);
#pragma unroll
        for (int m = 0; m < NZM; ++m) q.gq[m] = ldg<float>(grow, offI + 4u * (ekind[m] == 2 ? ecol[m] : 0));
        load_x2_raw(a.gxs, k_ + 1, q.gin);
        const RowZV zr = zv_rows(k_, ev_);
        // (saved form, event step: i0 as the forward call computed it, slot layout [nE,B,16] -- the row POINTER and the lane offset are
        //  selected, the load is one and unconditional: a branch around it would put a wait behind the join of every step)
        const bool evs = !REC && ev_ >= 0;
        const gptr<const float> irow = sbase(evs ? a.sevi + (size_t)(evs ? ev_ : 0) * a.B * 16 : ((REC && a.ti) ? a.itrue : a.is) + k_ * a.B * idim);
#pragma unroll
        for (int m = 0; m < NZM; ++m) {
            q.zr[m] = ldg<float>(zr.z, zr.zo + 4u * (ekind[m] == 0 ? ecol[m] : 0));
            q.vr[m] = ldg<float>(zr.v, zr.vo + 4u * (ekind[m] == 1 ? ecol[m] : 0));
            q.ir[m] = ldg<float>(irow, evs ? offS + 16u * m : offI + 4u * (ekind[m] == 2 ? ecol[m] : 0));
        }
        if constexpr (REC) load_x2_raw(a.tx ? a.xtrue : a.xs, k_, q.xr);
        if constexpr (AEW) {     // the inputs of the head at grid point k+1 (for dAW1): its x row and the un-jumped z | v rows
            load_x2_raw((REC && a.tx) ? a.xtrue : a.xs, k_ + 1, q.hx);
            const RowZV zh = zv_rows(k_ + 1, -1);
#pragma unroll
            for (int m = 0; m < NZA; ++m) {
                q.hz[m] = ldg<float>(zh.z, zh.zo + 4u * (akind[m] == 0 ? acol[m] : 0));
                q.hv[m] = ldg<float>(zh.v, zh.vo + 4u * (akind[m] == 1 ? acol[m] : 0));
            }
        }
    };
    auto finish_step = [&](const StepRaw& q, float (&gq_)[NZM], float (&gin_)[NX], float (&ext_)[NZM], float (&x0_)[NX], float (&xh_)[NX],
                           float (&zvh_)[NZA]) {
#pragma unroll
        for (int m = 0; m < NZA; ++m) zvh_[m] = AEW ? (akind[m] == 0 ? q.hz[m] : (akind[m] == 1 ? q.hv[m] : 0.0f)) : 0.0f;
#pragma unroll
        for (int r = 0; r < NX; ++r) xh_[r] = (AEW && 4 * r + g < xd) ? q.hx[r] : 0.0f;
#pragma unroll
        for (int m = 0; m < NZM; ++m) {
            gq_[m] = q.gq[m];
            ext_[m] = ekind[m] == 0 ? q.zr[m] : (ekind[m] == 1 ? q.vr[m] : (ekind[m] == 2 ? q.ir[m] : 0.0f));
        }
#pragma unroll
        for (int r = 0; r < NX; ++r) { gin_[r] = (4 * r + g < xd) ? q.gin[r] : 0.0f; x0_[r] = (REC && 4 * r + g < xd) ? q.xr[r] : 0.0f; }
    };
    int lane_zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));
    const bool has_ev = a.ev != nullptr;
    const int* evp = (has_ev ? a.ev : reinterpret_cast<const int*>(a.t.p)) + lane_zero;      // per-lane load: the entry stays in a VGPR until it is used
    StepRaw nxt = {};                        // raw inputs of the step about to run (requested a step ago)
    float x1_c[NX] = {};                     // recompute form: the state row of grid point k+1 (= the previous step's start row)
    float t_hi = 0.0f, t_lo = 0.0f;
    int ev_n = -1, ev_r = -1;
    float gz_pend[NZM] = {};                 // z | v gradient of the previous step, stored at the next one
    int ev_pend = -1;
    long long k_pend = AEW ? a.T - 1 : -1;   // (AEW: grid point T-1 has no DE part, only its head's)
    // gzh: (AEW) the share of the head at grid point k_pend, slot layout -- un-jumped rows, whatever the step did
    auto flush_gzv = [&](const f2 gzh) {
        if (k_pend >= 0 && w == 0 && valid && (a.gzv != nullptr)) {
#pragma unroll
            for (int m = 0; m < NZM; ++m) {
                const int q = 4 * m + g;
                if (q < nzv) {
                    if (ev_pend >= 0) a.gjump[(b * a.n_events + ev_pend) * nzv + q] = gz_pend[m];
                    a.gzv[(k_pend * a.B + b) * nzv + q] = (ev_pend >= 0 ? 0.0f : gz_pend[m]) + (m < 2 ? gzh[m < 2 ? m : 0] : 0.0f);
                }
            }
        }
    };
    if constexpr (AHEAD) {
        if (nT >= 2) {
            ev_n = has_ev ? __builtin_amdgcn_readfirstlane(a.ev[nT - 2]) : -1;
            ev_r = evp[nT >= 3 ? nT - 3 : 0];
            fetch_step(nT - 2, ev_n, nxt);
            t_hi = load_t(nT - 1);
            t_lo = load_t(nT - 2);
            if constexpr (REC) load_x2(a.tx ? a.xtrue : a.xs, nT - 1, x1_c);
        }
    }
    if constexpr (ROLES) lds_barrier();      // the first rows are in the mailboxes
    for (long long k = nT - 2; k >= 0; --k) {
        int ev;
        float gq[NZM], gin[NX], ext[NZM], x0[NX] = {}, x1[NX] = {}, xh[NX], zvh[NZA], h_;
        if constexpr (AHEAD) {
            ev = ev_n;
            finish_step(nxt, gq, gin, ext, x0, xh, zvh);
#pragma unroll
            for (int r = 0; r < NX; ++r) x1[r] = x1_c[r];
            h_ = t_hi - t_lo;
        } else {
            ev = a.ev ? __builtin_amdgcn_readfirstlane(a.ev[k]) : -1;
            if constexpr (REC) load_x2(a.tx ? a.xtrue : a.xs, k + 1, x1);
            StepRaw now;
            fetch_step(k, ev, now);
            finish_step(now, gq, gin, ext, x0, xh, zvh);
            h_ = load_t(k + 1) - load_t(k);
        }
        // ---- (1) AE head at grid point k+1 (my_solvers.py:121): adjoint = dL/dis[k+1] + the algebraic adjoint of step k+1's DE
        f2 gzh_k = f2{0.f, 0.f};
        {
            f4 a1, a2, a3;
            if constexpr (REC) {
                ae_hidden(x1, k + 1, -1, a1, a2, a3);
            } else if constexpr (ROLES) {
                mailbox(1, a1, a2, a3);
            } else if constexpr (HEAD_AHEAD) {
                a1 = hn1; a2 = hn2; a3 = hn3;
            } else {
                load_head(k + 1, a1, a2, a3);
            }
#pragma unroll
            for (int m = 0; m < NZM; ++m) gsl[m] += (has_gis && valid && ekind[m] == 2 && 4 * m + g >= ne) ? gq[m] : 0.0f;      // (add_gis, on the raw row)
            const HeadOut ho = ae_adjoint(a1, a2, a3, gsl, grid_rows, (size_t)(k + 1), xh, zvh);
            gcar[0] += (REC && a.tx) ? 0.0f : ho.gx[0];          // (teacher-forced x: the head read a dataset row)
            if constexpr (NX > 1) gcar[1] += (REC && a.tx) ? 0.0f : ho.gx[1];
            gzh_k = ho.gz;
        }
        if constexpr (AHEAD) {       // behind the head's row stores: last step's z | v gradient, then the requests for the next step
            flush_gzv(gzh_k);
            const long long kq = k > 0 ? k - 1 : 0;
            ev_n = has_ev ? __builtin_amdgcn_readfirstlane(ev_r) : -1;      // requested a step ago
            ev_r = evp[k >= 2 ? k - 2 : 0];
            fetch_step(kq, ev_n, nxt);
            t_hi = t_lo;
            t_lo = load_t(kq);
            if constexpr (REC) {
#pragma unroll
                for (int r = 0; r < NX; ++r) x1_c[r] = x0[r];       // this step's start row is the next head's grid point
            }
        }
        // ---- (2) DE step k
        if constexpr (REC)
        if (ev >= 0) {   // event: i0 = g(x0; z_jump, v_jump) (my_solvers.py:108-110); its rows travel through the event buffers
            f4 e1, e2, e3;
            float xe[NX];      // the event's head reads the RUNNING state, teacher forcing or not (my_solvers.py:108-110)
#pragma unroll
            for (int r = 0; r < NX; ++r) xe[r] = x0[r];
            if (a.tx) load_x2(a.xs, k, xe);
            ae_hidden(xe, k, ev, e1, e2, e3);
            f4 sb4 = ab4;
            float sw4[4] = {aw4[0], aw4[1], aw4[2], aw4[3]};
            if constexpr (AEG) {        // event steps are rare: their output layer is read where it is used, not kept for the whole launch
                const float* pwo = pwa;
                asm volatile("" : "+v"(pwo));
                const gptr<const float> pws = (gptr<const float>)pwo;
#pragma unroll
                for (int r = 0; r < 4; ++r) { sb4[r] = pws[(RA::B4 + r) * 64]; sw4[r] = pws[(RA::W4 + r) * 64]; }
            }
            const f4 i0 = allreduce4(own4(sw4, e3), sb4);
#pragma unroll
            for (int m = 0; m < NZM; ++m) if (ekind[m] == 2 && !a.ti) ext[m] = i0[m];      // (teacher-forced i: the DE keeps the dataset row)
            if (valid) {
                const size_t rb = (size_t)ev * a.B * H;
                stg<f4>(sbase(a.eact[0] + rb), offH, e1);
                stg<f4>(sbase(a.eact[1] + rb), offH, e2);
                stg<f4>(sbase(a.eact[2] + rb), offH, e3);
                if (w == 0) {
                    const gptr<float> er = sbase(a.ei + (size_t)ev * a.B * 16);
#pragma unroll
                    for (int m = 0; m < NZM; ++m) stg<float>(er, offS + 16u * m, i0[m]);
                }
            }
        }
        f4 cz = c0;
#pragma unroll
        for (int m = 0; m < NZM; ++m) cz = fm4(w1z[m], ext[m] - a0e[m], cz);

        // ---- phase A: stage evaluations (K1's plan)
        float X[S][NX], ks[S][NX];
        f4 h1[WREG ? S : 1], h2[WREG ? S : 1], h3[WREG ? S : 1];
        f4 la1 = zero4, la2 = zero4, la3 = zero4;    // ELU outputs of the last stage evaluated (the first one the backward half needs)
        if constexpr (STREAM) dma_wait();            // forward images of both DE layers (refilled behind their last use of the previous phase B)
#pragma unroll
        for (int s = 0; s < (REC ? S : 0); ++s) {
#pragma unroll
            for (int r = 0; r < NX; ++r) {
                float acc = 0.0f;
#pragma unroll
                for (int jj = 0; jj < s; ++jj) acc += rk_a(METHOD, s, jj) * ks[jj][r];
                X[s][r] = s == 0 ? x0[r] : x0[r] + h_ * acc;
            }
            f4 accA = cz, accB = zero4;
#pragma unroll
            for (int r = 0; r < NX; ++r) {
                if (r & 1) accB = fm4(w1xs[r], X[s][r], accB);
                else accA = fm4(w1xs[r], X[s][r], accA);
            }
            const f4 a1 = elu_quad(NX > 1 ? accA + accB : accA);
            f4 a2, a3;
            if constexpr (STREAM) {
                a2 = elu_quad(mid_lds(0, b2, a1));
                if (s == S - 1) dma_layer(pack_t, 0);      // W2's region is free until phase B's SECOND transposed layer
                a3 = elu_quad(mid_lds(NWV * NWV * 64, b3, a2));
                if (s == S - 1) dma_layer(pack_t, 1);
                if (s < S - 1) {     // park the stage's activations in the ring (each lane re-reads exactly what it wrote)
                    const size_t rb = (size_t)(3 * s) * nrow * H;
                    stg<f4>(sbase(a.ring + rb), offH, a1);
                    stg<f4>(sbase(a.ring + rb + nrow * H), offH, a2);
                    stg<f4>(sbase(a.ring + rb + 2 * nrow * H), offH, a3);
                }
            } else if constexpr (WREG) {
                a2 = elu_quad(mid(w2r, b2, a1)); a3 = elu_quad(mid(w3r, b3, a2));
                h1[s] = a1; h2[s] = a2; h3[s] = a3;
            }
            if (s == S - 1) { la1 = a1; la2 = a2; la3 = a3; }
            if (s < S - 1) {         // the last stage's derivative feeds x[k+1] only, which the backward does not need
                const f4 part = own4(w4, a3);
                const f2 kk = allreduce2(f2{part[0], part[1]}, f2{b4[0], b4[1]});
                ks[s][0] = kk[0];
                if constexpr (NX > 1) ks[s][1] = kk[1];
            }
        }

        // ---- phase B: stages backwards; D1 = sum over the stages of delta1 (the external inputs are frozen over the step)
        float gks[S][NX], gx0[NX];
#pragma unroll
        for (int r = 0; r < NX; ++r) {
            const float g1 = gcar[r] + (valid ? gin[r] : 0.0f);
            gx0[r] = g1;
#pragma unroll
            for (int s = 0; s < S; ++s) gks[s][r] = (h_ * rk_b(METHOD, s)) * g1;
        }
        f4 D1 = zero4;
        f4 na1 = la1, na2 = la2, na3 = la3;       // STREAM: activations of the stage handled next, requested late in the previous stage
#pragma unroll
        for (int s = S - 1; s >= 0; --s) {
            f4 a1, a2, a3;
            if constexpr (ROLES) {
                mailbox(0, a1, a2, a3);
                const f2 xq = reinterpret_cast<const f2*>(k7f_roles_xbox<NWV>(xb, w))[l];
#pragma unroll
                for (int r = 0; r < NX; ++r) X[s][r] = (4 * r + g < xd) ? xq[r & 1] : 0.0f;
            } else if constexpr (!REC) {
                a1 = sv1; a2 = sv2; a3 = sv3;
#pragma unroll
                for (int r = 0; r < NX; ++r) X[s][r] = (4 * r + g < xd) ? svx[r] : 0.0f;
            } else if constexpr (STREAM) { a1 = na1; a2 = na2; a3 = na3; }
            else { a1 = h1[s]; a2 = h2[s]; a3 = h3[s]; }
            const f2 gk = f2{gks[s][0], NX > 1 ? gks[s][1] : 0.0f};
            db4 += gk;
            if constexpr (RSM) {      // gk and the stage input for the gradient waves (every chain wave holds the same values); in front of the MFMAs
                if (w == 0) {
                    put(k7f_roles_g_tile<NWV>(xb), f4{gk[0], gk[1], 0.f, 0.f});
                    put(k7f_roles_u_tile<NWV>(xb), f4{X[s][0], NX > 1 ? X[s][1] : 0.0f, g < ne ? ext[0] : 0.0f, (NZM > 1 && 4 + g < ne) ? ext[NZM > 1 ? 1 : 0] : 0.0f});
                }
            }
            f4 g3 = zero4;
#pragma unroll
            for (int r = 0; r < NX; ++r) g3 = fm4(w4T[r], gks[s][r], g3);
            const f4 d3 = g3 * elu_grad_quad(a3);
            S3 += d3;
            if constexpr (!RSM) {   // dW4[x-dim of row][own unit] += gk (x) h3, contracted over the tile's trajectories
                const f4 gT = transpose(f4{gk[0], gk[1], 0.f, 0.f});
                const f4 hT = transpose(a3);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) accW4 = fm4(gT[kk], hT[kk], accW4);
            }
            const f4 h2T = ROLES ? zero4 : transpose(a2);
            if constexpr (STREAM) { if (s == S - 1) dma_wait(); }     // the transposed images must have landed
            const f4 d2 = midT(1, d3, h2T, accW3) * elu_grad_quad(a2);
            S2 += d2;
            if constexpr (STREAM) { if (s == 0 && k > 0) dma_layer(pack_f, 1); }     // W3's region: forward image for the next step
            const f4 h1T = ROLES ? zero4 : transpose(a1);
            const f4 d1 = midT(0, d2, h1T, accW2, &accW3) * elu_grad_quad(a1);
            D1 += d1;
            if constexpr (RSM) put(k7f_roles_d1_tile<NWV>(xb, w), d1);
            if constexpr (STREAM) { if (s == 0 && k > 0) dma_layer(pack_f, 0); }
            const f4 ft = own4(fT, d1);
            if constexpr (STREAM) {
                if (s > 0) {
                    const size_t rb = (size_t)(3 * (s - 1)) * nrow * H;
                    na1 = ldg<f4>(sbase(a.ring + rb), offH);
                    na2 = ldg<f4>(sbase(a.ring + rb + nrow * H), offH);
                    na3 = ldg<f4>(sbase(a.ring + rb + 2 * nrow * H), offH);
                }
            }
            if constexpr (!REC && !ROLES) {      // the next stage's rows, requested late in this one (K4f: PSNODE_K4F_SAVED_AHEAD)
                const long long idx = k * S + s;
                load_saved(idx > 0 ? idx - 1 : 0, sv1, sv2, sv3, svx);
                if constexpr (HEAD_AHEAD) { if (s == 0) load_head(k, hn1, hn2, hn3); }      // grid point k: the head of the next iteration (or the one behind the loop)
            }
            const f2 gx = allreduce2(f2{ft[0], ft[1]}, f2{0.f, 0.f}, &accW2);
            if constexpr (!RSM) {   // dW1 (`s` columns) += delta1 (x) s, s = (X_s | ext)
                const f4 dT = transpose(d1);
                put(scr, f4{X[s][0], NX > 1 ? X[s][1] : 0.0f, g < ne ? ext[0] : 0.0f, (NZM > 1 && 4 + g < ne) ? ext[NZM > 1 ? 1 : 0] : 0.0f});
                const f4 sT = srow >= 0 ? get_row(scr, 72 * (srow >> 2) + 4 * g + (srow & 3)) : zero4;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) accW1s = fm4(dT[kk], sT[kk], accW1s);
            }
#pragma unroll
            for (int r = 0; r < NX; ++r) {
                const float gxr = r == 0 ? gx[0] : gx[1];
                gx0[r] += gxr;
#pragma unroll
                for (int jj = 0; jj < s; ++jj) gks[jj][r] += (h_ * rk_a(METHOD, s, jj)) * gxr;
            }
        }
        S1 += D1;
        const f4 gE = allreduce4(own4(fE, D1), zero4);       // adjoint of this step's external inputs, slot layout
#pragma unroll
        for (int r = 0; r < NX; ++r) gcar[r] = (REC && a.tx) ? 0.0f : gx0[r];        // (the step started from a dataset row)
#pragma unroll
        for (int m = 0; m < NZM; ++m) gsl[m] = (REC && a.ti) ? 0.0f : gE[m];           // (the DE read a dataset row of i)
        // z | v columns: final layout (an event step's belong to the jump values); AHEAD: stored at the next step, in front of its requests
#pragma unroll
        for (int m = 0; m < NZM; ++m) gz_pend[m] = gE[m];
        ev_pend = ev;
        k_pend = k;
        if constexpr (!AHEAD) flush_gzv(f2{0.f, 0.f});
        // ---- (3) event: that adjoint belongs to the recomputed i0, whose head is run backwards here; grid point k's own head
        //          (is[k], un-jumped) then only sees dL/dis[k]
        if (ev >= 0) {
            f4 e1, e2, e3;
            if constexpr (REC) {
                const size_t rb = (size_t)ev * a.B * H;                          // this lane's own rows, written above
                e1 = ldg<f4>(sbase(a.eact[0] + rb), offH);
                e2 = ldg<f4>(sbase(a.eact[1] + rb), offH);
                e3 = ldg<f4>(sbase(a.eact[2] + rb), offH);
            } else {
                const float* rb = a.sevact + (size_t)ev * 3 * a.B * H;
                e1 = ldg<f4>(sbase(rb), offH);
                e2 = ldg<f4>(sbase(rb + a.B * H), offH);
                e3 = ldg<f4>(sbase(rb + 2 * a.B * H), offH);
            }
            float xev[NX] = {}, zvev[NZA] = {};
            if constexpr (AEW) {      // the event head's inputs (for dAW1): the RUNNING state of step k and the jump values
                load_x2(a.xs, k, xev);
                head_zv(k, ev, zvev);
            }
            const HeadOut he = ae_adjoint(e1, e2, e3, gsl, event_rows, (size_t)ev, xev, zvev);
            gcar[0] += he.gx[0];
            if constexpr (NX > 1) gcar[1] += he.gx[1];
            if constexpr (AEW) {      // its dL/d(z|v) belongs to the jump values, like the step's own
                gz_pend[0] += he.gz[0];
                if constexpr (NZM > 1) gz_pend[1] += he.gz[1];
            }
#pragma unroll
            for (int m = 0; m < NZM; ++m) gsl[m] = 0.0f;
        }
    }
    if constexpr (AHEAD && !AEW) flush_gzv(f2{0.f, 0.f});
    if constexpr (STREAM) dma_wait();
    {   // the head at grid point 0 (my_solvers.py:95): i_0 = g(x_init; z_0, v_0)
        f4 a1, a2, a3;
        if constexpr (REC) {
            float x1[NX];
            load_x2(a.tx ? a.xtrue : a.xs, 0, x1);
            ae_hidden(x1, 0, -1, a1, a2, a3);
        } else if constexpr (ROLES) {
            mailbox(1, a1, a2, a3);
        } else if constexpr (HEAD_AHEAD) {
            if (nT >= 2) { a1 = hn1; a2 = hn2; a3 = hn3; }
            else load_head(0, a1, a2, a3);
        } else {
            load_head(0, a1, a2, a3);
        }
        add_gis(0, gsl);
        float xh0[NX] = {}, zvh0[NZA] = {};
        if constexpr (AEW) { load_x2((REC && a.tx) ? a.xtrue : a.xs, 0, xh0); head_zv(0, -1, zvh0); }
        const HeadOut ho = ae_adjoint(a1, a2, a3, gsl, grid_rows, (size_t)0, xh0, zvh0);
        gcar[0] += (REC && a.tx) ? 0.0f : ho.gx[0];
        if constexpr (NX > 1) gcar[1] += (REC && a.tx) ? 0.0f : ho.gx[1];
        if constexpr (AEW) flush_gzv(ho.gz);       // grid point 0: step 0's DE part (pending) + this head's
    }

    // ---- epilogue
    if (w == 0 && valid) {
#pragma unroll
        for (int r = 0; r < NX; ++r) if (4 * r + g < xd) stg<float>(sbase(a.carry_x), offX + 16u * r, gcar[r]);
    }
    const int K1 = 3 * n;
    if constexpr (RSM) {      // for the gradient wave with these units (read behind the barriers below): padded layout
        put(scr, S1);
        put(k7f_roles_mailbox<NWV>(xb, 0, w), SA1);
    }
    {   // DE part of d all_initial[c] = sum_u (Wa - Wd)[u][c] S1[u]: split-K over the waves' own units, all-reduce, rows c = 4g + r
        float at[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int uu = 16 * w + 4 * g + r;
            at[r] = (i < n && uu < HR) ? a.w1[(size_t)uu * K1 + i] - a.w1[(size_t)uu * K1 + n + i] : 0.0f;
        }
        f4 ga = allreduce4(own4(at, S1), zero4);
        if constexpr (AEW) {      // + the AE's share: sum_u AW1[u][c] SA1[u] (its first-layer input starts with all_initial, no difference term)
            const int K1a = n + xd + nzv;
            float ata[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int uu = 16 * w + 4 * g + r;
                ata[r] = (i < n && uu < HR) ? a.aw1[(size_t)uu * K1a + i] : 0.0f;
            }
            ga += allreduce4(own4(ata, SA1), zero4);
        }
        if (w == 0 && valid) {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (4 * g + r < n) a.ga0[b * n + 4 * g + r] = ga[r];
        }
    }
    // parameter-gradient partials of this workgroup, nn.Linear order with the MLP's real width HR as row stride
    float* wp = a.wpart + (size_t)blockIdx.x * a.NP;
    const int oB1 = HR * K1, oW2 = oB1 + HR, oB2 = oW2 + HR * HR, oW3 = oB2 + HR, oB3 = oW3 + HR * HR, oW4 = oB3 + HR, oB4 = oW4 + xd * HR;
    if constexpr (!RSM) {
        // dW1: columns [a0 | s-a0 | s]; ca0 = sum(delta1) (x) a0 over the tile's trajectories
        const f4 sT = transpose(S1);
        f4 ca0 = zero4;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const long long tb = b0 + 4 * kk + g;
            const float av = (i < n && tb < a.B) ? a.a0[tb * n + i] : 0.0f;
            ca0 = fm4(sT[kk], av, ca0);
        }
        if (j < n) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int u = 16 * w + 4 * g + r;
                if (u < HR) {
                    float* row = wp + (size_t)u * K1;
                    row[j] = ca0[r];
                    row[n + j] = accW1s[r] - ca0[r];
                    row[2 * n + j] = accW1s[r];
                }
            }
        }
    }
    {
        const int v = 16 * w + j;        // own column
#pragma unroll
        for (int c = 0; c < (ROLES ? 0 : NWV); ++c) {      // (ROLES: written by the gradient waves)
            const int ub = 16 * ((w + c) & (NWV - 1)) + 4 * g;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (ub + r < HR && v < HR) {
                    wp[oW2 + (size_t)(ub + r) * HR + v] = accW2[c][r];
                    wp[oW3 + (size_t)(ub + r) * HR + v] = accW3[c][r];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < (RSM ? 0 : 4); ++r) {   // dW4 rows (g, r) <-> x-dim 4r+g, columns = own units
            const int dd = 4 * r + g;
            if (r < NX && dd < xd && v < HR) wp[oW4 + (size_t)dd * HR + v] = accW4[r];
        }
    }
    // biases: row sums over the 16 trajectories of a lane group
    f4 sb1 = S1, sb2 = S2, sb3 = S3;
    f2 sb4 = db4;
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sb1[r] += __shfl_xor(sb1[r], m, 64); sb2[r] += __shfl_xor(sb2[r], m, 64); sb3[r] += __shfl_xor(sb3[r], m, 64);
        }
        sb4[0] += __shfl_xor(sb4[0], m, 64); sb4[1] += __shfl_xor(sb4[1], m, 64);
    }
    if (j == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int u = 16 * w + 4 * g + r;
            if (u < HR) { wp[oB1 + u] = sb1[r]; wp[oB2 + u] = sb2[r]; wp[oB3 + u] = sb3[r]; }
        }
        if (w == 0) {
#pragma unroll
            for (int r = 0; r < NX; ++r) if (4 * r + g < xd) wp[oB4 + 4 * r + g] = sb4[r];
        }
    }
    if constexpr (AEW) {
        // the AE head's partials: [dAW1 (HR x K1a) | db1 | dAW2 | db2 | dAW3 | db3 | P3 (16 slots x HR) | sum gi (16 slots)] -- the caller
        // folds the two slots of an i-dim (P3, sum gi) into dAW4 / db4 as it did with K7h's output
        float* wa = a.wpart_ae + (size_t)blockIdx.x * a.NPA;
        const int K1a = n + xd + nzv;
        const int aB1 = HR * K1a, aW2 = aB1 + HR, aB2 = aW2 + HR * HR, aW3 = aB2 + HR, aB3 = aW3 + HR * HR, aP3 = aB3 + HR, aSG = aP3 + 16 * HR;
        if constexpr (!RSM) {
            const f4 sT = transpose(SA1);
            f4 ca0 = zero4;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const long long tb = b0 + 4 * kk + g;
                const float av = (i < n && tb < a.B) ? a.a0[tb * n + i] : 0.0f;
                ca0 = fm4(sT[kk], av, ca0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int u = 16 * w + 4 * g + r;
                if (u < HR) {
                    float* row = wa + (size_t)u * K1a;
                    if (j < n) row[j] = ca0[r];
                    if (j < xd + nzv) row[n + j] = accA1u[r];
                }
            }
        }
        {
            const int v = 16 * w + j;        // own column
#pragma unroll
            for (int c = 0; c < (ROLES ? 0 : NWV); ++c) {
                const int ub = 16 * ((w + c) & (NWV - 1)) + 4 * g;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (ub + r < HR && v < HR) {
                        wa[aW2 + (size_t)(ub + r) * HR + v] = accA2[c][r];
                        wa[aW3 + (size_t)(ub + r) * HR + v] = accA3[c][r];
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < (RSM ? 0 : 4); ++r) if (v < HR) wa[aP3 + (size_t)(4 * r + g) * HR + v] = accAP3[r];      // rows (g, r) <-> slot 4r+g
        }
        f4 s1 = SA1, s2 = SA2, s3 = SA3, sg = SGi;
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s1[r] += __shfl_xor(s1[r], m, 64); s2[r] += __shfl_xor(s2[r], m, 64); s3[r] += __shfl_xor(s3[r], m, 64);
                sg[r] += __shfl_xor(sg[r], m, 64);
            }
        }
        if (j == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int u = 16 * w + 4 * g + r;
                if (u < HR) { wa[aB1 + u] = s1[r]; wa[aB2 + u] = s2[r]; wa[aB3 + u] = s3[r]; }
                if (w == 0) wa[aSG + 4 * r + g] = sg[r];         // (the slot adjoints are the same in every wave)
            }
        }
    }
}

size_t k7f_lds_bytes(int nw, bool roles = false) { return (wide_t_floats(nw) * (aet_in_lds(nw) ? 2 : 1) + ((size_t)(roles ? 12 : 3) * nw + (roles ? 2 : 0)) * FTILE) * sizeof(float); }
size_t k7f_ring_floats(int nw, int method, long long B) { return nw >= 8 ? (size_t)rk_stages(method) * 3 * (size_t)B * 16 * nw : 0; }
int k7f_np(int hr, int xd, int ne) { const int n = xd + ne; return hr * 3 * n + hr + 2 * (hr * hr + hr) + xd * hr + xd; }
// <= 4 waves: the AE head's partials [dAW1 | db1 | dAW2 | db2 | dAW3 | db3 | P3 (16 x h) | sum gi (16)]
int k7f_npa(int nw, bool saved, int hr, int xd, int nzv, int ne) {
    return (nw <= 4 && saved) ? hr * (xd + ne + xd + nzv) + hr + 2 * (hr * hr + hr) + 16 * hr + 16 : 0;
}
size_t k7f_fwd_floats(int nw, int n) { return ((wide_fwd_floats(nw, n) + 63) / 64) * 64; }

template <int METHOD, int NWV>
hipError_t launch_k7f(const FusedDaeDev& a, int NZM, int NZA, const float* pde, const float* pae, const f4* pt, const f4* pf, const f4* pta,
                      const f4* pfa, int NA, hipStream_t s) {
    constexpr bool RL = PSNODE_K7F_ROLES && PSNODE_K7F_AE_GRADS_DEFAULT && NWV <= 4;
    const bool roles = RL && a.sact != nullptr;
    const dim3 grid((unsigned)((a.B + TBM - 1) / TBM)), block(64 * NWV * (roles ? 2 : 1));
    const size_t lds = k7f_lds_bytes(NWV, roles);
#define PSNODE_K7F(NZM_, NZA_)                                                                                                  \
    {                                                                                                                           \
        auto kern = a.sact ? &dae_backward_fused_kernel<METHOD, NZM_, NZA_, NWV, false, RL> : &dae_backward_fused_kernel<METHOD, NZM_, NZA_, NWV, true>; \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return e;                                                                                          \
        hipLaunchKernelGGL(kern, grid, block, lds, s, a, pde, pae, pt, pf, pta, pfa, NA);                                       \
        return hipGetLastError();                                                                                               \
    }
    switch (NZM * 10 + NZA) {
        case 11: PSNODE_K7F(1, 1)
        case 21: PSNODE_K7F(2, 1)
        case 31: PSNODE_K7F(3, 1)
        case 41: PSNODE_K7F(4, 1)
        case 32: PSNODE_K7F(3, 2)
        case 42: PSNODE_K7F(4, 2)
        default: return hipErrorNotSupported;
    }
#undef PSNODE_K7F
}

template <int NWV>
hipError_t launch_k7f_method(const FusedDaeDev& a, int NZM, int NZA, const float* pde, const float* pae, const f4* pt, const f4* pf,
                             const f4* pta, const f4* pfa, int NA, hipStream_t s) {
    switch (a.method) {
        case PSNODE_EULER: return launch_k7f<PSNODE_EULER, NWV>(a, NZM, NZA, pde, pae, pt, pf, pta, pfa, NA, s);
        case PSNODE_MIDPOINT: return launch_k7f<PSNODE_MIDPOINT, NWV>(a, NZM, NZA, pde, pae, pt, pf, pta, pfa, NA, s);
        default: return launch_k7f<PSNODE_RK4_38, NWV>(a, NZM, NZA, pde, pae, pt, pf, pta, pfa, NA, s);
    }
}

}  // namespace

// ---- entry points used by psnode_dae_backward_wide.hip (C ABI psnode_dae_backward_wide_f32 with grad_params_de set)
size_t dae_fused_bwd_workspace_floats(const psnode_dae_bwd_wide_args_f32* p) {
    const int nw = wide_hidden(p->de) / 16, ne = p->z_dim + p->v_dim + p->i_dim, n = p->x_dim + ne;
    const size_t nwg = (size_t)((p->B + TBM - 1) / TBM);
    return 2 * k7f_fwd_floats(nw, n) + 4 * wide_t_floats(nw) + ((nwg * k7f_np(p->de.out_dim[0], p->x_dim, ne) + 63) / 64) * 64 +
           ((nwg * k7f_npa(nw, p->saved_act != nullptr, p->de.out_dim[0], p->x_dim, p->z_dim + p->v_dim, ne) + 63) / 64) * 64 + k7f_ring_floats(nw, p->method, p->B) + 256;
}
// floats of psnode_dae_bwd_wide_args_f32::grad_params_ae_raw (0: this width leaves the AE head's contractions to the caller, K7h)
size_t dae_fused_bwd_ae_floats(const psnode_dae_bwd_wide_args_f32* p) {
    const int nw = wide_hidden(p->de) / 16;
    return (size_t)k7f_npa(nw, p->saved_act != nullptr, p->de.out_dim[0], p->x_dim, p->z_dim + p->v_dim, p->z_dim + p->v_dim + p->i_dim);
}

int dae_fused_bwd_launch(const psnode_dae_bwd_wide_args_f32* p, float* workspace, hipStream_t s) {
    const int H = wide_hidden(p->de), nw = H / 16, xd = p->x_dim, zd = p->z_dim, vd = p->v_dim, id = p->i_dim, HR = p->de.out_dim[0];
    const int nzv = zd + vd, ne = nzv + id, n = xd + ne;
    const int NZM = (2 * ne + 3) / 4, NZA = (nzv + 3) / 4, NA = (n + 3) / 4;
    float* pde = workspace;
    float* pae = pde + k7f_fwd_floats(nw, n);
    f4* pt = reinterpret_cast<f4*>(pae + k7f_fwd_floats(nw, n));
    f4* pf = pt + wide_t_floats(nw) / 4;
    f4* pta = pf + wide_t_floats(nw) / 4;
    f4* pfa = pta + wide_t_floats(nw) / 4;
    float* wpart = reinterpret_cast<float*>(pfa + wide_t_floats(nw) / 4);
    const size_t nwg = (size_t)((p->B + TBM - 1) / TBM);
    const int NP = k7f_np(HR, xd, ne), NPA = k7f_npa(nw, p->saved_act != nullptr, HR, xd, nzv, ne);
    float* wpart_ae = wpart + ((nwg * NP + 63) / 64) * 64;
    float* ring = wpart_ae + ((nwg * NPA + 63) / 64) * 64;
    if (NPA > 0 && !p->grad_params_ae_raw) return PSNODE_ERR_NULL;
    PackMfma f;
    memset(&f, 0, sizeof(f));
    f.ae = 0; f.nw = nw; f.xd = xd; f.ne = ne; f.n = n; f.nzv = nzv; f.NX = kNXc; f.NB = 0; f.NE = NZM; f.NA = NA; f.fold = 1;
    f.hreal = HR;
    f.w1 = p->de.weight[0]; f.b1 = p->de.bias[0]; f.w2 = p->de.weight[1]; f.b2 = p->de.bias[1];
    f.w3 = p->de.weight[2]; f.b3 = p->de.bias[2]; f.w4 = p->de.weight[3]; f.b4 = p->de.bias[3];
    f.out_dim = xd; f.out = pde;
    hipLaunchKernelGGL(pack_wide_fwd_kernel, dim3(32), dim3(256), 0, s, f);
    PackMfma q = f;
    q.ae = 1; q.NE = NZA; q.fold = 0;
    q.w1 = p->ae.weight[0]; q.b1 = p->ae.bias[0]; q.w2 = p->ae.weight[1]; q.b2 = p->ae.bias[1];
    q.w3 = p->ae.weight[2]; q.b3 = p->ae.bias[2]; q.w4 = p->ae.weight[3]; q.b4 = p->ae.bias[3];
    q.out_dim = id; q.out = pae;
    hipLaunchKernelGGL(pack_wide_fwd_kernel, dim3(32), dim3(256), 0, s, q);
    PackWideT t{nw, HR, p->de.weight[1], p->de.weight[2], pt};
    hipLaunchKernelGGL(pack_wide_t_kernel, dim3(64), dim3(256), 0, s, t);
    PackWideT ta{nw, HR, p->ae.weight[1], p->ae.weight[2], pta};
    hipLaunchKernelGGL(pack_wide_t_kernel, dim3(64), dim3(256), 0, s, ta);
    if (nw >= 8) {
        PackWideT tf{nw, HR, p->de.weight[1], p->de.weight[2], pf};
        hipLaunchKernelGGL(pack_wide_f_kernel, dim3(64), dim3(256), 0, s, tf);
        PackWideT tfa{nw, HR, p->ae.weight[1], p->ae.weight[2], pfa};
        hipLaunchKernelGGL(pack_wide_f_kernel, dim3(64), dim3(256), 0, s, tfa);
    }
    if (hipGetLastError() != hipSuccess) return PSNODE_ERR_HIP;
    FusedDaeDev a;
    memset(&a, 0, sizeof(a));
    a.method = p->method; a.xd = xd; a.zd = zd; a.vd = vd; a.id = id; a.hreal = HR; a.n_events = p->n_events; a.NP = NP; a.T = p->T; a.B = p->B;
    a.w1 = p->de.weight[0]; a.w4 = p->de.weight[3]; a.aw1 = p->ae.weight[0]; a.aw4 = p->ae.weight[3];
    a.t = ViewDev{p->t.ptr, p->t.stride_t, p->t.stride_b};
    a.z = ViewDev{p->z.ptr, p->z.stride_t, p->z.stride_b};
    a.v = ViewDev{p->v.ptr, p->v.stride_t, p->v.stride_b};
    a.a0 = p->all_initial; a.ev = p->event_idx;
    a.zj = p->z_jump; a.zjb = p->zj_stride_b; a.zje = p->zj_stride_e;
    a.vj = p->v_jump; a.vjb = p->vj_stride_b; a.vje = p->vj_stride_e;
    a.xs = p->xs; a.is = p->is; a.gxs = p->grad_xs; a.gis = p->grad_is;
    a.tx = (p->flags & PSNODE_FLAG_INPUT_TRUE_X) ? 1 : 0; a.ti = (p->flags & PSNODE_FLAG_INPUT_TRUE_I) ? 1 : 0;
    a.xtrue = p->x_true; a.itrue = p->i_true;
    if ((a.tx || a.ti) && p->saved_act) return PSNODE_ERR_UNSUPPORTED;      // a teacher-forced forward saves nothing: recompute form only
    if ((a.tx && !a.xtrue) || (a.ti && !a.itrue)) return PSNODE_ERR_NULL;
    a.carry_x = p->carry_x;
    a.gzv = p->grad_zv; a.gjump = p->grad_jump; a.ga0 = p->grad_all_initial_de;
    a.wpart = wpart; a.ring = ring; a.wpart_ae = wpart_ae; a.NPA = NPA;
    for (int l = 0; l < 3; ++l) {
        a.aact[l] = p->ae_act[l]; a.adelta[l] = p->ae_delta[l];
        a.eact[l] = p->ev_act[l]; a.edelta[l] = p->ev_delta[l];
    }
    a.agi = p->ae_gi; a.egi = p->ev_gi; a.ei = p->ev_i;
    a.sact = p->saved_act; a.sxst = p->saved_xstage; a.saeact = p->saved_ae_act; a.sevact = p->saved_ev_act; a.sevi = p->saved_ev_i;
    hipError_t e;
    switch (nw) {
        case 2: e = launch_k7f_method<2>(a, NZM, NZA, pde, pae, pt, pf, pta, pfa, NA, s); break;
        case 4: e = launch_k7f_method<4>(a, NZM, NZA, pde, pae, pt, pf, pta, pfa, NA, s); break;
        default: e = launch_k7f_method<8>(a, NZM, NZA, pde, pae, pt, pf, pta, pfa, NA, s); break;
    }
    if (e == hipErrorNotSupported) return PSNODE_ERR_UNSUPPORTED;
    if (e != hipSuccess) return PSNODE_ERR_HIP;
    if (launch_reduce_partials(wpart, p->grad_params_de, nullptr, NP, 0, (int)nwg, s) != hipSuccess) return PSNODE_ERR_HIP;
    if (NPA > 0 && launch_reduce_partials(wpart_ae, p->grad_params_ae_raw, nullptr, NPA, 0, (int)nwg, s) != hipSuccess) return PSNODE_ERR_HIP;
    return PSNODE_OK;
}

}  // namespace psnode
