// K4f -- backward pass through the ODE integrator for hidden widths 32 / 64 / 128 in ONE launch, parameter gradients included
// (neural_00_ODE_01_no_encode.py:358-360 -- loss.backward() through the unrolled loop of my_solvers.py:66-78 -- at the scripts'
// argparse default --hidden 128, :245-246).  Replaces the round-2 split (K4w adjoint sweep + library GEMMs over 12 KB of stored rows per
// state-step, psnode_backward_wide.hip + fused.ode_backward_wide): nothing but the inputs and the gradients touches HBM.
//
// Tile / waves / layer plan as K1 and K4w: one workgroup = NWV = hidden/16 waves = 16 trajectories, walked from the last step to the
// first; phase A recomputes the stage evaluations with the folded forward image, phase B sweeps the stages backwards with the
// transposed images of W2 / W3 in LDS.  What is new is where the weight gradients come from:
//   dW_l[u][v] = sum over (step, stage, trajectory) of delta_l[u] * h_{l-1}[v]
// contracts over the tile's 16 trajectories on MFMA (A = delta^T, B = h^T).  delta_l of EVERY wave is in the exchange buffer anyway
// (the all-gather in front of W_l^T delta_l); the buffer tiles are padded so that they can also be read TRANSPOSED (4 ds_read_b32,
// two-pass conflict-free) -- so the only extra data movement is one in-wave transpose of the wave's own h_{l-1}.  Wave w accumulates
// the COLUMNS of its own 16 units, dW_l[all u][16w..16w+15], in 4*NWV registers per layer for the whole launch; per-workgroup partials
// are summed by a second kernel in a fixed order (deterministic, as K4).
// Capacity at hidden 128 (8 waves, 256 registers per lane, 160 KB LDS): W2, W3, their transposes and the dW2 / dW3 accumulators are 3 x 128 KB.
// The accumulators take 64 registers per lane for the whole launch (+ 20 for the small gradients); the 128 KB weight region of the LDS
// holds the FORWARD images during phase A and the TRANSPOSED images during phase B, swapped twice per step by LDS-DMA
// (global_load_lds_dwordx4 from the packed images, L2-resident; every lane only ever reads back slots its own wave filled, so the swap
// needs no barrier: a layer's region is refilled right behind its last use of the phase and waited for with vmcnt(0) before its first
// use of the next one).  Holding the forward images in registers instead (round 2's K4w: re-read per step) made the compiler spill the
// accumulators around phase A -- 784 B of scratch per lane, 200 scratch instructions per step.  The stage activations travel through a
// per-workgroup ring in the workspace that is rewritten every step (24 KB per workgroup: it lives in L2, never in HBM).  Exchange tiles
// 18 KB + transpose tiles 9 KB of LDS.
#define PSNODE_ELU_LITERALS
#include <stdlib.h>
#include <type_traits>
#include <string.h>

#include "psnode_wide_pack.h"

namespace psnode {
namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f4 fm4(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

struct FusedDev {
    int method, xd, zd, hreal, n_events, NP;
    int true_x;                           // teacher-forced call: `xs` = the dataset rows, no adjoint carried from step to step (my_solvers.py:72-74)
    long long T, B;
    const float *w1, *w4;                 // raw nn.Linear tensors for the small transposed operands
    ViewDev t, z;
    const float* a0;
    const int* ev;
    const float* zj;
    long long zjb, zje;
    const float *xs, *gout;
    float *gx0, *gz, *gzj, *ga0;
    float* wpart;                         // [workgroups][NP]
    float* ring;                          // NWV >= 8: [S][3][B][H] stage activations of the step in flight
    const float *sact, *sxst;             // saved by the forward call ([T-1,S,3,B,H] / [T-1,S,B,xd]) or null: recompute
};

#ifndef PSNODE_K4F_BOUND
#define PSNODE_K4F_BOUND 3      // 0: never bound the scheduler's read-ahead in the layer loops, 1: always at 8 waves, 2: only at RK4, 3: RK4 + Midpoint
                                // (profiles/r03m_bwd_ab.txt, hidden-128 training step rk4 / midpoint / euler: 0 -> 65.1 / 28.0 / 11.95 ms, 1 -> 52.8 / 22.9 / 12.19)
#endif
#ifndef PSNODE_K4F_BOUND_SAVED
#define PSNODE_K4F_BOUND_SAVED 3      // the same knob for the instances that read saved activations (REC = false)
#endif
#ifndef PSNODE_K4F_EVERY_SAVED
#define PSNODE_K4F_EVERY_SAVED 2
#endif
#ifndef PSNODE_K4F_EVERY
#define PSNODE_K4F_EVERY 2      // a sched_barrier behind every EVERY-th chunk
#endif
constexpr int FTILE = 64 * 4 + 4 * 8;     // padded 16x16 tile (floats): lane l's four rows 4g..4g+3 of column j at 4l + 8g

#ifndef PSNODE_K4F_ROLES
#define PSNODE_K4F_ROLES 1      // <= 4 waves, saved activations: a second set of NWV waves per tile owns the H->H weight gradients (below)
#endif
#ifndef PSNODE_K4F_SAVED_AHEAD
#define PSNODE_K4F_SAVED_AHEAD 2      // see the comment at its use in the stage loop
#endif
#define PSNODE_K4F_SAVED_AHEAD_DEFAULT PSNODE_K4F_SAVED_AHEAD
#ifndef PSNODE_K4F_ROLES_SMALL
#define PSNODE_K4F_ROLES_SMALL 1  // the gradient waves also own dW4 and the s columns of dW1 (gk / s / delta1 handed over in LDS tiles): no in-wave
                                  // transpose is left on the chain
#endif
#ifndef PSNODE_K4F_ROLES_PRIO
#define PSNODE_K4F_ROLES_PRIO 1 // the chain waves run at a higher issue priority than the gradient waves
#endif

// ROLES (round 4): the two-role form of the saved-activation instances at <= 4 waves per tile.  One wave per SIMD cannot hide an LDS
// exchange or an MFMA result latency, and half of the backward's MFMAs -- the weight gradients dW2 / dW3 -- are not on the adjoint's
// critical path at all.  The workgroup gets NWV more waves (one more per SIMD): waves 0..NWV-1 are the CHAIN (the sweep, as before, minus
// the H->H weight gradients, their accumulators and the two in-wave transposes per stage), waves NWV..2NWV-1 are GRADIENT waves: each
// loads its own 16 units of the saved h1 / h2 rows straight from the forward's save area (it needs nothing from the chain for that),
// transposes them in its private tile, and contracts them with the delta tiles the chain's all-gathers publish anyway -- read TRANSPOSED
// between the barrier that publishes them and the next one (the chain rewrites a parity two exchanges later, so one barrier sequence
// shared by both roles is all the synchronisation there is).  The gradient waves' MFMAs fill the issue slots the chain leaves empty.
// LDS tiles of the two-role form (FTILE floats each, behind the exchange parities): chain-private NWV | gradient-private NWV | delta1 NWV | gk | s
template <int NWV> __device__ __forceinline__ float* roles_d1_tile(float* xb, const int wv) { return xb + (4 * NWV + wv) * FTILE; }
template <int NWV> __device__ __forceinline__ float* roles_gk_tile(float* xb) { return xb + 5 * NWV * FTILE; }
template <int NWV> __device__ __forceinline__ float* roles_s_tile(float* xb) { return xb + (5 * NWV + 1) * FTILE; }

template <int METHOD, int NZM, int NWV>
__device__ __forceinline__ void fused_gradient_wave(const FusedDev& a, float* __restrict__ xb, const int l, const int wg) {
    constexpr int S = rk_stages(METHOD), H = 16 * NWV, NX = kNXc;
    const int g = l >> 4, j = l & 15, i = j, HR = a.hreal, xd = a.xd, zd = a.zd, n = xd + zd;
    float* scr = xb + (3 * NWV + wg) * FTILE;                          // private transpose tile (behind the chain waves')
    const long long b0 = (long long)blockIdx.x * TBM;
    const long long b = b0 + j < a.B ? b0 + j : a.B - 1;               // padding trajectories: their deltas are zero
    const int toff = 4 * l + 8 * g, roff = 72 * (j >> 2) + 4 * g + (j & 3);
    // row of the s tile that holds column i of the stage input (as the chain's `srow`)
    const int srow = i < xd ? 4 * (i & 3) + (i >> 2) : (i < n ? 4 * ((i - xd) & 3) + 2 + ((i - xd) >> 2) : -1);
    const int sroff = srow >= 0 ? 72 * (srow >> 2) + 4 * g + (srow & 3) : 0;
    auto tile = [&](const int par, const int wv) -> const float* { return xb + (par * NWV + wv) * FTILE; };
    auto get_at = [&](const float* t_, const int ro) -> f4 { const float* s_ = t_ + ro; return f4{s_[0], s_[16], s_[32], s_[48]}; };
    auto get_row = [&](const float* t_) -> f4 { return get_at(t_, roff); };
    auto transpose = [&](const f4 v) -> f4 { *reinterpret_cast<f4*>(scr + toff) = v; return get_row(scr); };
    const unsigned offH = 4u * ((unsigned)(b * H) + 16 * wg + 4 * g);
    const long long act_layer = a.B * H, nT = a.T;
    auto load_saved = [&](const long long idx, f4& q1, f4& q2, f4& q3) {
        const float* rb = a.sact + (size_t)idx * 3 * act_layer;
        q1 = ldg<f4>(sbase(rb), offH);
        q2 = ldg<f4>(sbase(rb + act_layer), offH);
        if constexpr (PSNODE_K4F_ROLES_SMALL) q3 = ldg<f4>(sbase(rb + 2 * act_layer), offH);
    };
    const f4 zero4 = f4{0.f, 0.f, 0.f, 0.f};
    f4 accW2[NWV], accW3[NWV], accW4 = zero4, accW1s = zero4;
#pragma unroll
    for (int c = 0; c < NWV; ++c) { accW2[c] = zero4; accW3[c] = zero4; }
    f4 sv1 = zero4, sv2 = zero4, sv3 = zero4;
    if (nT >= 2) load_saved((nT - 1) * S - 1, sv1, sv2, sv3);
    int p = 0;
#ifndef PSNODE_K4F_ROLES_EARLY
#define PSNODE_K4F_ROLES_EARLY 1       // chunks (4 MFMAs each) of a contraction issued in the interval its tiles are published in; the rest waits for the
                                       // interval behind the stage's all-reduce, where the chain is latency-bound and leaves the MFMA pipe idle
#endif
    constexpr int E = PSNODE_K4F_ROLES_EARLY < NWV ? PSNODE_K4F_ROLES_EARLY : NWV;
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
    for (long long k = nT - 2; k >= 0; --k) {
#pragma unroll
        for (int s = S - 1; s >= 0; --s) {
            const long long idx = k * S + s;
            const f4 h3T = PSNODE_K4F_ROLES_SMALL ? transpose(sv3) : zero4;
            const f4 h2T = transpose(sv2), h1T = transpose(sv1);
            load_saved(idx > 0 ? idx - 1 : 0, sv1, sv2, sv3);     // the next stage's rows: a whole stage ahead (nothing here waits on them)
            f4 dT3[NWV], dT2[NWV], sT = zero4;
            lds_barrier();                                         // delta3 of every wave is in parity p; gk and s tiles of the stage
            read_tiles(p, dT3);
            if constexpr (PSNODE_K4F_ROLES_SMALL) {
                const f4 gT = get_row(roles_gk_tile<NWV>(xb));
                sT = get_at(roles_s_tile<NWV>(xb), sroff);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) accW4 = fm4(gT[kk], h3T[kk], accW4);
            }
            chunks(dT3, h2T, accW3, 0, E);
            __builtin_amdgcn_sched_barrier(0);
            p ^= 1;
            lds_barrier();                                         // delta2
            read_tiles(p, dT2);
            chunks(dT3, h2T, accW3, E, 2 * E);
            chunks(dT2, h1T, accW2, 0, E);
            __builtin_amdgcn_sched_barrier(0);
            p ^= 1;
            lds_barrier();                                         // the stage's all-reduce; delta1 of the chain wave with these units
            if constexpr (PSNODE_K4F_ROLES_SMALL) {
                const f4 dT = get_row(roles_d1_tile<NWV>(xb, wg));
                if (srow < 0) sT = zero4;
                chunks(dT3, h2T, accW3, 2 * E, NWV);
                chunks(dT2, h1T, accW2, E, NWV);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) accW1s = fm4(dT[kk], sT[kk], accW1s);
            } else {
                chunks(dT3, h2T, accW3, 2 * E, NWV);
                chunks(dT2, h1T, accW2, E, NWV);
            }
            p ^= 1;
        }
    }                                                              // (the step's dL/dz rides on the last stage's all-reduce)
    lds_barrier();                                                 // epilogue: dL/dall_initial all-reduce; sum(delta1) in the chain wave's private tile
    const int K1 = 3 * n;
    float* wp = a.wpart + (size_t)blockIdx.x * a.NP;
    const int oW2 = HR * K1 + HR, oW3 = oW2 + HR * HR + HR, oW4 = oW3 + HR * HR + HR;
    const int v = 16 * wg + j;
#pragma unroll
    for (int c = 0; c < NWV; ++c) {
        const int ub = 16 * ((wg + c) & (NWV - 1)) + 4 * g;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (ub + r < HR && v < HR) {
                wp[oW2 + (size_t)(ub + r) * HR + v] = accW2[c][r];
                wp[oW3 + (size_t)(ub + r) * HR + v] = accW3[c][r];
            }
        }
    }
    if constexpr (PSNODE_K4F_ROLES_SMALL) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {   // dW4 rows (g, r) <-> x-dim 4r+g, columns = own units
            const int dd = 4 * r + g;
            if (r < NX && dd < xd && v < HR) wp[oW4 + (size_t)dd * HR + v] = accW4[r];
        }
        // dW1: columns [a0 | s-a0 | s]; ca0 = sum(delta1) (x) a0 over the tile's trajectories
        const f4 s1T = get_row(xb + (2 * NWV + wg) * FTILE);
        f4 ca0 = zero4;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const long long tb = b0 + 4 * kk + g;
            const float av = (i < n && tb < a.B) ? a.a0[tb * n + i] : 0.0f;
            ca0 = fm4(s1T[kk], av, ca0);
        }
        if (j < n) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int u = 16 * wg + 4 * g + r;
                if (u < HR) {
                    float* row = wp + (size_t)u * K1;
                    row[j] = ca0[r];
                    row[n + j] = accW1s[r] - ca0[r];
                    row[2 * n + j] = accW1s[r];
                }
            }
        }
    }
}

// REC = false: the forward call saved the stage activations and stage inputs (psnode_ode_args_f32::save_act / save_xstage): no phase A,
// the transposed images stay in LDS for the whole launch, the rows of (step, stage) are requested one stage ahead along the sweep.
// ROLES: 2 NWV waves per tile, see fused_gradient_wave.
template <int METHOD, int NZM, int NWV, bool REC = true, bool ROLES = false>
__global__ __launch_bounds__(64 * NWV * (ROLES ? 2 : 1)) void ode_backward_fused_kernel(const FusedDev a, const float* __restrict__ pack_de,
                                                                        const f4* __restrict__ pack_t, const f4* __restrict__ pack_f,
                                                                        const int NA) {
    constexpr int NX = kNXc, S = rk_stages(METHOD), H = 16 * NWV;
    constexpr int NZ = NZM > 0 ? NZM : 1;
    using RD = Regs<NX, 0, NZM, NWV>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    f4* wT = reinterpret_cast<f4*>(lds);                      // [layer 0: W2^T | 1: W3^T][chunk][wave][lane]
    float* xb = lds + (size_t)2 * NWV * NWV * 64 * 4;         // [2][NWV] exchange tiles (padded)

    static_assert(!ROLES || (!REC && NWV <= 4), "two-role form: saved activations, <= 4 waves per tile");
    const int l = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if constexpr (ROLES) {
        if (w >= NWV) {
            fused_gradient_wave<METHOD, NZM, NWV>(a, xb, l, w - NWV);
            return;
        }
        if constexpr (PSNODE_K4F_ROLES_PRIO) __builtin_amdgcn_s_setprio(2);
    }
    const int g = l >> 4, j = l & 15, i = j;
    float* scr = xb + 2 * NWV * FTILE + w * FTILE;           // this wave's private transpose tile
    const long long b0 = (long long)blockIdx.x * TBM;
    const bool valid = b0 + j < a.B;
    const long long b = valid ? b0 + j : a.B - 1;
    const int xd = a.xd, zd = a.zd, ne = zd, n = xd + zd, HR = a.hreal;

    // ---- forward image -> registers (as K1), transposed images -> LDS
    const float* pw = pack_de + (size_t)w * (RD::COUNT + NA) * 64 + l;
    constexpr bool STREAM = NWV >= 8 && REC;     // forward / transposed images swapped in LDS per phase; activations through the ring
    constexpr int BMODE = REC ? PSNODE_K4F_BOUND : PSNODE_K4F_BOUND_SAVED;
    // (rounds 3's exception for the recompute instance <Midpoint, NZM = 0, 8 waves> is gone: its wrong dL/dall_initial / dW1 were a VGPR
    //  spill -- of `l & 15`, live from the prologue to the epilogue -- that the register allocator had placed inside the `4 + g < x_dim`
    //  arm of load_x2, a divergent region whose EXEC is EMPTY at x_dim = 8: nothing was saved, and the epilogue's `i < n` predicates read
    //  whatever the scratch slot held (zeros: every lane acted as column 0).  Dropping the sched_barriers only moved the spill.  load_x2
    //  is branch-free now (psnode_common.h: ldg_sel); profiles/scripts/isa_lint.py check A watches for the pattern.  DESIGN.md, round 4.)
    constexpr bool BOUND = NWV >= 8 && (BMODE == 1 || (BMODE == 2 && S >= 4) || (BMODE == 3 && S >= 2));
    constexpr int EVERY = REC ? PSNODE_K4F_EVERY : PSNODE_K4F_EVERY_SAVED;
    float w1xs[NX], w1z[NZ], w2r[STREAM ? 1 : 4 * NWV], w3r[STREAM ? 1 : 4 * NWV], w4[4];
    f4 b1r, b2, b3, b4;
#pragma unroll
    for (int r = 0; r < NX; ++r) w1xs[r] = pw[(RD::W1A + r) * 64];
#pragma unroll
    for (int m = 0; m < NZM; ++m) w1z[m] = pw[(RD::W1E + m) * 64];
#pragma unroll
    for (int k = 0; k < (STREAM ? 0 : 4 * NWV); ++k) { w2r[k] = pw[(RD::W2 + k) * 64]; w3r[k] = pw[(RD::W3 + k) * 64]; }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        w4[r] = pw[(RD::W4 + r) * 64];
        b1r[r] = pw[(RD::B1 + r) * 64]; b2[r] = pw[(RD::B2 + r) * 64]; b3[r] = pw[(RD::B3 + r) * 64]; b4[r] = pw[(RD::B4 + r) * 64];
    }
    // LDS-DMA of one layer's image (0: W2, 1: W3) into its region: slot (layer, c, w) <- 64 lanes x 16 B, lane-linear.  The statement
    // first drains this wave's LDS reads (they are the only readers of these slots) and is invisible to the compiler's wait counts:
    // dma_wait() before the first use.
    const unsigned wT_lds = __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<uintptr_t>(wT));
    const unsigned lane16 = 16u * (unsigned)l;
    auto dma_layer = [&](const f4* __restrict__ img, const int layer) {
#pragma unroll
        for (int c = 0; c < NWV; ++c) {
            const int slot = (layer * NWV + c) * NWV + w;
            // source = <uniform slot base in SGPRs> + <lane * 16 bytes>: one VGPR for all 32 slots (a per-lane 64-bit pointer per slot is
            // loop-invariant, and the compiler kept -- and spilled -- all of them)
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
    } else if constexpr (NWV >= 8) {      // saved activations: the transposed images, once
        dma_layer(pack_t, 0);
        dma_layer(pack_t, 1);
        dma_wait();
    } else {
#pragma unroll
        for (int c = 0; c < NWV; ++c) {
            wT[((0 * NWV + c) * NWV + w) * 64 + l] = pack_t[((0 * NWV + c) * NWV + w) * 64 + l];
            wT[((1 * NWV + c) * NWV + w) * 64 + l] = pack_t[((1 * NWV + c) * NWV + w) * 64 + l];
        }
    }
    // small transposed operands straight from the nn.Linear tensors (u = 16w + i: this lane's A-operand row)
    //   w4T[r] = W4[4r+g][u]                                   g3[u] = sum_d W4[d][u] gk[d]
    //   fT[r]  = (Ws+Wd)[16w+4g+r][col(i)]: output row i = 4g'+r' carries x-dim 4r'+g' (r' < 2), z-dim g' (r' = 2), z-dim 4+g' (r' = 3)
    float w4T[NX], fT[4];
    {
        const int u = 16 * w + i, K1 = 3 * n;
#pragma unroll
        for (int r = 0; r < NX; ++r) { const int d = 4 * r + g; w4T[r] = (d < xd && u < HR) ? a.w4[(size_t)d * HR + u] : 0.0f; }
        const int rr = i & 3, gr = i >> 2;
        const int col = rr < 2 ? (4 * rr + gr < xd ? 4 * rr + gr : -1) : (4 * (rr - 2) + gr < zd ? xd + 4 * (rr - 2) + gr : -1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int uu = 16 * w + 4 * g + r;
            fT[r] = (col >= 0 && uu < HR) ? a.w1[(size_t)uu * K1 + 2 * n + col] + a.w1[(size_t)uu * K1 + n + col] : 0.0f;
        }
    }

    // ---- per-trajectory constants (as K1)
    float a0e[NZ];
    int ecol[NZ];
    bool eon[NZ];
#pragma unroll
    for (int m = 0; m < NZM; ++m) {
        const int q = 4 * m + g, e = slot_ext(q, ne);
        eon[m] = e >= 0;
        ecol[m] = e >= 0 ? e : 0;
        a0e[m] = q < ne ? a.a0[b * n + xd + q] : 0.0f;
    }
    f4 c0 = b1r;
    for (int m = 0; m < NA; ++m) {
        const int q = 4 * m + g;
        c0 = fm4(pw[(RD::COUNT + m) * 64], q < n ? a.a0[b * n + q] : 0.0f, c0);
    }
    // row of a padded tile that holds column i of the stage input s = (x dims | z dims):  x-dim d sits in row 4(d&3) + (d>>2) (registers
    // 0..1 of lane group d&3), z-dim e in row 4(e&3) + 2 + (e>>2) (registers 2..3 of lane group e&3)
    const int srow = i < xd ? 4 * (i & 3) + (i >> 2) : (i < n ? 4 * ((i - xd) & 3) + 2 + ((i - xd) >> 2) : -1);

    // ---- LDS tiles
    const int toff = 4 * l + 8 * g;                                   // own slot of a tile
    const int roff = 72 * (i >> 2) + 4 * g + (i & 3);                 // row i of a tile, columns g, g+4, g+8, g+12
    auto tile = [&](const int par, const int wv) -> float* { return xb + (par * NWV + wv) * FTILE; };
    auto put = [&](float* t_, const f4 v) { *reinterpret_cast<f4*>(t_ + toff) = v; };
    auto getl = [&](const float* t_) -> f4 { return *reinterpret_cast<const f4*>(t_ + toff); };
    auto get_row = [&](const float* t_, const int ro) -> f4 { const float* s_ = t_ + ro; return f4{s_[0], s_[16], s_[32], s_[48]}; };
    auto transpose = [&](const f4 v) -> f4 {
        put(scr, v); return get_row(scr, roff);
    };   // own D tile -> operand layout (trajectory g + 4kk)

    int p = 0;
    constexpr bool PREFETCH_ALL = NWV <= 4;
    // forward H->H layer with the weights in registers (K1's `mid`), returns the pre-activation
    auto mid = [&](const float (&wm)[4 * NWV], const f4 bias, const f4 h) -> f4 {
        put(tile(p, w), h);
        f4 accA = bias, accB = f4{0.f, 0.f, 0.f, 0.f};
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
    // the same forward layer with the image of `layer` in LDS (8 waves)
    auto mid_lds = [&](const int layer, const f4 bias, const f4 h) -> f4 {
        put(tile(p, w), h);
        const f4* wl = wT + ((size_t)layer * NWV * NWV + w) * 64 + l;
        f4 wq = wl[0];
        f4 accA = bias, accB = f4{0.f, 0.f, 0.f, 0.f};
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
            if constexpr (BOUND) { if (c % EVERY == EVERY - 1) __builtin_amdgcn_sched_barrier(0); }   // 8 waves: bound the scheduler's read-ahead (registers)
        }
        p ^= 1;
        return accA + accB;
    };
    // transposed H->H layer (image in LDS): returns sum_k W[k][own] d[k]; and, from the very tiles the all-gather published, the weight
    // gradient of that layer: acc[c] += delta(block (w+c) % NWV)^T (x) hT, hT = this wave's own input activations in operand layout
#ifndef PSNODE_K4F_DEFER_DW
#define PSNODE_K4F_DEFER_DW 1        // <= 4 waves: the weight-gradient MFMAs of a layer are issued behind the NEXT exchange's LDS write, in front of its
                                     // barrier -- one wave per SIMD has nothing else to keep the MFMA pipe busy while the tile travels
#endif
#ifndef PSNODE_K4F_TREAD_AHEAD
#define PSNODE_K4F_TREAD_AHEAD 1
#endif
    constexpr bool RSM = ROLES && PSNODE_K4F_ROLES_SMALL;
    constexpr bool DEFER = PREFETCH_ALL && PSNODE_K4F_TREAD_AHEAD && PSNODE_K4F_DEFER_DW && !ROLES;
    f4 pendT[DEFER ? NWV : 1], pend_h = f4{0.f, 0.f, 0.f, 0.f};      // transposed tiles / own activations of the layer whose gradient is still owed
#ifndef PSNODE_K4F_DEFER8
#define PSNODE_K4F_DEFER8 1         // 8 waves: the same deferral without holding the transposed tiles: they are re-read from the previous exchange's parity
#endif
#ifndef PSNODE_K4F_DW_PAIR
#define PSNODE_K4F_DW_PAIR 4
#endif
    constexpr bool DEFER8 = NWV >= 8 && PSNODE_K4F_DEFER8 && PSNODE_K4F_DW_PAIR && !REC;
    int pend_par = 0;
    auto dw_groups = [&](const int par, const f4 hT, f4 (&acc)[NWV]) {
        constexpr int DG = PSNODE_K4F_DW_PAIR <= 1 ? 2 : PSNODE_K4F_DW_PAIR;      // chunks per group
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
    // pacc: the accumulators of the layer handled by the previous call (its gradient MFMAs run here, between this layer's LDS write and barrier)
    auto midT = [&](const int layer, const f4 d, const f4 hT, f4 (&acc)[NWV], f4 (*pacc)[NWV] = nullptr) -> f4 {
        put(tile(p, w), d);
        const f4* wl = wT + ((size_t)layer * NWV * NWV + w) * 64 + l;
        f4 wq = wl[0];
#ifndef PSNODE_K4F_W_AHEAD
#define PSNODE_K4F_W_AHEAD 1      // two-role chain: the layer's weight chunks are read from LDS BEFORE the exchange's barrier (they do not depend on it): behind it
                                  // only the three tiles of the other waves are left to read -- half of the LDS traffic on the critical path
#endif
        constexpr bool WAH = ROLES && PSNODE_K4F_W_AHEAD;
        f4 wah[WAH ? NWV : 1];
        if constexpr (WAH) {
#pragma unroll
            for (int c = 1; c < NWV; ++c) wah[c] = wl[c * NWV * 64];
        }
        f4 accA = fm4(wq[0], d[0], f4{0.f, 0.f, 0.f, 0.f}), accB = fm4(wq[1], d[1], f4{0.f, 0.f, 0.f, 0.f});
        accA = fm4(wq[2], d[2], accA); accB = fm4(wq[3], d[3], accB);
        if constexpr (DEFER || DEFER8) { if (pacc) flush(*pacc); }
        __builtin_amdgcn_sched_barrier(0);
        lds_barrier();
#ifndef PSNODE_K4F_TREAD_AHEAD
#define PSNODE_K4F_TREAD_AHEAD 1     // <= 4 waves (one wave per SIMD, registers to spare): every LDS read of the layer -- the other waves' tiles, the
                                     // weight chunks and the transposed tiles of the weight gradient -- is in flight before the first MFMA that
                                     // needs one.  Left to the scheduler the transposed reads recycled two registers, each pair of MFMAs behind
                                     // its own lgkmcnt(0) (found in the ISA)
#endif
        if constexpr (PREFETCH_ALL && PSNODE_K4F_TREAD_AHEAD) {
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
#ifndef PSNODE_K4F_DP_GROUP
#define PSNODE_K4F_DP_GROUP 7      // 8 waves, transposed layers: tiles and weight chunks of this many chunks read ahead of their MFMAs (0: as the forward layers).
                                     // hidden-128 training step RK4, 0 / 3 / 4 / 7: ODE (K4f) 35.9 / 34.6 / 36.0 / 34.5 ms, DAE (K7f) 52.1 / 51.9 / 51.9 / 52.0 (profiles/r03y_dp_group_ab.txt)
#endif
        if constexpr (PSNODE_K4F_DP_GROUP > 0 && NWV >= 8) {
            constexpr int PG = PSNODE_K4F_DP_GROUP > 0 ? PSNODE_K4F_DP_GROUP : 1;
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
#ifndef PSNODE_K4F_DW_PAIR
#define PSNODE_K4F_DW_PAIR 4     // 8 waves: the weight-gradient MFMAs of two chunks interleaved (two accumulator chains, both transposed tiles read first)
#endif
        {
        if constexpr (PSNODE_K4F_DW_PAIR && NWV >= 8 && !REC) {      // (recompute instance at RK4: 48.2 -> 50.3 ms with it, profiles/r03y_dw_pair_ab.txt)
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
        }
        p ^= 1;
        return accA + accB;
    };
    // 8-byte all-reduce of rows r < 2 over the waves (fixed order)
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
    // 16-byte form: the last stage's dL/dx rides with the step's dL/dz (one exchange less per step: a quarter of them at Euler)
    auto allreduce4 = [&](const f4 part, f4 (*pacc)[NWV] = nullptr) -> f4 {
        put(tile(p, w), part);
        if constexpr (DEFER || DEFER8) { if (pacc) flush(*pacc); }
        lds_barrier();
        f4 out = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NWV; ++c) out += getl(tile(p, c));
        p ^= 1;
        return out;
    };
    // split-K over the waves' own units (4 MFMAs)
    auto own4 = [&](const float (&wq)[4], const f4 h) -> f4 {
        f4 accA = fm4(wq[0], h[0], f4{0.f, 0.f, 0.f, 0.f}), accB = fm4(wq[1], h[1], f4{0.f, 0.f, 0.f, 0.f});
        accA = fm4(wq[2], h[2], accA); accB = fm4(wq[3], h[3], accB);
        return accA + accB;
    };

    // addressing: sbase(uniform row base) + 32-bit per-lane BYTE offset (psnode_common.h: ldg / stg)
    const unsigned offH = 4u * ((unsigned)(b * H) + 16 * w + 4 * g);
    const unsigned offX = 4u * ((unsigned)(b * xd) + g);
    unsigned offXc[NX];                                                    // the same with the column clamped into the row (ldg_sel)
#pragma unroll
    for (int r = 0; r < NX; ++r) offXc[r] = 4u * ((unsigned)(b * xd) + (4 * r + g < xd ? 4 * r + g : 0));
    const unsigned offT = 4u * (unsigned)(b * a.t.sb), offZ = 4u * (unsigned)(b * a.z.sb), offZJ = 4u * (unsigned)(b * a.zjb);
    auto load_ext = [&, offZ, offZJ](const long long k, const int ev, float (&dst)[NZ]) {
        if constexpr (NZM > 0) {
            const gptr<const float> row = sbase(ev >= 0 ? a.zj + (long long)ev * a.zje : a.z.p + k * a.z.st);
            const unsigned m_ = ev >= 0 ? ~0u : 0u, zo = (offZJ & m_) | (offZ & ~m_);
#pragma unroll
            for (int m = 0; m < NZM; ++m) dst[m] = ldg<float>(row, zo + 4u * ecol[m]);     // padding slots read column 0 against a zero weight
        }
    };
    auto x2on = [&](const int r) -> bool { return 4 * r + g < xd; };
    auto load_x2 = [&](const float* base, const long long k, float (&dst)[NX]) {
        const gptr<const float> row = sbase(base + k * a.B * xd);
#pragma unroll
        for (int r = 0; r < NX; ++r) dst[r] = ldg<float>(row, offXc[r]);      // RAW and branch-free (clamped column, psnode_common.h: ldg_sel):
                                                                                // every caller is a request a step / stage ahead, x2on() masks at the consumer
    };
    // clock and event index of a step are RAW prefetched values (one grid point / one table entry per step, requested a step ahead);
    // the difference and the readfirstlane happen a step later, at the consumer.  Subtracting / broadcasting right behind the load --
    // inside the `if (k > 0)` of the prefetch -- made the compiler wait out the memory round trip on the spot, every step (K4w did that).
    auto load_t = [&](const long long k) -> float { return ldg<float>(sbase(a.t.p + k * a.t.st), offT); };
    int lane_zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));
    const bool has_ev = a.ev != nullptr;
    const int* evp = (has_ev ? a.ev : reinterpret_cast<const int*>(a.t.p)) + lane_zero;   // per-lane load: the value stays in a VGPR until it is used

    // ---- accumulators (whole launch)
    const f4 zero4 = f4{0.f, 0.f, 0.f, 0.f};
    f4 accW3[NWV], accW2[NWV], accW4 = zero4, accW1s = zero4, S1 = zero4, S2 = zero4, S3 = zero4;
#pragma unroll
    for (int c = 0; c < NWV; ++c) { accW3[c] = zero4; accW2[c] = zero4; }
    f2 db4 = f2{0.f, 0.f};
    float gcar[NX];
#pragma unroll
    for (int r = 0; r < NX; ++r) gcar[r] = 0.0f;

    const long long nrow = a.B, nT = a.T;
    float x0n[NX] = {}, ginn[NX] = {}, extn[NZ] = {};
    float t_hi = 0.0f, t_lo = 0.0f;          // t[k+1], t[k] of the step about to run
    int evn = -1, evr = -1;                  // evn: event index of the step about to run; evr: raw table entry of the one after
    if (nT >= 2) {
        evn = a.ev ? __builtin_amdgcn_readfirstlane(a.ev[nT - 2]) : -1;
        evr = evp[nT >= 3 ? nT - 3 : 0];
        load_x2(a.xs, nT - 2, x0n);
        load_x2(a.gout, nT - 1, ginn);
        load_ext(nT - 2, evn, extn);
        t_hi = load_t(nT - 1);
        t_lo = load_t(nT - 2);
    }
    // REC = false: rows of (step, stage), linear index idx = k S + s, walked downwards; requested one stage ahead
    const long long act_layer = a.B * H;
    auto load_saved = [&](const long long idx, f4& q1, f4& q2, f4& q3, float (&xq)[NX]) {
        const float* rb = a.sact + (size_t)idx * 3 * act_layer;
        q1 = ldg_nt<f4, (NWV >= 8)>(sbase(rb), offH);
        q2 = ldg_nt<f4, (NWV >= 8)>(sbase(rb + act_layer), offH);
        q3 = ldg_nt<f4, (NWV >= 8)>(sbase(rb + 2 * act_layer), offH);
        load_x2(a.sxst, idx, xq);
    };
    f4 sv1 = zero4, sv2 = zero4, sv3 = zero4;
    float svx[NX] = {};
#ifndef PSNODE_K4F_SAVED_AHEAD_ROLES
#define PSNODE_K4F_SAVED_AHEAD_ROLES 3     // two-role form (registers to spare on the chain): 3 = the rows are requested TWO stages ahead
#endif
    constexpr int SAHEAD = ROLES ? PSNODE_K4F_SAVED_AHEAD_ROLES : PSNODE_K4F_SAVED_AHEAD_DEFAULT;
    // SAHEAD == 3: two register sets; the rows of linear index idx live in set (idx parity) -- a compile-time index: `s & 1` when the
    // stage count is even, the parity of the step body (KP) at Euler, whose time loop is therefore written out twice per iteration.  (Moving
    // the second set up with register copies instead would wait for the younger request at the copy.)
    f4 tv1 = zero4, tv2 = zero4, tv3 = zero4;
    float tvx[NX] = {};
    constexpr int QLAST = S == 1 ? 0 : ((S - 1) & 1);      // set of the very first rows consumed
    if constexpr (!REC) {
        if (nT >= 2) {
            const long long last = (nT - 2) * S + (S - 1);
            if constexpr (SAHEAD == 3) {
                if constexpr (QLAST == 0) { load_saved(last, sv1, sv2, sv3, svx); load_saved(last > 0 ? last - 1 : 0, tv1, tv2, tv3, tvx); }
                else { load_saved(last, tv1, tv2, tv3, tvx); load_saved(last > 0 ? last - 1 : 0, sv1, sv2, sv3, svx); }
            } else load_saved(last, sv1, sv2, sv3, svx);
        }
    }
    auto step = [&](const long long k, auto kp_tag) {
        constexpr int KP = decltype(kp_tag)::value;
        (void)KP;
        float x0[NX], gin[NX], ext[NZ];
#pragma unroll
        for (int r = 0; r < NX; ++r) { x0[r] = x2on(r) ? x0n[r] : 0.0f; gin[r] = x2on(r) ? ginn[r] : 0.0f; }
#pragma unroll
        for (int m = 0; m < NZ; ++m) ext[m] = extn[m];
        const float h_ = t_hi - t_lo;
        const int ev = evn;
        f4 cz = c0;
#pragma unroll
        for (int m = 0; m < NZM; ++m) cz = fm4(w1z[m], ext[m] - a0e[m], cz);

        // ---- phase A: stage evaluations (K1's plan)
        float X[S][NX], ks[S][NX];
        f4 h1[(STREAM || !REC) ? 1 : S], h2[(STREAM || !REC) ? 1 : S], h3[(STREAM || !REC) ? 1 : S];
        f4 la1 = zero4, la2 = zero4, la3 = zero4;    // ELU outputs of the last stage evaluated (the first one the backward half needs)
        if constexpr (STREAM) dma_wait();       // forward images of both layers (refilled behind their last use of the previous phase B)
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
                a3 = elu_quad(mid_lds(1, b3, a2));
                if (s == S - 1) dma_layer(pack_t, 1);
            } else { a2 = elu_quad(mid(w2r, b2, a1)); a3 = elu_quad(mid(w3r, b3, a2)); }
            if constexpr (STREAM) {
                if (s < S - 1) {     // park the stage's activations in the ring (each lane re-reads exactly what it wrote)
                    const size_t rb = (size_t)(3 * s) * nrow * H;
                    stg<f4>(sbase(a.ring + rb), offH, a1);
                    stg<f4>(sbase(a.ring + rb + nrow * H), offH, a2);
                    stg<f4>(sbase(a.ring + rb + 2 * nrow * H), offH, a3);
                }
            } else { h1[s] = a1; h2[s] = a2; h3[s] = a3; }
            if (s == S - 1) { la1 = a1; la2 = a2; la3 = a3; }
            if (s < S - 1) {         // the last stage's derivative feeds x[k+1] only, which the backward does not need
                const f4 part = own4(w4, a3);
                const f2 kk = allreduce2(f2{part[0], part[1]}, f2{b4[0], b4[1]});
                ks[s][0] = kk[0];
                if constexpr (NX > 1) ks[s][1] = kk[1];
            }
        }

        // ---- phase B: stages backwards.  The next step's inputs are requested here and consumed a whole backward half later.
        // (unconditional, with the row indices clamped at the first step: a branch around the loads makes their results phi values, the
        //  copies into the phi registers land inside the branch, and the wait for them with it -- the whole memory round trip, every step)
        auto prefetch_next = [&]() {
            const long long kp = k > 0 ? k - 1 : 0;
            evn = has_ev ? __builtin_amdgcn_readfirstlane(evr) : -1;          // requested a step ago
            evr = evp[k >= 2 ? k - 2 : 0];
            load_x2(a.xs, kp, x0n);
            load_x2(a.gout, kp + 1, ginn);
            load_ext(kp, evn, extn);
            t_hi = t_lo;
            t_lo = load_t(kp);
        };
#ifndef PSNODE_K4F_NEXT_LATE
#define PSNODE_K4F_NEXT_LATE 0
#endif
        constexpr bool NEXT_LATE = !REC && NWV >= 8 && PSNODE_K4F_NEXT_LATE;
        if constexpr (!STREAM && !NEXT_LATE) prefetch_next();
        (void)cz; (void)x0;
        float gks[S][NX], gx0[NX];
#pragma unroll
        for (int r = 0; r < NX; ++r) {
            const float g1 = (a.true_x ? 0.0f : gcar[r]) + (valid ? gin[r] : 0.0f);      // (teacher forcing: x[k+1] is an output only)
            gx0[r] = g1;
#pragma unroll
            for (int s = 0; s < S; ++s) gks[s][r] = (h_ * rk_b(METHOD, s)) * g1;
        }
        f2 gzp = f2{0.f, 0.f};                    // z rows of F^T delta1: this wave's partial, summed over the stages (z is frozen over them)
        f2 gzr = f2{0.f, 0.f};                    // ... all-reduced with the last stage's dL/dx
        f4 na1 = la1, na2 = la2, na3 = la3;       // STREAM: activations of the stage handled next, requested one stage ahead
#pragma unroll
        for (int s = S - 1; s >= 0; --s) {
            f4 a1, a2, a3;
            if constexpr (!REC) {
#ifndef PSNODE_K4F_SAVED_AHEAD
#define PSNODE_K4F_SAVED_AHEAD 2      // 0: rows of a stage requested at its top; 1: a whole stage ahead; 2: late in the previous stage (in front of its last
                                      // exchange).  vmcnt counts in order: every spilled-invariant reload behind an outstanding HBM request waits for it, so
                                      // requesting EARLY is what exposed the latency at RK4 (43.8 / 39.2 / 37.9 ms per hidden-128 training step for 1 / 0 / 2;
                                      // Euler, no spills: 9.3 / 10.5 / 9.4: profiles/r03t_ahead.txt)
#endif
                const long long idx = k * S + s;
                if constexpr (SAHEAD) {
                    if constexpr (SAHEAD == 3) {      // two stages ahead: this stage's set is consumed and refilled with the rows of idx - 2
                        const int Q = S == 1 ? KP : (s & 1);      // (compile-time once the stage loop is unrolled)
                        if (Q == 0) {
                            a1 = sv1; a2 = sv2; a3 = sv3;
#pragma unroll
                            for (int r = 0; r < NX; ++r) X[s][r] = x2on(r) ? svx[r] : 0.0f;
                            load_saved(idx > 1 ? idx - 2 : 0, sv1, sv2, sv3, svx);
                        } else {
                            a1 = tv1; a2 = tv2; a3 = tv3;
#pragma unroll
                            for (int r = 0; r < NX; ++r) X[s][r] = x2on(r) ? tvx[r] : 0.0f;
                            load_saved(idx > 1 ? idx - 2 : 0, tv1, tv2, tv3, tvx);
                        }
                    } else {
                    a1 = sv1; a2 = sv2; a3 = sv3;
#pragma unroll
                    for (int r = 0; r < NX; ++r) X[s][r] = x2on(r) ? svx[r] : 0.0f;
                    if constexpr (SAHEAD == 1) load_saved(idx > 0 ? idx - 1 : 0, sv1, sv2, sv3, svx);
                    }
                } else {
                    load_saved(idx, a1, a2, a3, svx);
#pragma unroll
                    for (int r = 0; r < NX; ++r) X[s][r] = x2on(r) ? svx[r] : 0.0f;
                }
            } else if constexpr (STREAM) {
#ifndef PSNODE_K4F_RING_LATE
#define PSNODE_K4F_RING_LATE 1
#endif
                a1 = na1; a2 = na2; a3 = na3;
                if constexpr (!PSNODE_K4F_RING_LATE) {
                    if (s > 0) {
                        const size_t rb = (size_t)(3 * (s - 1)) * nrow * H;
                        na1 = ldg<f4>(sbase(a.ring + rb), offH);
                        na2 = ldg<f4>(sbase(a.ring + rb + nrow * H), offH);
                        na3 = ldg<f4>(sbase(a.ring + rb + 2 * nrow * H), offH);
                    }
                }
            } else {
                a1 = h1[s]; a2 = h2[s]; a3 = h3[s];
            }
            const f2 gk = f2{gks[s][0], NX > 1 ? gks[s][1] : 0.0f};
            db4 += gk;
            if constexpr (RSM) {      // gk and the stage input for the gradient waves (every chain wave holds the same values)
                if (w == 0) {
                    put(roles_gk_tile<NWV>(xb), f4{gk[0], gk[1], 0.f, 0.f});
                    put(roles_s_tile<NWV>(xb), f4{X[s][0], NX > 1 ? X[s][1] : 0.0f, (NZM > 0 && g < ne) ? ext[0] : 0.0f,
                                                  (NZM > 1 && 4 + g < ne) ? ext[NZM > 1 ? 1 : 0] : 0.0f});
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
            if constexpr (STREAM) {
                if (s == S - 1) {    // the transposed images must have landed; the next step's inputs are requested BEHIND that wait
                    dma_wait();
                    prefetch_next();
                }
            }
            const f4 d2 = midT(1, d3, h2T, accW3) * elu_grad_quad(a2);
            S2 += d2;
            if constexpr (STREAM) { if (s == 0 && k > 0) dma_layer(pack_f, 1); }     // W3's region: forward image for the next step
            const f4 h1T = ROLES ? zero4 : transpose(a1);
            const f4 d1 = midT(0, d2, h1T, accW2, &accW3) * elu_grad_quad(a1);
            S1 += d1;
            if constexpr (RSM) put(roles_d1_tile<NWV>(xb, w), d1);
            if constexpr (STREAM) { if (s == 0 && k > 0) dma_layer(pack_f, 0); }
            const f4 ft = own4(fT, d1);           // rows 0..1: gX partial, rows 2..3: gz partial
            gzp += f2{ft[2], ft[3]};
            if constexpr (NEXT_LATE) { if (s == PSNODE_K4F_NEXT_LATE - 1) prefetch_next(); }
            if constexpr (STREAM && PSNODE_K4F_RING_LATE) {
                if (s > 0) {
                    const size_t rb = (size_t)(3 * (s - 1)) * nrow * H;
                    na1 = ldg<f4>(sbase(a.ring + rb), offH);
                    na2 = ldg<f4>(sbase(a.ring + rb + nrow * H), offH);
                    na3 = ldg<f4>(sbase(a.ring + rb + 2 * nrow * H), offH);
                }
            }
            if constexpr (!REC && SAHEAD == 2) {      // request the next stage's rows late in this one (see SAVED_AHEAD)
                const long long idx = k * S + s;
                load_saved(idx > 0 ? idx - 1 : 0, sv1, sv2, sv3, svx);
            }
            f2 gx;
            if (NZM > 0 && s == 0) {      // (compile-time once the stage loop is unrolled)
                const f4 r4 = allreduce4(f4{ft[0], ft[1], gzp[0], gzp[1]}, &accW2);
                gx = f2{r4[0], r4[1]};
                gzr = f2{r4[2], r4[3]};
            } else gx = allreduce2(f2{ft[0], ft[1]}, f2{0.f, 0.f}, &accW2);
            if constexpr (!RSM) {   // dW1 (`s` columns) += delta1 (x) s
                const f4 dT = transpose(d1);
                put(scr, f4{X[s][0], NX > 1 ? X[s][1] : 0.0f, (NZM > 0 && g < ne) ? ext[0] : 0.0f, (NZM > 1 && 4 + g < ne) ? ext[NZM > 1 ? 1 : 0] : 0.0f});
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
#pragma unroll
        for (int r = 0; r < NX; ++r) gcar[r] = gx0[r];
        if constexpr (NZM > 0) {     // gradient of this step's external input: row 0 -> z-dim g, row 1 -> z-dim 4+g

            if (w == 0 && valid) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int e = 4 * q + g;
                    if (e < ne) {
                        if (ev >= 0) { if (a.gzj) a.gzj[(b * a.n_events + ev) * ne + e] = gzr[q]; }
                        if (a.gz) a.gz[(k * a.B + b) * ne + e] = ev >= 0 ? 0.0f : gzr[q];
                    }
                }
            }
        }
    };
    if constexpr (S == 1 && SAHEAD == 3 && !REC) {
        long long k = nT - 2;
        for (; k >= 1; k -= 2) { step(k, std::integral_constant<int, 0>{}); step(k - 1, std::integral_constant<int, 1>{}); }
        if (k == 0) step(0, std::integral_constant<int, 0>{});
    } else {
        for (long long k = nT - 2; k >= 0; --k) step(k, std::integral_constant<int, 0>{});
    }

    // ---- epilogue
    if constexpr (STREAM) dma_wait();
    if (w == 0 && valid) {
#pragma unroll
        for (int r = 0; r < NX; ++r)
            if (4 * r + g < xd) a.gx0[b * xd + 4 * r + g] = gcar[r] + a.gout[b * xd + 4 * r + g];
        if (a.gz && nT >= 1) {       // z[T-1] is never read by the ODE loop
#pragma unroll
            for (int q = 0; q < 2; ++q) if (4 * q + g < ne) a.gz[((nT - 1) * a.B + b) * ne + 4 * q + g] = 0.0f;
        }
    }
    const int K1 = 3 * n;
    if constexpr (RSM) put(scr, S1);      // for the gradient wave with these units (read behind the barrier below)
    {   // d all_initial[c] = sum_u (Wa - Wd)[u][c] S1[u]: split-K over the waves' own units, 16-byte all-reduce, rows c = 4g + r
        float at[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int uu = 16 * w + 4 * g + r;
            at[r] = (i < n && uu < HR) ? a.w1[(size_t)uu * K1 + i] - a.w1[(size_t)uu * K1 + n + i] : 0.0f;
        }
        const f4 part = own4(at, S1);
        put(tile(p, w), part);
        lds_barrier();
        f4 ga = zero4;
#pragma unroll
        for (int c = 0; c < NWV; ++c) ga += getl(tile(p, c));
        p ^= 1;
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
}

size_t fused_lds_bytes(int nw, bool roles = false) { return (wide_t_floats(nw) + ((size_t)(roles ? 5 : 3) * nw + (roles ? 2 : 0)) * FTILE) * sizeof(float); }
size_t fused_ring_floats(int nw, int method, long long B) { return nw >= 8 ? (size_t)rk_stages(method) * 3 * (size_t)B * 16 * nw : 0; }
int fused_np(int hr, int xd, int zd) { const int n = xd + zd; return hr * 3 * n + hr + 2 * (hr * hr + hr) + xd * hr + xd; }

template <int METHOD, int NWV>
hipError_t launch_fused(const FusedDev& a, int NZM, const float* pde, const f4* pt, const f4* pf, int NA, hipStream_t s) {
    constexpr bool RL = PSNODE_K4F_ROLES && NWV <= 4;
    const bool roles = RL && a.sact != nullptr;
    const dim3 grid((unsigned)((a.B + TBM - 1) / TBM)), block(64 * NWV * (roles ? 2 : 1));
    const size_t lds = fused_lds_bytes(NWV, roles);
#define PSNODE_FUSED(NZM_)                                                                                                      \
    {                                                                                                                           \
        auto kern = a.sact ? &ode_backward_fused_kernel<METHOD, NZM_, NWV, false, RL> : &ode_backward_fused_kernel<METHOD, NZM_, NWV, true>; \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return e;                                                                                          \
        hipLaunchKernelGGL(kern, grid, block, lds, s, a, pde, pt, pf, NA);                                                      \
        return hipGetLastError();                                                                                               \
    }
    switch (NZM) {
        case 0: PSNODE_FUSED(0)
        case 1: PSNODE_FUSED(1)
        case 2: PSNODE_FUSED(2)
        case 3: PSNODE_FUSED(3)
        case 4: PSNODE_FUSED(4)
        default: return hipErrorNotSupported;
    }
#undef PSNODE_FUSED
}

template <int NWV>
hipError_t launch_fused_method(const FusedDev& a, int NZM, const float* pde, const f4* pt, const f4* pf, int NA, hipStream_t s) {
    switch (a.method) {
        case PSNODE_EULER: return launch_fused<PSNODE_EULER, NWV>(a, NZM, pde, pt, pf, NA, s);
        case PSNODE_MIDPOINT: return launch_fused<PSNODE_MIDPOINT, NWV>(a, NZM, pde, pt, pf, NA, s);
        default: return launch_fused<PSNODE_RK4_38, NWV>(a, NZM, pde, pt, pf, NA, s);
    }
}

}  // namespace

// ---- entry points used by psnode_backward.hip (C ABI psnode_ode_backward_f32)
bool fused_bwd_shape_ok(const psnode_ode_bwd_args_f32* a) {
    if (a->x_dim < 1 || a->x_dim > 4 * kNXc || a->z_dim < 0 || 2 * a->z_dim > 4 * kMaxNZM) return false;
    if (!wide_hidden(a->de)) return false;
    return a->de.in_dim == 3 * (a->x_dim + a->z_dim) && a->de.out_dim[3] == a->x_dim;
}

size_t fused_bwd_workspace_floats(const psnode_ode_bwd_args_f32* a) {
    const int nw = wide_hidden(a->de) / 16, n = a->x_dim + a->z_dim;
    const size_t nwg = (size_t)((a->B + TBM - 1) / TBM);
    return ((wide_fwd_floats(nw, n) + 63) / 64) * 64 + 2 * wide_t_floats(nw) + nwg * fused_np(a->de.out_dim[0], a->x_dim, a->z_dim) +
           fused_ring_floats(nw, a->method, a->B) + 256;
}

int fused_bwd_launch(const psnode_ode_bwd_args_f32* p, float* workspace, hipStream_t s) {
    const int H = wide_hidden(p->de), nw = H / 16, xd = p->x_dim, zd = p->z_dim, n = xd + zd, HR = p->de.out_dim[0];
    {   // per-lane offsets inside a row are 32-bit next to a scalar row base
        const int64_t lim = (int64_t)1 << 30, Bm = p->B;
        const int64_t sb[] = {H, p->t.stride_b, zd > 0 ? p->z.stride_b : 0, p->event_idx && zd > 0 ? p->zj_stride_b : 0};
        for (int64_t q : sb) if (q < 0 || Bm * q + 64 >= lim) return PSNODE_ERR_DIMS;
    }
    const int NZM = (2 * zd + 3) / 4, NA = (n + 3) / 4;
    float* pde = workspace;
    f4* pt = reinterpret_cast<f4*>(pde + ((wide_fwd_floats(nw, n) + 63) / 64) * 64);
    f4* pf = pt + wide_t_floats(nw) / 4;
    float* wpart = reinterpret_cast<float*>(pf) + wide_t_floats(nw);
    const size_t nwg = (size_t)((p->B + TBM - 1) / TBM);
    const int NP = fused_np(HR, xd, zd);
    float* ring = wpart + ((nwg * NP + 63) / 64) * 64;
    PackMfma f;
    memset(&f, 0, sizeof(f));
    f.ae = 0; f.nw = nw; f.xd = xd; f.ne = zd; f.n = n; f.nzv = zd; f.NX = kNXc; f.NB = 0; f.NE = NZM; f.NA = NA; f.fold = 1;
    f.hreal = HR;
    f.w1 = p->de.weight[0]; f.b1 = p->de.bias[0]; f.w2 = p->de.weight[1]; f.b2 = p->de.bias[1];
    f.w3 = p->de.weight[2]; f.b3 = p->de.bias[2]; f.w4 = p->de.weight[3]; f.b4 = p->de.bias[3];
    f.out_dim = xd; f.out = pde;
    hipLaunchKernelGGL(pack_wide_fwd_kernel, dim3(32), dim3(256), 0, s, f);
    PackWideT t{nw, HR, p->de.weight[1], p->de.weight[2], pt};
    hipLaunchKernelGGL(pack_wide_t_kernel, dim3(64), dim3(256), 0, s, t);
    if (nw >= 8) {
        PackWideT tf{nw, HR, p->de.weight[1], p->de.weight[2], pf};
        hipLaunchKernelGGL(pack_wide_f_kernel, dim3(64), dim3(256), 0, s, tf);
    }
    if (hipGetLastError() != hipSuccess) return PSNODE_ERR_HIP;
    FusedDev a;
    memset(&a, 0, sizeof(a));
    a.method = p->method; a.xd = xd; a.zd = zd; a.hreal = HR; a.n_events = p->n_events; a.NP = NP; a.T = p->T; a.B = p->B;
    a.w1 = p->de.weight[0]; a.w4 = p->de.weight[3];
    a.t = ViewDev{p->t.ptr, p->t.stride_t, p->t.stride_b};
    a.z = ViewDev{p->z.ptr, p->z.stride_t, p->z.stride_b};
    a.a0 = p->all_initial; a.ev = p->event_idx; a.zj = p->z_jump; a.zjb = p->zj_stride_b; a.zje = p->zj_stride_e;
    a.xs = p->xs; a.gout = p->grad_xs; a.gx0 = p->grad_x0; a.gz = p->grad_z; a.gzj = p->grad_z_jump; a.ga0 = p->grad_all_initial;
    a.wpart = wpart; a.ring = ring;
    a.sact = p->saved_act; a.sxst = p->saved_xstage;
    a.true_x = (p->flags & PSNODE_FLAG_INPUT_TRUE_X) ? 1 : 0;
    if (a.true_x && a.sact) return PSNODE_ERR_UNSUPPORTED;     // a teacher-forced forward saves nothing: recompute form only
    hipError_t e;
    switch (nw) {
        case 2: e = launch_fused_method<2>(a, NZM, pde, pt, pf, NA, s); break;
        case 4: e = launch_fused_method<4>(a, NZM, pde, pt, pf, NA, s); break;
        default: e = launch_fused_method<8>(a, NZM, pde, pt, pf, NA, s); break;
    }
    if (e == hipErrorNotSupported) return PSNODE_ERR_UNSUPPORTED;
    if (e != hipSuccess) return PSNODE_ERR_HIP;
    return launch_reduce_partials(wpart, p->grad_params, nullptr, NP, 0, (int)nwg, s) == hipSuccess ? PSNODE_OK : PSNODE_ERR_HIP;
}

}  // namespace psnode
