// K1x -- the exchange-free form of the fused ODE integrator (round 5; DESIGN.md "K1x"): ONE WAVE owns 4 trajectories and ALL hidden
// units of `3n -> H -> H -> H -> x_dim` (H <= 64, x_dim <= 8, z_dim <= 8), so nothing crosses waves: no LDS, no barrier.  At the headline
// batch (4096 trajectories = 1024 waves = one per SIMD) K1's 4-wave tile spends ~23 % of a stage in its three LDS exchanges
// (ds_write -> s_barrier -> ds_read with nothing else to issue); here the layer-to-layer re-layout is a 4 x 4 transpose inside every lane
// quad (8 v_cndmask_b32_dpp) and the L4 reduction a 4-stage lane butterfly (profiles/r05_ubench_4x4.txt: 325 vs 402 ns per H -> H layer).
//
// Arithmetic on v_mfma_f32_4x4x1_16B_f32 (16 blocks of D_b(4x4) += A_b(4x1) B_b(1x4); exact fp32, the same 64 flop/clk/SIMD peak as
// 16x16x4 -- it issues every 10 cycles instead of 8).  Lane l = (block b = l >> 2, c = l & 3).
//   hidden layers   A = activations, B = weights:   D[traj r][unit 4b'+c'] += act[k][traj r] * W[unit 4b'+c'][k]
//       A layout: four registers hA[cc], lane (b, t) holds act[k = 4b + cc][trajectory t]; CBSZ = 4 / ABID = b hands block b's rows to all
//       16 blocks, so ONE MFMA adds one k for all 64 units and 4 trajectories; B = one VGPR per k (64 per H -> H matrix, resident).
//       D layout: register r = trajectory, lane (b', c') = unit 4b' + c'.  ELU runs on D; the in-quad transpose turns D into the next A.
//   L4 (64 -> x_dim <= 8), split K over the BLOCKS, roles swapped (A = weights, B = activations, no broadcast):
//       P_j[dim 4j + r][traj c] (block b) += W4[4j + r][4b + cc] * act[4b + cc][c]      8 MFMAs, then the sum over the 16 blocks:
//       v_permlane32_swap (lane bit 5), v_permlane16_swap (bit 4) -- each folds TWO registers into one -- and two DPP row rotations
//       (bits 3, 2).  Result = the state layout below, so the RK update is 2 registers wide and feeds L1 without any data movement.
//   State layout: X01 / X23, lane in row rho = l >> 4, trajectory t = l & 3:  X01 = x[4 (rho >> 1) + 2 (rho & 1)][t],  X23 = the NEXT dim
//       (adjacent dims: a row's two state values are one 8-byte store).  L1 reads dim d with ABID = 4 rho(d) from X01 or X23 (8 MFMAs per
//       stage, folded image as K1).
// External inputs: slot q (K1's order: q < ne -> ext[q] - a0, ne <= q < 2 ne -> ext[q - ne]) lives in block q of ONE register (lane (q, t));
// the per-step constant of L1 is 4 NZM MFMAs with ABID = q.  Hidden layers run in the log2e-scaled domain (psnode_common.h:
// elu_quad_scaled).  Inference only (no saved activations); teacher forcing (input_true_x) is a uniform runtime branch.
#include <string.h>

#include "psnode_mfma_x.h"

namespace psnode {
namespace {

// register image: pack[reg][lane] (64 lanes)
struct XRegs {
    static constexpr int W2 = 0, W3 = 64, W1X = 128, W1E = 136, W1A = 152, B1 = 168, B2 = 169, B3 = 170, W4A = 171, B4C = 179, COUNT = 187;
};
struct PackX {
    int xd, zd, n, hreal;
    float sc;           // log2e: the scaled ELU domain of the inference kernel; 1: the training forward (plain domain, bit-exact weights)
    const float *w1, *b1, *w2, *b2, *w3, *b3, *w4, *b4;
    float* out;
};
__global__ void pack_x_kernel(const PackX p) {
    const int H = p.hreal, n = p.n, xd = p.xd, ne = p.zd, K1 = 3 * n;
    const float kLog2e = p.sc;                                 // (shadows the constant: 1 for the training forward's plain-domain image)
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < XRegs::COUNT * 64; idx += gridDim.x * blockDim.x) {
        const int reg = idx >> 6, l = idx & 63, b = l >> 2, c = l & 3, u = l;      // B operands: lane (b', c') = unit 4b' + c' = l
        float v = 0.0f;
        if (reg < XRegs::W1X) {                         // H -> H matrices, k in MFMA order: k = 4 bb + cc  (scale-free)
            const int k = reg & 63;
            const float* W = reg < XRegs::W3 ? p.w2 : p.w3;
            if (u < H && k < H) v = W[u * H + k];
        } else if (reg < XRegs::W1E) {                  // x columns of L1, folded: Ws + Wd
            const int d = l1_dim(reg - XRegs::W1X);
            if (u < H && d < xd) v = (p.w1[u * K1 + 2 * n + d] + p.w1[u * K1 + n + d]) * kLog2e;
        } else if (reg < XRegs::W1A) {                  // ext slots
            const int q = reg - XRegs::W1E;
            if (u < H && q < ne) v = p.w1[u * K1 + n + xd + q] * kLog2e;
            else if (u < H && q < 2 * ne) v = p.w1[u * K1 + 2 * n + xd + (q - ne)] * kLog2e;
        } else if (reg < XRegs::B1) {                   // a0 columns (folded: Wa - Wd on the x dims)
            const int q = reg - XRegs::W1A;
            if (u < H && q < n) {
                v = p.w1[u * K1 + q];
                if (q < xd) v -= p.w1[u * K1 + n + q];
                v *= kLog2e;
            }
        } else if (reg == XRegs::B1) { if (u < H) v = p.b1[u] * kLog2e;
        } else if (reg == XRegs::B2) { if (u < H) v = p.b2[u] * kLog2e;
        } else if (reg == XRegs::B3) { if (u < H) v = p.b3[u] * kLog2e;
        } else if (reg < XRegs::B4C) {                  // L4 as A operand: lane (b, r = c) = W4[dim 4j + r][k = 4b + cc] / log2e
            const int j = (reg - XRegs::W4A) >> 2, cc = (reg - XRegs::W4A) & 3, d = 4 * j + c, k = 4 * b + cc;
            if (d < xd && k < H) v = p.w4[d * H + k] / kLog2e;
        } else {                                        // C init of L4: D lane (b, c), register r -> dim 4j + r: the bias enters in block 0 only
            const int j = (reg - XRegs::B4C) >> 2, r = (reg - XRegs::B4C) & 3, d = 4 * j + r;
            if (b == 0 && d < xd) v = p.b4[d];
        }
        p.out[idx] = v;
    }
}


// FAST (a loop variant inside the kernel): no event in the table, no teacher forcing, even x_dim -- the plain inference call.  With one wave per SIMD every scalar instruction and
// every branch of the per-step bookkeeping is wall time (~4 / ~8 cycles each, nothing else to issue: 72 SALU + 19 branches were ~200 ns of
// a 1.05 us Euler step), so the common call gets a loop without the event / teacher-forcing / odd-width code, and BOTH forms peel the last
// step (which prefetches nothing) instead of clamping every row pointer every step.
// SAVE: the training forward (a.sact / a.sxst: what autograd would keep -- the three ELU layers' outputs [T-1,S,3,B,Hp] and the stage inputs
// [T-1,S,B,xd], the rows K4f reads; Hp = the padded width of the backward's tile, 32 or 64).  Plain ELU domain (the pack image is unscaled).
// The A layout after the in-quad transpose holds, per lane (b, t), units 4b .. 4b+3 of trajectory t: ONE 16-byte store per layer.
#ifndef PSNODE_K1X_SAVE_ABL
#define PSNODE_K1X_SAVE_ABL 0     // timing-only ablations of the saving forward (saved rows WRONG): 1 = no saved-row store is issued, 2 = every stage
#endif                            // stores to the rows of (step 0, stage 0) (the same instructions, no HBM traffic)
// ROLES (round 6): the two-role form of the SAVING forward.  A lone wave per SIMD has nothing to issue while one of its 13 saved-row
// stores per stage waits for a queue slot: the 13.1 GB of rows cost K1x +0.9 ms (0.6 of it only when the bytes really go to HBM,
// profiles/r06_k1x_save_ablations.txt) while the 4-wave tile kernels lose half of that in their exchange bubbles.  So the stores move to a
// PARTNER wave (wave w + 4 of the workgroup, one more wave per SIMD: K1x needs 211 registers): the compute wave drops a stage's rows -- the
// stage input and the three layers exactly as it would have stored them -- into an LDS ring (3 ds_write_b128 + 1 ds_write_b64), the partner
// drains it to memory.  Ring = 2 halves of 4 stage slots per pair (28 KB; 112 KB per workgroup), ONE workgroup barrier per 4 stages: the
// compute wave fills half h while the partner drains half h ^ 1; the partner arrives at the barrier with its ds_reads done and its global
// stores still in flight, so their acknowledgement never holds the compute wave.
constexpr int kXSlotFloats = 3 * 256 + 128;          // per stage: three layers [64 lanes][4] + the stage input [64 lanes][2]
constexpr int kXRingFloats = 2 * 4 * kXSlotFloats;   // per pair: 2 halves x 4 stage slots

template <int METHOD>
__device__ __forceinline__ void x_store_role(const IntegrateDev& a, const float* __restrict__ ring, const int l, const int pw) {
    constexpr int S = METHOD == PSNODE_EULER ? 1 : (METHOD == PSNODE_MIDPOINT ? 2 : 4);
    const int b = l >> 2, c = l & 3, rho = l >> 4;
    const long long tile = (long long)blockIdx.x * kXWaves + pw;
    const bool valid = tile * 4 + c < a.B;
    const long long tr = valid ? tile * 4 + c : a.B - 1;
    const int xd = a.xd, nT = (int)a.T;
    const int d01 = 4 * (rho >> 1) + 2 * (rho & 1), d23 = d01 + 1;
    const bool storer = (b & 3) == 0 && valid, st01 = storer && d01 < xd, st23 = storer && d23 < xd, pair_ok = (xd & 1) == 0;
    const int hp = padded_hidden(a.de.out_dim[0]);
    const size_t sa_layer = (size_t)a.B * hp;
    const long long xo_step = a.B * xd;
    const unsigned saoff = (unsigned)(tr * hp + 4 * b) * 4u, xooff = (unsigned)(tr * xd + d01) * 4u;
    const bool sa_on = valid && 4 * b < hp;
    float* sa_run = a.sact;
    float* sx_run = a.sxst;
    const int nstages = nT >= 2 ? (nT - 1) * S : 0, groups = (nstages + 3) >> 2;
    const float* mine = ring + (size_t)pw * kXRingFloats;
    for (int g = 0; g < groups; ++g) {
        lds_barrier();                                   // group g is complete in half g & 1
        const float* hb = mine + (g & 1) * 4 * kXSlotFloats;
        const int nslot = nstages - 4 * g < 4 ? nstages - 4 * g : 4;
        for (int sl = 0; sl < nslot; ++sl) {
            const float* sp = hb + sl * kXSlotFloats;
            const f4 h0 = *reinterpret_cast<const f4*>(sp + 4 * l), h1 = *reinterpret_cast<const f4*>(sp + 256 + 4 * l),
                     h2 = *reinterpret_cast<const f4*>(sp + 512 + 4 * l);
            const f2 xs = *reinterpret_cast<const f2*>(sp + 768 + 2 * l);
            if (pair_ok) {
                if (st01) stg<f2>((gptr<float>)(uintptr_t)sx_run, xooff, xs);
            } else {
                if (st01) stg<float>((gptr<float>)(uintptr_t)sx_run, xooff, xs[0]);
                if (st23) stg<float>((gptr<float>)(uintptr_t)sx_run, xooff + 4u, xs[1]);
            }
            if (sa_on) {
                stg<f4>((gptr<float>)(uintptr_t)sa_run, saoff, h0);
                stg<f4>((gptr<float>)(uintptr_t)(sa_run + sa_layer), saoff, h1);
                stg<f4>((gptr<float>)(uintptr_t)(sa_run + 2 * sa_layer), saoff, h2);
            }
            sx_run += xo_step;
            sa_run += 3 * sa_layer;
        }
    }
}

template <int METHOD, int NZM, bool SAVE, bool ROLES = false>
__global__ __launch_bounds__(64 * kXWaves * (ROLES ? 2 : 1)) void integrate_x_kernel(const IntegrateDev a, const float* __restrict__ pack) {
    static_assert(!ROLES || SAVE, "the two-role form is the saving forward's");
    extern __shared__ __attribute__((aligned(16))) float xring[];
    const int l = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if constexpr (ROLES) {
        if (wv >= kXWaves) { x_store_role<METHOD>(a, xring, l, wv - kXWaves); return; }
    }
    const int b = l >> 2, c = l & 3, rho = l >> 4;
    const long long tile = (long long)blockIdx.x * kXWaves + wv;
    if (!ROLES && tile * 4 >= a.B) return;                          // (nothing is shared between the waves: a surplus wave just leaves; with
                                                                    //  ROLES it stays for the barriers, every lane invalid)
    const bool valid = tile * 4 + c < a.B;
    const long long tr = valid ? tile * 4 + c : a.B - 1;
    const int xd = a.xd, zd = a.zd, ne = zd, n = xd + zd;
    const bool true_x = !SAVE && (a.flags & PSNODE_FLAG_INPUT_TRUE_X) != 0;

    // ---- weights -> registers (once per launch)
    const float* pw = pack + l;
    float w2[64], w3[64], w1x[8], w1e[16], w4a[8];
    f4 b4c[2];
#pragma unroll
    for (int k = 0; k < 64; ++k) { w2[k] = pw[(XRegs::W2 + k) * 64]; w3[k] = pw[(XRegs::W3 + k) * 64]; }
#pragma unroll
    for (int m = 0; m < 8; ++m) { w1x[m] = pw[(XRegs::W1X + m) * 64]; w4a[m] = pw[(XRegs::W4A + m) * 64]; }
#pragma unroll
    for (int q = 0; q < 16; ++q) w1e[q] = q < 4 * NZM ? pw[(XRegs::W1E + q) * 64] : 0.0f;
    const float b1 = pw[XRegs::B1 * 64], b2 = pw[XRegs::B2 * 64], b3 = pw[XRegs::B3 * 64];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) b4c[j][r] = pw[(XRegs::B4C + 4 * j + r) * 64];

    // ---- per-trajectory constants
    const int d01 = 4 * (rho >> 1) + 2 * (rho & 1), d23 = d01 + 1;  // the dims this lane's row carries in X01 / X23 (adjacent)
    const float* a0p = a.a0 + tr * n;
    f4 c0 = f4{b1, b1, b1, b1};                                     // bias + W1[:, a0 columns] . a0: constant for the whole launch
    {
        const float a0A = b < n ? a0p[b] : 0.0f;                    // block q = a0 column q
        float wa[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) wa[q] = pw[(XRegs::W1A + q) * 64];
        c0 = ext_mfmas<0, 16>(wa, a0A, c0);
    }
    const int ecol = b < ne ? b : (b < 2 * ne ? b - ne : 0);        // z column of this lane's ext slot (slot q = block b)
    const float a0e = b < ne ? a0p[xd + b] : 0.0f;
    const bool eon = b < 2 * ne;
    float X01 = 0.0f, X23 = 0.0f;
    {
        const float* x0 = a.x.p + tr * a.x.sb;
        if (d01 < xd) X01 = x0[d01];
        if (d23 < xd) X23 = x0[d23];
    }
    const bool storer = (b & 3) == 0 && valid;                      // the first block of each row writes the row's two dims
    if (a.T < 2 && storer) {                                         // (T >= 2: row 0 is stored by the first pass of the time loop)
        float* row = a.xo + tr * xd;
        if (d01 < xd) row[d01] = X01;
        if (d23 < xd) row[d23] = X23;
    }
    const int nT = (int)a.T;                                         // (32-bit loop counters: there is no 64-bit scalar compare, a `long long` k puts
    if (nT < 2) return;                                              //  every end-of-grid test on the VALU; the launcher refuses T >= 2^31)

    // Addressing of the time loop (psnode_common.h: sbase / ldg): <uniform row base in SGPRs> + <32-bit per-lane byte offset>.  With one wave
    // per SIMD every instruction of the per-step bookkeeping is wall time (~5 cycles each, nothing else to issue): the row bases advance by
    // scalar adds, a lane owns ONE clock entry, ONE external-input column and two state dims.
    const long long tst = a.t.st, xst = a.x.st;
    const bool has_z = zd > 0, has_zj = has_z && a.zj != nullptr;
    const long long zst = has_z ? a.z.st : 0, zje = has_zj ? a.zje : 0;
    const unsigned toff = (unsigned)(tr * a.t.sb) * 4u;
    const unsigned zoff = has_z ? (unsigned)(tr * a.z.sb + ecol) * 4u : toff;          // absent source: the trajectory's clock (meets a zero weight)
    const unsigned zjoff = has_zj ? (unsigned)(tr * a.zjb + ecol) * 4u : toff;
    const float* zbase = has_z ? a.z.p : a.t.p;
    const float* zjbase = has_zj ? a.zj : a.t.p;
    const int d01c = d01 < xd ? d01 : 0, d23c = d23 < xd ? d23 : 0;
    const unsigned xtoff01 = (unsigned)(tr * a.x.sb + d01c) * 4u, xtoff23 = (unsigned)(tr * a.x.sb + d23c) * 4u;
    const unsigned xooff = (unsigned)(tr * xd + d01) * 4u;
    // FAST is decided per launch, on the device: no teacher forcing, even x_dim, and an event table that is absent or holds no event -- the
    // scripts always pass an event list ("no events" = the time -1, neural_00_ODE_01_no_encode.py), so the table is scanned once (T / 64 loads
    // per lane, before the time loop) instead of trusting the pointer.
    bool fast_rt = !true_x && (xd & 1) == 0;
    if (fast_rt && a.ev) {
        int any = -1;
        for (int i = l; i + 1 < nT; i += 64) any = max(any, a.ev[i]);
        fast_rt = __builtin_amdgcn_ballot_w64(any >= 0) == 0;
    }
    // SAVE: uniform running row bases of the saved rows + this lane's byte offsets (its trajectory's row, its four units / its two dims)
    const int hp = SAVE ? padded_hidden(a.de.out_dim[0]) : 0;
    // (ablations 3 / 4, timing only: the same bytes in a TILE-major order -- a wave's three layer rows of a stage in one 3 KB run (3), all
    //  S stages of a step in one 3 S KB run (4) -- to see whether the saving forward's write stall depends on DRAM locality)
    constexpr int kStg = METHOD == PSNODE_EULER ? 1 : (METHOD == PSNODE_MIDPOINT ? 2 : 4);
    const size_t sa_layer = SAVE ? (PSNODE_K1X_SAVE_ABL >= 3 ? (size_t)256 : (size_t)a.B * hp) : 0;
    const size_t sa_stage = SAVE ? (PSNODE_K1X_SAVE_ABL == 4 ? (size_t)768 : 3 * (size_t)a.B * hp) : 0;
    float* sa_run = SAVE ? a.sact : nullptr;
    float* sx_run = SAVE ? a.sxst : nullptr;
    const unsigned saoff = !SAVE ? 0u : (PSNODE_K1X_SAVE_ABL == 3 ? (unsigned)(tile * 768 + c * 64 + 4 * b) * 4u
                                         : (PSNODE_K1X_SAVE_ABL == 4 ? (unsigned)(tile * 768 * kStg + c * 64 + 4 * b) * 4u
                                                                     : (unsigned)(tr * hp + 4 * b) * 4u));
    const bool sa_on = SAVE && valid && 4 * b < hp && PSNODE_K1X_SAVE_ABL != 1;
    float* ring_mine = ROLES ? xring + (size_t)wv * kXRingFloats : nullptr;
    int ring_sl = 0, ring_half = 0;                                  // ROLES: slot inside the half being filled, the half (uniform)
    auto time_loop = [&](auto fast_tag) {
    constexpr bool FAST = decltype(fast_tag)::value;
    auto as_g = [](const float* q) { return (gptr<const float>)(uintptr_t)q; };      // uniform row base (SGPR pair) as a global pointer
    auto load_evb = [&](const int blk) -> int {                       // event indices travel 64 steps at a time (lane i: step 64 blk + i)
        const int i = blk * 64 + l;
        const int v = (a.ev && i + 1 < nT) ? a.ev[i] : -1;
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
        return v;
    };
    int evb = FAST ? -1 : load_evb(0);
    int ev_cur = FAST ? -1 : __builtin_amdgcn_readlane(evb, 0);
    float t_cur = ldg<float>(as_g(a.t.p), toff), t_nxt = ldg<float>(as_g(a.t.p + tst), toff);
    float e_nxt = (!FAST && ev_cur >= 0) ? ldg<float>(as_g(zjbase + (long long)ev_cur * zje), zjoff) : ldg<float>(as_g(zbase), zoff);
    const float* trun = a.t.p + 2 * tst;                              // uniform running row bases: the clock entry / z row the next prefetch reads
    const float* zrun = zbase + zst;
    float* xo_run = a.xo;                                             // step k stores row k (the previous step's result; step 0: the start row)
    const long long xo_step = a.B * xd;
    const float* xt_run = a.x.p;                                      // teacher forcing: the dataset row of grid point k
    const bool st01 = storer && d01 < xd, st23 = storer && d23 < xd;
    const bool pair_ok = FAST || (xd & 1) == 0;                       // even x_dim: both dims exist together and the pair is 8-byte aligned
    auto store_row = [&](float* row) {
        if (pair_ok) {
            if (st01) stg<f2>((gptr<float>)(uintptr_t)row, xooff, f2{X01, X23});
        } else {
            if (st01) stg<float>((gptr<float>)(uintptr_t)row, xooff, X01);
            if (st23) stg<float>((gptr<float>)(uintptr_t)row, xooff + 4u, X23);
        }
    };

    // DE right-hand side in the state layout: k(X01, X23)
    auto rhs = [&](const float s01, const float s23, const f4 cz, float& k01, float& k23) {
        f4 accA = cz, accB = f4{0.f, 0.f, 0.f, 0.f};
        accA = mfx<0>(s01, w1x[0], accA);  accB = mfx<4>(s01, w1x[1], accB);
        accA = mfx<8>(s01, w1x[2], accA);  accB = mfx<12>(s01, w1x[3], accB);
        accA = mfx<0>(s23, w1x[4], accA);  accB = mfx<4>(s23, w1x[5], accB);
        accA = mfx<8>(s23, w1x[6], accA);  accB = mfx<12>(s23, w1x[7], accB);
        f4 hA = quad_transpose(elu_x<!SAVE>(accA + accB));
        float* slot = ROLES ? ring_mine + (ring_half * 4 + ring_sl) * kXSlotFloats : nullptr;
        if constexpr (ROLES) {                                       // the stage's rows into the ring slot; the partner wave stores them
            *reinterpret_cast<f2*>(slot + 768 + 2 * l) = f2{s01, s23};
            *reinterpret_cast<f4*>(slot + 4 * l) = hA;
        } else if constexpr (SAVE) {                                 // rows (step, stage): the stage input, then the three layers as they appear
            if (pair_ok) {
                if (st01) stg<f2>((gptr<float>)(uintptr_t)sx_run, xooff, f2{s01, s23});
            } else {
                if (st01) stg<float>((gptr<float>)(uintptr_t)sx_run, xooff, s01);
                if (st23) stg<float>((gptr<float>)(uintptr_t)sx_run, xooff + 4u, s23);
            }
            if constexpr (PSNODE_K1X_SAVE_ABL != 2) sx_run += xo_step;
            if (sa_on) stg<f4>((gptr<float>)(uintptr_t)sa_run, saoff, hA);
        }
        hA = hh_layer<!SAVE>(w2, b2, hA);
        if constexpr (ROLES) *reinterpret_cast<f4*>(slot + 256 + 4 * l) = hA;
        else if constexpr (SAVE) { if (sa_on) stg<f4>((gptr<float>)(uintptr_t)(sa_run + sa_layer), saoff, hA); }
        hA = hh_layer<!SAVE>(w3, b3, hA);
        if constexpr (ROLES) {
            *reinterpret_cast<f4*>(slot + 512 + 4 * l) = hA;
            if (++ring_sl == 4) { lds_barrier(); ring_sl = 0; ring_half ^= 1; }       // a half is complete: hand it over, fill the other one
        } else if constexpr (SAVE) {
            if (sa_on) stg<f4>((gptr<float>)(uintptr_t)(sa_run + 2 * sa_layer), saoff, hA);
            if constexpr (PSNODE_K1X_SAVE_ABL != 2) sa_run += sa_stage;
        }
        f4 p0 = b4c[0], p1 = b4c[1];
        p0 = mfn(w4a[0], hA[0], p0); p1 = mfn(w4a[4], hA[0], p1);
        p0 = mfn(w4a[1], hA[1], p0); p1 = mfn(w4a[5], hA[1], p1);
        p0 = mfn(w4a[2], hA[2], p0); p1 = mfn(w4a[6], hA[2], p1);
        p0 = mfn(w4a[3], hA[3], p0); p1 = mfn(w4a[7], hA[3], p1);
        // sum over the 16 blocks: lane bits 5, 4 fold two registers into one each; bits 3, 2 are row rotations
        const float q0 = fold32(p0[0], p1[0]), q1 = fold32(p0[1], p1[1]), q2 = fold32(p0[2], p1[2]), q3 = fold32(p0[3], p1[3]);
        k01 = fold16(q0, q2); k23 = fold16(q1, q3);          // (pairs (0,2) / (1,3): a row ends up with two ADJACENT dims)
        row_sum2(k01, k23);
    };

    // One step.  `tuse` / `euse`: the registers that hold t[k + 1] and the external value of step k; with PF the step re-issues loads INTO
    // them (the clock entry / z row `trun` / `zrun` point at).  WAITN = the vmcnt the step may start at: what was issued, in program order,
    // BEHIND the loads of tuse / euse -- per step two loads and, last, one store (even x_dim; otherwise the count is 0: wait for everything).
    //   one step of look-ahead (general form):  store(k-1)                                  -> vmcnt(1)
    //   two steps (FAST, Euler / Midpoint):     store(k-2), load t, load e, store(k-1)     -> vmcnt(4)
    // A 1 us Euler step does not cover a clock / z row that misses the caches (SQ_WAIT_ANY 28 % of the wave's cycles with one step of
    // look-ahead, profiles/r05g_k1x_euler_pmc_sq.txt); RK4's 3 us step does.
    auto step = [&](const int k, float& tuse, float& euse, auto pf_tag, auto wait_tag) {
        constexpr bool PF = decltype(pf_tag)::value;
        constexpr int WAITN = decltype(wait_tag)::value;
        // SAVE: a step also issued 4 stores per stage (xstage + three layers, every one from at least one lane of every wave) behind its prefetch
        constexpr int NSAVE = (SAVE && !ROLES) ? 4 * (METHOD == PSNODE_EULER ? 1 : (METHOD == PSNODE_MIDPOINT ? 2 : 4)) : 0;
        // ring of R steps of look-ahead (wait tag 100 + R): behind the loads in hand sit the rest of their own step (row store + saved rows) and
        // R - 1 whole steps (2 loads + row store + saved rows each); the legacy tags 1 / 4 are R = 1 / 2.  6 bits: [3:0], [15:14]
        constexpr int RDEPTH = WAITN >= 100 ? WAITN - 100 : (WAITN == 4 ? 2 : 1);
        constexpr int WN = WAITN == 0 ? 0 : (1 + NSAVE) + (RDEPTH - 1) * (3 + NSAVE);
        static_assert(WN <= 63, "vmcnt is 6 bits");
        if (pair_ok && WN > 0) __builtin_amdgcn_s_waitcnt(0x0F70 | (WN & 15) | ((WN >> 4) << 14));
        else __builtin_amdgcn_s_waitcnt(0x0F70);                              // vmcnt(0)
        const float h_ = tuse - t_cur;
        t_cur = tuse;
        const float eA = eon ? (b < ne ? euse - a0e : euse) : 0.0f;
        float s01 = X01, s23 = X23;
        if constexpr (!FAST) {
            if (true_x) {                                            // teacher forcing: the step starts from the dataset's x[k] (uniform branch)
                const float v01 = ldg<float>(as_g(xt_run), xtoff01), v23 = ldg<float>(as_g(xt_run), xtoff23);
                __builtin_amdgcn_s_waitcnt(0x0F70);
                s01 = d01 < xd ? v01 : 0.0f;
                s23 = d23 < xd ? v23 : 0.0f;
            }
            xt_run += xst;
        }
        if constexpr (PF) {   // prefetch: the clock entry and the external row the running bases point at (the caller keeps them inside the grid)
            tuse = ldg<float>(as_g(trun), toff);
            if constexpr (FAST) {
                euse = ldg<float>(as_g(zrun), zoff);
            } else {
                if (((k + 1) & 63) == 0) evb = load_evb((k + 1) >> 6);
                ev_cur = __builtin_amdgcn_readlane(evb, (k + 1) & 63);
                const bool jump = __builtin_amdgcn_readfirstlane(ev_cur) >= 0;
                const float* zr = jump ? zjbase + (long long)ev_cur * zje : zrun;      // an event step takes the jump row (uniform select)
                euse = ldg<float>(as_g(zr), jump ? zjoff : zoff);
            }
            trun += tst;
            zrun += zst;
        }
        store_row(xo_run);                                           // deferred store of the previous step's result, BEHIND the prefetch (above)
        xo_run += xo_step;
        const f4 cz = ext_mfmas<0, 4 * NZM>(w1e, eA, c0);            // per-step constant of L1
        float k1a, k1b;
        rhs(s01, s23, cz, k1a, k1b);
        if constexpr (METHOD == PSNODE_EULER) {
            X01 = s01 + h_ * k1a; X23 = s23 + h_ * k1b;
        } else if constexpr (METHOD == PSNODE_MIDPOINT) {
            const float hh = 0.5f * h_;
            float k2a, k2b;
            rhs(s01 + k1a * hh, s23 + k1b * hh, cz, k2a, k2b);
            X01 = s01 + h_ * k2a; X23 = s23 + h_ * k2b;
        } else {
            float k2a, k2b, k3a, k3b, k4a, k4b;
            rhs(s01 + h_ * k1a * kOneThird, s23 + h_ * k1b * kOneThird, cz, k2a, k2b);
            rhs(s01 + h_ * (k2a - k1a * kOneThird), s23 + h_ * (k2b - k1b * kOneThird), cz, k3a, k3b);
            rhs(s01 + h_ * (k1a - k2a + k3a), s23 + h_ * (k1b - k2b + k3b), cz, k4a, k4b);
            X01 = s01 + (k1a + 3.0f * (k2a + k3a) + k4a) * h_ * 0.125f;
            X23 = s23 + (k1b + 3.0f * (k2b + k3b) + k4b) * h_ * 0.125f;
        }
        if constexpr (SAVE && PSNODE_K1X_SAVE_ABL == 4) sa_run += 3 * (size_t)a.B * hp * kStg - (size_t)768 * kStg;
    };
    using W0 = std::integral_constant<int, 0>;
    using W1 = std::integral_constant<int, 1>;
    using W4 = std::integral_constant<int, 4>;
    constexpr bool TWO_AHEAD = FAST && METHOD != PSNODE_RK4_38;
    // The saving forward: its 4 S stores per step retire IN ORDER in front of the next loads, so with one step of look-ahead a step cannot start
    // before the previous step's rows are acknowledged -- 17 KB in flight per wave, 3.2 TB/s (a plain fill writes this HBM at 6.9 TB/s,
    // profiles/r05ai_hbm_write_bw.txt).  A ring of R steps gives the stores R steps to drain: R = 3 at RK4 (55 operations behind the loads; vmcnt
    // holds 63), 4 at Euler / Midpoint.
    constexpr int RING = (SAVE && !ROLES && FAST) ? (METHOD == PSNODE_RK4_38 ? 3 : (METHOD == PSNODE_MIDPOINT ? 4 : 4)) : 0;
    if constexpr (RING > 0) {
        using WR = std::integral_constant<int, 100 + RING>;
        float tq[RING], eq[RING];                    // slot j: t[k + 1] and the external value of step k, k = j (mod RING)
        tq[0] = t_nxt; eq[0] = e_nxt;
#pragma unroll
        for (int j = 1; j < RING; ++j) {
            const int rt = j + 1 < nT ? j + 1 : nT - 1, rz = j < nT ? j : nT - 1;
            tq[j] = ldg<float>(as_g(a.t.p + (long long)rt * tst), toff);
            eq[j] = ldg<float>(as_g(zbase + (long long)rz * zst), zoff);
        }
        trun = a.t.p + (long long)(RING + 1) * tst;   // the next refill: t[RING + 1], row RING -- requested by step 0
        zrun = zbase + (long long)RING * zst;
        __builtin_amdgcn_s_waitcnt(0x0F70);
        int k = 0;
        for (; k + 2 * RING + 1 <= nT; k += RING) {  // RING steps that all prefetch: the last one requests row k + 2 RING - 1 <= T - 2
#pragma unroll
            for (int j = 0; j < RING; ++j) step(k + j, tq[j], eq[j], std::true_type{}, WR{});
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);           // the last (up to 2 RING) steps: what the ring holds has arrived; refills wait on the spot
        int slot = 0;                                 // k is a multiple of RING here
        for (; k + 1 < nT; ++k) {
            const bool pf = k + RING + 2 <= nT;       // t[k + RING + 1] exists
#pragma unroll
            for (int j = 0; j < RING; ++j) {
                if (slot == j) {
                    if (pf) step(k, tq[j], eq[j], std::true_type{}, W0{});
                    else step(k, tq[j], eq[j], std::false_type{}, W0{});
                }
            }
            slot = slot + 1 == RING ? 0 : slot + 1;
        }
    } else if constexpr (TWO_AHEAD) {
        // ring of two register pairs: (t_nxt, e_nxt) serve the even steps, (t_n2, e_n2) the odd ones; a step reloads the pair it used
        float t_n2 = ldg<float>(as_g(a.t.p + (nT > 2 ? 2 : 1) * tst), toff);            // t[2]
        float e_n2 = ldg<float>(as_g(zbase + zst), zoff);                                 // external row of step 1 (row 1 exists: T >= 2)
        trun = a.t.p + 3 * tst;                                                           // next: t[3], row 2 -- loaded by step 0
        zrun = zbase + 2 * zst;
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
        int k = 0;
        for (; k + 5 <= nT; k += 2) {         // steps k and k + 1 both prefetch: k + 1 <= T - 4
            step(k, t_nxt, e_nxt, std::true_type{}, W4{});
            step(k + 1, t_n2, e_n2, std::true_type{}, W4{});
        }
        if (k + 4 <= nT) {                    // one more step that can still prefetch (t[k + 3] exists); k is even here
            step(k, t_nxt, e_nxt, std::true_type{}, W4{});
            ++k;
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);   // the last (up to two) steps: everything has arrived, nothing more is requested
        for (; k + 1 < nT; ++k) {
            if (k & 1) step(k, t_n2, e_n2, std::false_type{}, W0{});
            else step(k, t_nxt, e_nxt, std::false_type{}, W0{});
        }
    } else {
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the first step's inputs (no store is in flight yet for the loop's vmcnt(1) to skip)
        for (int k = 0; k + 2 < nT; ++k) step(k, t_nxt, e_nxt, std::true_type{}, W1{});
        step(nT - 2, t_nxt, e_nxt, std::false_type{}, W0{});
    }
    store_row(xo_run);
    };      // time_loop
    if (fast_rt) time_loop(std::true_type{});
    else time_loop(std::false_type{});
    if constexpr (ROLES) { if (ring_sl != 0) lds_barrier(); }        // the last, partial half
}

#ifndef PSNODE_K1X_SAVE_ROLES
#define PSNODE_K1X_SAVE_ROLES 0      // 1: the saving forward in a two-role form (a partner wave per SIMD drains an LDS ring of saved rows to memory).
                                     // Measured SLOWER on the same box (profiles/r06_k1x_save_roles_ab.txt: RK4 4.09 vs 4.00 ms, Euler 1.21 vs 1.11):
                                     // the cost of saving is the bytes reaching HBM, not the compute wave's store issue.  Kept as an experiment.
#endif
template <int METHOD, bool SAVE>
hipError_t launch_x_method(const IntegrateDev& a, const float* pack, hipStream_t s) {
    constexpr bool ROLES = SAVE && PSNODE_K1X_SAVE_ROLES;
    const long long tiles = (a.B + 3) / 4;
    const dim3 grid((unsigned)((tiles + kXWaves - 1) / kXWaves)), block(64 * kXWaves * (ROLES ? 2 : 1));
    const size_t lds = ROLES ? (size_t)kXWaves * kXRingFloats * sizeof(float) : 0;
#define PSNODE_X(NZM_)                                                                                                            \
    {                                                                                                                             \
        auto kern = &integrate_x_kernel<METHOD, NZM_, SAVE, ROLES>;                                                               \
        if (lds > 48 * 1024) {                                                                                                    \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return e;                                                                                        \
        }                                                                                                                         \
        hipLaunchKernelGGL(kern, grid, block, lds, s, a, pack);                                                                   \
    }                                                                                                                             \
    break;
    switch ((2 * a.zd + 3) / 4) {
        case 0: PSNODE_X(0)
        case 1: PSNODE_X(1)
        case 2: PSNODE_X(2)
        case 3: PSNODE_X(3)
        case 4: PSNODE_X(4)
        default: return hipErrorNotSupported;
    }
#undef PSNODE_X
    return hipGetLastError();
}

}  // namespace

// shapes K1x takes: the ODE's `3n -> h -> h -> h -> x_dim` with h <= 64 (zero-padded to 64), x_dim <= 8, z_dim <= 8, no saved activations
bool mfma_x_ode_supported(const IntegrateDev& a) {
    const MlpDev& m = a.de;
    if (a.xd < 1 || a.xd > 8 || a.zd < 0 || a.zd > 8 || a.T >= (1ll << 31)) return false;
    if (a.sact && ((a.flags & PSNODE_FLAG_INPUT_TRUE_X) || !a.sxst || ((uintptr_t)a.sact & 15) || ((uintptr_t)a.sxst & 7))) return false;
    if (m.n_layers != 4 || m.in_dim != 3 * (a.xd + a.zd) || m.out_dim[3] != a.xd) return false;
    const int h = m.out_dim[0];
    if (!(h >= 1 && h <= 64 && m.out_dim[1] == h && m.out_dim[2] == h)) return false;
    // 32-bit per-lane byte offsets (span32_ok): every row the time loop addresses that way, once the pointers are known
    if (a.t.p && !span32_ok(a.B, a.t.sb, 1)) return false;
    if (a.zd > 0 && a.z.p && !span32_ok(a.B, a.z.sb, a.zd)) return false;
    if (a.zd > 0 && a.zj && !span32_ok(a.B, a.zjb, a.zd)) return false;
    if ((a.flags & PSNODE_FLAG_INPUT_TRUE_X) && a.x.p && !span32_ok(a.B, a.x.sb, a.xd)) return false;
    if (!span32_ok(a.B, a.xd, a.xd) || (a.sact && !span32_ok(a.B, 64, 64))) return false;      // output rows [B,xd], saved rows [B,Hp]
    return true;
}
// ... and the calls it takes them for: forced (PSNODE_KERNEL_MFMA_WAVE), or up to one wave per SIMD unless the 4-wave tile is forced
bool mfma_x_ode_preferred(const IntegrateDev& a) {
    if (a.kern == PSNODE_KERNEL_MFMA_TILE || a.kern == PSNODE_KERNEL_GENERIC || !mfma_x_ode_supported(a)) return false;
    return a.kern == PSNODE_KERNEL_MFMA_WAVE || a.B <= 4608;
}
size_t mfma_x_pack_floats() { return (size_t)XRegs::COUNT * 64; }

hipError_t launch_mfma_x(const IntegrateDev& a, float* pack, hipStream_t stream) {
    PackX p;
    p.xd = a.xd; p.zd = a.zd; p.n = a.xd + a.zd; p.hreal = a.de.out_dim[0];
    p.w1 = a.de.w[0]; p.b1 = a.de.bias[0]; p.w2 = a.de.w[1]; p.b2 = a.de.bias[1];
    p.w3 = a.de.w[2]; p.b3 = a.de.bias[2]; p.w4 = a.de.w[3]; p.b4 = a.de.bias[3];
    p.out = pack;
    p.sc = a.sact ? 1.0f : kLog2e;
    hipLaunchKernelGGL(pack_x_kernel, dim3(16), dim3(256), 0, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (a.sact) {
        switch (a.method) {
            case PSNODE_EULER: return launch_x_method<PSNODE_EULER, true>(a, pack, stream);
            case PSNODE_MIDPOINT: return launch_x_method<PSNODE_MIDPOINT, true>(a, pack, stream);
            default: return launch_x_method<PSNODE_RK4_38, true>(a, pack, stream);
        }
    }
    switch (a.method) {
        case PSNODE_EULER: return launch_x_method<PSNODE_EULER, false>(a, pack, stream);
        case PSNODE_MIDPOINT: return launch_x_method<PSNODE_MIDPOINT, false>(a, pack, stream);
        default: return launch_x_method<PSNODE_RK4_38, false>(a, pack, stream);
    }
}

}  // namespace psnode
