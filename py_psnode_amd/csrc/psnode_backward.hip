// K4 -- backward (discretise-then-optimise) pass through the fused ODE integrator, K1 shape class
// (DE: 3n -> 64 -> 64 -> 64 -> x_dim, x_dim <= 8, z_dim <= 4).  Replaces what loss.backward() does when it walks the
// unrolled T-step autograd graph of integrate_ODE (neural_00_ODE_01_no_encode.py:358-360 through my_solvers.py:66-78).
//
// Same decomposition as the forward kernel: one workgroup = 4 waves = one tile of 16 trajectories, here walked from the
// last step to the first.  Per step k:
//   phase A  recompute the stage evaluations from the saved xs[k] (stage inputs need k1..k3), keeping every stage's own-unit
//            activations h1,h2,h3 in registers and publishing the TRANSPOSED activations of h1,h2 to LDS;
//   phase B  sweep the stages backwards.  Data path: delta_l = (W_{l+1}^T delta_{l+1}) * ELU'(pre_l) with the transposed
//            weights as MFMA A operands; each wave multiplies its own 16 units (split-K) and the partial sums are
//            reduce-scattered through LDS so that every wave ends up with delta for ITS units.  Weight gradients
//            dW_l += delta_l . h_{l-1}^T contract over the 16 trajectories of the tile: MFMA with A = delta^T, B = h^T (both
//            obtained by a 16x16 in-wave transpose through a padded, conflict-free LDS tile), accumulated in registers
//            over the whole launch and written once, as per-workgroup partials that a second kernel sums in a fixed
//            order (deterministic).  The a0- and (s-a0)-columns of dW1 are reconstructed at the end from the s-columns and
//            sum(delta_1), because a0 is constant over time.
//   RK adjoint: g_k[s] = h b_s g1 + h sum_{s'>s} a_{s's} gx[s'],  gx[s] = (df/dx)(x_s)^T g_k[s],  g0 = g1 + sum_s gx[s].
#define PSNODE_ELU_LITERALS   // register-bound kernels: ELU coefficients as literals, not as 8 resident VGPRs (psnode_common.h)
#include <string.h>

#include "psnode_pack.h"

namespace psnode {
namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

struct BwdDev {
    IntegrateDev a;          // t, z, a0, ev, zj (+strides), T, B, xd, zd, method; a.de.bias unused here
    const float* xs;         // [T,B,xd]
    const float* gout;       // [T,B,xd]
    float* gx0;              // [B,xd]
    float* gz;               // [T,B,zd] or null
    float* gzj;              // [B,nE,zd] or null
    int n_events;
    float* ga0;              // [B,n]
    float* wpart;            // [nWG][NP] per-workgroup parameter-gradient partials
    int NP;
};

// backward registers appended to the forward image
constexpr int BW4T = 0, BW3T = 2, BW2T = 18, BW1T0 = 34, BW1T1 = 38, BWCOUNT = 42;

struct PackBwd {
    PackMfma f;              // forward image description (NA a0 registers included)
    float* out;
};

__global__ void pack_bwd_kernel(const PackBwd pb) {
    const PackMfma& p = pb.f;
    const int RF = pack_fwd_count(p), R = RF + BWCOUNT;
    const int K1 = 3 * p.n;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < NW * R * 64; idx += gridDim.x * blockDim.x) {
        const int lane = idx & 63, reg = (idx >> 6) % R, w = (idx >> 6) / R;
        if (reg < RF) { pb.out[idx] = pack_fwd_value(p, w, reg, lane); continue; }
        const int br = reg - RF, i = lane & 15, g = lane >> 4;
        float v = 0.0f;
        if (br < BW3T) {                       // L4^T: rows = own units i, k-slot g <-> x-dim 4*br+g
            const int d = 4 * br + g;
            if (d < p.xd) v = p.w4[d * HID + 16 * w + i];
        } else if (br < BW1T0) {               // L3^T / L2^T: chunk c <-> output tile of wave (w+c)&3, k-slot g <-> own unit 4g+r
            const bool l3 = br < BW2T;
            const int kk = br - (l3 ? BW3T : BW2T), mt = (w + (kk >> 2)) & 3, r = kk & 3;
            v = (l3 ? p.w3 : p.w2)[(16 * w + 4 * g + r) * HID + 16 * mt + i];
        } else {
            const bool t1 = br >= BW1T1;
            const int r = br - (t1 ? BW1T1 : BW1T0), u = 16 * w + 4 * g + r;
            const int gr = i >> 2, rr = i & 3;
            const float* row = p.w1 + u * K1;
            if (!t1) {                         // tile 0: rows (gr, 0..1) -> gx[d], rows (gr, 2..3) -> ga0x[d]
                const int d = 4 * (rr & 1) + gr;
                if (d < p.xd) v = rr < 2 ? row[2 * p.n + d] + row[p.n + d] : row[d] - row[p.n + d];
            } else {                           // tile 1: row (e, 0) -> gz[e], row (e, 1) -> ga0z[e]
                const int e = gr;
                if (e < p.ne && rr == 0) v = row[p.n + p.xd + e] + row[2 * p.n + p.xd + e];
                if (e < p.ne && rr == 1) v = row[p.xd + e] - row[p.n + p.xd + e];
            }
        }
        pb.out[idx] = v;
    }
}

__device__ __forceinline__ f4 bmfma(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f4 elu4b(f4 v) { return elu_quad(v); }
// ELU'(pre) from h = ELU(pre): 1 for pre > 0 (h > 0), exp(pre) = h + 1 otherwise
__device__ __forceinline__ f4 dact(f4 h) {
    return elu_grad_quad(h);
}
constexpr int SCR = 64 * 4 + 4 * 8;   // padded transpose tile per wave (floats)

template <int METHOD, int NZM>
__global__ __launch_bounds__(256) void ode_backward_kernel(const BwdDev d, const float* __restrict__ pack, const int NA) {
    constexpr int NX = kNXc, S = rk_stages(METHOD);
    using RD = Regs<NX, NX, NZM>;
    const IntegrateDev& a = d.a;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    f4* xbuf = reinterpret_cast<f4*>(lds);                 // [2][4][64]
    f4* rsbuf = xbuf + 2 * NW * 64;                        // [2][4 dest][4 src][64]
    f4* hTb = rsbuf + 2 * NW * NW * 64;                    // [S][2][4][64]
    float* scr_all = reinterpret_cast<float*>(hTb + S * 2 * NW * 64);

    const int l = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = l >> 4, j = l & 15, i = l & 15;
    float* scr = scr_all + w * SCR;
    const long long b0 = (long long)blockIdx.x * TBM;
    const bool valid = b0 + j < a.B;
    const long long b = valid ? b0 + j : a.B - 1;
    const int xd = a.xd, ne = a.zd, n = xd + ne;

    // ---- weights -> registers
    const int RF = RD::COUNT + NA;
    const float* pw = pack + (size_t)w * (RF + BWCOUNT) * 64 + l;
    float w1xs[NX], w1xd[NX], w2[16], w3[16], w4[4], w4t[2], w3t[16], w2t[16], w1t0[4], w1t1[4];
    float w1z[NZM > 0 ? NZM : 1];
    f4 b1r, b2r, b3r, b4r;
#pragma unroll
    for (int r = 0; r < NX; ++r) { w1xs[r] = pw[(RD::W1A + r) * 64]; w1xd[r] = pw[(RD::W1B + r) * 64]; }
#pragma unroll
    for (int m = 0; m < NZM; ++m) w1z[m] = pw[(RD::W1E + m) * 64];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        w2[k] = pw[(RD::W2 + k) * 64]; w3[k] = pw[(RD::W3 + k) * 64];
        w3t[k] = pw[(RF + BW3T + k) * 64]; w2t[k] = pw[(RF + BW2T + k) * 64];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        w4[r] = pw[(RD::W4 + r) * 64];
        b1r[r] = pw[(RD::B1 + r) * 64]; b2r[r] = pw[(RD::B2 + r) * 64]; b3r[r] = pw[(RD::B3 + r) * 64]; b4r[r] = pw[(RD::B4 + r) * 64];
        w1t0[r] = pw[(RF + BW1T0 + r) * 64]; w1t1[r] = pw[(RF + BW1T1 + r) * 64];
    }
    w4t[0] = pw[(RF + BW4T) * 64]; w4t[1] = pw[(RF + BW4T + 1) * 64];

    // ---- per-trajectory constants
    float a0x[NX];
#pragma unroll
    for (int r = 0; r < NX; ++r) a0x[r] = 4 * r + g < xd ? a.a0[b * n + 4 * r + g] : 0.0f;
    int eidx[NZM > 0 ? NZM : 1];
    float a0e[NZM > 0 ? NZM : 1];
#pragma unroll
    for (int m = 0; m < NZM; ++m) {
        const int q = 4 * m + g, e = slot_ext(q, ne);
        eidx[m] = e < 0 ? 0 : e;
        a0e[m] = q < ne ? a.a0[b * n + xd + q] : 0.0f;
    }
    f4 c0 = b1r;
    for (int m = 0; m < NA; ++m) {
        const int q = 4 * m + g;
        c0 = bmfma(pw[(RD::COUNT + m) * 64], q < n ? a.a0[b * n + q] : 0.0f, c0);
    }
    // row of the padded transpose tile that holds column i of the `s` vector (x dims, then z dims), or -1
    const int srow = i < xd ? 4 * (i & 3) + (i >> 2) : (i < n ? 4 * (i - xd) + 2 : -1);

    const long long tst = a.t.st, nT = a.T, zst = a.z.st, zje = a.zje;
    const float* tp = a.t.p + b * a.t.sb;
    const float* zp = a.z.p + b * a.z.sb;
    const float* zjp = a.zj + b * a.zjb;

    // D-layout tile (rows 4g+r, col j) of this wave -> o[kk] = M[row][col 4kk+g]   (A/B operand layout), via LDS
    auto put_tile = [&](const f4 v) { *reinterpret_cast<f4*>(scr + 4 * l + 8 * g) = v; };
    auto get_row = [&](const int row) -> f4 {
        const float* s = scr + 4 * (16 * (row >> 2) + g) + 8 * (row >> 2) + (row & 3);
        return f4{s[0], s[16], s[32], s[48]};
    };
    auto transpose = [&](const f4 v) -> f4 { put_tile(v); return get_row(i); };

    int p = 0, q = 0;   // parities of the all-gather/all-reduce buffer and of the reduce-scatter buffer
    // all-gather of own activations h (D layout) + 64->64 layer, as in the forward kernel
    auto mid = [&](const float (&wm)[16], const f4 bias, const f4 h) -> f4 {
        xbuf[(p * NW + w) * 64 + l] = h;
        f4 accA = bias, accB = f4{0.f, 0.f, 0.f, 0.f};
        // own-quarter MFMAs: PSNODE_BWD_MID_PRE of them pinned in front of the barrier, the rest behind the read issue (K1's finding:
        // psnode_mfma_impl.h `mid`; -1 = leave the placement to the compiler, as rounds 1-2 did).  Same-box A/B of the training step
        // (profiles/r03m_bwd_ab.txt): unpinned 17.42 ms, 2 in front 17.17, 3 in front 17.25.
#ifndef PSNODE_BWD_MID_PRE
#define PSNODE_BWD_MID_PRE 2
#endif
        constexpr int PRE = PSNODE_BWD_MID_PRE < 0 ? 4 : PSNODE_BWD_MID_PRE;
        constexpr bool PIN = PSNODE_BWD_MID_PRE >= 0;
        if constexpr (PRE >= 1) accA = bmfma(wm[0], h[0], accA);
        if constexpr (PRE >= 2) accB = bmfma(wm[1], h[1], accB);
        if constexpr (PRE >= 3) accA = bmfma(wm[2], h[2], accA);
        if constexpr (PRE >= 4) accB = bmfma(wm[3], h[3], accB);
        if constexpr (PIN && PRE >= 1) asm volatile("" : "+v"(accA), "+v"(accB));
        lds_barrier();
        f4 vq[4];      // all three reads in flight before the first dependent MFMA (K1: -5 % launch time)
#pragma unroll
        for (int c = 1; c < 4; ++c) vq[c] = xbuf[(p * NW + ((w + c) & 3)) * 64 + l];
        if constexpr (PIN && PRE < 4) asm volatile("" : "+v"(accA), "+v"(accB));
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PRE < 1) accA = bmfma(wm[0], h[0], accA);
        if constexpr (PRE < 2) accB = bmfma(wm[1], h[1], accB);
        if constexpr (PRE < 3) accA = bmfma(wm[2], h[2], accA);
        if constexpr (PRE < 4) accB = bmfma(wm[3], h[3], accB);
#pragma unroll
        for (int c = 1; c < 4; ++c) {
            const f4 v = vq[c];
            accA = bmfma(wm[4 * c + 0], v[0], accA); accB = bmfma(wm[4 * c + 1], v[1], accB);
            accA = bmfma(wm[4 * c + 2], v[2], accA); accB = bmfma(wm[4 * c + 3], v[3], accB);
        }
        p ^= 1;
        return accA + accB;
    };
    // all-reduce of rows r < 2 over the four waves (fixed order)
    auto allreduce2 = [&](const f4 part, const f4 init) -> f2 {
        f2* xb2 = reinterpret_cast<f2*>(xbuf + p * NW * 64);
        xb2[w * 64 + l] = f2{part[0], part[1]};
        lds_barrier();
        f2 out = f2{init[0], init[1]};
#pragma unroll
        for (int c = 0; c < 4; ++c) out += xb2[c * 64 + l];
        p ^= 1;
        return out;
    };
    // reduce-scatter: part[c] is this wave's contribution to the tile of wave (w+c)&3; returns the full sum of the own tile
    auto reduce_scatter = [&](const f4 (&part)[4]) -> f4 {
#pragma unroll
        for (int c = 1; c < 4; ++c) rsbuf[((q * NW + ((w + c) & 3)) * NW + w) * 64 + l] = part[c];
        lds_barrier();
        f4 out = part[0];
#pragma unroll
        for (int c = 1; c < 4; ++c) out += rsbuf[((q * NW + w) * NW + ((w + c) & 3)) * 64 + l];
        q ^= 1;
        return out;
    };
    // partial products of a transposed 64x64 layer from this wave's 16 units
    auto layer_T = [&](const float (&wt)[16], const f4 dl, f4 (&part)[4]) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f4 acc = bmfma(wt[4 * c], dl[0], f4{0.f, 0.f, 0.f, 0.f});
            acc = bmfma(wt[4 * c + 1], dl[1], acc);
            acc = bmfma(wt[4 * c + 2], dl[2], acc);
            part[c] = bmfma(wt[4 * c + 3], dl[3], acc);
        }
    };

    // ---- accumulators (whole launch)
    f4 accW4 = {0.f, 0.f, 0.f, 0.f}, accW1s = accW4, S1 = accW4, S2 = accW4, S3 = accW4;
    f4 accW3[4], accW2[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { accW3[c] = accW4; accW2[c] = accW4; }
    f2 db4 = {0.f, 0.f}, ga0x = {0.f, 0.f};
    float ga0z = 0.f;
    f2 gcarry = {0.f, 0.f};

    // inputs of a step, prefetched one iteration ahead (the sweep runs k = T-2 .. 0).  Every load is UNCONDITIONAL (clamped row / column
    // indices, masks applied where the value is consumed a step later): a load under a predicate or inside `if (k >= 1)` is a phi with a
    // constant, the copy into the loop-carried register sits right behind the load, and the wait for it -- s_waitcnt vmcnt(0) on the whole
    // prefetch just issued, a full HBM round trip -- sat in every step (round 3, found in the ISA: 14.3 -> see DESIGN.md).
    auto load_ext = [&](long long k, int ev, float (&dst)[NZM > 0 ? NZM : 1]) {
        const float* src = ev >= 0 ? zjp + ev * zje : zp + k * zst;
#pragma unroll
        for (int m = 0; m < NZM; ++m) dst[m] = src[eidx[m]];     // padding slots read column 0 against a zero weight
    };
    int xcol[NX];
#pragma unroll
    for (int r = 0; r < NX; ++r) xcol[r] = 4 * r + g < xd ? 4 * r + g : 0;
    auto load_state = [&](long long k, float (&xk)[NX], float (&gk1)[NX]) {   // xs[k] and dL/dxs[k+1], raw (dims >= x_dim repeat dim 0)
#pragma unroll
        for (int r = 0; r < NX; ++r) {
            xk[r] = d.xs[(k * a.B + b) * xd + xcol[r]];
            gk1[r] = d.gout[((k + 1) * a.B + b) * xd + xcol[r]];
        }
    };
    int lane_zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));
    const bool has_ev = a.ev != nullptr;
    const int* evp = (has_ev ? a.ev : reinterpret_cast<const int*>(a.t.p)) + lane_zero;   // per-lane load: the value stays in a VGPR until it is used
    float t_hi = nT >= 2 ? tp[(nT - 1) * tst] : 0.0f, t_lo = nT >= 2 ? tp[(nT - 2) * tst] : 0.0f;
    int ev_cur = (has_ev && nT >= 2) ? a.ev[nT - 2] : -1;
    int ev_raw = evp[nT >= 3 ? nT - 3 : 0];               // raw table entry of the step after; masked with has_ev when it becomes ev_cur
    float ext_n[NZM > 0 ? NZM : 1] = {}, x_n[NX] = {}, g_n[NX] = {};
    if (nT >= 2) { load_ext(nT - 2, ev_cur, ext_n); load_state(nT - 2, x_n, g_n); }
    bool gon[NX];
#pragma unroll
    for (int r = 0; r < NX; ++r) gon[r] = valid && 4 * r + g < xd;

    for (long long k = nT - 2; k >= 0; --k) {
        // ---- inputs of step k (already in registers); issue the loads of step k-1
        const float h_ = t_hi - t_lo;
        const int ev = ev_cur;
        float extv[NZM > 0 ? NZM : 1], x0[NX];
        f2 g1 = gcarry;
#pragma unroll
        for (int m = 0; m < (NZM > 0 ? NZM : 1); ++m) extv[m] = ext_n[m];
#pragma unroll
        for (int r = 0; r < NX; ++r) { x0[r] = x_n[r]; g1[r] += gon[r] ? g_n[r] : 0.0f; }
        {
            const long long kp = k >= 1 ? k - 1 : 0;
            t_hi = t_lo;
            t_lo = tp[kp * tst];
            ev_cur = has_ev ? ev_raw : -1;
            load_ext(kp, ev_cur, ext_n);
            load_state(kp, x_n, g_n);
            ev_raw = evp[k >= 2 ? k - 2 : 0];
        }
        f4 cz = c0;
#pragma unroll
        for (int m = 0; m < NZM; ++m) cz = bmfma(w1z[m], extv[m] - a0e[m], cz);

        // ---- phase A: stage evaluations, activations kept
        f4 h1[S], h2[S], h3[S];
        f2 ks[S];
        float xst[S][NX];
#pragma unroll
        for (int s = 0; s < S; ++s) {
#pragma unroll
            for (int r = 0; r < NX; ++r) {
                float acc = 0.0f;
#pragma unroll
                for (int jj = 0; jj < s; ++jj) acc += rk_a(METHOD, s, jj) * ks[jj][r];
                xst[s][r] = s == 0 ? x0[r] : x0[r] + h_ * acc;
            }
            f4 accA = cz, accB = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < NX; ++r) {
                accA = bmfma(w1xs[r], xst[s][r], accA);
                accB = bmfma(w1xd[r], xst[s][r] - a0x[r], accB);
            }
            h1[s] = elu4b(accA + accB);
            hTb[((s * 2 + 0) * NW + w) * 64 + l] = transpose(h1[s]);
            h2[s] = elu4b(mid(w2, b2r, h1[s]));
            hTb[((s * 2 + 1) * NW + w) * 64 + l] = transpose(h2[s]);
            h3[s] = elu4b(mid(w3, b3r, h2[s]));
            f4 pa = bmfma(w4[0], h3[s][0], f4{0.f, 0.f, 0.f, 0.f}), pb = bmfma(w4[1], h3[s][1], f4{0.f, 0.f, 0.f, 0.f});
            pa = bmfma(w4[2], h3[s][2], pa);
            pb = bmfma(w4[3], h3[s][3], pb);
            ks[s] = allreduce2(pa + pb, b4r);
        }

        // ---- phase B: stages backwards
        f2 gks[S], gx0 = g1;
#pragma unroll
        for (int s = 0; s < S; ++s) gks[s] = (h_ * rk_b(METHOD, s)) * g1;
        f4 l1t1 = {0.f, 0.f, 0.f, 0.f};   // z rows of W1^T delta1, this wave's partial, summed over the stages
#pragma unroll
        for (int s = S - 1; s >= 0; --s) {
            const f2 gk = gks[s];
            db4 += gk;
            // delta3 = (W4^T gk) * ELU'(pre3)
            f4 t3 = bmfma(w4t[0], gk[0], f4{0.f, 0.f, 0.f, 0.f});
            t3 = bmfma(w4t[1], gk[1], t3);
            const f4 d3 = t3 * dact(h3[s]);
            S3 += d3;
            // dW4 += gk^T (x) h3^T
            {
                const f4 gT = transpose(f4{gk[0], gk[1], 0.f, 0.f});
                const f4 hT = transpose(h3[s]);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) accW4 = bmfma(gT[kk], hT[kk], accW4);
            }
            // delta2, dW3
            f4 part[4];
            layer_T(w3t, d3, part);
            const f4 d2 = reduce_scatter(part) * dact(h2[s]);
            S2 += d2;
            {
                const f4 dT = transpose(d3);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f4 hT = hTb[((s * 2 + 1) * NW + ((w + c) & 3)) * 64 + l];
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) accW3[c] = bmfma(dT[kk], hT[kk], accW3[c]);
                }
            }
            // delta1, dW2
            layer_T(w2t, d2, part);
            const f4 d1 = reduce_scatter(part) * dact(h1[s]);
            S1 += d1;
            {
                const f4 dT = transpose(d2);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f4 hT = hTb[((s * 2 + 0) * NW + ((w + c) & 3)) * 64 + l];
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) accW2[c] = bmfma(dT[kk], hT[kk], accW2[c]);
                }
            }
            // W1^T delta1: tile 0 rows 0..1 -> gx (all-reduced), rows 2..3 -> ga0x partial; tile 1 -> gz / ga0z partials
            f4 t0 = bmfma(w1t0[0], d1[0], f4{0.f, 0.f, 0.f, 0.f});
            t0 = bmfma(w1t0[1], d1[1], t0); t0 = bmfma(w1t0[2], d1[2], t0); t0 = bmfma(w1t0[3], d1[3], t0);
            if (NZM > 0) {
                l1t1 = bmfma(w1t1[0], d1[0], l1t1); l1t1 = bmfma(w1t1[1], d1[1], l1t1);
                l1t1 = bmfma(w1t1[2], d1[2], l1t1); l1t1 = bmfma(w1t1[3], d1[3], l1t1);
            }
            ga0x += f2{t0[2], t0[3]};
            const f2 gx = allreduce2(t0, f4{0.f, 0.f, 0.f, 0.f});
            // dW1 (`s` columns) += delta1^T (x) s^T
            {
                const f4 dT = transpose(d1);
                put_tile(f4{xst[s][0], xst[s][1], g < ne ? extv[0] : 0.0f, 0.0f});
                const f4 sT = srow >= 0 ? get_row(srow) : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) accW1s = bmfma(dT[kk], sT[kk], accW1s);
            }
            // RK adjoint
            gx0 += gx;
#pragma unroll
            for (int jj = 0; jj < s; ++jj) gks[jj] += (h_ * rk_a(METHOD, s, jj)) * gx;
        }
        gcarry = gx0;
        // ---- gradient of this step's external input (sum over waves), a0-z accumulation
        if (NZM > 0) {
            ga0z += l1t1[1];
            const f2 gzr = allreduce2(l1t1, f4{0.f, 0.f, 0.f, 0.f});   // row 0: gz[e = g]
            if (w == 0 && valid && g < ne) {
                if (ev >= 0) { if (d.gzj) d.gzj[(b * d.n_events + ev) * ne + g] = gzr[0]; }
                if (d.gz) d.gz[(k * a.B + b) * ne + g] = ev >= 0 ? 0.0f : gzr[0];
            }
        }
    }

    // ---- epilogue
    if (w == 0 && valid) {
#pragma unroll
        for (int r = 0; r < NX; ++r)
            if (4 * r + g < xd) d.gx0[b * xd + 4 * r + g] = gcarry[r] + d.gout[b * xd + 4 * r + g];
        if (d.gz && g < ne && nT >= 1) d.gz[((nT - 1) * a.B + b) * ne + g] = 0.0f;   // z[T-1] is never read by the ODE loop
    }
    // d all_initial: x dims from ga0x (rows 2..3 of tile 0), z dims from ga0z; sum over waves
    {
        const f2 ax = allreduce2(f4{ga0x[0], ga0x[1], 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f});
        const f2 az = allreduce2(f4{ga0z, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f});
        if (w == 0 && valid) {
#pragma unroll
            for (int r = 0; r < NX; ++r) if (4 * r + g < xd) d.ga0[b * n + 4 * r + g] = ax[r];
            if (g < ne) d.ga0[b * n + xd + g] = az[0];
        }
    }
    // parameter-gradient partials of this workgroup
    float* wp = d.wpart + (size_t)blockIdx.x * d.NP;
    const int K1 = 3 * n;
    const int oB1 = HID * K1, oW2 = oB1 + HID, oB2 = oW2 + HID * HID, oW3 = oB2 + HID, oB3 = oW3 + HID * HID, oW4 = oB3 + HID, oB4 = oW4 + xd * HID;
    {
        // dW1: columns [a0 | s-a0 | s]; ca0 = sum(delta1)^T (x) a0^T
        const f4 sT = transpose(S1);
        f4 ca0 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const long long tb = b0 + 4 * kk + g;
            const float av = (i < n && tb < a.B) ? a.a0[tb * n + i] : 0.0f;
            ca0 = bmfma(sT[kk], av, ca0);
        }
        if (j < n) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* row = wp + (16 * w + 4 * g + r) * K1;
                row[j] = ca0[r];
                row[n + j] = accW1s[r] - ca0[r];
                row[2 * n + j] = accW1s[r];
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int nt = (w + c) & 3;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            wp[oW2 + (16 * w + 4 * g + r) * HID + 16 * nt + j] = accW2[c][r];
            wp[oW3 + (16 * w + 4 * g + r) * HID + 16 * nt + j] = accW3[c][r];
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {   // dW4 rows (g, r) <-> x-dim 4r+g, columns = own units
        const int dd = 4 * r + g;
        if (dd < xd) wp[oW4 + dd * HID + 16 * w + j] = accW4[r];
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
            wp[oB1 + 16 * w + 4 * g + r] = sb1[r]; wp[oB2 + 16 * w + 4 * g + r] = sb2[r]; wp[oB3 + 16 * w + 4 * g + r] = sb3[r];
        }
        if (w == 0) {
#pragma unroll
            for (int r = 0; r < NX; ++r) if (4 * r + g < xd) wp[oB4 + 4 * r + g] = sb4[r];
        }
    }
}

int bwd_np(int xd, int zd) {
    const int n = xd + zd;
    return HID * 3 * n + HID + 2 * (HID * HID + HID) + xd * HID + xd;
}
int bwd_nzm(int zd) { return (2 * zd + 3) / 4; }

template <int METHOD>
hipError_t launch_bwd(const BwdDev& d, const float* pack, int NA, int NZM, size_t lds, hipStream_t s) {
    const dim3 grid((unsigned)((d.a.B + TBM - 1) / TBM)), block(256);
#define PSNODE_BWD(NZM_)                                                                                                         \
    {                                                                                                                            \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ode_backward_kernel<METHOD, NZM_>),                   \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                               \
        if (e != hipSuccess) return e;                                                                                          \
        hipLaunchKernelGGL((ode_backward_kernel<METHOD, NZM_>), grid, block, lds, s, d, pack, NA);                              \
        return hipGetLastError();                                                                                               \
    }
    switch (NZM) {
        case 0: PSNODE_BWD(0)
        case 1: PSNODE_BWD(1)
        case 2: PSNODE_BWD(2)
        default: return hipErrorNotSupported;
    }
#undef PSNODE_BWD
}

bool bwd_shape_ok(const psnode_ode_bwd_args_f32* a) {
    const psnode_mlp_f32& m = a->de;
    if (a->x_dim < 1 || a->x_dim > 4 * kNXc || a->z_dim < 0 || a->z_dim > 4) return false;
    return m.n_layers == 4 && m.in_dim == 3 * (a->x_dim + a->z_dim) && m.out_dim[0] == HID && m.out_dim[1] == HID &&
           m.out_dim[2] == HID && m.out_dim[3] == a->x_dim;
}

}  // namespace
}  // namespace psnode

using namespace psnode;

namespace {
int generic_np(const psnode_mlp_f32& m) {
    int np = 0, k = m.in_dim;
    for (int l = 0; l < m.n_layers; ++l) { np += m.out_dim[l] * (k + 1); k = m.out_dim[l]; }
    return np;
}
bool ode_generic_ok(const psnode_ode_bwd_args_f32* a) {
    const psnode_mlp_f32& m = a->de;
    if (a->x_dim < 1 || a->z_dim < 0 || m.n_layers < 1 || m.n_layers > kMaxLayers) return false;
    if (m.in_dim != 3 * (a->x_dim + a->z_dim) || m.out_dim[m.n_layers - 1] != a->x_dim) return false;
    return generic_bwd_fits(&a->de, nullptr, a->x_dim, a->z_dim, 0, 0) != 0;
}
bool use_mfma_bwd(const psnode_ode_bwd_args_f32* a) {     // (saved activations are K4f's: K4 always recomputes)
    return a->kernel != PSNODE_KERNEL_GENERIC && a->kernel != PSNODE_KERNEL_MFMA_WIDE && !a->saved_act && bwd_shape_ok(a);
}
// K4f: every other width <= 128 (and z_dim up to 8); at hidden 64 exactly the specialised K4 is faster (14.3 vs ~16 ms) and keeps AUTO
bool use_fused_bwd(const psnode_ode_bwd_args_f32* a) { return a->kernel != PSNODE_KERNEL_GENERIC && !use_mfma_bwd(a) && fused_bwd_shape_ok(a); }
// K8 needs 16-byte aligned rows; with pointers not yet known (dims-only queries) the shape decides
bool use_latent_bwd(const psnode_ode_bwd_args_f32* a) {
    return a->kernel != PSNODE_KERNEL_GENERIC && latent_bwd_shape_ok(a) && (!a->xs || latent_bwd_ptrs_ok(a));
}
bool use_latent64_bwd(const psnode_ode_bwd_args_f32* a) {   // K9: the only fused backward for this shape (K5's LDS budget is too small)
    return a->kernel != PSNODE_KERNEL_GENERIC && latent64_ode_bwd_shape_ok(a) && (!a->xs || latent64_ode_bwd_ptrs_ok(a));
}
}  // namespace

extern "C" int32_t psnode_ode_backward_supported(const psnode_ode_bwd_args_f32* a) {
    if (!a || a->method < PSNODE_EULER || a->method > PSNODE_RK4_38) return 0;
    if (a->kernel == PSNODE_KERNEL_MFMA_WIDE) return fused_bwd_shape_ok(a);
    if (a->kernel == PSNODE_KERNEL_MFMA) return bwd_shape_ok(a) || fused_bwd_shape_ok(a) || use_latent_bwd(a) || use_latent64_bwd(a);
    return use_mfma_bwd(a) || use_fused_bwd(a) || use_latent_bwd(a) || use_latent64_bwd(a) || ode_generic_ok(a);
}

extern "C" int64_t psnode_ode_backward_param_count(const psnode_ode_bwd_args_f32* a) {
    return a && a->de.n_layers >= 1 && a->de.n_layers <= kMaxLayers ? generic_np(a->de) : 0;
}

extern "C" size_t psnode_ode_backward_workspace_bytes(const psnode_ode_bwd_args_f32* a) {
    if (!a || !psnode_ode_backward_supported(a)) return 0;
    size_t floats = ode_generic_ok(a) ? generic_bwd_workspace_floats(&a->de, nullptr, a->B) : 0;
    if (latent_bwd_shape_ok(a)) floats = latent_bwd_workspace_floats(a->B) > floats ? latent_bwd_workspace_floats(a->B) : floats;
    if (latent64_ode_bwd_shape_ok(a)) floats = latent64_ode_bwd_workspace_floats(a->B) > floats ? latent64_ode_bwd_workspace_floats(a->B) : floats;
    if (fused_bwd_shape_ok(a)) { const size_t f3 = fused_bwd_workspace_floats(a); floats = f3 > floats ? f3 : floats; }
    if (bwd_shape_ok(a)) {
        const int n = a->x_dim + a->z_dim;
        const size_t pack = (size_t)NW * (kMaxRegs + (n + 3) / 4 + BWCOUNT) * 64;
        const size_t nwg = (size_t)((a->B + TBM - 1) / TBM);
        const size_t f2 = pack + nwg * bwd_np(a->x_dim, a->z_dim) + 64;
        floats = f2 > floats ? f2 : floats;
    }
    return floats * sizeof(float);
}

extern "C" int32_t psnode_ode_backward_f32(const psnode_ode_bwd_args_f32* a, void* workspace, size_t workspace_bytes, void* stream) {
    if (!a) return PSNODE_ERR_NULL;
    if (a->method < PSNODE_EULER || a->method > PSNODE_RK4_38) return PSNODE_ERR_METHOD;
    if (a->T < 1 || a->B < 1) return PSNODE_ERR_DIMS;
    if (!psnode_ode_backward_supported(a)) return PSNODE_ERR_UNSUPPORTED;
    for (int l = 0; l < a->de.n_layers; ++l) if (!a->de.weight[l] || !a->de.bias[l]) return PSNODE_ERR_NULL;
    if (!a->t.ptr || !a->all_initial || !a->xs || !a->grad_xs || !a->grad_x0 || !a->grad_all_initial || !a->grad_params) return PSNODE_ERR_NULL;
    if (a->z_dim > 0 && !a->z.ptr) return PSNODE_ERR_NULL;
    if (a->event_idx && a->z_dim > 0 && !a->z_jump) return PSNODE_ERR_NULL;
    if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 255u) || workspace_bytes < psnode_ode_backward_workspace_bytes(a))
        return PSNODE_ERR_WORKSPACE;
    if ((a->saved_act != nullptr) != (a->saved_xstage != nullptr)) return PSNODE_ERR_NULL;
    if (a->saved_act && !use_fused_bwd(a) && !use_latent64_bwd(a)) return PSNODE_ERR_UNSUPPORTED;      // only K4f and K9 read them
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (a->flags & ~PSNODE_FLAG_INPUT_TRUE_X) return PSNODE_ERR_UNSUPPORTED;
    if (a->flags & PSNODE_FLAG_INPUT_TRUE_X) {      // teacher-forced backward: K4f (recompute form) is the kernel that has it
        if (a->kernel == PSNODE_KERNEL_GENERIC || a->saved_act || !fused_bwd_shape_ok(a)) return PSNODE_ERR_UNSUPPORTED;
        return fused_bwd_launch(a, static_cast<float*>(workspace), s);
    }
    if (use_latent_bwd(a)) return latent_bwd_launch(a, static_cast<float*>(workspace), s);
    if (use_latent64_bwd(a)) return latent64_ode_bwd_launch(a, static_cast<float*>(workspace), s);
    if (use_fused_bwd(a)) return fused_bwd_launch(a, static_cast<float*>(workspace), s);
    if (a->kernel == PSNODE_KERNEL_MFMA && !bwd_shape_ok(a)) return PSNODE_ERR_UNSUPPORTED;   // latent shape, unaligned views
    if (!use_mfma_bwd(a)) {
        return generic_backward_launch(a->method, a->x_dim, a->z_dim, 0, 0, a->T, a->B, &a->de, nullptr,
                                       ViewDev{a->t.ptr, a->t.stride_t, a->t.stride_b}, ViewDev{a->z.ptr, a->z.stride_t, a->z.stride_b},
                                       ViewDev{nullptr, 0, 0}, a->all_initial, a->event_idx, a->z_jump, a->zj_stride_b, a->zj_stride_e, nullptr,
                                       0, 0, a->n_events, a->xs, nullptr, a->grad_xs, nullptr, a->grad_x0, a->grad_z, nullptr, a->grad_z_jump,
                                       nullptr, a->grad_all_initial, a->grad_params, nullptr, static_cast<float*>(workspace), s);
    }
    const int xd = a->x_dim, zd = a->z_dim, n = xd + zd, NZM = bwd_nzm(zd), NA = (n + 3) / 4;
    float* pack = static_cast<float*>(workspace);
    float* wpart = pack + (size_t)NW * (kMaxRegs + NA + BWCOUNT) * 64;
    BwdDev d;
    memset(&d, 0, sizeof(d));
    d.a.method = a->method; d.a.xd = xd; d.a.zd = zd; d.a.T = a->T; d.a.B = a->B;
    d.a.t = ViewDev{a->t.ptr, a->t.stride_t, a->t.stride_b};
    d.a.z = ViewDev{a->z.ptr, a->z.stride_t, a->z.stride_b};
    d.a.a0 = a->all_initial; d.a.ev = a->event_idx; d.a.zj = a->z_jump; d.a.zjb = a->zj_stride_b; d.a.zje = a->zj_stride_e;
    d.xs = a->xs; d.gout = a->grad_xs; d.gx0 = a->grad_x0; d.gz = a->grad_z; d.gzj = a->grad_z_jump; d.n_events = a->n_events;
    d.ga0 = a->grad_all_initial; d.wpart = wpart; d.NP = bwd_np(xd, zd);
    PackBwd pb;
    pb.f.fold = 0; pb.f.hreal = HID; pb.f.ae = 0; pb.f.nw = NW; pb.f.xd = xd; pb.f.ne = zd; pb.f.n = n; pb.f.nzv = zd; pb.f.NX = kNXc; pb.f.NB = kNXc; pb.f.NE = NZM; pb.f.NA = NA;
    pb.f.w1 = a->de.weight[0]; pb.f.b1 = a->de.bias[0]; pb.f.w2 = a->de.weight[1]; pb.f.b2 = a->de.bias[1];
    pb.f.w3 = a->de.weight[2]; pb.f.b3 = a->de.bias[2]; pb.f.w4 = a->de.weight[3]; pb.f.b4 = a->de.bias[3];
    pb.f.out_dim = xd; pb.f.out = nullptr;
    pb.out = pack;
    hipLaunchKernelGGL(pack_bwd_kernel, dim3(32), dim3(256), 0, s, pb);
    if (hipGetLastError() != hipSuccess) return PSNODE_ERR_HIP;
    const int S = a->method == PSNODE_EULER ? 1 : (a->method == PSNODE_MIDPOINT ? 2 : 4);
    const size_t lds = (size_t)(2 * NW * 64 + 2 * NW * NW * 64 + S * 2 * NW * 64) * sizeof(f4) + (size_t)NW * SCR * sizeof(float);
    hipError_t e;
    switch (a->method) {
        case PSNODE_EULER: e = launch_bwd<PSNODE_EULER>(d, pack, NA, NZM, lds, s); break;
        case PSNODE_MIDPOINT: e = launch_bwd<PSNODE_MIDPOINT>(d, pack, NA, NZM, lds, s); break;
        default: e = launch_bwd<PSNODE_RK4_38>(d, pack, NA, NZM, lds, s); break;
    }
    if (e != hipSuccess) return e == hipErrorNotSupported ? PSNODE_ERR_UNSUPPORTED : PSNODE_ERR_HIP;
    const int nwg = (int)((a->B + TBM - 1) / TBM);
    return launch_reduce_partials(wpart, a->grad_params, nullptr, d.NP, 0, nwg, s) == hipSuccess ? PSNODE_OK : PSNODE_ERR_HIP;
}

namespace {
bool use_mfma_dae_bwd(const psnode_dae_bwd_args_f32* a) { return a->kernel != PSNODE_KERNEL_GENERIC && dae_mfma_bwd_shape_ok(a); }
bool use_latent16_dae_bwd(const psnode_dae_bwd_args_f32* a) {
    return a->kernel != PSNODE_KERNEL_GENERIC && latent16_dae_bwd_shape_ok(a) && (!a->xs || latent16_dae_bwd_ptrs_ok(a));
}
bool use_latent64_dae_bwd(const psnode_dae_bwd_args_f32* a) {
    return a->kernel != PSNODE_KERNEL_GENERIC && latent64_dae_bwd_shape_ok(a) && (!a->xs || latent64_dae_bwd_ptrs_ok(a));
}
}  // namespace

extern "C" int32_t psnode_dae_backward_supported(const psnode_dae_bwd_args_f32* a) {
    if (!a || a->method < PSNODE_EULER || a->method > PSNODE_RK4_38) return 0;
    if (a->x_dim < 1 || a->z_dim < 0 || a->v_dim < 0 || a->i_dim < 1) return 0;
    if (a->kernel == PSNODE_KERNEL_MFMA) return dae_mfma_bwd_shape_ok(a) || use_latent64_dae_bwd(a) || use_latent16_dae_bwd(a);
    if (use_mfma_dae_bwd(a) || use_latent64_dae_bwd(a) || use_latent16_dae_bwd(a)) return 1;
    const int n = a->x_dim + a->z_dim + a->v_dim + a->i_dim;
    const psnode_mlp_f32 &d = a->de, &g = a->ae;
    if (d.n_layers < 1 || d.n_layers > kMaxLayers || g.n_layers < 1 || g.n_layers > kMaxLayers) return 0;
    if (d.in_dim != 3 * n || d.out_dim[d.n_layers - 1] != a->x_dim) return 0;
    if (g.in_dim != n + a->x_dim + a->z_dim + a->v_dim || g.out_dim[g.n_layers - 1] != a->i_dim) return 0;
    return generic_bwd_fits(&a->de, &a->ae, a->x_dim, a->z_dim, a->v_dim, a->i_dim);
}

extern "C" size_t psnode_dae_backward_workspace_bytes(const psnode_dae_bwd_args_f32* a) {
    if (!a || !psnode_dae_backward_supported(a)) return 0;
    if (use_mfma_dae_bwd(a)) return dae_mfma_bwd_workspace_floats(a) * sizeof(float);
    if (latent64_dae_bwd_shape_ok(a) && a->kernel != PSNODE_KERNEL_GENERIC) {
        // sized for every kernel this launch can end up on: K9, or K5 when the pointers turn out unaligned and K5 fits the shape
        size_t f = latent64_dae_bwd_workspace_floats(a);
        if (generic_bwd_fits(&a->de, &a->ae, a->x_dim, a->z_dim, a->v_dim, a->i_dim)) {
            const size_t k5 = generic_bwd_workspace_floats(&a->de, &a->ae, a->B);
            f = k5 > f ? k5 : f;
        }
        return f * sizeof(float);
    }
    if (latent16_dae_bwd_shape_ok(a)) {      // K8 (DAE) or, for unaligned views / kernel = generic, K5: the larger of the two
        const size_t k8 = latent16_dae_bwd_workspace_floats(a), k5 = generic_bwd_workspace_floats(&a->de, &a->ae, a->B);
        return (k8 > k5 ? k8 : k5) * sizeof(float);
    }
    return generic_bwd_workspace_floats(&a->de, &a->ae, a->B) * sizeof(float);
}

extern "C" int32_t psnode_dae_backward_f32(const psnode_dae_bwd_args_f32* a, void* workspace, size_t workspace_bytes, void* stream) {
    if (!a) return PSNODE_ERR_NULL;
    if (a->method < PSNODE_EULER || a->method > PSNODE_RK4_38) return PSNODE_ERR_METHOD;
    if (a->T < 1 || a->B < 1) return PSNODE_ERR_DIMS;
    if (!psnode_dae_backward_supported(a)) return PSNODE_ERR_UNSUPPORTED;
    for (int l = 0; l < a->de.n_layers; ++l) if (!a->de.weight[l] || !a->de.bias[l]) return PSNODE_ERR_NULL;
    for (int l = 0; l < a->ae.n_layers; ++l) if (!a->ae.weight[l] || !a->ae.bias[l]) return PSNODE_ERR_NULL;
    if (!a->t.ptr || !a->all_initial || !a->xs || !a->is || !a->grad_xs || !a->grad_x_init || !a->grad_all_initial || !a->grad_params_de ||
        !a->grad_params_ae)
        return PSNODE_ERR_NULL;
    if ((a->z_dim > 0 && !a->z.ptr) || (a->v_dim > 0 && !a->v.ptr)) return PSNODE_ERR_NULL;
    if (a->event_idx && ((a->z_dim > 0 && !a->z_jump) || (a->v_dim > 0 && !a->v_jump))) return PSNODE_ERR_NULL;
    if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 255u) || workspace_bytes < psnode_dae_backward_workspace_bytes(a))
        return PSNODE_ERR_WORKSPACE;
    {
        const bool sv = a->saved_act != nullptr;
        if ((a->saved_xstage != nullptr) != sv || (a->saved_ae_act != nullptr) != sv) return PSNODE_ERR_NULL;
        if (sv && a->event_idx && (!a->saved_ev_act || !a->saved_ev_i)) return PSNODE_ERR_NULL;
        if (sv && (use_mfma_dae_bwd(a) || !use_latent64_dae_bwd(a))) return PSNODE_ERR_UNSUPPORTED;      // only K9 reads them here
    }
    if (use_mfma_dae_bwd(a)) return dae_mfma_bwd_launch(a, static_cast<float*>(workspace), static_cast<hipStream_t>(stream));
    if (use_latent64_dae_bwd(a)) return latent64_dae_bwd_launch(a, static_cast<float*>(workspace), static_cast<hipStream_t>(stream));
    if (use_latent16_dae_bwd(a)) return latent16_dae_bwd_launch(a, static_cast<float*>(workspace), static_cast<hipStream_t>(stream));
    if (a->kernel == PSNODE_KERNEL_MFMA) return PSNODE_ERR_UNSUPPORTED;
    return generic_backward_launch(a->method, a->x_dim, a->z_dim, a->v_dim, a->i_dim, a->T, a->B, &a->de, &a->ae,
                                   ViewDev{a->t.ptr, a->t.stride_t, a->t.stride_b}, ViewDev{a->z.ptr, a->z.stride_t, a->z.stride_b},
                                   ViewDev{a->v.ptr, a->v.stride_t, a->v.stride_b}, a->all_initial, a->event_idx, a->z_jump, a->zj_stride_b,
                                   a->zj_stride_e, a->v_jump, a->vj_stride_b, a->vj_stride_e, a->n_events, a->xs, a->is, a->grad_xs, a->grad_is,
                                   a->grad_x_init, a->grad_z, a->grad_v, a->grad_z_jump, a->grad_v_jump, a->grad_all_initial,
                                   a->grad_params_de, a->grad_params_ae, static_cast<float*>(workspace), static_cast<hipStream_t>(stream));
}
