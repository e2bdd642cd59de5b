// K4w -- the sequential part of the backward pass through the ODE integrator for hidden widths 32 / 64 / 128 (C ABI
// psnode_ode_backward_wide_f32): recomputed stage evaluations + adjoint recursion through stages and steps, writing the rows the
// parameter gradients contract over.  The parameter gradients themselves are plain GEMMs over those rows (host side).
//
// Why split: K4 (psnode_backward.hip, hidden 64) keeps W2, W3, their transposes AND the dW accumulators on chip; at hidden 128 that is
// 3 x 128 KB next to 96 KB of transposed activation tiles -- more than the register file + LDS of a CU hold -- so shapes outside K4
// trained on the generic K5 (603 ms per 4096x1000 batch at hidden 128).  The adjoint recursion is the only part that is sequential in
// time; dW_l = sum over (step, stage, trajectory) of delta_l (x) h_{l-1} is one big GEMM with K = 16 M rows, which a library GEMM
// runs at > 100 TFLOP/s (profiles/scripts/gemm_probe.py).
//
// The kernel is K1 (psnode_mfma_impl.h) run backwards: same tile (16 trajectories, NWV = hidden/16 waves), same folded forward
// image and layer plan for the recompute, and the data path of the backward is ANOTHER evaluation of the same layer plan with
// transposed images and ELU' multipliers in place of ELU:
//   g3 = W4^T gk          (2 MFMAs, like L1)            delta3 = g3 * ELU'(h3)
//   delta2 = (W3^T delta3) * ELU'(h2)                   (all-gather + 4*NWV MFMAs, the mid layer; transposed image in LDS)
//   delta1 = (W2^T delta2) * ELU'(h1)
//   gX = F_x^T delta1     (4 MFMAs split-K + 8-byte all-reduce, like L4)
// W2 / W3 stay in VGPRs (forward image), W2^T / W3^T live in LDS (2 * NWV^2 KB; each lane reads back the A-operand values it wrote:
// conflict-free ds_read_b128, no barrier), exchanges as K1 (one barrier each, all reads in flight before the dependent MFMAs at 4 waves).
#include <string.h>

#include "psnode_wide_pack.h"

namespace psnode {
namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f4 wm4(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f4 wdact(f4 h) {   // ELU'(pre) from h = ELU(pre)
    return elu_grad_quad(h);
}

struct WideDev {
    int method, xd, zd, hreal;            // hreal: the MLP's hidden width (<= H = 16 * NWV; units beyond it are zero padding)
    long long T, B, k0, k1;
    const float *w1, *w4;                 // raw nn.Linear tensors for the small transposed operands
    ViewDev t, z;
    const float* a0;
    const int* ev;
    const float* zj;
    long long zjb, zje;
    const float *xs, *gout;
    float* carry;
    float *act[3], *delta[3], *gk, *xst, *dsum[3];
};

template <int METHOD, int NZM, int NWV>
__global__ __launch_bounds__(64 * NWV) void ode_backward_wide_kernel(const WideDev a, const float* __restrict__ pack_de,
                                                                       const f4* __restrict__ pack_t, const int NA) {
    constexpr int NX = kNXc, S = rk_stages(METHOD), H = 16 * NWV;
    using RD = Regs<NX, 0, NZM, NWV>;
    __shared__ f4 xbuf[2][NWV][64];
    extern __shared__ f4 wT[];   // [layer 0: W2^T | 1: W3^T][chunk][wave][lane]

    const int l = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = l >> 4, j = l & 15;
    const long long b0 = (long long)blockIdx.x * TBM;
    const bool valid = b0 + j < a.B;
    const long long b = valid ? b0 + j : a.B - 1;
    const int xd = a.xd, zd = a.zd, ne = zd, n = xd + zd;

    // ---- forward image -> registers (as K1), transposed images -> LDS
    const float* pw = pack_de + (size_t)w * (RD::COUNT + NA) * 64 + l;
    // 8 waves per tile leave 256 registers per lane: W2 / W3 (64 registers) are then re-read from the packed image (L2-resident,
    // 128 KB per workgroup and step) at the top of every step, so that their registers are free during the backward half, and the
    // hidden activations travel through the act[] rows this lane writes anyway instead of through 48 registers.
    constexpr bool STREAM = NWV >= 8;
    float w1xs[NX], w1z[NZM > 0 ? NZM : 1], w2r[STREAM ? 1 : 4 * NWV], w3r[STREAM ? 1 : 4 * NWV], w4[4];
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
#pragma unroll
    for (int c = 0; c < NWV; ++c) {
        wT[((0 * NWV + c) * NWV + w) * 64 + l] = pack_t[((0 * NWV + c) * NWV + w) * 64 + l];
        wT[((1 * NWV + c) * NWV + w) * 64 + l] = pack_t[((1 * NWV + c) * NWV + w) * 64 + l];
    }
    // small transposed operands straight from the nn.Linear tensors (u = 16w + i: this lane's A-operand row)
    //   w4T[r]  = W4[4r+g][u]                        g3[u] = sum_d W4[d][u] gk[d]
    //   fT[r]   = (Ws+Wd)[16w+4g+r][o],  o = x-dim carried by output row i (as the W4 rows of the forward image)
    float w4T[NX], fT[4];
    {
        const int i = j, u = 16 * w + i, K1 = 3 * n, HR = a.hreal;
#pragma unroll
        for (int r = 0; r < NX; ++r) { const int d = 4 * r + g; w4T[r] = (d < xd && u < HR) ? a.w4[(size_t)d * HR + u] : 0.0f; }
        const int o = 4 * (i & 3) + (i >> 2);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int uu = 16 * w + 4 * g + r;
            fT[r] = (o < xd && uu < HR) ? a.w1[(size_t)uu * K1 + 2 * n + o] + a.w1[(size_t)uu * K1 + n + o] : 0.0f;
        }
    }

    // ---- per-trajectory constants (as K1)
    float a0e[NZM > 0 ? NZM : 1];
    int ecol[NZM > 0 ? NZM : 1];
    bool eon[NZM > 0 ? NZM : 1];
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
        c0 = wm4(pw[(RD::COUNT + m) * 64], q < n ? a.a0[b * n + q] : 0.0f, c0);
    }

    int p = 0;
    constexpr bool PREFETCH_ALL = NWV <= 4;
    // forward H->H layer with the weights in registers (K1's `mid`), returns the pre-activation
    auto mid = [&](const float (&wm)[4 * NWV], const f4 bias, const f4 h) -> f4 {
        xbuf[p][w][l] = h;
        f4 accA = bias, accB = f4{0.f, 0.f, 0.f, 0.f};
        accA = wm4(wm[0], h[0], accA); accB = wm4(wm[1], h[1], accB);
        accA = wm4(wm[2], h[2], accA); accB = wm4(wm[3], h[3], accB);
        __builtin_amdgcn_sched_barrier(0);
        lds_barrier();
        f4 vq[NWV];
        if constexpr (PREFETCH_ALL) {
#pragma unroll
            for (int c = 1; c < NWV; ++c) vq[c] = xbuf[p][(w + c) & (NWV - 1)][l];
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int c = 1; c < NWV; ++c) {
            const f4 v = PREFETCH_ALL ? vq[c] : xbuf[p][(w + c) & (NWV - 1)][l];
            accA = wm4(wm[4 * c + 0], v[0], accA); accB = wm4(wm[4 * c + 1], v[1], accB);
            accA = wm4(wm[4 * c + 2], v[2], accA); accB = wm4(wm[4 * c + 3], v[3], accB);
        }
        p ^= 1;
        return accA + accB;
    };
    // transposed H->H layer with the image in LDS: out[own units] = sum_k W[k][own] d[k]
    auto midT = [&](const int layer, const f4 d) -> f4 {
        xbuf[p][w][l] = d;
        const f4* wl = wT + ((size_t)layer * NWV * NWV + w) * 64 + l;
        f4 wq = wl[0];
        f4 accA = wm4(wq[0], d[0], f4{0.f, 0.f, 0.f, 0.f}), accB = wm4(wq[1], d[1], f4{0.f, 0.f, 0.f, 0.f});
        accA = wm4(wq[2], d[2], accA); accB = wm4(wq[3], d[3], accB);
        __builtin_amdgcn_sched_barrier(0);
        lds_barrier();
#pragma unroll
        for (int c = 1; c < NWV; ++c) {
            const f4 v = xbuf[p][(w + c) & (NWV - 1)][l];
            wq = wl[c * NWV * 64];
            accA = wm4(wq[0], v[0], accA); accB = wm4(wq[1], v[1], accB);
            accA = wm4(wq[2], v[2], accA); accB = wm4(wq[3], v[3], accB);
        }
        p ^= 1;
        return accA + accB;
    };
    // split-K over the waves' own units + 8-byte all-reduce of output rows r < 2 (x-layout: dim 4r+g in lane group g, register r)
    auto out2 = [&](const float (&wq)[4], const f4 h, const f4 init) -> f2 {
        f4 accA = wm4(wq[0], h[0], f4{0.f, 0.f, 0.f, 0.f}), accB = wm4(wq[1], h[1], f4{0.f, 0.f, 0.f, 0.f});
        accA = wm4(wq[2], h[2], accA); accB = wm4(wq[3], h[3], accB);
        const f4 part = accA + accB;
        f2* xb2 = reinterpret_cast<f2*>(&xbuf[p][0][0]);
        xb2[w * 64 + l] = f2{part[0], part[1]};
        lds_barrier();
        f2 out = f2{init[0], init[1]};
#pragma unroll
        for (int c = 0; c < NWV; ++c) { const f2 q = xb2[c * 64 + l]; out[0] += q[0]; out[1] += q[1]; }
        p ^= 1;
        return out;
    };

    // addressing: sbase(uniform row base) + 32-bit per-lane BYTE offset (psnode_common.h: ldg / stg)
    const unsigned offH = 4u * ((unsigned)(b * H) + 16 * w + 4 * g);      // rows of H floats: this lane's 4 units of its trajectory
    const unsigned offX = 4u * ((unsigned)(b * xd) + g);                  // rows of x_dim floats (+ 16 r)
    unsigned offXc[NX];                                                    // the same with the column clamped into the row (ldg_sel)
#pragma unroll
    for (int r = 0; r < NX; ++r) offXc[r] = 4u * ((unsigned)(b * xd) + (4 * r + g < xd ? 4 * r + g : 0));
    const unsigned offT = 4u * (unsigned)(b * a.t.sb), offZ = 4u * (unsigned)(b * a.z.sb), offZJ = 4u * (unsigned)(b * a.zjb);
    auto load_ext = [&, offZ, offZJ](const long long k, const int ev, float (&dst)[NZM > 0 ? NZM : 1]) {
        if constexpr (NZM > 0) {
            const gptr<const float> row = sbase(ev >= 0 ? a.zj + (long long)ev * a.zje : a.z.p + k * a.z.st);
            const unsigned m_ = ev >= 0 ? ~0u : 0u, zo = (offZJ & m_) | (offZ & ~m_);
#pragma unroll
            for (int m = 0; m < NZM; ++m) dst[m] = ldg_sel(row, zo + 4u * ecol[m], eon[m]);      // (ecol is 0 on padding slots)
        }
    };
    auto load_x2 = [&](const float* base, const long long k, float (&dst)[NX]) {
        const gptr<const float> row = sbase(base + k * a.B * xd);
#pragma unroll
        for (int r = 0; r < NX; ++r) dst[r] = ldg_sel(row, offXc[r], 4 * r + g < xd);      // branch-free (psnode_common.h: ldg_sel)
    };
    auto load_dt = [&](const long long k) -> float { return ldg<float>(sbase(a.t.p + (k + 1) * a.t.st), offT) - ldg<float>(sbase(a.t.p + k * a.t.st), offT); };

    float gcar[NX];
#pragma unroll
    for (int r = 0; r < NX; ++r) gcar[r] = (valid && 4 * r + g < xd) ? a.carry[b * xd + 4 * r + g] : 0.0f;

    const long long nrow = a.B;           // rows per (step, stage) in the stored tensors
    auto event_of = [&](const long long k) -> int { return a.ev ? __builtin_amdgcn_readfirstlane(a.ev[k]) : -1; };
    // inputs of a step are requested one step ahead (at the top of the previous step's backward half)
    float x0n[NX], ginn[NX], extn[NZM > 0 ? NZM : 1], hn;
    load_x2(a.xs, a.k1 - 1, x0n);
    load_x2(a.gout, a.k1, ginn);
    load_ext(a.k1 - 1, event_of(a.k1 - 1), extn);
    hn = load_dt(a.k1 - 1);
    for (long long k = a.k1 - 1; k >= a.k0; --k) {
        float x0[NX], gin[NX], ext[NZM > 0 ? NZM : 1];
#pragma unroll
        for (int r = 0; r < NX; ++r) { x0[r] = x0n[r]; gin[r] = ginn[r]; }
#pragma unroll
        for (int m = 0; m < NZM; ++m) ext[m] = extn[m];
        const float h_ = hn;
        f4 cz = c0;
#pragma unroll
        for (int m = 0; m < NZM; ++m) cz = wm4(w1z[m], ext[m] - a0e[m], cz);

        // ---- phase A: stage evaluations (K1's plan; pre-activations are not kept, the ELU outputs are)
        float X[S][NX], ks[S][NX];
        f4 h1[STREAM ? 1 : S], h2[STREAM ? 1 : S], h3[STREAM ? 1 : S];
        auto row_blk = [&](const int s) -> size_t { return (size_t)((k - a.k0) * S + s) * nrow; };   // uniform
        f4 la1, la2, la3;    // ELU outputs of the last stage evaluated (the first one the backward half needs)
        {
            float w2s[STREAM ? 4 * NWV : 1], w3s[STREAM ? 4 * NWV : 1];
            if constexpr (STREAM) {
#pragma unroll
                for (int q = 0; q < 4 * NWV; ++q) { w2s[q] = pw[(RD::W2 + q) * 64]; w3s[q] = pw[(RD::W3 + q) * 64]; }
            }
#pragma unroll
            for (int s = 0; s < S; ++s) {
#pragma unroll
                for (int r = 0; r < NX; ++r) {
                    float acc = 0.0f;
#pragma unroll
                    for (int jj = 0; jj < s; ++jj) acc += rk_a(METHOD, s, jj) * ks[jj][r];
                    X[s][r] = s == 0 ? x0[r] : x0[r] + h_ * acc;
                }
                f4 accA = cz, accB = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < NX; ++r) {
                    if (r & 1) accB = wm4(w1xs[r], X[s][r], accB);
                    else accA = wm4(w1xs[r], X[s][r], accA);
                }
                const f4 a1 = elu_quad(NX > 1 ? accA + accB : accA);
                f4 a2, a3;
                if constexpr (STREAM) { a2 = elu_quad(mid(w2s, b2, a1)); a3 = elu_quad(mid(w3s, b3, a2)); }
                else { a2 = elu_quad(mid(w2r, b2, a1)); a3 = elu_quad(mid(w3r, b3, a2)); }
                if (valid) {     // rows for the parameter-gradient GEMMs: this wave's 16 units of trajectory j, 16 bytes per lane
                    const size_t rb = row_blk(s) * H;
                    stg<f4>(sbase(a.act[0] + rb), offH, a1);
                    stg<f4>(sbase(a.act[1] + rb), offH, a2);
                    stg<f4>(sbase(a.act[2] + rb), offH, a3);
                }
                if constexpr (!STREAM) { h1[s] = a1; h2[s] = a2; h3[s] = a3; }
                if (s == S - 1) { la1 = a1; la2 = a2; la3 = a3; }
                const f2 kk = out2(w4, a3, b4);
                ks[s][0] = kk[0];
                if constexpr (NX > 1) ks[s][1] = kk[1];
            }
        }

        // ---- phase B: stages backwards.  The next step's inputs are requested here and consumed a whole backward half later.
        if (k > a.k0) {
            load_x2(a.xs, k - 1, x0n);
            load_x2(a.gout, k, ginn);
            load_ext(k - 1, event_of(k - 1), extn);
            hn = load_dt(k - 1);
        }
        float gks[S][NX], gx0[NX];
#pragma unroll
        for (int r = 0; r < NX; ++r) {
            const float g1 = gcar[r] + (valid ? gin[r] : 0.0f);
            gx0[r] = g1;
#pragma unroll
            for (int s = 0; s < S; ++s) gks[s][r] = (h_ * rk_b(METHOD, s)) * g1;
        }
        f4 D1 = f4{0.f, 0.f, 0.f, 0.f}, D2 = D1, D3 = D1;
        f4 na1 = la1, na2 = la2, na3 = la3;       // STREAM: rows of the stage handled next, requested one stage ahead
#pragma unroll
        for (int s = S - 1; s >= 0; --s) {
            f4 a1, a2, a3;
            if constexpr (STREAM) {     // this lane's own rows, written in phase A (lanes of a ragged tile read trajectory B-1's)
                a1 = na1; a2 = na2; a3 = na3;
                if (s > 0) {
                    const size_t rb = row_blk(s - 1) * H;
                    na1 = ldg<f4>(sbase(a.act[0] + rb), offH);
                    na2 = ldg<f4>(sbase(a.act[1] + rb), offH);
                    na3 = ldg<f4>(sbase(a.act[2] + rb), offH);
                }
            } else {
                a1 = h1[s]; a2 = h2[s]; a3 = h3[s];
            }
            f4 g3 = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < NX; ++r) g3 = wm4(w4T[r], gks[s][r], g3);
            const f4 d3 = g3 * wdact(a3);
            const f4 d2 = midT(1, d3) * wdact(a2);
            const f4 d1 = midT(0, d2) * wdact(a1);
            D1 += d1; D2 += d2; D3 += d3;
            const f2 gx = out2(fT, d1, f4{0.f, 0.f, 0.f, 0.f});
            if (valid) {
                const size_t rb = row_blk(s) * H;
                stg<f4>(sbase(a.delta[0] + rb), offH, d1);
                stg<f4>(sbase(a.delta[1] + rb), offH, d2);
                stg<f4>(sbase(a.delta[2] + rb), offH, d3);
                if (w == 0) {
                    const gptr<float> gkr = sbase(a.gk + row_blk(s) * xd);
                    const gptr<float> xsr = sbase(a.xst + row_blk(s) * xd);
#pragma unroll
                    for (int r = 0; r < NX; ++r) {
                        if (4 * r + g < xd) {
                            stg<float>(gkr, offX + 16u * r, gks[s][r]);
                            stg<float>(xsr, offX + 16u * r, X[s][r]);
                        }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < NX; ++r) {
                const float gxr = r == 0 ? gx[0] : gx[1];
                gx0[r] += gxr;
#pragma unroll
                for (int jj = 0; jj < s; ++jj) gks[jj][r] += (h_ * rk_a(METHOD, s, jj)) * gxr;
            }
        }
        if (valid) {     // per-step sums over the stages: what the bias / input gradients contract over (a quarter of the rows at RK4)
            const size_t rb = (size_t)(k - a.k0) * nrow * H;
            stg<f4>(sbase(a.dsum[0] + rb), offH, D1);
            stg<f4>(sbase(a.dsum[1] + rb), offH, D2);
            stg<f4>(sbase(a.dsum[2] + rb), offH, D3);
        }
#pragma unroll
        for (int r = 0; r < NX; ++r) gcar[r] = gx0[r];
    }
    if (w == 0 && valid) {
#pragma unroll
        for (int r = 0; r < NX; ++r) if (4 * r + g < xd) a.carry[b * xd + 4 * r + g] = gcar[r];
    }
}

template <int METHOD, int NWV>
hipError_t launch_wide(const WideDev& a, int NZM, const float* pde, const f4* pt, int NA, hipStream_t s) {
    const dim3 grid((unsigned)((a.B + TBM - 1) / TBM)), block(64 * NWV);
    const size_t lds = wide_t_floats(NWV) * sizeof(float);
#define PSNODE_WIDE(NZM_)                                                                                                       \
    {                                                                                                                           \
        auto kern = &ode_backward_wide_kernel<METHOD, NZM_, NWV>;                                                               \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return e;                                                                                          \
        hipLaunchKernelGGL(kern, grid, block, lds, s, a, pde, pt, NA);                                                          \
        return hipGetLastError();                                                                                               \
    }
    switch (NZM) {
        case 0: PSNODE_WIDE(0)
        case 1: PSNODE_WIDE(1)
        case 2: PSNODE_WIDE(2)
        case 3: PSNODE_WIDE(3)
        case 4: PSNODE_WIDE(4)
        default: return hipErrorNotSupported;
    }
#undef PSNODE_WIDE
}

template <int NWV>
hipError_t launch_wide_method(const WideDev& a, int NZM, const float* pde, const f4* pt, int NA, hipStream_t s) {
    switch (a.method) {
        case PSNODE_EULER: return launch_wide<PSNODE_EULER, NWV>(a, NZM, pde, pt, NA, s);
        case PSNODE_MIDPOINT: return launch_wide<PSNODE_MIDPOINT, NWV>(a, NZM, pde, pt, NA, s);
        default: return launch_wide<PSNODE_RK4_38, NWV>(a, NZM, pde, pt, NA, s);
    }
}

}  // namespace
}  // namespace psnode

using namespace psnode;

extern "C" {

int32_t psnode_ode_backward_wide_supported(const psnode_ode_bwd_wide_args_f32* a) {
    if (!a || a->method < PSNODE_EULER || a->method > PSNODE_RK4_38) return 0;
    if (a->x_dim < 1 || a->x_dim > 4 * kNXc || a->z_dim < 0 || 2 * a->z_dim > 4 * kMaxNZM) return 0;
    if (!wide_hidden(a->de)) return 0;
    return a->de.in_dim == 3 * (a->x_dim + a->z_dim) && a->de.out_dim[3] == a->x_dim;
}

size_t psnode_ode_backward_wide_workspace_bytes(const psnode_ode_bwd_wide_args_f32* a) {
    if (!psnode_ode_backward_wide_supported(a)) return 0;
    const int nw = wide_hidden(a->de) / 16;
    return (wide_fwd_floats(nw, a->x_dim + a->z_dim) + wide_t_floats(nw) + 128) * sizeof(float);
}

int32_t psnode_ode_backward_wide_f32(const psnode_ode_bwd_wide_args_f32* p, void* workspace, size_t workspace_bytes, void* stream) {
    if (!p) return PSNODE_ERR_NULL;
    if (p->method < PSNODE_EULER || p->method > PSNODE_RK4_38) return PSNODE_ERR_METHOD;
    if (!psnode_ode_backward_wide_supported(p)) return PSNODE_ERR_UNSUPPORTED;
    if (p->T < 2 || p->B < 1 || p->k0 < 0 || p->k1 <= p->k0 || p->k1 > p->T - 1) return PSNODE_ERR_DIMS;
    for (int l = 0; l < 4; ++l) if (!p->de.weight[l] || !p->de.bias[l]) return PSNODE_ERR_NULL;
    if (!p->t.ptr || !p->all_initial || !p->xs || !p->grad_xs || !p->carry || !p->gk || !p->xstage) return PSNODE_ERR_NULL;
    for (int l = 0; l < 3; ++l) if (!p->act[l] || !p->delta[l] || !p->dsum[l]) return PSNODE_ERR_NULL;
    if (p->z_dim > 0 && !p->z.ptr) return PSNODE_ERR_NULL;
    if (p->event_idx && p->z_dim > 0 && !p->z_jump) return PSNODE_ERR_NULL;
    if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 255u) || workspace_bytes < psnode_ode_backward_wide_workspace_bytes(p))
        return PSNODE_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int H = wide_hidden(p->de), nw = H / 16, xd = p->x_dim, zd = p->z_dim, n = xd + zd;
    {   // per-lane offsets inside a row are 32-bit next to a scalar row base
        const int64_t lim = (int64_t)1 << 30, Bm = p->B;
        const int64_t sb[] = {H, p->t.stride_b, zd > 0 ? p->z.stride_b : 0, p->event_idx && zd > 0 ? p->zj_stride_b : 0};
        for (int64_t q : sb) if (q < 0 || Bm * q + 64 >= lim) return PSNODE_ERR_DIMS;
    }
    const int NZM = (2 * zd + 3) / 4, NA = (n + 3) / 4;
    float* pde = static_cast<float*>(workspace);
    f4* pt = reinterpret_cast<f4*>(pde + ((wide_fwd_floats(nw, n) + 63) / 64) * 64);
    PackMfma f;
    memset(&f, 0, sizeof(f));
    f.ae = 0; f.nw = nw; f.xd = xd; f.ne = zd; f.n = n; f.nzv = zd; f.NX = kNXc; f.NB = 0; f.NE = NZM; f.NA = NA; f.fold = 1;
    f.hreal = p->de.out_dim[0];
    f.w1 = p->de.weight[0]; f.b1 = p->de.bias[0]; f.w2 = p->de.weight[1]; f.b2 = p->de.bias[1];
    f.w3 = p->de.weight[2]; f.b3 = p->de.bias[2]; f.w4 = p->de.weight[3]; f.b4 = p->de.bias[3];
    f.out_dim = xd; f.out = pde;
    hipLaunchKernelGGL(pack_wide_fwd_kernel, dim3(32), dim3(256), 0, s, f);
    PackWideT t{nw, p->de.out_dim[0], p->de.weight[1], p->de.weight[2], pt};
    hipLaunchKernelGGL(pack_wide_t_kernel, dim3(64), dim3(256), 0, s, t);
    if (hipGetLastError() != hipSuccess) return PSNODE_ERR_HIP;
    WideDev a;
    memset(&a, 0, sizeof(a));
    a.method = p->method; a.xd = xd; a.zd = zd; a.hreal = p->de.out_dim[0]; a.T = p->T; a.B = p->B; a.k0 = p->k0; a.k1 = p->k1;
    a.w1 = p->de.weight[0]; a.w4 = p->de.weight[3];
    a.t = ViewDev{p->t.ptr, p->t.stride_t, p->t.stride_b};
    a.z = ViewDev{p->z.ptr, p->z.stride_t, p->z.stride_b};
    a.a0 = p->all_initial; a.ev = p->event_idx; a.zj = p->z_jump; a.zjb = p->zj_stride_b; a.zje = p->zj_stride_e;
    a.xs = p->xs; a.gout = p->grad_xs; a.carry = p->carry;
    for (int l = 0; l < 3; ++l) { a.act[l] = p->act[l]; a.delta[l] = p->delta[l]; a.dsum[l] = p->dsum[l]; }
    a.gk = p->gk; a.xst = p->xstage;
    hipError_t e;
    switch (nw) {
        case 2: e = launch_wide_method<2>(a, NZM, pde, pt, NA, s); break;
        case 4: e = launch_wide_method<4>(a, NZM, pde, pt, NA, s); break;
        default: e = launch_wide_method<8>(a, NZM, pde, pt, NA, s); break;
    }
    if (e == hipErrorNotSupported) return PSNODE_ERR_UNSUPPORTED;
    return e == hipSuccess ? PSNODE_OK : PSNODE_ERR_HIP;
}

}  // extern "C"
