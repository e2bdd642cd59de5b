// K7w -- the sequential part of the backward pass through the DAE integrator for hidden widths 32 / 64 / 128 (C ABI
// psnode_dae_backward_wide_f32): K4w (psnode_backward_wide.hip) extended by the AE head.  Per grid point j the head
// i_j = g(x_j; z_j, v_j) is recomputed and run backwards with the output adjoint dL/dis[j] + (what step j's DE returned through its
// algebraic inputs); per step the DE stages as in K4w, with the algebraic inputs taken from the saved is[k] (or recomputed from the
// jump values at an event step, whose head then takes the DE's algebraic adjoint instead of grid point k's).  Everything that is a
// contraction over the stored rows (parameter gradients, dL/dz, dL/dv, dL/dall_initial) is a library GEMM on the host side
// (py_psnode_amd/fused.py:dae_backward_wide).
//
// Layouts as K2 (psnode_mfma_impl.h): the AE's output rows are the DE's ext slots (row (g, m) = slot q = 4m+g), so the head's value
// feeds the DE's per-step MFMAs and the DE's ext adjoint feeds the head's W4^T product without any data movement; an i-dim occupies
// two slots (the `s - a0` block and the `s` block), whose adjoints the transposed W4 image sums by carrying the same row for both.
// Weights: the DE's forward image in registers (8 waves: re-read per step), its transposes in LDS; the AE's forward image likewise,
// its transposes in LDS up to hidden 64 and read from the packed image (L2) at hidden 128, where the LDS is full.
#include <string.h>

#include "psnode_pack.h"

namespace psnode {
namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f4 wm4(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f4 wdact(f4 h) {   // ELU'(pre) from h = ELU(pre)
    return elu_grad_quad(h);
}

struct WideDaeDev {
    int method, xd, zd, vd, id, hreal;    // hreal: the MLPs' hidden width (<= H = 16 * NWV; units beyond it are zero padding)
    long long T, B, k0, k1;
    const float *w1, *w4, *aw1, *aw4;     // raw nn.Linear tensors for the small transposed operands
    ViewDev t, z, v;
    const float* a0;
    const int* ev;
    const float *zj, *vj;
    long long zjb, zje, vjb, vje;
    const float *xs, *is, *gxs, *gis;
    float *carry_x, *carry_i;
    float *act[3], *delta[3], *gk, *xst, *dsum[3];
    float *aact[3], *adelta[3], *agi;
    float *eact[3], *edelta[3], *egi, *ei;
};

struct PackWideT {
    int nw, hreal;
    const float *w2, *w3;
    f4* out;
};
// [(layer * NWV + c) * NWV + w][lane] (f4): reg r = W[16((w+c) % NWV) + 4g + r][16w + i]   (as psnode_backward_wide.hip)
__global__ void pack_wide_t_kernel(const PackWideT p) {
    const int H = p.hreal, total = 2 * p.nw * p.nw * 64;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int lane = idx & 63, w = (idx >> 6) % p.nw, c = ((idx >> 6) / p.nw) % p.nw, layer = (idx >> 6) / (p.nw * p.nw);
        const int i = lane & 15, g = lane >> 4, ws = (w + c) & (p.nw - 1);
        const float* W = layer ? p.w3 : p.w2;
        f4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * ws + 4 * g + r, col = 16 * w + i;
            v[r] = (row < H && col < H) ? W[(size_t)row * H + col] : 0.0f;
        }
        p.out[idx] = v;
    }
}
__global__ void pack_wide_fwd_kernel(const PackMfma p) {
    const int R = pack_fwd_count(p);
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < p.nw * R * 64; idx += gridDim.x * blockDim.x)
        p.out[idx] = pack_fwd_value(p, (idx >> 6) / R, (idx >> 6) % R, idx & 63);
}

__host__ __device__ constexpr bool aet_in_lds(int nwv) { return nwv <= 4; }

// Addressing: every global access of the time loop is sbase(uniform row base) + 32-bit per-lane offset (psnode_common.h).

template <int METHOD, int NZM, int NZA, int NWV>
__global__ __launch_bounds__(64 * NWV) void dae_backward_wide_kernel(const WideDaeDev a, const float* __restrict__ pack_de,
                                                                       const float* __restrict__ pack_ae, const f4* __restrict__ pack_t,
                                                                       const f4* __restrict__ pack_ta, const int NA) {
    constexpr int NX = kNXc, S = rk_stages(METHOD), H = 16 * NWV;
    using RD = Regs<NX, 0, NZM, NWV>;
    using RA = Regs<NX, 0, NZA, NWV>;
    constexpr bool STREAM = NWV >= 8;        // 256 registers per lane: H->H forward weights re-read where they are used
    constexpr bool AET_LDS = aet_in_lds(NWV);
    constexpr int TSZ = 2 * NWV * NWV * 64;  // f4 per transposed pair
    __shared__ f4 xbuf[2][NWV][64];
    extern __shared__ f4 wT[];               // DE: [W2^T | W3^T][chunk][wave][lane]; AET_LDS: the AE's behind it

    const int l = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = l >> 4, j = l & 15;
    const long long b0 = (long long)blockIdx.x * TBM;
    const bool valid = b0 + j < a.B;
    const long long b = valid ? b0 + j : a.B - 1;
    const int xd = a.xd, zd = a.zd, vd = a.vd, idim = a.id;
    const int nzv = zd + vd, ne = nzv + idim, n = xd + ne;

    // ---- forward images -> registers, transposed images -> LDS
    const float* pw = pack_de + (size_t)w * (RD::COUNT + NA) * 64 + l;
    const float* pwa = pack_ae + (size_t)w * (RA::COUNT + NA) * 64 + l;
    float w1xs[NX], w1z[NZM], w2r[STREAM ? 1 : 4 * NWV], w3r[STREAM ? 1 : 4 * NWV], w4[4];
    float aw1x[NX], aw1e[NZA], aw2r[STREAM ? 1 : 4 * NWV], aw3r[STREAM ? 1 : 4 * NWV], aw4[4];
    f4 b1r, b2, b3, b4, ab1r, ab2, ab3, ab4;
#pragma unroll
    for (int r = 0; r < NX; ++r) { w1xs[r] = pw[(RD::W1A + r) * 64]; aw1x[r] = pwa[(RA::W1A + r) * 64]; }
#pragma unroll
    for (int m = 0; m < NZM; ++m) w1z[m] = pw[(RD::W1E + m) * 64];
#pragma unroll
    for (int m = 0; m < NZA; ++m) aw1e[m] = pwa[(RA::W1E + m) * 64];
#pragma unroll
    for (int k = 0; k < (STREAM ? 0 : 4 * NWV); ++k) {
        w2r[k] = pw[(RD::W2 + k) * 64]; w3r[k] = pw[(RD::W3 + k) * 64];
        aw2r[k] = pwa[(RA::W2 + k) * 64]; aw3r[k] = pwa[(RA::W3 + k) * 64];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        w4[r] = pw[(RD::W4 + r) * 64]; aw4[r] = STREAM ? 0.0f : pwa[(RA::W4 + r) * 64];
        b1r[r] = pw[(RD::B1 + r) * 64]; ab1r[r] = pwa[(RA::B1 + r) * 64];
        if constexpr (!STREAM) {     // (STREAM: re-read next to the H->H weights)
            b2[r] = pw[(RD::B2 + r) * 64]; b3[r] = pw[(RD::B3 + r) * 64]; b4[r] = pw[(RD::B4 + r) * 64];
            ab2[r] = pwa[(RA::B2 + r) * 64]; ab3[r] = pwa[(RA::B3 + r) * 64]; ab4[r] = pwa[(RA::B4 + r) * 64];
        } else {
            b2[r] = b3[r] = b4[r] = ab2[r] = ab3[r] = ab4[r] = 0.0f;
        }
    }
#pragma unroll
    for (int c = 0; c < 2 * NWV; ++c) {
        wT[(c * NWV + w) * 64 + l] = pack_t[(c * NWV + w) * 64 + l];
        if constexpr (AET_LDS) wT[TSZ + (c * NWV + w) * 64 + l] = pack_ta[(c * NWV + w) * 64 + l];
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
    //   fE[r]   = W1[16w+4g+r][column of ext slot of row i], i-kind slots only:  adjoint of the DE's algebraic inputs, slot layout
    //   aw4T[m] = AW4[i-dim of slot 4m+g][16w+i]          g3a = AW4^T (slot adjoints): both slots of an i-dim carry its row
    //   afT[r]  = AW1[16w+4g+r][n + x-dim of row i]
    float w4T[NX], fT[4], fE[4], aw4T[NZM], afT[4];
    {
        const int i = j, u = 16 * w + i, K1 = 3 * n, K1a = n + xd + nzv, HR = a.hreal;
#pragma unroll
        for (int r = 0; r < NX; ++r) { const int d = 4 * r + g; w4T[r] = (d < xd && u < HR) ? a.w4[(size_t)d * HR + u] : 0.0f; }
#pragma unroll
        for (int m = 0; m < NZM; ++m) aw4T[m] = (ekind[m] == 2 && u < HR) ? a.aw4[(size_t)ecol[m] * HR + u] : 0.0f;
        const int o = 4 * (i & 3) + (i >> 2);           // x-dim / ext slot carried by output row i
        const int eo = slot_ext(o, ne);
        const int ecolumn = o < ne ? n + xd + o : 2 * n + xd + (o - ne);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int uu = 16 * w + 4 * g + r;
            const bool on = uu < HR;
            fT[r] = (o < xd && on) ? a.w1[(size_t)uu * K1 + 2 * n + o] + a.w1[(size_t)uu * K1 + n + o] : 0.0f;
            fE[r] = (eo >= nzv && on) ? a.w1[(size_t)uu * K1 + ecolumn] : 0.0f;
            afT[r] = (o < xd && on) ? a.aw1[(size_t)uu * K1a + n + o] : 0.0f;
        }
    }
    // bias + W1[:, a0 columns] . a0
    f4 c0 = b1r, c0a = ab1r;
    for (int m = 0; m < NA; ++m) {
        const int q = 4 * m + g;
        const float av = q < n ? a.a0[b * n + q] : 0.0f;
        c0 = wm4(pw[(RD::COUNT + m) * 64], av, c0);
        c0a = wm4(pwa[(RA::COUNT + m) * 64], av, c0a);
    }

    int p = 0;
    constexpr bool PREFETCH_ALL = NWV <= 4;
    // forward H->H layer, weights in registers (K1's `mid`); returns the pre-activation
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
    // transposed H->H layer, image in LDS at f4 offset `base` of wT: out[own units] = sum_k W[k][own] d[k]
    auto midT = [&](const int base, const f4 d) -> f4 {
        xbuf[p][w][l] = d;
        const f4* wl = wT + base + w * 64 + l;
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
    // the same with the image read from the packed tensor in global memory (hidden 128's AE: the LDS holds the DE's transposes);
    // all chunks are requested before the exchange so that their latency overlaps it
    auto midTg = [&](const f4* __restrict__ img, const f4 d) -> f4 {
        const f4* wlo = img + w * 64 + l;
        asm volatile("" : "+v"(wlo));
        const gptr<const f4> wl = (gptr<const f4>)wlo;
        f4 wq[NWV];
#pragma unroll
        for (int c = 0; c < NWV; ++c) wq[c] = wl[c * NWV * 64];
        xbuf[p][w][l] = d;
        f4 accA = wm4(wq[0][0], d[0], f4{0.f, 0.f, 0.f, 0.f}), accB = wm4(wq[0][1], d[1], f4{0.f, 0.f, 0.f, 0.f});
        accA = wm4(wq[0][2], d[2], accA); accB = wm4(wq[0][3], d[3], accB);
        lds_barrier();
#pragma unroll
        for (int c = 1; c < NWV; ++c) {
            const f4 v = xbuf[p][(w + c) & (NWV - 1)][l];
            accA = wm4(wq[c][0], v[0], accA); accB = wm4(wq[c][1], v[1], accB);
            accA = wm4(wq[c][2], v[2], accA); accB = wm4(wq[c][3], v[3], accB);
        }
        p ^= 1;
        return accA + accB;
    };
    // split-K over the waves' own units + all-reduce of the output rows: 2 rows (x-layout) or 4 rows (slot layout)
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
    auto out4 = [&](const float (&wq)[4], const f4 h, const f4 init) -> f4 {
        f4 accA = wm4(wq[0], h[0], f4{0.f, 0.f, 0.f, 0.f}), accB = wm4(wq[1], h[1], f4{0.f, 0.f, 0.f, 0.f});
        accA = wm4(wq[2], h[2], accA); accB = wm4(wq[3], h[3], accB);
        xbuf[p][w][l] = accA + accB;
        lds_barrier();
        f4 out = init;
#pragma unroll
        for (int c = 0; c < NWV; ++c) out += xbuf[p][c][l];
        p ^= 1;
        return out;
    };

    // per-lane BYTE offsets (psnode_common.h: at)
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
        const unsigned m = ev >= 0 ? ~0u : 0u;     // (bit select: a ?: between the two captured offsets becomes a select between
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
        for (int r = 0; r < NX; ++r) acc = wm4(aw1x[r], xa[r], acc);
#pragma unroll
        for (int m = 0; m < NZA; ++m) acc = wm4(aw1e[m], zv_val(zr, akind[m], acol[m]), acc);
        a1 = elu_quad(acc);
        if constexpr (STREAM) {
            // (the pointer is made opaque at every use: left visible, the loads are loop-invariant, get hoisted out of the time loop
            //  and the 64 registers they were meant to free are spilled to scratch instead)
            const float* pwo = pwa;
            asm volatile("" : "+v"(pwo));
            const gptr<const float> pws = (gptr<const float>)pwo;     // (global, not generic: an opaque generic pointer loads flat_*)
            float ws2[4 * NWV], ws3[4 * NWV];
#pragma unroll
            for (int q = 0; q < 4 * NWV; ++q) { ws2[q] = pws[(RA::W2 + q) * 64]; ws3[q] = pws[(RA::W3 + q) * 64]; }
            f4 sb2, sb3;
#pragma unroll
            for (int r = 0; r < 4; ++r) { sb2[r] = pws[(RA::B2 + r) * 64]; sb3[r] = pws[(RA::B3 + r) * 64]; }
            a2 = elu_quad(mid(ws2, sb2, a1));
            a3 = elu_quad(mid(ws3, sb3, a2));
        } else {
            a2 = elu_quad(mid(aw2r, ab2, a1));
            a3 = elu_quad(mid(aw3r, ab3, a2));
        }
    };
    // AE head backwards: output adjoint gs (slot layout), rows written at row index `row` of (ract, rdelta, rgi); returns dL/dxa
    struct HeadRows { float *a1, *a2, *a3, *d1, *d2, *d3, *gi; };
    auto ae_adjoint = [&](const f4 a1, const f4 a2, const f4 a3, const float (&gs)[NZM], const HeadRows hr, const size_t row) -> f2 {
        f4 g3 = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < NZM; ++m) g3 = wm4(aw4T[m], gs[m], g3);
        const f4 d3 = g3 * wdact(a3);
        f4 d2, d1;
        if constexpr (AET_LDS) {
            d2 = midT(TSZ + NWV * NWV * 64, d3) * wdact(a2);
            d1 = midT(TSZ, d2) * wdact(a1);
        } else {
            d2 = midTg(pack_ta + NWV * NWV * 64, d3) * wdact(a2);
            d1 = midTg(pack_ta, d2) * wdact(a1);
        }
        const f2 gx = out2(afT, d1, f4{0.f, 0.f, 0.f, 0.f});
        if (valid) {
            const size_t rb = row * a.B * H;
            stg<f4>(sbase(hr.a1 + rb), offH, a1);
            stg<f4>(sbase(hr.a2 + rb), offH, a2);
            stg<f4>(sbase(hr.a3 + rb), offH, a3);
            stg<f4>(sbase(hr.d1 + rb), offH, d1);
            stg<f4>(sbase(hr.d2 + rb), offH, d2);
            stg<f4>(sbase(hr.d3 + rb), offH, d3);
            if (w == 0) {
                const gptr<float> gr = sbase(hr.gi + row * a.B * 16);
#pragma unroll
                for (int m = 0; m < NZM; ++m) stg<float>(gr, offS + 16u * m, gs[m]);
            }
        }
        return gx;
    };
    // dL/dis[k] enters through the `s`-block slot of its i-dim (one slot per i-dim)
    auto add_gis = [&](const long long k, float (&gs)[NZM]) {
        // grad_is == NULL (no loss term on the algebraic outputs) reads the rows of `is` instead -- same shape -- and masks them out:
        // NO branch here.  Rounds 2-3 had `if (a.gis) { .. }`: the compiler moved that uniform branch between the last MFMA of the head's
        // third layer and the v_pk_add that sums its two accumulator chains; the hazard recognizer pads the fall-through path with
        // s_nop but not the taken edge, so with NULL the add read registers 2..3 of the accumulator one instruction behind the MFMA that
        // writes them (wrong a3 in half of the units -> wrong AE gradients; profiles/r04_defect_b_*.txt).
        const bool has = a.gis != nullptr;
        const gptr<const float> row = sbase((has ? a.gis : a.is) + k * a.B * idim);
#pragma unroll
        for (int m = 0; m < NZM; ++m) {
            const float q = ldg<float>(row, offI + 4u * (ekind[m] == 2 ? ecol[m] : 0));
            gs[m] += (has && ekind[m] == 2 && 4 * m + g >= ne) ? q : 0.0f;
        }
    };

    const HeadRows grid_rows{a.aact[0], a.aact[1], a.aact[2], a.adelta[0], a.adelta[1], a.adelta[2], a.agi};
    const HeadRows event_rows{a.eact[0], a.eact[1], a.eact[2], a.edelta[0], a.edelta[1], a.edelta[2], a.egi};
    float gcar[NX], gsl[NZM];
#pragma unroll
    for (int r = 0; r < NX; ++r) gcar[r] = (4 * r + g < xd) ? a.carry_x[b * xd + 4 * r + g] : 0.0f;
#pragma unroll
    for (int m = 0; m < NZM; ++m) gsl[m] = a.carry_i[b * 16 + 4 * m + g];

    const long long nrow = a.B;
    for (long long k = a.k1 - 1; k >= a.k0; --k) {
        const int ev = a.ev ? __builtin_amdgcn_readfirstlane(a.ev[k]) : -1;
        // ---- (1) AE head at grid point k+1 (my_solvers.py:121): adjoint = dL/dis[k+1] + the algebraic adjoint of step k+1's DE
        {
            float x1[NX];
            load_x2(a.xs, k + 1, x1);
            f4 a1, a2, a3;
            ae_hidden(x1, k + 1, -1, a1, a2, a3);
            add_gis(k + 1, gsl);
            const f2 gxa = ae_adjoint(a1, a2, a3, gsl, grid_rows, (size_t)(k + 1 - a.k0));
            gcar[0] += gxa[0];
            if constexpr (NX > 1) gcar[1] += gxa[1];
        }
        // ---- (2) DE step k
        float x0[NX], gin[NX], ext[NZM];
        load_x2(a.xs, k, x0);
        load_x2(a.gxs, k + 1, gin);
        {
            const RowZV zr = zv_rows(k, ev);
            const gptr<const float> irow = sbase(a.is + k * a.B * idim);
#pragma unroll
            for (int m = 0; m < NZM; ++m) {
                ext[m] = zv_val(zr, ekind[m], ecol[m]);
                if (ekind[m] == 2) ext[m] = ldg<float>(irow, offI + 4u * ecol[m]);
            }
        }
        if (ev >= 0) {   // event: i0 = g(x0; z_jump, v_jump) (my_solvers.py:108-110); its rows travel through the event buffers
            f4 e1, e2, e3;
            ae_hidden(x0, k, ev, e1, e2, e3);
            f4 sb4 = ab4;
            float sw4[4] = {aw4[0], aw4[1], aw4[2], aw4[3]};
            if constexpr (STREAM) {     // event steps are rare: their output layer is read where it is used, not kept for the whole launch
                const float* pwo = pwa;
                asm volatile("" : "+v"(pwo));
                const gptr<const float> pws = (gptr<const float>)pwo;
#pragma unroll
                for (int r = 0; r < 4; ++r) { sb4[r] = pws[(RA::B4 + r) * 64]; sw4[r] = pws[(RA::W4 + r) * 64]; }
            }
            const f4 i0 = out4(sw4, e3, sb4);
#pragma unroll
            for (int m = 0; m < NZM; ++m) if (ekind[m] == 2) ext[m] = i0[m];
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
        const float h_ = ldg<float>(sbase(a.t.p + (k + 1) * a.t.st), offT) - ldg<float>(sbase(a.t.p + k * a.t.st), offT);
        f4 cz = c0;
#pragma unroll
        for (int m = 0; m < NZM; ++m) cz = wm4(w1z[m], ext[m] - a0e[m], cz);

        // phase A: stage evaluations (K1's plan; the ELU outputs are kept)
        float X[S][NX], ks[S][NX];
        f4 h1[STREAM ? 1 : S], h2[STREAM ? 1 : S], h3[STREAM ? 1 : S];
        f4 la1, la2, la3;    // ELU outputs of the last stage evaluated (the first one the backward half needs)
        auto row_blk = [&](const int s) -> size_t { return (size_t)((k - a.k0) * S + s) * nrow; };   // uniform
        {
            float w2s[STREAM ? 4 * NWV : 1], w3s[STREAM ? 4 * NWV : 1];
            f4 sb2 = b2, sb3 = b3, sb4 = b4;
            if constexpr (STREAM) {
                const float* pwo = pw;
                asm volatile("" : "+v"(pwo));
                const gptr<const float> pws = (gptr<const float>)pwo;
#pragma unroll
                for (int q = 0; q < 4 * NWV; ++q) { w2s[q] = pws[(RD::W2 + q) * 64]; w3s[q] = pws[(RD::W3 + q) * 64]; }
#pragma unroll
                for (int r = 0; r < 4; ++r) { sb2[r] = pws[(RD::B2 + r) * 64]; sb3[r] = pws[(RD::B3 + r) * 64]; sb4[r] = pws[(RD::B4 + r) * 64]; }
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
                if constexpr (STREAM) { a2 = elu_quad(mid(w2s, sb2, a1)); a3 = elu_quad(mid(w3s, sb3, a2)); }
                else { a2 = elu_quad(mid(w2r, b2, a1)); a3 = elu_quad(mid(w3r, b3, a2)); }
                if (valid) {
                    const size_t rb = row_blk(s) * H;
                    stg<f4>(sbase(a.act[0] + rb), offH, a1);
                    stg<f4>(sbase(a.act[1] + rb), offH, a2);
                    stg<f4>(sbase(a.act[2] + rb), offH, a3);
                    if (w == 0) {     // the stage's state input goes out here, so that X[] does not have to live through the backward half
                        const gptr<float> xsr = sbase(a.xst + row_blk(s) * xd);
#pragma unroll
                        for (int r = 0; r < NX; ++r) if (4 * r + g < xd) stg<float>(xsr, offX + 16u * r, X[s][r]);
                    }
                }
                if constexpr (!STREAM) { h1[s] = a1; h2[s] = a2; h3[s] = a3; }
                if (s == S - 1) { la1 = a1; la2 = a2; la3 = a3; }
                const f2 kk = out2(w4, a3, sb4);
                ks[s][0] = kk[0];
                if constexpr (NX > 1) ks[s][1] = kk[1];
            }
        }
        // phase B: stages backwards; D1 = sum over the stages of delta1 (the external inputs are frozen over the step)
        float gks[S][NX], gx0[NX];
#pragma unroll
        for (int r = 0; r < NX; ++r) {
            const float g1 = gcar[r] + gin[r];
            gx0[r] = g1;
#pragma unroll
            for (int s = 0; s < S; ++s) gks[s][r] = (h_ * rk_b(METHOD, s)) * g1;
        }
        f4 D1 = f4{0.f, 0.f, 0.f, 0.f}, D2 = D1, D3 = D1;
        f4 na1 = la1, na2 = la2, na3 = la3;       // STREAM: rows of the stage handled next, requested one stage ahead
#pragma unroll
        for (int s = S - 1; s >= 0; --s) {
            f4 a1, a2, a3;
            if constexpr (STREAM) {
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
            const f4 d2 = midT(NWV * NWV * 64, d3) * wdact(a2);
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
#pragma unroll
                    for (int r = 0; r < NX; ++r) if (4 * r + g < xd) stg<float>(gkr, offX + 16u * r, gks[s][r]);
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
        const f4 gE = out4(fE, D1, f4{0.f, 0.f, 0.f, 0.f});      // adjoint of the algebraic inputs of this step, slot layout
#pragma unroll
        for (int r = 0; r < NX; ++r) gcar[r] = gx0[r];
#pragma unroll
        for (int m = 0; m < NZM; ++m) gsl[m] = gE[m];
        // ---- (3) event: that adjoint belongs to the recomputed i0, whose head is run backwards here; grid point k's own head
        //          (is[k], un-jumped) then only sees dL/dis[k]
        if (ev >= 0) {
            const size_t rb = (size_t)ev * a.B * H;                              // this lane's own rows, written above
            const f4 e1 = ldg<f4>(sbase(a.eact[0] + rb), offH);
            const f4 e2 = ldg<f4>(sbase(a.eact[1] + rb), offH);
            const f4 e3 = ldg<f4>(sbase(a.eact[2] + rb), offH);
            const f2 gxa = ae_adjoint(e1, e2, e3, gsl, event_rows, (size_t)ev);
            gcar[0] += gxa[0];
            if constexpr (NX > 1) gcar[1] += gxa[1];
#pragma unroll
            for (int m = 0; m < NZM; ++m) gsl[m] = 0.0f;
        }
    }
    if (a.k0 == 0) {   // the head at grid point 0 (my_solvers.py:95): i_0 = g(x_init; z_0, v_0)
        float x1[NX];
        load_x2(a.xs, 0, x1);
        f4 a1, a2, a3;
        ae_hidden(x1, 0, -1, a1, a2, a3);
        add_gis(0, gsl);
        const f2 gxa = ae_adjoint(a1, a2, a3, gsl, grid_rows, (size_t)0);
        gcar[0] += gxa[0];
        if constexpr (NX > 1) gcar[1] += gxa[1];
#pragma unroll
        for (int m = 0; m < NZM; ++m) gsl[m] = 0.0f;
    }
    if (w == 0 && valid) {
#pragma unroll
        for (int r = 0; r < NX; ++r) if (4 * r + g < xd) stg<float>(sbase(a.carry_x), offX + 16u * r, gcar[r]);
#pragma unroll
        for (int m = 0; m < NZM; ++m) stg<float>(sbase(a.carry_i), offS + 16u * m, gsl[m]);
    }
}

int wide_hidden(const psnode_mlp_f32& m) {
    if (m.n_layers != 4) return 0;
    const int h = m.out_dim[0];
    if (m.out_dim[1] != h || m.out_dim[2] != h) return 0;
    return padded_hidden(h);      // the width class the kernel runs at (zero-padded units beyond h)
}
size_t wide_fwd_floats(int nw, int n) { return (((size_t)nw * (max_regs(nw) + (n + 3) / 4) * 64 + 63) / 64) * 64; }
size_t wide_t_floats(int nw) { return (size_t)2 * nw * nw * 64 * 4; }

template <int METHOD, int NWV>
hipError_t launch_wide(const WideDaeDev& a, int NZM, int NZA, const float* pde, const float* pae, const f4* pt, const f4* pta, int NA,
                       hipStream_t s) {
    const dim3 grid((unsigned)((a.B + TBM - 1) / TBM)), block(64 * NWV);
    const size_t lds = wide_t_floats(NWV) * sizeof(float) * (aet_in_lds(NWV) ? 2 : 1);
#define PSNODE_WIDE(NZM_, NZA_)                                                                                                 \
    {                                                                                                                           \
        auto kern = &dae_backward_wide_kernel<METHOD, NZM_, NZA_, NWV>;                                                         \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return e;                                                                                          \
        hipLaunchKernelGGL(kern, grid, block, lds, s, a, pde, pae, pt, pta, NA);                                                \
        return hipGetLastError();                                                                                               \
    }
    switch (NZM * 10 + NZA) {
        case 11: PSNODE_WIDE(1, 1)
        case 21: PSNODE_WIDE(2, 1)
        case 31: PSNODE_WIDE(3, 1)
        case 41: PSNODE_WIDE(4, 1)
        case 32: PSNODE_WIDE(3, 2)
        case 42: PSNODE_WIDE(4, 2)
        default: return hipErrorNotSupported;
    }
#undef PSNODE_WIDE
}

template <int NWV>
hipError_t launch_wide_method(const WideDaeDev& a, int NZM, int NZA, const float* pde, const float* pae, const f4* pt, const f4* pta,
                              int NA, hipStream_t s) {
    switch (a.method) {
        case PSNODE_EULER: return launch_wide<PSNODE_EULER, NWV>(a, NZM, NZA, pde, pae, pt, pta, NA, s);
        case PSNODE_MIDPOINT: return launch_wide<PSNODE_MIDPOINT, NWV>(a, NZM, NZA, pde, pae, pt, pta, NA, s);
        default: return launch_wide<PSNODE_RK4_38, NWV>(a, NZM, NZA, pde, pae, pt, pta, NA, s);
    }
}

}  // namespace
}  // namespace psnode

using namespace psnode;

extern "C" {

int32_t psnode_dae_backward_wide_supported(const psnode_dae_bwd_wide_args_f32* a) {
    if (!a || a->method < PSNODE_EULER || a->method > PSNODE_RK4_38) return 0;
    const int xd = a->x_dim, zd = a->z_dim, vd = a->v_dim, id = a->i_dim;
    if (xd < 1 || xd > 4 * kNXc || zd < 0 || vd < 0 || id < 1) return 0;
    const int nzv = zd + vd, ne = nzv + id, n = xd + ne;
    if (ne > 8) return 0;
    const int NZM = (2 * ne + 3) / 4, NZA = (nzv + 3) / 4;
    if (NZA < 1 || NZA > 2 || NZM > kMaxNZM || (NZA == 2 && NZM < 3)) return 0;
    const int h = wide_hidden(a->de);
    if (!h || wide_hidden(a->ae) != h || a->ae.out_dim[0] != a->de.out_dim[0]) return 0;
    return a->de.in_dim == 3 * n && a->de.out_dim[3] == xd && a->ae.in_dim == n + xd + nzv && a->ae.out_dim[3] == id;
}

size_t psnode_dae_backward_wide_ae_floats(const psnode_dae_bwd_wide_args_f32* a) {
    if (!psnode_dae_backward_wide_supported(a) || !a->grad_params_de) return 0;
    return dae_fused_bwd_ae_floats(a);
}

size_t psnode_dae_backward_wide_workspace_bytes(const psnode_dae_bwd_wide_args_f32* a) {
    if (!psnode_dae_backward_wide_supported(a)) return 0;
    if (a->grad_params_de) return dae_fused_bwd_workspace_floats(a) * sizeof(float);
    const int nw = wide_hidden(a->de) / 16, n = a->x_dim + a->z_dim + a->v_dim + a->i_dim;
    return (2 * wide_fwd_floats(nw, n) + 2 * wide_t_floats(nw) + 128) * sizeof(float);
}

int32_t psnode_dae_backward_wide_f32(const psnode_dae_bwd_wide_args_f32* p, void* workspace, size_t workspace_bytes, void* stream) {
    if (!p) return PSNODE_ERR_NULL;
    if (p->method < PSNODE_EULER || p->method > PSNODE_RK4_38) return PSNODE_ERR_METHOD;
    if (!psnode_dae_backward_wide_supported(p)) return PSNODE_ERR_UNSUPPORTED;
    if (p->T < 2 || p->B < 1 || p->k0 < 0 || p->k1 <= p->k0 || p->k1 > p->T - 1) return PSNODE_ERR_DIMS;
    for (int l = 0; l < 4; ++l) if (!p->de.weight[l] || !p->de.bias[l] || !p->ae.weight[l] || !p->ae.bias[l]) return PSNODE_ERR_NULL;
    const bool fused = p->grad_params_de != nullptr;
    if ((p->flags & ~(PSNODE_FLAG_INPUT_TRUE_X | PSNODE_FLAG_INPUT_TRUE_I)) || (p->flags && !fused)) return PSNODE_ERR_UNSUPPORTED;   // teacher forcing: K7f only
    const bool ae_in_kernel = fused && dae_fused_bwd_ae_floats(p) > 0;      // hidden <= 64: no head rows are written at all
    if (!p->t.ptr || !p->all_initial || !p->xs || !p->is || !p->grad_xs || !p->carry_x || (!ae_in_kernel && !p->ae_gi)) return PSNODE_ERR_NULL;
    if (fused) {
        if (p->k0 != 0 || p->k1 != p->T - 1) return PSNODE_ERR_DIMS;
        const bool sv = p->saved_act != nullptr;
        if ((p->saved_xstage != nullptr) != sv || (p->saved_ae_act != nullptr) != sv) return PSNODE_ERR_NULL;
        if (sv && p->event_idx && (!p->saved_ev_act || !p->saved_ev_i)) return PSNODE_ERR_NULL;
        if (!p->grad_all_initial_de || (p->z_dim + p->v_dim > 0 && (!p->grad_zv || (p->event_idx && !p->grad_jump)))) return PSNODE_ERR_NULL;
    } else if (!p->carry_i || !p->gk || !p->xstage) {
        return PSNODE_ERR_NULL;
    }
    for (int l = 0; l < 3; ++l) {
        if (!ae_in_kernel && (!p->ae_act[l] || !p->ae_delta[l])) return PSNODE_ERR_NULL;
        if (!fused && (!p->act[l] || !p->delta[l] || !p->dsum[l])) return PSNODE_ERR_NULL;
    }
    if ((p->z_dim > 0 && !p->z.ptr) || (p->v_dim > 0 && !p->v.ptr)) return PSNODE_ERR_NULL;
    if (p->event_idx) {
        if (p->n_events < 1 || (p->z_dim > 0 && !p->z_jump) || (p->v_dim > 0 && !p->v_jump) || !p->ev_i) return PSNODE_ERR_NULL;
        if (!ae_in_kernel && !p->ev_gi) return PSNODE_ERR_NULL;
        for (int l = 0; l < 3; ++l) if (!p->ev_act[l] || (!ae_in_kernel && !p->ev_delta[l])) return PSNODE_ERR_NULL;
    }
    if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 255u) || workspace_bytes < psnode_dae_backward_wide_workspace_bytes(p))
        return PSNODE_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int H = wide_hidden(p->de), nw = H / 16, xd = p->x_dim, zd = p->z_dim, vd = p->v_dim, id = p->i_dim;
    {   // per-lane offsets inside a row are 32-bit byte offsets next to a scalar row base
        const int64_t lim = (int64_t)1 << 30, Bm = p->B;
        const int64_t sb[] = {H, p->t.stride_b, zd > 0 ? p->z.stride_b : 0, vd > 0 ? p->v.stride_b : 0,
                              p->event_idx && zd > 0 ? p->zj_stride_b : 0, p->event_idx && vd > 0 ? p->vj_stride_b : 0};
        for (int64_t q : sb) if (q < 0 || Bm * q + 64 >= lim) return PSNODE_ERR_DIMS;
    }
    if (fused) return dae_fused_bwd_launch(p, static_cast<float*>(workspace), s);
    const int nzv = zd + vd, ne = nzv + id, n = xd + ne;
    const int NZM = (2 * ne + 3) / 4, NZA = (nzv + 3) / 4, NA = (n + 3) / 4;
    float* pde = static_cast<float*>(workspace);
    float* pae = pde + wide_fwd_floats(nw, n);
    f4* pt = reinterpret_cast<f4*>(pae + wide_fwd_floats(nw, n));
    f4* pta = pt + wide_t_floats(nw) / 4;
    PackMfma f;
    memset(&f, 0, sizeof(f));
    f.ae = 0; f.nw = nw; f.xd = xd; f.ne = ne; f.n = n; f.nzv = nzv; f.NX = kNXc; f.NB = 0; f.NE = NZM; f.NA = NA; f.fold = 1;
    f.hreal = p->de.out_dim[0];
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
    PackWideT t{nw, p->de.out_dim[0], p->de.weight[1], p->de.weight[2], pt};
    hipLaunchKernelGGL(pack_wide_t_kernel, dim3(64), dim3(256), 0, s, t);
    PackWideT ta{nw, p->de.out_dim[0], p->ae.weight[1], p->ae.weight[2], pta};
    hipLaunchKernelGGL(pack_wide_t_kernel, dim3(64), dim3(256), 0, s, ta);
    if (hipGetLastError() != hipSuccess) return PSNODE_ERR_HIP;
    WideDaeDev a;
    memset(&a, 0, sizeof(a));
    a.method = p->method; a.xd = xd; a.zd = zd; a.vd = vd; a.id = id; a.hreal = p->de.out_dim[0]; a.T = p->T; a.B = p->B; a.k0 = p->k0; a.k1 = p->k1;
    a.w1 = p->de.weight[0]; a.w4 = p->de.weight[3]; a.aw1 = p->ae.weight[0]; a.aw4 = p->ae.weight[3];
    a.t = ViewDev{p->t.ptr, p->t.stride_t, p->t.stride_b};
    a.z = ViewDev{p->z.ptr, p->z.stride_t, p->z.stride_b};
    a.v = ViewDev{p->v.ptr, p->v.stride_t, p->v.stride_b};
    a.a0 = p->all_initial; a.ev = p->event_idx;
    a.zj = p->z_jump; a.zjb = p->zj_stride_b; a.zje = p->zj_stride_e;
    a.vj = p->v_jump; a.vjb = p->vj_stride_b; a.vje = p->vj_stride_e;
    a.xs = p->xs; a.is = p->is; a.gxs = p->grad_xs; a.gis = p->grad_is;
    a.carry_x = p->carry_x; a.carry_i = p->carry_i;
    for (int l = 0; l < 3; ++l) {
        a.act[l] = p->act[l]; a.delta[l] = p->delta[l]; a.dsum[l] = p->dsum[l];
        a.aact[l] = p->ae_act[l]; a.adelta[l] = p->ae_delta[l];
        a.eact[l] = p->ev_act[l]; a.edelta[l] = p->ev_delta[l];
    }
    a.gk = p->gk; a.xst = p->xstage; a.agi = p->ae_gi; a.egi = p->ev_gi; a.ei = p->ev_i;
    hipError_t e;
    switch (nw) {
        case 2: e = launch_wide_method<2>(a, NZM, NZA, pde, pae, pt, pta, NA, s); break;
        case 4: e = launch_wide_method<4>(a, NZM, NZA, pde, pae, pt, pta, NA, s); break;
        default: e = launch_wide_method<8>(a, NZM, NZA, pde, pae, pt, pta, NA, s); break;
    }
    if (e == hipErrorNotSupported) return PSNODE_ERR_UNSUPPORTED;
    return e == hipSuccess ? PSNODE_OK : PSNODE_ERR_HIP;
}

}  // extern "C"
