// K8 -- backward (discretise-then-optimise) pass through the latent ODE integrator of the direct_encode variant at
// hidden_dim 16 (K3a's shape: DE = Linear(6H,H) ELU Linear(H,H), state Xh[H], external Zh[H];
// neural_00_ODE_02_direct_encode.py:49-57, trained by :267-275).  Replaces what loss.backward() does on the unrolled
// integrate_ODE graph between the encoders and the decoder.
//
// As in K3a one wave owns a tile of 16 trajectories end to end: no barrier, no cross-wave exchange; the only LDS use
// is a private padded 16x16 transpose tile per operand (weight gradients contract over the tile's 16 trajectories,
// so delta and activation tiles are needed with the trajectory index on the K axis).  Per step, backwards:
//   phase A  stage evaluations from the saved xs[k] (stage inputs need k1..k3), h1 and the stage inputs kept;
//   phase B  per stage: delta1 = (W2^T gk) * ELU'(pre1), gx = (W1s+W1d)[:, x]^T delta1, RK adjoint; dW2 += gk (x) h1,
//            dW1[:, s-columns of x] += delta1 (x) xs.  The external block is frozen over the stages, so its gradient
//            and its dW1 columns use sum_s(delta1) once per step.  The a0 and (s-a0) columns of dW1 and d all_initial
//            are reconstructed once at the end from sum_t(delta1) (a0 is constant over time).
// Layout: D row 4g+r of every tile is unit/dim 4g+r and the K order of a 16-wide block is column 4g+m for MFMA m
// (K3a's convention): a lane's four D registers are its four B operands of the next product, and every global access of
// a lane (state, z block, gradients) is one aligned float4.
#define PSNODE_ELU_LITERALS   // register-bound kernels: ELU coefficients as literals, not as 8 resident VGPRs (psnode_common.h)
#include <string.h>

#include "psnode_common.h"

namespace psnode {
namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int LH = 16, LTB = 16, LN = 2 * LH, LK1 = 3 * LN;   // hidden, tile, all_initial width, in_features
constexpr int LSCR = 64 * 4 + 4 * 8;                           // padded transpose tile (floats)

__device__ __forceinline__ f4 km(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f4 kelu4(f4 v) { return elu_quad(v); }
__device__ __forceinline__ f4 kdact(f4 h) {
    return elu_grad_quad(h);
}
__device__ __forceinline__ f4 kz4() { return f4{0.f, 0.f, 0.f, 0.f}; }

// The ODE case (Linear(96,16) ELU Linear(16,16)) moved to K8f in psnode_latent_dpp.hip (lane = (trajectory, unit), VALU + DPP
// row broadcasts, 4 trajectories per wave): this file keeps the single-wave MFMA backward of the latent DAE.

// ---------------------------------------------------------------------------------------------------------------
// K8 (DAE): the same single-wave scheme for the latent DAE at hidden 16 (K3a's DAE shapes: blocks x | [z] | v | i of width 16,
// DE = Linear(12H|9H,H) ELU Linear(H,H), AE = Linear(7H|5H,H) ELU Linear(H,H)).  Every matrix is a set of 16x16 blocks held in
// three register forms: forward (Blk[i][4g+m]), transposed (Blk[4g+m][i]) and a gradient accumulator tile; F_b = Ws_b + Wd_b is
// folded as in K3c/K9.  Control flow as K7/K9: AE head VJP per grid point, DE stages with frozen external blocks, event
// recompute + chained VJP, jump gradients.  The weights are read straight from the nn.Linear tensors (no packed image).
struct LatentDaeBwdDev {
    IntegrateDev a;          // t, z, v, a0, ev, zj, vj (+strides), T, B, zd, method; a.de / a.ae weight pointers (w[], bias[])
    const float *xs, *is_, *gxs, *gis;
    float *gx0, *gz, *gv, *gzj, *gvj, *ga0, *wpart;
    int n_events;
};

struct Q4 { float m[4]; };

template <int METHOD, int NBE>
__global__ __launch_bounds__(64) void latent16_dae_backward_kernel(const LatentDaeBwdDev d) {
    constexpr int S = rk_stages(METHOD);
    constexpr int NBLK = 1 + NBE, NZV = NBE - 1, NAE = NBE, n = LH * NBLK, K1 = 3 * n, K1A = n + LH * NAE;
    constexpr int NPD = LH * K1 + LH + LH * LH + LH, NPA = LH * K1A + LH + LH * LH + LH;
    const IntegrateDev& a = d.a;
    __shared__ __attribute__((aligned(16))) float scr[4][LSCR];
    const int l = threadIdx.x, g = l >> 4, j = l & 15, i = j;
    const long long b0 = (long long)blockIdx.x * LTB;
    const bool valid = b0 + j < a.B;
    const long long b = valid ? b0 + j : a.B - 1;

    // ---- blocks -> registers
    const float *w1 = a.de.w[0], *w2p = a.de.w[1], *aw1 = a.ae.w[0], *aw2p = a.ae.w[1];
    Q4 F[NBLK], FT[NBLK], AF[NAE], AFT[NAE], W2, W2T, AW2, AW2T;
    f4 b1r, b2r, ab1r, ab2r;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int blk = 0; blk < NBLK; ++blk) {
            const float* rf = w1 + i * K1 + LH * blk + 4 * g + m;             // row i, column 4g+m of block blk
            F[blk].m[m] = rf[2 * n] + rf[n];
            const float* rt = w1 + (4 * g + m) * K1 + LH * blk + i;           // row 4g+m, column i
            FT[blk].m[m] = rt[2 * n] + rt[n];
        }
#pragma unroll
        for (int bb = 0; bb < NAE; ++bb) {
            AF[bb].m[m] = aw1[i * K1A + n + LH * bb + 4 * g + m];
            AFT[bb].m[m] = aw1[(4 * g + m) * K1A + n + LH * bb + i];
        }
        W2.m[m] = w2p[i * LH + 4 * g + m]; W2T.m[m] = w2p[(4 * g + m) * LH + i];
        AW2.m[m] = aw2p[i * LH + 4 * g + m]; AW2T.m[m] = aw2p[(4 * g + m) * LH + i];
        b1r[m] = a.de.bias[0][4 * g + m]; b2r[m] = a.de.bias[1][4 * g + m];
        ab1r[m] = a.ae.bias[0][4 * g + m]; ab2r[m] = a.ae.bias[1][4 * g + m];
    }
    auto mat = [&](const Q4& wq, const f4 v, const f4 init) -> f4 {
        f4 accA = km(wq.m[0], v[0], init), accB = km(wq.m[1], v[1], kz4());
        accA = km(wq.m[2], v[2], accA);
        accB = km(wq.m[3], v[3], accB);
        return accA + accB;
    };
    // c0 = b1 + sum_blk A0_blk . a0_blk   (A0 = Wa0 - Wd for the DE, Wa0 for the AE)
    f4 c0 = b1r, c0a = ab1r;
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) {
        const f4 a0v = *reinterpret_cast<const f4*>(a.a0 + b * n + LH * blk + 4 * g);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float* rf = w1 + i * K1 + LH * blk + 4 * g + m;
            c0 = km(rf[0] - rf[n], a0v[m], c0);
            c0a = km(aw1[i * K1A + LH * blk + 4 * g + m], a0v[m], c0a);
        }
    }
    auto put_tile = [&](const int q, const f4 v) { *reinterpret_cast<f4*>(scr[q] + 4 * l + 8 * g) = v; };
    auto get_tile = [&](const int q) -> f4 {
        const float* s = scr[q] + 4 * (16 * (i >> 2) + g) + 8 * (i >> 2) + (i & 3);
        return f4{s[0], s[16], s[32], s[48]};
    };
    auto tr = [&](const int q, const f4 v) -> f4 { put_tile(q, v); return get_tile(q); };
    auto outer = [&](f4& acc, const f4 aT, const f4 bT) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc = km(aT[kk], bT[kk], acc);
    };

    const long long tst = a.t.st, nT = a.T;
    const float* tp = a.t.p + b * a.t.sb;
    const bool has_z = a.zd > 0;
    const float* vbase = a.v.p + b * a.v.sb;
    const float* vjbase = a.vj + b * a.vjb;
    const float* sp[2] = {has_z ? a.z.p + b * a.z.sb : vbase, vbase};
    const long long sst[2] = {has_z ? a.z.st : a.v.st, a.v.st};
    const float* jp[2] = {has_z ? a.zj + b * a.zjb : vjbase, vjbase};
    const long long jse[2] = {has_z ? a.zje : a.vje, a.vje};
    float* gdst[2] = {has_z ? d.gz : d.gv, d.gv};
    float* gjdst[2] = {has_z ? d.gzj : d.gvj, d.gvj};
    auto load_zv = [&](const int s, const long long k, const int ev) -> f4 {
        return *reinterpret_cast<const f4*>((ev >= 0 ? jp[s] + ev * jse[s] : sp[s] + k * sst[s]) + 4 * g);
    };
    auto row_of = [&](const float* base, const long long k) -> f4 { return *reinterpret_cast<const f4*>(base + (k * a.B + b) * LH + 4 * g); };
    auto store_zv = [&](const int s, const long long grid, const int ev, const f4 val) {
        if (!valid) return;
        if (ev >= 0) { if (gjdst[s]) *reinterpret_cast<f4*>(gjdst[s] + (b * d.n_events + ev) * LH + 4 * g) = val; }
        else if (gdst[s]) *reinterpret_cast<f4*>(gdst[s] + (grid * a.B + b) * LH + 4 * g) = val;
    };

    f4 accF[NBLK], accAF[NAE], accW2 = kz4(), accW2a = kz4(), S1 = kz4(), SB2 = kz4(), AS1 = kz4(), ASB2 = kz4();
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) accF[blk] = kz4();
#pragma unroll
    for (int bb = 0; bb < NAE; ++bb) accAF[bb] = kz4();

    struct ZV { f4 b[NZV]; };
    f4 ah1 = kz4();
    auto ae_hidden = [&](const f4 xo, const ZV& zv) {
        f4 acc = mat(AF[0], xo, c0a);
#pragma unroll
        for (int s = 0; s < NZV; ++s) acc = mat(AF[1 + s], zv.b[s], acc);
        ah1 = kelu4(acc);
    };
    auto ae_vjp = [&](const f4 xo, const ZV& zv, const f4 gi, ZV& gzv) -> f4 {
        ae_hidden(xo, zv);
        ASB2 += gi;
        const f4 d1 = mat(AW2T, gi, kz4()) * kdact(ah1);
        AS1 += d1;
        outer(accW2a, tr(0, gi), tr(1, ah1));
        const f4 dT = tr(0, d1);
        outer(accAF[0], dT, tr(1, xo));
        const f4 gx = mat(AFT[0], d1, kz4());
#pragma unroll
        for (int s = 0; s < NZV; ++s) {
            outer(accAF[1 + s], dT, tr(1, zv.b[s]));
            gzv.b[s] = mat(AFT[1 + s], d1, kz4());
        }
        return gx;
    };

    f4 gcarry = kz4(), gicarry = kz4();
    ZV dezv;
#pragma unroll
    for (int s = 0; s < NZV; ++s) dezv.b[s] = kz4();

    for (long long jg = nT - 1; jg >= 0; --jg) {
        f4 g1 = gcarry + (valid ? row_of(d.gxs, jg) : kz4());
        {
            const f4 xj = row_of(d.xs, jg);
            ZV zvj, gzv;
#pragma unroll
            for (int s = 0; s < NZV; ++s) zvj.b[s] = load_zv(s, jg, -1);
            const f4 gi = gicarry + ((d.gis && valid) ? row_of(d.gis, jg) : kz4());
            g1 += ae_vjp(xj, zvj, gi, gzv);
#pragma unroll
            for (int s = 0; s < NZV; ++s) store_zv(s, jg, -1, dezv.b[s] + gzv.b[s]);
        }
        if (jg == 0) { gcarry = g1; break; }
        const long long k = jg - 1;
        const int ev = a.ev ? a.ev[k] : -1;
        const float h_ = tp[jg * tst] - tp[k * tst];
        const f4 x0 = row_of(d.xs, k);
        f4 ext[NBE];
#pragma unroll
        for (int s = 0; s < NZV; ++s) ext[s] = load_zv(s, k, ev);
        if (ev >= 0) {
            ZV zq;
#pragma unroll
            for (int s = 0; s < NZV; ++s) zq.b[s] = ext[s];
            ae_hidden(x0, zq);
            ext[NBE - 1] = mat(AW2, ah1, ab2r);
        } else {
            ext[NBE - 1] = row_of(d.is_, k);
        }
        f4 cz = c0;
#pragma unroll
        for (int e = 0; e < NBE; ++e) cz = mat(F[1 + e], ext[e], cz);

        f4 xst[S], h1[S], ks[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            f4 acc = kz4();
#pragma unroll
            for (int jj = 0; jj < s; ++jj) acc += rk_a(METHOD, s, jj) * ks[jj];
            xst[s] = s == 0 ? x0 : x0 + h_ * acc;
            h1[s] = kelu4(mat(F[0], xst[s], cz));
            ks[s] = mat(W2, h1[s], b2r);
        }
        f4 gks[S], gx0 = g1, D1 = kz4();
#pragma unroll
        for (int s = 0; s < S; ++s) gks[s] = (h_ * rk_b(METHOD, s)) * g1;
#pragma unroll
        for (int s = S - 1; s >= 0; --s) {
            const f4 gk = gks[s];
            SB2 += gk;
            const f4 d1 = mat(W2T, gk, kz4()) * kdact(h1[s]);
            D1 += d1;
            outer(accW2, tr(0, gk), tr(1, h1[s]));
            outer(accF[0], tr(2, d1), tr(3, xst[s]));
            const f4 gx = mat(FT[0], d1, kz4());
            gx0 += gx;
#pragma unroll
            for (int jj = 0; jj < s; ++jj) gks[jj] += (h_ * rk_a(METHOD, s, jj)) * gx;
        }
        S1 += D1;
        const f4 DT = tr(0, D1);
        f4 gext[NBE];
#pragma unroll
        for (int e = 0; e < NBE; ++e) {
            outer(accF[1 + e], DT, tr(1 + (e & 1), ext[e]));
            gext[e] = mat(FT[1 + e], D1, kz4());
        }
        if (ev >= 0) {
            ZV zq, gq;
#pragma unroll
            for (int s = 0; s < NZV; ++s) zq.b[s] = ext[s];
            gx0 += ae_vjp(x0, zq, gext[NBE - 1], gq);
#pragma unroll
            for (int s = 0; s < NZV; ++s) { store_zv(s, k, ev, gext[s] + gq.b[s]); dezv.b[s] = kz4(); }
            gicarry = kz4();
        } else {
#pragma unroll
            for (int s = 0; s < NZV; ++s) dezv.b[s] = gext[s];
            gicarry = gext[NBE - 1];
        }
        gcarry = gx0;
    }

    // ---- epilogue
    if (valid) *reinterpret_cast<f4*>(d.gx0 + b * LH + 4 * g) = gcarry;
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) {       // d all_initial block = A0_blk^T sum_t(delta1)  (DE + AE)
        Q4 at, aat;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float* rt = w1 + (4 * g + m) * K1 + LH * blk + i;
            at.m[m] = rt[0] - rt[n];
            aat.m[m] = aw1[(4 * g + m) * K1A + LH * blk + i];
        }
        const f4 ga = mat(at, S1, kz4()) + mat(aat, AS1, kz4());
        if (valid) *reinterpret_cast<f4*>(d.ga0 + b * n + LH * blk + 4 * g) = ga;
    }
    float* wp = d.wpart + (size_t)blockIdx.x * (NPD + NPA);
    const f4 sT = tr(0, S1), asT = tr(1, AS1);
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) {
        f4 ca0 = kz4(), ca0a = kz4();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const long long tb = b0 + 4 * kk + g;
            const float av = tb < a.B ? a.a0[tb * n + LH * blk + j] : 0.0f;
            ca0 = km(sT[kk], av, ca0);
            ca0a = km(asT[kk], av, ca0a);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float* row = wp + (4 * g + r) * K1 + LH * blk + j;
            row[0] = ca0[r];
            row[n] = accF[blk][r] - ca0[r];
            row[2 * n] = accF[blk][r];
            wp[NPD + (4 * g + r) * K1A + LH * blk + j] = ca0a[r];
        }
    }
#pragma unroll
    for (int bb = 0; bb < NAE; ++bb)
#pragma unroll
        for (int r = 0; r < 4; ++r) wp[NPD + (4 * g + r) * K1A + n + LH * bb + j] = accAF[bb][r];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        wp[LH * K1 + LH + (4 * g + r) * LH + j] = accW2[r];
        wp[NPD + LH * K1A + LH + (4 * g + r) * LH + j] = accW2a[r];
    }
    f4 sb1 = S1, sb2 = SB2, asb1 = AS1, asb2 = ASB2;
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sb1[r] += __shfl_xor(sb1[r], m, 64); sb2[r] += __shfl_xor(sb2[r], m, 64);
            asb1[r] += __shfl_xor(asb1[r], m, 64); asb2[r] += __shfl_xor(asb2[r], m, 64);
        }
    }
    if (j == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            wp[LH * K1 + 4 * g + r] = sb1[r]; wp[LH * K1 + LH + LH * LH + 4 * g + r] = sb2[r];
            wp[NPD + LH * K1A + 4 * g + r] = asb1[r]; wp[NPD + LH * K1A + LH + LH * LH + 4 * g + r] = asb2[r];
        }
    }
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

bool latent_bwd_shape_ok(const psnode_ode_bwd_args_f32* a) {
    const psnode_mlp_f32& m = a->de;
    return a->x_dim == LH && a->z_dim == LH && m.n_layers == 2 && m.in_dim == LK1 && m.out_dim[0] == LH && m.out_dim[1] == LH;
}

bool latent_bwd_ptrs_ok(const psnode_ode_bwd_args_f32*) { return true; }   // K8f reads lane-granular: any alignment

size_t latent_bwd_workspace_floats(long long B) { return latent_bwd_dpp_workspace_floats(B); }

int latent_bwd_launch(const psnode_ode_bwd_args_f32* a, float* workspace, hipStream_t s) { return latent_bwd_dpp_launch(a, workspace, s); }

// ---- DAE entry points
static bool two16(const psnode_mlp_f32& m, int in_dim) { return m.n_layers == 2 && m.in_dim == in_dim && m.out_dim[0] == LH && m.out_dim[1] == LH; }

bool latent16_dae_bwd_shape_ok(const psnode_dae_bwd_args_f32* a) {
    if (a->x_dim != LH || a->v_dim != LH || a->i_dim != LH || (a->z_dim != LH && a->z_dim != 0)) return false;
    const int nblk = a->z_dim ? 4 : 3;
    return two16(a->de, 3 * nblk * LH) && two16(a->ae, (2 * nblk - 1) * LH);
}

bool latent16_dae_bwd_ptrs_ok(const psnode_dae_bwd_args_f32* a) {
    auto view_ok = [](const psnode_view_f32& v) { return v.ptr && al16(v.ptr) && v.stride_t % 4 == 0 && v.stride_b % 4 == 0; };
    if (!al16(a->all_initial) || !al16(a->xs) || !al16(a->is) || !al16(a->grad_xs) || !al16(a->grad_x_init) || !al16(a->grad_all_initial)) return false;
    if ((a->grad_is && !al16(a->grad_is)) || !view_ok(a->v) || (a->z_dim && !view_ok(a->z))) return false;
    if ((a->grad_z && !al16(a->grad_z)) || (a->grad_v && !al16(a->grad_v))) return false;
    if (a->event_idx) {
        if (a->z_dim && (!al16(a->z_jump) || a->zj_stride_b % 4 || a->zj_stride_e % 4 || (a->grad_z_jump && !al16(a->grad_z_jump)))) return false;
        if (!al16(a->v_jump) || a->vj_stride_b % 4 || a->vj_stride_e % 4 || (a->grad_v_jump && !al16(a->grad_v_jump))) return false;
    }
    return true;
}

static int np16(int k1) { return LH * k1 + LH + LH * LH + LH; }

size_t latent16_dae_bwd_workspace_floats(const psnode_dae_bwd_args_f32* a) {
    const int nblk = a->z_dim ? 4 : 3;
    return (size_t)((a->B + LTB - 1) / LTB) * (np16(3 * nblk * LH) + np16((2 * nblk - 1) * LH)) + 64;
}

int latent16_dae_bwd_launch(const psnode_dae_bwd_args_f32* a, float* workspace, hipStream_t s) {
    LatentDaeBwdDev d;
    memset(&d, 0, sizeof(d));
    d.a.method = a->method; d.a.xd = LH; d.a.zd = a->z_dim; d.a.vd = LH; d.a.id = LH; d.a.T = a->T; d.a.B = a->B;
    d.a.t = ViewDev{a->t.ptr, a->t.stride_t, a->t.stride_b};
    d.a.z = ViewDev{a->z.ptr, a->z.stride_t, a->z.stride_b};
    d.a.v = ViewDev{a->v.ptr, a->v.stride_t, a->v.stride_b};
    d.a.a0 = a->all_initial; d.a.ev = a->event_idx;
    d.a.zj = a->z_jump; d.a.zjb = a->zj_stride_b; d.a.zje = a->zj_stride_e;
    d.a.vj = a->v_jump; d.a.vjb = a->vj_stride_b; d.a.vje = a->vj_stride_e;
    for (int l = 0; l < 2; ++l) { d.a.de.w[l] = a->de.weight[l]; d.a.de.bias[l] = a->de.bias[l]; d.a.ae.w[l] = a->ae.weight[l]; d.a.ae.bias[l] = a->ae.bias[l]; }
    d.xs = a->xs; d.is_ = a->is; d.gxs = a->grad_xs; d.gis = a->grad_is;
    d.gx0 = a->grad_x_init; d.gz = a->grad_z; d.gv = a->grad_v; d.gzj = a->grad_z_jump; d.gvj = a->grad_v_jump; d.ga0 = a->grad_all_initial;
    d.wpart = workspace; d.n_events = a->n_events;
    const int nblk = a->z_dim ? 4 : 3, npd = np16(3 * nblk * LH), npa = np16((2 * nblk - 1) * LH);
    const int nwg = (int)((a->B + LTB - 1) / LTB);
    const dim3 grid((unsigned)nwg), block(64);
#define PSNODE_L16(M_)                                                                                              \
    if (a->z_dim) hipLaunchKernelGGL((latent16_dae_backward_kernel<M_, 3>), grid, block, 0, s, d);                 \
    else hipLaunchKernelGGL((latent16_dae_backward_kernel<M_, 2>), grid, block, 0, s, d);
    switch (a->method) {
        case PSNODE_EULER: PSNODE_L16(PSNODE_EULER) break;
        case PSNODE_MIDPOINT: PSNODE_L16(PSNODE_MIDPOINT) break;
        default: PSNODE_L16(PSNODE_RK4_38) break;
    }
#undef PSNODE_L16
    if (hipGetLastError() != hipSuccess) return PSNODE_ERR_HIP;
    return launch_reduce_partials(workspace, a->grad_params_de, a->grad_params_ae, npd, npa, nwg, s) == hipSuccess ? PSNODE_OK : PSNODE_ERR_HIP;
}

}  // namespace psnode
