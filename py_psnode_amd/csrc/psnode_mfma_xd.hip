// K2x -- the exchange-free form of the fused DAE integrator (round 5): K1x's decomposition (psnode_mfma_x.hip: one WAVE owns 4 trajectories and
// all hidden units on v_mfma_f32_4x4x1_16B_f32, no cross-wave traffic) extended by the AE head of integrate_DAE (my_solvers.py:94-129):
//   DE  3n -> h -> h -> h -> x_dim,   AE  n + x + z + v -> h -> h -> h -> i_dim,   h <= 64, x_dim <= 8, z + v + i <= 8, i_dim <= 4,
// inference without teacher forcing (everything else stays on K2, psnode_mfma_impl.h).
// What is new against K1x:
//   * the DE's H -> H weights stay in VGPRs (128); the AE's live in the wave's AccVGPRs (128 more: one wave per SIMD owns all 512
//     registers) and are READ from there as the MFMA's B operand (psnode_mfma_x.h: hh_layer_acc -- inline asm, because left to the register
//     allocator they come back through a v_accvgpr_read per MFMA).  A first version kept them in LDS (one 32 KB image per workgroup, a
//     ds_read_b128 per 4 MFMAs): 5.16 ms at dae01 / RK4 against K2's 4.66; from AccVGPRs 4.41 (profiles/r05n_dae_tile_vs_wave.txt).
//   * the AE's output layer (64 -> i_dim <= 4) is K1x's split-K layer with ONE accumulator; its lane folds (v_permlane32_swap,
//     v_permlane16_swap, row rotations: 8 VALU instructions) leave algebraic variable d in every lane of ROW drow^-1(d), drow = {0,2,1,3};
//   * the DE's external slots (z | v | i, K2's order: slot q < ne carries ext[q] - a0, ne <= q < 2 ne carries ext[q - ne]) are PLACED for that:
//     block b = slot_of_block(b) -- a row's first two blocks are the two slots of the algebraic variable the row holds, the z | v slots fill
//     the rest -- so the feedback i -> DE input is one v_cndmask per step, and the per-step constant is 16 MFMAs with ABID = block (the
//     permutation lives in the packed weight image, not in an immediate).
// Step k (as K2): [event: i0 = g(x_k; z_jump, v_jump)], the DE stages with (z, v, i) frozen, then i_{k+1} = g(x_{k+1}; z[k+1], v[k+1]).
#include <string.h>

#include "psnode_mfma_x.h"

namespace psnode {
namespace {

// algebraic variable a row's lanes hold after the AE's output folds, and the placement of the DE's external slots (see above)
__host__ __device__ constexpr int xd_drow(int rho) { return rho == 1 ? 2 : (rho == 2 ? 1 : rho); }
__host__ __device__ inline int xd_slot_of_block(int b, int nzv, int id) {
    const int ne = nzv + id, rho = b >> 2, j = b & 3, d = xd_drow(rho);
    const bool has = d < id;
    if (has && j < 2) return j == 0 ? nzv + d : ne + nzv + d;
    int idx = j - (has ? 2 : 0);
    for (int r = 0; r < rho; ++r) idx += 4 - (xd_drow(r) < id ? 2 : 0);
    if (idx >= 2 * nzv) return -1;
    return idx < nzv ? idx : ne + (idx - nzv);
}

struct XDRegs {      // register image pack[reg][lane]; the AE's H -> H matrices follow as [layer][k / 4][lane][k % 4]
    static constexpr int W2 = 0, W3 = 64, W1X = 128, W1E = 136, W1A = 152, B1 = 168, B2 = 169, B3 = 170, W4A = 171, B4C = 179,
                         AW1X = 187, AW1E = 195, AW1A = 203, AB1 = 219, AB2 = 220, AB3 = 221, AW4A = 222, AB4C = 226, COUNT = 230;
    static constexpr int HH_F4 = 2 * 16 * 64;
};
struct PackXD {
    int xd, zd, vd, id, hreal;
    float sc;           // log2e: the scaled ELU domain of the inference kernel; 1: the training forward (plain domain, bit-exact weights)
    const float *w1, *b1, *w2, *b2, *w3, *b3, *w4, *b4;             // DE
    const float *aw1, *ab1, *aw2, *ab2, *aw3, *ab3, *aw4, *ab4;     // AE
    float* out;
};

__global__ void pack_xd_kernel(const PackXD p) {
    const int H = p.hreal, xd = p.xd, nzv = p.zd + p.vd, id = p.id, ne = nzv + id, n = xd + ne, K1 = 3 * n, K1a = n + xd + nzv;
    const int total = XDRegs::COUNT * 64 + XDRegs::HH_F4 * 4;
    const float kLog2e = p.sc;                                 // (shadows the constant: 1 for the training forward's plain-domain image)
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        float v = 0.0f;
        if (idx >= XDRegs::COUNT * 64) {          // AE H -> H image: [layer][kq][lane] f4, component cc = W[unit = lane][k = 4 kq + cc]
            const int e = idx - XDRegs::COUNT * 64, cc = e & 3, lane = (e >> 2) & 63, kq = (e >> 8) & 15, layer = e >> 12, k = 4 * kq + cc;
            const float* W = layer ? p.aw3 : p.aw2;
            if (lane < H && k < H) v = W[lane * H + k];
            p.out[idx] = v;
            continue;
        }
        const int reg = idx >> 6, l = idx & 63, b = l >> 2, c = l & 3, u = l;
        if (reg < XDRegs::W1X) {
            const int k = reg & 63;
            const float* W = reg < XDRegs::W3 ? p.w2 : p.w3;
            if (u < H && k < H) v = W[u * H + k];
        } else if (reg < XDRegs::W1E) {                 // DE x columns, folded: Ws + Wd
            const int d = l1_dim(reg - XDRegs::W1X);
            if (u < H && d < xd) v = (p.w1[u * K1 + 2 * n + d] + p.w1[u * K1 + n + d]) * kLog2e;
        } else if (reg < XDRegs::W1A) {                 // DE external slots, BLOCK order
            const int q = xd_slot_of_block(reg - XDRegs::W1E, nzv, id);
            if (u < H && q >= 0) v = (q < ne ? p.w1[u * K1 + n + xd + q] : p.w1[u * K1 + 2 * n + xd + (q - ne)]) * kLog2e;
        } else if (reg < XDRegs::B1) {                  // DE a0 columns (folded: Wa - Wd on the x dims)
            const int q = reg - XDRegs::W1A;
            if (u < H && q < n) {
                v = p.w1[u * K1 + q];
                if (q < xd) v -= p.w1[u * K1 + n + q];
                v *= kLog2e;
            }
        } else if (reg == XDRegs::B1) { if (u < H) v = p.b1[u] * kLog2e;
        } else if (reg == XDRegs::B2) { if (u < H) v = p.b2[u] * kLog2e;
        } else if (reg == XDRegs::B3) { if (u < H) v = p.b3[u] * kLog2e;
        } else if (reg < XDRegs::B4C) {                 // DE L4 as A operand
            const int j = (reg - XDRegs::W4A) >> 2, cc = (reg - XDRegs::W4A) & 3, d = 4 * j + c, k = 4 * b + cc;
            if (d < xd && k < H) v = p.w4[d * H + k] / kLog2e;
        } else if (reg < XDRegs::AW1X) {
            const int j = (reg - XDRegs::B4C) >> 2, r = (reg - XDRegs::B4C) & 3, d = 4 * j + r;
            if (b == 0 && d < xd) v = p.b4[d];
        } else if (reg < XDRegs::AW1E) {                // AE x columns
            const int d = l1_dim(reg - XDRegs::AW1X);
            if (u < H && d < xd) v = p.aw1[u * K1a + n + d] * kLog2e;
        } else if (reg < XDRegs::AW1A) {                // AE z | v columns: block qa = column qa
            const int qa = reg - XDRegs::AW1E;
            if (u < H && qa < nzv) v = p.aw1[u * K1a + n + xd + qa] * kLog2e;
        } else if (reg < XDRegs::AB1) {                 // AE a0 columns
            const int q = reg - XDRegs::AW1A;
            if (u < H && q < n) v = p.aw1[u * K1a + q] * kLog2e;
        } else if (reg == XDRegs::AB1) { if (u < H) v = p.ab1[u] * kLog2e;
        } else if (reg == XDRegs::AB2) { if (u < H) v = p.ab2[u] * kLog2e;
        } else if (reg == XDRegs::AB3) { if (u < H) v = p.ab3[u] * kLog2e;
        } else if (reg < XDRegs::AB4C) {                // AE L4 as A operand: lane (b, r = c) = AW4[i-dim r][k = 4b + cc] / log2e
            const int cc = reg - XDRegs::AW4A, k = 4 * b + cc;
            if (c < id && k < H) v = p.aw4[c * H + k] / kLog2e;
        } else {                                        // C init of the AE's L4: the bias enters in block 0
            const int r = reg - XDRegs::AB4C;
            if (b == 0 && r < id) v = p.ab4[r];
        }
        p.out[idx] = v;
    }
}

// x += row_ror:8 (x); x += row_ror:4 (x): the sum over the four blocks of a row, in every lane
__device__ __forceinline__ void row_sum1(float& a) {
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf"
        : "+v"(a));
}

// SAVE (round 6): the DAE training forward -- what autograd would keep, in the formats K2's saving instances write and K7f / K7h read:
// DE rows a.sact [T-1,S,3,B,Hp] + stage inputs a.sxst [T-1,S,B,xd] as K1x; the AE head's three layers per grid point a.saeact [3,T,B,Hp]
// and per event a.sevact [nE,3,B,Hp]; the event's i0 in DE-slot layout a.sevi [nE,B,16].  Plain ELU domain (unscaled pack image).
template <int METHOD, bool SAVE>
__global__ __launch_bounds__(64 * kXWaves) void integrate_xd_kernel(const IntegrateDev a, const float* __restrict__ pack) {
    const int l = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = l >> 2, c = l & 3, rho = l >> 4;
    const long long tile = (long long)blockIdx.x * kXWaves + wv;
    if (tile * 4 >= a.B) return;
    const bool valid = tile * 4 + c < a.B;
    const long long tr = valid ? tile * 4 + c : a.B - 1;
    const int xd = a.xd, zd = a.zd, vd = a.vd, idim = a.id, nzv = zd + vd, ne = nzv + idim, n = xd + ne;

    // ---- weights -> registers (once per launch)
    const float* pw = pack + l;
    float w2[64], w3[64], w1x[8], w1e[16], w4a[8], aw1x[8], aw1e[8], aw4a[4];
    f4 b4c[2], ab4c;
#pragma unroll
    for (int k = 0; k < 64; ++k) { w2[k] = pw[(XDRegs::W2 + k) * 64]; w3[k] = pw[(XDRegs::W3 + k) * 64]; }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        w1x[m] = pw[(XDRegs::W1X + m) * 64]; w4a[m] = pw[(XDRegs::W4A + m) * 64];
        aw1x[m] = pw[(XDRegs::AW1X + m) * 64]; aw1e[m] = pw[(XDRegs::AW1E + m) * 64];
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) w1e[q] = pw[(XDRegs::W1E + q) * 64];
    const float b1 = pw[XDRegs::B1 * 64], b2 = pw[XDRegs::B2 * 64], b3 = pw[XDRegs::B3 * 64];
    const float ab1 = pw[XDRegs::AB1 * 64], ab2 = pw[XDRegs::AB2 * 64], ab3 = pw[XDRegs::AB3 * 64];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        b4c[0][r] = pw[(XDRegs::B4C + r) * 64]; b4c[1][r] = pw[(XDRegs::B4C + 4 + r) * 64];
        aw4a[r] = pw[(XDRegs::AW4A + r) * 64]; ab4c[r] = pw[(XDRegs::AB4C + r) * 64];
    }
    float aw2[64], aw3[64];                                    // the AE's H -> H weights: AccVGPRs (hh_layer_acc)
    {
        const float* pa = pack + XDRegs::COUNT * 64 + 4 * l;   // [layer][kq][lane][cc]
#pragma unroll
        for (int k = 0; k < 64; ++k) { aw2[k] = pa[(k >> 2) * 256 + (k & 3)]; aw3[k] = pa[4096 + (k >> 2) * 256 + (k & 3)]; }
    }

    // ---- per-trajectory constants
    const int d01 = 4 * (rho >> 1) + 2 * (rho & 1), d23 = d01 + 1;
    const float* a0p = a.a0 + tr * n;
    f4 c0 = f4{b1, b1, b1, b1}, c0a = f4{ab1, ab1, ab1, ab1};
    {
        const float a0A = b < n ? a0p[b] : 0.0f;
        float wa[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) wa[q] = pw[(XDRegs::W1A + q) * 64];
        c0 = ext_mfmas<0, 16>(wa, a0A, c0);
#pragma unroll
        for (int q = 0; q < 16; ++q) wa[q] = pw[(XDRegs::AW1A + q) * 64];
        c0a = ext_mfmas<0, 16>(wa, a0A, c0a);
    }
    // this lane's DE slot (block b): source kind 0 = z column, 1 = v column, 2 = the row's algebraic variable, 3 = none
    const int q_de = xd_slot_of_block(b, nzv, idim);
    const int e_de = q_de < 0 ? -1 : (q_de < ne ? q_de : q_de - ne);                 // index into ext = z | v | i
    const int kind_de = e_de < 0 ? 3 : (e_de < zd ? 0 : (e_de < nzv ? 1 : 2));
    const int col_de = kind_de == 0 ? e_de : (kind_de == 1 ? e_de - zd : 0);
    const bool sub_de = q_de >= 0 && q_de < ne;
    const float a0e = sub_de ? a0p[xd + q_de] : 0.0f;
    // ... and its AE slot (block b = z | v column b)
    const int kind_ae = b < zd ? 0 : (b < nzv ? 1 : 3);
    const int col_ae = kind_ae == 0 ? b : (kind_ae == 1 ? b - zd : 0);
    float X01 = d01 < xd ? a.x_init[tr * xd + d01] : 0.0f, X23 = d23 < xd ? a.x_init[tr * xd + d23] : 0.0f;
    const bool storer = (b & 3) == 0 && valid;
    const bool st01 = storer && d01 < xd, st23 = storer && d23 < xd, sti = storer && xd_drow(rho) < idim;
    const bool pair_ok = (xd & 1) == 0;
    const unsigned xooff = (unsigned)(tr * xd + d01) * 4u, iooff = (unsigned)(tr * idim + xd_drow(rho)) * 4u;
    const int nT = (int)a.T;

    const long long tst = a.t.st;
    const bool has_z = zd > 0, has_v = vd > 0, has_zj = has_z && a.zj != nullptr, has_vj = has_v && a.vj != nullptr;
    const long long zst = has_z ? a.z.st : 0, vst = has_v ? a.v.st : 0, zje = has_zj ? a.zje : 0, vje = has_vj ? a.vje : 0;
    const unsigned toff = (unsigned)(tr * a.t.sb) * 4u;
    const float* zbase = has_z ? a.z.p : a.t.p;
    const float* vbase = has_v ? a.v.p : a.t.p;
    const float* zjbase = has_zj ? a.zj : a.t.p;
    const float* vjbase = has_vj ? a.vj : a.t.p;
    // One load per lane and source row.  A lane's slot reads a z OR a v column; lanes without a column (algebraic / padding slots) read column
    // 0 of a row that exists (z's, else v's, else the clock) -- the value is never selected.
    const bool de_v = kind_de == 1 || (kind_de >= 2 && !has_z && has_v), ae_v = kind_ae == 1 || (kind_ae >= 2 && !has_z && has_v);
    auto lane_off = [&](const bool isv, const int kind, const int col, const bool jump) -> unsigned {
        const int cc = kind <= 1 ? col : 0;                      // (selects, not branches: the lanes of a wave differ in `isv`)
        const unsigned ov = (unsigned)(tr * (jump ? a.vjb : a.v.sb) + cc) * 4u, oz = (unsigned)(tr * (jump ? a.zjb : a.z.sb) + cc) * 4u;
        return isv ? ov : (has_z ? oz : toff);
    };
    const unsigned off_de = lane_off(de_v, kind_de, col_de, false), off_ae = lane_off(ae_v, kind_ae, col_ae, false);
    const unsigned joff_de = ((de_v && has_vj) || (!de_v && has_zj)) ? lane_off(de_v, kind_de, col_de, true) : toff;
    const unsigned joff_ae = ((ae_v && has_vj) || (!ae_v && has_zj)) ? lane_off(ae_v, kind_ae, col_ae, true) : toff;
    auto as_g = [](const float* q) { return (gptr<const float>)(uintptr_t)q; };
    // the lane's value in row (zr | vr): the row base is uniform per SOURCE, the lane picks its source's (two v_cndmask on the pointer) and loads once
    auto row_val = [&](const float* zr, const float* vr, const bool isv, const unsigned off) -> float {
        const float* rp = isv ? vr : zr;
        return *(gptr<const float>)((gptr<const char>)(uintptr_t)rp + off);
    };

    // SAVE: uniform running row bases + this lane's byte offset (its trajectory's row, its four units); xo_step doubles as the stage-input stride
    const long long xo_step = a.B * xd, io_step = a.B * idim;
    const int hp = SAVE ? padded_hidden(a.de.out_dim[0]) : 0;
    const size_t sa_layer = SAVE ? (size_t)a.B * hp : 0, sae_layer = SAVE ? (size_t)a.T * a.B * hp : 0;
    float* sa_run = SAVE ? a.sact : nullptr;
    float* sx_run = SAVE ? a.sxst : nullptr;
    float* sae_run = SAVE ? a.saeact : nullptr;                     // the AE head's rows of the NEXT grid point to be evaluated
    const unsigned saoff = SAVE ? (unsigned)(tr * hp + 4 * b) * 4u : 0u;
    const bool sa_on = SAVE && valid && 4 * b < hp;
    // DE right-hand side in the state layout (K1x's rhs)
    auto rhs = [&](const float s01, const float s23, const f4 cz, float& k01, float& k23) {
        f4 accA = cz, accB = f4{0.f, 0.f, 0.f, 0.f};
        accA = mfx<0>(s01, w1x[0], accA);  accB = mfx<4>(s01, w1x[1], accB);
        accA = mfx<8>(s01, w1x[2], accA);  accB = mfx<12>(s01, w1x[3], accB);
        accA = mfx<0>(s23, w1x[4], accA);  accB = mfx<4>(s23, w1x[5], accB);
        accA = mfx<8>(s23, w1x[6], accA);  accB = mfx<12>(s23, w1x[7], accB);
        f4 hA = quad_transpose(elu_x<!SAVE>(accA + accB));
        if constexpr (SAVE) {                                        // rows (step, stage): the stage input, then the three layers as they appear
            if (pair_ok) {
                if (st01) stg<f2>((gptr<float>)(uintptr_t)sx_run, xooff, f2{s01, s23});
            } else {
                if (st01) stg<float>((gptr<float>)(uintptr_t)sx_run, xooff, s01);
                if (st23) stg<float>((gptr<float>)(uintptr_t)sx_run, xooff + 4u, s23);
            }
            sx_run += xo_step;
            if (sa_on) stg<f4>((gptr<float>)(uintptr_t)sa_run, saoff, hA);
        }
        hA = hh_layer<!SAVE>(w2, b2, hA);
        if constexpr (SAVE) { if (sa_on) stg<f4>((gptr<float>)(uintptr_t)(sa_run + sa_layer), saoff, hA); }
        hA = hh_layer<!SAVE>(w3, b3, hA);
        if constexpr (SAVE) {
            if (sa_on) stg<f4>((gptr<float>)(uintptr_t)(sa_run + 2 * sa_layer), saoff, hA);
            sa_run += 3 * sa_layer;
        }
        f4 p0 = b4c[0], p1 = b4c[1];
        p0 = mfn(w4a[0], hA[0], p0); p1 = mfn(w4a[4], hA[0], p1);
        p0 = mfn(w4a[1], hA[1], p0); p1 = mfn(w4a[5], hA[1], p1);
        p0 = mfn(w4a[2], hA[2], p0); p1 = mfn(w4a[6], hA[2], p1);
        p0 = mfn(w4a[3], hA[3], p0); p1 = mfn(w4a[7], hA[3], p1);
        const float q0 = fold32(p0[0], p1[0]), q1 = fold32(p0[1], p1[1]), q2 = fold32(p0[2], p1[2]), q3 = fold32(p0[3], p1[3]);
        k01 = fold16(q0, q2); k23 = fold16(q1, q3);
        row_sum2(k01, k23);
    };
    // AE head g(x; z | v): returns, in every lane of row rho, algebraic variable drow(rho) of the lane's trajectory
    // (SAVE: `rows` = uniform base of the head's layer-0 rows [B,Hp] of this grid point / event, `lstride` floats to the next layer)
    auto ae_eval = [&](const float s01, const float s23, const float eAE, float* rows = nullptr, const size_t lstride = 0) -> float {
        f4 accA = c0a, accB = f4{0.f, 0.f, 0.f, 0.f};
        accA = mfx<0>(s01, aw1x[0], accA);  accB = mfx<4>(s01, aw1x[1], accB);
        accA = mfx<8>(s01, aw1x[2], accA);  accB = mfx<12>(s01, aw1x[3], accB);
        accA = mfx<0>(s23, aw1x[4], accA);  accB = mfx<4>(s23, aw1x[5], accB);
        accA = mfx<8>(s23, aw1x[6], accA);  accB = mfx<12>(s23, aw1x[7], accB);
        accA = mfx<0>(eAE, aw1e[0], accA);  accB = mfx<1>(eAE, aw1e[1], accB);
        accA = mfx<2>(eAE, aw1e[2], accA);  accB = mfx<3>(eAE, aw1e[3], accB);
        accA = mfx<4>(eAE, aw1e[4], accA);  accB = mfx<5>(eAE, aw1e[5], accB);
        accA = mfx<6>(eAE, aw1e[6], accA);  accB = mfx<7>(eAE, aw1e[7], accB);
        f4 hA = quad_transpose(elu_x<!SAVE>(accA + accB));
        if constexpr (SAVE) { if (sa_on) stg<f4>((gptr<float>)(uintptr_t)rows, saoff, hA); }
        hA = hh_layer_acc<!SAVE>(aw2, ab2, hA);
        if constexpr (SAVE) { if (sa_on) stg<f4>((gptr<float>)(uintptr_t)(rows + lstride), saoff, hA); }
        hA = hh_layer_acc<!SAVE>(aw3, ab3, hA);
        if constexpr (SAVE) { if (sa_on) stg<f4>((gptr<float>)(uintptr_t)(rows + 2 * lstride), saoff, hA); }
        f4 p0 = ab4c;
        p0 = mfn(aw4a[0], hA[0], p0);
        p0 = mfn(aw4a[1], hA[1], p0);
        p0 = mfn(aw4a[2], hA[2], p0);
        p0 = mfn(aw4a[3], hA[3], p0);
        // rows after the folds: row 0 <- register 0, row 1 <- register 2, row 2 <- register 1, row 3 <- register 3  (xd_drow)
        float w = fold16(fold32(p0[0], p0[1]), fold32(p0[2], p0[3]));
        row_sum1(w);
        return w;
    };

    auto load_evb = [&](const int blk) -> int {
        const int i = blk * 64 + l;
        const int v = (a.ev && i + 1 < nT) ? a.ev[i] : -1;
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
        return v;
    };
    auto store_rows = [&](float* xrow, float* irow, const float iw) {
        if (pair_ok) {
            if (st01) stg<f2>((gptr<float>)(uintptr_t)xrow, xooff, f2{X01, X23});
        } else {
            if (st01) stg<float>((gptr<float>)(uintptr_t)xrow, xooff, X01);
            if (st23) stg<float>((gptr<float>)(uintptr_t)xrow, xooff + 4u, X23);
        }
        if (sti) stg<float>((gptr<float>)(uintptr_t)irow, iooff, iw);
    };

    // ---- grid point 0: i_0 = g(x_init; z[0], v[0])  (my_solvers.py:95)
    float e_ae = row_val(zbase, vbase, ae_v, off_ae);
    float t_cur = ldg<float>(as_g(a.t.p), toff);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    float icw = ae_eval(X01, X23, kind_ae == 3 ? 0.0f : e_ae, sae_run, sae_layer);
    if constexpr (SAVE) sae_run += sa_layer;
    float* xo_run = a.xo;
    float* io_run = a.io;
    if (nT < 2) { store_rows(xo_run, io_run, icw); return; }

    int evb = load_evb(0);
    int ev_cur = __builtin_amdgcn_readlane(evb, 0);
    float t_nxt = ldg<float>(as_g(a.t.p + tst), toff);
    // the DE's external value of step 0: row 0, or the jump row when step 0 takes an event
    float e_de_nxt = ev_cur >= 0 ? row_val(zjbase + (long long)ev_cur * zje, vjbase + (long long)ev_cur * vje, de_v, joff_de)
                                 : row_val(zbase, vbase, de_v, off_de);
    const float* trun = a.t.p + 2 * tst;
    const float* zrun = zbase + zst;        // row k + 1
    const float* vrun = vbase + vst;

    // vmcnt counts in order and a step issues its loads IN FRONT of its stores (x row: one 8-byte store at even x_dim; i row: one store -- both
    // from at least lane 0 of every wave), so "the loads have arrived" is vmcnt(2) and the stores' acknowledgement stays out of the step.
    // Measured, same box (profiles/r05u_dae_tile_vs_wave.txt against r05n_...): RK4 4.41 -> 4.24 ms at dae01, but Euler 2.05 -> 2.09 and
    // Midpoint 2.74 -> 2.82 (and requesting the AE head's row a step earlier, with one wait per step, lost at all three: r05v_...) -- so RK4
    // alone takes it; the others, and odd x_dim (a second x store that not every wave issues), wait for everything.
    // SAVE: behind a step's loads sit its two row stores and 4 S saved-row stores (every one issued by every wave); the AE head at the end of
    // the step adds three more in front of the next step's first wait
    constexpr int kStg = METHOD == PSNODE_EULER ? 1 : (METHOD == PSNODE_MIDPOINT ? 2 : 4);
    auto wait_loads = [&](auto top_tag) {
        constexpr bool TOP = decltype(top_tag)::value;
        if constexpr (SAVE) {
            constexpr int WN = 2 + 4 * kStg + (TOP ? 3 : 0);
            if (pair_ok) __builtin_amdgcn_s_waitcnt(0x0F70 | (WN & 15) | ((WN >> 4) << 14));
            else __builtin_amdgcn_s_waitcnt(0x0F70);
        } else {
            if (METHOD == PSNODE_RK4_38 && pair_ok) __builtin_amdgcn_s_waitcnt(0x0F72);
            else __builtin_amdgcn_s_waitcnt(0x0F70);
        }
    };
    __builtin_amdgcn_s_waitcnt(0x0F70);         // the first step's inputs (no store is in flight yet for vmcnt(2) to skip)
    auto step = [&](const int k, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        wait_loads(std::true_type{});           // this step's inputs were requested a whole step ago
        const float h_ = t_nxt - t_cur;
        t_cur = t_nxt;
        const int ev_now = ev_cur;
        const float i_k = icw;                                   // i_solution[k]: the event below replaces i0 for the DE only
        if (__builtin_amdgcn_readfirstlane(ev_now) >= 0) {       // event: i0 = g(x_k; z_jump, v_jump) with the RUNNING state (my_solvers.py:108-110)
            const float ej = row_val(zjbase + (long long)ev_now * zje, vjbase + (long long)ev_now * vje, ae_v, joff_ae);
            __builtin_amdgcn_s_waitcnt(0x0F70);
            icw = ae_eval(X01, X23, kind_ae == 3 ? 0.0f : ej, SAVE ? a.sevact + (size_t)ev_now * 3 * sa_layer : nullptr, sa_layer);
            if constexpr (SAVE) {     // the event's i0 in the DE's slot layout [nE,B,16]: slots nzv + d (the `s - a0` block) and ne + nzv + d
                const int d = xd_drow(rho);
                if (storer && d < idim) {
                    float* er = a.sevi + ((size_t)ev_now * a.B + tr) * 16;
                    er[nzv + d] = icw;
                    er[ne + nzv + d] = icw;
                }
                __builtin_amdgcn_s_waitcnt(0x0F70);      // (an event step: the counted waits below assume the regular store sequence)
            }
        }
        // the lane's DE slot: a z | v column (loaded) or the row's algebraic variable; the `s - a0` block subtracts a0
        float eA = kind_de == 2 ? icw : e_de_nxt;
        eA = sub_de ? eA - a0e : eA;
        eA = kind_de == 3 ? 0.0f : eA;
        // prefetch: t[k + 2], row k + 1 for the AE head at the end of this step and for the DE of the next one (its jump row on an event step)
        if constexpr (!LAST) {
            t_nxt = ldg<float>(as_g(trun), toff);
            trun += tst;
            if (((k + 1) & 63) == 0) evb = load_evb((k + 1) >> 6);
            ev_cur = __builtin_amdgcn_readlane(evb, (k + 1) & 63);
        }
        e_ae = row_val(zrun, vrun, ae_v, off_ae);
        if constexpr (!LAST) {
            if (__builtin_amdgcn_readfirstlane(ev_cur) >= 0)
                e_de_nxt = row_val(zjbase + (long long)ev_cur * zje, vjbase + (long long)ev_cur * vje, de_v, joff_de);
            else
                e_de_nxt = row_val(zrun, vrun, de_v, off_de);
        }
        zrun += zst;
        vrun += vst;
        store_rows(xo_run, io_run, i_k);      // deferred store of grid point k (the previous step's result)
        xo_run += xo_step;
        io_run += io_step;
        const f4 cz = ext_mfmas<0, 16>(w1e, eA, c0);
        const float s01 = X01, s23 = X23;
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
        // i_{k+1} = g(x_{k+1}; z[k+1], v[k+1]): un-jumped inputs of the right grid point (my_solvers.py:121)
        wait_loads(std::false_type{});          // (requested at the top of this step, in front of its stores)
        icw = ae_eval(X01, X23, kind_ae == 3 ? 0.0f : e_ae, sae_run, sae_layer);
        if constexpr (SAVE) sae_run += sa_layer;
    };
#pragma unroll 1
    for (int k = 0; k + 2 < nT; ++k) step(k, std::false_type{});
    step(nT - 2, std::true_type{});
    store_rows(xo_run, io_run, icw);
}

}  // namespace

// shapes K2x takes (K2's classes at hidden <= 64, i_dim <= 4), inference without teacher forcing
bool mfma_x_dae_supported(const IntegrateDev& a) {
    const int nzv = a.zd + a.vd, ne = nzv + a.id, n = a.xd + ne;
    if (a.flags || a.xd < 1 || a.xd > 8 || a.zd < 0 || a.vd < 0 || a.id < 1 || a.id > 4 || nzv < 1 || ne > 8 || a.T >= (1ll << 31)) return false;
    const MlpDev &d = a.de, &g = a.ae;
    if (d.n_layers != 4 || g.n_layers != 4 || d.in_dim != 3 * n || d.out_dim[3] != a.xd || g.in_dim != n + a.xd + nzv || g.out_dim[3] != a.id) return false;
    const int h = d.out_dim[0];
    if (!(h >= 1 && h <= 64 && d.out_dim[1] == h && d.out_dim[2] == h && g.out_dim[0] == h && g.out_dim[1] == h && g.out_dim[2] == h)) return false;
    // 32-bit per-lane byte offsets (psnode_mfma_x.h: span32_ok): every row the time loop addresses that way, once the pointers are known
    if (a.t.p && !span32_ok(a.B, a.t.sb, 1)) return false;
    if (a.zd > 0 && a.z.p && !span32_ok(a.B, a.z.sb, a.zd)) return false;
    if (a.vd > 0 && a.v.p && !span32_ok(a.B, a.v.sb, a.vd)) return false;
    if (a.zd > 0 && a.zj && !span32_ok(a.B, a.zjb, a.zd)) return false;
    if (a.vd > 0 && a.vj && !span32_ok(a.B, a.vjb, a.vd)) return false;
    if (a.sact) {      // the training forward: every save buffer present and aligned for the 16-byte / 8-byte stores, 32-bit row offsets
        if (!a.sxst || !a.saeact || (a.ev && (!a.sevact || !a.sevi))) return false;
        if (((uintptr_t)a.sact & 15) || ((uintptr_t)a.sxst & 7) || ((uintptr_t)a.saeact & 15) || ((uintptr_t)a.sevact & 15)) return false;
        if (!span32_ok(a.B, 64, 64)) return false;
    }
    return span32_ok(a.B, a.xd, a.xd) && span32_ok(a.B, a.id, a.id);                           // output rows [B,xd] / [B,id]
}
bool mfma_x_dae_preferred(const IntegrateDev& a) {
    if (a.kern == PSNODE_KERNEL_MFMA_TILE || a.kern == PSNODE_KERNEL_GENERIC || !mfma_x_dae_supported(a)) return false;
    // The SAVING instance is forced-only: measured (profiles/r06_k2x_save_time.txt, B = 4096 x 1000 steps) it is 5.54 ms at RK4 / 2.42 ms at
    // Euler against K2's 5.10 / 2.19 -- the 16.3 GB of rows cost a lone wave 1.3 ms of write back-pressure (nothing else to issue), K2's waves
    // only 0.4 ms (their exchange bubbles absorb it).  Inference (no rows): K2x 4.23 vs 4.68 ms.
    if (a.sact) return a.kern == PSNODE_KERNEL_MFMA_WAVE;
    return a.kern == PSNODE_KERNEL_MFMA_WAVE || a.B <= 4608;
}
size_t mfma_xd_pack_floats() { return (size_t)XDRegs::COUNT * 64 + (size_t)XDRegs::HH_F4 * 4; }

hipError_t launch_mfma_xd(const IntegrateDev& a, float* pack, hipStream_t stream) {
    PackXD p;
    p.xd = a.xd; p.zd = a.zd; p.vd = a.vd; p.id = a.id; p.hreal = a.de.out_dim[0];
    p.w1 = a.de.w[0]; p.b1 = a.de.bias[0]; p.w2 = a.de.w[1]; p.b2 = a.de.bias[1];
    p.w3 = a.de.w[2]; p.b3 = a.de.bias[2]; p.w4 = a.de.w[3]; p.b4 = a.de.bias[3];
    p.aw1 = a.ae.w[0]; p.ab1 = a.ae.bias[0]; p.aw2 = a.ae.w[1]; p.ab2 = a.ae.bias[1];
    p.aw3 = a.ae.w[2]; p.ab3 = a.ae.bias[2]; p.aw4 = a.ae.w[3]; p.ab4 = a.ae.bias[3];
    p.out = pack;
    p.sc = a.sact ? 1.0f : kLog2e;
    hipLaunchKernelGGL(pack_xd_kernel, dim3(32), dim3(256), 0, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const long long tiles = (a.B + 3) / 4;
    const dim3 grid((unsigned)((tiles + kXWaves - 1) / kXWaves)), block(64 * kXWaves);
#define PSNODE_XD(M_)                                                                                          \
    if (a.sact) hipLaunchKernelGGL((integrate_xd_kernel<M_, true>), grid, block, 0, stream, a, pack);          \
    else hipLaunchKernelGGL((integrate_xd_kernel<M_, false>), grid, block, 0, stream, a, pack);
    switch (a.method) {
        case PSNODE_EULER: PSNODE_XD(PSNODE_EULER) break;
        case PSNODE_MIDPOINT: PSNODE_XD(PSNODE_MIDPOINT) break;
        default: PSNODE_XD(PSNODE_RK4_38) break;
    }
#undef PSNODE_XD
    return hipGetLastError();
}

}  // namespace psnode
