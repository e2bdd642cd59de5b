// K4x -- the exchange-free backward through the ODE integrator at hidden <= 64 (round 6): the form of K1x
// (psnode_mfma_x.hip) applied to loss.backward() through the unrolled loop of my_solvers.py:66-78 (neural_00_ODE_01_no_encode.py:358-360).
// ONE WAVE owns 4 trajectories and ALL 64 hidden units, sweeps the grid from the last step to the first and reads the rows the training
// forward saved (K1x SAVE: h1 | h2 | h3 [T-1,S,3,B,64] and the stage inputs [T-1,S,B,xd]).  Nothing crosses waves: no LDS, no barrier, no
// second role -- K4f's two-role tile spends its time in 3 exchanges per stage and in feeding a second wave per SIMD through LDS tiles.
//
// Everything is v_mfma_f32_4x4x1_16B_f32 (16 blocks of D_b(4x4) += A_b(4x1) B_b(1x4); lane l = (block b = l >> 2, c = l & 3)) and every
// operand is in one of K1x's three layouts:
//   D layout   f4, register r = trajectory, lane = unit                    (what an MFMA layer returns; what the saved rows are loaded as:
//                                                                           one 256-byte row per register)
//   A layout   f4, register cc, lane (b, t) = value[unit 4b + cc][traj t]  (quad_transpose of D: the A operand of the next layer)
//   S layout   2 registers, lane in row rho, t = l & 3: dims 4(rho>>1) + 2(rho&1) and the next one of trajectory t (state / adjoint)
// The adjoint chain of a stage is K1x's forward with transposed images:
//   t3 = W4^T gk        8 MFMAs, A = gk (S layout, ABID = 4 rho(d)), B = W4[d][unit]                   -> D;  delta3 = t3 * ELU'(h3)
//   t2 = W3^T delta3    64 MFMAs, A = quad_transpose(delta3), B = W3[u = k-slot][unit = lane]          -> D;  delta2 = t2 * ELU'(h2)
//   t1 = W2^T delta2    64 MFMAs                                                                       -> D;  delta1 = t1 * ELU'(h1)
//   ds = F^T delta1     8 MFMAs split-K over the blocks + the lane folds of K1x's L4 (F = Ws + Wd, x columns)  -> S
// and the weight gradients need NO re-layout at all: with A = a saved row in D layout and ABID = kk, block kk's four lanes hand
// h[t][4kk .. 4kk+3] to all 16 blocks, B = delta in D layout:
//   dW^T tile kk:  acc[lane (b, c)][reg i] += h[t][4kk + i] * delta[t][4b + c]        one MFMA per trajectory, 16 tiles x 4 = 64 per H -> H matrix
//   dW4 (2 tiles): A = gk re-laid by 2 v_permlane16_swap + one in-quad transpose, B = h3;   dW1 (<= 4 tiles): A = ONE register holding the
//   stage input | z row of trajectory t in row t of the wave (loaded straight from the saved stage inputs; ABID = 4t + tile), B = delta1.
// The 152 accumulators live in AccVGPRs for the whole launch (inline asm, `+a`), the two transposed H -> H images in 128 VGPRs.
// Per stage: 144 chain + 128 + 8 + 4 ceil(n/4) gradient MFMAs.  Deterministic: per-wave partials, summed in a fixed order
// (launch_reduce_partials), then mapped into nn.Linear order.
#include <string.h>

#include "psnode_mfma_x.h"

namespace psnode {
namespace {

#ifndef PSNODE_K4X_ABL
#define PSNODE_K4X_ABL 0      // timing-only ablations (results WRONG): 1 = no weight-gradient MFMAs, 2 = every stage reads the rows of (step 0, stage 0)
#endif

// register image: pack[reg][lane]
struct BXRegs {
    static constexpr int W3T = 0, W2T = 64, W4T = 128, FT = 136, FZ = 144, A0 = 152, COUNT = 168;
};
// per-wave partial record: part[reg][lane]
struct BXPart {
    static constexpr int W2 = 0, W3 = 64, W4 = 128, W1 = 136, CA0 = 152, B1 = 168, B2 = 169, B3 = 170, B4 = 171, COUNT = 173;
};

struct PackBX {
    int xd, zd, n, hreal;
    const float *w1, *w2, *w3, *w4;
    float* out;
};
__global__ void pack_bx_kernel(const PackBX p) {
    const int H = p.hreal, n = p.n, xd = p.xd, zd = p.zd, K1 = 3 * n;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < BXRegs::COUNT * 64; idx += gridDim.x * blockDim.x) {
        const int reg = idx >> 6, l = idx & 63, b = l >> 2, c = l & 3;
        float v = 0.0f;
        if (reg < BXRegs::W4T) {                        // transposed H -> H images: B operand, lane = input unit of the layer, register = its output unit
            const int u = reg & 63;
            const float* W = reg < BXRegs::W2T ? p.w3 : p.w2;
            if (u < H && l < H) v = W[u * H + l];
        } else if (reg < BXRegs::FT) {                  // W4^T: B operand of the 8 -> 64 layer, register m <-> x-dim l1_dim(m)
            const int d = l1_dim(reg - BXRegs::W4T);
            if (d < xd && l < H) v = p.w4[d * H + l];
        } else if (reg < BXRegs::FZ) {                  // F^T (x columns, Ws + Wd): A operand of the split-K layer, lane (b, r = c), register 4j + cc
            const int j = (reg - BXRegs::FT) >> 2, cc = (reg - BXRegs::FT) & 3, d = 4 * j + c, k = 4 * b + cc;
            if (d < xd && k < H) v = p.w1[k * K1 + 2 * n + d] + p.w1[k * K1 + n + d];
        } else if (reg < BXRegs::A0) {                  // the same for the z columns (dL/dz)
            const int j = (reg - BXRegs::FZ) >> 2, cc = (reg - BXRegs::FZ) & 3, e = 4 * j + c, k = 4 * b + cc;
            if (e < zd && k < H) v = p.w1[k * K1 + 2 * n + xd + e] + p.w1[k * K1 + n + xd + e];
        } else {                                        // (Wa - Wd)^T, all n <= 16 columns (dL/dall_initial)
            const int j = (reg - BXRegs::A0) >> 2, cc = (reg - BXRegs::A0) & 3, q = 4 * j + c, k = 4 * b + cc;
            if (q < n && k < H) v = p.w1[k * K1 + q] - p.w1[k * K1 + n + q];
        }
        p.out[idx] = v;
    }
}

struct BwdXDev {
    int xd, zd, hreal, n_events;
    long long T, B;
    ViewDev t, z;
    const float* a0;
    const int* ev;
    const float* zj;
    long long zjb, zje;
    const float* gout;
    float *gx0, *gz, *gzj, *ga0;
    float* wpart;                 // [tiles][BXPart::COUNT][64]
    const float *sact, *sxst;
};

// Two gradient tiles, four trajectories: acc_x += A_t (block K_x + TS t) (x) B_t.  The accumulators are AccVGPRs for the whole launch (the
// allocator would bring VGPR-form accumulators back and forth); a tile's dependent MFMAs are two issue slots apart (the other tile's + s_nop 0).
// Wait states inside the block are its own business (the hazard recognizer does not look): s_nop 1 = VALU write -> MFMA operand read.
// LEAD: the block may directly follow the VALU instruction that wrote one of its operands (s_nop 1); the later blocks of a run over the
// same operands follow an MFMA block and need none (76 blocks per stage: the two cycles each were 4 % of a step).
#define PSNODE_WG_BODY                                                          \
        "v_mfma_f32_4x4x1_16b_f32 %0, %2, %6, %0 cbsz:4 abid:%10\n\t"            \
        "v_mfma_f32_4x4x1_16b_f32 %1, %2, %6, %1 cbsz:4 abid:%14\n\t"            \
        "s_nop 0\n\t"                                                            \
        "v_mfma_f32_4x4x1_16b_f32 %0, %3, %7, %0 cbsz:4 abid:%11\n\t"            \
        "v_mfma_f32_4x4x1_16b_f32 %1, %3, %7, %1 cbsz:4 abid:%15\n\t"            \
        "s_nop 0\n\t"                                                            \
        "v_mfma_f32_4x4x1_16b_f32 %0, %4, %8, %0 cbsz:4 abid:%12\n\t"            \
        "v_mfma_f32_4x4x1_16b_f32 %1, %4, %8, %1 cbsz:4 abid:%16\n\t"            \
        "s_nop 0\n\t"                                                            \
        "v_mfma_f32_4x4x1_16b_f32 %0, %5, %9, %0 cbsz:4 abid:%13\n\t"            \
        "v_mfma_f32_4x4x1_16b_f32 %1, %5, %9, %1 cbsz:4 abid:%17\n\t"            \
        "s_nop 0"
#define PSNODE_WG_OPS                                                                                                     \
        : "+a"(acc0), "+a"(acc1)                                                                                          \
        : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(bv[0]), "v"(bv[1]), "v"(bv[2]), "v"(bv[3]),                            \
          "n"(K0), "n"(K0 + TS), "n"(K0 + 2 * TS), "n"(K0 + 3 * TS), "n"(K1), "n"(K1 + TS), "n"(K1 + 2 * TS), "n"(K1 + 3 * TS)
template <int K0, int K1, int TS, bool LEAD = true>
__device__ __forceinline__ void wg_pair(f4& acc0, f4& acc1, const float a0, const float a1, const float a2, const float a3, const f4 bv) {
    if constexpr (PSNODE_K4X_ABL == 1) return;
    if constexpr (LEAD) asm volatile("s_nop 1\n\t" PSNODE_WG_BODY PSNODE_WG_OPS);
    else asm volatile(PSNODE_WG_BODY PSNODE_WG_OPS);
}
#undef PSNODE_WG_BODY
#undef PSNODE_WG_OPS
// all 16 tiles of one H -> H matrix: dW^T[k-slot][unit] += h (x) delta over the wave's four trajectories
template <int KK>
__device__ __forceinline__ void wgrad_tiles(f4 (&acc)[16], const f4 h, const f4 d) {
    if constexpr (KK < 16) {
        wg_pair<KK, KK + 1, 0, (KK == 0)>(acc[KK], acc[KK + 1], h[0], h[1], h[2], h[3], d);
        wgrad_tiles<KK + 2>(acc, h, d);
    }
}

// transposed H -> H layer: A = delta in A layout, B = the transposed image; no bias, no activation
__device__ __forceinline__ f4 hh_layer_t(const float (&wk)[64], const f4 dA) {
    f4 accA = f4{0.f, 0.f, 0.f, 0.f}, accB = f4{0.f, 0.f, 0.f, 0.f};
    hh_block<0>(wk, dA, accA, accB); hh_block<1>(wk, dA, accA, accB); hh_block<2>(wk, dA, accA, accB); hh_block<3>(wk, dA, accA, accB);
    hh_block<4>(wk, dA, accA, accB); hh_block<5>(wk, dA, accA, accB); hh_block<6>(wk, dA, accA, accB); hh_block<7>(wk, dA, accA, accB);
    hh_block<8>(wk, dA, accA, accB); hh_block<9>(wk, dA, accA, accB); hh_block<10>(wk, dA, accA, accB); hh_block<11>(wk, dA, accA, accB);
    hh_block<12>(wk, dA, accA, accB); hh_block<13>(wk, dA, accA, accB); hh_block<14>(wk, dA, accA, accB); hh_block<15>(wk, dA, accA, accB);
    return accA + accB;
}
// 64 -> 8 split-K layer (K1x's L4): A = image (lane (b, r), register 4j + cc), B = the input in A layout; the sum over the 16 blocks
// leaves the S layout
__device__ __forceinline__ void l4_like(const float (&wa)[8], const f4 hA, float& k01, float& k23) {
    f4 p0 = f4{0.f, 0.f, 0.f, 0.f}, p1 = f4{0.f, 0.f, 0.f, 0.f};
    p0 = mfn(wa[0], hA[0], p0); p1 = mfn(wa[4], hA[0], p1);
    p0 = mfn(wa[1], hA[1], p0); p1 = mfn(wa[5], hA[1], p1);
    p0 = mfn(wa[2], hA[2], p0); p1 = mfn(wa[6], hA[2], p1);
    p0 = mfn(wa[3], hA[3], p0); p1 = mfn(wa[7], hA[3], p1);
    const float q0 = fold32(p0[0], p1[0]), q1 = fold32(p0[1], p1[1]), q2 = fold32(p0[2], p1[2]), q3 = fold32(p0[3], p1[3]);
    k01 = fold16(q0, q2); k23 = fold16(q1, q3);
    row_sum2(k01, k23);
}
// S layout -> the A operands of the dW4 tiles: out[t], lane (block 0, i) = g[dim i][t], lane (block 8, i) = g[dim 4 + i][t]
__device__ __forceinline__ f4 s_to_rows(const float g01, const float g23) {
    const auto r01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(g01), __float_as_uint(g01), false, false);
    const auto r23 = __builtin_amdgcn_permlane16_swap(__float_as_uint(g23), __float_as_uint(g23), false, false);
    // rows 0 / 2 of r[0] = rows 0 / 2 of the source, of r[1] = rows 1 / 3: in row 0 the four registers are dims 0..3, in row 2 dims 4..7
    return quad_transpose(f4{__uint_as_float(r01[0]), __uint_as_float(r23[0]), __uint_as_float(r01[1]), __uint_as_float(r23[1])});
}
__device__ __forceinline__ float sum4(const f4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }
// delta = t * ELU'(pre) from h = ELU(pre) > -1:  ELU' = min(h, 0) + 1, and min(h, 0) = med3(h, -2, 0) -- ONE VOP3 instruction per value (fminf
// costs a canonicalising v_max in front of the v_min for a value that comes from memory), then t * m + t as a packed fma: 6 instructions
// per tile instead of 12.  With one wave per SIMD every VALU instruction is wall time.
__device__ __forceinline__ f4 times_elu_grad(const f4 t, const f4 h) {
    const f2 m0 = f2{__builtin_amdgcn_fmed3f(h[0], -2.0f, 0.0f), __builtin_amdgcn_fmed3f(h[1], -2.0f, 0.0f)};
    const f2 m1 = f2{__builtin_amdgcn_fmed3f(h[2], -2.0f, 0.0f), __builtin_amdgcn_fmed3f(h[3], -2.0f, 0.0f)};
    const f2 t0 = f2{t[0], t[1]}, t1 = f2{t[2], t[3]};
    const f2 r0 = __builtin_elementwise_fma(t0, m0, t0), r1 = __builtin_elementwise_fma(t1, m1, t1);
    return f4{r0[0], r0[1], r1[0], r1[1]};
}

// GZ: dL/dz (or dL/dz_jump) is wanted -- one more split-K layer per step on the sum of the step's delta1
template <int METHOD, bool GZ>
__global__ __launch_bounds__(64 * kXWaves) void ode_backward_x_kernel(const BwdXDev a, const float* __restrict__ pack) {
    constexpr int S = rk_stages(METHOD);
    const int l = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = l >> 2, c = l & 3, rho = l >> 4, j16 = l & 15;
    const long long tile = (long long)blockIdx.x * kXWaves + wv;
    if (tile * 4 >= a.B) return;                                    // (nothing is shared between the waves)
    const int xd = a.xd, zd = a.zd, n = xd + zd;
    const long long nB = a.B;
    const int nT = (int)a.T;                                        // 32-bit loop counters (no 64-bit scalar compare; the launcher bounds T)
    const bool validS = tile * 4 + c < nB;
    const long long trS = validS ? tile * 4 + c : nB - 1;           // trajectory of this lane in the S layout
    const long long trG = tile * 4 + rho < nB ? tile * 4 + rho : nB - 1;   // ... in the row-per-trajectory registers (stage input, a0)
    const int d01 = 4 * (rho >> 1) + 2 * (rho & 1), d23 = d01 + 1;
    const bool storer = (b & 3) == 0 && validS;

    // ---- images -> registers
    const float* pw = pack + l;
    float w3t[64], w2t[64], w4t[8], ft[8], fz[GZ ? 8 : 1];
#pragma unroll
    for (int k = 0; k < 64; ++k) { w3t[k] = pw[(BXRegs::W3T + k) * 64]; w2t[k] = pw[(BXRegs::W2T + k) * 64]; }
#pragma unroll
    for (int m = 0; m < 8; ++m) { w4t[m] = pw[(BXRegs::W4T + m) * 64]; ft[m] = pw[(BXRegs::FT + m) * 64]; }
    if constexpr (GZ) {
#pragma unroll
        for (int m = 0; m < 8; ++m) fz[m] = pw[(BXRegs::FZ + m) * 64];
    }

    // ---- accumulators (whole launch)
    const f4 zero4 = f4{0.f, 0.f, 0.f, 0.f};
    f4 aW2[16], aW3[16], aW4[2], aW1[4];
#pragma unroll
    for (int k = 0; k < 16; ++k) { aW2[k] = zero4; aW3[k] = zero4; }
    aW4[0] = zero4; aW4[1] = zero4;
#pragma unroll
    for (int k = 0; k < 4; ++k) aW1[k] = zero4;
    f4 S1D = zero4, S2D = zero4, S3D = zero4;                       // sums of delta1..3 per trajectory (D layout): biases, a0 columns, dL/da0
    float db4a = 0.0f, db4b = 0.0f;
    float lam01 = 0.0f, lam23 = 0.0f;                               // the adjoint carried from step to step (S layout)

    // ---- addressing: <uniform row base> + <32-bit per-lane byte offset> (psnode_common.h: sbase / ldg)
    unsigned offH[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long long tr = tile * 4 + r < nB ? tile * 4 + r : nB - 1;
        offH[r] = 4u * ((unsigned)(tr * 64) + (unsigned)l);
    }
    const size_t act_layer = (size_t)nB * 64, xst_row = (size_t)nB * xd;
    const bool c_s = j16 < xd, c_z = !c_s && j16 < n;
    const unsigned offC = 4u * ((unsigned)(trG * xd) + (unsigned)(c_s ? j16 : 0));
    const bool has_z = zd > 0;
    const unsigned offT = 4u * (unsigned)(trS * a.t.sb);
    const int zc = c_z ? j16 - xd : 0;
    const unsigned offZ = has_z ? 4u * (unsigned)(trG * a.z.sb + zc) : offT;
    const unsigned offZJ = (has_z && a.zj) ? 4u * (unsigned)(trG * a.zjb + zc) : offT;
    const int d01c = d01 < xd ? d01 : 0, d23c = d23 < xd ? d23 : 0;
    const unsigned offG01 = 4u * (unsigned)(trS * xd + d01c), offG23 = 4u * (unsigned)(trS * xd + d23c);
    const bool on01 = validS && d01 < xd, on23 = validS && d23 < xd;

    if (nT >= 2) {
    // the rows of (step, stage) idx: uniform running bases (the sweep walks idx downwards: one 64-bit subtraction per stage and array), a
    // lane offset made opaque ONCE per register and stage (ldg()'s per-load copy is a v_mov per load: 13 per stage)
    auto ld = [](const gptr<const float> base, const unsigned off) -> float { return *(gptr<const float>)((gptr<const char>)base + off); };
    const float* srun = a.sact;            // rows the NEXT request reads
    const float* xrun = a.sxst;
    auto load_rows = [&](f4 (&h)[3], float& cs) {
        const gptr<const float> b0 = sbase(srun), b1 = sbase(srun + act_layer), b2 = sbase(srun + 2 * act_layer);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            unsigned o = offH[r];
            asm volatile("" : "+v"(o));
            h[0][r] = ld(b0, o); h[1][r] = ld(b1, o); h[2][r] = ld(b2, o);
        }
        cs = ldg<float>(sbase(xrun), offC);
    };
    auto rows_back = [&](const int idx) {      // move the running bases one (step, stage) down, clamped at the first one
        if (idx > 0) { srun -= 3 * act_layer; xrun -= xst_row; }
    };
    auto load_z = [&](const int k, const int ev) -> float {
        const bool jump = ev >= 0 && a.zj != nullptr;
        const float* row = !has_z ? a.t.p : (jump ? a.zj + (long long)ev * a.zje : a.z.p + (long long)k * a.z.st);
        return ldg<float>(sbase(row), jump ? offZJ : offZ);
    };
    auto load_t = [&](const int k) -> float { return ldg<float>(sbase(a.t.p + (long long)k * a.t.st), offT); };
    auto load_g = [&](const int k, float& g01, float& g23) {
        const gptr<const float> row = sbase(a.gout + (size_t)k * xst_row);
        g01 = ldg<float>(row, offG01);
        g23 = ldg<float>(row, offG23);
    };

    // rows of (step, stage), linear index idx = k S + s, walked downwards: set (idx & 1) -- a compile-time index (s & 1 for an even stage
    // count, the parity of the unrolled step body at Euler) -- holds them, requested one stage ahead
    f4 hq[2][3];
    float cq[2] = {0.0f, 0.0f};
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int L = 0; L < 3; ++L) hq[q][L] = zero4;
    constexpr int QLAST = S == 1 ? 0 : ((S - 1) & 1);
    {
        const int last = PSNODE_K4X_ABL == 2 ? 0 : (nT - 2) * S + (S - 1);
        srun += (size_t)last * 3 * act_layer;
        xrun += (size_t)last * xst_row;
        load_rows(hq[QLAST], cq[QLAST]);
        rows_back(last);
    }
    // per-step inputs, requested a step ahead; the event index two steps ahead as a RAW per-lane load (a scalar load would be waited for at
    // its first use -- the address of the z row -- every step)
    unsigned lane_zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));
    const bool has_ev = a.ev != nullptr;
    const int* evbase = has_ev ? a.ev : reinterpret_cast<const int*>(a.t.p);        // (a GLOBAL load: a flat one would make every wait vmcnt(0))
    auto load_ev = [&](const int k) -> int { return ldg<int>(sbase(evbase + k), lane_zero); };
    int evn = has_ev ? __builtin_amdgcn_readfirstlane(a.ev[nT - 2]) : -1;
    int evr = load_ev(nT >= 3 ? nT - 3 : 0);
    float t_hi = load_t(nT - 1), t_lo = load_t(nT - 2);
    float gin01, gin23;
    load_g(nT - 1, gin01, gin23);
    float zin = load_z(nT - 2, evn);

    auto step = [&](const int k, auto kp_tag) {
        constexpr int KP = decltype(kp_tag)::value;
        asm volatile("" : "+v"(t_lo), "+v"(gin01), "+v"(gin23), "+v"(zin));      // (the same for the step's inputs, requested a step ago)
        const float h_ = t_hi - t_lo;
        const int ev = evn;
        const float zrow = c_z ? zin : 0.0f;                         // the step's external row in the stage-input register's z columns
        const float g1a = lam01 + (on01 ? gin01 : 0.0f), g1b = lam23 + (on23 ? gin23 : 0.0f);
        float gka[S], gkb[S];
#pragma unroll
        for (int s = 0; s < S; ++s) { gka[s] = (h_ * rk_b(METHOD, s)) * g1a; gkb[s] = (h_ * rk_b(METHOD, s)) * g1b; }
        float gxa = g1a, gxb = g1b;
        {   // the next step's inputs (unconditional, clamped at the first step)
            const int kp = k > 0 ? k - 1 : 0;
            evn = has_ev ? __builtin_amdgcn_readfirstlane(evr) : -1;      // requested a step ago
            evr = load_ev(k >= 2 ? k - 2 : 0);
            t_hi = t_lo;
            t_lo = load_t(kp);
            load_g(kp + 1, gin01, gin23);
            zin = load_z(kp, evn);
        }
        f4 sstepA = zero4;
#pragma unroll
        for (int s = S - 1; s >= 0; --s) {
            const int Q = S == 1 ? KP : (s & 1);                     // (compile-time once the stage loop is unrolled)
            const int idx = k * S + s;
            f4 h1, h2, h3;
            float cs;
            if (Q == 0) { h1 = hq[0][0]; h2 = hq[0][1]; h3 = hq[0][2]; cs = cq[0]; load_rows(hq[1], cq[1]); }
            else { h1 = hq[1][0]; h2 = hq[1][1]; h3 = hq[1][2]; cs = cq[1]; load_rows(hq[0], cq[0]); }
            if constexpr (PSNODE_K4X_ABL != 2) rows_back(idx - 1);
            // The rows are CONSUMED here and not before: left to the scheduler, the first VALU instructions that read them (ELU' has no other
            // input) are hoisted to right behind their loads -- a stage early -- and the wave waits out the memory round trip there, every
            // stage (found in the ISA: s_waitcnt vmcnt(11..0) + v_med3 interleaved with the load block; 4.83 ms of chain for K1x's 3.1).
            // Volatile asm statements keep their order, so this one pins the first use behind the previous stage's last transpose.
            asm volatile("" : "+v"(h1), "+v"(h2), "+v"(h3), "+v"(cs));
            const float cin = c_s ? cs : zrow;
            const float g01 = gka[s], g23 = gkb[s];
            db4a += g01; db4b += g23;
            // delta3 = (W4^T gk) * ELU'(h3)
            f4 accA = zero4, accB = zero4;
            accA = mfx<0>(g01, w4t[0], accA);  accB = mfx<4>(g01, w4t[1], accB);
            accA = mfx<8>(g01, w4t[2], accA);  accB = mfx<12>(g01, w4t[3], accB);
            accA = mfx<0>(g23, w4t[4], accA);  accB = mfx<4>(g23, w4t[5], accB);
            accA = mfx<8>(g23, w4t[6], accA);  accB = mfx<12>(g23, w4t[7], accB);
            const f4 d3 = times_elu_grad(accA + accB, h3);
            S3D += d3;
            {   // dW4 += gk (x) h3
                const f4 gA = s_to_rows(g01, g23);
                wg_pair<0, 8, 0>(aW4[0], aW4[1], gA[0], gA[1], gA[2], gA[3], h3);
            }
            const f4 d3A = quad_transpose(d3);
            const f4 d2 = times_elu_grad(hh_layer_t(w3t, d3A), h2);
            wgrad_tiles<0>(aW3, h2, d3);                              // dW3 += delta3 (x) h2
            S2D += d2;
            const f4 d2A = quad_transpose(d2);
            const f4 d1 = times_elu_grad(hh_layer_t(w2t, d2A), h1);
            wgrad_tiles<0>(aW2, h1, d2);                              // dW2 += delta2 (x) h1
            S1D += d1;
            wg_pair<0, 1, 4>(aW1[0], aW1[1], cin, cin, cin, cin, d1);  // dW1[:, stage input | z] += delta1 (x) (s | z)
            if (n > 8) wg_pair<2, 3, 4, false>(aW1[2], aW1[3], cin, cin, cin, cin, d1);
            const f4 d1A = quad_transpose(d1);
            if constexpr (GZ) { if constexpr (S > 1) sstepA += d1A; else sstepA = d1A; }
            float dsa, dsb;
            l4_like(ft, d1A, dsa, dsb);                               // dL/d(stage input)
            gxa += dsa; gxb += dsb;
#pragma unroll
            for (int jj = 0; jj < s; ++jj) { gka[jj] += (h_ * rk_a(METHOD, s, jj)) * dsa; gkb[jj] += (h_ * rk_a(METHOD, s, jj)) * dsb; }
        }
        lam01 = gxa; lam23 = gxb;
        if constexpr (GZ) {     // gradient of this step's external input (frozen over the stages): F_z^T sum(delta1)
            float gza, gzb;
            l4_like(fz, sstepA, gza, gzb);
            int lq = l;                                                // (the row's z columns recomputed from an opaque lane id: see the epilogue)
            asm volatile("" : "+v"(lq));
            const int e01 = 4 * (lq >> 5) + 2 * ((lq >> 4) & 1), e23 = e01 + 1;
            if (storer) {
                if (e01 < zd) {
                    if (ev >= 0) { if (a.gzj) a.gzj[(trS * a.n_events + ev) * zd + e01] = gza; }
                    if (a.gz) a.gz[((long long)k * nB + trS) * zd + e01] = ev >= 0 ? 0.0f : gza;
                }
                if (e23 < zd) {
                    if (ev >= 0) { if (a.gzj) a.gzj[(trS * a.n_events + ev) * zd + e23] = gzb; }
                    if (a.gz) a.gz[((long long)k * nB + trS) * zd + e23] = ev >= 0 ? 0.0f : gzb;
                }
            }
        }
    };
    if constexpr (S == 1) {
        int k = nT - 2;
        for (; k >= 1; k -= 2) { step(k, std::integral_constant<int, 0>{}); step(k - 1, std::integral_constant<int, 1>{}); }
        if (k == 0) step(0, std::integral_constant<int, 0>{});
    } else {
        for (int k = nT - 2; k >= 0; --k) step(k, std::integral_constant<int, 0>{});
    }
    }   // nT >= 2

    // ---- epilogue
    // (lane-derived values are RECOMPUTED from an opaque lane id: kept live from the prologue across the time loop they cost registers
    //  the loop does not have -- the Euler instance with dL/dz spilled six of them to scratch)
    int l_e = threadIdx.x & 63;
    asm volatile("" : "+v"(l_e));
    const int rho_e = l_e >> 4, j16_e = l_e & 15, c_e = l_e & 3;
    const bool validS_e = tile * 4 + c_e < nB;
    const long long trS_e = validS_e ? tile * 4 + c_e : nB - 1;
    const long long trG_e = tile * 4 + rho_e < nB ? tile * 4 + rho_e : nB - 1;
    const int d01_e = 4 * (rho_e >> 1) + 2 * (rho_e & 1), d23_e = d01_e + 1;
    const bool storer_e = ((l_e >> 2) & 3) == 0 && validS_e;
    const float* pw_e = pack + l_e;
    if (storer_e) {
        const float* g0 = a.gout + trS_e * xd;
        if (d01_e < xd) a.gx0[trS_e * xd + d01_e] = lam01 + g0[d01_e];
        if (d23_e < xd) a.gx0[trS_e * xd + d23_e] = lam23 + g0[d23_e];
        if (GZ && a.gz && nT >= 1) {        // z[T-1] is never read by the ODE loop
            if (d01_e < zd) a.gz[((long long)(nT - 1) * nB + trS_e) * zd + d01_e] = 0.0f;
            if (d23_e < zd) a.gz[((long long)(nT - 1) * nB + trS_e) * zd + d23_e] = 0.0f;
        }
    }
    {   // dL/dall_initial[c] = sum_u (Wa - Wd)[u][c] S1[u], n <= 16 columns: two split-K layers
        const f4 s1A = quad_transpose(S1D);
        float wa[8];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int m = 0; m < 8; ++m) wa[m] = pw_e[(BXRegs::A0 + 8 * half + m) * 64];
            float ga, gb;
            l4_like(wa, s1A, ga, gb);
            if (storer_e) {
                if (8 * half + d01_e < n) a.ga0[trS_e * n + 8 * half + d01_e] = ga;
                if (8 * half + d23_e < n) a.ga0[trS_e * n + 8 * half + d23_e] = gb;
            }
        }
    }
    // a0 columns of dW1: ca0 = sum(delta1) (x) a0 over the wave's trajectories
    f4 ca[4] = {zero4, zero4, zero4, zero4};
    {
        const bool live = tile * 4 + rho_e < nB && j16_e < n;
        const float a0in = live ? a.a0[trG_e * n + j16_e] : 0.0f;
        wg_pair<0, 1, 4>(ca[0], ca[1], a0in, a0in, a0in, a0in, S1D);
        wg_pair<2, 3, 4>(ca[2], ca[3], a0in, a0in, a0in, a0in, S1D);
    }
    float* wp = a.wpart + (size_t)tile * (BXPart::COUNT * 64) + l_e;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            wp[(BXPart::W2 + 4 * kk + i) * 64] = aW2[kk][i];
            wp[(BXPart::W3 + 4 * kk + i) * 64] = aW3[kk][i];
        }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) wp[(BXPart::W4 + 4 * j + i) * 64] = aW4[j][i];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            wp[(BXPart::W1 + 4 * j + i) * 64] = aW1[j][i];
            wp[(BXPart::CA0 + 4 * j + i) * 64] = ca[j][i];
        }
    wp[BXPart::B1 * 64] = sum4(S1D);
    wp[BXPart::B2 * 64] = sum4(S2D);
    wp[BXPart::B3 * 64] = sum4(S3D);
    // db4: S layout, summed over the quad's four trajectories (every lane of the quad then holds the sum)
    db4a += __shfl_xor(db4a, 1, 64); db4a += __shfl_xor(db4a, 2, 64);
    db4b += __shfl_xor(db4b, 1, 64); db4b += __shfl_xor(db4b, 2, 64);
    wp[BXPart::B4 * 64] = db4a;
    wp[(BXPart::B4 + 1) * 64] = db4b;
}

// reduced record [BXPart::COUNT][64] -> nn.Linear order [W1 (H x 3n), b1, W2, b2, W3, b3, W4 (xd x H), b4]
__global__ void scatter_bx_kernel(const float* __restrict__ red, float* __restrict__ out, const int H, const int xd, const int n) {
    const int K1 = 3 * n, oB1 = H * K1, oW2 = oB1 + H, oB2 = oW2 + H * H, oW3 = oB2 + H, oB3 = oW3 + H * H, oW4 = oB3 + H, oB4 = oW4 + xd * H,
              NP = oB4 + xd;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= NP) return;
    auto R = [&](const int reg, const int lane) -> float { return red[reg * 64 + lane]; };
    float v;
    if (p < oB1) {
        const int u = p / K1, col = p - u * K1;
        if (col < n) v = R(BXPart::CA0 + col, u);
        else if (col < 2 * n) v = R(BXPart::W1 + col - n, u) - R(BXPart::CA0 + col - n, u);
        else v = R(BXPart::W1 + col - 2 * n, u);
    } else if (p < oW2) v = R(BXPart::B1, p - oB1);
    else if (p < oB2) { const int q = p - oW2, u = q / H, k = q - u * H; v = R(BXPart::W2 + k, u); }
    else if (p < oW3) v = R(BXPart::B2, p - oB2);
    else if (p < oB3) { const int q = p - oW3, u = q / H, k = q - u * H; v = R(BXPart::W3 + k, u); }
    else if (p < oW4) v = R(BXPart::B3, p - oB3);
    else if (p < oB4) { const int q = p - oW4, d = q / H, k = q - d * H; v = R(BXPart::W4 + d, k); }
    else { const int d = p - oB4; v = R(BXPart::B4 + (d & 1), 16 * (d >> 1)); }
    out[p] = v;
}

}  // namespace

// shapes K4x takes: the saved-activation backward of `3n -> h -> h -> h -> x_dim`, 33 <= h <= 64 (rows saved at width 64), x_dim <= 8,
// z_dim <= 8 (n <= 16), no teacher forcing
bool bwd_x_shape_ok(const psnode_ode_bwd_args_f32* a) {
    const psnode_mlp_f32& m = a->de;
    if (a->x_dim < 1 || a->x_dim > 8 || a->z_dim < 0 || a->z_dim > 8) return false;
    if (m.n_layers != 4 || m.in_dim != 3 * (a->x_dim + a->z_dim) || m.out_dim[3] != a->x_dim) return false;
    const int h = m.out_dim[0];
    return h > 32 && h <= 64 && m.out_dim[1] == h && m.out_dim[2] == h;
}
// ... and the calls it takes them for: saved rows present, 32-bit per-lane offsets, and -- one wave owns a SIMD's whole register file --
// up to one wave per SIMD unless the call forces it (PSNODE_KERNEL_MFMA_WAVE)
bool bwd_x_preferred(const psnode_ode_bwd_args_f32* a) {
    if (a->kernel == PSNODE_KERNEL_GENERIC || a->kernel == PSNODE_KERNEL_MFMA_WIDE || a->kernel == PSNODE_KERNEL_MFMA_TILE) return false;
    if (!bwd_x_shape_ok(a) || !a->saved_act || !a->saved_xstage || (a->flags & PSNODE_FLAG_INPUT_TRUE_X)) return false;
    if (a->T >= (1ll << 28)) return false;                          // 32-bit (step, stage) counters
    if (!span32_ok(a->B, 64, 64) || !span32_ok(a->B, a->t.stride_b, 1)) return false;
    if (a->z_dim > 0 && !span32_ok(a->B, a->z.stride_b, a->z_dim)) return false;
    if (a->z_dim > 0 && a->event_idx && a->z_jump && !span32_ok(a->B, a->zj_stride_b, a->z_dim)) return false;
    return a->kernel == PSNODE_KERNEL_MFMA_WAVE || a->B <= 4608;
}
size_t bwd_x_workspace_floats(const psnode_ode_bwd_args_f32* a) {
    const size_t tiles = (size_t)((a->B + 3) / 4);
    return (size_t)BXRegs::COUNT * 64 + (tiles + 1) * (size_t)BXPart::COUNT * 64 + 256;
}

int bwd_x_launch(const psnode_ode_bwd_args_f32* p, float* workspace, hipStream_t s) {
    const int xd = p->x_dim, zd = p->z_dim, n = xd + zd, HR = p->de.out_dim[0];
    float* pack = workspace;
    float* wpart = pack + (size_t)BXRegs::COUNT * 64;
    const long long tiles = (p->B + 3) / 4;
    float* red = wpart + (size_t)tiles * BXPart::COUNT * 64;
    PackBX pk{xd, zd, n, HR, p->de.weight[0], p->de.weight[1], p->de.weight[2], p->de.weight[3], pack};
    hipLaunchKernelGGL(pack_bx_kernel, dim3(16), dim3(256), 0, s, pk);
    if (hipGetLastError() != hipSuccess) return PSNODE_ERR_HIP;
    BwdXDev a;
    memset(&a, 0, sizeof(a));
    a.xd = xd; a.zd = zd; a.hreal = HR; a.n_events = p->n_events; a.T = p->T; a.B = p->B;
    a.t = ViewDev{p->t.ptr, p->t.stride_t, p->t.stride_b};
    a.z = ViewDev{p->z.ptr, p->z.stride_t, p->z.stride_b};
    a.a0 = p->all_initial; a.ev = p->event_idx; a.zj = p->z_jump; a.zjb = p->zj_stride_b; a.zje = p->zj_stride_e;
    a.gout = p->grad_xs; a.gx0 = p->grad_x0; a.gz = p->grad_z; a.gzj = p->grad_z_jump; a.ga0 = p->grad_all_initial;
    a.wpart = wpart; a.sact = p->saved_act; a.sxst = p->saved_xstage;
    const bool gz = zd > 0 && (a.gz || a.gzj);
    const dim3 grid((unsigned)((tiles + kXWaves - 1) / kXWaves)), block(64 * kXWaves);
#define PSNODE_BX(M_)                                                                                   \
    if (gz) hipLaunchKernelGGL((ode_backward_x_kernel<M_, true>), grid, block, 0, s, a, pack);          \
    else hipLaunchKernelGGL((ode_backward_x_kernel<M_, false>), grid, block, 0, s, a, pack);
    switch (p->method) {
        case PSNODE_EULER: PSNODE_BX(PSNODE_EULER) break;
        case PSNODE_MIDPOINT: PSNODE_BX(PSNODE_MIDPOINT) break;
        default: PSNODE_BX(PSNODE_RK4_38) break;
    }
#undef PSNODE_BX
    if (hipGetLastError() != hipSuccess) return PSNODE_ERR_HIP;
    if (launch_reduce_partials(wpart, red, nullptr, BXPart::COUNT * 64, 0, (int)tiles, s) != hipSuccess) return PSNODE_ERR_HIP;
    const int NP = HR * 3 * n + HR + 2 * (HR * HR + HR) + xd * HR + xd;
    hipLaunchKernelGGL(scatter_bx_kernel, dim3((NP + 255) / 256), dim3(256), 0, s, red, p->grad_params, HR, xd, n);
    return hipGetLastError() == hipSuccess ? PSNODE_OK : PSNODE_ERR_HIP;
}

}  // namespace psnode
