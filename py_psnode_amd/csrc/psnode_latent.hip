// K3a -- latent-space integrator of the direct_encode variants (hidden_dim = 16, the scripts' default):
//   ODE_02:  DE = Linear(6H,H) ELU Linear(H,H),  state Xh[H], external Zh[H]      neural_00_ODE_02_direct_encode.py:49-57
//   DAE_02:  DE = Linear(12H|9H,H) ELU Linear(H,H), AE = Linear(7H|5H,H) ELU Linear(H,H), blocks x | [z] | v | i of
//            width H each                                                        neural_01_DAE_02_direct_encode.py:70-100
//
// With H = 16 the hidden layer is ONE 16x16 MFMA tile, so a single wave owns a tile of 16 trajectories end to end:
// no LDS, no barrier, no cross-wave exchange.  D row 4g+r of every layer is unit/dim 4g+r, and the K order of every
// 16-wide block is chosen as column 4g+m for MFMA m, lane group g -- so a lane's four D registers ARE its four B
// operands for the next layer, and every per-step global access of a lane (state, z|v blocks, outputs) is one
// aligned float4.  Weights stay in VGPRs; the a0 columns of L1 are folded into a per-trajectory constant, the
// external-input columns into a per-step constant (zero-order hold).
#include "psnode_common.h"

namespace psnode {
namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int LH = 16;     // latent / hidden width
constexpr int LTB = 16;    // trajectories per wave

__device__ __forceinline__ f4 lmfma(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

__device__ __forceinline__ f4 lelu4(f4 v) { return elu_quad(v); }

// Packed image per lane (identical for every wave): pack[reg][lane]
//   DE: S[blk][m] (NBLK*4) | D[blk][m] (NBLK*4) | B1 (4) | W2 (4) | B2 (4) | A0[m] (n/4)          NBLK = 1 + NBE
//   AE: X/ZV[blk][m] (NBE*4) | B1 (4) | W2 (4) | B2 (4) | A0[m] (n/4)                              blocks x | z|v
struct PackLatent {
    int ae, nblk, n, k1;    // nblk = blocks after a0 in this MLP's input; k1 = in_features
    const float *w1, *b1, *w2, *b2;
    float* out;
};

__global__ void pack_latent_kernel(const PackLatent p) {
    const int nS = p.nblk * 4, nD = p.ae ? 0 : p.nblk * 4;
    const int B1 = nS + nD, W2 = B1 + 4, B2 = W2 + 4, A0 = B2 + 4, R = A0 + p.n / 4;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < R * 64; idx += gridDim.x * blockDim.x) {
        const int lane = idx & 63, reg = idx >> 6, i = lane & 15, g = lane >> 4;
        float v;
        if (reg < nS) {              // DE: `s` group (offset 2n) ; AE: blocks right after a0 (offset n)
            const int blk = reg >> 2, m = reg & 3;
            v = p.w1[i * p.k1 + (p.ae ? p.n : 2 * p.n) + LH * blk + 4 * g + m];
        } else if (reg < B1) {       // DE: `s - a0` group (offset n)
            const int blk = (reg - nS) >> 2, m = (reg - nS) & 3;
            v = p.w1[i * p.k1 + p.n + LH * blk + 4 * g + m];
        } else if (reg < W2) {
            v = p.b1[4 * g + (reg - B1)];
        } else if (reg < B2) {
            v = p.w2[i * LH + 4 * g + (reg - W2)];
        } else if (reg < A0) {
            v = p.b2[4 * g + (reg - B2)];
        } else {
            const int m = reg - A0;
            v = p.w1[i * p.k1 + LH * (m >> 2) + 4 * g + (m & 3)];
        }
        p.out[idx] = v;
    }
}

template <int N> struct F4s { f4 v[N > 0 ? N : 1]; };

// NBE = number of external blocks of the DE (ODE: 1 = z; DAE: 3 = z,v,i or 2 = v,i).  For the DAE the last block is
// the algebraic variable (register resident), the first NBE-1 are z|v streamed from memory.
template <int METHOD, int NBE, bool DAE>
__global__ __launch_bounds__(64) void latent_kernel(const IntegrateDev a, const float* __restrict__ pack_de,
                                                    const float* __restrict__ pack_ae) {
    constexpr int NBLK = 1 + NBE;            // x + ext blocks
    constexpr int NZV = DAE ? NBE - 1 : NBE; // streamed blocks
    constexpr int n = LH * NBLK;
    const int l = threadIdx.x, g = l >> 4, j = l & 15;
    const long long b0 = (long long)blockIdx.x * LTB;
    const bool valid = b0 + j < a.B;
    const long long b = valid ? b0 + j : a.B - 1;

    // ---- weights -> registers
    float ws[NBLK * 4], wd[NBLK * 4], w2[4];
    f4 b1r, b2r;
    {
        const float* pw = pack_de + l;
#pragma unroll
        for (int k = 0; k < NBLK * 4; ++k) { ws[k] = pw[k * 64]; wd[k] = pw[(NBLK * 4 + k) * 64]; }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            b1r[r] = pw[(NBLK * 8 + r) * 64]; w2[r] = pw[(NBLK * 8 + 4 + r) * 64]; b2r[r] = pw[(NBLK * 8 + 8 + r) * 64];
        }
    }
    float aw[DAE ? NBE * 4 : 1], aw2[4];
    f4 ab1r = f4{0.f, 0.f, 0.f, 0.f}, ab2r = ab1r;
    if constexpr (DAE) {
        const float* pw = pack_ae + l;
#pragma unroll
        for (int k = 0; k < NBE * 4; ++k) aw[k] = pw[k * 64];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ab1r[r] = pw[(NBE * 4 + r) * 64]; aw2[r] = pw[(NBE * 4 + 4 + r) * 64]; ab2r[r] = pw[(NBE * 4 + 8 + r) * 64];
        }
    }

    // ---- per-trajectory constants: a0 blocks (this lane's columns 4g..4g+3 of every block), folded a0 columns of L1
    f4 a0b[NBLK];
    f4 c0 = b1r, c0a = ab1r;
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) {
        a0b[blk] = *reinterpret_cast<const f4*>(a.a0 + b * n + LH * blk + 4 * g);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            c0 = lmfma(pack_de[(NBLK * 8 + 12 + 4 * blk + m) * 64 + l], a0b[blk][m], c0);
            if constexpr (DAE) c0a = lmfma(pack_ae[(NBE * 4 + 12 + 4 * blk + m) * 64 + l], a0b[blk][m], c0a);
        }
    }
    f4 x = *reinterpret_cast<const f4*>((DAE ? a.x_init + b * LH : a.x.p + b * a.x.sb) + 4 * g);

    const long long tst = a.t.st, nT = a.T;
    const float* tp = a.t.p + b * a.t.sb;
    // streamed block s (0 = z or, when the model has no z, v ; 1 = v): base pointers, strides, jump tables
    const bool has_z = a.zd > 0;
    const float* vbase = DAE ? a.v.p + b * a.v.sb : nullptr;
    const float* vjbase = DAE ? a.vj + b * a.vjb : nullptr;
    const float* sp[2] = {has_z ? a.z.p + b * a.z.sb : vbase, vbase};
    const long long sst[2] = {has_z ? a.z.st : a.v.st, a.v.st};
    const float* jp[2] = {has_z ? a.zj + b * a.zjb : vjbase, vjbase};
    const long long jse[2] = {has_z ? a.zje : a.vje, a.vje};
    auto load_blocks = [&](long long k, int ev, F4s<NZV>& dst) {
#pragma unroll
        for (int s = 0; s < NZV; ++s) {
            const long long off = ev >= 0 ? ev * jse[s] : k * sst[s];
            dst.v[s] = *reinterpret_cast<const f4*>((ev >= 0 ? jp[s] : sp[s]) + off + 4 * g);
        }
    };

    // L2 + ELU from an L1 pre-activation: the lane's four hidden units are its four B operands
    auto layer2 = [&](const f4 pre, const float (&wq)[4], const f4 bias) -> f4 {
        const f4 h = lelu4(pre);
        f4 accA = lmfma(wq[0], h[0], bias), accB = lmfma(wq[1], h[1], f4{0.f, 0.f, 0.f, 0.f});
        accA = lmfma(wq[2], h[2], accA);
        accB = lmfma(wq[3], h[3], accB);
        return accA + accB;
    };
    auto rhs = [&](const f4 xs, const f4 cz) -> f4 {
        f4 accA = cz, accB = f4{0.f, 0.f, 0.f, 0.f};
        const f4 xdiff = xs - a0b[0];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            accA = lmfma(ws[m], xs[m], accA);
            accB = lmfma(wd[m], xdiff[m], accB);
        }
        return layer2(accA + accB, w2, b2r);
    };
    auto ae_eval = [&](const f4 xa, const F4s<NZV>& zv) -> f4 {
        f4 accA = c0a, accB = f4{0.f, 0.f, 0.f, 0.f};
        if constexpr (DAE) {
#pragma unroll
            for (int m = 0; m < 4; ++m) accA = lmfma(aw[m], xa[m], accA);
#pragma unroll
            for (int s = 0; s < NZV; ++s) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    if (s & 1) accA = lmfma(aw[4 * (1 + s) + m], zv.v[s][m], accA);
                    else accB = lmfma(aw[4 * (1 + s) + m], zv.v[s][m], accB);
                }
            }
            return layer2(accA + accB, aw2, ab2r);
        }
        return accA;
    };
    auto store4 = [&](float* base, long long k, const f4 v) {
        if (valid) *reinterpret_cast<f4*>(base + (k * a.B + b) * LH + 4 * g) = v;
    };

    store4(a.xo, 0, x);
    f4 icur = f4{0.f, 0.f, 0.f, 0.f};
    F4s<NZV> zva_nxt = {};
    if constexpr (DAE) {
        F4s<NZV> zv0;
        load_blocks(0, -1, zv0);
        icur = ae_eval(x, zv0);
        store4(a.io, 0, icur);
        if (nT > 1) load_blocks(1, -1, zva_nxt);
    }
    if (nT < 2) return;

    float t_cur = tp[0], t_nxt = tp[tst];
    int lane_zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));
    // (unconditional prefetch with clamped indices and a raw event-table entry, as K3c: a load inside `if (k + 2 < nT)` is a phi with a
    //  default, the copy into the loop-carried register and the wait for the whole prefetch sat right behind it)
    const bool has_ev = a.ev != nullptr;
    const int* evp = (has_ev ? a.ev : reinterpret_cast<const int*>(a.t.p)) + lane_zero;
    int ev_cur = has_ev ? a.ev[0] : -1;
    int ev_raw = evp[nT > 2 ? 1 : 0];
    F4s<NZV> ext_nxt = {};
    load_blocks(0, ev_cur, ext_nxt);

    for (long long k = 0; k + 1 < nT; ++k) {
        const float h_ = t_nxt - t_cur;
        t_cur = t_nxt;
        const F4s<NZV> extv = ext_nxt;
        const F4s<NZV> zva = zva_nxt;
        const int ev_now = ev_cur;
        {
            const bool more = k + 2 < nT;
            const long long kn = more ? k + 1 : k;                    // the last step re-reads its own inputs (unused)
            t_nxt = tp[(kn + 1) * tst];
            ev_cur = (has_ev && more) ? ev_raw : -1;
            load_blocks(kn, ev_cur, ext_nxt);
            if constexpr (DAE) load_blocks(kn + 1, -1, zva_nxt);
            ev_raw = evp[k + 3 < nT ? k + 2 : 0];
        }
        if constexpr (DAE) {
            if (__builtin_amdgcn_readfirstlane(ev_now) >= 0) {   // i0 = g(x0; jumped z, v)  (my_solvers.py:108-110)
                F4s<NZV> zvj;
                load_blocks(k, ev_now, zvj);
                icur = ae_eval(x, zvj);
            }
        }
        // per-step constant: c0 + W1[:, ext columns of `s-a0`].(ext - a0) + W1[:, ext columns of `s`].ext
        f4 czA = c0, czB = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < NBE; ++e) {
            const f4 val = (DAE && e == NBE - 1) ? icur : extv.v[e < NZV ? e : 0];
            const f4 dif = val - a0b[1 + e];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                czA = lmfma(ws[4 * (1 + e) + m], val[m], czA);
                czB = lmfma(wd[4 * (1 + e) + m], dif[m], czB);
            }
        }
        const f4 cz = czA + czB;

        const f4 k1 = rhs(x, cz);
        if constexpr (METHOD == PSNODE_EULER) {
            x = x + h_ * k1;
        } else if constexpr (METHOD == PSNODE_MIDPOINT) {
            const f4 k2 = rhs(x + k1 * (0.5f * h_), cz);
            x = x + h_ * k2;
        } else {
            const f4 k2 = rhs(x + h_ * k1 * kOneThird, cz);
            const f4 k3 = rhs(x + h_ * (k2 - k1 * kOneThird), cz);
            const f4 k4 = rhs(x + h_ * (k1 - k2 + k3), cz);
            x = x + (k1 + 3.0f * (k2 + k3) + k4) * h_ * 0.125f;
        }
        store4(a.xo, k + 1, x);
        if constexpr (DAE) {
            icur = ae_eval(x, zva);
            store4(a.io, k + 1, icur);
        }
    }
}

bool two_layer(const MlpDev& m, int in_dim) {
    return m.n_layers == 2 && m.in_dim == in_dim && m.out_dim[0] == LH && m.out_dim[1] == LH;
}
bool aligned4(const ViewDev& v) { return v.p && (reinterpret_cast<uintptr_t>(v.p) & 15) == 0 && v.st % 4 == 0 && v.sb % 4 == 0; }

}  // namespace

// raw-pointer alignment is only known at launch time; shape support is decided on dims alone and the launch
// re-checks the pointers (falling back to the generic kernel through PSNODE_KERNEL_AUTO is the caller's job).
bool latent_shape_ok(const IntegrateDev& a, bool dae) {
    if (a.flags) return false;   // teacher forcing never occurs on the latent path of the scripts: generic kernel
    if (!dae) return a.xd == LH && a.zd == LH && two_layer(a.de, 6 * LH);
    if (a.xd != LH || a.vd != LH || a.id != LH || (a.zd != LH && a.zd != 0)) return false;
    const int nblk = a.zd ? 4 : 3;
    return two_layer(a.de, 3 * nblk * LH) && two_layer(a.ae, (2 * nblk - 1) * LH);
}

bool latent_ptrs_ok(const IntegrateDev& a, bool dae) {
    if (!dae) return true;   // K3f reads lane-granular
    if ((reinterpret_cast<uintptr_t>(a.a0) & 15) || (reinterpret_cast<uintptr_t>(a.xo) & 15)) return false;
    if ((reinterpret_cast<uintptr_t>(a.x_init) & 15) || (reinterpret_cast<uintptr_t>(a.io) & 15)) return false;
    if (a.zd && !aligned4(a.z)) return false;
    if (!aligned4(a.v)) return false;
    if (a.ev) {
        if (a.zd && ((reinterpret_cast<uintptr_t>(a.zj) & 15) || a.zjb % 4 || a.zje % 4)) return false;
        if ((reinterpret_cast<uintptr_t>(a.vj) & 15) || a.vjb % 4 || a.vje % 4) return false;
    }
    return true;
}

size_t latent_pack_floats() { return 2 * (size_t)(8 * 4 + 12 + 16) * 64; }

template <int METHOD>
static hipError_t launch_latent_method(const IntegrateDev& a, bool dae, const float* pde, const float* pae, hipStream_t s) {
    const dim3 grid((unsigned)((a.B + LTB - 1) / LTB)), block(64);
    if (!dae) hipLaunchKernelGGL((latent_kernel<METHOD, 1, false>), grid, block, 0, s, a, pde, pae);
    else if (a.zd) hipLaunchKernelGGL((latent_kernel<METHOD, 3, true>), grid, block, 0, s, a, pde, pae);
    else hipLaunchKernelGGL((latent_kernel<METHOD, 2, true>), grid, block, 0, s, a, pde, pae);
    return hipGetLastError();
}

hipError_t launch_latent(const IntegrateDev& a, bool dae, float* pack, hipStream_t stream) {
    if (!dae) return launch_latent_dpp(a, stream);   // ODE: K3f (4 trajectories per wave, every SIMD busy); K3a below serves the DAE
    const int nblk = dae ? (a.zd ? 4 : 3) : 2;
    PackLatent p;
    p.ae = 0; p.nblk = nblk; p.n = nblk * LH; p.k1 = 3 * p.n;
    p.w1 = a.de.w[0]; p.b1 = a.de.bias[0]; p.w2 = a.de.w[1]; p.b2 = a.de.bias[1];
    p.out = pack;
    hipLaunchKernelGGL(pack_latent_kernel, dim3(4), dim3(256), 0, stream, p);
    float* pack_ae = pack + latent_pack_floats() / 2;
    if (dae) {
        PackLatent q;
        q.ae = 1; q.nblk = nblk - 1; q.n = p.n; q.k1 = p.n + (nblk - 1) * LH;
        q.w1 = a.ae.w[0]; q.b1 = a.ae.bias[0]; q.w2 = a.ae.w[1]; q.b2 = a.ae.bias[1];
        q.out = pack_ae;
        hipLaunchKernelGGL(pack_latent_kernel, dim3(4), dim3(256), 0, stream, q);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    switch (a.method) {
        case PSNODE_EULER: return launch_latent_method<PSNODE_EULER>(a, dae, pack, pack_ae, stream);
        case PSNODE_MIDPOINT: return launch_latent_method<PSNODE_MIDPOINT>(a, dae, pack, pack_ae, stream);
        default: return launch_latent_method<PSNODE_RK4_38>(a, dae, pack, pack_ae, stream);
    }
}

}  // namespace psnode
