// K3f -- the hidden-16 latent ODE of the direct_encode scripts on the VALU with DPP row broadcasts, and the WHOLE
// ODE_Model.forward of neural_00_ODE_02_direct_encode.py:74-89 (x_encoder, z_encoder, latent integrate_ODE, x_decoder on the
// solution, x_decoder(x_encoder(x)) reconstruction) fused into one launch.
//
// Why not MFMA here (K3a was): with H = 16 every layer is ONE 16x16 MFMA tile, so a wave is a serial chain
// MFMA -> ELU -> MFMA with nothing to overlap, 16 trajectories per wave = 256 waves at B = 4096 = ONE wave per CU, three of four
// SIMDs idle (round 1: 1.96 ms, latency-bound).  And on gfx950 VALU work next to fp32 MFMA is additive anyway
// (profiles/r02a_ubench_mfma.txt).  Here a lane is one (trajectory, unit) pair:
//   * a wave = 4 trajectories x 16 units (one DPP row of 16 lanes per trajectory) -> 1024 waves at B = 4096, every SIMD busy;
//   * y[u] = b[u] + sum_j W[u][j] * h[j]  is 16 x  v_fmac_f32_dpp acc, h, w_j  row_newbcast:j  -- the broadcast of lane j's
//     value to its row rides on the FMA's own operand fetch (same 5 issue cycles as a plain v_fma_f32,
//     profiles/r02a_ubench_valu.txt): no LDS, no barrier, no shuffle instructions, no cross-wave traffic;
//   * lane u keeps ROW u of every weight matrix in VGPRs for the whole launch (read once from the nn.Linear tensors);
//   * L1 of the DE is folded:  W.cat(a0, s - a0, s) = (Ws + Wd).s + (Wa - Wd).a0  -- the a0 term is a per-trajectory constant,
//     the z block a per-step constant (zero-order hold), so a stage costs 16 + 16 FMAs and one 12-instruction scalar ELU.
// Fused model (ENC = true): per time step the wave also encodes this step's raw z (zd -> H -> H) and decodes the new state
// (H -> H -> xd); a second set of four waves in the workgroup runs x_decoder(x_encoder(x[t])) for the same 16 trajectories over
// all T rows (independent of the integration; it shares the SIMDs with the latency-bound chain waves and finishes in their
// shadow).  HBM traffic is the algorithmic minimum of SURVEY 8(d): read t 4 + z 8 + x 32, write x_pred 32 + x_re 32 = 108 B per
// state-step; Xh, Zh and Xh_solution never exist in memory.
#include <string.h>

#include "psnode_common.h"

namespace psnode {
namespace {

constexpr int LH = 16;
constexpr int DTB = 16;   // trajectories per workgroup (4 waves x 4 DPP rows)
#ifndef PSNODE_DPP_PF
#define PSNODE_DPP_PF 4
#endif
constexpr int PF = PSNODE_DPP_PF;   // look-ahead of the step inputs, in time steps

// One accumulator chain: a v_fmac_f32_dpp that depends on the previous one issues back to back at the plain VALU rate
// (profiles/r02b_ubench_dpp.txt: 16 dependent terms = 80 cycles; 2 / 4 chains only add their final v_add).
// Hazard: a VALU write of the broadcast source must be 2 wait states old before a DPP read, and inline asm is opaque to the
// compiler's hazard recogniser -- the accumulator's initialisation (v_mov) and one v_nop are those two wait states
// (an s_nop 1 in front of the block measured +14 cycles per dot product).
#define PSNODE_DPP_HEAD "v_mov_b32 %0, %1\n\tv_nop\n\t"
#define PSNODE_DPP_FMAC(N, W) "v_fmac_f32_dpp %0, %2, %" #W " row_newbcast:" #N " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ float dot4(const float init, const float src, const float (&w)[16]) {
    float acc;
    asm(PSNODE_DPP_HEAD PSNODE_DPP_FMAC(0, 3) PSNODE_DPP_FMAC(1, 4) PSNODE_DPP_FMAC(2, 5) PSNODE_DPP_FMAC(3, 6)
        : "=&v"(acc) : "v"(init), "v"(src), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]));
    return acc;
}
__device__ __forceinline__ float dot8(const float init, const float src, const float (&w)[16]) {
    float acc;
    asm(PSNODE_DPP_HEAD PSNODE_DPP_FMAC(0, 3) PSNODE_DPP_FMAC(1, 4) PSNODE_DPP_FMAC(2, 5) PSNODE_DPP_FMAC(3, 6)
        PSNODE_DPP_FMAC(4, 7) PSNODE_DPP_FMAC(5, 8) PSNODE_DPP_FMAC(6, 9) PSNODE_DPP_FMAC(7, 10)
        : "=&v"(acc) : "v"(init), "v"(src), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]));
    return acc;
}
__device__ __forceinline__ float dot16(const float init, const float src, const float (&w)[16]) {
    float acc;
    asm(PSNODE_DPP_HEAD PSNODE_DPP_FMAC(0, 3) PSNODE_DPP_FMAC(1, 4) PSNODE_DPP_FMAC(2, 5) PSNODE_DPP_FMAC(3, 6)
        PSNODE_DPP_FMAC(4, 7) PSNODE_DPP_FMAC(5, 8) PSNODE_DPP_FMAC(6, 9) PSNODE_DPP_FMAC(7, 10)
        PSNODE_DPP_FMAC(8, 11) PSNODE_DPP_FMAC(9, 12) PSNODE_DPP_FMAC(10, 13) PSNODE_DPP_FMAC(11, 14)
        PSNODE_DPP_FMAC(12, 15) PSNODE_DPP_FMAC(13, 16) PSNODE_DPP_FMAC(14, 17) PSNODE_DPP_FMAC(15, 18)
        : "=&v"(acc) : "v"(init), "v"(src), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]),
          "v"(w[8]), "v"(w[9]), "v"(w[10]), "v"(w[11]), "v"(w[12]), "v"(w[13]), "v"(w[14]), "v"(w[15]));
    return acc;
}
#undef PSNODE_DPP_FMAC
#undef PSNODE_DPP_HEAD
// first layer of an encoder: in-features `quads`*4 <= 16 (the weights beyond in_dim are zero), wave-uniform choice
__device__ __forceinline__ float dot_in(const float init, const float src, const float (&w)[16], const int quads) {
    if (quads == 1) return dot4(init, src, w);
    if (quads == 2) return dot8(init, src, w);
    return dot16(init, src, w);
}

// scalar form of psnode_common.h:elu_pair (same clamps, same polynomial, same exact cancellation of the exp term)
struct EluS {
    float knee, neg_t0;
    __device__ __forceinline__ float operator()(const float x) const {
#ifndef PSNODE_ELU_EXPM1
        // (round 5: the clamp of the negative side rides on v_exp_f32's output modifier -- bit-identical, psnode_common.h: elu_pair)
        return fmaxf(x, __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(x * kLog2e), 0.0f, 1.0f) - 1.0f);
#endif
        const float xc = __builtin_amdgcn_fmed3f(x, knee, 0.0f), xe = fminf(x, knee), xp = fmaxf(x, 0.0f);
        const float u = __builtin_amdgcn_exp2f(xe * kLog2e) + neg_t0;
        float q = fmaf(xc, 0.007513605989515781f, 0.04149065539240837f);
        q = fmaf(xc, q, 0.16665108501911163f);
        q = fmaf(xc, q, 0.4999995231628418f);
        q = fmaf(xc, q, 1.0f);
        return xp + fmaf(xc, q, u);
    }
};

struct Mlp2Dev {          // Linear(in, H) ELU Linear(H, out): raw nn.Linear tensors
    const float *w1, *b1, *w2, *b2;
    int in, out;
};

struct LatentDppDev {
    int method, xd, zd;   // ENC: raw widths; latent-only: both = LH
    long long T, B;
    Mlp2Dev xenc, zenc, xdec;            // ENC only
    const float *de_w1, *de_b1, *de_w2, *de_b2;
    ViewDev t, x, z;
    const float* a0;                     // latent-only: [B, 2H]
    const int* ev;
    const float* zj;
    long long zjb, zje;
    float* xo;                           // ENC: x_pred [T,B,xd]; latent-only: xs [T,B,H]
    float* xre;                          // ENC: reconstruction view (may be null)
    long long xre_st, xre_sb;
    float* xh_out;                       // ENC: optional latent trajectory [T,B,H]
};

// row u of a [rows, ld] matrix, `count` valid columns starting at column c0, zero-padded to 16 registers
__device__ __forceinline__ void load_row(float (&w)[16], const float* m, const int ld, const int u, const int c0, const int count,
                                         const bool row_ok = true) {
#pragma unroll
    for (int j = 0; j < 16; ++j) w[j] = (row_ok && j < count) ? m[(long long)u * ld + c0 + j] : 0.0f;
}


// ---------------------------------------------------------------------------------------------------------------------------
// K3f, two-role form of the fused model (round 5): the plain inference call -- no event in the table, no latent trajectory wanted, 32-bit row
// offsets.  The integration wave keeps ONLY the latent stages; everything that is a row-wise MLP moves to its partner wave (wave w + 4, same
// SIMD) and onto MFMA tiles of 16 grid points (psnode_rows.hip's plan: D rows of one layer are the B operands of the next):
//   ahead of the chain   cz'[r] = F_z . z_encoder(z[r-1])          (zd -> H -> H, the two linear maps folded: 1 + 4 MFMAs per 16 rows and trajectory)
//   behind it            x_pred[r] = x_decoder(Xh_solution[r])     (H -> H -> xd: 4 + 4)
//   beside it            x_re[r] = x_decoder(x_encoder(x[r]))      (2 + 4 + 4, folded likewise)
// handed over through two LDS rings of 2 x 16 rows per chain wave (cz' in, latent state out), ONE workgroup barrier per 16 steps.  Per step the
// chain wave is left with  cz = c0 + cz'[r] (one ds_read),  S x (dot16, ELU, dot16),  the RK update and one ds_write: 185 instead of 250 VALU
// instructions at RK4, 60 instead of 125 at Euler -- the encoder of z, the decoder and their ELUs were 30 + 43 dependent VALU instructions
// of every step of a latency-bound wave.
typedef float f4l __attribute__((ext_vector_type(4)));
constexpr int kRingRows = 32, kRingFloats = kRingRows * 64;       // per chain wave and ring: [row & 31][trajectory 4][unit 16]

__device__ __forceinline__ f4l mf16(const float wa, const float vb, const f4l c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(wa, vb, c, 0, 0, 0); }
// one H -> (<= 16) layer of a tile: A = the lane's four weights (k-slots 4g .. 4g+3 of output row j), B = the four rows of the input tile
__device__ __forceinline__ f4l layer16(const float (&w)[4], const f4l in, const f4l bias) {
    f4l pA = mf16(w[0], in[0], bias), pB = mf16(w[1], in[1], f4l{0.f, 0.f, 0.f, 0.f});
    pA = mf16(w[2], in[2], pA); pB = mf16(w[3], in[3], pB);
    return pA + pB;
}

template <int METHOD>
__device__ __forceinline__ void enc_two_role(const LatentDppDev& a, float* __restrict__ lds, const int lane, const int wv) {
    const int nT = (int)a.T, nblk = (nT + 15) >> 4, cw = wv & 3;
    float* czr = lds + cw * kRingFloats;
    float* xr = lds + 4 * kRingFloats + cw * kRingFloats;
    const long long bw = (long long)blockIdx.x * DTB + cw * 4;           // the pair's four trajectories
    const int xd = a.xd, zd = a.zd;
    if (wv >= 4) {
        // ------------------------------------------------------------------ rows wave
        const int g = lane >> 4, j = lane & 15;
        const int nme = (xd + 3) >> 2, nmz = (zd + 3) >> 2;
        float w1e[4], w1d[4], w2d[4], w1z[4];
        f4l b1e, b1d, b2d, b1z;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int ce = nme * g + m, cz_ = nmz * g + m, k4 = 4 * g + m;
            w1e[m] = (m < nme && ce < xd) ? a.xenc.w1[j * xd + ce] : 0.0f;
            w1d[m] = a.xdec.w1[j * LH + k4];
            w2d[m] = j < xd ? a.xdec.w2[j * LH + k4] : 0.0f;
            w1z[m] = (m < nmz && cz_ < zd) ? a.zenc.w1[j * zd + cz_] : 0.0f;
            b1e[m] = a.xenc.b1[k4]; b1d[m] = a.xdec.b1[k4]; b2d[m] = k4 < xd ? a.xdec.b2[k4] : 0.0f;
            b1z[m] = a.zenc.b1[k4];
        }
        // Two pairs of layers have no ELU between them and are folded once per launch (one 16 x 16 x 16 product each, fp32 FMAs in k order):
        //   cz' = F_z (W2z h + b2z)           = (F_z W2z) h + F_z b2z                  z_encoder's output layer and the DE's z block
        //   dec L1 (enc L2 (h)) in x_re       = (W1d W2e) h + (W1d b2e + b1d)          x_encoder's output layer and x_decoder's input layer
        // 8 MFMAs less per 16 rows and trajectory (31 -> 23); each product is rounded once more than the unfolded chain (1e-7 relative).
        float mz[4], mr[4];
        f4l bzp, brp;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int k4 = 4 * g + m;
            float sz_ = 0.0f, sr_ = 0.0f, bz_ = 0.0f, br_ = a.xdec.b1[k4];
            for (int k = 0; k < LH; ++k) {
                const float fjk = a.de_w1[j * 6 * LH + 5 * LH + k] + a.de_w1[j * 6 * LH + 3 * LH + k];
                sz_ = fmaf(fjk, a.zenc.w2[k * LH + k4], sz_);
                sr_ = fmaf(a.xdec.w1[j * LH + k], a.xenc.w2[k * LH + k4], sr_);
                const float fuk = a.de_w1[k4 * 6 * LH + 5 * LH + k] + a.de_w1[k4 * 6 * LH + 3 * LH + k];
                bz_ = fmaf(fuk, a.zenc.b2[k], bz_);
                br_ = fmaf(a.xdec.w1[k4 * LH + k], a.xenc.b2[k], br_);
            }
            mz[m] = sz_; mr[m] = sr_; bzp[m] = bz_; brp[m] = br_;
        }
        const int ntraj = a.B - bw >= 4 ? 4 : (a.B - bw > 0 ? (int)(a.B - bw) : 0);
        const bool recon = a.xre != nullptr;
        // Raw inputs of a block travel a block ahead of their use: z rows r - 1 (the external input of the step that ENDS at row r), x rows r.
        // A tile is 4 grid points x the pair's 4 trajectories (row j = (step j >> 2, trajectory j & 3)): a block of 16 steps is four tiles, written as
        // straight-line code -- four independent MFMA chains for the scheduler to interleave (one tile alone is a dependent chain of 10) -- and the
        // rows a store instruction writes are the 4 trajectories of one grid point side by side (128 contiguous bytes of the time-major output;
        // 16 grid points of ONE trajectory per tile scattered every 32-byte row into its own DRAM page).  A trajectory beyond the batch re-reads
        // the first one and stores nothing.
        const int tj = j & 3, ts = j >> 2;
        const bool tok = tj < ntraj;
        const long long btj_raw = bw + (tok ? tj : 0);
        const long long btj = btj_raw < a.B ? btj_raw : a.B - 1;      // (a pair whose trajectories all lie beyond the batch: bw itself is out of range)
        float vz[4][4], vx[4][4];
        auto request_z = [&](const int blk) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = 16 * blk + 4 * q + ts, rz = r - 1 < 0 ? 0 : (r - 1 < nT ? r - 1 : nT - 1);
                const float* sz = a.z.p + btj * a.z.sb + (long long)rz * a.z.st + nmz * g;
#pragma unroll
                for (int m = 0; m < 4; ++m) vz[q][m] = (m < nmz && nmz * g + m < zd) ? sz[m] : 0.0f;
            }
        };
        auto request_x = [&](const int blk) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = 16 * blk + 4 * q + ts, rx = r < nT ? r : nT - 1;
                const float* sx = a.x.p + btj * a.x.sb + (long long)rx * a.x.st + nme * g;
#pragma unroll
                for (int m = 0; m < 4; ++m) vx[q][m] = (m < nme && nme * g + m < xd) ? sx[m] : 0.0f;
            }
        };
        auto ring_at = [&](float* ring_, const int blk, const int q) -> float* {      // ring row of tile q's row of this lane: [row & 31][trajectory][unit]
            return ring_ + ((16 * blk + 4 * q + ts) & (kRingRows - 1)) * 64 + tj * 16 + 4 * g;
        };
        auto ztiles = [&](const int blk) {           // from the registers request_z(blk) filled
            f4l acc[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[q] = b1z;
#pragma unroll
                for (int m = 0; m < 4; ++m) if (m < nmz) acc[q] = mf16(w1z[m], vz[q][m], acc[q]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = layer16(mz, elu_quad(acc[q]), bzp);
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<f4l*>(ring_at(czr, blk, q)) = acc[q];
        };
        auto decode4 = [&](f4l (&xh)[4], float* base, const long long st_t, const long long st_b, const int blk, auto l1_done) {   // rows of x_decoder, stored
            if constexpr (!decltype(l1_done)::value) {
#pragma unroll
                for (int q = 0; q < 4; ++q) xh[q] = layer16(w1d, xh[q], b1d);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) xh[q] = layer16(w2d, elu_quad(xh[q]), b2d);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = 16 * blk + 4 * q + ts;
                if (tok && r < nT) {
                    float* dst = base + (bw + tj) * st_b + (long long)r * st_t + 4 * g;
#pragma unroll
                    for (int m = 0; m < 4; ++m) if (4 * g + m < xd) dst[m] = xh[q][m];
                }
            }
        };
        auto recon_tiles = [&](const int blk) {
            f4l acc[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[q] = b1e;
#pragma unroll
                for (int m = 0; m < 4; ++m) if (m < nme) acc[q] = mf16(w1e[m], vx[q][m], acc[q]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = layer16(mr, elu_quad(acc[q]), brp);      // = x_decoder's first layer of the encoded row
            decode4(acc, a.xre, a.xre_st, a.xre_sb, blk, std::true_type{});
        };
        auto decode_tiles = [&](const int blk) {
            f4l xh[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) xh[q] = *reinterpret_cast<const f4l*>(ring_at(xr, blk, q));
            decode4(xh, a.xo, a.B * xd, xd, blk, std::false_type{});
        };
        request_z(0);
        ztiles(0);
        request_z(1);
        if (recon) request_x(0);
        __syncthreads();                             // cz' of block 0 is in the ring
        for (int n = 0; n < nblk; ++n) {
            if (n + 1 < nblk) {
                ztiles(n + 1);                       // (z rows requested during block n - 1)
                request_z(n + 2);
            }
            if (recon) {
                recon_tiles(n);                      // (x rows requested during block n - 1)
                request_x(n + 1);
            }
            if (n >= 1) decode_tiles(n - 1);
            __syncthreads();                         // the chain has written block n's states and may read block n + 1's cz'
        }
        decode_tiles(nblk - 1);
        return;
    }
    // ---------------------------------------------------------------------- integration wave
    const int u = lane & 15, row = lane >> 4;
    const long long b_raw = bw + row;
    const long long b = b_raw < a.B ? b_raw : a.B - 1;
    EluS elu;
    elu.knee = elu_knee();
    elu.neg_t0 = -__builtin_amdgcn_exp2f(elu.knee * kLog2e);
    float fx[16], w2[16];
    {
        float ws[16], wd[16];
        load_row(ws, a.de_w1, 6 * LH, u, 4 * LH, LH);
        load_row(wd, a.de_w1, 6 * LH, u, 2 * LH, LH);
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) fx[jj] = ws[jj] + wd[jj];
    }
    load_row(w2, a.de_w2, LH, u, 0, LH);
    const float b2 = a.de_b2[u];
    float x, c0 = a.de_b1[u];
    {   // Xh[0], Zh[0] (all_initial) and the constant part of L1 -- once per launch, on the DPP dot products
        float wa[16], wb[16];
        load_row(wa, a.xenc.w1, xd, u, 0, xd);
        load_row(wb, a.xenc.w2, LH, u, 0, LH);
        const float xraw = u < xd ? a.x.p[b * a.x.sb + u] : 0.0f;
        const float a0x = dot16(a.xenc.b2[u], elu(dot_in(a.xenc.b1[u], xraw, wa, (xd + 3) >> 2)), wb);
        load_row(wa, a.zenc.w1, zd, u, 0, zd);
        load_row(wb, a.zenc.w2, LH, u, 0, LH);
        const float zraw = a.z.p[b * a.z.sb + (u < zd ? u : 0)];
        const float a0z = dot16(a.zenc.b2[u], elu(dot_in(a.zenc.b1[u], zraw, wa, (zd + 3) >> 2)), wb);
        x = a0x;
        load_row(wa, a.de_w1, 6 * LH, u, 0, LH);
        load_row(wb, a.de_w1, 6 * LH, u, 2 * LH, LH);
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) wa[jj] -= wb[jj];
        c0 = dot16(c0, a0x, wa);
        load_row(wa, a.de_w1, 6 * LH, u, LH, LH);
        load_row(wb, a.de_w1, 6 * LH, u, 3 * LH, LH);
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) wa[jj] -= wb[jj];
        c0 = dot16(c0, a0z, wa);
    }
    auto rhs = [&](const float xs, const float cz) -> float { return dot16(b2, elu(dot16(cz, xs, fx)), w2); };
    auto as_g = [](const float* q) { return (gptr<const float>)(uintptr_t)q; };
    const unsigned toff = (unsigned)(b * a.t.sb) * 4u;
    const long long tst = a.t.st;
    xr[lane] = x;                                    // row 0 = the start state
    float t_cur = ldg<float>(as_g(a.t.p), toff);
    float tq[4];                                     // slot j of a chunk of four rows: t[row]
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) tq[jj] = ldg<float>(as_g(a.t.p + (long long)(jj < nT ? jj : nT - 1) * tst), toff);
    const float* trun = a.t.p + 4 * tst;             // the row the next refill reads (full chunks only)
    __syncthreads();                                 // cz' of block 0
    float czv = czr[64 + lane];                      // cz' of the row in hand, read a step ahead of its use (row 1 first; rows of the NEXT block
                                                     // only behind the barrier that completes them)
    auto step_row = [&](const int r, const float tr) {        // the step that ends at row r >= 1
        const float h_ = tr - t_cur;
        const int so = (r & (kRingRows - 1)) * 64 + lane;
        const float cz = c0 + czv;
        if (((r + 1) & 15) != 0) czv = czr[((r + 1) & (kRingRows - 1)) * 64 + lane];
        const float k1 = rhs(x, cz);
        if constexpr (METHOD == PSNODE_EULER) {
            x = x + h_ * k1;
        } else if constexpr (METHOD == PSNODE_MIDPOINT) {
            const float k2 = rhs(x + k1 * (0.5f * h_), cz);
            x = x + h_ * k2;
        } else {
            const float k2 = rhs(x + h_ * k1 * kOneThird, cz);
            const float k3 = rhs(x + h_ * (k2 - k1 * kOneThird), cz);
            const float k4 = rhs(x + h_ * (k1 - k2 + k3), cz);
            x = x + (k1 + 3.0f * (k2 + k3) + k4) * h_ * 0.125f;
        }
        xr[so] = x;
    };
    int r0 = 0;
    for (int n = 0; n < nblk; ++n) {
        for (int c = 0; c < 4; ++c, r0 += 4) {
            if (r0 >= 4 && r0 + 8 <= nT) {           // a full chunk whose refills stay inside the grid: no per-row tests
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float tr = tq[jj];
                    tq[jj] = ldg<float>(as_g(trun), toff);
                    trun += tst;
                    step_row(r0 + jj, tr);
                    t_cur = tr;
                }
            } else {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int r = r0 + jj;
                    const float tr = tq[jj];
                    const int rn = r + 4 < nT ? r + 4 : nT - 1;
                    tq[jj] = ldg<float>(as_g(a.t.p + (long long)rn * tst), toff);
                    if (r >= 1 && r < nT) step_row(r, tr);
                    if (r < nT) t_cur = tr;
                }
                trun = a.t.p + (long long)(r0 + 8) * tst;
            }
        }
        __syncthreads();
        czv = czr[(r0 & (kRingRows - 1)) * 64 + lane];           // first row of the next block
    }
}

template <int METHOD, bool ENC>
__global__ __launch_bounds__(ENC ? 512 : 256) void latent_dpp_kernel(const LatentDppDev a) {
    const int lane = threadIdx.x & 63, u = lane & 15, row = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long b_raw = (long long)blockIdx.x * DTB + (wv & 3) * 4 + row;
    const bool valid = b_raw < a.B;
    const long long b = valid ? b_raw : a.B - 1;
    const long long nT = a.T;
    EluS elu;
    elu.knee = elu_knee();
    elu.neg_t0 = -__builtin_amdgcn_exp2f(elu.knee * kLog2e);

    if constexpr (ENC) {
        // the plain inference call takes the two-role form (enc_two_role above); the decision is uniform over the workgroup
        __shared__ float ring[ENC ? 8 * kRingFloats : 1];
        {
            const unsigned long long span_t = (unsigned long long)a.B * (unsigned long long)(a.t.sb < 0 ? 0 : a.t.sb) * 4ull;
            bool two = nT >= 2 && nT < (1ll << 31) && !a.xh_out && a.t.sb >= 0 && span_t < (1ull << 32);
            if (two && a.ev) {
                int any = -1;
                for (int i = lane; i + 1 < (int)nT; i += 64) any = max(any, a.ev[i]);
                two = __builtin_amdgcn_ballot_w64(any >= 0) == 0;
            }
            if (two) {
                enc_two_role<METHOD>(a, ring, lane, wv);
                return;
            }
        }
        if (wv >= 4) {
            // ---------------- reconstruction waves: x_re[t] = x_decoder(x_encoder(x[t]))  (neural_00_ODE_02_direct_encode.py:87)
            // Round 5: on MFMA, K3b's way (psnode_rows.hip) -- a tile is 16 grid points of ONE trajectory, the hidden / latent tiles stay in
            // registers through all four layers (D rows of one layer are the B operands of the next: no exchange), 14 MFMAs + 2 ELU quads per
            // 16 rows.  The DPP form of rounds 2-4 (a lane per (trajectory, unit), 4 rows per 76 v_fmac_f32_dpp) cost ~110 issue cycles per
            // row on the SIMD the latency-bound integration wave shares; this is ~35, and the waves are done after a tenth of the launch.
            if (!a.xre) return;
            typedef float f4 __attribute__((ext_vector_type(4)));
            auto mf = [](const float wa, const float vb, const f4 c) -> f4 { return __builtin_amdgcn_mfma_f32_16x16x4f32(wa, vb, c, 0, 0, 0); };
            const int g = lane >> 4, j = lane & 15;      // lane group g = k-slot of the MFMAs, j = the tile's row (B, D) / the weight's row (A)
            const int xd = a.xd, nm = (xd + 3) >> 2;     // first-layer MFMAs of the encoder: lane group g supplies input columns nm g + m
            float w1e[4], w2e[4], w1d[4], w2d[4];
            f4 b1e, b2e, b1d, b2d;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int c = nm * g + m;
                w1e[m] = (m < nm && c < xd) ? a.xenc.w1[j * xd + c] : 0.0f;
                w2e[m] = a.xenc.w2[j * LH + 4 * g + m];
                w1d[m] = a.xdec.w1[j * LH + 4 * g + m];
                w2d[m] = j < xd ? a.xdec.w2[j * LH + 4 * g + m] : 0.0f;
                b1e[m] = a.xenc.b1[4 * g + m];
                b2e[m] = a.xenc.b2[4 * g + m];
                b1d[m] = a.xdec.b1[4 * g + m];
                b2d[m] = 4 * g + m < xd ? a.xdec.b2[4 * g + m] : 0.0f;
            }
            const long long bw = (long long)blockIdx.x * DTB + (wv & 3) * 4;      // this wave's four trajectories
            const int ntraj = a.B - bw >= 4 ? 4 : (int)(a.B - bw);
            if (ntraj <= 0) return;
            const long long xst = a.x.st;
            auto load_tile = [&](const int q, const long long t0, float (&v)[4]) {          // rows t0 .. t0 + 15 of trajectory bw + q
                const long long r = t0 + j, rc = r < nT ? r : nT - 1;
                const float* src = a.x.p + (bw + q) * a.x.sb + rc * xst + nm * g;
#pragma unroll
                for (int m = 0; m < 4; ++m) v[m] = (m < nm && nm * g + m < xd) ? src[m] : 0.0f;
            };
            float vn[4];
            load_tile(0, 0, vn);
            int q = 0, qn = 0;              // (q, t0): the tile in hand; (qn, tn): the one requested ahead
            long long t0 = 0, tn = 0;
            for (;;) {
                float v[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) v[m] = vn[m];
                tn += 16;
                if (tn >= nT) { tn = 0; ++qn; }
                const bool more = qn < ntraj;
                if (more) load_tile(qn, tn, vn);                     // one tile ahead
                f4 acc = b1e;
#pragma unroll
                for (int m = 0; m < 4; ++m) if (m < nm) acc = mf(w1e[m], v[m], acc);
                const f4 he = elu_quad(acc);
                f4 pA = mf(w2e[0], he[0], b2e), pB = mf(w2e[1], he[1], f4{0.f, 0.f, 0.f, 0.f});
                pA = mf(w2e[2], he[2], pA); pB = mf(w2e[3], he[3], pB);
                const f4 xh = pA + pB;                               // the latent rows: unit 4g + r of row j
                pA = mf(w1d[0], xh[0], b1d); pB = mf(w1d[1], xh[1], f4{0.f, 0.f, 0.f, 0.f});
                pA = mf(w1d[2], xh[2], pA); pB = mf(w1d[3], xh[3], pB);
                const f4 hd = elu_quad(pA + pB);
                pA = mf(w2d[0], hd[0], b2d); pB = mf(w2d[1], hd[1], f4{0.f, 0.f, 0.f, 0.f});
                pA = mf(w2d[2], hd[2], pA); pB = mf(w2d[3], hd[3], pB);
                const f4 o = pA + pB;
                const long long r = t0 + j;
                if (r < nT) {
                    float* dst = a.xre + (bw + q) * a.xre_sb + r * a.xre_st + 4 * g;
#pragma unroll
                    for (int m = 0; m < 4; ++m) if (4 * g + m < xd) dst[m] = o[m];
                }
                if (!more) break;
                q = qn; t0 = tn;
            }
            return;
        }
    }

    // ---------------- integration waves
    float fx[16], fz[16], w2[16];
    {
        float ws[16], wd[16];
        load_row(ws, a.de_w1, 6 * LH, u, 4 * LH, LH);
        load_row(wd, a.de_w1, 6 * LH, u, 2 * LH, LH);
#pragma unroll
        for (int j = 0; j < 16; ++j) fx[j] = ws[j] + wd[j];
        load_row(ws, a.de_w1, 6 * LH, u, 5 * LH, LH);
        load_row(wd, a.de_w1, 6 * LH, u, 3 * LH, LH);
#pragma unroll
        for (int j = 0; j < 16; ++j) fz[j] = ws[j] + wd[j];
    }
    load_row(w2, a.de_w2, LH, u, 0, LH);
    const float b2 = a.de_b2[u];

    // encoders / decoder of the fused model (dead code in the latent-only instantiation)
    float w1z[16], w2z[16], w1d[16], w2d[16];
    float b1z = 0.f, b2z = 0.f, b1d = 0.f, b2d = 0.f;
    int zq = 4;
    if constexpr (ENC) {
        load_row(w1z, a.zenc.w1, a.zd, u, 0, a.zd);
        load_row(w2z, a.zenc.w2, LH, u, 0, LH);
        load_row(w1d, a.xdec.w1, LH, u, 0, LH);
        load_row(w2d, a.xdec.w2, LH, u, 0, LH, u < a.xd);
        b1z = a.zenc.b1[u]; b2z = a.zenc.b2[u]; b1d = a.xdec.b1[u]; b2d = u < a.xd ? a.xdec.b2[u] : 0.0f;
        zq = (a.zd + 3) >> 2;
    }
    const int zlanes = ENC ? a.zd : LH;      // lanes of a row that carry one z column each
    auto encode_z = [&](const float zraw) -> float {   // ENC: z_encoder on this row's raw z; latent-only: already encoded
        if constexpr (!ENC) return zraw;
        return dot16(b2z, elu(dot_in(b1z, zraw, w1z, zq)), w2z);
    };
    auto emit = [&](const long long k, const float xlat) {   // output row k from the latent state
        if constexpr (ENC) {
            const float o = dot16(b2d, elu(dot16(b1d, xlat, w1d)), w2d);
            if (valid && u < a.xd) a.xo[(k * a.B + b) * a.xd + u] = o;
            if (a.xh_out && valid) a.xh_out[(k * a.B + b) * LH + u] = xlat;
        } else {
            if (valid) a.xo[(k * a.B + b) * LH + u] = xlat;
        }
    };

    const float* zp = a.z.p + b * a.z.sb + (u < zlanes ? u : 0);
    const float* zjp = a.zj ? a.zj + b * a.zjb + (u < zlanes ? u : 0) : zp;
    // the two strides are pinned in SGPRs: left to itself the compiler selects between their KERNARG ADDRESSES and issues a scalar
    // load + s_waitcnt lgkmcnt(0) inside every time step (470 stalled cycles per step, profiles/r02b_latent16_pmc_sq.txt)
    long long zst = a.z.st, zje = a.zje;
    asm volatile("" : "+s"(zst), "+s"(zje));
    auto load_z = [&](const long long k, const int ev) -> float {
        if (u >= zlanes) return 0.0f;
        const long long off = ev >= 0 ? ev * zje : k * zst;
        return (ev >= 0 ? zjp : zp)[off];
    };

    // ---- initial state and the constant part of L1
    float x, a0x, a0z;
    if constexpr (ENC) {
        float w1e[16], w2e[16];
        load_row(w1e, a.xenc.w1, a.xd, u, 0, a.xd);
        load_row(w2e, a.xenc.w2, LH, u, 0, LH);
        const float xraw = u < a.xd ? a.x.p[b * a.x.sb + u] : 0.0f;
        a0x = dot16(a.xenc.b2[u], elu(dot_in(a.xenc.b1[u], xraw, w1e, (a.xd + 3) >> 2)), w2e);      // Xh[0]
        a0z = encode_z(load_z(0, -1));                                                             // Zh[0] (never jumped)
        x = a0x;
    } else {
        a0x = a.a0[b * 2 * LH + u];
        a0z = a.a0[b * 2 * LH + LH + u];
        x = a.x.p[b * a.x.sb + u];
    }
    float c0 = a.de_b1[u];
    {
        float wa[16], wd[16];
        load_row(wa, a.de_w1, 6 * LH, u, 0, LH);
        load_row(wd, a.de_w1, 6 * LH, u, 2 * LH, LH);
#pragma unroll
        for (int j = 0; j < 16; ++j) wa[j] -= wd[j];
        c0 = dot16(c0, a0x, wa);
        load_row(wa, a.de_w1, 6 * LH, u, LH, LH);
        load_row(wd, a.de_w1, 6 * LH, u, 3 * LH, LH);
#pragma unroll
        for (int j = 0; j < 16; ++j) wa[j] -= wd[j];
        c0 = dot16(c0, a0z, wa);
    }
    emit(0, x);
    if (nT < 2) return;

    auto rhs = [&](const float xs, const float cz) -> float { return dot16(b2, elu(dot16(cz, xs, fx)), w2); };

    const float* tp = a.t.p + b * a.t.sb;
    const long long tst = a.t.st;
    // Event indices travel 64 steps at a time: lane i of `evb` holds event_idx[64*blk + i] (one coalesced 256-byte load per 64
    // steps, waited for on the spot -- one memory round trip per 64 steps), a step reads its entry with v_readlane.  A per-step
    // scalar or vector load of the table put a full memory round trip inside EVERY step (s_waitcnt right behind the load: 470 of
    // 1700 cycles per step).
    auto load_evb = [&](const long long blk) -> int {
        const long long i = blk * 64 + lane;
        const int v = (a.ev && i + 1 < nT) ? a.ev[i] : -1;
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
        return v;
    };
    int evb = load_evb(0);
    // A step is ~1.2 k cycles (0.5 us): one step of look-ahead does not cover an HBM miss (a new 64-byte line of t every 16 steps,
    // of z every 16/zd steps per trajectory), so the step inputs run PF steps ahead in a register ring (time loop unrolled by PF).
    float t_cur = tp[0];
    float tq[PF], zring[PF];      // tq[j] = t[s+1], zring[j] = external input of step s, for the step s = chunk base + j
#pragma unroll
    for (int j = 0; j < PF; ++j) {
        const bool live = j + 1 < nT;
        tq[j] = live ? tp[(j + 1) * tst] : 0.0f;
        zring[j] = live ? load_z(j, __builtin_amdgcn_readlane(evb, j)) : 0.0f;
    }
    // FAST main loop (round 5, K1x's lesson: with one latency-bound wave per SIMD every scalar instruction and branch of the per-step bookkeeping is
    // wall time -- 64-bit end-of-grid tests on the VALU, the event-table block test, the divergent `lane carries a z column` load, 64-bit
    // address products were ~20 branches and ~50 SALU instructions per step): whole chunks of PF steps whose prefetches stay inside the grid,
    // no event in the table (scanned once), 32-bit counters, uniform running row bases + one 32-bit lane offset per array.  The general loop
    // below finishes the grid from wherever this one stops (same ring invariant: slot j holds the inputs of step k + j).
    long long k = 0;
    {
        const int nTi = (int)nT;
        const unsigned long long span_t = (unsigned long long)a.B * (unsigned long long)(a.t.sb < 0 ? 0 : a.t.sb) * 4ull;
        const unsigned long long span_z = (unsigned long long)a.B * (unsigned long long)(a.z.sb < 0 ? 0 : a.z.sb) * 4ull + 64ull;
        const unsigned long long span_o = (unsigned long long)a.B * (ENC ? a.xd : LH) * 4ull;
        bool fast = nT < (1ll << 31) && !a.xh_out && a.t.sb >= 0 && a.z.sb >= 0 && span_t < (1ull << 32) && span_z < (1ull << 32) && span_o < (1ull << 32);
        if (fast && a.ev) {
            int any = -1;
            for (int i = lane; i + 1 < nTi; i += 64) any = max(any, a.ev[i]);
            fast = __builtin_amdgcn_ballot_w64(any >= 0) == 0;
        }
        if (fast && 2 * PF < nTi) {
            auto as_g = [](const float* q) { return (gptr<const float>)(uintptr_t)q; };
            const unsigned toff = (unsigned)(b * a.t.sb) * 4u;
            const unsigned zoff = (unsigned)(b * a.z.sb + (u < zlanes ? u : 0)) * 4u;     // lanes without a column read column 0 (finite) against a zero weight
            const int od = ENC ? a.xd : LH;
            const unsigned ooff = (unsigned)(b * od + (u < od ? u : 0)) * 4u;
            const bool st_ok = valid && u < od;
            const float* trun = a.t.p + (PF + 1) * tst;          // step 0 refills its slot with t[PF + 1] / the z row of step PF
            const float* zrun = a.z.p + PF * zst;
            float* orun = a.xo + a.B * od;                       // step 0 writes row 1
            const long long ostep = a.B * od;
            int ki = 0;
            for (; ki + 2 * PF < nTi; ki += PF) {
#pragma unroll
                for (int j = 0; j < PF; ++j) {
                    const float h_ = tq[j] - t_cur;
                    t_cur = tq[j];
                    const float zraw = zring[j];
                    tq[j] = ldg<float>(as_g(trun), toff);
                    zring[j] = ldg<float>(as_g(zrun), zoff);
                    trun += tst;
                    zrun += zst;
                    const float cz = dot16(c0, encode_z(zraw), fz);
                    const float k1 = rhs(x, cz);
                    if constexpr (METHOD == PSNODE_EULER) {
                        x = x + h_ * k1;
                    } else if constexpr (METHOD == PSNODE_MIDPOINT) {
                        const float k2 = rhs(x + k1 * (0.5f * h_), cz);
                        x = x + h_ * k2;
                    } else {
                        const float k2 = rhs(x + h_ * k1 * kOneThird, cz);
                        const float k3 = rhs(x + h_ * (k2 - k1 * kOneThird), cz);
                        const float k4 = rhs(x + h_ * (k1 - k2 + k3), cz);
                        x = x + (k1 + 3.0f * (k2 + k3) + k4) * h_ * 0.125f;
                    }
                    float o = x;
                    if constexpr (ENC) o = dot16(b2d, elu(dot16(b1d, x, w1d)), w2d);
                    if (st_ok) stg<float>((gptr<float>)(uintptr_t)orun, ooff, o);
                    orun += ostep;
                }
            }
            k = ki;
        }
    }
    for (; k + 1 < nT; k += PF) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const long long st = k + j;
            if (st + 1 < nT) {
                const float h_ = tq[j] - t_cur;
                t_cur = tq[j];
                const float zraw = zring[j];
                const long long sn = st + PF;      // refill this ring slot for the step PF ahead
                if (sn + 1 < nT) {
                    tq[j] = tp[(sn + 1) * tst];
                    if ((sn & 63) == 0) evb = load_evb(sn >> 6);
                    zring[j] = load_z(sn, __builtin_amdgcn_readlane(evb, (int)(sn & 63)));
                }
                const float cz = dot16(c0, encode_z(zraw), fz);   // zero-order hold: constant over the stages
                const float k1 = rhs(x, cz);
                if constexpr (METHOD == PSNODE_EULER) {
                    x = x + h_ * k1;
                } else if constexpr (METHOD == PSNODE_MIDPOINT) {
                    const float k2 = rhs(x + k1 * (0.5f * h_), cz);
                    x = x + h_ * k2;
                } else {
                    const float k2 = rhs(x + h_ * k1 * kOneThird, cz);
                    const float k3 = rhs(x + h_ * (k2 - k1 * kOneThird), cz);
                    const float k4 = rhs(x + h_ * (k1 - k2 + k3), cz);
                    x = x + (k1 + 3.0f * (k2 + k3) + k4) * h_ * 0.125f;
                }
                emit(st + 1, x);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// K8f -- backward (discretise-then-optimise) through the hidden-16 latent ODE, same lane = (trajectory, unit) mapping: what
// loss.backward() computes through integrate_ODE between the encoders and the decoder of neural_00_ODE_02_direct_encode.py:74-89.
// Replaces K8 (one MFMA wave per 16 trajectories = one wave per CU at B = 4096, 3.85 ms) -- 1024 waves instead of 256.
//   forward recompute per stage  pre = cz + F_x . X_s ; h = ELU(pre) ; k_s = b2 + W2 . h                       (dot16 x2 + ELU)
//   backward per stage           dW2[u][:] += gk[u] * h[:]              (one MFMA over the wave's four trajectories, see below)
//                                d1 = (W2^T gk) * ELU'(pre)             (dot16 with the lane's COLUMN of W2)
//                                dF_x[u][:] += d1[u] * X_s[:] ;  gX_s = F_x^T d1
//   per step                     dF_z[u][:] += D1[u] * z[:] ; gz = F_z^T D1 ,  D1 = sum_s d1 (the external block is frozen)
//   per launch                   d all_initial = (Wa - Wd)^T sum_t D1 ;  dWa = sum_t D1 (x) a0 ;  dWd = dF - dWa ;  dWs = dF
// Weight gradients (round 5): dW[u][j] += sum over the wave's four trajectories of own[traj][u] * src[traj][j] IS one v_mfma_f32_16x16x4_f32
// with the two registers as they stand -- lane (traj, u) is element A[u][k = traj] of the first and B[k = traj][j = u] of the second -- so
// every rank-1 update of rounds 2-4 (16 v_fmac_f32_dpp + 2 v_nop per block and stage: 9 x 18 = 162 of the ~600 VALU instructions of an RK4
// step) is ONE MFMA, the accumulators are 4 registers per block instead of 16 (D: lane (g, j) holds dW[4g + r][j]) and already summed over
// the trajectories.  One partial vector per wave, summed by reduce_partials in a fixed order (deterministic).  No LDS, no barrier, any alignment.
struct LatentBwdDppDev {
    int method, n_events;
    long long T, B;
    const float *de_w1, *de_b1, *de_w2, *de_b2;
    ViewDev t, z;
    const float* a0;
    const int* ev;
    const float* zj;
    long long zjb, zje;
    const float *xs, *gout;
    float *gx0, *gz, *gzj, *ga0, *wpart;
};
constexpr int BK1 = 6 * LH, BNP = LH * BK1 + LH + LH * LH + LH;   // in_features of L1; parameters W1, b1, W2, b2

// Two roles (round 5, the plain call: no event in the table, 32-bit row offsets): phase A -- the recomputation of the stage inputs X_s and
// hidden rows h_s of a step from the saved state -- does not depend on the adjoint, so the steps of a block are independent of each other:
// the PARTNER wave (wave w + 4, same SIMD) runs it for 4 steps x 4 trajectories at a time as ONE 16-row MFMA tile (K3b's plan: cz = c0 + F_z Zh,
// then per stage pre = cz + F_x X_s, h_s = ELU, k_s = b2 + W2 h_s: 4 + 8 S MFMAs per 4 steps) and hands X_1.., h_0.. over through an LDS ring of
// two blocks, ONE workgroup barrier per 4 steps.  On the sweep wave phase A was 8 dependent 16-term DPP dot products + 4 ELUs of every RK4
// step (~190 of ~440 VALU instructions); it becomes 2 S - 1 ds_read_b32.
constexpr int kBwdVals = 7;                                  // values per (row, unit) in the ring: h_0 .. h_{S-1}, X_1 .. X_{S-1}
constexpr int kBwdRing = 2 * 16 * kBwdVals * 16;             // floats per sweep wave: [block & 1][trajectory 4 x step 4][value][unit]

template <int METHOD>
__global__ __launch_bounds__(512) void latent_ode_backward_dpp_kernel(const LatentBwdDppDev a) {
    constexpr int S = rk_stages(METHOD);
    const int lane = threadIdx.x & 63, u = lane & 15, row = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long b_raw = (long long)blockIdx.x * DTB + (wv & 3) * 4 + row;
    const bool valid = b_raw < a.B;
    const long long b = valid ? b_raw : a.B - 1;
    const long long nT = a.T;
    __shared__ float pring[4 * kBwdRing];
    // the FAST form and its number of 4-step blocks: one decision for the whole workgroup
    int nchunk = 0;
    {
        const unsigned long long rows_b = (unsigned long long)a.B * LH * 4ull;
        const unsigned long long span_t = (unsigned long long)a.B * (unsigned long long)(a.t.sb < 0 ? 0 : a.t.sb) * 4ull;
        const unsigned long long span_z = (unsigned long long)a.B * (unsigned long long)(a.z.sb < 0 ? 0 : a.z.sb) * 4ull + 64ull;
        bool fast = nT < (1ll << 31) && nT >= 2 * PF + 2 && a.t.sb >= 0 && a.z.sb >= 0 && rows_b < (1ull << 32) && span_t < (1ull << 32) && span_z < (1ull << 32);
        if (fast && a.ev) {
            int any = -1;
            for (int i = lane; i + 1 < (int)nT; i += 64) any = max(any, a.ev[i]);
            fast = __builtin_amdgcn_ballot_w64(any >= 0) == 0;
        }
        if (fast) nchunk = ((int)nT - 2 - (2 * PF - 1)) / PF + 1;      // blocks whose prefetches stay inside the grid (nT >= 2 PF + 2: >= 1)
    }
    if (wv >= 4) {
        // ------------------------------------------------------------------ partner wave: phase A on MFMA tiles
        if (nchunk == 0) return;
        const int g = lane >> 4, j = lane & 15, q = j >> 2, si = j & 3;      // tile row j = (trajectory q, step si of the block)
        float* ring = pring + (wv - 4) * kBwdRing;
        const long long bq_raw = (long long)blockIdx.x * DTB + (wv - 4) * 4 + q;
        const long long bq = bq_raw < a.B ? bq_raw : a.B - 1;
        float FxA[4], FzA[4], W2A[4], WaxA[4], WazA[4];
        f4l b2t, c0t, a0xv, a0zv;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int k4 = 4 * g + m;
            const float* r = a.de_w1 + (long long)j * BK1;                     // A operands: lane (g, i = j) holds M[i][4g + m]
            FxA[m] = r[4 * LH + k4] + r[2 * LH + k4];
            FzA[m] = r[5 * LH + k4] + r[3 * LH + k4];
            W2A[m] = a.de_w2[j * LH + k4];
            WaxA[m] = r[k4] - r[2 * LH + k4];
            WazA[m] = r[LH + k4] - r[3 * LH + k4];
            b2t[m] = a.de_b2[k4];
            c0t[m] = a.de_b1[k4];
            a0xv[m] = a.a0[bq * 2 * LH + k4];
            a0zv[m] = a.a0[bq * 2 * LH + LH + k4];
        }
        c0t = layer16(WazA, a0zv, layer16(WaxA, a0xv, c0t));                  // b1 + (Wa - Wd) a0: constant per trajectory
        const int ki0 = (int)nT - 2;
        f4l xn, zn;
        float tln, thn;
        auto request = [&](const int bi) {                                     // the saved state, Zh row and clock of step k = ki0 - 4 bi - si
            const long long k = ki0 - 4 * bi - si;
            const float* xr_ = a.xs + (k * a.B + bq) * LH + 4 * g;
            const float* zr_ = a.z.p + bq * a.z.sb + k * a.z.st + 4 * g;
            const float* tr_ = a.t.p + bq * a.t.sb + k * a.t.st;
#pragma unroll
            for (int m = 0; m < 4; ++m) { xn[m] = xr_[m]; zn[m] = zr_[m]; }
            tln = tr_[0]; thn = tr_[a.t.st];
        };
        auto produce = [&](const int bi, const f4l x0v, const f4l zv, const float h_) {
            float* dst = ring + (((bi & 1) * 16 + j) * kBwdVals) * 16 + 4 * g;
            const f4l cz = layer16(FzA, zv, c0t);
            f4l ks[S];
#pragma unroll
            for (int st = 0; st < S; ++st) {
                f4l X = x0v;
                if (st > 0) {
                    f4l acc = f4l{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int jj = 0; jj < st; ++jj) acc += rk_a(METHOD, st, jj) * ks[jj];
                    X = x0v + h_ * acc;
                    *reinterpret_cast<f4l*>(dst + (S + st - 1) * 16) = X;
                }
                const f4l hh = elu_quad(layer16(FxA, X, cz));
                *reinterpret_cast<f4l*>(dst + st * 16) = hh;
                ks[st] = layer16(W2A, hh, b2t);
            }
        };
        request(0);
        {
            const f4l x0v = xn, zv = zn;
            const float h_ = thn - tln;
            if (nchunk > 1) request(1);
            produce(0, x0v, zv, h_);
        }
        __syncthreads();
        for (int bi = 0; bi < nchunk; ++bi) {
            if (bi + 1 < nchunk) {
                const f4l x0v = xn, zv = zn;
                const float h_ = thn - tln;
                if (bi + 2 < nchunk) request(bi + 2);
                produce(bi + 1, x0v, zv, h_);
            }
            __syncthreads();
        }
        return;
    }
    EluS elu;
    elu.knee = elu_knee();
    elu.neg_t0 = -__builtin_amdgcn_exp2f(elu.knee * kLog2e);
    auto dact = [](const float h) -> float { return elu_grad(h); };   // ELU'(pre) from h = ELU(pre)

    // rows (forward) and columns (transposed products) of the folded blocks F = Ws + Wd and of W2
    float fx[16], fz[16], w2[16], fxT[16], fzT[16], w2T[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float* r = a.de_w1 + (long long)u * BK1;
        fx[j] = r[4 * LH + j] + r[2 * LH + j];
        fz[j] = r[5 * LH + j] + r[3 * LH + j];
        w2[j] = a.de_w2[u * LH + j];
        const float* c = a.de_w1 + (long long)j * BK1;
        fxT[j] = c[4 * LH + u] + c[2 * LH + u];
        fzT[j] = c[5 * LH + u] + c[3 * LH + u];
        w2T[j] = a.de_w2[j * LH + u];
    }
    const float b2 = a.de_b2[u];
    const float a0x = a.a0[b * 2 * LH + u], a0z = a.a0[b * 2 * LH + LH + u];
    float c0 = a.de_b1[u];
    {
        float wa[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) wa[j] = a.de_w1[(long long)u * BK1 + j] - a.de_w1[(long long)u * BK1 + 2 * LH + j];
        c0 = dot16(c0, a0x, wa);
#pragma unroll
        for (int j = 0; j < 16; ++j) wa[j] = a.de_w1[(long long)u * BK1 + LH + j] - a.de_w1[(long long)u * BK1 + 3 * LH + j];
        c0 = dot16(c0, a0z, wa);
    }

    typedef float f4b __attribute__((ext_vector_type(4)));
    auto rank1 = [](f4b& acc, const float src, const float own) { acc = __builtin_amdgcn_mfma_f32_16x16x4f32(own, src, acc, 0, 0, 0); };
    f4b aW2 = f4b{0.f, 0.f, 0.f, 0.f}, aFx = aW2, aFz = aW2;
    float S1 = 0.0f, SB2 = 0.0f, gcarry = 0.0f;

    const float* tp = a.t.p + b * a.t.sb;
    const long long tst = a.t.st;
    const float* zp = a.z.p + b * a.z.sb + u;
    const float* zjp = a.zj ? a.zj + b * a.zjb + u : zp;
    long long zst = a.z.st, zje = a.zje;
    asm volatile("" : "+s"(zst), "+s"(zje));
    auto load_z = [&](const long long k, const int ev) -> float {
        const long long off = ev >= 0 ? ev * zje : k * zst;
        return (ev >= 0 ? zjp : zp)[off];
    };
    auto load_evb = [&](const long long blk) -> int {      // event indices of steps 64 blk .. 64 blk + 63, one per lane
        const long long i = blk * 64 + lane;
        const int v = (a.ev && i + 1 < nT) ? a.ev[i] : -1;
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
        return v;
    };
    auto row_of = [&](const float* base, const long long k) -> float { return base[(k * a.B + b) * LH + u]; };

    // The sweep runs k = T-2 .. 0; step inputs PF steps ahead in a register ring (as the forward kernel).
    float tq[PF], zq_[PF], xq[PF], gq[PF];   // slot j of a chunk: t[k], z of step k, xs[k], dL/dxs[k+1]
    int evq[PF];
    int evb = 0;
    long long evb_blk = -1;
    auto fetch = [&](const int j, const long long k) {       // k >= 0
        if ((k >> 6) != evb_blk) { evb_blk = k >> 6; evb = load_evb(evb_blk); }
        evq[j] = __builtin_amdgcn_readlane(evb, (int)(k & 63));
        tq[j] = tp[k * tst];
        zq_[j] = load_z(k, evq[j]);
        xq[j] = row_of(a.xs, k);
        gq[j] = valid ? row_of(a.gout, k + 1) : 0.0f;
    };
    float t_hi = nT >= 2 ? tp[(nT - 1) * tst] : 0.0f;
#pragma unroll
    for (int j = 0; j < PF; ++j) {
        tq[j] = zq_[j] = xq[j] = gq[j] = 0.0f; evq[j] = -1;
        if (nT - 2 - j >= 0) fetch(j, nT - 2 - j);
    }
    // one step of the sweep from its inputs; `emit(gz of the step)` stores the external input's gradient
    // phase A on the sweep wave itself (the general loop): stage evaluations from the saved state
    auto phase_a_dpp = [&](const float h_, const float zk, const float x0, float (&X)[S], float (&hh)[S]) {
        const float cz = dot16(c0, zk, fz);
        float ks[S];
#pragma unroll
        for (int st = 0; st < S; ++st) {
            float acc = 0.0f;
#pragma unroll
            for (int jj = 0; jj < st; ++jj) acc += rk_a(METHOD, st, jj) * ks[jj];
            X[st] = st == 0 ? x0 : x0 + h_ * acc;
            hh[st] = elu(dot16(cz, X[st], fx));
            ks[st] = dot16(b2, hh[st], w2);
        }
    };
    auto sweep_step = [&](const float h_, const float zk, const float x0, const float g1, auto phase_a, auto emit) {
        float X[S], hh[S];
        phase_a(h_, zk, x0, X, hh);
        // ---- phase B: stages backwards
        float gks[S], gx0 = g1, D1 = 0.0f;
#pragma unroll
        for (int st = 0; st < S; ++st) gks[st] = (h_ * rk_b(METHOD, st)) * g1;
#pragma unroll
        for (int st = S - 1; st >= 0; --st) {
            const float gk = gks[st];
            SB2 += gk;
            rank1(aW2, hh[st], gk);                                // dW2[u][:] += gk[u] h[:]
            const float d1 = dot16(0.0f, gk, w2T) * dact(hh[st]);  // (W2^T gk)[u] ELU'
            D1 += d1;
            rank1(aFx, X[st], d1);                                 // dF_x[u][:] += d1[u] X_s[:]
            const float gx = dot16(0.0f, d1, fxT);
            gx0 += gx;
#pragma unroll
            for (int jj = 0; jj < st; ++jj) gks[jj] += (h_ * rk_a(METHOD, st, jj)) * gx;
        }
        S1 += D1;
        // ---- external block: frozen over the stages
        const float gzv = dot16(0.0f, D1, fzT);
        rank1(aFz, zk, D1);
        emit(gzv);
        gcarry = gx0;
    };
    // FAST main loop (round 5, as the forward kernel's): no event in the table, 32-bit counters and row offsets, uniform running row bases,
    // whole chunks of PF steps whose prefetches stay inside the grid; the general loop below takes over wherever this one stops.
    long long kc = nT - 2;
    if (nchunk > 0) {
        auto as_g = [](const float* q) { return (gptr<const float>)(uintptr_t)q; };
        const unsigned toff = (unsigned)(b * a.t.sb) * 4u, zoff = (unsigned)(b * a.z.sb + u) * 4u, roff = (unsigned)(b * LH + u) * 4u;
        const long long rstep = a.B * LH;
        int ki = (int)nT - 2;                                     // the step in hand; its slot is refilled with step ki - PF
        const float* trun = a.t.p + (long long)(ki - PF) * tst;
        const float* zrun = a.z.p + (long long)(ki - PF) * zst;
        const float* xrun = a.xs + (long long)(ki - PF) * rstep;
        const float* grun = a.gout + (long long)(ki - PF + 1) * rstep;
        float* gzrun = (a.gz ? a.gz : a.gx0) + (long long)(a.gz ? ki : 0) * rstep;      // (no gz wanted: a base that is never stored through)
        const bool st_gz = valid && a.gz != nullptr;
        const float* ring = pring + wv * kBwdRing + (row * 4) * kBwdVals * 16 + u;      // this lane's trajectory, step 0 of block slot 0
        __syncthreads();                                          // block 0 of the partner's rows is in the ring
        for (int bi = 0; bi < nchunk; ++bi, ki -= PF) {
            const float* rb = ring + (bi & 1) * 16 * kBwdVals * 16;
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                const float h_ = t_hi - tq[j];
                t_hi = tq[j];
                const float zk = zq_[j], x0 = xq[j], g1 = gcarry + gq[j];
                tq[j] = ldg<float>(as_g(trun), toff);
                zq_[j] = ldg<float>(as_g(zrun), zoff);
                xq[j] = ldg<float>(as_g(xrun), roff);
                const float gl = ldg<float>(as_g(grun), roff);
                gq[j] = valid ? gl : 0.0f;
                trun -= tst; zrun -= zst; xrun -= rstep; grun -= rstep;
                sweep_step(h_, zk, x0, g1,
                           [&](const float, const float, const float x0_, float (&X)[S], float (&hh)[S]) {
                               const float* rv = rb + j * kBwdVals * 16;
#pragma unroll
                               for (int st = 0; st < S; ++st) {
                                   hh[st] = rv[st * 16];
                                   X[st] = st == 0 ? x0_ : rv[(S + st - 1) * 16];
                               }
                           },
                           [&](const float gzv) {
                               if (st_gz) stg<float>((gptr<float>)(uintptr_t)gzrun, roff, gzv);
                           });
                if (st_gz || a.gz) gzrun -= rstep;
            }
            __syncthreads();                                      // the partner may overwrite this block's slot; the next block is complete
        }
        kc = ki;
    }
    for (; kc >= 0; kc -= PF) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const long long k = kc - j;
            if (k >= 0) {
                const float h_ = t_hi - tq[j];
                t_hi = tq[j];
                const float zk = zq_[j], x0 = xq[j], g1 = gcarry + gq[j];
                const int ev = evq[j];
                if (k - PF >= 0) fetch(j, k - PF);
                sweep_step(h_, zk, x0, g1, phase_a_dpp, [&](const float gzv) {
                    if (valid) {
                        if (ev >= 0) { if (a.gzj) a.gzj[(b * a.n_events + ev) * LH + u] = gzv; }
                        if (a.gz) a.gz[(k * a.B + b) * LH + u] = ev >= 0 ? 0.0f : gzv;
                    }
                });
            }
        }
    }

    // ---- epilogue
    if (valid) {
        a.gx0[b * LH + u] = gcarry + row_of(a.gout, 0);
        if (a.gz && nT >= 1) a.gz[((nT - 1) * a.B + b) * LH + u] = 0.0f;      // z[T-1] is never read
    }
    f4b cax = f4b{0.f, 0.f, 0.f, 0.f}, caz = cax;     // dWa: sum over the trajectories of sum_t(D1)[u] * a0[:]
    rank1(cax, a0x, S1);
    rank1(caz, a0z, S1);
    {   // d all_initial = (Wa - Wd)^T sum_t(D1): the lane's column of each a0 block
        float wt[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) wt[j] = a.de_w1[(long long)j * BK1 + u] - a.de_w1[(long long)j * BK1 + 2 * LH + u];
        const float gax = dot16(0.0f, S1, wt);
#pragma unroll
        for (int j = 0; j < 16; ++j) wt[j] = a.de_w1[(long long)j * BK1 + LH + u] - a.de_w1[(long long)j * BK1 + 3 * LH + u];
        const float gaz = dot16(0.0f, S1, wt);
        if (valid) { a.ga0[b * 2 * LH + u] = gax; a.ga0[b * 2 * LH + LH + u] = gaz; }
    }
    // ---- one partial per wave in nn.Linear order [W1 (16 x 96), b1, W2, b2]: the MFMA accumulators hold rows 4g .. 4g+3, column j of every block,
    //      summed over the wave's trajectories; the two bias sums take the xor-shuffles
    auto rows4 = [](float v) -> float {
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        return v;
    };
    S1 = rows4(S1);
    SB2 = rows4(SB2);
    {
        float* wp = a.wpart + ((size_t)blockIdx.x * 4 + wv) * BNP;
        const int g = lane >> 4, j = lane & 15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float* rr = wp + (4 * g + r) * BK1;
            rr[j] = cax[r];                       rr[LH + j] = caz[r];
            rr[2 * LH + j] = aFx[r] - cax[r];     rr[3 * LH + j] = aFz[r] - caz[r];
            rr[4 * LH + j] = aFx[r];              rr[5 * LH + j] = aFz[r];
            wp[LH * BK1 + LH + (4 * g + r) * LH + j] = aW2[r];
        }
        if (row == 0) {
            wp[LH * BK1 + u] = S1;
            wp[LH * BK1 + LH + LH * LH + u] = SB2;
        }
    }
}

template <bool ENC>
hipError_t launch_dpp(const LatentDppDev& a, hipStream_t s) {
    const dim3 grid((unsigned)((a.B + DTB - 1) / DTB)), block(ENC ? 512 : 256);
    switch (a.method) {
        case PSNODE_EULER: hipLaunchKernelGGL((latent_dpp_kernel<PSNODE_EULER, ENC>), grid, block, 0, s, a); break;
        case PSNODE_MIDPOINT: hipLaunchKernelGGL((latent_dpp_kernel<PSNODE_MIDPOINT, ENC>), grid, block, 0, s, a); break;
        default: hipLaunchKernelGGL((latent_dpp_kernel<PSNODE_RK4_38, ENC>), grid, block, 0, s, a); break;
    }
    return hipGetLastError();
}

bool mlp2_ok(const psnode_mlp_f32& m, int in, int out) {
    return m.n_layers == 2 && m.in_dim == in && m.out_dim[0] == LH && m.out_dim[1] == out && m.weight[0] && m.bias[0] &&
           m.weight[1] && m.bias[1];
}
Mlp2Dev bind2(const psnode_mlp_f32& m) { return Mlp2Dev{m.weight[0], m.bias[0], m.weight[1], m.bias[1], m.in_dim, m.out_dim[1]}; }

}  // namespace

// latent-only ODE (integrate_ODE with x_dim = z_dim = 16, DE = Linear(96,16) ELU Linear(16,16)): replaces K3a for the ODE
hipError_t launch_latent_dpp(const IntegrateDev& d, hipStream_t stream) {
    LatentDppDev a;
    memset(&a, 0, sizeof(a));
    a.method = d.method; a.xd = LH; a.zd = LH; a.T = d.T; a.B = d.B;
    a.de_w1 = d.de.w[0]; a.de_b1 = d.de.bias[0]; a.de_w2 = d.de.w[1]; a.de_b2 = d.de.bias[1];
    a.t = d.t; a.x = d.x; a.z = d.z; a.a0 = d.a0; a.ev = d.ev; a.zj = d.zj; a.zjb = d.zjb; a.zje = d.zje;
    a.xo = d.xo;
    return launch_dpp<false>(a, stream);
}

size_t latent_bwd_dpp_workspace_floats(long long B) { return (size_t)((B + DTB - 1) / DTB) * 4 * BNP + 64; }

int latent_bwd_dpp_launch(const psnode_ode_bwd_args_f32* p, float* workspace, hipStream_t s) {
    LatentBwdDppDev a;
    memset(&a, 0, sizeof(a));
    a.method = p->method; a.n_events = p->n_events; a.T = p->T; a.B = p->B;
    a.de_w1 = p->de.weight[0]; a.de_b1 = p->de.bias[0]; a.de_w2 = p->de.weight[1]; a.de_b2 = p->de.bias[1];
    a.t = ViewDev{p->t.ptr, p->t.stride_t, p->t.stride_b};
    a.z = ViewDev{p->z.ptr, p->z.stride_t, p->z.stride_b};
    a.a0 = p->all_initial; a.ev = p->event_idx; a.zj = p->z_jump; a.zjb = p->zj_stride_b; a.zje = p->zj_stride_e;
    a.xs = p->xs; a.gout = p->grad_xs; a.gx0 = p->grad_x0; a.gz = p->grad_z; a.gzj = p->grad_z_jump; a.ga0 = p->grad_all_initial;
    a.wpart = workspace;
    const int nwg = (int)((p->B + DTB - 1) / DTB);
    const dim3 grid((unsigned)nwg), block(512);      // 4 sweep waves + their 4 partner waves (phase A on MFMA tiles)
    switch (p->method) {
        case PSNODE_EULER: hipLaunchKernelGGL((latent_ode_backward_dpp_kernel<PSNODE_EULER>), grid, block, 0, s, a); break;
        case PSNODE_MIDPOINT: hipLaunchKernelGGL((latent_ode_backward_dpp_kernel<PSNODE_MIDPOINT>), grid, block, 0, s, a); break;
        default: hipLaunchKernelGGL((latent_ode_backward_dpp_kernel<PSNODE_RK4_38>), grid, block, 0, s, a); break;
    }
    if (hipGetLastError() != hipSuccess) return PSNODE_ERR_HIP;
    return launch_reduce_partials(workspace, p->grad_params, nullptr, BNP, 0, nwg * 4, s) == hipSuccess ? PSNODE_OK : PSNODE_ERR_HIP;
}

}  // namespace psnode

using namespace psnode;

extern "C" {

int32_t psnode_ode_encoded_supported(const psnode_ode_encoded_args_f32* p) {
    if (!p) return 0;
    if (p->x_dim < 1 || p->x_dim > LH || p->z_dim < 1 || p->z_dim > LH) return 0;
    return mlp2_ok(p->x_encoder, p->x_dim, LH) && mlp2_ok(p->z_encoder, p->z_dim, LH) && mlp2_ok(p->x_decoder, LH, p->x_dim) &&
           mlp2_ok(p->de, 6 * LH, LH);
}

int32_t psnode_ode_encoded_integrate_f32(const psnode_ode_encoded_args_f32* p, void* stream) {
    if (!p) return PSNODE_ERR_NULL;
    if (p->method < PSNODE_EULER || p->method > PSNODE_RK4_38) return PSNODE_ERR_METHOD;
    if (p->T < 1 || p->B < 1 || p->x_dim < 1 || p->z_dim < 1) return PSNODE_ERR_DIMS;
    if (!p->t.ptr || !p->x.ptr || !p->z.ptr || !p->x_pred) return PSNODE_ERR_NULL;
    if (p->event_idx && !p->z_jump) return PSNODE_ERR_NULL;
    if (!psnode_ode_encoded_supported(p)) return PSNODE_ERR_UNSUPPORTED;
    LatentDppDev a;
    memset(&a, 0, sizeof(a));
    a.method = p->method; a.xd = p->x_dim; a.zd = p->z_dim; a.T = p->T; a.B = p->B;
    a.xenc = bind2(p->x_encoder); a.zenc = bind2(p->z_encoder); a.xdec = bind2(p->x_decoder);
    a.de_w1 = p->de.weight[0]; a.de_b1 = p->de.bias[0]; a.de_w2 = p->de.weight[1]; a.de_b2 = p->de.bias[1];
    a.t = ViewDev{p->t.ptr, p->t.stride_t, p->t.stride_b};
    a.x = ViewDev{p->x.ptr, p->x.stride_t, p->x.stride_b};
    a.z = ViewDev{p->z.ptr, p->z.stride_t, p->z.stride_b};
    a.ev = p->event_idx; a.zj = p->z_jump; a.zjb = p->zj_stride_b; a.zje = p->zj_stride_e;
    a.xo = p->x_pred; a.xre = p->x_re; a.xre_st = p->xre_stride_t; a.xre_sb = p->xre_stride_b; a.xh_out = p->xh_out;
    return launch_dpp<true>(a, static_cast<hipStream_t>(stream)) == hipSuccess ? PSNODE_OK : PSNODE_ERR_HIP;
}

}  // extern "C"
