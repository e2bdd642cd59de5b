// Packed register images of the MFMA kernels: shared by the forward kernels (psnode_mfma.hip) and the backward
// kernel (psnode_backward.hip).  pack[wave][reg][lane]; lane l: i = l&15 (A-operand row / B-operand column),
// g = l>>4 (k-slot of A and B operands, row group of D).
#pragma once
#include "psnode_common.h"

namespace psnode {

constexpr int HID = 64;        // hidden width of the backward kernel (the forward kernels take NWV = hidden / 16)
constexpr int NW = HID / 16;   // waves per workgroup at hidden 64
constexpr int TBM = 16;        // trajectories per workgroup
constexpr int kNXc = 2;        // x registers per lane: x_dim <= 4*kNXc = 8 (every kernel family)
constexpr int kNXw = 4;        // ... and 4 for x_dim 9..16 (x_dim is data-defined upstream, neural_00_ODE_01_no_encode.py:293): the ODE forward K1
constexpr int kMaxNZM = 4;     // per-step external MFMAs of the DE: 2*(z+v+i) <= 16

// Forward image of one MLP.
// DE: W1A = columns of the `s` block (x dims), W1B = columns of the `s-a0` block (x dims), W1E = NE ext registers.
//     The forward kernels use the FOLDED image (PackMfma::fold, NB = 0): W1A = Ws + Wd on the x dims and the a0 columns carry
//     Wa - Wd there, so a stage multiplies 2*NX fewer MFMAs (one extra rounding of the summed weights, 6e-8 relative, as K3c/K3f).
//     The backward kernels keep the unfolded image (they need Ws and Wd separately for the weight gradients).
// AE: W1A = columns of x, W1B unused (count 0), W1E = NE registers of the z|v columns.
template <int NX, int NB, int NE, int NWV = NW>
struct Regs {
    static constexpr int W1A = 0;
    static constexpr int W1B = NX;
    static constexpr int W1E = NX + NB;
    static constexpr int B1 = W1E + NE;
    static constexpr int W2 = B1 + 4;         // (4*NWV) chunk c = source wave (w+c) % NWV
    static constexpr int B2 = W2 + 4 * NWV;
    static constexpr int W3 = B2 + 4;
    static constexpr int B3 = W3 + 4 * NWV;
    static constexpr int W4 = B3 + 4;         // (4) this wave's K slice
    static constexpr int B4 = W4 + 4;
    static constexpr int COUNT = B4 + 4;      // followed by NA registers of the a0 columns of L1
};
__host__ __device__ constexpr int max_regs(int nw) { return 2 * kNXc + kMaxNZM + 20 + 8 * nw; }
constexpr int kMaxRegs = max_regs(NW);

// ext slot q -> index into ext = z | v | i, or -1 (padding)
__host__ __device__ inline int slot_ext(int q, int ne) { return q < ne ? q : (q < 2 * ne ? q - ne : -1); }

struct PackMfma {
    int ae;                 // 0: DE image, 1: AE image
    int nw;                 // waves per tile = hidden / 16
    int xd, ne, n, nzv;     // ne = z+v+i (DE ext), n = xd+ne, nzv = z+v
    int NX, NB, NE, NA;
    int fold;               // DE forward image only: W1A = Ws + Wd on the x dims, no W1B registers (NB = 0), a0 columns = Wa - Wd there
    int hreal;              // the MLP's hidden width (row stride of its H->H tensors); 0 = 16 * nw.  Widths between the kernels'
                            // 32 / 64 / 128 run zero-padded: units >= hreal have zero weights and biases, ELU(0) = 0, so they
                            // contribute exact zeros to every sum
    const float *w1, *b1, *w2, *b2, *w3, *b3, *w4, *b4;
    int out_dim;            // x_dim (DE) or i_dim (AE)
    float* out;
    int scaled = 0;         // forward image of the inference kernels (round 5): the hidden layers run in the log2(e)-SCALED domain --
                            // every L1 register and the biases b1, b2, b3 carry a factor log2e, the L4 slice a factor 1/log2e, the
                            // H->H matrices are untouched (their input g = log2e * ELU(p) and their output log2e * p carry the same
                            // factor) -- so that the ELU's exp2 argument IS the pre-activation (psnode_common.h: elu_quad_scaled)
};

// the hidden width the MFMA kernels run a width-h MLP at (0: none)
__host__ __device__ inline int padded_hidden(int h) { return h < 1 ? 0 : (h <= 32 ? 32 : (h <= 64 ? 64 : (h <= 128 ? 128 : 0))); }
// the width class a backward kernel runs the 4-layer MLP `in -> h -> h -> h -> out` at (zero-padded units beyond h); 0: another shape
inline int wide_hidden(const psnode_mlp_f32& m) {
    if (m.n_layers != 4) return 0;
    const int h = m.out_dim[0];
    if (m.out_dim[1] != h || m.out_dim[2] != h) return 0;
    return padded_hidden(h);
}
// ... and the FORWARD kernels K1 / K2 alone (round 4): 129..192 -> 12 waves, 193..256 -> 16 waves per tile, the H->H weights of these
// two classes streamed from the L2-resident image every layer (no CU holds them: two 256 x 256 fp32 matrices are the whole register file)
__host__ __device__ inline int padded_hidden_fwd(int h) { return h <= 128 ? padded_hidden(h) : (h <= 192 ? 192 : (h <= 256 ? 256 : 0)); }
// source wave of chunk c in wave w: (w + c) mod nw  (w, c < nw; nw need not be a power of two: 12 waves at hidden 192)
__host__ __device__ constexpr int wrap_wave(int x, int nw) { return (nw & (nw - 1)) == 0 ? (x & (nw - 1)) : (x >= nw ? x - nw : x); }

__host__ __device__ inline int pack_fwd_count(const PackMfma& p) { return p.NX + p.NB + p.NE + 20 + 8 * p.nw + p.NA; }

// value of forward-image register `reg` (0 .. pack_fwd_count) for wave w, lane `lane`
__device__ inline float pack_fwd_value(const PackMfma& p, int w, int reg, int lane) {
    const int H = p.hreal ? p.hreal : 16 * p.nw;      // real width: row stride and bound of the unit indices
    const int W1B = p.NX, W1E = p.NX + p.NB, B1 = W1E + p.NE, W2 = B1 + 4, B2 = W2 + 4 * p.nw, W3 = B2 + 4, B3 = W3 + 4 * p.nw,
              W4 = B3 + 4, B4 = W4 + 4, COUNT = B4 + 4;
    const int K1 = p.ae ? p.n + p.xd + p.nzv : 3 * p.n;
    const int i = lane & 15, g = lane >> 4, u = 16 * w + i;
    const bool urow = u < H;                          // padding units: every register of their rows is zero
    float v = 0.0f;
    if (reg < W1B) {                      // x columns: DE `s` block / AE x block
        const int d = 4 * reg + g;
        if (d < p.xd && urow) {
            v = p.w1[u * K1 + (p.ae ? p.n : 2 * p.n) + d];
            if (p.fold) v += p.w1[u * K1 + p.n + d];      // W.cat(a0, s-a0, s) = (Ws+Wd).s + (Wa-Wd).a0 on the x dims
        }
    } else if (reg < W1E) {               // DE `s - a0` block, x dims
        const int d = 4 * (reg - W1B) + g;
        if (d < p.xd && urow) v = p.w1[u * K1 + p.n + d];
    } else if (reg < B1) {                // external-input columns
        const int q = 4 * (reg - W1E) + g;
        if (!urow) {
        } else if (p.ae) {
            if (q < p.nzv) v = p.w1[u * K1 + p.n + p.xd + q];
        } else {
            if (q < p.ne) v = p.w1[u * K1 + p.n + p.xd + q];
            else if (q < 2 * p.ne) v = p.w1[u * K1 + 2 * p.n + p.xd + (q - p.ne)];
        }
    } else if (reg < W2) {
        const int uu = 16 * w + 4 * g + (reg - B1);
        if (uu < H) v = p.b1[uu];
    } else if (reg < B2) {
        const int kk = reg - W2, ws = wrap_wave(w + (kk >> 2), p.nw), col = 16 * ws + 4 * g + (kk & 3);
        if (urow && col < H) v = p.w2[u * H + col];
    } else if (reg < W3) {
        const int uu = 16 * w + 4 * g + (reg - B2);
        if (uu < H) v = p.b2[uu];
    } else if (reg < B3) {
        const int kk = reg - W3, ws = wrap_wave(w + (kk >> 2), p.nw), col = 16 * ws + 4 * g + (kk & 3);
        if (urow && col < H) v = p.w3[u * H + col];
    } else if (reg < W4) {
        const int uu = 16 * w + 4 * g + (reg - B3);
        if (uu < H) v = p.b3[uu];
    } else if (reg < COUNT) {
        // output row rho = 4*gr + rr (A operand: rho = i; bias in D layout: gr = g, rr = reg - B4)
        const bool bias = reg >= B4;
        const int gr = bias ? g : (i >> 2), rr = bias ? reg - B4 : (i & 3);
        int o;                            // which output the row carries, -1 = none
        if (p.ae) {                       // row (gr, m) <- the i-dim DE ext slot q = 4m+gr needs
            const int e = slot_ext(4 * rr + gr, p.ne);
            o = e >= p.nzv ? e - p.nzv : -1;
        } else {                          // row (gr, rr) <- x-dim 4*rr+gr
            o = 4 * rr + gr;
        }
        const int col = 16 * w + 4 * g + (reg - W4);
        if (o >= 0 && o < p.out_dim) v = bias ? p.b4[o] : (col < H ? p.w4[o * H + col] : 0.0f);
    } else {
        const int q = 4 * (reg - COUNT) + g;
        if (q < p.n && urow) {
            v = p.w1[u * K1 + q];
            if (p.fold && q < p.xd) v -= p.w1[u * K1 + p.n + q];
        }
    }
    if (p.scaled) {
        const bool mid = (reg >= W2 && reg < B2) || (reg >= W3 && reg < B3);   // H->H matrices: scale-free
        const bool out = reg >= W4 && reg < COUNT;                                // L4 slice (/ log2e) and b4 (unscaled output)
        if (out) { if (reg < B4) v = v / kLog2e; }
        else if (!mid) v = v * kLog2e;
    }
    return v;
}

}  // namespace psnode
