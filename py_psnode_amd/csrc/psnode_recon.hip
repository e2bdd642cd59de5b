// K3r -- the RECONSTRUCTION branch of the direct_encode ODE model at hidden 16 as one row kernel forward and one backward:
//   x_re = x_decoder(x_encoder(x))        neural_00_ODE_02_direct_encode.py:87        (x_encoder: in <= 16 -> 16 -> 16, x_decoder: 16 -> 16 -> out <= 16)
// and what loss.backward() does with it (:267-275).  The branch is a row-wise function of the DATA alone: with the integrator started from the
// first row's own encoding (integrate_ODE's x_init) the encoded rows Xh = x_encoder(x) have no other consumer, so they need not exist in
// memory.  Unfused, a training step moved them five times (encoder forward writes 262 MB, decoder forward reads, decoder backward reads them
// and writes their gradient, encoder backward reads that): K3b encoder forward + decoder forward + decoder backward + encoder backward =
// 84 + 78 + 202 + 172 us per 4096 x 1001 rows; here the forward reads x and writes x_re, the backward reads x and dL/dx_re -- nothing else.
//
// Plan (K3b's, psnode_rows.hip): one wave per 16-row tile; D rows of one layer are the B operands of the next (lane (g, j): units 4g .. 4g+3
// of row j), so the four layers chain in registers.  Backward: recompute h1, Xh, h2; the adjoint chain gh2 = W2d^T g, d3 = gh2 * ELU'(h2),
// gXh = W1d^T d3, gh1 = W2e^T gXh, d1 = gh1 * ELU'(h1) on MFMA with transposed weights; the four weight gradients contract over the tile's 16
// rows on MFMA (k-slot g <-> row 4g + m of the m-th MFMA): both operands in the TRANSPOSED tile layout (lane (g, i): P[i][row 4g + m]) --
// six 16 x 16 transposes through a wave-private padded LDS tile, the raw x and dL/dx_re rows read straight from memory in that layout.
// Gradients accumulate in registers over all of a wave's tiles; per-wave partials, summed in a fixed order (deterministic).
#include <string.h>

#include "psnode_common.h"

namespace psnode {
namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int RH = 16;                       // hidden = latent width of this kernel
constexpr int TPITCH = 20;                   // transpose tile [unit 16][row 16 + 4 pad] (floats): writes and 16-byte reads conflict-free
constexpr int kReconSlices = 32;

__device__ __forceinline__ f4 rmf(const float a, const float b, const f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f4 layer(const float (&w)[4], const f4 in, const f4 bias) {
    f4 pA = rmf(w[0], in[0], bias), pB = rmf(w[1], in[1], f4{0.f, 0.f, 0.f, 0.f});
    pA = rmf(w[2], in[2], pA); pB = rmf(w[3], in[3], pB);
    return pA + pB;
}

struct ReconDev {
    const float *w1e, *b1e, *w2e, *b2e, *w1d, *b1d, *w2d, *b2d;
    const float *in, *gout;
    float *out, *wpart;
    long long rows, in_stride, in_outer, out_stride, gout_stride;
    unsigned in_inner;
    int in_dim, out_dim;
};

// input offset of row r (flat, or the two-level addressing of psnode_rows.hip: the [B,T,D] batch read as time-major rows in place)
__device__ __forceinline__ long long in_off(const ReconDev& a, const long long r) {
    if (a.in_inner == 0) return r * a.in_stride;
    const unsigned o = (unsigned)r / a.in_inner;
    return (long long)o * a.in_outer + (long long)((unsigned)r - o * a.in_inner) * a.in_stride;
}

// forward weights of a lane: A operands (lane (g, i): M[i][k-slot columns of group g]) and D-layout biases
struct FwdW {
    float w1e[4], w2e[4], w1d[4];
    f4 b1e, b2e, b1d;
    int nm;
    __device__ __forceinline__ FwdW(const ReconDev& a, const int g, const int i) {
        nm = (a.in_dim + 3) >> 2;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int c = nm * g + m, k4 = 4 * g + m;
            w1e[m] = (m < nm && c < a.in_dim) ? a.w1e[i * a.in_dim + c] : 0.0f;
            w2e[m] = a.w2e[i * RH + k4];
            w1d[m] = a.w1d[i * RH + k4];
            b1e[m] = a.b1e[k4]; b2e[m] = a.b2e[k4]; b1d[m] = a.b1d[k4];
        }
    }
};

__global__ __launch_bounds__(256) void recon_fwd_kernel(const ReconDev a) {
    const int l = threadIdx.x & 63, g = l >> 4, j = l & 15;
    const FwdW W(a, g, j);
    float w2d[4];
    f4 b2d;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        w2d[m] = j < a.out_dim ? a.w2d[j * RH + 4 * g + m] : 0.0f;
        b2d[m] = 4 * g + m < a.out_dim ? a.b2d[4 * g + m] : 0.0f;
    }
    const long long tiles = (a.rows + 15) / 16;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    for (long long t = wave; t < tiles; t += nwaves) {
        const long long row = t * 16 + j;
        const bool valid = row < a.rows;
        const float* src = a.in + in_off(a, valid ? row : a.rows - 1) + W.nm * g;
        float v[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) v[m] = (m < W.nm && W.nm * g + m < a.in_dim) ? src[m] : 0.0f;
        f4 acc = W.b1e;
#pragma unroll
        for (int m = 0; m < 4; ++m) acc = rmf(W.w1e[m], v[m], acc);      // (m >= nm: zero weight x zero input -- no uniform branch between MFMAs)
        const f4 xh = layer(W.w2e, elu_quad(acc), W.b2e);
        const f4 y = layer(w2d, elu_quad(layer(W.w1d, xh, W.b1d)), b2d);
        if (valid) {
            float* dst = a.out + row * a.out_stride + 4 * g;
#pragma unroll
            for (int m = 0; m < 4; ++m) if (4 * g + m < a.out_dim) dst[m] = y[m];
        }
    }
}

// parameter layout of a partial vector: [W1e (16 x in), b1e, W2e (16 x 16), b2e | W1d (16 x 16), b1d, W2d (out x 16), b2d]  (nn.Linear order)
__host__ __device__ inline int recon_np_enc(int in_dim) { return RH * in_dim + RH + RH * RH + RH; }
__host__ __device__ inline int recon_np_dec(int out_dim) { return RH * RH + RH + out_dim * RH + out_dim; }

__global__ __launch_bounds__(256, 3) void recon_bwd_kernel(const ReconDev a) {
    __shared__ __attribute__((aligned(16))) float scr_all[4][2][16 * TPITCH];
    const int l = threadIdx.x & 63, wv = threadIdx.x >> 6, g = l >> 4, j = l & 15;
    float* scrA = scr_all[wv][0];
    float* scrB = scr_all[wv][1];
    const FwdW W(a, g, j);
    // transposed A operands of the adjoint chain: lane (g, i): M^T[i][4g + m] = M[4g + m][i]
    float w2dT[4], w1dT[4], w2eT[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int k4 = 4 * g + m;
        w2dT[m] = k4 < a.out_dim ? a.w2d[k4 * RH + j] : 0.0f;
        w1dT[m] = a.w1d[k4 * RH + j];
        w2eT[m] = a.w2e[k4 * RH + j];
    }
    const f4 z4 = f4{0.f, 0.f, 0.f, 0.f};
    f4 aW1e = z4, aW2e = z4, aW1d = z4, aW2d = z4, sb1e = z4, sb2e = z4, sb1d = z4, sb2d = z4;
    // 16 x 16 transpose through the wave's LDS tile: D layout (lane (g, j): P[4g + r][row j]) -> lane (g, i): P[i][row 4g + m], m = 0 .. 3
    auto transpose = [&](float* scr, const f4 p) -> f4 {
#pragma unroll
        for (int r = 0; r < 4; ++r) scr[(4 * g + r) * TPITCH + j] = p[r];
        return *reinterpret_cast<const f4*>(scr + j * TPITCH + 4 * g);      // (LDS serves a wave's accesses in order; no other wave touches this tile)
    };
    auto contract = [&](f4& acc, const f4 aT, const f4 bT) {          // acc[u][k] += sum over the tile's rows of A[u][row] B[k][row]
#pragma unroll
        for (int m = 0; m < 4; ++m) acc = rmf(aT[m], bT[m], acc);
    };
    const long long tiles = (a.rows + 15) / 16;
    const long long wave = (long long)blockIdx.x * 4 + wv, nwaves = (long long)gridDim.x * 4;
    // raw inputs of a tile, requested one tile ahead: x in the first layer's B layout, dL/dx_re in the D layout (lane (g, j): g[row j][4g + r]) and
    // transposed (lane (g, i = out dim): g[row 4g + m][i]), the raw rows transposed (lane (g, i = column): x[row 4g + m][i]); rows beyond the
    // set contribute zeros
    struct Raw { float v[4]; f4 gy, gyT; };
    auto request = [&](const long long t, Raw& q) {
        const long long row = t * 16 + j;
        const bool valid = row < a.rows;
        const float* src = a.in + in_off(a, valid ? row : a.rows - 1) + W.nm * g;
#pragma unroll
        for (int m = 0; m < 4; ++m) q.v[m] = (m < W.nm && W.nm * g + m < a.in_dim) ? src[m] : 0.0f;
        q.gy = z4; q.gyT = z4;
        if (valid) {
            const float* gs = a.gout + row * a.gout_stride + 4 * g;
#pragma unroll
            for (int r = 0; r < 4; ++r) if (4 * g + r < a.out_dim) q.gy[r] = gs[r];
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const long long rm = t * 16 + 4 * g + m;
            if (rm < a.rows && j < a.out_dim) q.gyT[m] = a.gout[rm * a.gout_stride + j];
        }
    };
    Raw nxt;
    if (wave < tiles) request(wave, nxt);
    for (long long t = wave; t < tiles; t += nwaves) {
        const Raw cur = nxt;
        if (t + nwaves < tiles) request(t + nwaves, nxt);
        const float (&v)[4] = cur.v;
        const f4 gy = cur.gy, gyT = cur.gyT;
        f4 xT = z4;                                   // (the rows this tile's first-layer loads just brought in: cache hits)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const long long rm = t * 16 + 4 * g + m;
            if (rm < a.rows && j < a.in_dim) xT[m] = a.in[in_off(a, rm) + j];
        }
        // ---- recompute
        f4 acc = W.b1e;
#pragma unroll
        for (int m = 0; m < 4; ++m) acc = rmf(W.w1e[m], v[m], acc);      // (m >= nm: zero weight x zero input -- no uniform branch between MFMAs)
        const f4 h1 = elu_quad(acc);
        const f4 xh = layer(W.w2e, h1, W.b2e);
        const f4 h2 = elu_quad(layer(W.w1d, xh, W.b1d));
        // ---- adjoint chain (rows beyond the set: gy = 0 -> every delta is 0)
        const f4 d3 = layer(w2dT, gy, z4) * elu_grad_quad(h2);
        const f4 gxh = layer(w1dT, d3, z4);
        const f4 d1 = layer(w2eT, gxh, z4) * elu_grad_quad(h1);
        sb2d += gy; sb1d += d3; sb2e += gxh; sb1e += d1;
        // ---- weight gradients: contraction over the tile's rows (activations of rows beyond the set meet zero deltas)
        contract(aW2d, gyT, transpose(scrA, h2));
        contract(aW1d, transpose(scrA, d3), transpose(scrB, xh));
        contract(aW2e, transpose(scrA, gxh), transpose(scrB, h1));
        contract(aW1e, transpose(scrA, d1), xT);
        // pin the accumulators in front of the loop's back edge: the next tile's predicated loads open with branches, and the taken edge of a
        // branch carries no wait states for an MFMA result (round 4's defect (b); isa_lint check B)
        asm volatile("" : "+v"(aW1e), "+v"(aW2e), "+v"(aW1d), "+v"(aW2d));
    }
    // ---- the wave's partial vector.  Weight accumulators: lane (g, j) holds dW[4g + r][j]; bias sums: lane (g, j) holds the tile-row-j share
    //      of db[4g + r] -- summed over j with four xor-shuffles (fixed order)
    auto over_rows = [](f4 v) -> f4 {
#pragma unroll
        for (int s = 1; s < 16; s <<= 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += __shfl_xor(v[r], s, 64);
        }
        return v;
    };
    sb1e = over_rows(sb1e); sb2e = over_rows(sb2e); sb1d = over_rows(sb1d); sb2d = over_rows(sb2d);
    const int npe = recon_np_enc(a.in_dim), npd = recon_np_dec(a.out_dim);
    float* wp = a.wpart + ((size_t)blockIdx.x * 4 + wv) * (npe + npd);
    float* pe = wp;
    float* pd = wp + npe;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int u = 4 * g + r;
        if (j < a.in_dim) pe[u * a.in_dim + j] = aW1e[r];
        pe[RH * a.in_dim + RH + u * RH + j] = aW2e[r];
        pd[u * RH + j] = aW1d[r];
        if (u < a.out_dim) pd[RH * RH + RH + u * RH + j] = aW2d[r];
        if (j == 0) {
            pe[RH * a.in_dim + u] = sb1e[r];
            pe[RH * a.in_dim + RH + RH * RH + u] = sb2e[r];
            pd[RH * RH + u] = sb1d[r];
            if (u < a.out_dim) pd[RH * RH + RH + a.out_dim * RH + u] = sb2d[r];
        }
    }
}

__global__ void recon_reduce_stage1(const float* __restrict__ part, float* __restrict__ mid, int np, int nparts) {
    const int pidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (pidx >= np) return;
    const int per = (nparts + kReconSlices - 1) / kReconSlices, q0 = blockIdx.y * per, q1 = q0 + per < nparts ? q0 + per : nparts;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int q = q0;
    for (; q + 8 <= q1; q += 8) {
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) acc[jj] += part[(size_t)(q + jj) * np + pidx];
    }
    for (int jj = 0; q < q1; ++q, ++jj) acc[jj] += part[(size_t)q * np + pidx];
    mid[(size_t)blockIdx.y * np + pidx] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
}
__global__ void recon_reduce_stage2(const float* __restrict__ mid, float* __restrict__ out, int np) {
    const int pidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (pidx >= np) return;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < kReconSlices; ++q) acc[q & 7] += mid[(size_t)q * np + pidx];
    out[pidx] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
}

long long recon_blocks(long long rows) {      // backward: three workgroups per CU (168 registers per lane)
    const long long tiles = (rows + 15) / 16;
    long long blocks = (tiles + 3) / 4;
    return blocks > 768 ? 768 : (blocks < 1 ? 1 : blocks);
}

bool recon_shape_ok(const psnode_mlp_f32* e, const psnode_mlp_f32* d) {
    return e && d && e->n_layers == 2 && d->n_layers == 2 && e->in_dim >= 1 && e->in_dim <= 16 && e->out_dim[0] == RH && e->out_dim[1] == RH &&
           d->in_dim == RH && d->out_dim[0] == RH && d->out_dim[1] >= 1 && d->out_dim[1] <= 16;
}
bool recon_ptrs_ok(const psnode_mlp_f32* e, const psnode_mlp_f32* d) {
    return e->weight[0] && e->bias[0] && e->weight[1] && e->bias[1] && d->weight[0] && d->bias[0] && d->weight[1] && d->bias[1];
}
ReconDev bind(const psnode_mlp_f32* e, const psnode_mlp_f32* d) {
    ReconDev a;
    memset(&a, 0, sizeof(a));
    a.w1e = e->weight[0]; a.b1e = e->bias[0]; a.w2e = e->weight[1]; a.b2e = e->bias[1];
    a.w1d = d->weight[0]; a.b1d = d->bias[0]; a.w2d = d->weight[1]; a.b2d = d->bias[1];
    a.in_dim = e->in_dim; a.out_dim = d->out_dim[1];
    return a;
}
int addressing_ok(int64_t rows, int64_t in_row_stride, int64_t in_inner_rows, int64_t in_outer_stride, int in_dim) {
    if (rows < 0 || in_row_stride < in_dim) return 0;
    if (in_inner_rows < 0 || in_inner_rows > 0xffffffffll || (in_inner_rows > 0 && (rows > 0xffffffffll || in_outer_stride < 0))) return 0;
    return 1;
}

}  // namespace
}  // namespace psnode

using namespace psnode;

extern "C" int32_t psnode_recon_rows_supported(const psnode_mlp_f32* enc, const psnode_mlp_f32* dec) { return recon_shape_ok(enc, dec) ? 1 : 0; }

extern "C" int32_t psnode_recon_rows_f32(const psnode_mlp_f32* enc, const psnode_mlp_f32* dec, int64_t rows, const float* in, int64_t in_row_stride,
                                         int64_t in_inner_rows, int64_t in_outer_stride, float* out, int64_t out_row_stride, void* stream) {
    if (!enc || !dec || !in || !out) return PSNODE_ERR_NULL;
    if (!recon_shape_ok(enc, dec)) return PSNODE_ERR_UNSUPPORTED;
    if (!recon_ptrs_ok(enc, dec)) return PSNODE_ERR_NULL;
    if (!addressing_ok(rows, in_row_stride, in_inner_rows, in_outer_stride, enc->in_dim) || out_row_stride < dec->out_dim[1]) return PSNODE_ERR_DIMS;
    if (rows == 0) return PSNODE_OK;
    ReconDev a = bind(enc, dec);
    a.in = in; a.out = out; a.rows = rows; a.in_stride = in_row_stride; a.in_inner = (unsigned)in_inner_rows;
    a.in_outer = in_inner_rows > 0 ? in_outer_stride : 0; a.out_stride = out_row_stride;
    const long long tiles = (rows + 15) / 16;
    long long blocks = (tiles + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(recon_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? PSNODE_OK : PSNODE_ERR_HIP;
}

extern "C" int64_t psnode_recon_rows_param_count(const psnode_mlp_f32* enc, const psnode_mlp_f32* dec) {
    return recon_shape_ok(enc, dec) ? recon_np_enc(enc->in_dim) + recon_np_dec(dec->out_dim[1]) : 0;
}
extern "C" size_t psnode_recon_rows_backward_workspace_bytes(const psnode_mlp_f32* enc, const psnode_mlp_f32* dec, int64_t rows) {
    if (!recon_shape_ok(enc, dec) || rows < 0) return 0;
    const size_t np = (size_t)recon_np_enc(enc->in_dim) + recon_np_dec(dec->out_dim[1]);
    return ((size_t)recon_blocks(rows) * 4 + kReconSlices) * np * sizeof(float);
}
extern "C" int32_t psnode_recon_rows_backward_f32(const psnode_mlp_f32* enc, const psnode_mlp_f32* dec, int64_t rows, const float* in,
                                                  int64_t in_row_stride, int64_t in_inner_rows, int64_t in_outer_stride, const float* grad_out,
                                                  int64_t gout_row_stride, float* grad_params, void* workspace, size_t workspace_bytes, void* stream) {
    if (!enc || !dec || !in || !grad_out || !grad_params) return PSNODE_ERR_NULL;
    if (!recon_shape_ok(enc, dec)) return PSNODE_ERR_UNSUPPORTED;
    if (!recon_ptrs_ok(enc, dec)) return PSNODE_ERR_NULL;
    if (!addressing_ok(rows, in_row_stride, in_inner_rows, in_outer_stride, enc->in_dim) || gout_row_stride < dec->out_dim[1]) return PSNODE_ERR_DIMS;
    const size_t need = psnode_recon_rows_backward_workspace_bytes(enc, dec, rows);
    if (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15)) return PSNODE_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    ReconDev a = bind(enc, dec);
    a.in = in; a.gout = grad_out; a.rows = rows; a.in_stride = in_row_stride; a.in_inner = (unsigned)in_inner_rows;
    a.in_outer = in_inner_rows > 0 ? in_outer_stride : 0; a.gout_stride = gout_row_stride;
    a.wpart = static_cast<float*>(workspace);
    const long long blocks = recon_blocks(rows);
    const int np = recon_np_enc(enc->in_dim) + recon_np_dec(dec->out_dim[1]);
    hipLaunchKernelGGL(recon_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
    if (hipGetLastError() != hipSuccess) return PSNODE_ERR_HIP;
    float* mid = static_cast<float*>(workspace) + (size_t)blocks * 4 * np;
    hipLaunchKernelGGL(recon_reduce_stage1, dim3((np + 255) / 256, kReconSlices), dim3(256), 0, s, static_cast<const float*>(workspace), mid, np,
                       (int)(blocks * 4));
    hipLaunchKernelGGL(recon_reduce_stage2, dim3((np + 255) / 256), dim3(256), 0, s, mid, grad_params, np);
    return hipGetLastError() == hipSuccess ? PSNODE_OK : PSNODE_ERR_HIP;
}
