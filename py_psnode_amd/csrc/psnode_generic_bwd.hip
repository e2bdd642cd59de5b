// K5 -- generic fused backward pass (discretise-then-optimise) for ODE and DAE, any layer count / widths that fit LDS.
// The always-available HIP path for training, as psnode_generic.hip is for the forward: shapes with an MFMA backward
// (K4, psnode_backward.hip) use that one.
//
// One workgroup = 16 trajectories walked from the last grid point to the first, everything in LDS:
//   * activations of ONE MLP evaluation for all layers, [unit][TP] with a padded row stride TP = 20 floats: 16 lanes
//     reading 16 consecutive rows with ds_read_b128 then hit 16 disjoint bank quads (stride 16 would be 4..16-way);
//   * parameter-gradient accumulators for every weight and bias (each element owned by one thread: no atomics),
//     written once at the end as a per-workgroup partial and summed in a fixed order by reduce_partials (deterministic);
//   * the RK adjoint state (stage inputs, k_s, g_k[s], carries) and the external-input gradients.
// Per step: stage forwards (to rebuild the stage inputs), then for each stage in reverse a forward with stored
// activations followed by the VJP:  delta_in = W^T delta_out * ELU'(.) (weights read row-major, coalesced over the input
// index) and dW += delta (x) act in 4x4 register blocks.  DAE: the AE head's VJP is chained in through the algebraic
// variable (i_{k+1} = g(x_{k+1}; z,v) feeds the DE of step k+1; at event steps i0 = g(x_k; jumps) instead).
#include <string.h>

#include "psnode_common.h"

namespace psnode {
namespace {

constexpr int TB = 16;    // trajectories per workgroup
constexpr int TP = 20;    // padded row stride (floats)
constexpr int NT = 256;

struct GMlp {
    int L, in_dim;
    int out_dim[kMaxLayers];
    const float* w[kMaxLayers];    // row-major [out][in] (caller's nn.Linear weight)
    const float* wt[kMaxLayers];   // transposed [in][out] (workspace)
    const float* b[kMaxLayers];
    int gw[kMaxLayers], gb[kMaxLayers];   // offsets of dW, db in the flat gradient vector (nn.Linear order)
    int act[kMaxLayers + 1];              // row offsets of the layer activations (act[0] = input) in the acts buffer
    int np;
};

struct GBwd {
    int method, dae;
    int xd, zd, vd, id;
    long long T, B;
    GMlp de, ae;
    ViewDev t, z, v;
    const float* a0;
    const int* ev;
    const float* zj; long long zjb, zje;
    const float* vj; long long vjb, vje;
    int n_events;
    const float *xs, *is_, *gxs, *gis;
    float *gx0, *gz, *gv, *gzj, *gvj, *ga0, *wpart;
    int maxw, act_rows;
    int gacc_global;   // parameter-gradient accumulators in this workgroup's slice of wpart (global, L2) instead of LDS: 0 = none,
                       // 1 = both MLPs', 2 = the AE's only (register path of a DAE: the DE's tile-major accumulators stay in LDS)
    // register path of the DE (round 6): <= 4 layers of <= 64 units, 3 n <= 128 input columns.  Plain and transposed MFMA images (workspace;
    // psnode_generic.hip: launch_pack_plain_images); the wave's A operands of both stay in VGPRs for the launch.
    int de_reg;
    const float* fimg[kMaxLayers];
    const float* timg[kMaxLayers];
    // streamed path (round 6): the MLPs that are not on the register path -- str 1: the AE head of a DAE whose DE is, 2: both MLPs -- read
    // their MFMA A operands from the same kind of images (L2-resident), one chunk ahead; no staging through LDS
    int str;
    float* tmpart;     // per-workgroup TILE-MAJOR global accumulators of the MLPs off the staged path (gacc_global != 0): tm_total(de) + tm_total(ae)
    const float* fimgA[kMaxLayers];
    const float* timgA[kMaxLayers];
};

__device__ __forceinline__ float delu(float h) { return elu_grad(h); }   // ELU'(pre) from h = ELU(pre)

typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4v gm(float a, float b, f4v c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// All three matrix products below run on v_mfma_f32_16x16x4_f32 with operands read straight from LDS (lane l:
// i = j = l&15, k-slot g = l>>4).  Output tiles of 16 rows are dealt round-robin to the four waves.

// forward with stored activations: acts[act[0]] = input rows; writes acts[act[l+1]].
// out[u][traj] = sum_k W[u][k] in[k][traj]:  A[i][g] = W^T staged in `wbuf` as [k][N] (chunks of input rows, partial sums
// of multi-chunk layers live in `out`), B[g][j] = in[k = 4q+g][traj j].  Barrier after every chunk.
// (forceinline: as separate functions the buffers arrive as GENERIC pointers and every LDS access becomes a flat_load / flat_store)
__device__ __forceinline__ void g_forward(const GMlp& m, float* acts, float* wbuf) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, i = lane & 15;
    int K = m.in_dim;
    for (int l = 0; l < m.L; ++l) {
        const int N = m.out_dim[l];
        const float* __restrict__ wt = m.wt[l];
        const float* __restrict__ bias = m.b[l];
        const float* in = acts + m.act[l] * TP;
        float* out = acts + m.act[l + 1] * TP;
        const bool last = (l + 1 == m.L);
        const int KC = (kWBuf / N) & ~3;   // input rows per chunk: a multiple of 4, >= 4 because N <= PSNODE_MAX_WIDTH = kWBuf / 4
        for (int k0 = 0; k0 < K; k0 += KC) {
            const int kc = K - k0 < KC ? K - k0 : KC;
            stage_weights(wt + (size_t)k0 * N, wbuf, kc * N);
            const bool first = k0 == 0, final = k0 + kc >= K;
            for (int mt = wave; mt * 16 < N; mt += 4) {
                const int u = 16 * mt + i;
                f4v accA, accB = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int uu = 16 * mt + 4 * g + r, uc = uu < N ? uu : N - 1;
                    const float bv = bias[uc], ov = out[uc * TP + i];      // two address spaces: load both, select the value
                    accA[r] = uu < N ? (first ? bv : ov) : 0.0f;
                }
                for (int kq = 0; kq < kc; kq += 8) {
                    const int ka = kq + g, kb = kq + 4 + g;
                    // clamped addresses, unconditional loads, selects: a predicated LDS load is an exec-masked branch around every operand
                    const int kac = ka < kc ? ka : kc - 1, kbc = kb < kc ? kb : kc - 1, uc = u < N ? u : N - 1;
                    const float wa = wbuf[kac * N + uc], ia = in[(k0 + kac) * TP + i], wb = wbuf[kbc * N + uc], ib = in[(k0 + kbc) * TP + i];
                    const float a0 = (ka < kc && u < N) ? wa : 0.0f, b0 = ka < kc ? ia : 0.0f;
                    const float a1 = (kb < kc && u < N) ? wb : 0.0f, b1 = kb < kc ? ib : 0.0f;
                    accA = gm(a0, b0, accA);
                    accB = gm(a1, b1, accB);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int uu = 16 * mt + 4 * g + r;
                    if (uu < N) {
                        float v = accA[r] + accB[r];
                        if (final && !last) v = elu1(v);
                        out[uu * TP + i] = v;
                    }
                }
            }
            __syncthreads();
        }
        K = N;
    }
}

// VJP of the MLP: `din` holds delta of the output [N_L][TP]; returns the buffer with the input gradient [in_dim][TP].
// Accumulates dW, db into gacc.  Ends with a barrier.
// gacc_l / gacc_g: the parameter-gradient accumulators in LDS or in this workgroup's global slice (gg, a template parameter: a runtime
// choice between the two pointers makes every access a flat one)
template <bool gg>
__device__ __forceinline__ float* g_vjp(const GMlp& m, const float* acts, float* din, float* dout, float* gacc_l, float* gacc_g, float* wbuf) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i = lane & 15;
    for (int l = m.L - 1; l >= 0; --l) {
        const int N = m.out_dim[l], K = l == 0 ? m.in_dim : m.out_dim[l - 1];
        const float* a_in = acts + m.act[l] * TP;
        // ---- dW[j][k] += sum_tr delta[j][tr] * a_in[k][tr]: 16x16 tiles, contraction over the 16 trajectories
        //      A[i][g] = delta[16 mt + i][tr = 4q+g], B[g][j] = a_in[16 kt + j][tr]; each tile is owned by one wave
        float* gw_l = gacc_l + m.gw[l];
        float* gw_g = gacc_g + m.gw[l];
        const int ntk = (K + 15) / 16, ntiles = ((N + 15) / 16) * ntk;
        for (int tile = wave; tile < ntiles; tile += 4) {
            const int mt = tile / ntk, kt = tile % ntk;
            const int ju = 16 * mt + i, ku = 16 * kt + i;
            f4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int tr = 4 * q + g;
                const float dv = din[(ju < N ? ju : N - 1) * TP + tr], av = a_in[(ku < K ? ku : K - 1) * TP + tr];
                acc = gm(ju < N ? dv : 0.0f, ku < K ? av : 0.0f, acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int jr = 16 * mt + 4 * g + r;
                if (jr < N && ku < K) { if constexpr (gg) gw_g[jr * K + ku] += acc[r]; else gw_l[jr * K + ku] += acc[r]; }
            }
        }
        // ---- db[j] += sum_tr delta[j][tr]
        for (int j = tid; j < N; j += NT) {
            float s = 0.0f;
#pragma unroll
            for (int c = 0; c < TB; ++c) s += din[j * TP + c];
            if constexpr (gg) gacc_g[m.gb[l] + j] += s; else gacc_l[m.gb[l] + j] += s;
        }
        // ---- delta_in[k] = sum_j W[j][k] delta[j]  (* ELU'(a_in[k]) for hidden layers): A[i][g] = W[j = 4q+g][16 kt + i] from the
        //      row-major weights staged in chunks of output rows, B[g][j] = delta[4q+g][traj]; partial sums in dout
        __syncthreads();
        const float* __restrict__ w = m.w[l];
        const int JC = (kWBuf / K) & ~3;
        for (int j0 = 0; j0 < N; j0 += JC) {
            const int jc = N - j0 < JC ? N - j0 : JC;
            stage_weights(w + (size_t)j0 * K, wbuf, jc * K);
            const bool first = j0 == 0, final = j0 + jc >= N;
            for (int kt = wave; kt * 16 < K; kt += 4) {
                const int ku = 16 * kt + i;
                f4v accA, accB = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kr = 16 * kt + 4 * g + r;
                    const float dv = dout[(kr < K ? kr : K - 1) * TP + i];
                    accA[r] = (!first && kr < K) ? dv : 0.0f;
                }
                for (int jq = 0; jq < jc; jq += 8) {
                    const int ja = jq + g, jb = jq + 4 + g;
                    const int jac = ja < jc ? ja : jc - 1, jbc = jb < jc ? jb : jc - 1, kc_ = ku < K ? ku : K - 1;
                    const float wa = wbuf[jac * K + kc_], da = din[(j0 + jac) * TP + i], wb = wbuf[jbc * K + kc_], db_ = din[(j0 + jbc) * TP + i];
                    const float a0 = (ja < jc && ku < K) ? wa : 0.0f, b0 = ja < jc ? da : 0.0f;
                    const float a1 = (jb < jc && ku < K) ? wb : 0.0f, b1 = jb < jc ? db_ : 0.0f;
                    accA = gm(a0, b0, accA);
                    accB = gm(a1, b1, accB);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kr = 16 * kt + 4 * g + r;
                    if (kr < K) {
                        float v = accA[r] + accB[r];
                        if (final && l > 0) v *= delu(a_in[kr * TP + i]);
                        dout[kr * TP + i] = v;
                    }
                }
            }
            __syncthreads();
        }
        float* tmp = din; din = dout; dout = tmp;
    }
    return din;
}

// ---- register path of the DE (the forward recomputation and the delta propagation of g_vjp): what K0's register form is for the forward
// pass (psnode_generic.hip).  Activations and deltas additionally live in QUAD-ROW buffers (float index ((col / 4) * 16 + traj) * 4 + col % 4:
// the B operands of four MFMA steps are one lane-linear ds_read_b128, a D tile one ds_write_b128); the [unit][TP] copies stay, they are what
// the weight-gradient MFMAs (contraction over the trajectories) and the step's glue read.
typedef float f4 __attribute__((ext_vector_type(4)));
__host__ __device__ constexpr int up16(int v) { return (v + 15) & ~15; }
__device__ __forceinline__ int qi(int r, int c) { return ((((r >> 2) * TB) + c) << 2) | (r & 3); }
__device__ __forceinline__ void mfma_quad(const f4 av, const f4 bv, f4& acc) {
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[e], acc, 0, 0, 0);
}
template <int Q, int QM>
__device__ __forceinline__ f4 tile_reg(const f4* bq, const f4 (&wa)[QM]) {
    f4 bv[Q];
#pragma unroll
    for (int c = 0; c < Q; ++c) bv[c] = bq[c * 64];
    __builtin_amdgcn_sched_barrier(0);
    f4 acc = f4{0.f, 0.f, 0.f, 0.f}, acc2 = acc;
#pragma unroll
    for (int c = 0; c < Q; ++c) mfma_quad(wa[c], bv[c], (c & 1) ? acc2 : acc);
    return Q > 1 ? acc + acc2 : acc;
}
template <int QM>
__device__ __forceinline__ f4 tile_reg_any(int S4, const f4* bq, const f4 (&wa)[QM]) {
    if constexpr (QM > 4) {
        switch (S4) {
            case 5: return tile_reg<5, QM>(bq, wa);
            case 6: return tile_reg<6, QM>(bq, wa);
            case 7: return tile_reg<7, QM>(bq, wa);
            case 8: return tile_reg<8, QM>(bq, wa);
            default: break;
        }
    }
    switch (S4) {
        case 1: return tile_reg<1, QM>(bq, wa);
        case 2: return tile_reg<2, QM>(bq, wa);
        case 3: return tile_reg<3, QM>(bq, wa);
        default: return tile_reg<4, QM>(bq, wa);
    }
}

struct RegFwd { f4 first[8]; f4 rest[3][4]; };          // layer 0: <= 128 input columns; layers 1..3: <= 64
struct RegBwd { f4 first[2][4]; f4 rest[3][4]; };       // transposed: layer 0 has <= 8 tiles over its inputs (two per wave), the others <= 4

__device__ __forceinline__ void load_reg_images(const GBwd& a, RegFwd& fw, RegBwd& bw) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const f4 zero = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        const bool on = l < a.de.L;
        const int K = on ? (l ? a.de.out_dim[l - 1] : a.de.in_dim) : 0, N = on ? a.de.out_dim[l] : 0;
        const int SK = (K + 15) >> 4, SN = (N + 15) >> 4;           // quads of the forward contraction / of the transposed one
        const f4* __restrict__ F = reinterpret_cast<const f4*>(a.fimg[on ? l : 0]) + lane;
        const f4* __restrict__ Tm = reinterpret_cast<const f4*>(a.timg[on ? l : 0]) + lane;
#pragma unroll
        for (int q = 0; q < (l ? 4 : 8); ++q) {
            const f4 v = (w < SN && q < SK) ? F[((size_t)(w < SN ? w : 0) * SK + (q < SK ? q : 0)) * 64] : zero;
            if (l == 0) fw.first[q] = v; else fw.rest[l - 1][q] = v;
        }
#pragma unroll
        for (int j = 0; j < (l ? 1 : 2); ++j) {
            const int kt = w + 4 * j;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f4 v = (kt < SK && q < SN) ? Tm[((size_t)(kt < SK ? kt : 0) * SN + (q < SN ? q : 0)) * 64] : zero;
                if (l == 0) bw.first[j][q] = v; else bw.rest[l - 1][q] = v;
            }
        }
    }
}

// quad-row buffers of the register path (float offsets from qb): the DE input, the hidden activations, two delta buffers
struct QOff { int in, act[kMaxLayers - 1], d0, d1, total; };
__host__ __device__ inline QOff q_offsets(const GMlp& m) {
    QOff q;
    int o = 0;
    q.in = o; o += up16(m.in_dim) * TB;
    int mx = 16;
    for (int l = 0; l < kMaxLayers - 1; ++l) {
        q.act[l] = o;
        if (l + 1 < m.L) o += up16(m.out_dim[l]) * TB;
    }
    for (int l = 0; l < m.L; ++l) mx = up16(m.out_dim[l]) > mx ? up16(m.out_dim[l]) : mx;
    q.d0 = o; o += mx * TB;
    q.d1 = o; o += mx * TB;
    q.total = o;
    return q;
}

// Register path with the accumulators in LDS: the DE's weight gradients are kept TILE-MAJOR -- [tile = mt * ntk + kt][lane][4], the MFMA D layout:
// one ds_read_b128 + one ds_write_b128 per tile instead of four predicated b32 read-modify-writes (13 of 77 ms at x_dim 20, hidden 64) -- and
// un-permuted into nn.Linear order once, when the workgroup's partial is written out.  Per layer: 16 x 16 tiles padded, then all the biases.
__host__ __device__ inline int tm_dw_off(const GMlp& m, int l) {
    int o = 0, k = m.in_dim;
    for (int q = 0; q < l; ++q) { o += up16(m.out_dim[q]) * up16(k); k = m.out_dim[q]; }
    return o;
}
__host__ __device__ inline int tm_db_off(const GMlp& m, int l) {
    int o = tm_dw_off(m, m.L);
    for (int q = 0; q < l; ++q) o += m.out_dim[q];
    return o;
}
__host__ __device__ inline int tm_total(const GMlp& m) { return (tm_db_off(m, m.L) + 3) & ~3; }

// forward with stored activations, the DE in registers: acts[act[0]] = input rows; writes acts[act[l + 1]] and the quad-row copies
__device__ __forceinline__ void g_forward_reg(const GBwd& a, float* acts, float* qb, const QOff& qo, const RegFwd& fw) {
    const GMlp& m = a.de;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, j = lane & 15;
    {   // the input rows -> quad-row, pad columns zero
        const float* u = acts + m.act[0] * TP;
        for (int idx = tid; idx < up16(m.in_dim) * TB; idx += NT)
            qb[qo.in + qi(idx / TB, idx % TB)] = idx / TB < m.in_dim ? u[(idx / TB) * TP + idx % TB] : 0.0f;
        __syncthreads();
    }
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        if (l >= m.L) break;
        const int K = l ? m.out_dim[l - 1] : m.in_dim, N = m.out_dim[l];
        const int S4 = (K + 15) >> 4, NTL = (N + 15) >> 4;
        const bool last = (l + 1 == m.L);
        if (w < NTL) {
            const f4* bq = reinterpret_cast<const f4*>(qb + (l ? qo.act[l - 1 < 3 ? l - 1 : 0] : qo.in)) + lane;
            const f4 bias = *reinterpret_cast<const f4*>(a.fimg[l] + (size_t)NTL * S4 * 256 + 16 * w + 4 * g);
            f4 acc;
            if (l == 0) acc = tile_reg_any<8>(S4, bq, fw.first);
            else acc = tile_reg_any<4>(S4, bq, fw.rest[l - 1 < 3 ? l - 1 : 0]);
            acc = acc + bias;
            const f4 e = last ? acc : elu_quad(acc);
            float* out = acts + m.act[l + 1] * TP;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int uu = 16 * w + 4 * g + r;
                if (uu < N) out[uu * TP + j] = e[r];
            }
            if (!last) reinterpret_cast<f4*>(qb + qo.act[l < 3 ? l : 0])[w * 64 + lane] = e;
        }
        __syncthreads();
    }
}

// VJP of the DE with the delta propagation in registers (the weight-gradient part is g_vjp's)
template <bool gg>
__device__ __forceinline__ float* g_vjp_reg(const GBwd& a, const float* acts, float* din, float* dout, float* gacc_l, float* gacc_g, float* qb,
                                            const QOff& qo, const RegBwd& bw) {
    const GMlp& m = a.de;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i = lane & 15;
    const int w = __builtin_amdgcn_readfirstlane(wave);
    int qd = qo.d0, qn = qo.d1;
    {   // the output gradient -> quad-row, pad columns zero
        const int N = m.out_dim[m.L - 1];
        for (int idx = tid; idx < up16(N) * TB; idx += NT) qb[qd + qi(idx / TB, idx % TB)] = idx / TB < N ? din[(idx / TB) * TP + idx % TB] : 0.0f;
        __syncthreads();
    }
#pragma unroll
    for (int l = 3; l >= 0; --l) {
        if (l >= m.L) continue;
        const int N = m.out_dim[l], K = l == 0 ? m.in_dim : m.out_dim[l - 1];
        const float* a_in = acts + m.act[l] * TP;
        // ---- dW[j][k] += sum_tr delta[j][tr] * a_in[k][tr],  db[j] += sum_tr delta[j][tr]      (as g_vjp)
        const int ntk = (K + 15) / 16, ntiles = ((N + 15) / 16) * ntk;
#ifndef PSNODE_K5_ABL
#define PSNODE_K5_ABL 0      // timing-only builds: 1 = no accumulation into gacc, 2 = no weight-gradient tiles at all, 3 = no forward recomputation
#endif
        // weight-gradient tiles.  Global accumulators (gg) are tile-major like the LDS ones -- one 16-byte read-modify-write per lane and tile
        // -- and the NEXT tile's old value is requested before this tile's MFMAs: four predicated b32 read-modify-writes per tile with the
        // L2 round trip exposed cost 56 of 95 ms at hidden 128.
        f4* T4 = reinterpret_cast<f4*>(gg ? gacc_g + tm_dw_off(m, l) : gacc_l + tm_dw_off(m, l)) + lane;
        f4 oldn = f4{0.f, 0.f, 0.f, 0.f};
        if constexpr (gg) { if (wave < ntiles) oldn = T4[wave * 64]; }
        for (int tile = wave; tile < (PSNODE_K5_ABL == 2 ? 0 : ntiles); tile += 4) {
            const int mt = tile / ntk, kt = tile % ntk;
            const int ju = 16 * mt + i, ku = 16 * kt + i;
            // MFMA step q contracts the trajectories 4 g + q (slot g): a lane's four operands are ONE 16-byte read of its row (TP = 20 floats:
            // 80-byte rows, 16-byte aligned)
            const f4 dv = *reinterpret_cast<const f4*>(din + (ju < N ? ju : N - 1) * TP + 4 * g);
            const f4 av = *reinterpret_cast<const f4*>(a_in + (ku < K ? ku : K - 1) * TP + 4 * g);
            f4 old = oldn;
            if constexpr (gg) oldn = T4[(tile + 4 < ntiles ? tile + 4 : tile) * 64];
            else old = T4[tile * 64];
            f4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = gm(ju < N ? dv[q] : 0.0f, ku < K ? av[q] : 0.0f, acc);
            if (PSNODE_K5_ABL == 1) { if (acc[0] == 123.456f) T4[0] = acc; continue; }
            T4[tile * 64] = old + acc;                   // rows / columns beyond the matrix accumulate zeros
        }
        for (int jj = tid; jj < N; jj += NT) {
            float s = 0.0f;
#pragma unroll
            for (int c = 0; c < TB; ++c) s += din[jj * TP + c];
            if constexpr (gg) gacc_g[tm_db_off(m, l) + jj] += s; else gacc_l[tm_db_off(m, l) + jj] += s;
        }
        // ---- delta_in[k] = sum_j W[j][k] delta[j]  (* ELU'(a_in[k]) for hidden layers): tiles over k, A operands (W^T) in registers
        const int SN = (N + 15) >> 4, NTK = (K + 15) >> 4;
        const f4* bq = reinterpret_cast<const f4*>(qb + qd) + lane;
#pragma unroll
        for (int jt = 0; jt < (l ? 1 : 2); ++jt) {
            const int kt = w + 4 * jt;
            if (kt < NTK) {
                f4 acc = l == 0 ? tile_reg_any<4>(SN, bq, bw.first[jt]) : tile_reg_any<4>(SN, bq, bw.rest[l > 0 ? l - 1 : 0]);
                // layer 0 stores the accumulator as it is: on the taken edge of the switch's exit branch the compiler's hazard count is one
                // wait state short of the MFMA's write (ISA lint check B); the tied nop puts the distance on every path
                asm volatile("s_nop 3" : "+v"(acc));
                if (l > 0) {
                    const f4 h = reinterpret_cast<const f4*>(qb + qo.act[l - 1 >= 0 ? l - 1 : 0])[kt * 64 + lane];
                    acc = acc * elu_grad_quad(h);
                    reinterpret_cast<f4*>(qb + qn)[kt * 64 + lane] = acc;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kr = 16 * kt + 4 * g + r;
                    if (kr < K) dout[kr * TP + i] = acc[r];
                }
            }
        }
        __syncthreads();
        { float* tmp = din; din = dout; dout = tmp; }
        { const int t_ = qd; qd = qn; qn = t_; }
    }
    return din;
}

// ---- streamed path: one output tile with its A operands read from the image (L2), one chunk of four quads ahead of the MFMAs that use
// them; B operands from the quad-row buffer.  (The loads are unconditional on clamped addresses: see psnode_generic.hip, mlp_eval.)
__device__ __forceinline__ f4 tile_stream(const f4* __restrict__ A, const int S4, const f4* bq) {
    f4 acc = f4{0.f, 0.f, 0.f, 0.f}, acc2 = acc;
    f4 nxt[4];
    auto fetch4 = [&](int q0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) nxt[c] = A[(q0 + c < S4 ? q0 + c : S4 - 1) * 64];
    };
    fetch4(0);
    for (int q0 = 0; q0 < S4; q0 += 4) {
        f4 cur[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) cur[c] = nxt[c];
        fetch4(q0 + 4 < S4 ? q0 + 4 : S4 - 1);
        if (q0 + 4 <= S4) {
            f4 bv[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) bv[c] = bq[(q0 + c) * 64];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < 4; ++c) mfma_quad(cur[c], bv[c], (c & 1) ? acc2 : acc);
        } else {
            for (int c = 0; q0 + c < S4; ++c) mfma_quad(c == 0 ? cur[0] : (c == 1 ? cur[1] : cur[2]), bq[(q0 + c) * 64], acc);
        }
    }
    return acc + acc2;
}

// forward with stored activations, streamed: acts[act[0]] = input rows; writes acts[act[l + 1]] and the quad-row copies
__device__ __forceinline__ void g_forward_str(const GMlp& m, const float* const* fimg, float* acts, float* qb, const QOff& qo) {
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4, j = lane & 15;
    {
        const float* u = acts + m.act[0] * TP;
        for (int idx = tid; idx < up16(m.in_dim) * TB; idx += NT)
            qb[qo.in + qi(idx / TB, idx % TB)] = idx / TB < m.in_dim ? u[(idx / TB) * TP + idx % TB] : 0.0f;
        __syncthreads();
    }
    for (int l = 0; l < m.L; ++l) {
        const int K = l ? m.out_dim[l - 1] : m.in_dim, N = m.out_dim[l];
        const int S4 = (K + 15) >> 4, NTL = (N + 15) >> 4;
        const bool last = (l + 1 == m.L);
        const f4* bq = reinterpret_cast<const f4*>(qb + (l ? qo.act[l - 1] : qo.in)) + lane;
        float* out = acts + m.act[l + 1] * TP;
        for (int nt = w; nt < NTL; nt += 4) {
            const f4 bias = *reinterpret_cast<const f4*>(fimg[l] + (size_t)NTL * S4 * 256 + 16 * nt + 4 * g);
            f4 acc = tile_stream(reinterpret_cast<const f4*>(fimg[l]) + (size_t)nt * S4 * 64 + lane, S4, bq) + bias;
            const f4 e = last ? acc : elu_quad(acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int uu = 16 * nt + 4 * g + r;
                if (uu < N) out[uu * TP + j] = e[r];
            }
            if (!last) reinterpret_cast<f4*>(qb + qo.act[l])[nt * 64 + lane] = e;
        }
        __syncthreads();
    }
}

// VJP, streamed: weight gradients as on the register path (tile-major LDS accumulators at gacc_l when !gg), delta propagation on the
// transposed images
template <bool gg>
__device__ __forceinline__ float* g_vjp_str(const GMlp& m, const float* const* timg, const float* acts, float* din, float* dout, float* gacc_l,
                                            float* gacc_g, float* qb, const QOff& qo) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i = lane & 15;
    const int w = __builtin_amdgcn_readfirstlane(wave);
    int qd = qo.d0, qn = qo.d1;
    {
        const int N = m.out_dim[m.L - 1];
        for (int idx = tid; idx < up16(N) * TB; idx += NT) qb[qd + qi(idx / TB, idx % TB)] = idx / TB < N ? din[(idx / TB) * TP + idx % TB] : 0.0f;
        __syncthreads();
    }
    for (int l = m.L - 1; l >= 0; --l) {
        const int N = m.out_dim[l], K = l == 0 ? m.in_dim : m.out_dim[l - 1];
        const float* a_in = acts + m.act[l] * TP;
        const int ntk = (K + 15) / 16, ntiles = ((N + 15) / 16) * ntk;
        // weight-gradient tiles.  Global accumulators (gg) are tile-major like the LDS ones -- one 16-byte read-modify-write per lane and tile
        // -- and the NEXT tile's old value is requested before this tile's MFMAs: four predicated b32 read-modify-writes per tile with the
        // L2 round trip exposed cost 56 of 95 ms at hidden 128.
        f4* T4 = reinterpret_cast<f4*>(gg ? gacc_g + tm_dw_off(m, l) : gacc_l + tm_dw_off(m, l)) + lane;
        f4 oldn = f4{0.f, 0.f, 0.f, 0.f};
        if constexpr (gg) { if (wave < ntiles) oldn = T4[wave * 64]; }
        for (int tile = wave; tile < (PSNODE_K5_ABL == 2 ? 0 : ntiles); tile += 4) {
            const int mt = tile / ntk, kt = tile % ntk;
            const int ju = 16 * mt + i, ku = 16 * kt + i;
            // MFMA step q contracts the trajectories 4 g + q (slot g): a lane's four operands are ONE 16-byte read of its row (TP = 20 floats:
            // 80-byte rows, 16-byte aligned)
            const f4 dv = *reinterpret_cast<const f4*>(din + (ju < N ? ju : N - 1) * TP + 4 * g);
            const f4 av = *reinterpret_cast<const f4*>(a_in + (ku < K ? ku : K - 1) * TP + 4 * g);
            f4 old = oldn;
            if constexpr (gg) oldn = T4[(tile + 4 < ntiles ? tile + 4 : tile) * 64];
            else old = T4[tile * 64];
            f4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = gm(ju < N ? dv[q] : 0.0f, ku < K ? av[q] : 0.0f, acc);
            if (PSNODE_K5_ABL == 1) { if (acc[0] == 123.456f) T4[0] = acc; continue; }
            T4[tile * 64] = old + acc;                   // rows / columns beyond the matrix accumulate zeros
        }
        for (int jj = tid; jj < N; jj += NT) {
            float s = 0.0f;
#pragma unroll
            for (int c = 0; c < TB; ++c) s += din[jj * TP + c];
            if constexpr (gg) gacc_g[tm_db_off(m, l) + jj] += s; else gacc_l[tm_db_off(m, l) + jj] += s;
        }
        const int SN = (N + 15) >> 4, NTK = (K + 15) >> 4;
        const f4* bq = reinterpret_cast<const f4*>(qb + qd) + lane;
        for (int kt = w; kt < NTK; kt += 4) {
            f4 acc = tile_stream(reinterpret_cast<const f4*>(timg[l]) + (size_t)kt * SN * 64 + lane, SN, bq);
            asm volatile("s_nop 3" : "+v"(acc));       // (as on the register path: the store below may sit on a taken branch edge)
            if (l > 0) {
                const f4 h = reinterpret_cast<const f4*>(qb + qo.act[l - 1])[kt * 64 + lane];
                acc = acc * elu_grad_quad(h);
                reinterpret_cast<f4*>(qb + qn)[kt * 64 + lane] = acc;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kr = 16 * kt + 4 * g + r;
                if (kr < K) dout[kr * TP + i] = acc[r];
            }
        }
        __syncthreads();
        { float* tmp = din; din = dout; dout = tmp; }
        { const int t_ = qd; qd = qn; qn = t_; }
    }
    return din;
}

// gg / ggA: the DE's / the AE's accumulators live in the workgroup's global slice.  REG: the DE on the register path.  STR: 1 = the AE
// head streamed, 2 = both MLPs streamed (0: whatever is not on the register path stages its weights through LDS).
template <bool gg, bool REG, bool ggA = gg, int STR = 0>
__global__ __launch_bounds__(NT) void generic_backward_kernel(const GBwd a) {
    constexpr bool DE_TM = REG || STR == 2;      // the DE's LDS accumulators are tile-major
    constexpr bool AE_TM = STR >= 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const long long b0 = (long long)blockIdx.x * TB;
    const bool dae = a.dae != 0;
    const int xd = a.xd, zd = a.zd, vd = dae ? a.vd : 0, id = dae ? a.id : 0;
    const int nzv = zd + vd, ne = nzv + id, n = xd + ne;
    const int S = rk_stages(a.method);
    const int nx = xd * TP;

    float* acts = lds;                            // [act_rows][TP]
    float* dA = acts + a.act_rows * TP;           // [maxw][TP]
    float* dB = dA + a.maxw * TP;
    float* a0s = dB + a.maxw * TP;                // [n][TP]
    float* ga0s = a0s + n * TP;                   // [n][TP]
    float* ext = ga0s + n * TP;                   // [ne][TP]  z | v | i fed to the DE of this step
    float* gext = ext + ne * TP;                  // [ne][TP]
    float* x0 = gext + ne * TP;                   // [xd][TP]
    float* xst = x0 + nx;                         // [4][xd][TP]
    float* ks = xst + 4 * nx;                     // [4][xd][TP]
    float* gks = ks + 4 * nx;                     // [4][xd][TP]
    float* gx0 = gks + 4 * nx;                    // [xd][TP]
    float* gxc = gx0 + nx;                        // [xd][TP]  carried dL/dx_{k+1}
    float* gic = gxc + nx;                        // [id][TP]  carried dL/di_{k+1}
    float* dts = gic + id * TP;                   // [TP]
    float* wbuf = dts + TP;                       // [kWBuf] staged weights
    // [np_de + np_ae]: in LDS when it fits, else this workgroup's partial slice in global memory (each element is owned by
    // one thread either way, so the read-modify-write needs no atomics)
    float* gacc_g = a.wpart + (size_t)blockIdx.x * (a.de.np + (a.dae ? a.ae.np : 0));      // (used when gg)
    const bool stages = (!REG && STR != 2) || (a.dae && STR == 0);      // some MLP still stages its weights through LDS
    float* gacc_l = wbuf + (stages ? kWBuf : 0);
    // register path of the DE: quad-row buffers behind the accumulators, the wave's MFMA operands of both passes in VGPRs
    const int de_acc = (DE_TM && !gg) ? tm_total(a.de) : a.de.np;      // floats of the DE's accumulators in LDS (tile-major off the staged path)
    const int ae_at = gg ? 0 : de_acc;                                 // the AE's accumulators in LDS (when !ggA) sit behind the DE's
    const int ae_acc = (AE_TM && !ggA) ? tm_total(a.ae) : a.ae.np;
    const int np_all = ae_at + ((a.dae && !ggA) ? ae_acc : 0);        // floats of LDS accumulators
    float* qb = gacc_l + ((np_all + 3) & ~3);
    const QOff qo = q_offsets(a.de), qoA = q_offsets(a.ae);           // (one region: the two MLPs' evaluations never overlap in time)
    RegFwd rfw;
    RegBwd rbw;
    if constexpr (REG) load_reg_images(a, rfw, rbw);

    auto gb = [&](int c) -> long long { const long long b = b0 + c; return b < a.B ? b : a.B - 1; };
    auto on = [&](int c) -> bool { return b0 + c < a.B; };
    // loops over [rows][TB] tiles: idx -> (r, c)
#define TILE_LOOP(rows) for (int idx = tid, r = tid / TB, c = tid % TB; idx < (rows) * TB; idx += NT, r = idx / TB, c = idx % TB)

    for (int e = tid; e < np_all; e += NT) gacc_l[e] = 0.0f;
    // global accumulators: tile-major slices (tmpart) for the MLPs off the staged path, the natural partial slice itself for a staged one
    float* tmg = a.tmpart + (size_t)blockIdx.x * (tm_total(a.de) + (a.dae ? tm_total(a.ae) : 0));
    float* tmgA = tmg + tm_total(a.de);
    if constexpr (gg) {
        if constexpr (DE_TM) { for (int e = tid; e < tm_total(a.de); e += NT) tmg[e] = 0.0f; }
        else { for (int e = tid; e < a.de.np; e += NT) gacc_g[e] = 0.0f; }
    }
    if constexpr (ggA) {
        if (dae) {
            if constexpr (AE_TM) { for (int e = tid; e < tm_total(a.ae); e += NT) tmgA[e] = 0.0f; }
            else { for (int e = tid; e < a.ae.np; e += NT) gacc_g[a.de.np + e] = 0.0f; }
        }
    }
    TILE_LOOP(n) { a0s[r * TP + c] = a.a0[gb(c) * n + r]; ga0s[r * TP + c] = 0.0f; }
    TILE_LOOP(xd) gxc[r * TP + c] = on(c) ? a.gxs[((a.T - 1) * a.B + gb(c)) * xd + r] : 0.0f;
    TILE_LOOP(id) gic[r * TP + c] = (on(c) && a.gis) ? a.gis[((a.T - 1) * a.B + gb(c)) * id + r] : 0.0f;
    TILE_LOOP(nzv) {   // the last grid point's z|v only receive the AE part (DAE) or nothing (ODE)
        if (!on(c)) continue;
        const bool isz = r < zd;
        float* dst = isz ? a.gz : a.gv;
        if (dst) dst[((a.T - 1) * a.B + b0 + c) * (isz ? zd : vd) + (isz ? r : r - zd)] = 0.0f;
    }
    __syncthreads();

    // DE input rows of acts: a0 | s - a0 | s  with s = x | ext
    auto de_input = [&](const float* xs_rows) {
        float* u = acts + a.de.act[0] * TP;
        TILE_LOOP(n) {
            const float s = r < xd ? xs_rows[r * TP + c] : ext[(r - xd) * TP + c];
            const float i0 = a0s[r * TP + c];
            u[r * TP + c] = i0;
            u[(n + r) * TP + c] = s - i0;
            u[(2 * n + r) * TP + c] = s;
        }
        __syncthreads();
    };
    // AE input rows: a0 | x | z | v ; x from xrows (LDS) ; z|v from grid point jzv (>= 0) or from ext
    auto ae_input = [&](const float* xrows, long long jzv) {
        float* u = acts + a.ae.act[0] * TP;
        TILE_LOOP(n + xd + nzv) {
            float v;
            if (r < n) v = a0s[r * TP + c];
            else if (r < n + xd) v = xrows[(r - n) * TP + c];
            else if (jzv < 0) v = ext[(r - n - xd) * TP + c];
            else if (r < n + xd + zd) v = a.z.p[jzv * a.z.st + gb(c) * a.z.sb + (r - n - xd)];
            else v = a.v.p[jzv * a.v.st + gb(c) * a.v.sb + (r - n - xd - zd)];
            u[r * TP + c] = v;
        }
        __syncthreads();
    };
    // VJP of the AE head at (xrows; z|v of grid point jzv or the jumped ext rows) with output gradient `gi`:
    // adds to gx_dst, ga0s, and to the z|v gradients (global gz/gv at jzv, or the jump gradients of event ev)
    auto ae_vjp = [&](const float* xrows, long long jzv, int ev, const float* gi, float* gx_dst) {
        ae_input(xrows, jzv);
        if constexpr (STR >= 1) g_forward_str(a.ae, a.fimgA, acts, qb, qoA); else g_forward(a.ae, acts, wbuf);
        TILE_LOOP(id) dA[r * TP + c] = gi[r * TP + c];
        __syncthreads();
        const float* gu = STR >= 1 ? g_vjp_str<ggA>(a.ae, a.timgA, acts, dA, dB, gacc_l + ae_at, tmgA, qb, qoA)
                                   : g_vjp<ggA>(a.ae, acts, dA, dB, gacc_l + ae_at, gacc_g + a.de.np, wbuf);
        TILE_LOOP(n) ga0s[r * TP + c] += gu[r * TP + c];
        TILE_LOOP(xd) gx_dst[r * TP + c] += gu[(n + r) * TP + c];
        TILE_LOOP(nzv) {
            if (!on(c)) continue;
            const float g = gu[(n + xd + r) * TP + c];
            const bool isz = r < zd;
            const int d_ = isz ? r : r - zd, w_ = isz ? zd : vd;
            if (jzv >= 0) {
                float* dst = isz ? a.gz : a.gv;
                if (dst) dst[(jzv * a.B + b0 + c) * w_ + d_] += g;
            } else {
                float* dst = isz ? a.gzj : a.gvj;
                if (dst) dst[((b0 + c) * a.n_events + ev) * w_ + d_] += g;
            }
        }
        __syncthreads();
    };

    // Look-ahead (round 6): the rows a step reads from HBM -- the clocks, the dataset z | v, xs[k], the incoming gradient of grid point k --
    // are requested one step early into registers (items tid + 256 j, j < LA: up to 32 rows each; rows beyond that are loaded where they are
    // used), so that their latency hides behind the previous step instead of standing at the top and the bottom of every step.
    constexpr int LA = 2;
    float la_x[LA], la_g[LA], la_zv[LA], la_t = 0.0f, la_tn = 0.0f;
    auto look_ahead = [&](long long kk) {       // grid point kk >= 0
#pragma unroll
        for (int j = 0; j < LA; ++j) {
            const int idx = tid + NT * j;
            const int ix = idx < xd * TB ? idx : 0, rx = ix / TB, cx = ix % TB;
            la_x[j] = a.xs[(kk * a.B + gb(cx)) * xd + rx];
            la_g[j] = a.gxs[(kk * a.B + gb(cx)) * xd + rx];
            const int iz = idx < nzv * TB ? idx : 0, rz = iz / TB;
            const long long b = gb(iz % TB);
            la_zv[j] = nzv == 0 ? 0.0f : (rz < zd ? a.z.p[kk * a.z.st + b * a.z.sb + rz] : a.v.p[kk * a.v.st + b * a.v.sb + (rz - zd)]);
        }
        if (tid < TB) la_t = a.t.p[kk * a.t.st + gb(tid) * a.t.sb];
    };
    if (a.T >= 2) {
        if (tid < TB) la_tn = a.t.p[(a.T - 1) * a.t.st + gb(tid) * a.t.sb];
        look_ahead(a.T - 2);
    }
    for (long long k = a.T - 2; k >= 0; --k) {
        const int ev = a.ev ? a.ev[k] : -1;
        if (tid < TB) dts[tid] = la_tn - la_t;
        float gx_in[LA];                             // the incoming gradient of grid point k, consumed at the bottom of the step
#pragma unroll
        for (int j = 0; j < LA; ++j) {
            const int idx = tid + NT * j;
            gx_in[j] = la_g[j];
            if (idx < xd * TB) x0[(idx / TB) * TP + idx % TB] = la_x[j];
            if (idx < nzv * TB && ev < 0) ext[(idx / TB) * TP + idx % TB] = la_zv[j];
        }
        for (int idx = tid + NT * LA; idx < xd * TB; idx += NT) x0[(idx / TB) * TP + idx % TB] = a.xs[(k * a.B + gb(idx % TB)) * xd + idx / TB];
        TILE_LOOP(nzv) {
            if (ev < 0 && idx < NT * LA) continue;                                      // (came through the look-ahead registers)
            const long long b = gb(c);
            float v;
            if (r < zd) v = ev >= 0 ? a.zj[b * a.zjb + ev * a.zje + r] : a.z.p[k * a.z.st + b * a.z.sb + r];
            else v = ev >= 0 ? a.vj[b * a.vjb + ev * a.vje + (r - zd)] : a.v.p[k * a.v.st + b * a.v.sb + (r - zd)];
            ext[r * TP + c] = v;
        }
        if (tid < TB) la_tn = la_t;
        if (k > 0) look_ahead(k - 1);
        __syncthreads();
        if (dae) {
            // (1) AE head at the end of step k: i_{k+1} = g(x_{k+1}; z[k+1], v[k+1]) carries gic
            TILE_LOOP(xd) xst[r * TP + c] = a.xs[((k + 1) * a.B + gb(c)) * xd + r];
            __syncthreads();
            ae_vjp(xst, k + 1, -1, gic, gxc);
            // (2) algebraic input of this step's DE
            if (ev >= 0) {
                ae_input(x0, -1);
                if constexpr (STR >= 1) g_forward_str(a.ae, a.fimgA, acts, qb, qoA); else g_forward(a.ae, acts, wbuf);
                const float* out = acts + a.ae.act[a.ae.L] * TP;
                TILE_LOOP(id) ext[(nzv + r) * TP + c] = out[r * TP + c];
            } else {
                TILE_LOOP(id) ext[(nzv + r) * TP + c] = a.is_[(k * a.B + gb(c)) * id + r];
            }
            __syncthreads();
        }
        // (3a) stage inputs and slopes
        for (int s = 0; s < S; ++s) {
            TILE_LOOP(xd) {
                float acc = 0.0f;
                for (int j = 0; j < s; ++j) acc += rk_a(a.method, s, j) * ks[j * nx + r * TP + c];
                xst[s * nx + r * TP + c] = s == 0 ? x0[r * TP + c] : x0[r * TP + c] + dts[c] * acc;
            }
            __syncthreads();
            if (s + 1 < S) {               // (the last stage's slope feeds no stage input: its evaluation is (3b)'s first, not done here)
                de_input(xst + s * nx);
                if constexpr (REG) g_forward_reg(a, acts, qb, qo, rfw);
                else if constexpr (STR == 2) g_forward_str(a.de, a.fimg, acts, qb, qo);
                else g_forward(a.de, acts, wbuf);
                const float* out = acts + a.de.act[a.de.L] * TP;
                TILE_LOOP(xd) ks[s * nx + r * TP + c] = out[r * TP + c];
                __syncthreads();
            }
        }
        // (3b) stages backwards
        TILE_LOOP(xd) {
            const float g1 = gxc[r * TP + c];
            gx0[r * TP + c] = g1;
            for (int s = 0; s < S; ++s) gks[s * nx + r * TP + c] = dts[c] * rk_b(a.method, s) * g1;
        }
        TILE_LOOP(ne) gext[r * TP + c] = 0.0f;
        __syncthreads();
        for (int s = S - 1; s >= 0; --s) {
            de_input(xst + s * nx);
            if constexpr (REG) { if (PSNODE_K5_ABL != 3) g_forward_reg(a, acts, qb, qo, rfw); }
            else if constexpr (STR == 2) g_forward_str(a.de, a.fimg, acts, qb, qo);
            else g_forward(a.de, acts, wbuf);
            TILE_LOOP(xd) dA[r * TP + c] = gks[s * nx + r * TP + c];
            __syncthreads();
            const float* gu = REG ? g_vjp_reg<gg>(a, acts, dA, dB, gacc_l, tmg, qb, qo, rbw)
                                  : (STR == 2 ? g_vjp_str<gg>(a.de, a.timg, acts, dA, dB, gacc_l, tmg, qb, qo) : g_vjp<gg>(a.de, acts, dA, dB, gacc_l, gacc_g, wbuf));
            TILE_LOOP(n) {
                const float gs = gu[(n + r) * TP + c] + gu[(2 * n + r) * TP + c];
                ga0s[r * TP + c] += gu[r * TP + c] - gu[(n + r) * TP + c];
                if (r < xd) {
                    gx0[r * TP + c] += gs;
                    for (int j = 0; j < s; ++j) gks[j * nx + r * TP + c] += dts[c] * rk_a(a.method, s, j) * gs;
                } else {
                    gext[(r - xd) * TP + c] += gs;
                }
            }
            __syncthreads();
        }
        // (4) gradients of this step's external inputs
        TILE_LOOP(nzv) {
            if (!on(c)) continue;
            const float g = gext[r * TP + c];
            const bool isz = r < zd;
            const int d_ = isz ? r : r - zd, w_ = isz ? zd : vd;
            float* dst = isz ? a.gz : a.gv;
            float* dj = isz ? a.gzj : a.gvj;
            if (ev >= 0) {
                if (dj) dj[((b0 + c) * a.n_events + ev) * w_ + d_] = g;
                if (dst) dst[(k * a.B + b0 + c) * w_ + d_] = 0.0f;
            } else if (dst) {
                dst[(k * a.B + b0 + c) * w_ + d_] = g;
            }
        }
        if (dae) {
            __syncthreads();
            if (ev >= 0) {   // i_in = g(x_k; jumps): its gradient flows into x_k and the jump inputs; i_k itself was unused
                ae_vjp(x0, -1, ev, gext + nzv * TP, gx0);
                TILE_LOOP(id) gic[r * TP + c] = (on(c) && a.gis) ? a.gis[(k * a.B + gb(c)) * id + r] : 0.0f;
            } else {
                TILE_LOOP(id) gic[r * TP + c] = gext[(nzv + r) * TP + c] + ((on(c) && a.gis) ? a.gis[(k * a.B + gb(c)) * id + r] : 0.0f);
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < LA; ++j) {
            const int idx = tid + NT * j;
            if (idx < xd * TB) gxc[(idx / TB) * TP + idx % TB] = gx0[(idx / TB) * TP + idx % TB] + (on(idx % TB) ? gx_in[j] : 0.0f);
        }
        for (int idx = tid + NT * LA; idx < xd * TB; idx += NT) {
            const int r = idx / TB, c = idx % TB;
            gxc[r * TP + c] = gx0[r * TP + c] + (on(c) ? a.gxs[(k * a.B + gb(c)) * xd + r] : 0.0f);
        }
        __syncthreads();
    }
    if (dae) {   // i_0 = g(x_0; z[0], v[0])   (my_solvers.py:95)
        TILE_LOOP(xd) x0[r * TP + c] = a.xs[gb(c) * xd + r];
        __syncthreads();
        ae_vjp(x0, 0, -1, gic, gxc);
    }
    TILE_LOOP(xd) if (on(c)) a.gx0[(b0 + c) * xd + r] = gxc[r * TP + c];
    TILE_LOOP(n) if (on(c)) a.ga0[(b0 + c) * n + r] = ga0s[r * TP + c];
    float* wp = a.wpart + (size_t)blockIdx.x * (a.de.np + (dae ? a.ae.np : 0));
    // the LDS accumulators -> this workgroup's partial in nn.Linear order (tile-major ones un-permuted)
    // (volatile: the global tile-major slices were written by other lanes of this workgroup; read them past the vector L1)
    auto unpermute = [&](const GMlp& m, const volatile float* base, float* dst) {
        for (int l = 0; l < m.L; ++l) {
            const int N = m.out_dim[l], K = l ? m.out_dim[l - 1] : m.in_dim, ntk = (K + 15) / 16;
            const volatile float* tw = base + tm_dw_off(m, l);
            for (int e = tid; e < N * K; e += NT) {
                const int j = e / K, k = e % K;
                dst[m.gw[l] + e] = tw[(((j >> 4) * ntk + (k >> 4)) * 64 + ((j & 15) >> 2) * 16 + (k & 15)) * 4 + (j & 3)];
            }
            for (int e = tid; e < N; e += NT) dst[m.gb[l] + e] = base[tm_db_off(m, l) + e];
        }
    };
    if constexpr (!ggA) {
        if (dae) {
            if constexpr (AE_TM) unpermute(a.ae, gacc_l + ae_at, wp + a.de.np);
            else for (int e = tid; e < a.ae.np; e += NT) wp[a.de.np + e] = gacc_l[ae_at + e];
        }
    }
    if constexpr (!gg) {
        if constexpr (DE_TM) unpermute(a.de, gacc_l, wp);
        else for (int e = tid; e < a.de.np; e += NT) wp[e] = gacc_l[e];
    }
    if constexpr (gg && DE_TM) { __threadfence(); __syncthreads(); unpermute(a.de, tmg, wp); }
    if constexpr (ggA && AE_TM) { if (dae) { __threadfence(); __syncthreads(); unpermute(a.ae, tmgA, wp + a.de.np); } }
#undef TILE_LOOP
}

int fill_gmlp(const psnode_mlp_f32& m, GMlp& g, float*& ws) {
    g.L = m.n_layers;
    g.in_dim = m.in_dim;
    int k = m.in_dim, off = 0, rows = 0;
    g.act[0] = 0;
    rows = m.in_dim;
    for (int l = 0; l < m.n_layers; ++l) {
        g.out_dim[l] = m.out_dim[l];
        g.w[l] = m.weight[l];
        g.b[l] = m.bias[l];
        g.wt[l] = ws;
        ws += ((size_t)k * m.out_dim[l] + 63) / 64 * 64;
        g.gw[l] = off; off += m.out_dim[l] * k;
        g.gb[l] = off; off += m.out_dim[l];
        g.act[l + 1] = rows;
        rows += m.out_dim[l];
        k = m.out_dim[l];
    }
    g.np = off;
    return rows;
}

size_t gbwd_lds_floats(const GBwd& a) {
    const int vd = a.dae ? a.vd : 0, id = a.dae ? a.id : 0, ne = a.zd + vd + id, n = a.xd + ne;
    const bool de_tm = a.de_reg || a.str == 2, ae_tm = a.str >= 1;
    const size_t de_acc = a.gacc_global == 1 ? 0 : (size_t)(de_tm ? tm_total(a.de) : a.de.np);
    const size_t np_all = de_acc + ((a.dae && a.gacc_global == 0) ? (size_t)(ae_tm ? tm_total(a.ae) : a.ae.np) : 0);
    const bool stages = (!a.de_reg && a.str != 2) || (a.dae && a.str == 0);
    size_t q = 0;
    if (de_tm) q = (size_t)q_offsets(a.de).total;
    if (a.dae && ae_tm && (size_t)q_offsets(a.ae).total > q) q = (size_t)q_offsets(a.ae).total;
    return (size_t)a.act_rows * TP + 2 * (size_t)a.maxw * TP + 2 * (size_t)n * TP + 2 * (size_t)ne * TP + (size_t)a.xd * TP * (1 + 12 + 2) +
           (size_t)id * TP + TP + (stages ? kWBuf : 0) + ((np_all + 3) & ~(size_t)3) + q;
}
// the DE's shape class of the register path
bool de_reg_class(const psnode_mlp_f32& de) {
    if (de.n_layers > 4 || de.in_dim > 128) return false;
    for (int l = 0; l < de.n_layers; ++l)
        if (de.out_dim[l] > 64) return false;
    return true;
}
size_t tm_floats(const psnode_mlp_f32& de, const psnode_mlp_f32* ae) {      // one workgroup's tile-major global accumulators (both MLPs)
    size_t tot = 0;
    for (int m = 0; m < (ae ? 2 : 1); ++m) {
        const psnode_mlp_f32& mm = m ? *ae : de;
        size_t t = 0;
        int k = mm.in_dim;
        for (int l = 0; l < mm.n_layers; ++l) { t += (size_t)up16(mm.out_dim[l]) * up16(k) + mm.out_dim[l]; k = mm.out_dim[l]; }
        tot += (t + 3) & ~(size_t)3;
    }
    return tot;
}
size_t reg_image_floats(const psnode_mlp_f32& de) {       // plain + transposed images of every layer
    size_t tot = 0;
    int k = de.in_dim;
    for (int l = 0; l < de.n_layers; ++l) {
        tot += (generic_image_floats(k, de.out_dim[l]) + 63) / 64 * 64 + (generic_image_floats(de.out_dim[l], k) + 63) / 64 * 64;
        k = de.out_dim[l];
    }
    return tot;
}
// 1: everything in LDS; 2: only with the parameter-gradient accumulators in global memory; 0: does not fit.  a.de_reg (the DE's class
// allows the register path) is kept when its quad-row buffers fit next to the LDS accumulators, else dropped.
int gbwd_mode(GBwd& a) {
    const int want_reg = a.de_reg;
    // paths in order of preference: register DE (+ streamed AE head), everything streamed, then the staged paths; for each, the accumulators
    // in LDS, the AE's in the global slice, both there
    const int cand[4][2] = {{want_reg, a.dae ? 1 : 0}, {0, 2}, {want_reg, 0}, {0, 0}};      // {de_reg, str}
    for (int c = 0; c < 4; ++c) {
        if (c == 0 && !want_reg) continue;
        if (c == 2 && (!want_reg || !a.dae)) continue;
        a.de_reg = cand[c][0]; a.str = cand[c][1];
        a.gacc_global = 0;
        if (gbwd_lds_floats(a) * sizeof(float) <= 160 * 1024) return 1;
        if (a.dae && (a.de_reg || a.str == 2)) {
            a.gacc_global = 2;
            if (gbwd_lds_floats(a) * sizeof(float) <= 160 * 1024) return 2;
        }
        a.gacc_global = 1;
        if (gbwd_lds_floats(a) * sizeof(float) <= 160 * 1024) return 2;
    }
    return 0;
}

int mlp_maxw(const psnode_mlp_f32& m) {
    int w = m.in_dim;
    for (int l = 0; l < m.n_layers; ++l) w = m.out_dim[l] > w ? m.out_dim[l] : w;
    return w;
}
size_t mlp_wt_floats(const psnode_mlp_f32& m) {
    size_t tot = 0;
    int k = m.in_dim;
    for (int l = 0; l < m.n_layers; ++l) { tot += ((size_t)k * m.out_dim[l] + 63) / 64 * 64; k = m.out_dim[l]; }
    return tot;
}
int mlp_np(const psnode_mlp_f32& m) {
    int np = 0, k = m.in_dim;
    for (int l = 0; l < m.n_layers; ++l) { np += m.out_dim[l] * (k + 1); k = m.out_dim[l]; }
    return np;
}
bool mlp_ok(const psnode_mlp_f32& m, int in_dim, int out_dim) {
    if (m.n_layers < 1 || m.n_layers > kMaxLayers || m.in_dim != in_dim || m.out_dim[m.n_layers - 1] != out_dim) return false;
    for (int l = 0; l < m.n_layers; ++l)
        if (m.out_dim[l] < 1 || m.out_dim[l] > PSNODE_MAX_WIDTH || !m.weight[l] || !m.bias[l]) return false;
    return true;
}

}  // namespace

// shared by the ODE and DAE entry points (psnode_backward.hip calls this for kernel = generic / unsupported MFMA shapes)
size_t generic_bwd_workspace_floats(const psnode_mlp_f32* de, const psnode_mlp_f32* ae, long long B) {
    const size_t nwg = (size_t)((B + TB - 1) / TB);
    return mlp_wt_floats(*de) + (ae ? mlp_wt_floats(*ae) : 0) + nwg * (size_t)(mlp_np(*de) + (ae ? mlp_np(*ae) : 0)) + 64 +
           reg_image_floats(*de) + 64 + (ae ? reg_image_floats(*ae) + 64 : 0) + nwg * tm_floats(*de, ae) + 64;
}

int generic_bwd_fits(const psnode_mlp_f32* de, const psnode_mlp_f32* ae, int xd, int zd, int vd, int id) {
    GBwd a;
    memset(&a, 0, sizeof(a));
    a.dae = ae != nullptr; a.xd = xd; a.zd = zd; a.vd = vd; a.id = id;
    float* ws = nullptr;
    int rows = fill_gmlp(*de, a.de, ws);
    a.maxw = mlp_maxw(*de);
    if (ae) {
        const int r2 = fill_gmlp(*ae, a.ae, ws);
        rows = r2 > rows ? r2 : rows;
        a.maxw = mlp_maxw(*ae) > a.maxw ? mlp_maxw(*ae) : a.maxw;
    }
    a.act_rows = rows;
    a.de_reg = de_reg_class(*de) ? 1 : 0;
    return gbwd_mode(a);
}

// launches pack (transpose), the backward kernel and the partial reduction
int generic_backward_launch(int method, int xd, int zd, int vd, int id, long long T, long long B, const psnode_mlp_f32* de,
                            const psnode_mlp_f32* ae, ViewDev t, ViewDev z, ViewDev v, const float* a0, const int* ev, const float* zj,
                            long long zjb, long long zje, const float* vj, long long vjb, long long vje, int n_events, const float* xs,
                            const float* is_, const float* gxs, const float* gis, float* gx0, float* gz, float* gv, float* gzj, float* gvj,
                            float* ga0, float* gparams_de, float* gparams_ae, float* workspace, hipStream_t stream) {
    const bool dae = ae != nullptr;
    const int n = xd + zd + (dae ? vd + id : 0);
    if (!mlp_ok(*de, 3 * n, xd)) return PSNODE_ERR_DIMS;
    if (dae && !mlp_ok(*ae, n + xd + zd + vd, id)) return PSNODE_ERR_DIMS;
    GBwd a;
    memset(&a, 0, sizeof(a));
    a.method = method; a.dae = dae; a.xd = xd; a.zd = zd; a.vd = vd; a.id = id; a.T = T; a.B = B;
    float* ws = workspace;
    int rows = fill_gmlp(*de, a.de, ws);
    a.maxw = mlp_maxw(*de);
    if (dae) {
        const int r2 = fill_gmlp(*ae, a.ae, ws);
        rows = r2 > rows ? r2 : rows;
        a.maxw = mlp_maxw(*ae) > a.maxw ? mlp_maxw(*ae) : a.maxw;
    }
    a.act_rows = rows;
    a.t = t; a.z = z; a.v = v; a.a0 = a0; a.ev = ev; a.zj = zj; a.zjb = zjb; a.zje = zje; a.vj = vj; a.vjb = vjb; a.vje = vje;
    a.n_events = n_events; a.xs = xs; a.is_ = is_; a.gxs = gxs; a.gis = gis; a.gx0 = gx0; a.gz = gz; a.gv = gv; a.gzj = gzj; a.gvj = gvj;
    a.ga0 = ga0;
    a.de_reg = de_reg_class(*de) ? 1 : 0;
    float* img[kMaxLayers] = {}, *imgT[kMaxLayers] = {}, *imgA[kMaxLayers] = {}, *imgTA[kMaxLayers] = {};
    {                           // the plain / transposed images of both MLPs sit in front of the per-workgroup partials
        ws = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255);
        for (int m = 0; m < (dae ? 2 : 1); ++m) {
            const psnode_mlp_f32* mm = m ? ae : de;
            int k = mm->in_dim;
            for (int l = 0; l < mm->n_layers; ++l) {
                float* f = ws; ws += (generic_image_floats(k, mm->out_dim[l]) + 63) / 64 * 64;
                float* t_ = ws; ws += (generic_image_floats(mm->out_dim[l], k) + 63) / 64 * 64;
                if (m) { imgA[l] = f; imgTA[l] = t_; a.fimgA[l] = f; a.timgA[l] = t_; }
                else { img[l] = f; imgT[l] = t_; a.fimg[l] = f; a.timg[l] = t_; }
                k = mm->out_dim[l];
            }
        }
    }
    a.wpart = ws;
    {
        const size_t nwg_ = (size_t)((B + TB - 1) / TB);
        float* tm = ws + nwg_ * (size_t)(a.de.np + (dae ? a.ae.np : 0));
        a.tmpart = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(tm) + 255) & ~(uintptr_t)255);
    }
    if (!gbwd_mode(a)) return PSNODE_ERR_UNSUPPORTED;
    const size_t lds = gbwd_lds_floats(a) * sizeof(float);
    // transposed weights for the forward recomputation
    MlpDev mde, mae;
    memset(&mde, 0, sizeof(mde));
    memset(&mae, 0, sizeof(mae));
    auto to_dev = [](const GMlp& g, MlpDev& m) {
        m.n_layers = g.L; m.in_dim = g.in_dim;
        for (int l = 0; l < g.L; ++l) { m.out_dim[l] = g.out_dim[l]; m.w[l] = g.w[l]; m.wt[l] = g.wt[l]; m.bias[l] = g.b[l]; }
    };
    to_dev(a.de, mde);
    if (dae) to_dev(a.ae, mae);
    if (launch_pack_transpose(mde, dae ? &mae : nullptr, stream) != hipSuccess) return PSNODE_ERR_HIP;
    if ((a.de_reg || a.str == 2) && launch_pack_plain_images(mde, img, imgT, stream) != hipSuccess) return PSNODE_ERR_HIP;
    if (dae && a.str >= 1 && launch_pack_plain_images(mae, imgA, imgTA, stream) != hipSuccess) return PSNODE_ERR_HIP;
    // <DE accumulators global, DE on the register path, AE accumulators global, streamed MLPs>
    void (*kern)(const GBwd) = nullptr;
    const int g = a.gacc_global;
    if (a.de_reg && a.str == 1) kern = g == 1 ? &generic_backward_kernel<true, true, true, 1> : (g == 2 ? &generic_backward_kernel<false, true, true, 1> : &generic_backward_kernel<false, true, false, 1>);
    else if (a.de_reg) kern = g == 1 ? &generic_backward_kernel<true, true, true, 0> : (g == 2 ? &generic_backward_kernel<false, true, true, 0> : &generic_backward_kernel<false, true, false, 0>);
    else if (a.str == 2) kern = g == 1 ? &generic_backward_kernel<true, false, true, 2> : (g == 2 ? &generic_backward_kernel<false, false, true, 2> : &generic_backward_kernel<false, false, false, 2>);
    else kern = g ? &generic_backward_kernel<true, false, true, 0> : &generic_backward_kernel<false, false, false, 0>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return PSNODE_ERR_HIP;
    const unsigned nwg = (unsigned)((B + TB - 1) / TB);
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(NT), lds, stream, a);
    if (hipGetLastError() != hipSuccess) return PSNODE_ERR_HIP;
    return launch_reduce_partials(a.wpart, gparams_de, gparams_ae, a.de.np, dae ? a.ae.np : 0, (int)nwg, stream) == hipSuccess ? PSNODE_OK : PSNODE_ERR_HIP;
}

}  // namespace psnode
