// K6: masked, column-weighted squared-error loss + its gradient in one pass over (pred, target, mask).
// HBM-bound: per (t,b) row it reads pred (4D B) + target (4D B) + mask (4 B) and writes grad (4D B); nothing else.
// The two operands arrive in different layouts -- pred is the integrator's time-major [T,B,D], target/mask are the
// scripts' B-major [B,T,D] viewed as [T,B,D] -- so one of them is always strided for a thread-per-row mapping.  Each
// workgroup therefore walks tiles of TT grid points x TB trajectories (TB*D = 256 floats): target/mask are read in
// THEIR contiguous order and transposed into a time-major LDS tile (row pitch chosen so that both the transposing
// writes and the float4 reads are bank-conflict-free), then pred is streamed with float4 loads against the LDS copy and
// grad is streamed out with float4 stores.  Sums: registers across tiles -> fixed-order wave butterflies -> one partial
// row per workgroup -> a second tiny kernel adds the partial rows in double, fixed order (run-to-run bit-identical).
// Shapes off the fast path (D not a power of two, unaligned or oddly strided views) take the scalar kernel below it.
// Replaces: neural_00_ODE_01_no_encode.py:353-355, neural_01_DAE_01_no_encode.py:414-419 (include/psnode_hip.h).
#include "psnode_common.h"

namespace psnode {
namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int NT = 256;        // threads per workgroup
constexpr int TT = 16;         // grid points per tile
constexpr int kMaxLossD = 64;  // widest row
constexpr int kMaxBlocks = 2048;

struct LossDev {
    long long T, B;
    int D, mask_w, TB, pitch, mpitch;   // scalar kernel: TB = trajectories per tile; LDS pitches per trajectory (floats)
    ViewDev pred, target, mask;
    const float* col_w;
    const float* inv_norm;
    float scale, t0_coef;
    float* grad;
    float* partial;    // [n_blocks][D+1]
    float* out;
    long long tiles_b, n_tiles;
};

// ---------------------------------------------------------------------------------------------------------------
// fast path: D in {1,2,4,...,64}; pred dense time-major (stride_b == D, 16-B aligned rows); target (and a width-D mask)
// dense along time (stride_t == D).  VT: target/mask rows are 16-B aligned, stage them with float4 loads.
template <int D, bool VT>
__global__ __launch_bounds__(NT) void masked_mse_fast_kernel(const LossDev a) {
    constexpr int TB = NT / D;                      // trajectories per tile: one tile row = 256 floats
    constexpr int PK = NT + (D > 4 ? D : 4);        // LDS pitch of one grid point (floats)
    constexpr int MPK = TB + 4;                     // pitch of the width-1 mask tile
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* tg = lds;                                // [TT][PK]
    float* mk = lds + TT * PK;                      // [TT][PK] (mask width D) or [TT][MPK] (width 1)

    const int tid = threadIdx.x, kq = tid >> 6, c = tid & 63;
    const int mw = a.mask_w;
    const float norm = a.scale * (a.inv_norm ? a.inv_norm[0] : 1.0f);
    const float t0c = a.t0_coef;
    float cw[4], acc[4] = {0.f, 0.f, 0.f, 0.f}, acc0 = 0.0f;
#pragma unroll
    for (int e = 0; e < 4; ++e) cw[e] = a.col_w ? a.col_w[(4 * c + e) % D] : 1.0f;

    for (long long tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const long long b0 = (tile % a.tiles_b) * TB, k0 = (tile / a.tiles_b) * TT;
        const int nb = (int)(a.B - b0 < TB ? a.B - b0 : TB), nk = (int)(a.T - k0 < TT ? a.T - k0 : TT);

        // ---- pred: issue the tile's loads first (4 grid points per thread, floats 4c..4c+3 of the 256-float tile row)
        const bool colv = (4 * c) / D < nb;
        f4 p4[TT / 4];
#pragma unroll
        for (int it = 0; it < TT / 4; ++it) {
            const int kk = kq + 4 * it;
            p4[it] = (colv && kk < nk) ? *reinterpret_cast<const f4*>(a.pred.p + (k0 + kk) * a.pred.st + b0 * D + 4 * c)
                                       : f4{0.f, 0.f, 0.f, 0.f};
        }
        // ---- target (and width-D mask): contiguous along time per trajectory -> time-major LDS tile
        auto stage = [&](const ViewDev& v, float* dst) {
            if constexpr (VT) {
                constexpr int V = TT * D / 4, PER = TB * V / NT;   // float4 per trajectory; per thread
                f4 r[PER];
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const int idx = tid + NT * i, bl = idx / V, q = idx % V, kk = (4 * q) / D, d0 = (4 * q) % D;
                    r[i] = (bl < nb && kk < nk) ? *reinterpret_cast<const f4*>(v.p + (b0 + bl) * v.sb + (k0 + kk) * D + d0)
                                                : f4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const int idx = tid + NT * i, bl = idx / V, q = idx % V, kk = (4 * q) / D, d0 = (4 * q) % D;
                    *reinterpret_cast<f4*>(dst + kk * PK + bl * D + d0) = r[i];
                }
            } else {
                constexpr int R = TT * D, PER = TB * R / NT;       // floats per trajectory; per thread (= TT)
                float r[PER];
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const int idx = tid + NT * i, bl = idx / R, q = idx % R;
                    r[i] = (bl < nb && q / D < nk) ? v.p[(b0 + bl) * v.sb + k0 * D + q] : 0.0f;
                }
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const int idx = tid + NT * i, bl = idx / R, q = idx % R;
                    dst[(q / D) * PK + bl * D + (q % D)] = r[i];
                }
            }
        };
        stage(a.target, tg);
        if (mw == D && D > 1) stage(a.mask, mk);
        else if (mw == 1) {
            for (int idx = tid; idx < TB * TT; idx += NT) {
                const int bl = idx / TT, kk = idx % TT;
                mk[kk * MPK + bl] = (bl < nb && kk < nk) ? a.mask.p[(b0 + bl) * a.mask.sb + (k0 + kk) * a.mask.st] : 0.0f;
            }
        }
        __syncthreads();

        // ---- stream against the tile
#pragma unroll
        for (int it = 0; it < TT / 4; ++it) {
            const int kk = kq + 4 * it;
            const long long k = k0 + kk;
            const f4 t4 = *reinterpret_cast<const f4*>(tg + kk * PK + 4 * c);
            f4 m4 = f4{1.f, 1.f, 1.f, 1.f};
            if (mw == D && D > 1) m4 = *reinterpret_cast<const f4*>(mk + kk * PK + 4 * c);
            else if (mw == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) m4[e] = mk[kk * MPK + (4 * c + e) / D];
            }
            f4 g4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float er = p4[it][e] - t4[e];
                const float wm = cw[e] * m4[e];
                acc[e] += wm * er * er;
                g4[e] = 2.0f * norm * wm * er;
                if (k == 0) { acc0 += er * er; g4[e] += 2.0f * t0c * er; }
            }
            if (a.grad && colv && kk < nk) *reinterpret_cast<f4*>(a.grad + (k * a.B + b0) * D + 4 * c) = g4;
        }
        __syncthreads();
    }

    // ---- workgroup sums, fixed order.  Column of acc[e] = (4c+e) % D: lanes that differ only in bits >= log2(D/4)
    //      of c hold the same columns -> xor butterflies over those bits.
    constexpr int LOW = D >= 4 ? D / 4 : 1;
#pragma unroll
    for (int m = 32; m >= LOW; m >>= 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += __shfl_xor(acc[e], m, 64);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc0 += __shfl_xor(acc0, m, 64);
    float* red = lds;                               // [4 waves][D + 1]  (tiles are done: the last loop ended on a barrier)
    if (c < LOW) {
        if constexpr (D >= 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) red[kq * (D + 1) + 4 * c + e] = acc[e];
        } else if constexpr (D == 2) {
            red[kq * 3 + 0] = acc[0] + acc[2];
            red[kq * 3 + 1] = acc[1] + acc[3];
        } else {
            red[kq * 2] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        }
        if (c == 0) red[kq * (D + 1) + D] = acc0;
    }
    __syncthreads();
    if (tid <= D) {
        const float s = (red[tid] + red[(D + 1) + tid]) + (red[2 * (D + 1) + tid] + red[3 * (D + 1) + tid]);
        a.partial[(long long)blockIdx.x * (D + 1) + tid] = s * (tid < D ? norm : t0c);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// scalar path: any D <= 64, any strides.  One tile per workgroup, thread = (trajectory, column).
__global__ __launch_bounds__(NT) void masked_mse_kernel(const LossDev a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* tg = lds;                                   // [TB][pitch]   target tile  (b, k, d)
    float* mk = tg + (size_t)a.TB * a.pitch;           // [TB][mpitch]  mask tile
    float* red = mk + (size_t)a.TB * a.mpitch;         // [2][NT]       block reduction

    const int tid = threadIdx.x, D = a.D, TB = a.TB;
    const long long tile = blockIdx.x;
    const long long b0 = (tile % a.tiles_b) * TB, k0 = (tile / a.tiles_b) * TT;
    const int nb = (int)(a.B - b0 < TB ? a.B - b0 : TB), nk = (int)(a.T - k0 < TT ? a.T - k0 : TT);

    // target / mask tile -> LDS, walked in the operand's own contiguous order
    const bool b_major = a.target.st <= a.target.sb;   // [B,T,D] memory: time is the inner run
    const int row = TT * D;
    for (int idx = tid; idx < TB * row; idx += NT) {
        int bl, kk, d;
        if (b_major) { bl = idx / row; kk = (idx % row) / D; d = idx % D; }
        else { kk = idx / (TB * D); bl = (idx % (TB * D)) / D; d = idx % D; }
        if (bl < nb && kk < nk) tg[bl * a.pitch + kk * D + d] = a.target.p[(k0 + kk) * a.target.st + (b0 + bl) * a.target.sb + d];
    }
    if (a.mask_w > 0) {
        const int mw = a.mask_w, mrow = TT * mw;
        const bool mb_major = a.mask.st <= a.mask.sb;
        for (int idx = tid; idx < TB * mrow; idx += NT) {
            int bl, kk, d;
            if (mb_major) { bl = idx / mrow; kk = (idx % mrow) / mw; d = idx % mw; }
            else { kk = idx / (TB * mw); bl = (idx % (TB * mw)) / mw; d = idx % mw; }
            if (bl < nb && kk < nk) mk[bl * a.mpitch + kk * mw + d] = a.mask.p[(k0 + kk) * a.mask.st + (b0 + bl) * a.mask.sb + d];
        }
    }
    __syncthreads();

    const int bl = tid / D, d = tid % D;
    const float norm = a.scale * (a.inv_norm ? a.inv_norm[0] : 1.0f);
    float acc = 0.0f, acc0 = 0.0f;
    if (bl < nb) {
        const float cw = a.col_w ? a.col_w[d] : 1.0f;
        const float* pp = a.pred.p + (b0 + bl) * a.pred.sb + d;
        float* gp = a.grad ? a.grad + (b0 + bl) * D + d : nullptr;
        const float* tl = tg + bl * a.pitch + d;
        const float* ml = mk + bl * a.mpitch + (a.mask_w == D ? d : 0);
        const int mw = a.mask_w;
#pragma unroll 4
        for (int kk = 0; kk < nk; ++kk) {
            const long long k = k0 + kk;
            const float e = pp[k * a.pred.st] - tl[kk * D];
            const float wm = cw * (mw > 0 ? ml[kk * mw] : 1.0f);
            acc += wm * e * e;
            float g = 2.0f * norm * wm * e;
            if (k == 0) { acc0 = e * e; g += 2.0f * a.t0_coef * e; }
            if (gp) gp[k * a.B * D] = g;
        }
    }
    red[tid] = acc;
    red[NT + tid] = acc0;
    __syncthreads();
    if (tid <= D) {
        float s = 0.0f;
        if (tid < D) { for (int q = 0; q < nb; ++q) s += red[q * D + tid]; s *= norm; }
        else { for (int q = 0; q < nb * D; ++q) s += red[NT + q]; s *= a.t0_coef; }
        a.partial[tile * (D + 1) + tid] = s;
    }
}

// out[c] = sum over rows of partial[row][c] (double, fixed order); out[D+1] = total.
// Thread (part, col): part-strided row sums, then column `col` adds its parts in order.
__global__ __launch_bounds__(NT) void masked_mse_finish_kernel(const float* __restrict__ partial, long long rows, int D, float* out) {
    __shared__ double red[NT];
    __shared__ double cols[kMaxLossD + 1];
    const int tid = threadIdx.x, C = D + 1, P = NT / C;
    const int part = tid / C, col = tid % C;
    double s = 0.0;
    if (part < P) {      // four independent chains (the loads of one chain are dependent round trips: 20 us for 2 k partial rows); fixed order
        double s4[4] = {0.0, 0.0, 0.0, 0.0};
        long long q = part;
        for (; q + 3 * P < rows; q += 4 * P) {
#pragma unroll
            for (int j = 0; j < 4; ++j) s4[j] += (double)partial[(q + (long long)j * P) * C + col];
        }
        for (int j = 0; q < rows; q += P, ++j) s4[j] += (double)partial[q * C + col];
        s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    }
    red[tid] = s;
    __syncthreads();
    if (tid < C) {
        double tsum = 0.0;
        for (int q = 0; q < P; ++q) tsum += red[q * C + tid];
        cols[tid] = tsum;
        out[tid] = (float)tsum;
    }
    __syncthreads();
    if (tid == 0) {
        double tot = 0.0;
        for (int q = 0; q < C; ++q) tot += cols[q];
        out[D + 1] = (float)tot;
    }
}

bool plan(const psnode_loss_args_f32* a, LossDev& d) {
    if (a->T < 1 || a->B < 1 || a->D < 1 || a->D > kMaxLossD) return false;
    if (a->mask_width != 0 && a->mask_width != 1 && a->mask_width != a->D) return false;
    d.T = a->T; d.B = a->B; d.D = a->D; d.mask_w = a->mask_width;
    d.TB = NT / a->D;
    // pitch == D (mod 64): lanes (bl, d) of one wave hit 64 distinct LDS banks
    auto pitch_for = [&](int w) { return w > 0 ? ((TT * w - w + 63) / 64) * 64 + w : 0; };
    d.pitch = pitch_for(a->D);
    d.mpitch = pitch_for(a->mask_width);
    d.tiles_b = (a->B + d.TB - 1) / d.TB;
    d.n_tiles = d.tiles_b * ((a->T + TT - 1) / TT);
    return true;
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// 0: scalar kernel; 1: fast kernel, scalar staging; 2: fast kernel, float4 staging
int fast_mode(const psnode_loss_args_f32* a) {
    const int D = a->D;
    if (D & (D - 1)) return 0;
    if (a->pred.stride_b != D || (a->pred.stride_t & 3) || !aligned16(a->pred.ptr) || ((a->B * D) & 3)) return 0;
    if (a->grad_pred && !aligned16(a->grad_pred)) return 0;
    if (a->target.stride_t != D) return 0;
    if (a->mask_width == D && D > 1 && a->mask.stride_t != D) return 0;
    bool vt = D >= 4 && !(a->target.stride_b & 3) && aligned16(a->target.ptr);
    if (a->mask_width == D && D > 1) vt = vt && !(a->mask.stride_b & 3) && aligned16(a->mask.ptr);
    return vt ? 2 : 1;
}

long long block_rows(const psnode_loss_args_f32* a, const LossDev& d) {
    if (!fast_mode(a)) return d.n_tiles;
    return d.n_tiles < kMaxBlocks ? d.n_tiles : kMaxBlocks;
}

template <int D>
hipError_t launch_fast(const LossDev& d, int mode, unsigned blocks, hipStream_t s) {
    constexpr int PK = NT + (D > 4 ? D : 4), MPK = NT / D + 4;
    size_t fl = (size_t)TT * PK + (d.mask_w == D && D > 1 ? (size_t)TT * PK : (d.mask_w == 1 ? (size_t)TT * MPK : 0));
    if (fl < 4 * (D + 1)) fl = 4 * (D + 1);
    if constexpr (D >= 4) {
        if (mode == 2) {
            hipLaunchKernelGGL((masked_mse_fast_kernel<D, true>), dim3(blocks), dim3(NT), fl * sizeof(float), s, d);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL((masked_mse_fast_kernel<D, false>), dim3(blocks), dim3(NT), fl * sizeof(float), s, d);
    return hipGetLastError();
}

}  // namespace
}  // namespace psnode

using namespace psnode;

extern "C" size_t psnode_masked_mse_workspace_bytes(const psnode_loss_args_f32* a) {
    LossDev d;
    if (!a || !plan(a, d)) return 0;
    // sized for the scalar kernel (one row per tile) so that the same workspace serves either kernel
    return (size_t)d.n_tiles * (a->D + 1) * sizeof(float);
}

extern "C" int32_t psnode_masked_mse_f32(const psnode_loss_args_f32* a, void* workspace, size_t workspace_bytes, void* stream) {
    if (!a) return PSNODE_ERR_NULL;
    LossDev d;
    if (a->T < 1 || a->B < 1 || a->D < 1) return PSNODE_ERR_DIMS;
    if (a->mask_width != 0 && a->mask_width != 1 && a->mask_width != a->D) return PSNODE_ERR_DIMS;
    if (!plan(a, d)) return PSNODE_ERR_UNSUPPORTED;
    if (!a->pred.ptr || !a->target.ptr || !a->out || (a->mask_width > 0 && !a->mask.ptr)) return PSNODE_ERR_NULL;
    if (d.n_tiles > 0x7fffffffLL) return PSNODE_ERR_UNSUPPORTED;
    const size_t need = (size_t)d.n_tiles * (a->D + 1) * sizeof(float);
    if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 15)) return PSNODE_ERR_WORKSPACE;
    d.pred = {a->pred.ptr, a->pred.stride_t, a->pred.stride_b};
    d.target = {a->target.ptr, a->target.stride_t, a->target.stride_b};
    d.mask = {a->mask.ptr, a->mask.stride_t, a->mask.stride_b};
    d.col_w = a->col_weight; d.inv_norm = a->inv_norm; d.scale = a->scale; d.t0_coef = a->t0_coef;
    d.grad = a->grad_pred; d.partial = static_cast<float*>(workspace); d.out = a->out;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int mode = fast_mode(a);
    const long long rows = block_rows(a, d);
    hipError_t e;
    if (mode) {
        switch (a->D) {
            case 1: e = launch_fast<1>(d, mode, (unsigned)rows, s); break;
            case 2: e = launch_fast<2>(d, mode, (unsigned)rows, s); break;
            case 4: e = launch_fast<4>(d, mode, (unsigned)rows, s); break;
            case 8: e = launch_fast<8>(d, mode, (unsigned)rows, s); break;
            case 16: e = launch_fast<16>(d, mode, (unsigned)rows, s); break;
            case 32: e = launch_fast<32>(d, mode, (unsigned)rows, s); break;
            default: e = launch_fast<64>(d, mode, (unsigned)rows, s); break;
        }
    } else {
        const size_t lds = ((size_t)d.TB * (d.pitch + d.mpitch) + 2 * NT) * sizeof(float);
        hipLaunchKernelGGL(masked_mse_kernel, dim3((unsigned)d.n_tiles), dim3(NT), lds, s, d);
        e = hipGetLastError();
    }
    if (e != hipSuccess) return PSNODE_ERR_HIP;
    hipLaunchKernelGGL(masked_mse_finish_kernel, dim3(1), dim3(NT), 0, s, d.partial, rows, a->D, a->out);
    return hipGetLastError() == hipSuccess ? PSNODE_OK : PSNODE_ERR_HIP;
}
