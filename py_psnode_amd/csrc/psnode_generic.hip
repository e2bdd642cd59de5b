// Generic fused fixed-grid integrator for gfx950: any layer widths / state dims within the ABI limits.
//
// One launch integrates ALL T-1 steps: a workgroup owns TB trajectories (they never interact,
// my_solvers.py:66 is row-wise over the batch) and walks the time grid with its state in LDS.
// This is the always-available HIP path; shapes that have an MFMA specialisation use psnode_mfma.hip.
//
// Work split per Linear layer: item (j, g) = output unit j x group of 4 trajectories; activations live
// in LDS as [unit][TB] so the 4 trajectories of a group are one ds_read_b128, weights are read
// transposed ([in][out], packed into the workspace) so consecutive lanes read consecutive floats.
// Every dot product is an fp32 fmaf chain in input order.
#include "psnode_common.h"

namespace psnode {

namespace {

constexpr int TB = 16;    // trajectories per workgroup
constexpr int NT = 256;   // threads per workgroup (4 waves)

// MLP over the TB columns. `in` holds [K][TB]; returns the buffer holding [N_last][TB].
// Each layer's transposed weights ([in][out]) are staged through the LDS buffer `wbuf` in chunks of whole input rows,
// so the inner loop reads one conflict-free ds_read_b32 (weight) and one broadcast ds_read_b128 (4 trajectories) per
// FMA quad instead of a global load.  Partial sums of multi-chunk layers live in `out`.  Barrier after every chunk.
__device__ float* mlp_eval(const MlpDev& m, float* in, float* out, float* wbuf) {
    int K = m.in_dim;
    for (int l = 0; l < m.n_layers; ++l) {
        const int N = m.out_dim[l];
        const float* __restrict__ wt = m.wt[l];
        const float* __restrict__ bias = m.bias[l];
        const bool last = (l + 1 == m.n_layers);
        const int KC = kWBuf / N > 0 ? kWBuf / N : 1;            // input rows per staged chunk (N <= 1024 < kWBuf)
        for (int k0 = 0; k0 < K; k0 += KC) {
            const int kc = K - k0 < KC ? K - k0 : KC;
            stage_weights(wt + (size_t)k0 * N, wbuf, kc * N);
            const bool first = k0 == 0, final = k0 + kc >= K;
            for (int item = threadIdx.x; item < N * (TB / 4); item += NT) {
                const int j = item % N, g = item / N;
                float4 acc;
                if (first) { const float b = bias[j]; acc = make_float4(b, b, b, b); }
                else acc = *reinterpret_cast<const float4*>(out + j * TB + g * 4);
                const float* col = in + (k0 * TB) + g * 4;
                const float* w = wbuf + j;
#pragma unroll 8
                for (int k = 0; k < kc; ++k) {
                    const float wk = w[k * N];
                    const float4 v = *reinterpret_cast<const float4*>(col + k * TB);
                    acc.x = fmaf(wk, v.x, acc.x); acc.y = fmaf(wk, v.y, acc.y); acc.z = fmaf(wk, v.z, acc.z); acc.w = fmaf(wk, v.w, acc.w);
                }
                if (final && !last) { acc.x = elu1(acc.x); acc.y = elu1(acc.y); acc.z = elu1(acc.z); acc.w = elu1(acc.w); }
                *reinterpret_cast<float4*>(out + j * TB + g * 4) = acc;
            }
            __syncthreads();
        }
        float* tmp = in; in = out; out = tmp;
        K = N;
    }
    return in;
}

template <bool DAE>
__global__ __launch_bounds__(NT) void generic_kernel(const IntegrateDev a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const long long b0 = (long long)blockIdx.x * TB;
    const int xd = a.xd, zd = a.zd;
    const int vd = DAE ? a.vd : 0, id = DAE ? a.id : 0;
    const int n = xd + zd + vd + id;   // width of all_initial
    const int ne = n - xd;             // external rows: z | v | i

    float* actA = lds;
    float* actB = actA + a.maxw * TB;  // ping-pong partner: holds layer outputs only -> maxo rows
    float* a0 = actB + a.maxo * TB;    // [n][TB]
    float* ext = a0 + n * TB;          // [ne][TB] z | v | i fed to the DE stages of this step
    float* xcur = ext + ne * TB;       // [xd][TB] running state
    float* xsrc = xcur + xd * TB;      // [xd][TB] start of this step (xcur, or dataset x under teacher forcing)
    float* xst = xsrc + xd * TB;       // [xd][TB] stage argument
    float* kbuf = xst + xd * TB;       // [4][xd][TB]
    float* icur = kbuf + 4 * xd * TB;  // [id][TB]
    float* dts = icur + id * TB;       // [TB]
    float* wbuf = dts + TB;            // [kWBuf] staged weights

    auto gb = [&](int c) -> long long { const long long b = b0 + c; return b < a.B ? b : a.B - 1; };
    const bool true_x = (a.flags & PSNODE_FLAG_INPUT_TRUE_X) != 0;
    const bool true_i = DAE && (a.flags & PSNODE_FLAG_INPUT_TRUE_I) != 0;
    const int nx = xd * TB;

    // ---- per-trajectory constants and the initial state
    for (int idx = tid; idx < n * TB; idx += NT) {
        const int r = idx / TB, c = idx % TB;
        a0[idx] = a.a0[gb(c) * n + r];
    }
    for (int idx = tid; idx < nx; idx += NT) {
        const int r = idx / TB, c = idx % TB;
        const long long b = gb(c);
        const float v = DAE ? a.x_init[b * xd + r] : a.x.p[b * a.x.sb + r];
        xcur[idx] = v;
        if (b0 + c < a.B) a.xo[b * xd + r] = v;
    }
    __syncthreads();

    // AE head g(x; z, v) -> icur.  jx >= 0: x from the dataset at grid point jx, else xcur.
    // jzv >= 0: z, v from the dataset at grid point jzv, else the (possibly jumped) rows of `ext`.
    auto ae_eval = [&](long long jx, long long jzv) {
        if constexpr (DAE) {
            const int m = n + xd + zd + vd;
            for (int idx = tid; idx < m * TB; idx += NT) {
                const int r = idx / TB, c = idx % TB;
                const long long b = gb(c);
                float v;
                if (r < n) v = a0[idx];
                else if (r < n + xd) v = jx >= 0 ? a.x.p[jx * a.x.st + b * a.x.sb + (r - n)] : xcur[(r - n) * TB + c];
                else if (r < n + xd + zd) v = jzv >= 0 ? a.z.p[jzv * a.z.st + b * a.z.sb + (r - n - xd)] : ext[(r - n - xd) * TB + c];
                else v = jzv >= 0 ? a.v.p[jzv * a.v.st + b * a.v.sb + (r - n - xd - zd)] : ext[(r - n - xd) * TB + c];
                actA[idx] = v;
            }
            __syncthreads();
            const float* out = mlp_eval(a.ae, actA, actB, wbuf);
            for (int idx = tid; idx < id * TB; idx += NT) icur[idx] = out[idx];
            __syncthreads();
        }
    };

    if constexpr (DAE) {
        ae_eval(true_x ? 0 : -1, 0);   // my_solvers.py:95
        for (int idx = tid; idx < id * TB; idx += NT) {
            const int r = idx / TB, c = idx % TB;
            if (b0 + c < a.B) a.io[(b0 + c) * id + r] = icur[idx];
        }
    }

    // DE right-hand side at xst with this step's frozen externals -> pointer to [xd][TB]
    auto de_eval = [&]() -> const float* {
        for (int idx = tid; idx < n * TB; idx += NT) {
            const int r = idx / TB;
            const float s = r < xd ? xst[idx] : ext[idx - nx];
            const float i0 = a0[idx];
            actA[idx] = i0;
            actA[n * TB + idx] = s - i0;
            actA[2 * n * TB + idx] = s;
        }
        __syncthreads();
        return mlp_eval(a.de, actA, actB, wbuf);
    };

    const int nstage = a.method == PSNODE_EULER ? 1 : (a.method == PSNODE_MIDPOINT ? 2 : 4);

    for (long long k = 0; k + 1 < a.T; ++k) {
        const int ev = a.ev ? a.ev[k] : -1;
        // ---- this step's inputs (zero-order hold: the left grid point feeds every stage)
        if (tid < TB) {
            const long long b = gb(tid);
            dts[tid] = a.t.p[(k + 1) * a.t.st + b * a.t.sb] - a.t.p[k * a.t.st + b * a.t.sb];
        }
        for (int idx = tid; idx < (zd + vd) * TB; idx += NT) {
            const int r = idx / TB, c = idx % TB;
            const long long b = gb(c);
            float v;
            if (r < zd) v = ev >= 0 ? a.zj[b * a.zjb + ev * a.zje + r] : a.z.p[k * a.z.st + b * a.z.sb + r];
            else v = ev >= 0 ? a.vj[b * a.vjb + ev * a.vje + (r - zd)] : a.v.p[k * a.v.st + b * a.v.sb + (r - zd)];
            ext[idx] = v;
        }
        for (int idx = tid; idx < nx; idx += NT) {
            const int r = idx / TB, c = idx % TB;
            const float v = true_x ? a.x.p[k * a.x.st + gb(c) * a.x.sb + r] : xcur[idx];
            xsrc[idx] = v;
            xst[idx] = v;
        }
        __syncthreads();
        if constexpr (DAE) {
            if (ev >= 0) ae_eval(-1, -1);   // my_solvers.py:110: i0 = i_func(x0, z0_jump, v0_jump)
            for (int idx = tid; idx < id * TB; idx += NT) {
                const int r = idx / TB, c = idx % TB;
                ext[(zd + vd) * TB + idx] = true_i ? a.i.p[k * a.i.st + gb(c) * a.i.sb + r] : icur[idx];
            }
            __syncthreads();
        }

        // ---- stages (my_fixed_grid.py:15-18, 23-32, 38-51)
        for (int s = 0; s < nstage; ++s) {
            const float* f = de_eval();
            for (int idx = tid; idx < nx; idx += NT) {
                const float h = dts[idx % TB];
                const float x0 = xsrc[idx];
                const float ks = f[idx];
                kbuf[s * nx + idx] = ks;
                if (a.method == PSNODE_EULER) {
                    xcur[idx] = x0 + h * ks;
                } else if (a.method == PSNODE_MIDPOINT) {
                    if (s == 0) xst[idx] = x0 + ks * (0.5f * h);
                    else xcur[idx] = x0 + h * ks;
                } else {
                    const float k1 = kbuf[idx];
                    if (s == 0) xst[idx] = x0 + h * k1 * kOneThird;
                    else if (s == 1) xst[idx] = x0 + h * (ks - k1 * kOneThird);
                    else if (s == 2) xst[idx] = x0 + h * (k1 - kbuf[nx + idx] + ks);
                    else xcur[idx] = x0 + (k1 + 3.0f * (kbuf[nx + idx] + kbuf[2 * nx + idx]) + ks) * h * 0.125f;
                }
            }
            __syncthreads();
        }

        for (int idx = tid; idx < nx; idx += NT) {
            const int r = idx / TB, c = idx % TB;
            if (b0 + c < a.B) a.xo[((k + 1) * a.B + b0 + c) * xd + r] = xcur[idx];
        }
        if constexpr (DAE) {
            ae_eval(true_x ? k + 1 : -1, k + 1);   // my_solvers.py:121
            for (int idx = tid; idx < id * TB; idx += NT) {
                const int r = idx / TB, c = idx % TB;
                if (b0 + c < a.B) a.io[((k + 1) * a.B + b0 + c) * id + r] = icur[idx];
            }
        }
    }
}

}  // namespace

namespace {
struct PackArgs {
    int n;                       // layers in total (de then ae)
    int K[2 * kMaxLayers], N[2 * kMaxLayers];
    const float* w[2 * kMaxLayers];
    float* wt[2 * kMaxLayers];
};

// wt[k][j] = w[j][k] for every layer of both MLPs in one launch (blockIdx.y = layer).
__global__ void pack_transpose_kernel(const PackArgs p) {
    const int l = blockIdx.y;
    const int K = p.K[l], N = p.N[l];
    const float* __restrict__ w = p.w[l];
    float* __restrict__ wt = p.wt[l];
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < K * N; idx += gridDim.x * blockDim.x) {
        const int k = idx / N, j = idx % N;
        wt[idx] = w[(size_t)j * K + k];
    }
}

void add_pack(PackArgs& p, const MlpDev& d) {
    int k = d.in_dim;
    for (int l = 0; l < d.n_layers; ++l) {
        p.K[p.n] = k;
        p.N[p.n] = d.out_dim[l];
        p.w[p.n] = d.w[l];
        p.wt[p.n] = const_cast<float*>(d.wt[l]);
        ++p.n;
        k = d.out_dim[l];
    }
}

}  // namespace

// wt[k][j] = w[j][k] for every layer of one or two MLPs, one launch
hipError_t launch_pack_transpose(const MlpDev& de, const MlpDev* ae, hipStream_t stream) {
    PackArgs p;
    p.n = 0;
    add_pack(p, de);
    if (ae) add_pack(p, *ae);
    hipLaunchKernelGGL(pack_transpose_kernel, dim3(8, p.n), dim3(256), 0, stream, p);
    return hipGetLastError();
}

size_t generic_lds_bytes(const IntegrateDev& a, bool dae) {
    const int vd = dae ? a.vd : 0, id = dae ? a.id : 0;
    const int n = a.xd + a.zd + vd + id;
    const size_t rows = (size_t)a.maxw + a.maxo + n + (n - a.xd) + 3 * (size_t)a.xd + 4 * (size_t)a.xd + id + 1;
    return (rows * TB + kWBuf) * sizeof(float);
}

hipError_t launch_generic(const IntegrateDev& a, bool dae, hipStream_t stream) {
    const size_t lds = generic_lds_bytes(a, dae);
    const unsigned grid = (unsigned)((a.B + TB - 1) / TB);
    hipError_t e;
    if (dae) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&generic_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(generic_kernel<true>, dim3(grid), dim3(NT), lds, stream, a);
    } else {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&generic_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(generic_kernel<false>, dim3(grid), dim3(NT), lds, stream, a);
    }
    return hipGetLastError();
}

}  // namespace psnode
