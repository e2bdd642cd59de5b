// Micro-benchmark behind DESIGN.md "K1 at B = 4096: why 0.57": what one cross-wave exchange of a 16-trajectory tile costs on gfx950 when
// every SIMD holds ONE wave (B = 4096 = 256 tiles of 16 trajectories = one 4-wave workgroup per CU, nothing else to switch to).
//   hipcc --offload-arch=gfx950 -O3 profiles/scripts/ubench_exchange.hip -o /tmp/ubx && /tmp/ubx
// One iteration = one H -> H layer of K1 at hidden 64 as the kernel runs it: publish the wave's 16 x 16 activations (ds_write_b128),
// issue the 4 MFMAs that only need the wave's own block, barrier (LDS-only fence), read the three other waves' blocks (ds_read_b128),
// 12 MFMAs, then NV independent VALU instructions standing in for the ELU.  Variants:
//   MODE 0  the layer as described (16 MFMA + exchange + NV VALU)
//   MODE 1  the same without the exchange (16 MFMA + NV VALU): the issue-bound floor
//   MODE 2  the exchange alone (write, barrier, 3 reads, one dependent v_add per block): its latency
//   MODE 3  the layer with NO own-block MFMAs in front of the barrier (all 16 behind it): what the overlap buys
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f4 mf(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int MODE, int NV>
__global__ __launch_bounds__(256) void bench(float* out, long long* cyc, int niter) {
    __shared__ f4 xb[2][4][64];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    float wreg[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) wreg[i] = 0.001f * (i + 1) + 1e-4f * l;
    f4 h = f4{0.1f + 0.001f * l, 0.2f, 0.3f, 0.4f};
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = 0.5f + i;
    int p = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < niter; ++it) {
        f4 accA = f4{0.f, 0.f, 0.f, 0.f}, accB = accA;
        if constexpr (MODE != 1) xb[p][w][l] = h;
        if constexpr (MODE == 0 || MODE == 1) {
            accA = mf(wreg[0], h[0], accA); accB = mf(wreg[1], h[1], accB);
            accA = mf(wreg[2], h[2], accA); accB = mf(wreg[3], h[3], accB);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MODE != 1) lds_barrier();
        f4 v[4];
        v[0] = h;
#pragma unroll
        for (int c = 1; c < 4; ++c) v[c] = MODE == 1 ? h : xb[p][(w + c) & 3][l];
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MODE == 2) {
#pragma unroll
            for (int c = 1; c < 4; ++c) accA += v[c];
        } else {
#pragma unroll
            for (int c = (MODE == 3 ? 0 : 1); c < 4; ++c) {
                accA = mf(wreg[4 * c + 0], v[c][0], accA); accB = mf(wreg[4 * c + 1], v[c][1], accB);
                accA = mf(wreg[4 * c + 2], v[c][2], accA); accB = mf(wreg[4 * c + 3], v[c][3], accB);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        h = accA + accB;
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            float& x = f[q & 7];
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(h[q & 3]), "v"(wreg[q & 15]));
        }
        h[0] = h[0] * 1e-3f + f[0] * 1e-6f;       // keep the chain finite and dependent on the VALU block
        p ^= 1;
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = h[0] + h[1] + h[2] + h[3];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE, int NV>
void run(const char* name, float* out, long long* cyc) {
    const int niter = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    bench<MODE, NV><<<256, 256>>>(out, cyc, 10);
    hipEventRecord(e0);
    bench<MODE, NV><<<256, 256>>>(out, cyc, niter);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    printf("%-46s NV=%2d : %7.1f ns per layer (wall, 256 workgroups = 1 per CU), s_memtime %.2f ticks (x24 = %.0f shader cycles at 2.4 GHz)\n", name, NV,
           ms * 1e6 / niter, (double)c / niter, (double)c / niter * 24.0);
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * sizeof(float)); hipMalloc(&cyc, 8);
    run<1, 0>("16 MFMA, no exchange", out, cyc);
    run<2, 0>("exchange alone (write, barrier, 3 reads)", out, cyc);
    run<0, 0>("layer: 4 MFMA | exchange | 12 MFMA", out, cyc);
    run<3, 0>("layer: exchange | 16 MFMA (no overlap)", out, cyc);
    run<1, 24>("16 MFMA + 24 VALU, no exchange", out, cyc);
    run<0, 24>("layer + 24 VALU (K1's ELU per layer)", out, cyc);
    run<3, 24>("layer, no overlap, + 24 VALU", out, cyc);
    run<1, 48>("16 MFMA + 48 VALU, no exchange", out, cyc);
    run<0, 48>("layer + 48 VALU", out, cyc);
    return 0;
}
