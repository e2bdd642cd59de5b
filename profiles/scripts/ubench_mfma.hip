// Micro-benchmark behind DESIGN.md's K1 decomposition choice: issue cost of fp32 MFMA shapes next to VALU work on gfx950.
//   hipcc --offload-arch=gfx950 -O3 profiles/scripts/ubench_mfma.hip -o /tmp/ubench && /tmp/ubench
// Every kernel runs NITER iterations of a body of NM MFMAs (NACC independent accumulators) each followed by NV independent
// v_fma_f32; cycles per MFMA are taken from s_memtime of wave 0 of block 0 and from the wall clock of the whole grid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int NV, int NACC>
__global__ void bench(float* out, long long* cyc, int niter) {
    f4 acc[NACC];
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    float f[8];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = a + i;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < niter; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if constexpr (SHAPE == 0) acc[m % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m % NACC], 0, 0, 0);
            else acc[m % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[m % NACC], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                float& x = f[(m * NV + v) & 7];
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}

template <int SHAPE, int NV, int NACC>
void run(const char* name, int waves_per_wg, float* out, long long* cyc) {
    const int niter = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    bench<SHAPE, NV, NACC><<<256, 64 * waves_per_wg>>>(out, cyc, 10);
    hipEventRecord(e0);
    bench<SHAPE, NV, NACC><<<256, 64 * waves_per_wg>>>(out, cyc, niter);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    const double nm = 16.0 * niter;
    printf("%-10s NV=%d NACC=%d waves/SIMD=%d : s_memtime %.1f ticks/MFMA (100 MHz ticks x24 = %.1f shader cycles @2.4GHz), wall %.2f ns/MFMA/wave-slot\n", name, NV, NACC,
           waves_per_wg / 4, c / nm, c / nm * 24.0, ms * 1e6 / nm);
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 1024 * sizeof(float)); hipMalloc(&cyc, 8);
#define ROW(S, NAME, NACC) \
    run<S, 0, NACC>(NAME, 4, out, cyc); run<S, 1, NACC>(NAME, 4, out, cyc); run<S, 2, NACC>(NAME, 4, out, cyc); run<S, 3, NACC>(NAME, 4, out, cyc); \
    run<S, 4, NACC>(NAME, 4, out, cyc); run<S, 6, NACC>(NAME, 4, out, cyc); run<S, 8, NACC>(NAME, 4, out, cyc); \
    run<S, 0, NACC>(NAME, 8, out, cyc); run<S, 1, NACC>(NAME, 8, out, cyc); run<S, 2, NACC>(NAME, 8, out, cyc); run<S, 4, NACC>(NAME, 8, out, cyc);
    ROW(0, "16x16x4", 2)
    ROW(1, "4x4x1", 4)
    ROW(1, "4x4x1", 1)
    ROW(1, "4x4x1", 2)
    return 0;
}
