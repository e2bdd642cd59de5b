// Issue cost of the VALU instructions the kernels lean on (gfx950), one and two waves per SIMD, alone and after an fp32 MFMA.
//   hipcc --offload-arch=gfx950 -O3 -w profiles/scripts/ubench_valu.hip -o /tmp/ubench_valu && /tmp/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define OPS 64
template <int KIND, int WITH_MFMA>
__global__ void bench(float* out, long long* cyc, int niter) {
    float a = threadIdx.x * 0.001f - 0.03f, b = 1.0f + threadIdx.x * 0.002f;
    float f[8];
    f2 p[8];
    f4 acc = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) { f[i] = a + i; p[i] = f2{a + i, b - i}; }
    f2 pa = f2{a, b}, pb = f2{b, a};
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < niter; ++it) {
#pragma unroll
        for (int m = 0; m < OPS; ++m) {
            float& x = f[m & 7];
            f2& q = p[m & 7];
            if constexpr (WITH_MFMA) { if ((m & 7) == 0) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0); }
            if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b));
            if constexpr (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(q) : "v"(pa), "v"(pb));
            if constexpr (KIND == 2) asm volatile("v_exp_f32 %0, %1" : "=v"(x) : "v"(a));
            if constexpr (KIND == 3) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(a), "v"(b));
            if constexpr (KIND == 4) asm volatile("v_med3_f32 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b));
            if constexpr (KIND == 5) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(q) : "v"(pa));
            if constexpr (KIND == 6) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(q) : "v"(pa));
            if constexpr (KIND == 7) asm volatile("v_max_f32 %0, %1, %0" : "+v"(x) : "v"(a));
            if constexpr (KIND == 8) asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "=v"(x) : "v"(a));
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = acc[0] + acc[1] + acc[2] + acc[3];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += f[i] + p[i][0] + p[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND, int WITH_MFMA>
void run(const char* name, int waves_per_wg, float* out, long long* cyc) {
    const int niter = 1000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    bench<KIND, WITH_MFMA><<<256, 64 * waves_per_wg>>>(out, cyc, 10);
    hipEventRecord(e0);
    bench<KIND, WITH_MFMA><<<256, 64 * waves_per_wg>>>(out, cyc, niter);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    const double n = (double)OPS * niter;
    printf("%-28s %s waves/SIMD=%d : %.2f cycles per op per wave (s_memtime), wall %.3f ns/op\n", name,
           WITH_MFMA ? "[+1 mfma16x16x4 per 8 ops]" : "[alone]                  ", waves_per_wg / 4, c / n, ms * 1e6 / n);
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 1024 * sizeof(float)); hipMalloc(&cyc, 8);
#define ROW(K, NAME) run<K, 0>(NAME, 4, out, cyc); run<K, 0>(NAME, 8, out, cyc); run<K, 1>(NAME, 4, out, cyc); run<K, 1>(NAME, 8, out, cyc);
    ROW(0, "v_fma_f32")
    ROW(1, "v_pk_fma_f32")
    ROW(2, "v_exp_f32")
    ROW(3, "v_fmac_f32_dpp row_newbcast")
    ROW(4, "v_med3_f32")
    ROW(5, "v_pk_add_f32")
    ROW(6, "v_pk_mul_f32")
    ROW(7, "v_max_f32")
    ROW(8, "v_mov_b32_dpp row_newbcast")
    return 0;
}
