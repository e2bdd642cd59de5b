// Issue cost of the DPP dot product used by K3f (psnode_latent_dpp.hip): 16 x v_fmac_f32_dpp row_newbcast with 1 / 2 / 4
// accumulator chains, each dot depending on the previous one's result (like layer -> layer), 1 and 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -w profiles/scripts/ubench_dpp.hip -o /tmp/ubench_dpp && /tmp/ubench_dpp
#include <hip/hip_runtime.h>
#include <cstdio>
#define T(ACC, N, W) "v_fmac_f32_dpp %" #ACC ", %[s], %[w" #W "] row_newbcast:" #N " row_mask:0xf bank_mask:0xf\n\t"
#define M(ACC, N, W) "v_mul_f32_dpp %" #ACC ", %[s], %[w" #W "] row_newbcast:" #N " row_mask:0xf bank_mask:0xf\n\t"
#define WOPS [w0] "v"(w[0]), [w1] "v"(w[1]), [w2] "v"(w[2]), [w3] "v"(w[3]), [w4] "v"(w[4]), [w5] "v"(w[5]), [w6] "v"(w[6]), [w7] "v"(w[7]), \
             [w8] "v"(w[8]), [w9] "v"(w[9]), [w10] "v"(w[10]), [w11] "v"(w[11]), [w12] "v"(w[12]), [w13] "v"(w[13]), [w14] "v"(w[14]), [w15] "v"(w[15])
template <int CH, int NOP>
__device__ __forceinline__ float dot16(float acc, float src, const float (&w)[16]) {
    float b, c, d;
    if constexpr (CH == 1) {
        if constexpr (NOP) asm volatile("s_nop 1");
        asm volatile(T(0,0,0) T(0,1,1) T(0,2,2) T(0,3,3) T(0,4,4) T(0,5,5) T(0,6,6) T(0,7,7) T(0,8,8) T(0,9,9) T(0,10,10) T(0,11,11) T(0,12,12) T(0,13,13) T(0,14,14) T(0,15,15)
            : "+v"(acc) : [s] "v"(src), WOPS);
    } else if constexpr (CH == 2) {
        if constexpr (NOP) asm volatile("s_nop 1");
        asm volatile(T(0,0,0) M(1,1,1) T(0,2,2) T(1,3,3) T(0,4,4) T(1,5,5) T(0,6,6) T(1,7,7) T(0,8,8) T(1,9,9) T(0,10,10) T(1,11,11) T(0,12,12) T(1,13,13) T(0,14,14) T(1,15,15)
            "v_add_f32 %0, %0, %1" : "+v"(acc), "=&v"(b) : [s] "v"(src), WOPS);
    } else {
        if constexpr (NOP) asm volatile("s_nop 1");
        asm volatile(T(0,0,0) M(1,1,1) M(2,2,2) M(3,3,3) T(0,4,4) T(1,5,5) T(2,6,6) T(3,7,7) T(0,8,8) T(1,9,9) T(2,10,10) T(3,11,11) T(0,12,12) T(1,13,13) T(2,14,14) T(3,15,15)
            "v_add_f32 %0, %0, %1\n\tv_add_f32 %2, %2, %3\n\tv_add_f32 %0, %0, %2" : "+v"(acc), "=&v"(b), "=&v"(c), "=&v"(d) : [s] "v"(src), WOPS);
    }
    return acc;
}
template <int CH, int NOP>
__global__ void bench(float* out, long long* cyc, int niter) {
    float w[16];
    for (int i = 0; i < 16; ++i) w[i] = 0.01f * (threadIdx.x + i) - 0.3f;
    float x = threadIdx.x * 0.001f;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < niter; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) { x = dot16<CH, NOP>(0.1f, x, w); asm volatile("v_max_f32 %0, %0, %0" : "+v"(x)); }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}
template <int CH, int NOP>
void run(int wpw, float* out, long long* cyc) {
    const int niter = 2000;
    bench<CH, NOP><<<256, 64 * wpw>>>(out, cyc, 10);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); bench<CH, NOP><<<256, 64 * wpw>>>(out, cyc, niter); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("chains=%d s_nop=%d waves/SIMD=%d : %.1f cycles per dependent dot16 (+1 v_max) per wave, wall %.1f ns\n", CH, NOP, wpw / 4, c / (8.0 * niter), ms * 1e6 / (8.0 * niter));
}
int main() {
    float* out; long long* cyc; hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8);
    run<1, 1>(4, out, cyc); run<2, 1>(4, out, cyc); run<4, 1>(4, out, cyc); run<1, 0>(4, out, cyc); run<2, 0>(4, out, cyc);
    run<1, 1>(8, out, cyc); run<2, 1>(8, out, cyc); run<4, 1>(8, out, cyc);
    return 0;
}
