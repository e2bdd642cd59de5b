// Micro-benchmark behind DESIGN.md "K1 at B = 4096: the exchange-free 4x4x1 tile" (round 5).
//   hipcc --offload-arch=gfx950 -O3 profiles/scripts/ubench_4x4.hip -o /tmp/ub4 && /tmp/ub4
// The candidate decomposition: ONE wave owns 4 trajectories and ALL 64 hidden units -- no LDS, no barrier.  One H -> H layer is
//   D[traj r][unit 4b+c] += act[k][traj r] * W[unit 4b+c][k]      64 x v_mfma_f32_4x4x1_16B_f32 (16 blocks of 4x4x1),
// A = activations (lane (b,t) register c holds act[k = 4b+c][traj t]; CBSZ = 4 / ABID = b broadcasts block b's four rows to all 16 blocks),
// B = weights (lane (b',c') holds W[4b'+c'][k]: one VGPR per k, 64 per layer), D lane (b,c) register r = out[unit 4b+c][traj r].  The
// layer-to-layer re-layout (D -> A) is a 4 x 4 transpose inside every lane quad: 8 v_cndmask_b32_dpp (quad_perm), no LDS.
// Compared against profiles/scripts/ubench_exchange.hip (today's 4-wave tile: 16 x v_mfma_f32_16x16x4_f32 + one LDS exchange per layer;
// 343.8 ns bare, 391.0 ns with 24 VALU instructions of ELU).  Modes:
//   0  64 MFMA (2 accumulator chains) only                                   : the issue floor of the 4x4x1 form
//   1  + the in-quad transpose (8 v_cndmask_b32_dpp)
//   2  + the ELU of the wave's 4 values per lane (elu_quad of psnode_common.h: the same values per lane as today)
//   3  as 2 with FOUR accumulator chains (3 packed adds more)
//   4  as 2 with the round-5 ELU (the clamp of the negative side rides on v_exp_f32's output modifier: 3 instead of 4 issue slots per value)
//   5  64 MFMA without the A broadcast (CBSZ = 0): is the 10-cycle issue interval a property of the broadcast?
//   6  the whole layer with the scaled-domain ELU;  7 / 8  64 MFMA with one / two SALU instructions behind each: are they free?
// It also checks, bit for bit over 2^24 inputs, that the round-5 ELU form equals round 3's.
// Grid: 1024 single-wave workgroups (= B 4096: one wave per SIMD), and 2048 (two per SIMD) for reference.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int ABID, bool BC = true>
__device__ __forceinline__ f4 mf(float a, float b, f4 c) {
    if constexpr (BC) return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, ABID, 0);
    else return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f4 elu_quad(f4 v) {
    const float L2E = 1.44269504088896340736f;
    f4 o;
#pragma unroll
    for (int i = 0; i < 4; i += 2) {
        f2 xn = f2{fminf(v[i], 0.f), fminf(v[i + 1], 0.f)};
        f2 y = xn * L2E;
        f2 t = f2{__builtin_amdgcn_exp2f(y[0]), __builtin_amdgcn_exp2f(y[1])};
        f2 u = t - 1.0f;
        o[i] = fmaxf(v[i], u[0]); o[i + 1] = fmaxf(v[i + 1], u[1]);
    }
    return o;
}

// round 5: exp2(min(y, 0)) == clamp(exp2(y)) to [0, 1] -- the VOP3 output modifier of v_exp_f32 replaces the v_min_f32
__device__ __forceinline__ f4 elu_quad_clamp(f4 v) {
    const float L2E = 1.44269504088896340736f;
    f4 o;
#pragma unroll
    for (int i = 0; i < 4; i += 2) {
        f2 y = f2{v[i], v[i + 1]} * L2E;
        f2 t = f2{__builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(y[0]), 0.f, 1.f), __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(y[1]), 0.f, 1.f)};
        f2 u = t - 1.0f;
        o[i] = fmaxf(v[i], u[0]); o[i + 1] = fmaxf(v[i + 1], u[1]);
    }
    return o;
}
__device__ __forceinline__ f4 elu_quad_scaled(f4 v) {
    const f2 c = f2{1.44269504088896340736f, 1.44269504088896340736f}, nc = -c;
    const f2 ea = f2{__builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(v[0]), 0.f, 1.f), __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(v[1]), 0.f, 1.f)};
    const f2 eb = f2{__builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(v[2]), 0.f, 1.f), __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(v[3]), 0.f, 1.f)};
    const f2 ua = __builtin_elementwise_fma(c, ea, nc), ub = __builtin_elementwise_fma(c, eb, nc);
    return f4{fmaxf(v[0], ua[0]), fmaxf(v[1], ua[1]), fmaxf(v[2], ub[0]), fmaxf(v[3], ub[1])};
}
__global__ void elu_forms(unsigned* ndiff) {
    // every float whose low 8 mantissa bits are a fixed pattern: 2^24 values over the whole exponent range, both signs
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned d = 0;
#pragma unroll 1
    for (unsigned lo = 0; lo < 256; lo += 85) {
        const float x = __uint_as_float((i << 8) | lo);
        if (x != x) continue;
        const f4 a = elu_quad(f4{x, x, x, x}), b = elu_quad_clamp(f4{x, x, x, x});
        d += __float_as_uint(a[0]) != __float_as_uint(b[0]);
    }
    if (d) atomicAdd(ndiff, d);
}

// 4x4 transpose of (r0..r3) x (lanes 4q..4q+3): two butterfly stages of 4 v_cndmask_b32_dpp each (D = vcc ? src1 : dpp(src0)).
// One asm block: the hazard recognizer does not look inside, so it carries its own wait states (VALU write -> DPP read: 2).
__device__ __forceinline__ f4 quad_transpose(f4 v, const int) {
    float a0, a1, a2, a3, o0, o1, o2, o3;
    const unsigned long long EVEN = 0x5555555555555555ull, LO = 0x3333333333333333ull;
    asm volatile(
        "s_nop 1\n\t"
        "s_mov_b64 vcc, %12\n\t"
        "v_cndmask_b32_dpp %0, %9, %8, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %2, %11, %10, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_not_b64 vcc, vcc\n\t"
        "v_cndmask_b32_dpp %1, %8, %9, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %3, %10, %11, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_mov_b64 vcc, %13\n\t"
        "v_cndmask_b32_dpp %4, %2, %0, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %5, %3, %1, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_not_b64 vcc, vcc\n\t"
        "v_cndmask_b32_dpp %6, %0, %2, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %7, %1, %3, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
        : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "s"(EVEN), "s"(LO)
        : "vcc", "scc");
    return f4{o0, o1, o2, o3};
}

template <int C, int B0, int NACC, bool BC = true, int FILL = 0>
__device__ __forceinline__ void quarter(const float (&wreg)[64], const float hc, f4 (&acc)[4], unsigned& g_sfill) {
    // the 16 k's that register c of the A operand carries: k = 4b + c, ABID = b
#define STEP(B) acc[(B) % NACC] = mf<B, BC>(hc, wreg[4 * (B) + C], acc[(B) % NACC]); \
    if constexpr (FILL == 1) { asm volatile("s_add_u32 %0, %0, 1" : "+s"(g_sfill)); } \
    if constexpr (FILL == 2) { asm volatile("s_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 3" : "+s"(g_sfill)); }
    STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7) STEP(8) STEP(9) STEP(10) STEP(11) STEP(12) STEP(13) STEP(14) STEP(15)
#undef STEP
}

// mode 9: the 64 MFMAs of a layer with the B operands READ FROM AccVGPRs (inline asm, `a` constraint -- K2x's hh_layer_acc): is an AGPR source slower?
template <int BB>
__device__ __forceinline__ void block_acc(const float (&wk)[64], const f4 hA, f4& accA, f4& accB) {
    asm volatile(
        "v_mfma_f32_4x4x1_16b_f32 %0, %2, %6, %0 cbsz:4 abid:%10\n\t"
        "v_mfma_f32_4x4x1_16b_f32 %1, %3, %7, %1 cbsz:4 abid:%10\n\t"
        "s_nop 0\n\t"
        "v_mfma_f32_4x4x1_16b_f32 %0, %4, %8, %0 cbsz:4 abid:%10\n\t"
        "v_mfma_f32_4x4x1_16b_f32 %1, %5, %9, %1 cbsz:4 abid:%10\n\t"
        "s_nop 0"
        : "+v"(accA), "+v"(accB)
        : "v"(hA[0]), "v"(hA[1]), "v"(hA[2]), "v"(hA[3]), "a"(wk[4 * BB + 0]), "a"(wk[4 * BB + 1]), "a"(wk[4 * BB + 2]), "a"(wk[4 * BB + 3]), "n"(BB));
}
// mode 10: the same asm with the B operands in VGPRs (`v` constraint): the asm form's own cost
template <int BB>
__device__ __forceinline__ void block_vgpr(const float (&wk)[64], const f4 hA, f4& accA, f4& accB) {
    asm volatile(
        "v_mfma_f32_4x4x1_16b_f32 %0, %2, %6, %0 cbsz:4 abid:%10\n\t"
        "v_mfma_f32_4x4x1_16b_f32 %1, %3, %7, %1 cbsz:4 abid:%10\n\t"
        "s_nop 0\n\t"
        "v_mfma_f32_4x4x1_16b_f32 %0, %4, %8, %0 cbsz:4 abid:%10\n\t"
        "v_mfma_f32_4x4x1_16b_f32 %1, %5, %9, %1 cbsz:4 abid:%10\n\t"
        "s_nop 0"
        : "+v"(accA), "+v"(accB)
        : "v"(hA[0]), "v"(hA[1]), "v"(hA[2]), "v"(hA[3]), "v"(wk[4 * BB + 0]), "v"(wk[4 * BB + 1]), "v"(wk[4 * BB + 2]), "v"(wk[4 * BB + 3]), "n"(BB));
}
template <bool ACC>
__global__ __launch_bounds__(64) void bench_asm(float* out, long long* cyc, int niter, const float* __restrict__ wsrc) {
    const int l = threadIdx.x;
    float wreg[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) wreg[i] = wsrc[i * 64 + l];
    f4 h = f4{0.1f + 0.001f * l, 0.2f, 0.3f, 0.4f};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < niter; ++it) {
        f4 a0 = f4{0.f, 0.f, 0.f, 0.f}, a1 = a0;
#define BLK(B_) if constexpr (ACC) block_acc<B_>(wreg, h, a0, a1); else block_vgpr<B_>(wreg, h, a0, a1);
        BLK(0) BLK(1) BLK(2) BLK(3) BLK(4) BLK(5) BLK(6) BLK(7) BLK(8) BLK(9) BLK(10) BLK(11) BLK(12) BLK(13) BLK(14) BLK(15)
#undef BLK
        asm volatile("s_nop 3" : "+v"(a0), "+v"(a1));
        h = (a0 + a1) * 1e-3f;
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + l] = h[0] + h[1] + h[2] + h[3];
    if (blockIdx.x == 0 && l == 0) *cyc = t1 - t0;
}
template <bool ACC>
void run_asm(const char* name, int nwg, float* out, long long* cyc, const float* w) {
    const int niter = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    bench_asm<ACC><<<nwg, 64>>>(out, cyc, 10, w);
    hipEventRecord(e0);
    bench_asm<ACC><<<nwg, 64>>>(out, cyc, niter, w);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    printf("%-58s waves/SIMD=%d : %7.1f ns per layer (wall), s_memtime %.2f ticks\n", name, nwg / 1024, ms * 1e6 / niter, (double)c / niter);
}

template <int MODE>
__global__ __launch_bounds__(64) void bench(float* out, long long* cyc, int niter, const float* __restrict__ wsrc) {
    const int l = threadIdx.x;
    float wreg[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) wreg[i] = wsrc[i * 64 + l];
    f4 h = f4{0.1f + 0.001f * l, 0.2f, 0.3f, 0.4f};
    constexpr int NACC = MODE == 3 ? 4 : 2;
    unsigned sfill = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < niter; ++it) {
        f4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
        constexpr int FILL = MODE == 7 ? 1 : (MODE == 8 ? 2 : 0);      // modes 7 / 8: one / two SALU instructions behind every MFMA
        quarter<0, 0, NACC, MODE != 5, FILL>(wreg, h[0], acc, sfill);
        quarter<1, 0, NACC, MODE != 5, FILL>(wreg, h[1], acc, sfill);
        quarter<2, 0, NACC, MODE != 5, FILL>(wreg, h[2], acc, sfill);
        quarter<3, 0, NACC, MODE != 5, FILL>(wreg, h[3], acc, sfill);
        __builtin_amdgcn_sched_barrier(0);
        f4 s = acc[0] + acc[1];
        if constexpr (NACC == 4) s += acc[2] + acc[3];
        if constexpr (MODE == 6) s = elu_quad_scaled(s);
        else if constexpr (MODE == 4) s = elu_quad_clamp(s);
        else if constexpr (MODE >= 2 && MODE < 5) s = elu_quad(s);
        if constexpr (MODE >= 1 && MODE != 5 && MODE < 7) s = quad_transpose(s, l);
        h = s * 1e-3f;      // keep the chain finite (one packed multiply pair; K1 has the accumulator sum in its place)
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + l] = h[0] + h[1] + h[2] + h[3] + (float)sfill;
    if (blockIdx.x == 0 && l == 0) *cyc = t1 - t0;
}

// Today's decomposition with the SAME ELU, for a like-for-like figure: the H -> H layer of K1 at hidden 64 as the kernel runs it (4 waves
// per 16-trajectory tile, v_mfma_f32_16x16x4_f32, ds_write_b128 -> 3 own-quarter MFMAs -> s_barrier -> 3 ds_read_b128 -> 13 MFMAs,
// accumulator sum, ELU).  ELUF: 0 = round 3's (min + mul + exp + add + max), 1 = round 5's clamp form, 2 = the scaled-domain form.
__device__ __forceinline__ f4 mf16(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
template <int ELUF, bool XCH>
__global__ __launch_bounds__(256) void bench16(float* out, long long* cyc, int niter, const float* __restrict__ wsrc) {
    __shared__ f4 xb[2][4][64];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    float wm[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) wm[i] = wsrc[(16 * w + i) * 64 + l];
    f4 h = f4{0.1f + 0.001f * l, 0.2f, 0.3f, 0.4f};
    int p = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < niter; ++it) {
        if constexpr (XCH) xb[p][w][l] = h;
        f4 accA = f4{0.f, 0.f, 0.f, 0.f}, accB = accA;
        accA = mf16(wm[0], h[0], accA); accB = mf16(wm[1], h[1], accB); accA = mf16(wm[2], h[2], accA);
        asm volatile("" : "+v"(accA), "+v"(accB));
        __builtin_amdgcn_sched_barrier(0);
        f4 vq[4];
        if constexpr (XCH) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
#pragma unroll
            for (int c = 1; c < 4; ++c) vq[c] = xb[p][(w + c) & 3][l];
        } else {
#pragma unroll
            for (int c = 1; c < 4; ++c) vq[c] = h;
        }
        asm volatile("" : "+v"(accA), "+v"(accB));
        __builtin_amdgcn_sched_barrier(0);
        accB = mf16(wm[3], h[3], accB);
#pragma unroll
        for (int c = 1; c < 4; ++c) {
            accA = mf16(wm[4 * c + 0], vq[c][0], accA); accB = mf16(wm[4 * c + 1], vq[c][1], accB);
            accA = mf16(wm[4 * c + 2], vq[c][2], accA); accB = mf16(wm[4 * c + 3], vq[c][3], accB);
        }
        f4 s = accA + accB;
        s = ELUF == 0 ? elu_quad(s) : (ELUF == 1 ? elu_quad_clamp(s) : elu_quad_scaled(s));
        h = s * 1e-3f;
        p ^= 1;
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = h[0] + h[1] + h[2] + h[3];
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}
template <int ELUF, bool XCH>
void run16(const char* name, float* out, long long* cyc, const float* w) {
    const int niter = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    bench16<ELUF, XCH><<<256, 256>>>(out, cyc, 10, w);
    hipEventRecord(e0);
    bench16<ELUF, XCH><<<256, 256>>>(out, cyc, niter, w);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s 4-wave tile   : %7.1f ns per layer (wall)\n", name, ms * 1e6 / niter);
}

// functional check of the mapping on one wave: out[unit][traj] = sum_k W[unit][k] act[k][traj], then transposed back into A layout
__global__ void check(const float* __restrict__ W, const float* __restrict__ act, float* __restrict__ res) {
    const int l = threadIdx.x, b = l >> 2, t = l & 3;
    float wreg[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) wreg[k] = W[l * 64 + k];                 // B operand of k: lane (b',c') = W[unit 4b'+c' = lane][k]
    f4 h;
#pragma unroll
    for (int c = 0; c < 4; ++c) h[c] = act[(4 * b + c) * 4 + t];          // A layout: lane (b,t) register c = act[k = 4b+c][t]
    f4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    unsigned sf = 0;
    quarter<0, 0, 2>(wreg, h[0], acc, sf);
    quarter<1, 0, 2>(wreg, h[1], acc, sf);
    quarter<2, 0, 2>(wreg, h[2], acc, sf);
    quarter<3, 0, 2>(wreg, h[3], acc, sf);
    f4 s = acc[0] + acc[1];                                               // D: lane (b,c) register r = out[unit 4b+c][traj r]
    s = quad_transpose(s, l);                                             // A layout again: lane (b,t) register c = out[unit 4b+c][traj t]
#pragma unroll
    for (int c = 0; c < 4; ++c) res[(4 * b + c) * 4 + t] = s[c];
}

template <int MODE>
void run(const char* name, int nwg, float* out, long long* cyc, const float* w) {
    const int niter = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    bench<MODE><<<nwg, 64>>>(out, cyc, 10, w);
    hipEventRecord(e0);
    bench<MODE><<<nwg, 64>>>(out, cyc, niter, w);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    printf("%-58s waves/SIMD=%d : %7.1f ns per layer (wall), s_memtime %.2f ticks\n", name, nwg / 1024, ms * 1e6 / niter, (double)c / niter);
}

int main() {
    float *out, *w, *act, *res; long long* cyc;
    hipMalloc(&out, 2048 * 64 * sizeof(float)); hipMalloc(&cyc, 8);
    hipMalloc(&w, 64 * 64 * 4); hipMalloc(&act, 64 * 4 * 4); hipMalloc(&res, 64 * 4 * 4);
    float hw[64 * 64], ha[64 * 4], hr[64 * 4];
    for (int i = 0; i < 64 * 64; ++i) hw[i] = (float)((i * 7919) % 1000) * 1e-3f - 0.5f;
    for (int i = 0; i < 64 * 4; ++i) ha[i] = (float)((i * 104729) % 1000) * 1e-3f - 0.5f;
    hipMemcpy(w, hw, sizeof(hw), hipMemcpyHostToDevice);
    hipMemcpy(act, ha, sizeof(ha), hipMemcpyHostToDevice);
    check<<<1, 64>>>(w, act, res);
    hipMemcpy(hr, res, sizeof(hr), hipMemcpyDeviceToHost);
    double worst = 0;
    for (int u = 0; u < 64; ++u)
        for (int t = 0; t < 4; ++t) {
            double ref = 0;
            for (int k = 0; k < 64; ++k) ref += (double)hw[u * 64 + k] * ha[k * 4 + t];
            const double d = fabs(ref - hr[u * 4 + t]);
            if (d > worst) worst = d;
        }
    printf("mapping check (64 x 4x4x1 CBSZ=4/ABID=b + in-quad transpose vs a double-precision matvec): max abs diff %.3g %s\n", worst,
           worst < 1e-4 ? "OK" : "WRONG");
    run<0>("64 x 4x4x1 (2 chains), nothing else", 1024, out, cyc, w);
    run_asm<false>("64 x 4x4x1 as ONE asm block per 4 k, B in VGPRs", 1024, out, cyc, w);
    run_asm<true>("64 x 4x4x1 as ONE asm block per 4 k, B read from AccVGPRs", 1024, out, cyc, w);
    run<1>("64 x 4x4x1 + in-quad transpose", 1024, out, cyc, w);
    run<2>("64 x 4x4x1 + ELU + transpose (the whole layer)", 1024, out, cyc, w);
    run<3>("the whole layer, 4 accumulator chains", 1024, out, cyc, w);
    run<4>("the whole layer with the round-5 ELU (v_exp clamp)", 1024, out, cyc, w);
    run<6>("the whole layer with the scaled-domain ELU", 1024, out, cyc, w);
    run<5>("64 x 4x4x1 without the A broadcast (CBSZ = 0)", 1024, out, cyc, w);
    run<7>("64 x (4x4x1 + 1 SALU instruction), nothing else", 1024, out, cyc, w);
    run<8>("64 x (4x4x1 + 2 SALU instructions), nothing else", 1024, out, cyc, w);
    run<0>("64 x 4x4x1 (2 chains), nothing else", 2048, out, cyc, w);
    run<2>("64 x 4x4x1 + ELU + transpose (the whole layer)", 2048, out, cyc, w);
    run16<0, true>("16x16x4 layer as K1 runs it, round-3 ELU", out, cyc, w);
    run16<1, true>("16x16x4 layer as K1 runs it, clamp ELU", out, cyc, w);
    run16<2, true>("16x16x4 layer as K1 runs it, scaled-domain ELU", out, cyc, w);
    run16<1, false>("16x16x4 layer WITHOUT the exchange, clamp ELU", out, cyc, w);
    run16<2, false>("16x16x4 layer WITHOUT the exchange, scaled-domain ELU", out, cyc, w);
    unsigned* nd; hipMalloc(&nd, 4); hipMemset(nd, 0, 4);
    elu_forms<<<(1u << 24) / 256, 256>>>(nd);
    unsigned hnd; hipMemcpy(&hnd, nd, 4, hipMemcpyDeviceToHost);
    printf("ELU forms (min + exp2 vs exp2 with clamp) over 3 x 2^24 inputs of every exponent and sign: %u bitwise differences\n", hnd);
    return 0;
}
