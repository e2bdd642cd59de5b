// Device helpers shared by the one-wave-per-4-trajectories integrators K1x (psnode_mfma_x.hip, ODE) and K2x (psnode_mfma_xd.hip, DAE):
// the 4x4x1 MFMA wrappers, the in-quad transpose, the lane folds of the split-K output layer, one H -> H layer.  Layouts: psnode_mfma_x.hip.
#pragma once
#include <type_traits>

#include "psnode_pack.h"

namespace psnode {
namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// dim that MFMA m of L1 multiplies: register X01 (m < 4) or X23, ABID = 4 (m & 3)
__host__ __device__ constexpr int l1_dim(int m) { return 4 * ((m & 3) >> 1) + 2 * ((m & 3) & 1) + (m >= 4 ? 1 : 0); }

template <int ABID>
__device__ __forceinline__ f4 mfx(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, ABID, 0); }
__device__ __forceinline__ f4 mfn(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }

// 4 x 4 transpose of registers (r0..r3) x lanes (4q..4q+3): two butterfly stages of four v_cndmask_b32_dpp (D = vcc ? src1 : dpp(src0)).
// One asm block (the hazard recognizer does not look inside): s_nop 1 covers VALU write -> DPP read (2 wait states); inside, every DPP
// source was written at least two instructions earlier.
__device__ __forceinline__ f4 quad_transpose(const f4 v) {
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

// (a, b) -> sums over the four blocks of each row of 16 lanes, in every lane: x += row_ror:8 (x); x += row_ror:4 (x) as v_add_f32_dpp.
// One asm block (left to the compiler the rotation is a v_mov_b32_dpp into a zeroed register in front of a packed add: 4 instructions
// per rotation); it carries its own wait states (VALU write -> DPP read: 2).
__device__ __forceinline__ void row_sum2(float& a, float& b) {
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf"
        : "+v"(a), "+v"(b));
}
// (a, b) -> one register: lanes 0..31 hold a[l] + a[l + 32], lanes 32..63 hold b[l - 32] + b[l]
__device__ __forceinline__ float fold32(const float a, const float b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// (a, b) -> one register: even rows hold a[row] + a[row + 1], odd rows hold b[row - 1] + b[row]
__device__ __forceinline__ float fold16(const float a, const float b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// all 64 k of one H -> H layer: k = 4 bb + cc <-> A register cc, ABID bb
template <int BB>
__device__ __forceinline__ void hh_block(const float (&wk)[64], const f4 hA, f4& accA, f4& accB) {
    accA = mfx<BB>(hA[0], wk[4 * BB + 0], accA);
    accB = mfx<BB>(hA[1], wk[4 * BB + 1], accB);
    accA = mfx<BB>(hA[2], wk[4 * BB + 2], accA);
    accB = mfx<BB>(hA[3], wk[4 * BB + 3], accB);
}
// ELU of a D tile: the log2e-scaled domain of the inference kernels (SC) or the plain one (the training forwards: their saved rows are
// what the backward kernels read)
template <bool SC>
__device__ __forceinline__ f4 elu_x(const f4 v) {
    if constexpr (SC) return elu_quad_scaled(v);
    else return elu_quad(v);
}
template <bool SC = true>
__device__ __forceinline__ f4 hh_layer(const float (&wk)[64], const float bias, const f4 hA) {
    f4 accA = f4{bias, bias, bias, bias}, accB = f4{0.f, 0.f, 0.f, 0.f};
    hh_block<0>(wk, hA, accA, accB); hh_block<1>(wk, hA, accA, accB); hh_block<2>(wk, hA, accA, accB); hh_block<3>(wk, hA, accA, accB);
    hh_block<4>(wk, hA, accA, accB); hh_block<5>(wk, hA, accA, accB); hh_block<6>(wk, hA, accA, accB); hh_block<7>(wk, hA, accA, accB);
    hh_block<8>(wk, hA, accA, accB); hh_block<9>(wk, hA, accA, accB); hh_block<10>(wk, hA, accA, accB); hh_block<11>(wk, hA, accA, accB);
    hh_block<12>(wk, hA, accA, accB); hh_block<13>(wk, hA, accA, accB); hh_block<14>(wk, hA, accA, accB); hh_block<15>(wk, hA, accA, accB);
    return quad_transpose(elu_x<SC>(accA + accB));
}

// The same layer with the weights in AccVGPRs (K2x's AE head: 128 B operands that do not fit the 256 architectural VGPRs next to the DE's).
// Left to the register allocator the AGPR-resident weights come back through v_accvgpr_read in front of every MFMA and the scheduler
// serialises the two accumulator chains; here the B operand is READ from the AGPR (`a` constraint) and the order is fixed.  One asm block
// per 4 k; wait states inside (the hazard recognizer does not look): a dependent SrcC needs 2 (the other chain's MFMA + s_nop 0).
template <int BB>
__device__ __forceinline__ void hh_block_acc(const float (&wk)[64], const f4 hA, f4& accA, f4& accB) {
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
template <bool SC = true>
__device__ __forceinline__ f4 hh_layer_acc(const float (&wk)[64], const float bias, const f4 hA) {
    f4 accA = f4{bias, bias, bias, bias}, accB = f4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_nop 1" : "+v"(accA), "+v"(accB));           // VALU write -> MFMA SrcC read
    hh_block_acc<0>(wk, hA, accA, accB); hh_block_acc<1>(wk, hA, accA, accB); hh_block_acc<2>(wk, hA, accA, accB); hh_block_acc<3>(wk, hA, accA, accB);
    hh_block_acc<4>(wk, hA, accA, accB); hh_block_acc<5>(wk, hA, accA, accB); hh_block_acc<6>(wk, hA, accA, accB); hh_block_acc<7>(wk, hA, accA, accB);
    hh_block_acc<8>(wk, hA, accA, accB); hh_block_acc<9>(wk, hA, accA, accB); hh_block_acc<10>(wk, hA, accA, accB); hh_block_acc<11>(wk, hA, accA, accB);
    hh_block_acc<12>(wk, hA, accA, accB); hh_block_acc<13>(wk, hA, accA, accB); hh_block_acc<14>(wk, hA, accA, accB); hh_block_acc<15>(wk, hA, accA, accB);
    asm volatile("s_nop 3" : "+v"(accA), "+v"(accB));           // MFMA write (2 passes) -> VALU read
    return quad_transpose(elu_x<SC>(accA + accB));
}

template <int Q0, int NQ>
__device__ __forceinline__ f4 ext_mfmas(const float (&we)[16], const float eA, f4 acc) {
    if constexpr (NQ > 0) {
        acc = mfx<Q0>(eA, we[Q0], acc);
        return ext_mfmas<Q0 + 1, NQ - 1>(we, eA, acc);
    } else {
        return acc;
    }
}

constexpr int kXWaves = 4;      // independent waves per workgroup (one per SIMD)

// K1x / K2x address a row as <uniform 64-bit row base> + <32-bit per-lane BYTE offset>: (trajectory * batch stride + column) * 4 must fit
// 32 bits and the batch stride must not be negative (a B-major dataset view of >= 4 GiB, or a negative stride from a C
// caller, would wrap silently).  Views that do not qualify stay on K1 / K2 (64-bit indexing); a forced _WAVE call returns UNSUPPORTED.
inline bool span32_ok(const long long B, const long long sb, const long long cols) {
    if (sb < 0 || B < 1) return false;
    const unsigned long long last = (unsigned long long)(B - 1) * (unsigned long long)sb + (unsigned long long)cols;
    return last < (1ull << 30);                                  // elements: x 4 bytes < 2^32
}

}  // namespace
}  // namespace psnode
