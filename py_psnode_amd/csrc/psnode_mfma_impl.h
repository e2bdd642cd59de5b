// MFMA fused integrators (K1 = ODE, K2 = DAE) for the reference's default right-hand sides:
//   DE: in -> H -> H -> H -> x_dim   and, for the DAE,   AE: in -> H -> H -> H -> i_dim   (ELU between layers),
//   H = 16 * NWV in {32, 64, 128}: the scripts' --hidden knob (64 under their flg_debug override, 128 the argparse
//   default, neural_00_ODE_01_no_encode.py:259,276).  The text below is written for H = 64 (NWV = 4 waves).
//
// Mapping (DESIGN.md "K1/K2"):  one workgroup = NWV waves = one tile of 16 trajectories, walked through ALL T-1 steps.
//   D[unit][traj] = W[unit][k] * act[k][traj]   on v_mfma_f32_16x16x4_f32 (exact fp32, 256 flop/clk/CU):
//   A operand = weights (lane l: unit l&15 of the wave's 16-unit slice, k-slot l>>4)  -- resident in VGPRs for the
//               whole launch, so the weights are read from HBM/L2 once per workgroup;
//   B operand = activations (lane l: k-slot g = l>>4, trajectory l&15);
//   D         = lane l holds units 4*g+r (r = 0..3) of trajectory l&15.
// Layer plan per MLP evaluation (4 waves, w = wave id):
//   L1  in->64   split-N: wave w makes hidden units 16w..16w+15.  The input cat(a0, s-a0, s) (DE) / cat(a0, x, z, v)
//                (AE) is never materialised: the a0 columns (+bias) are a per-trajectory constant computed once, the
//                external-input columns once per step (zero-order hold), only the x-dependent MFMAs run per stage --
//                same products, different summation order.
//   L2,L3 64->64 split-N: ELU -> every wave publishes its 4 values/lane with ONE lane-linear ds_write_b128, one
//                s_barrier, three ds_read_b128.  The k-order is permuted (k = 16w' + 4g + r) so that lane (g, traj)
//                needs exactly what lane (g, traj) of the other waves holds: no shuffles, no bank conflicts.
//                The wave's own quarter of K is multiplied before the barrier.
//   L4  64->out  split-K: wave w multiplies ITS OWN 16 hidden units (no exchange between L3 and L4), partial sums are
//                all-reduced through LDS in a fixed order so every wave holds the identical result.
//   => 3 exchanges per evaluation, 40 MFMAs per wave per DE evaluation.
// Register-resident state, replicated in the four waves: x-dim d = 4r+g sits in lane group g, register r (L4's output
// rows feed L1's B operands directly).  External-input "slot" q = 4m+g of the per-step MFMA m carries ext[q]-a0 for
// q < ne and ext[q-ne] for ne <= q < 2ne (ext = z | v | i).  The AE's output rows are laid out so that the algebraic
// variable an ext slot needs appears in that very lane, row m: the DAE feedback i -> DE input needs no data movement.
// External inputs are prefetched one step ahead straight from the caller's strided (B-major) memory.
// This header is the kernel template; one translation unit per hidden width instantiates it (psnode_mfma.hip = 64,
// psnode_mfma_h32.hip, psnode_mfma_h128.hip) so the widths compile in parallel.
#pragma once
#include <type_traits>

#include "psnode_pack.h"

namespace psnode {
namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void pack_mfma_kernel(const PackMfma p) {
    const int R = pack_fwd_count(p);
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < p.nw * R * 64; idx += gridDim.x * blockDim.x)
        p.out[idx] = pack_fwd_value(p, (idx >> 6) / R, (idx >> 6) % R, idx & 63);
}

// Stream image of the H->H layers for the widths whose weights no CU can hold (hidden 129..256: 12 / 16 waves per tile):
// [layer 2|3][wave w][chunk c][lane] f4 -- component r is forward-image register W2|W3 + 4c + r -- so that chunk c of a wave is ONE
// coalesced 1 KB global_load_dwordx4 and a wave's layer is one contiguous 16 KB run: scalar base + one lane offset + immediates
// (the image stays in L2: 0.5 MB per MLP at hidden 256).
__global__ void pack_stream_kernel(const PackMfma p, f4* __restrict__ out) {
    const int W2 = p.NX + p.NB + p.NE + 4, W3 = W2 + 4 * p.nw + 4;
    const int n = 2 * p.nw * p.nw * 64;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
        const int lane = idx & 63, c = (idx >> 6) % p.nw, w = ((idx >> 6) / p.nw) % p.nw, layer = (idx >> 6) / (p.nw * p.nw);
        f4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = pack_fwd_value(p, w, (layer ? W3 : W2) + 4 * c + r, lane);
        out[idx] = v;
    }
}
__host__ __device__ constexpr size_t stream_image_floats(int nwv) { return nwv > 8 ? (size_t)2 * nwv * nwv * 64 * 4 : 0; }

__device__ __forceinline__ f4 mfma4(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// SC: the log2e-scaled domain of the inference forwards (PackMfma::scaled); the training forwards (SAVE) store ELU outputs for the
// backward kernels and stay in the plain domain
#ifndef PSNODE_SCALED_ELU
#define PSNODE_SCALED_ELU 1
#endif
template <bool SC>
__device__ __forceinline__ f4 elu4(f4 v) { if constexpr (SC && PSNODE_SCALED_ELU) return elu_quad_scaled(v); else return elu_quad(v); }

// layers 2..4 of one MLP, in registers (NWV = waves per tile = hidden / 16)
template <int NWV>
struct Tail {
    float w2[4 * NWV], w3[4 * NWV], w4[4];
    f4 b2, b3, b4;
    int m;             // NWV > 8: 0 = DE, 1 = AE -- which set of the LDS-parked layer constants (tw4 / tb in the kernel) is this MLP's
    const float* gw;   // NWV > 8: this WAVE's run of the stream image (pack_stream_kernel; wave-uniform), layer 2 at +0, layer 3 at + NWV * NWV * 256 floats
};

__host__ __device__ constexpr bool ae_weights_in_lds(bool dae, int nwv) { return dae && nwv == 8; }
__host__ __device__ constexpr bool weights_streamed(int nwv) { return nwv > 8; }
// Which shape classes the streamed widths carry: the ones that compile WITHOUT spills at 168 (12 waves) / 128 (16 waves) VGPRs per lane
// (a spill in a kernel with divergent regions is what round 4's defect (a) was; the ISA lint gates the build on it).  16 waves: the ODE
// with x_dim <= 8; 12 waves: the ODE at every x_dim, the DAE with 2 (z+v+i) <= 12.  Everything else at these widths takes K0.
__host__ __device__ constexpr bool streamed_class_ok(int nwv, bool dae, bool wide_x, int nzm) {
    return nwv <= 8 || (nwv == 12 ? (!dae || nzm <= 3) : (!dae && !wide_x));
}
__host__ __device__ constexpr size_t ae_lds_bytes(int nwv) { return (size_t)2 * nwv * nwv * 64 * sizeof(f4); }

template <int NX, int NB, int NE, int NWV, bool MID = true, bool SMALL = true>
__device__ __forceinline__ void load_tail(const float* pw, Tail<NWV>& t) {
    using R = Regs<NX, NB, NE, NWV>;
    if constexpr (MID) {
#pragma unroll
        for (int k = 0; k < 4 * NWV; ++k) { t.w2[k] = pw[(R::W2 + k) * 64]; t.w3[k] = pw[(R::W3 + k) * 64]; }
    }
    if constexpr (SMALL) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            t.w4[r] = pw[(R::W4 + r) * 64];
            t.b2[r] = pw[(R::B2 + r) * 64]; t.b3[r] = pw[(R::B3 + r) * 64]; t.b4[r] = pw[(R::B4 + r) * 64];
        }
    }
}

template <int N> struct Arr { float v[N > 0 ? N : 1]; };

template <int METHOD, int NX, int NZM, int NZA, bool DAE, int NWV, bool SAVE = false>
__global__ __launch_bounds__(64 * NWV) void integrate_mfma_kernel(const IntegrateDev a, const float* __restrict__ pack_de,
                                                                  const float* __restrict__ pack_ae, const int NA) {
    using RD = Regs<NX, 0, NZM, NWV>;     // folded DE image: no `s - a0` registers for the x dims (psnode_pack.h)
    using RA = Regs<NX, 0, NZA, NWV>;
    // Exchange buffers: three, used in a FIXED rotation by the three exchanges of an MLP evaluation (L2's gather -> 0, L3's gather -> 1,
    // L4's all-reduce -> 2), so every LDS address of the time loop is a loop-invariant immediate.  A buffer is rewritten two barriers
    // after its last read.  With two buffers and a running parity bit the Euler step (3 exchanges: odd) flipped the parity from step to
    // step and paid ~10 address instructions per step for it (RK4 and Midpoint happen to have an even count).
    __shared__ f4 xbuf[3][NWV][64];
    // 8 waves of 64 lanes leave 256 VGPRs per lane: the DAE's second weight set does not fit next to the DE's, so the
    // AE's H->H weights live in (dynamic) LDS, each lane reading back exactly the A-operand values it wrote.
    constexpr bool AE_LDS = ae_weights_in_lds(DAE, NWV);
    extern __shared__ f4 aew[];   // AE_LDS: [layer 2|3][chunk][wave][lane]
    // 12 / 16 waves (hidden 129..256): 3 / 4 waves per SIMD, 168 / 128 VGPRs per lane, and one H->H matrix is 144 / 256 KB -- neither
    // registers nor LDS hold the weights.  Every H->H layer streams its A operands from the L2-resident stream image (one coalesced
    // dwordx4 per chunk); the other waves of the SIMD cover the latency.
    constexpr bool W_GLB = weights_streamed(NWV);
    constexpr int MODE_DE = W_GLB ? 2 : 0, MODE_AE = W_GLB ? 2 : (AE_LDS ? 1 : 0);
    // ... and at 128 / 168 VGPRs per lane the 16 registers per MLP of layer constants (b2, b3, b4, this wave's L4 slice) are the
    // difference between spilling and not: they are parked in LDS and read where a layer starts (one ds_read_b128 each)
    __shared__ f4 tw4[W_GLB ? (DAE ? 2 : 1) : 1][W_GLB ? NWV : 1][W_GLB ? 64 : 1];
    __shared__ f4 tb[W_GLB ? (DAE ? 2 : 1) : 1][3][W_GLB ? NWV : 1][4];

    const int l = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = l >> 4, j = l & 15;
    auto park_tail = [&](const float* pwl, const int m, const int w4r, const int b2r, const int b3r, const int b4r) {
        f4 q;
#pragma unroll
        for (int r = 0; r < 4; ++r) q[r] = pwl[(w4r + r) * 64];
        tw4[m][w][l] = q;
        const int regs[3] = {b2r, b3r, b4r};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) q[r] = pwl[(regs[i] + r) * 64];
            if ((l & 15) == 0) tb[m][i][w][l >> 4] = q;
        }
    };
    const long long b0 = (long long)blockIdx.x * TBM;
    const bool valid = b0 + j < a.B;
    const long long b = valid ? b0 + j : a.B - 1;
    const int xd = a.xd, zd = a.zd, vd = DAE ? a.vd : 0, idim = DAE ? a.id : 0;
    const int nzv = zd + vd, ne = nzv + idim, n = xd + ne;
    const bool true_i = DAE && (a.flags & PSNODE_FLAG_INPUT_TRUE_I) != 0;
    // teacher forcing of x is a RUNTIME flag since round 5 (it was a template parameter: a second instance of every inference kernel,
    // ~190 of the library's 1100): a uniform branch around two loads that are waited for inside it, as the teacher-forced i always was
    const bool true_x = !SAVE && (a.flags & PSNODE_FLAG_INPUT_TRUE_X) != 0;

    // ---- weights -> registers (once per launch)
    const float* pw = pack_de + (size_t)w * (RD::COUNT + NA) * 64 + l;
    float w1xs[NX];
    Arr<NZM> w1z;
    f4 b1r;
    Tail<NWV> de;
#pragma unroll
    for (int r = 0; r < NX; ++r) w1xs[r] = pw[(RD::W1A + r) * 64];
#pragma unroll
    for (int m = 0; m < NZM; ++m) w1z.v[m] = pw[(RD::W1E + m) * 64];
#pragma unroll
    for (int r = 0; r < 4; ++r) b1r[r] = pw[(RD::B1 + r) * 64];
    load_tail<NX, 0, NZM, NWV, !W_GLB, !W_GLB>(pw, de);
    de.m = 0;
    if constexpr (W_GLB) park_tail(pw, 0, RD::W4, RD::B2, RD::B3, RD::B4);
    de.gw = W_GLB ? pack_de + (size_t)NWV * (RD::COUNT + NA) * 64 + (size_t)w * NWV * 256 : nullptr;

    const float* pwa = pack_ae + (size_t)w * (RA::COUNT + NA) * 64 + l;
    float aw1x[NX];
    Arr<NZA> aw1e;
    f4 ab1r = f4{0.f, 0.f, 0.f, 0.f};
    Tail<NWV> ae;
    if constexpr (DAE) {
#pragma unroll
        for (int r = 0; r < NX; ++r) aw1x[r] = pwa[(RA::W1A + r) * 64];
#pragma unroll
        for (int m = 0; m < NZA; ++m) aw1e.v[m] = pwa[(RA::W1E + m) * 64];
#pragma unroll
        for (int r = 0; r < 4; ++r) ab1r[r] = pwa[(RA::B1 + r) * 64];
        load_tail<NX, 0, NZA, NWV, !AE_LDS && !W_GLB, !W_GLB>(pwa, ae);
        ae.m = 1;
        if constexpr (W_GLB) park_tail(pwa, 1, RA::W4, RA::B2, RA::B3, RA::B4);
        ae.gw = W_GLB ? pack_ae + (size_t)NWV * (RA::COUNT + NA) * 64 + (size_t)w * NWV * 256 : nullptr;
        if constexpr (AE_LDS) {
#pragma unroll
            for (int c = 0; c < NWV; ++c) {
                f4 q2, q3;
#pragma unroll
                for (int r = 0; r < 4; ++r) { q2[r] = pwa[(RA::W2 + 4 * c + r) * 64]; q3[r] = pwa[(RA::W3 + 4 * c + r) * 64]; }
                aew[((0 * NWV + c) * NWV + w) * 64 + l] = q2;
                aew[((1 * NWV + c) * NWV + w) * 64 + l] = q3;
            }
        }
    }

    // ---- per-trajectory constants
    float x[NX], a0x[NX];
#pragma unroll
    for (int r = 0; r < NX; ++r) {
        const int d = 4 * r + g;
        a0x[r] = d < xd ? a.a0[b * n + d] : 0.0f;
        x[r] = d < xd ? (DAE ? a.x_init[b * xd + d] : a.x.p[b * a.x.sb + d]) : 0.0f;
    }
    // DE ext slots of this lane: kind 0 = z column, 1 = v column, 2 = algebraic variable (register), 3 = padding
    int ekind[NZM > 0 ? NZM : 1], ecol[NZM > 0 ? NZM : 1];
    Arr<NZM> a0e;
#pragma unroll
    for (int m = 0; m < NZM; ++m) {
        const int q = 4 * m + g, e = slot_ext(q, ne);
        ekind[m] = e < 0 ? 3 : (e < zd ? 0 : (e < nzv ? 1 : 2));
        ecol[m] = e < 0 ? 0 : (e < zd ? e : (e < nzv ? e - zd : e - nzv));
        a0e.v[m] = q < ne ? a.a0[b * n + xd + q] : 0.0f;
    }
    // AE ext slots (z | v at the right grid point, never jumped except in the event-time recompute)
    int akind[NZA > 0 ? NZA : 1], acol[NZA > 0 ? NZA : 1];
#pragma unroll
    for (int m = 0; m < NZA; ++m) {
        const int q = 4 * m + g;
        akind[m] = q < zd ? 0 : (q < nzv ? 1 : 3);
        acol[m] = q < zd ? q : (q < nzv ? q - zd : 0);
    }
    f4 c0 = b1r, c0a = ab1r;   // bias + W1[:, a0 columns] . a0 : constant for the whole launch
    for (int m = 0; m < NA; ++m) {
        const int q = 4 * m + g;
        const float av = q < n ? a.a0[b * n + q] : 0.0f;
        c0 = mfma4(pw[(RD::COUNT + m) * 64], av, c0);
        if constexpr (DAE) c0a = mfma4(pwa[(RA::COUNT + m) * 64], av, c0a);
    }

    const long long tst = a.t.st, nT = a.T;
    const float* tp = a.t.p + b * a.t.sb;
    // Every per-step input load is UNCONDITIONAL and its row base is computed once per step: a source that does not exist (z_dim = 0,
    // v_dim = 0, no jump array) points at this trajectory's clock with stride 0 -- nothing selects its value -- instead of hiding
    // behind a runtime branch.  Branches around loads cost twice in the time loop: eight taken scalar branches per DAE step, and at
    // every join the compiler's wait-count pass can only assume the worst (s_waitcnt vmcnt(0) -- a full memory round trip on the
    // prefetches just issued).  The DAE's teacher-forced algebraic input was such a branch: ~800 cycles per step of K2 for a load that
    // the no-teacher-forcing call never issues (profiles/r03a_dae01_euler_pmc_sq.txt: SQ_WAIT_ANY 2.4 k cycles per step, 6 exchanges).
    const bool has_z = zd > 0, has_v = DAE && vd > 0;
    const long long zst = has_z ? a.z.st : 0, zje = (has_z && a.zj) ? a.zje : 0;
    const long long vst = has_v ? a.v.st : 0, vje = (has_v && a.vj) ? a.vje : 0;
    const float* zp = has_z ? a.z.p + b * a.z.sb : tp;
    const float* zjp = (has_z && a.zj) ? a.zj + b * a.zjb : tp;
    const float* vp = has_v ? a.v.p + b * a.v.sb : tp;
    const float* vjp = (has_v && a.vj) ? a.vj + b * a.vjb : tp;
    // column a slot reads in the row of ITS source (z row or v row; algebraic / padding slots read column 0 of the z row: the value is
    // replaced by i, or meets a zero weight)
    int esc[NZM > 0 ? NZM : 1], asc[NZA > 0 ? NZA : 1];
#pragma unroll
    for (int m = 0; m < NZM; ++m) esc[m] = ((has_z && ekind[m] == 0) || (has_v && ekind[m] == 1)) ? ecol[m] : 0;
#pragma unroll
    for (int m = 0; m < NZA; ++m) asc[m] = ((has_z && akind[m] == 0) || (has_v && akind[m] == 1)) ? acol[m] : 0;

    // z|v columns of grid point k for the ext slots (ev >= 0: the batch takes the jump values for this step).  In the DAE a lane's
    // slot may be a z or a v column: the lane picks the ROW POINTER of its source (two v_cndmask on the step's row pointers) and issues
    // ONE load -- reading both sources and selecting the value put an s_waitcnt vmcnt(0) behind the prefetch (rounds 1-2 deferred that
    // select by a step at the price of a second register per slot), and selecting between the loop-invariant BASE pointers makes the
    // compiler spill a pointer table to scratch.  `ev` is wave-uniform (read with v_readlane from a 64-step block of the event table,
    // as K3f): the row offset is scalar arithmetic; a per-lane copy of the index cost 64-bit per-lane multiplies per load.
    struct RowPtr { const float* z; const float* v; };
    auto rows_at = [&](long long k, int ev) -> RowPtr {
        RowPtr r;
        r.z = ev >= 0 ? zjp + (long long)ev * zje : zp + k * zst;
        r.v = DAE ? (ev >= 0 ? vjp + (long long)ev * vje : vp + k * vst) : r.z;
        return r;
    };
    auto pick = [&](int, float val, float) -> float { return val; };     // the load already came from the slot's own source
    auto load_de_rows = [&](const RowPtr rp, Arr<NZM>& dz) {
#pragma unroll
        for (int m = 0; m < NZM; ++m) dz.v[m] = ((DAE && ekind[m] == 1) ? rp.v : rp.z)[esc[m]];
    };
    auto load_ae_rows = [&](const RowPtr rp, Arr<NZA>& dz) {
#pragma unroll
        for (int m = 0; m < NZA; ++m) dz.v[m] = (akind[m] == 1 ? rp.v : rp.z)[asc[m]];
    };
    auto load_de_raw = [&](long long k, int ev, Arr<NZM>& dz, Arr<NZM>&) { load_de_rows(rows_at(k, ev), dz); };
    auto load_ae_raw = [&](long long k, int ev, Arr<NZA>& dz, Arr<NZA>&) { load_ae_rows(rows_at(k, ev), dz); };
    auto pick_ae = [&](const Arr<NZA>& dz, const Arr<NZA>& dv) -> Arr<NZA> {
        Arr<NZA> o = {};
#pragma unroll
        for (int m = 0; m < NZA; ++m) o.v[m] = pick(akind[m], dz.v[m], dv.v[m]);
        return o;
    };
    // Event indices travel 64 steps at a time: lane i holds event_idx[64 blk + i] (one 256-byte load per 64 steps, waited for on the spot)
    auto load_evb = [&](const long long blk) -> int {
        const long long i = blk * 64 + l;
        const int v = (a.ev && i + 1 < a.T) ? a.ev[i] : -1;
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
        return v;
    };

    int p = 0;   // exchange buffer of the next H->H layer (0, 1, 0, 1, ...: two per evaluation); the output all-reduce uses buffer 2
    // 4 waves per tile = one per SIMD: nothing else hides the exchange latency, so all reads are put in flight at once (12 VGPRs).
    // 8 waves (hidden 128) have a second wave per SIMD to hide it and no registers to spare.
    constexpr bool PREFETCH_ALL = NWV <= 4;

    // one H->H layer: publish own activations, multiply own K slice, barrier, multiply the other NWV-1 slices
    auto mid = [&](const float (&wm)[4 * NWV], const f4 bias, const f4 h) -> f4 {
        xbuf[p][w][l] = h;
        f4 accA = bias, accB = f4{0.f, 0.f, 0.f, 0.f};
        // PSNODE_MID_PRE of the wave's own K-quarter MFMAs are PINNED in front of the barrier (they cover the LDS write), the rest go
        // behind the reads (they cover the read latency).  Where an MFMA ends up relative to s_barrier is otherwise decided before the
        // machine scheduler runs (the intrinsic is pure, the barrier only orders memory): sched_barrier does not hold it.  Same-box A/B of the split
        // (profiles/r03i_mid_pre_ab.txt, r03j_mid_pre_ab.txt; K1 RK4 ms): 0 in front 4.04, 1: 3.88, 2: 3.83, 3: 3.77, 4: 3.90 -- 3 / 1 it is.  Round 3 found
        // 1 in front / 3 behind in the ISA of a source that says 4 in front; the empty asm statements make the split explicit.
#ifndef PSNODE_MID_PRE
#define PSNODE_MID_PRE 3
#endif
        constexpr int PRE = PSNODE_MID_PRE;
        if constexpr (PRE >= 1) accA = mfma4(wm[0], h[0], accA);
        if constexpr (PRE >= 2) accB = mfma4(wm[1], h[1], accB);
        if constexpr (PRE >= 3) accA = mfma4(wm[2], h[2], accA);
        if constexpr (PRE >= 4) accB = mfma4(wm[3], h[3], accB);
        if constexpr (PRE >= 1) asm volatile("" : "+v"(accA), "+v"(accB));
        __builtin_amdgcn_sched_barrier(0);
        lds_barrier();
        f4 vq[NWV];
        if constexpr (PREFETCH_ALL) {
#pragma unroll
            for (int c = 1; c < NWV; ++c) vq[c] = xbuf[p][wrap_wave(w + c, NWV)][l];
            if constexpr (PRE < 4) asm volatile("" : "+v"(accA), "+v"(accB));      // the remaining own-quarter MFMAs stay behind the read issue
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (PRE < 1) accA = mfma4(wm[0], h[0], accA);
        if constexpr (PRE < 2) accB = mfma4(wm[1], h[1], accB);
        if constexpr (PRE < 3) accA = mfma4(wm[2], h[2], accA);
        if constexpr (PRE < 4) accB = mfma4(wm[3], h[3], accB);
#pragma unroll
        for (int c = 1; c < NWV; ++c) {
            const f4 v = PREFETCH_ALL ? vq[c] : xbuf[p][wrap_wave(w + c, NWV)][l];
            accA = mfma4(wm[4 * c + 0], v[0], accA);
            accB = mfma4(wm[4 * c + 1], v[1], accB);
            accA = mfma4(wm[4 * c + 2], v[2], accA);
            accB = mfma4(wm[4 * c + 3], v[3], accB);
        }
        p ^= 1;
        return elu4<!SAVE>(accA + accB);
    };
    // the same layer with the weights of `layer` (0: L2, 1: L3) read from aew
    auto mid_lds = [&](const int layer, const f4 bias, const f4 h) -> f4 {
        xbuf[p][w][l] = h;
        const f4* wl = aew + ((size_t)layer * NWV * NWV + w) * 64 + l;
        f4 wq = wl[0];
        f4 accA = bias, accB = f4{0.f, 0.f, 0.f, 0.f};
        accA = mfma4(wq[0], h[0], accA);
        accB = mfma4(wq[1], h[1], accB);
        accA = mfma4(wq[2], h[2], accA);
        accB = mfma4(wq[3], h[3], accB);
        __builtin_amdgcn_sched_barrier(0);
        lds_barrier();
#pragma unroll
        for (int c = 1; c < NWV; ++c) {
            const f4 v = xbuf[p][wrap_wave(w + c, NWV)][l];
            wq = wl[c * NWV * 64];
            accA = mfma4(wq[0], v[0], accA);
            accB = mfma4(wq[1], v[1], accB);
            accA = mfma4(wq[2], v[2], accA);
            accB = mfma4(wq[3], v[3], accB);
        }
        p ^= 1;
        return elu4<!SAVE>(accA + accB);
    };
    // the same layer with the A operands streamed from the stream image (`wl` = this lane's f4 of chunk 0 of the layer): all NWV chunk
    // loads are requested first -- they travel while the tile publishes / gathers the activations
    auto mid_glb = [&](const float* __restrict__ lay, const f4 bias, const f4 h) -> f4 {
        // `lay`: the wave's run of this layer (uniform).  The lane offset goes through an empty asm: these loads are loop-invariant to
        // the compiler, and hoisted out of the time loop they would be the register-resident weights that do not fit (first build:
        // 330 spilled VGPRs).  Laundering the POINTER instead made every load a flat_load behind 64-bit per-lane address arithmetic.
        int loff = 4 * l;
        asm volatile("" : "+v"(loff));
        const f4* wl = reinterpret_cast<const f4*>(lay + loff);
        // a ring of WD chunk slots: chunk c + WD is requested into the slot chunk c just left, the gathered activations of chunk c + 1
        // are read while chunk c multiplies -- the order is pinned per chunk (left alone the scheduler hoists every read and every
        // load to the top of the layer and spills ~100 VGPRs at 16 waves)
#ifndef PSNODE_STREAM_DEPTH
#define PSNODE_STREAM_DEPTH 4
#endif
        constexpr int WD = PSNODE_STREAM_DEPTH < NWV ? PSNODE_STREAM_DEPTH : NWV;
        f4 wq[WD];
#pragma unroll
        for (int c = 0; c < WD; ++c) wq[c] = wl[c * 64];
        xbuf[p][w][l] = h;
        f4 accA = bias, accB = f4{0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_sched_barrier(0);
        lds_barrier();
        f4 vn = xbuf[p][wrap_wave(w + 1, NWV)][l];
#pragma unroll
        for (int c = 0; c < NWV; ++c) {
            const f4 v = c == 0 ? h : vn;
            if (c >= 1 && c + 1 < NWV) vn = xbuf[p][wrap_wave(w + c + 1, NWV)][l];
            const f4 wc = wq[c % WD];
            accA = mfma4(wc[0], v[0], accA);
            accB = mfma4(wc[1], v[1], accB);
            accA = mfma4(wc[2], v[2], accA);
            accB = mfma4(wc[3], v[3], accB);
            if (c + WD < NWV) wq[c % WD] = wl[(c + WD) * 64];
            __builtin_amdgcn_sched_barrier(0);
        }
        p ^= 1;
        return elu4<!SAVE>(accA + accB);
    };
    // layers 2..4 from the L1 pre-activation; every wave returns the identical output rows.
    // ROWS2: only rows r < 2 of the output carry data (the DE with x_dim <= 8): all-reduce 8 bytes per lane instead of 16.
    // `wmode`: where the H->H weights are -- 0 registers, 1 LDS (the AE at 8 waves), 2 the stream image (more than 8 waves)
    auto tail = [&](const f4 pre1, const Tail<NWV>& t, auto rows2, auto wmode, auto&& keep) -> f4 {
        constexpr bool ROWS2 = decltype(rows2)::value;
        f4 h = elu4<!SAVE>(pre1);
        keep(0, h);
        if constexpr (decltype(wmode)::value == 1) {
            h = mid_lds(0, t.b2, h);
            keep(1, h);
            h = mid_lds(1, t.b3, h);
            keep(2, h);
        } else if constexpr (decltype(wmode)::value == 2) {
            h = mid_glb(t.gw, tb[t.m][0][w][g], h);
            keep(1, h);
            h = mid_glb(t.gw + (size_t)NWV * NWV * 256, tb[t.m][1][w][g], h);
            keep(2, h);
        } else {
            h = mid(t.w2, t.b2, h);
            keep(1, h);
            h = mid(t.w3, t.b3, h);
            keep(2, h);
        }
        f4 w4v, out;
        if constexpr (decltype(wmode)::value == 2) { w4v = tw4[t.m][w][l]; out = tb[t.m][2][w][g]; }
        else { w4v = f4{t.w4[0], t.w4[1], t.w4[2], t.w4[3]}; out = t.b4; }
        f4 accA = mfma4(w4v[0], h[0], f4{0.f, 0.f, 0.f, 0.f});
        f4 accB = mfma4(w4v[1], h[1], f4{0.f, 0.f, 0.f, 0.f});
        accA = mfma4(w4v[2], h[2], accA);
        accB = mfma4(w4v[3], h[3], accB);
        const f4 part = accA + accB;
        if constexpr (ROWS2) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2* xb2 = reinterpret_cast<f2*>(&xbuf[2][0][0]);
            xb2[w * 64 + l] = f2{part[0], part[1]};
            lds_barrier();
#pragma unroll
            for (int c = 0; c < NWV; ++c) {
                const f2 q = xb2[c * 64 + l]; out[0] += q[0]; out[1] += q[1];
                if constexpr (W_GLB) { if ((c & 3) == 3) __builtin_amdgcn_sched_barrier(0); }    // four reads in flight, not NWV (registers)
            }
        } else {
            xbuf[2][w][l] = part;
            lds_barrier();
#pragma unroll
            for (int c = 0; c < NWV; ++c) {
                out += xbuf[2][c][l];
                if constexpr (W_GLB) { if ((c & 3) == 3) __builtin_amdgcn_sched_barrier(0); }
            }
        }
        return out;
    };
    // SAVE: running row pointers of the saved activations [T-1,S,3,B,H] (this lane's four units) and stage inputs [T-1,S,B,xd]
    const size_t sa_layer = SAVE ? (size_t)a.B * (16 * NWV) : 0;
    const long long sx_step = SAVE ? a.B * xd : 0;
    float* sa_run = SAVE ? a.sact + (size_t)b * (16 * NWV) + 16 * w + 4 * g : nullptr;
    float* sx_run = SAVE ? a.sxst + b * xd : nullptr;
    // DE right-hand side at xs with this step's constant part cz
    auto rhs = [&](const float (&xs)[NX], const f4 cz) -> f4 {
        // folded L1: (Ws + Wd) . xs, the -Wd . a0x term lives in c0 (two accumulator chains for NX = 2)
        f4 accA = cz, accB = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < NX; ++r) {
            if (r & 1) accB = mfma4(w1xs[r], xs[r], accB);
            else accA = mfma4(w1xs[r], xs[r], accA);
        }
        if constexpr (SAVE) {
            // training forward: what autograd would save.  This wave's 16 units of the three ELU layers (16 bytes per lane and layer)
            // and, from wave 0, the stage input -- rows (step, stage) of a.sact / a.sxst; the row pointers advance by one stage per call
            if (w == 0 && valid) {
#pragma unroll
                for (int r = 0; r < NX; ++r) if (4 * r + g < xd) sx_run[4 * r + g] = xs[r];
            }
            sx_run += sx_step;
            const f4 out = tail(NX > 1 ? accA + accB : accA, de, std::integral_constant<bool, (NX <= 2)>{}, std::integral_constant<int, MODE_DE>{},
                                [&](const int q, const f4 hq) { if (valid) store_nt<(NWV >= 8)>(reinterpret_cast<f4*>(sa_run + (size_t)q * sa_layer), hq); });
            sa_run += 3 * sa_layer;
            return out;
        }
        return tail(NX > 1 ? accA + accB : accA, de, std::integral_constant<bool, (NX <= 2)>{}, std::integral_constant<int, MODE_DE>{}, [](int, f4) {});
    };
    // AE head g(xa; zv): rows (g, m) of the result carry the i-dim that DE ext slot (m, g) consumes
    // SAVE: `rows` = this lane's four units in layer 0 of the head's saved activations, `lstride` floats to the next layer (null: not saved)
    auto ae_eval = [&](const float (&xa)[NX], const Arr<NZA>& zv, float* rows, const size_t lstride) -> f4 {
        f4 acc = c0a;
        if constexpr (DAE) {
#pragma unroll
            for (int r = 0; r < NX; ++r) acc = mfma4(aw1x[r], xa[r], acc);
#pragma unroll
            for (int m = 0; m < NZA; ++m) acc = mfma4(aw1e.v[m], zv.v[m], acc);
            if constexpr (SAVE)
                return tail(acc, ae, std::false_type{}, std::integral_constant<int, MODE_AE>{},
                            [&](const int q, const f4 hq) { if (valid) store_nt<(NWV >= 8)>(reinterpret_cast<f4*>(rows + (size_t)q * lstride), hq); });
            return tail(acc, ae, std::false_type{}, std::integral_constant<int, MODE_AE>{}, [](int, f4) {});
        }
        return acc;
    };
    // SAVE (DAE): AE head activations per grid point [3,T,B,H], per event [nE,3,B,H]; the event's i0 in slot layout [nE,B,16]
    const size_t sae_layer = (SAVE && DAE) ? (size_t)a.T * a.B * (16 * NWV) : 0;
    float* sae_lane = (SAVE && DAE) ? a.saeact + (size_t)b * (16 * NWV) + 16 * w + 4 * g : nullptr;
    auto sae_rows = [&](const long long kk) -> float* { return (SAVE && DAE) ? sae_lane + (size_t)kk * a.B * (16 * NWV) : nullptr; };
    auto load_x = [&](long long k, float (&dst)[NX]) {
#pragma unroll
        for (int r = 0; r < NX; ++r) {      // branch-free: a clamped column, the predicate only selects the value (no divergent region)
            const int d = 4 * r + g;
            const float v = a.x.p[k * a.x.st + b * a.x.sb + (d < xd ? d : 0)];
            dst[r] = d < xd ? v : 0.0f;
        }
    };
    auto store_x_at = [&](float* o) {        // o = this trajectory's output row
        if (w == 0 && valid) {
#pragma unroll
            for (int r = 0; r < NX; ++r) if (4 * r + g < xd) o[4 * r + g] = x[r];
        }
    };
    auto store_x = [&](long long k) { store_x_at(a.xo + (k * a.B + b) * xd); };
    // i_out: each i-dim is stored from its `s`-block slot (q >= ne) so that exactly one lane group writes it
    auto store_i_at = [&](float* o, const f4 iv) {
        if constexpr (DAE) {
            if (w == 1 && valid) {
#pragma unroll
                for (int m = 0; m < NZM; ++m) if (ekind[m] == 2 && 4 * m + g >= ne) o[ecol[m]] = iv[m];
            }
        }
    };
    auto store_i = [&](long long k, const f4 iv) { if constexpr (DAE) store_i_at(a.io + (k * a.B + b) * idim, iv); };

    if constexpr (W_GLB) __syncthreads();      // the LDS-parked layer constants are in place
    store_x(0);
    f4 icur = f4{0.f, 0.f, 0.f, 0.f};
    Arr<NZA> zaz_nxt = {}, zav_nxt = {};      // raw z / v reads of grid point k+1 for the AE head (selected at use)
    if constexpr (DAE) {   // my_solvers.py:95
        Arr<NZA> rz, rv;
        load_ae_raw(0, -1, rz, rv);
        float xa[NX];
#pragma unroll
        for (int r = 0; r < NX; ++r) xa[r] = x[r];
        if (true_x) load_x(0, xa);
        icur = ae_eval(xa, pick_ae(rz, rv), sae_rows(0), sae_layer);
        store_i(0, icur);
        if (nT > 1) load_ae_raw(1, -1, zaz_nxt, zav_nxt);
    }
    if (nT < 2) return;

    float t_cur = tp[0], t_nxt = tp[tst];
#ifndef PSNODE_UNIFORM_EV
#define PSNODE_UNIFORM_EV 1
#endif
    // UNIFORM_EV: the event index of a step is wave-uniform (block of 64 steps in a VGPR, read with v_readlane).  Otherwise it is
    // fetched per step through a formally per-lane address two steps ahead (round 1's scheme: no block reload, but per-lane
    // address arithmetic for every input row).  Chosen per kernel family by measurement.
    constexpr bool UNIFORM_EV = (PSNODE_UNIFORM_EV == 1) || (PSNODE_UNIFORM_EV == 2 && DAE);
    int lane_zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));
    const int* evp = a.ev + lane_zero;
    int evb = UNIFORM_EV ? load_evb(0) : 0;
    int ev_cur = UNIFORM_EV ? __builtin_amdgcn_readlane(evb, 0) : (a.ev ? a.ev[0] : -1);
    int ev_n1 = (!UNIFORM_EV && a.ev && nT > 2) ? evp[1] : -1;
    Arr<NZM> exz_nxt = {}, exv_nxt = {};
    load_de_raw(0, ev_cur, exz_nxt, exv_nxt);
    // Running row pointers of the time loop (one 64-bit add per array and step instead of a 64-bit multiply per load): the clock entry the
    // next prefetch reads (t[k+2]), the z / v rows of grid point k+1, the output rows of the deferred store.  They stop advancing at the
    // last row, so the prefetch of the final step re-reads valid rows (its results are never used).
    const float* trun = tp + (nT > 2 ? 2 : nT - 1) * tst;
    const float* zrun = zp + zst;
    const float* vrun = vp + vst;
    float* xo_run = a.xo + (a.B + b) * xd;                            // row 1
    float* io_run = DAE ? a.io + (a.B + b) * idim : nullptr;
    const long long xo_step = a.B * xd, io_step = a.B * idim;

    for (long long k = 0; k + 1 < nT; ++k) {
        // Everything still in flight here was issued a whole step ago (the prefetch of this step's inputs, the previous result's
        // store): wait for it HERE, once.  Left to the compiler, the first use of a prefetched value may be scheduled behind this
        // step's stores, and since those sit in branches (only wave 0 stores) the only wait count that is safe there also waits
        // for the stores just issued (ODE_01 Euler: 1.26 -> 1.35 ms per launch when that happened).
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
        const float h_ = t_nxt - t_cur;
        t_cur = t_nxt;
        Arr<NZM> extv;
#pragma unroll
        for (int m = 0; m < NZM; ++m) extv.v[m] = pick(ekind[m], exz_nxt.v[m], exv_nxt.v[m]);
        const Arr<NZA> zva = pick_ae(zaz_nxt, zav_nxt);     // z|v of grid point k+1 for the AE head at the end of this step
        const int ev_now = ev_cur;
        // Deferred store of the PREVIOUS step's result (still in x / icur): issued behind the consumption of this step's prefetched
        // inputs and in front of the next prefetch, it has a full step to drain.  Stored at the end of its own step it sat in
        // front of the loop-top s_waitcnt vmcnt(0) (the only count that is safe on the waves that do not store), and the storing
        // wave -- hence, through the barriers, the whole tile -- waited out the store's round trip every step.
        if (k > 0) {
            store_x_at(xo_run);
            xo_run += xo_step;
            if constexpr (DAE) { store_i_at(io_run, icur); io_run += io_step; }
        }

        float xsrc[NX];
#pragma unroll
        for (int r = 0; r < NX; ++r) xsrc[r] = x[r];
        if (true_x) {                            // teacher forcing: the step starts from the dataset's x[k]; read HERE, in front of the
            load_x(k, xsrc);                     // prefetch, and waited for inside the branch (cf. the teacher-forced i below)
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
        }
        // teacher-forced algebraic input i[k]: read HERE, in front of the prefetch, and waited for inside the branch -- what is still
        // in flight at this point are the inputs of this very step.  Read where it is consumed (behind the prefetch) the join of this
        // branch put an s_waitcnt vmcnt(0) on the prefetch of EVERY step, teacher forcing or not.
        Arr<NZM> itrue = {};
        if constexpr (DAE) {
            if (true_i) {
                const float* ir = a.i.p + b * a.i.sb + k * a.i.st;
#pragma unroll
                for (int m = 0; m < NZM; ++m) itrue.v[m] = ir[ekind[m] == 2 ? ecol[m] : 0];
                __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
            }
        }
        // prefetch the next step's inputs (consumed one full step later): unconditional -- inside `if (k + 2 < nT)` the results are phis
        // whose copies (and the wait for the loads) land in the branch
        {
            const bool more = k + 2 < nT;
            t_nxt = *trun;
            if constexpr (UNIFORM_EV) {
                if (((k + 1) & 63) == 0) evb = load_evb((k + 1) >> 6);
                ev_cur = __builtin_amdgcn_readlane(evb, (int)((k + 1) & 63));
            } else {
                ev_cur = ev_n1;
                ev_n1 = (a.ev && k + 3 < nT) ? evp[k + 2] : -1;
            }
            RowPtr rp;
            rp.z = zrun; rp.v = vrun;
            if (__builtin_amdgcn_readfirstlane(ev_cur) >= 0) rp = rows_at(k + 1, ev_cur);    // an event step takes the jump rows
            load_de_rows(rp, exz_nxt);
            if constexpr (DAE) {
                RowPtr ra;
                ra.z = zrun + (more ? zst : 0); ra.v = vrun + (more ? vst : 0);            // raw z|v of grid point k+2
                load_ae_rows(ra, zaz_nxt);
            }
            trun += (k + 3 < nT) ? tst : 0;
            zrun += more ? zst : 0;
            vrun += more ? vst : 0;
        }
        if constexpr (DAE) {
            // event: i0 = g(x0; z_jump, v_jump) with the RUNNING state (my_solvers.py:108-110)
            if (__builtin_amdgcn_readfirstlane(ev_now) >= 0) {
                Arr<NZA> rz, rv;
                load_ae_raw(k, ev_now, rz, rv);
                float* erows = nullptr;
                if constexpr (SAVE) erows = a.sevact + ((size_t)ev_now * 3 * a.B + b) * (16 * NWV) + 16 * w + 4 * g;
                icur = ae_eval(x, pick_ae(rz, rv), erows, (size_t)a.B * (16 * NWV));
                if constexpr (SAVE) {
                    if (w == 0 && valid) {
                        float* er = a.sevi + ((size_t)ev_now * a.B + b) * 16 + g;
#pragma unroll
                        for (int m = 0; m < NZM; ++m) er[4 * m] = icur[m];
                    }
                }
            }
        }
        // per-step constant of L1: c0 + W1[:, ext columns] . (ext - a0 | ext)
        f4 cz = c0;
#pragma unroll
        for (int m = 0; m < NZM; ++m) {
            float e = extv.v[m];
            if constexpr (DAE) {
                if (ekind[m] == 2) e = true_i ? itrue.v[m] : icur[m];
            }
            cz = mfma4(w1z.v[m], e - a0e.v[m], cz);
        }

        const f4 k1 = rhs(xsrc, cz);
        if constexpr (METHOD == PSNODE_EULER) {
#pragma unroll
            for (int r = 0; r < NX; ++r) x[r] = xsrc[r] + h_ * k1[r];
        } else if constexpr (METHOD == PSNODE_MIDPOINT) {
            float xs[NX];
            const float hh = 0.5f * h_;
#pragma unroll
            for (int r = 0; r < NX; ++r) xs[r] = xsrc[r] + k1[r] * hh;
            const f4 k2 = rhs(xs, cz);
#pragma unroll
            for (int r = 0; r < NX; ++r) x[r] = xsrc[r] + h_ * k2[r];
        } else {
            float xs[NX];
#pragma unroll
            for (int r = 0; r < NX; ++r) xs[r] = xsrc[r] + h_ * k1[r] * kOneThird;
            const f4 k2 = rhs(xs, cz);
#pragma unroll
            for (int r = 0; r < NX; ++r) xs[r] = xsrc[r] + h_ * (k2[r] - k1[r] * kOneThird);
            const f4 k3 = rhs(xs, cz);
#pragma unroll
            for (int r = 0; r < NX; ++r) xs[r] = xsrc[r] + h_ * (k1[r] - k2[r] + k3[r]);
            const f4 k4 = rhs(xs, cz);
#pragma unroll
            for (int r = 0; r < NX; ++r) x[r] = xsrc[r] + (k1[r] + 3.0f * (k2[r] + k3[r]) + k4[r]) * h_ * 0.125f;
        }
        if constexpr (DAE) {   // my_solvers.py:121: i1 at the right grid point, un-jumped inputs
            float xa[NX];
#pragma unroll
            for (int r = 0; r < NX; ++r) xa[r] = x[r];
            if (true_x) {
                load_x(k + 1, xa);
                __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) (the step's prefetch is a whole step old here)
            }
            icur = ae_eval(xa, zva, sae_rows(k + 1), sae_layer);
        }
    }
    store_x_at(xo_run);
    if constexpr (DAE) store_i_at(io_run, icur);
}

inline int nzm_of(const IntegrateDev& a, bool dae) { return (2 * (a.zd + (dae ? a.vd + a.id : 0)) + 3) / 4; }
inline int nza_of(const IntegrateDev& a) { return (a.zd + a.vd + 3) / 4; }
inline int na_of(const IntegrateDev& a, bool dae) { return (a.xd + a.zd + (dae ? a.vd + a.id : 0) + 3) / 4; }

template <int NWV, int METHOD, int NXR = kNXc>
hipError_t launch_shape(const IntegrateDev& a, bool dae, const float* pde, const float* pae, int NA, hipStream_t s) {
    const dim3 grid((unsigned)((a.B + TBM - 1) / TBM)), block(64 * NWV);
    const int NZM = nzm_of(a, dae), NZA = dae ? nza_of(a) : 0;
#define PSNODE_LAUNCH(NZM_, NZA_, DAE_)                                                                                    \
    {                                                                                                                      \
        auto kern = &integrate_mfma_kernel<METHOD, NXR, NZM_, NZA_, DAE_, NWV>;                                           \
        const size_t lds = ae_weights_in_lds(DAE_, NWV) ? ae_lds_bytes(NWV) : 0;                                           \
        if (lds) {                                                                                                         \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                        \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                      \
            if (e != hipSuccess) return e;                                                                                 \
        }                                                                                                                  \
        hipLaunchKernelGGL(kern, grid, block, lds, s, a, pde, pae, NA);                                                    \
        return hipGetLastError();                                                                                          \
    }
    if constexpr (NXR == kNXc && !weights_streamed(NWV)) {      // the training forwards exist where a fused backward does (hidden <= 128)
        if (!dae && a.sact) {
            if (a.flags & PSNODE_FLAG_INPUT_TRUE_X) return hipErrorNotSupported;
#define PSNODE_LAUNCH_SAVE(NZM_)                                                                                           \
    {                                                                                                                      \
        hipLaunchKernelGGL((integrate_mfma_kernel<METHOD, NXR, NZM_, 0, false, NWV, true>), grid, block, 0, s, a, pde, pae, NA); \
        return hipGetLastError();                                                                                          \
    }
            switch (NZM) {
                case 0: PSNODE_LAUNCH_SAVE(0)
                case 1: PSNODE_LAUNCH_SAVE(1)
                case 2: PSNODE_LAUNCH_SAVE(2)
                case 3: PSNODE_LAUNCH_SAVE(3)
                case 4: PSNODE_LAUNCH_SAVE(4)
                default: return hipErrorNotSupported;
            }
#undef PSNODE_LAUNCH_SAVE
        }
    }
    if constexpr (NXR == kNXc && !weights_streamed(NWV)) {
        if (dae && a.sact) {
            if (a.flags & (PSNODE_FLAG_INPUT_TRUE_I | PSNODE_FLAG_INPUT_TRUE_X)) return hipErrorNotSupported;
#define PSNODE_LAUNCH_SAVE(NZM_, NZA_)                                                                                     \
    {                                                                                                                      \
        auto kern = &integrate_mfma_kernel<METHOD, NXR, NZM_, NZA_, true, NWV, true>;                              \
        const size_t lds = ae_weights_in_lds(true, NWV) ? ae_lds_bytes(NWV) : 0;                                           \
        if (lds) {                                                                                                         \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                        \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                      \
            if (e != hipSuccess) return e;                                                                                 \
        }                                                                                                                  \
        hipLaunchKernelGGL(kern, grid, block, lds, s, a, pde, pae, NA);                                                    \
        return hipGetLastError();                                                                                          \
    }
            switch (NZM * 10 + NZA) {
                case 11: PSNODE_LAUNCH_SAVE(1, 1)
                case 21: PSNODE_LAUNCH_SAVE(2, 1)
                case 31: PSNODE_LAUNCH_SAVE(3, 1)
                case 41: PSNODE_LAUNCH_SAVE(4, 1)
                case 32: PSNODE_LAUNCH_SAVE(3, 2)
                case 42: PSNODE_LAUNCH_SAVE(4, 2)
                default: return hipErrorNotSupported;
            }
#undef PSNODE_LAUNCH_SAVE
        }
    }
    if (a.sact) return hipErrorNotSupported;
    if (!streamed_class_ok(NWV, dae, NXR != kNXc, NZM)) return hipErrorNotSupported;
    if (!dae) {
        if constexpr (!streamed_class_ok(NWV, false, NXR != kNXc, 0)) return hipErrorNotSupported;
        else
        switch (NZM) {
            case 0: PSNODE_LAUNCH(0, 0, false)
            case 1: PSNODE_LAUNCH(1, 0, false)
            case 2: PSNODE_LAUNCH(2, 0, false)
            case 3: PSNODE_LAUNCH(3, 0, false)
            case 4: PSNODE_LAUNCH(4, 0, false)
            default: return hipErrorNotSupported;
        }
    }
    if constexpr (NXR != kNXc || !streamed_class_ok(NWV, true, false, 1)) return hipErrorNotSupported;     // the DAE kernels hold two register-resident MLPs: x_dim <= 8
    else
    switch (NZM * 10 + NZA) {
        case 11: PSNODE_LAUNCH(1, 1, true)
        case 21: PSNODE_LAUNCH(2, 1, true)
        case 31: PSNODE_LAUNCH(3, 1, true)
        case 32: PSNODE_LAUNCH(3, 2, true)
        default:
            if constexpr (streamed_class_ok(NWV, true, false, 4)) {
                switch (NZM * 10 + NZA) {
                    case 41: PSNODE_LAUNCH(4, 1, true)
                    case 42: PSNODE_LAUNCH(4, 2, true)
                    default: return hipErrorNotSupported;
                }
            }
            return hipErrorNotSupported;
    }
#undef PSNODE_LAUNCH
}

template <int NWV, int METHOD>
hipError_t launch_method(const IntegrateDev& a, bool dae, const float* pde, const float* pae, int NA, hipStream_t s) {
    if (a.xd > 4 * kNXc) return launch_shape<NWV, METHOD, kNXw>(a, dae, pde, pae, NA, s);     // x_dim 9..16: four x registers per lane (ODE only), 16-byte L4 all-reduce
    return launch_shape<NWV, METHOD>(a, dae, pde, pae, NA, s);
}

// pack the weights into the register images, then run the integrator (both on `stream`)
template <int NWV>
hipError_t launch_mfma_nw(const IntegrateDev& a, bool dae, float* pack, hipStream_t stream) {
    const int NZM = nzm_of(a, dae), NA = na_of(a, dae);
    const int ne = a.zd + (dae ? a.vd + a.id : 0);
    PackMfma p;
    p.ae = 0; p.nw = NWV; p.xd = a.xd; p.ne = ne; p.n = a.xd + ne; p.nzv = a.zd + (dae ? a.vd : 0);
    p.NX = a.xd > 4 * kNXc ? kNXw : kNXc; p.NB = 0; p.NE = NZM; p.NA = NA; p.fold = 1; p.hreal = a.de.out_dim[0];
    p.w1 = a.de.w[0]; p.b1 = a.de.bias[0]; p.w2 = a.de.w[1]; p.b2 = a.de.bias[1];
    p.w3 = a.de.w[2]; p.b3 = a.de.bias[2]; p.w4 = a.de.w[3]; p.b4 = a.de.bias[3];
    p.out_dim = a.xd;
    p.out = pack;
    p.scaled = (a.sact || !PSNODE_SCALED_ELU) ? 0 : 1;        // the inference instances (SAVE = false) run the hidden layers in the log2e-scaled domain
    hipLaunchKernelGGL(pack_mfma_kernel, dim3(16), dim3(256), 0, stream, p);
    const size_t one = (size_t)NWV * (max_regs(NWV) + NA) * 64 + stream_image_floats(NWV);
    if constexpr (weights_streamed(NWV))     // the stream image sits right behind the register image of its MLP (Tail::gw)
        hipLaunchKernelGGL(pack_stream_kernel, dim3(64), dim3(256), 0, stream, p,
                           reinterpret_cast<f4*>(pack + (size_t)NWV * (pack_fwd_count(p)) * 64));
    float* pack_ae = pack + one;
    if (dae) {
        PackMfma q = p;
        q.ae = 1; q.NB = 0; q.NE = nza_of(a); q.fold = 0;
        q.w1 = a.ae.w[0]; q.b1 = a.ae.bias[0]; q.w2 = a.ae.w[1]; q.b2 = a.ae.bias[1];
        q.w3 = a.ae.w[2]; q.b3 = a.ae.bias[2]; q.w4 = a.ae.w[3]; q.b4 = a.ae.bias[3];
        q.out_dim = a.id;
        q.out = pack_ae;
        hipLaunchKernelGGL(pack_mfma_kernel, dim3(16), dim3(256), 0, stream, q);
        if constexpr (weights_streamed(NWV))
            hipLaunchKernelGGL(pack_stream_kernel, dim3(64), dim3(256), 0, stream, q,
                               reinterpret_cast<f4*>(pack_ae + (size_t)NWV * (pack_fwd_count(q)) * 64));
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    switch (a.method) {
        case PSNODE_EULER: return launch_method<NWV, PSNODE_EULER>(a, dae, pack, pack_ae, NA, stream);
        case PSNODE_MIDPOINT: return launch_method<NWV, PSNODE_MIDPOINT>(a, dae, pack, pack_ae, NA, stream);
        default: return launch_method<NWV, PSNODE_RK4_38>(a, dae, pack, pack_ae, NA, stream);
    }
}

}  // namespace
}  // namespace psnode
