// Generic fused fixed-grid integrator for gfx950 (K0): any layer count, layer widths and state dims within the ABI limits.
//
// One launch integrates ALL T-1 steps: a workgroup of four waves owns TB = 16 trajectories (they never interact, my_solvers.py:66 is
// row-wise over the batch) and walks the time grid with its state in LDS.  This is the always-available HIP path: the shapes outside
// the specialised integrators' classes (x_dim > 16, z + v + i > 8, depth != 3 hidden layers, mixed or very wide layers -- all of them
// data- or user-defined upstream, neural_00_ODE_01_no_encode.py:293) run here.
//
// Round 6: the Linear layers run on v_mfma_f32_16x16x4_f32 (before: one fp32 fmaf chain per (unit, 4 trajectories) item with the
// weights re-staged through LDS for every layer of every evaluation).  D[i = unit][j = trajectory] += A[i][k] * B[k][j]:
//   * activations live in LDS in QUAD-ROW order, float index ((col / 4) * 16 + traj) * 4 + col % 4: MFMA step (q, c) -- c = 0..3 -- takes
//     the columns 16 q + 4 k + c in its k-slot k, so lane (k, j)'s B operands of four consecutive steps are ONE lane-linear ds_read_b128
//     (f4 index 64 q + lane), and a D tile (lane (g, j), register r = unit 16 nt + 4 g + r) goes back as ONE lane-linear ds_write_b128
//     (f4 index 64 nt + lane): no transposes, no bank conflicts;
//   * the weights come from an image in the workspace (pack_image_kernel), [tile nt][q][lane] f4 with the same column order, zero-padded
//     to 16 rows x 16 columns.  Layers whose images fit the LDS left over (greedy in layer order, DE first: generic_plan) are copied there
//     once per launch and read like the activations; the others are STREAMED: one coalesced 1 KB global load per four MFMAs, L2-resident
//     (every workgroup reads the same image every evaluation), issued one chunk of 16 MFMAs ahead and in flight across the layer barrier
//     (lds_barrier waits for LDS traffic only);
//   * output tile nt of a layer belongs to wave nt % 4; bias (padded, in LDS) added behind the MFMAs; ELU on the
//     accumulator; one barrier per layer.
// The two MLP inputs (DE: a0 | s - a0 | s, AE: a0 | x | z | v) have their own buffers and keep their constant columns between evaluations:
// a0 is written once, the externals once per step, and per stage only the state's columns -- by the same pass that applies the stage's
// update, so an evaluation costs its layers' barriers plus one.  The next grid point's clocks, event index and z | v rows are loaded a
// step ahead (registers), the AE head's input of the end-of-step evaluation is written by the last stage's update.
// Padded columns: the image holds zeros there and the input builders write zeros into the pad columns of the first layer's input; a
// hidden layer's pad units come out of the MFMA as ELU(0 + 0) = 0.
#include "psnode_common.h"

namespace psnode {

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int TB = 16;    // trajectories per workgroup
constexpr int NT = 256;   // threads per workgroup (4 waves)

__host__ __device__ constexpr int up16(int v) { return (v + 15) & ~15; }
// float offset of (column r, trajectory c) in a quad-row activation buffer
__device__ __forceinline__ int qi(int r, int c) { return ((((r >> 2) * TB) + c) << 2) | (r & 3); }
// Column order of the two MLP inputs (the first layers' images use the same one): the columns that change most often come first and end on
// a quad boundary, so that the register forms can fold everything behind them into a per-step (DE) / per-trajectory (AE) constant.
//   DE  cat(a0, s - a0, s) (DE_Func.forward):   [ (s - a0)_x | s_x | pad to SX ] [ a0 | (s - a0)_ext | s_ext | pad ]    ext = z | v | i
//   AE  cat(a0, x, z, v)   (AE_Func.forward):   [ x | z | v | pad to SA ] [ a0 | pad ]
__host__ __device__ constexpr int de_sx(int xd) { return up16(2 * xd); }
__host__ __device__ constexpr int de_k16(int xd, int n) { return de_sx(xd) + up16(n + 2 * (n - xd)); }
__host__ __device__ inline int de_orig_col(int k, int xd, int n) {      // -> column of the nn.Linear weight, -1: pad
    const int ne = n - xd;
    if (k < xd) return n + k;
    if (k < 2 * xd) return 2 * n + (k - xd);
    if (k < de_sx(xd)) return -1;
    k -= de_sx(xd);
    if (k < n) return k;
    if (k < n + ne) return n + xd + (k - n);
    if (k < n + 2 * ne) return 2 * n + xd + (k - n - ne);
    return -1;
}
__host__ __device__ constexpr int ae_sa(int xd, int nzv) { return up16(xd + nzv); }
__host__ __device__ constexpr int ae_k16(int xd, int nzv, int n) { return ae_sa(xd, nzv) + up16(n); }
__host__ __device__ inline int ae_orig_col(int k, int xd, int nzv, int n) {
    if (k < xd + nzv) return n + k;
    if (k < ae_sa(xd, nzv)) return -1;
    k -= ae_sa(xd, nzv);
    return k < n ? k : -1;
}

// floats of one layer's image: N16 x K16 weights + N16 biases
// (+ 16 columns: the two padded blocks of a first layer)
__host__ __device__ constexpr size_t image_floats(int K, int N) { return (size_t)up16(N) * (up16(K) + 16) + up16(N); }
// floats of the padded biases of every layer (LDS region behind the kernel's state)
__host__ __device__ inline int generic_bias_floats(const IntegrateDev& a, bool dae) {
    int tot = 0;
    for (int l = 0; l < a.de.n_layers; ++l) tot += up16(a.de.out_dim[l]);
    if (dae) for (int l = 0; l < a.ae.n_layers; ++l) tot += up16(a.ae.out_dim[l]);
    return tot;
}

// One MLP as the time loop sees it: wave-uniform scalars, built once per launch so that no kernel-argument load (a scalar-cache round trip,
// and an lgkmcnt wait that also drains the LDS queue) sits between two chunks.  The layer loop of mlp_eval is fully unrolled over
// ML (4 or kMaxLayers: the kernel is instantiated for both), which makes every index below a constant.
template <int ML>
struct Tab {
    int L;
    unsigned dims[ML];     // resident << 31 | quads of the contraction (ceil(K / 16)) << 16 | output tiles (ceil(N / 16))
    unsigned off[ML];      // streamed layer: f4 offset of its image from `base`; resident layer: FLOAT offset of its copy in LDS
    unsigned boff[ML];     // float offset of the padded bias in LDS
    const f4* base;                // image of layer 0 (workspace)
    int qx;                        // quads of layer 0's leading block (DE: the state's columns, AE: x | z | v)
    unsigned first_off;            // this wave's first STREAMED chunk of an evaluation: f4 offset ...
    int first_q;                   // ... and the quads of that tile (0: the wave owns no tile in a streamed layer of this MLP)
};
__device__ __forceinline__ int tab_tiles(unsigned d) { return (int)(d & 0xffffu); }
__device__ __forceinline__ int tab_quads(unsigned d) { return (int)((d >> 16) & 0x7fffu); }
__device__ __forceinline__ bool tab_res(unsigned d) { return (d >> 31) != 0; }

// `res`: bit l = layer l's image is resident in LDS; `bias_at` / `img_at`: running float offsets of the LDS regions (advanced)
// k0 / qx: layer 0's padded contraction length (de_k16 / ae_k16) and the quads of its leading block
template <int ML>
__device__ __forceinline__ Tab<ML> make_tab(const MlpDev& m, int w, unsigned res, unsigned& bias_at, unsigned& img_at, int k0, int qx) {
    Tab<ML> t;
    t.L = m.n_layers;
    t.qx = qx;
    t.base = reinterpret_cast<const f4*>(m.wt[0]);
    t.first_off = 0; t.first_q = 0;
#pragma unroll
    for (int l = 0; l < ML; ++l) {
        const int K = l ? m.out_dim[l - 1] : k0, N = m.out_dim[l];
        const unsigned S4 = (K + 15) >> 4, NTL = (N + 15) >> 4;
        const bool on = l < m.n_layers, r = on && ((res >> l) & 1u);
        t.dims[l] = on ? ((r ? 1u << 31 : 0u) | S4 << 16 | NTL) : 0u;
        t.off[l] = !on ? 0u : (r ? img_at : (unsigned)((m.wt[l] - m.wt[0]) >> 2));
        t.boff[l] = bias_at;
        if (on) bias_at += 16u * NTL;
        if (r) img_at += 256u * NTL * S4;
    }
#pragma unroll
    for (int l = ML - 1; l >= 0; --l)
        if (l < m.n_layers && !tab_res(t.dims[l]) && tab_tiles(t.dims[l]) > w) {
            t.first_off = t.off[l] + (unsigned)w * tab_quads(t.dims[l]) * 64u;
            t.first_q = tab_quads(t.dims[l]);
        }
    return t;
}

// field-by-field select (a ternary on the structs goes through a stack copy: scratch)
template <int ML>
__device__ __forceinline__ Tab<ML> pick_tab(bool first, const Tab<ML>& x, const Tab<ML>& y) {
    // readfirstlane: the results are wave-uniform and have to stay scalar -- kept in VGPRs (the compiler does that under SGPR pressure)
    // every loop bound of mlp_eval turns into exec-mask control flow
    auto u = [](unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
    Tab<ML> t;
    t.L = (int)u(first ? x.L : y.L);
    t.qx = (int)u(first ? x.qx : y.qx);
    const unsigned long long pb = reinterpret_cast<unsigned long long>(first ? x.base : y.base);
    t.base = reinterpret_cast<const f4*>((unsigned long long)u((unsigned)(pb >> 32)) << 32 | u((unsigned)pb));
    t.first_off = u(first ? x.first_off : y.first_off);
    t.first_q = (int)u(first ? x.first_q : y.first_q);
#pragma unroll
    for (int l = 0; l < ML; ++l) {
        t.dims[l] = u(first ? x.dims[l] : y.dims[l]);
        t.off[l] = u(first ? x.off[l] : y.off[l]);
        t.boff[l] = u(first ? x.boff[l] : y.boff[l]);
    }
    return t;
}

// copies the biases (always) and the resident images into LDS; no barrier
template <int ML>
__device__ __forceinline__ void load_resident(const MlpDev& m, const Tab<ML>& t, float* lds) {
#pragma unroll
    for (int l = 0; l < ML; ++l) {
        if (l >= t.L) break;
        const int NTL = tab_tiles(t.dims[l]), S4 = tab_quads(t.dims[l]);
        const float* __restrict__ src = m.wt[l];
        for (int i = threadIdx.x; i < 16 * NTL; i += NT) lds[t.boff[l] + i] = src[(size_t)NTL * S4 * 256 + i];
        if (tab_res(t.dims[l]))
            for (int i = threadIdx.x; i < NTL * S4 * 64; i += NT) reinterpret_cast<f4*>(lds + t.off[l])[i] = reinterpret_cast<const f4*>(src)[i];
    }
}

// A-operand look-ahead carried from one layer / MLP evaluation into the next: one chunk (up to four f4 = 16 MFMA steps).
//   STREAM kernels: the wave's next chunk among the STREAMED layers, from the workspace image (hides the L2 latency);
//   all-resident kernels: chunk 0 of the wave's first tile of the next layer, read from LDS in front of the layer barrier, so that only
//   the activations are read behind it.
// `tag`: the MLP (its workspace image) whose first chunk `a` holds at the start of an evaluation; a wrong guess costs one more read.
struct Pref {
    f4 a[4];
    const f4* tag;
};

#ifndef PSNODE_K0_ABL      // ablation builds (timing only, wrong results): 1 = a quarter of the MFMAs, 2 = no ELU, 3 = no layer barrier, 4 = no MLP at all, 5 = no operand reads / MFMAs, 6 = no look-ahead read
#define PSNODE_K0_ABL 0
#endif
__device__ __forceinline__ void mfma_quad(const f4 av, const f4 bv, f4& acc) {
#pragma unroll
    for (int e = 0; e < (PSNODE_K0_ABL == 1 ? 1 : 4); ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[e], acc, 0, 0, 0);
}

// One output tile with both operands in LDS, Q quads, straight-line: every read in flight before the first MFMA, two accumulator chains.
// PA: the A operands of quads 0 .. 3 come from `pa` (read in front of the layer barrier).  With one wave per SIMD nothing hides a taken
// branch (an instruction-cache round trip each): a loop over the quads with its guards and tails cost more than the MFMAs it issued
// (profiles/r06_k0_generic_mfma.txt), so the contraction lengths up to 128 columns get a body each and the layer picks one with a switch.
template <int Q, bool PA>
__device__ __forceinline__ f4 tile_body(const f4* at, const f4* bq, const f4 (&pa)[4]) {
    f4 av[Q], bv[Q];
#pragma unroll
    for (int c = 0; c < Q; ++c) {
        bv[c] = bq[c * 64];
        if (PA && c < 4) av[c] = pa[c]; else av[c] = at[c * 64];
    }
    __builtin_amdgcn_sched_barrier(0);
    f4 acc = f4{0.f, 0.f, 0.f, 0.f}, acc2 = acc;
#pragma unroll
    for (int c = 0; c < Q; ++c) mfma_quad(av[c], bv[c], (c & 1) ? acc2 : acc);
    return Q > 1 ? acc + acc2 : acc;
}
template <bool PA>
__device__ __forceinline__ f4 tile_any(int S4, const f4* at, const f4* bq, const f4 (&pa)[4]) {
    switch (S4) {
        case 1: return tile_body<1, PA>(at, bq, pa);
        case 2: return tile_body<2, PA>(at, bq, pa);
        case 3: return tile_body<3, PA>(at, bq, pa);
        case 4: return tile_body<4, PA>(at, bq, pa);
        case 5: return tile_body<5, PA>(at, bq, pa);
        case 6: return tile_body<6, PA>(at, bq, pa);
        case 7: return tile_body<7, PA>(at, bq, pa);
        case 8: return tile_body<8, PA>(at, bq, pa);
        default: break;
    }
    // longer contractions: eight quads straight, then chunks of four and single quads
    f4 acc = tile_body<8, PA>(at, bq, pa), acc2 = f4{0.f, 0.f, 0.f, 0.f};
    int q0 = 8;
    for (; q0 + 4 <= S4; q0 += 4) {
        f4 av[4], bv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { av[c] = at[(q0 + c) * 64]; bv[c] = bq[(q0 + c) * 64]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 4; ++c) mfma_quad(av[c], bv[c], (c & 1) ? acc2 : acc);
    }
    for (; q0 < S4; ++q0) mfma_quad(at[q0 * 64], bq[q0 * 64], acc);
    return acc + acc2;
}

// MLP over the TB columns: layer 0 reads the quad-row buffer at float offset `in`, the layers write `ping` / `pong` alternately; returns
// the offset of the last layer's output.  Ends with a barrier.  `nx`: the MLP evaluated after this one.
template <bool STREAM, int ML>
__device__ __forceinline__ int mlp_eval(const Tab<ML>& T, float* lds, int in, const int ping, const int pong, Pref& pf, const Tab<ML>& nx) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // STREAM: chunk = quads q0 .. q0 + 3 of one tile, clamped inside the tile's run of the image (in bounds, unused beyond the tile's quads).
    // The address comes out of scalar selects and the four loads are unconditional straight-line code: a load inside a conditional block
    // gets its result copied (and waited for) at the end of that block, in front of the MFMAs it should overlap.
    auto fetch = [&](const f4* base, unsigned off, int rem, f4 (&a)[4]) {
        const f4* __restrict__ A = base + off + lane;
#pragma unroll
        for (int c = 0; c < 4; ++c) a[c] = A[(c < rem ? c : rem - 1) * 64];
    };
    // all-resident: chunk 0 of tile w of layer l of t, from LDS (nothing if the wave has no tile there)
    auto peek = [&](const Tab<ML>& t, int l, f4 (&a)[4]) {
        const int S4 = tab_quads(t.dims[l]);
        if (w < tab_tiles(t.dims[l])) {
            const f4* at = reinterpret_cast<const f4*>(lds + t.off[l]) + w * S4 * 64 + lane;
#pragma unroll
            for (int c = 0; c < 4; ++c) a[c] = at[(c < S4 ? c : S4 - 1) * 64];
        }
    };
    if constexpr (STREAM) {
        if (pf.tag != T.base && T.first_q > 0) fetch(T.base, T.first_off, T.first_q, pf.a);
    } else {
        if (pf.tag != T.base) peek(T, 0, pf.a);
    }
    int out = ping;
    if (PSNODE_K0_ABL == 4) { lds_barrier(); return out; }
#pragma unroll
    for (int l = 0; l < ML; ++l) {
        if (l >= T.L) break;
        out = (l & 1) ? pong : ping;
        const int S4 = tab_quads(T.dims[l]), NTL = tab_tiles(T.dims[l]);
        const bool last = (l + 1 == T.L);
        const f4* bq = reinterpret_cast<const f4*>(lds + in) + lane;
        const f4* b16 = reinterpret_cast<const f4*>(lds + T.boff[l]) + (lane >> 4);
        f4* oq = reinterpret_cast<f4*>(lds + out) + lane;
        auto finish = [&](f4 acc, const f4 bias, int nt) {
            acc = acc + bias;
            const f4 e = PSNODE_K0_ABL != 2 ? elu_quad(acc) : acc;
            oq[nt * 64] = last ? acc : e;         // a select, not a branch
        };
        if (!STREAM || tab_res(T.dims[l])) {
            const f4* aq = reinterpret_cast<const f4*>(lds + T.off[l]) + lane;
            int nt = w;
            if constexpr (!STREAM) {
                if (nt < NTL) {                              // first tile: the A operands of its first four quads were read in front of the barrier
                    const f4 bias = b16[4 * nt];
                    finish(PSNODE_K0_ABL == 5 ? bias : tile_any<true>(S4, aq + nt * S4 * 64, bq, pf.a), bias, nt);
                    nt += 4;
                }
            }
            for (; nt < NTL; nt += 4) {
                const f4 bias = b16[4 * nt];
                finish(tile_any<false>(S4, aq + nt * S4 * 64, bq, pf.a), bias, nt);
            }
        } else {
            // where this wave's A stream continues behind its last chunk of layer l: its first tile of a later streamed layer, else of the
            // next evaluation
            const f4* tbase = nx.base;
            unsigned toff = nx.first_off;
            int tq = nx.first_q > 0 ? nx.first_q : 1;
#pragma unroll
            for (int nl = ML - 1; nl > l; --nl)
                if (nl < T.L && !tab_res(T.dims[nl]) && tab_tiles(T.dims[nl]) > w) {
                    tbase = T.base; toff = T.off[nl] + (unsigned)w * tab_quads(T.dims[nl]) * 64u; tq = tab_quads(T.dims[nl]);
                }
            for (int nt = w; nt < NTL; nt += 4) {
                const f4 bias = b16[4 * nt];
                f4 acc = f4{0.f, 0.f, 0.f, 0.f};
                const unsigned coff = T.off[l] + (unsigned)(nt * S4) * 64u;
                for (int q0 = 0; q0 < S4; q0 += 4) {
                    f4 cur[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) cur[c] = pf.a[c];
                    // ---- the next chunk: same tile, the wave's next tile, or the continuation behind this layer
                    const bool same = q0 + 4 < S4, more = nt + 4 < NTL;
                    const f4* nb = (same || more) ? T.base : tbase;
                    const unsigned no = same ? coff + (unsigned)(q0 + 4) * 64u : (more ? coff + (unsigned)(4 * S4) * 64u : toff);
                    const int nr = same ? S4 - q0 - 4 : (more ? S4 : tq);
                    fetch(nb, no, nr, pf.a);
                    if (q0 + 4 <= S4) {
                        f4 bv[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) bv[c] = bq[(q0 + c) * 64];
                        __builtin_amdgcn_sched_barrier(0);      // all four reads in flight before the first MFMA
#pragma unroll
                        for (int c = 0; c < 4; ++c) mfma_quad(cur[c], bv[c], acc);
                    } else {
                        for (int c = 0; q0 + c < S4; ++c) mfma_quad(c == 0 ? cur[0] : (c == 1 ? cur[1] : cur[2]), bq[(q0 + c) * 64], acc);
                    }
                }
                finish(acc, bias, nt);
            }
        }
        if constexpr (!STREAM && PSNODE_K0_ABL != 6) {       // the next layer's first A operands, in front of the barrier
            if (!last) peek(T, l + 1 < ML ? l + 1 : l, pf.a);
            else peek(nx, 0, pf.a);
        }
        if (PSNODE_K0_ABL != 3) lds_barrier();
        in = out;
    }
    if constexpr (STREAM) { if (T.first_q > 0) pf.tag = nx.base; }      // a wave without a streamed tile in T has fetched nothing
    else pf.tag = nx.base;
    return out;
}

// ---- register mode: every layer has at most four output tiles (one per wave), the first contraction at most 16 QM columns (QM = 4 or 8),
// the others at most 64.  The wave's A operands of the WHOLE MLP stay in registers for the launch (wr[l][q]: quad q of its tile of layer l, zero where
// the wave has no tile or the tile is shorter), as in the specialised tile integrators: a layer then reads only the activations from LDS.
// Why it matters: a ds_read_b128 costs a wave about 64 cycles of LDS-to-register transfer, so reading both operands from LDS (8 reads per
// 16 MFMAs) takes as long as the MFMAs themselves (ablation builds, profiles/r06_k0_generic_mfma.txt).
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

// the wave's operand registers of one MLP: layer 0 (the only one whose contraction can exceed 64 columns: the other layers read a layer
// output of at most 64 units) holds QM quads, the others four
template <int ML, int QM>
struct WReg {
    f4 first[QM];
    f4 rest[ML - 1][4];
};

// Layer 0 behind its leading block: bias + the products of quads qx .. of tile nt with the input columns that are constant over a step
// (DE: a0 and the externals) or a trajectory (AE: a0).  Once per step / launch: the A operands come from the workspace image (L2), one
// chunk ahead.  The register forms start layer 0's accumulator from this value and multiply only the leading block per evaluation.
template <int ML>
__device__ __forceinline__ f4 fold0(const Tab<ML>& T, const float* lds, int in, int nt) {
    const int lane = threadIdx.x & 63;
    const int S4 = tab_quads(T.dims[0]);
    const f4* __restrict__ A = T.base + T.off[0] + (unsigned)(nt * S4) * 64u + lane;
    const f4* bq = reinterpret_cast<const f4*>(lds + in) + lane;
    f4 acc = reinterpret_cast<const f4*>(lds + T.boff[0])[4 * nt + (lane >> 4)];
    f4 nxt[4];
    auto fetch4 = [&](int q0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) nxt[c] = A[(q0 + c < S4 ? q0 + c : S4 - 1) * 64];
    };
    fetch4(T.qx < S4 ? T.qx : S4 - 1);
    for (int q0 = T.qx; q0 < S4; q0 += 4) {
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
            for (int c = 0; c < 4; ++c) mfma_quad(cur[c], bv[c], acc);
        } else {
            for (int c = 0; q0 + c < S4; ++c) mfma_quad(c == 0 ? cur[0] : (c == 1 ? cur[1] : cur[2]), bq[(q0 + c) * 64], acc);
        }
    }
    return acc;
}

template <int ML, int QM>
__device__ __forceinline__ void load_regs(const MlpDev& m, const Tab<ML>& t, WReg<ML, QM>& wr) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
    for (int l = 0; l < ML; ++l) {
        const int S4 = l < t.L ? tab_quads(t.dims[l]) : 0, NTL = l < t.L ? tab_tiles(t.dims[l]) : 0;
        const f4* __restrict__ A = reinterpret_cast<const f4*>(m.wt[l < t.L ? l : 0]) + (size_t)(w < NTL ? w : 0) * S4 * 64 + lane;
#pragma unroll
        for (int q = 0; q < (l ? 4 : QM); ++q) {
            const f4 v = (w < NTL && q < (l ? S4 : t.qx)) ? A[q * 64] : f4{0.f, 0.f, 0.f, 0.f};      // layer 0: its leading block only
            if (l == 0) wr.first[q] = v; else wr.rest[l - 1][q] = v;
        }
    }
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

// c0: fold0 of this wave's tile of layer 0 (bias included)
template <int ML, int QM>
__device__ __forceinline__ int mlp_reg(const Tab<ML>& T, float* lds, int in, const int ping, const int pong, const WReg<ML, QM>& wr, const f4 c0) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int out = ping;
#pragma unroll
    for (int l = 0; l < ML; ++l) {
        if (l >= T.L) break;
        out = (l & 1) ? pong : ping;
        const int S4 = tab_quads(T.dims[l]);
        const bool last = (l + 1 == T.L);
        if (w < tab_tiles(T.dims[l])) {
            const f4* bq = reinterpret_cast<const f4*>(lds + in) + lane;
            const f4 bias = reinterpret_cast<const f4*>(lds + T.boff[l])[4 * w + (lane >> 4)];
            f4 acc;
            if (l == 0) acc = tile_reg_any<QM>(T.qx, bq, wr.first) + c0;
            else acc = tile_reg_any<4>(S4, bq, wr.rest[l ? l - 1 : 0]) + bias;
            const f4 e = elu_quad(acc);
            reinterpret_cast<f4*>(lds + out)[w * 64 + lane] = last ? acc : e;
        }
        lds_barrier();
        in = out;
    }
    return out;
}

// ---- wide register form: hidden layers of up to 128 units = up to EIGHT tiles, two per wave (nt = w and w + 4), contractions up to 128
// columns; the last layer at most four tiles.  2 .. 4 layers.  The wave's two tiles share the activation reads: half the LDS traffic per
// MFMA of the one-tile form.  224 operand registers at four layers (the compiler parks part of them in AGPRs).
template <int ML>
struct WRegW {
    f4 first[2][8];
    f4 mid[ML - 2][2][8];
    f4 last[8];
};

template <int ML>
__device__ __forceinline__ void load_regs_wide(const MlpDev& m, const Tab<ML>& t, WRegW<ML>& wr) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const f4 zero = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int l = 0; l < ML; ++l) {
        const bool on = l < t.L, is_last = l + 1 == t.L;
        const int S4 = on ? tab_quads(t.dims[l]) : 0, NTL = on ? tab_tiles(t.dims[l]) : 0;
        const f4* __restrict__ A = reinterpret_cast<const f4*>(m.wt[on ? l : 0]) + lane;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int nt = w + 4 * j;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int QL = l ? S4 : t.qx;               // layer 0: its leading block only
                const f4 v = (nt < NTL && q < QL) ? A[((size_t)(nt < NTL ? nt : 0) * S4 + (q < QL ? q : 0)) * 64] : zero;
                if (is_last) { if (j == 0) wr.last[q] = v; }
                else if (l == 0) wr.first[j][q] = v;
                else if (l < ML - 1) wr.mid[l - 1 < ML - 2 ? l - 1 : 0][j][q] = v;
            }
        }
    }
}

// Q quads, one or two tiles on the same activation reads
template <int Q, bool TWO>
__device__ __forceinline__ void tile2_reg(const f4* bq, const f4 (&wa)[8], const f4 (&wb)[8], f4& ra, f4& rb) {
    f4 bv[Q];
#pragma unroll
    for (int c = 0; c < Q; ++c) bv[c] = bq[c * 64];
    __builtin_amdgcn_sched_barrier(0);
    f4 a0 = f4{0.f, 0.f, 0.f, 0.f}, a1 = a0, b0 = a0, b1 = a0;
#pragma unroll
    for (int c = 0; c < Q; ++c) {
        mfma_quad(wa[c], bv[c], (c & 1) ? a1 : a0);
        if (TWO) mfma_quad(wb[c], bv[c], (c & 1) ? b1 : b0);
    }
    ra = Q > 1 ? a0 + a1 : a0;
    rb = Q > 1 ? b0 + b1 : b0;
}
template <bool TWO>
__device__ __forceinline__ void tile2_any(int S4, const f4* bq, const f4 (&wa)[8], const f4 (&wb)[8], f4& ra, f4& rb) {
    switch (S4) {
        case 1: tile2_reg<1, TWO>(bq, wa, wb, ra, rb); break;
        case 2: tile2_reg<2, TWO>(bq, wa, wb, ra, rb); break;
        case 3: tile2_reg<3, TWO>(bq, wa, wb, ra, rb); break;
        case 4: tile2_reg<4, TWO>(bq, wa, wb, ra, rb); break;
        case 5: tile2_reg<5, TWO>(bq, wa, wb, ra, rb); break;
        case 6: tile2_reg<6, TWO>(bq, wa, wb, ra, rb); break;
        case 7: tile2_reg<7, TWO>(bq, wa, wb, ra, rb); break;
        default: tile2_reg<8, TWO>(bq, wa, wb, ra, rb); break;
    }
}

// c0a / c0b: fold0 of the wave's two tiles of layer 0 (bias included)
template <int ML>
__device__ __forceinline__ int mlp_regw(const Tab<ML>& T, float* lds, int in, const int ping, const int pong, const WRegW<ML>& wr, const f4 c0a,
                                        const f4 c0b) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int out = ping;
#pragma unroll
    for (int l = 0; l < ML; ++l) {
        if (l >= T.L) break;
        out = (l & 1) ? pong : ping;
        const int S4 = tab_quads(T.dims[l]), NTL = tab_tiles(T.dims[l]);
        const bool last = (l + 1 == T.L);
        if (w < NTL) {
            const f4* bq = reinterpret_cast<const f4*>(lds + in) + lane;
            const f4* b16 = reinterpret_cast<const f4*>(lds + T.boff[l]) + (lane >> 4);
            f4* oq = reinterpret_cast<f4*>(lds + out) + lane;
            const bool two = !last && w + 4 < NTL;
            const f4 bias0 = b16[4 * w], bias1 = b16[4 * (two ? w + 4 : w)];
            f4 ra, rb;
            const int QL = l ? S4 : T.qx;
            if (last && l) tile2_any<false>(QL, bq, wr.last, wr.last, ra, rb);
            else if (l == 0) { if (two) tile2_any<true>(QL, bq, wr.first[0], wr.first[1], ra, rb); else tile2_any<false>(QL, bq, wr.first[0], wr.first[0], ra, rb); }
            else {
                constexpr int MI = ML - 2;
                const int mi = l - 1 < MI ? l - 1 : 0;
                if (two) tile2_any<true>(S4, bq, wr.mid[mi][0], wr.mid[mi][1], ra, rb); else tile2_any<false>(S4, bq, wr.mid[mi][0], wr.mid[mi][0], ra, rb);
            }
            ra = ra + (l ? bias0 : c0a);
            oq[w * 64] = last ? ra : elu_quad(ra);
            if (two) { rb = rb + (l ? bias1 : c0b); oq[(w + 4) * 64] = elu_quad(rb); }
        }
        lds_barrier();
        in = out;
    }
    return out;
}

// PF: z | v values a thread keeps in flight for the next grid point (items tid + 256 j); rows beyond 16 PF are loaded where they are used
constexpr int PF = 4;

// MODE 0: weights in registers (mlp_reg; QM = 4 or 8 quads per tile), 1: every image resident in LDS, 2: some layers streamed,
// 3: the DE in the wide register form (mlp_regw)
template <bool DAE, int MODE, int ML, int QM = 4>
__global__ __launch_bounds__(NT) void generic_kernel(const IntegrateDev a) {
    constexpr bool STREAM = MODE == 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const long long b0 = (long long)blockIdx.x * TB;
    const int xd = a.xd, zd = a.zd;
    const int vd = DAE ? a.vd : 0, id = DAE ? a.id : 0;
    const int n = xd + zd + vd + id;   // width of all_initial
    const int ne = n - xd;             // external rows: z | v | i
    const int nzv = zd + vd;

    // quad-row buffers (float offsets): the two MLP inputs keep their constant columns (a0; the externals of a step) between evaluations,
    // only the columns that change are rewritten -- per stage that is the state x alone
    constexpr int inDE = 0;
    const int SX = de_sx(xd), SA = ae_sa(xd, nzv);      // first column of the second block of the DE / AE input
    const int inAE = de_k16(xd, n) * TB;
    const int ping = inAE + (DAE ? ae_k16(xd, nzv, n) * TB : 0);
    const int pong = ping + up16(a.maxo) * TB;
    float* a0 = lds + pong + up16(a.maxo) * TB;   // [n][TB]
    float* ext = a0 + n * TB;          // [ne][TB] z | v | i fed to the DE stages of this step
    float* xcur = ext + ne * TB;       // [xd][TB] running state
    float* xsrc = xcur + xd * TB;      // [xd][TB] start of this step (xcur, or dataset x under teacher forcing)
    float* kbuf = xsrc + xd * TB;      // [4][xd][TB]
    float* icur = kbuf + 4 * xd * TB;  // [id][TB]
    float* zvn = icur + id * TB;       // [nzv][TB] dataset z | v of the NEXT grid point
    float* dts = zvn + nzv * TB;       // [TB]

#ifdef PSNODE_K0_PROF      // discriminator builds only: cycles per phase of workgroup 0, printed by its first thread
    long long prof[6] = {0, 0, 0, 0, 0, 0};
    long long pt = clock64();
#define K0_PROF(i) { const long long now_ = clock64(); prof[i] += now_ - pt; pt = now_; }
#else
#define K0_PROF(i)
#endif
    Pref pf;
    pf.tag = nullptr;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned bias_at = (unsigned)(dts + TB - lds), img_at = bias_at + (unsigned)generic_bias_floats(a, DAE);
    const Tab<ML> tde = make_tab<ML>(a.de, wv, a.k0_res & 0xffu, bias_at, img_at, de_k16(xd, n), SX >> 4);
    const Tab<ML> tae = DAE ? make_tab<ML>(a.ae, wv, (a.k0_res >> 8) & 0xffu, bias_at, img_at, ae_k16(xd, nzv, n), SA >> 4) : tde;
    load_resident(a.de, tde, lds);
    if constexpr (DAE) load_resident(a.ae, tae, lds);
    WReg<MODE == 0 ? ML : 2, MODE == 0 ? QM : 1> wde;
    WRegW<MODE == 3 ? ML : 3> wdw;
    if constexpr (MODE == 3) load_regs_wide<ML>(a.de, tde, wdw);
    const f4 fzero = f4{0.f, 0.f, 0.f, 0.f};
    f4 cde = fzero, cde2 = fzero, cae = fzero;       // fold0 of the wave's tile(s) of the DE's / AE's first layer (register forms)
    WReg<(MODE == 0 && DAE) ? ML : 2, (MODE == 0 && DAE) ? QM : 1> wae;
    if constexpr (MODE == 0) {
        load_regs<ML, QM>(a.de, tde, wde);
        if constexpr (DAE) load_regs<ML, QM>(a.ae, tae, wae);
    }

    auto gb = [&](int c) -> long long { const long long b = b0 + c; return b < a.B ? b : a.B - 1; };
    const bool true_x = (a.flags & PSNODE_FLAG_INPUT_TRUE_X) != 0;
    const bool true_i = DAE && (a.flags & PSNODE_FLAG_INPUT_TRUE_I) != 0;
    const int nx = xd * TB;
    // dataset z | v row r of trajectory column c at grid point j
    auto zv_at = [&](long long j, int r, int c) -> float {
        const long long b = gb(c);
        return r < zd ? a.z.p[j * a.z.st + b * a.z.sb + r] : a.v.p[j * a.v.st + b * a.v.sb + (r - zd)];
    };
    // external row r (z | v | i order) of the DE input: columns n + xd + r (s - a0) and 2 n + xd + r (s)
    auto put_ext = [&](int r, int c, float v) {
        ext[r * TB + c] = v;
        lds[inDE + qi(SX + n + r, c)] = v - a0[(xd + r) * TB + c];
        lds[inDE + qi(SX + n + ne + r, c)] = v;
    };
    auto put_x = [&](int r, int c, float v) {
        lds[inDE + qi(r, c)] = v - a0[r * TB + c];
        lds[inDE + qi(xd + r, c)] = v;
    };

    // ---- per-trajectory constants, the constant and pad columns of both inputs, the initial state, z | v of grid point 0
    for (int idx = tid; idx < n * TB; idx += NT) {
        const int r = idx / TB, c = idx % TB;
        const float v = a.a0[gb(c) * n + r];
        a0[idx] = v;
        lds[inDE + qi(SX + r, c)] = v;
        if constexpr (DAE) lds[inAE + qi(SA + r, c)] = v;
    }
    for (int idx = tid; idx < de_k16(xd, n) * TB; idx += NT)       // pad columns (the others are overwritten below / at the top of every step)
        if (de_orig_col(idx / TB, xd, n) < 0) lds[inDE + qi(idx / TB, idx % TB)] = 0.0f;
    if constexpr (DAE)
        for (int idx = tid; idx < ae_k16(xd, nzv, n) * TB; idx += NT)
            if (ae_orig_col(idx / TB, xd, nzv, n) < 0) lds[inAE + qi(idx / TB, idx % TB)] = 0.0f;
    for (int idx = tid; idx < nx; idx += NT) {
        const int r = idx / TB, c = idx % TB;
        const long long b = gb(c);
        const float v = DAE ? a.x_init[b * xd + r] : a.x.p[b * a.x.sb + r];
        xcur[idx] = v;
        if (b0 + c < a.B) a.xo[b * xd + r] = v;
        if constexpr (DAE) lds[inAE + qi(r, c)] = true_x ? a.x.p[b * a.x.sb + r] : v;      // my_solvers.py:95
    }
    for (int idx = tid; idx < nzv * TB; idx += NT) {
        const float v = zv_at(0, idx / TB, idx % TB);
        zvn[idx] = v;
        if constexpr (DAE) lds[inAE + qi(xd + idx / TB, idx % TB)] = v;
    }
    lds_barrier();

    if constexpr (MODE == 0 && DAE) {       // the AE's a0 block: constant over the trajectory
        if (wv < tab_tiles(tae.dims[0])) cae = fold0<ML>(tae, lds, inAE, wv);
    }
    const int nstage = a.method == PSNODE_EULER ? 1 : (a.method == PSNODE_MIDPOINT ? 2 : 4);
    // look-ahead registers: clocks (threads < TB), the next step's event index, the next grid point's z | v
    float tc = 0.0f, tn = 0.0f;
    if (tid < TB) {
        tc = a.t.p[gb(tid) * a.t.sb];
        tn = a.T > 1 ? a.t.p[a.t.st + gb(tid) * a.t.sb] : tc;
    }
    // the next step's event index travels through a VECTOR load (every lane the same address): a scalar load would be waited for by the
    // very next LDS wait (SMEM returns out of order, so any lgkmcnt wait is lgkmcnt(0)) -- an L2 round trip at the top of every step
    const int* evp = a.ev ? a.ev : reinterpret_cast<const int*>(a.t.p);      // (a valid address when there are no events)
    int evn_v = (a.ev && a.T > 1) ? __builtin_nontemporal_load(evp) : -1;

    float pz[PF];

    // One MLP call site for every evaluation (the unrolled layer code exists once: it has to stay inside the instruction cache).  Slots of
    // step k: 0 = the AE head at an event (my_solvers.py:110), 1 .. S = the DE stages, S + 1 = the AE head at grid point k + 1
    // (my_solvers.py:95, 121); the pseudo-step k = -1 of the DAE is that last slot alone, for grid point 0.
    for (long long k = DAE ? -1 : 0; k + 1 < a.T; ++k) {
        K0_PROF(5)
        const int ev = k >= 0 ? __builtin_amdgcn_readfirstlane(evn_v) : -1;
        if (k >= 0) {
            // ---- this step's inputs (zero-order hold: the left grid point feeds every stage)
            if (tid < TB) dts[tid] = tn - tc;
            for (int idx = tid; idx < nzv * TB; idx += NT) {
                const int r = idx / TB, c = idx % TB;
                float v;
                if (ev >= 0) { const long long b = gb(c); v = r < zd ? a.zj[b * a.zjb + ev * a.zje + r] : a.vj[b * a.vjb + ev * a.vje + (r - zd)]; }
                else v = zvn[idx];
                put_ext(r, c, v);
                if constexpr (DAE) if (ev >= 0) lds[inAE + qi(xd + r, c)] = v;      // the event's AE evaluation sees the jumped rows
            }
            for (int idx = tid; idx < nx; idx += NT) {
                const int r = idx / TB, c = idx % TB;
                const float v = true_x ? a.x.p[k * a.x.st + gb(c) * a.x.sb + r] : xcur[idx];
                xsrc[idx] = v;
                put_x(r, c, v);
                if constexpr (DAE) if (ev >= 0) lds[inAE + qi(r, c)] = xcur[idx];   // ... and the computed state
            }
            if constexpr (DAE) {
                if (ev < 0)
                    for (int idx = tid; idx < id * TB; idx += NT) {
                        const int r = idx / TB, c = idx % TB;
                        put_ext(nzv + r, c, true_i ? a.i.p[k * a.i.st + gb(c) * a.i.sb + r] : icur[idx]);
                    }
            }
            // ---- look-ahead for grid point k + 1
            const long long k1 = k + 1, k2 = k + 2 < a.T ? k + 2 : a.T - 1;
            if (tid < TB) { tc = tn; tn = a.t.p[k2 * a.t.st + gb(tid) * a.t.sb]; }
            evn_v = (a.ev && k1 + 1 < a.T) ? evp[k1] : -1;
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                const int idx = tid + NT * j;
                const int ii = idx < nzv * TB ? idx : 0;
                pz[j] = nzv > 0 ? zv_at(k1, ii / TB, ii % TB) : 0.0f;
            }
            lds_barrier();
        }
        K0_PROF(0)
        const int e_end = DAE ? nstage + 1 : nstage;
        for (int e = k < 0 ? nstage + 1 : ((DAE && ev >= 0) ? 0 : 1); e <= e_end; ++e) {
            const bool is_ae = DAE && (e == 0 || e == nstage + 1);
            const bool ae_next = DAE && e == nstage;
            int f;
            if constexpr (MODE == 0 || MODE == 3) {
                if (e == 1) {       // the step's externals are in place (behind an event's AE evaluation too): the DE's per-step constant
                    const int nt0 = tab_tiles(tde.dims[0]);
                    if (wv < nt0) cde = fold0<ML>(tde, lds, inDE, wv);
                    if (MODE == 3 && wv + 4 < nt0) cde2 = fold0<ML>(tde, lds, inDE, wv + 4);
                }
            }
            if constexpr (MODE == 3) {
                f = mlp_regw<ML>(tde, lds, inDE, ping, pong, wdw, cde, cde2);
            } else if constexpr (MODE == 0) {      // two call sites: the operands are two different register sets
                if (is_ae) { if constexpr (DAE) f = mlp_reg<ML, QM>(tae, lds, inAE, ping, pong, wae, cae); else f = ping; }
                else f = mlp_reg<ML, QM>(tde, lds, inDE, ping, pong, wde, cde);
            } else {
                f = DAE ? mlp_eval<STREAM, ML>(pick_tab(is_ae, tae, tde), lds, is_ae ? inAE : inDE, ping, pong, pf, pick_tab(ae_next, tae, tde))
                        : mlp_eval<STREAM, ML>(tde, lds, inDE, ping, pong, pf, tde);
            }
            K0_PROF(2)
            if (is_ae) {
                if constexpr (DAE) {
                    for (int idx = tid; idx < id * TB; idx += NT) {
                        const int r = idx / TB, c = idx % TB;
                        const float v = lds[f + qi(r, c)];
                        icur[idx] = v;       // read back by the same thread (below, or at the top of the next step)
                        if (e == 0) put_ext(nzv + r, c, true_i ? a.i.p[k * a.i.st + gb(c) * a.i.sb + r] : v);
                        else if (b0 + c < a.B) a.io[((k + 1) * a.B + b0 + c) * id + r] = v;
                    }
                    if (e == 0) lds_barrier();
                }
                K0_PROF(4)
                continue;
            }
            // ---- stage s: ONE pass that forms the next stage's argument straight into the DE input (my_fixed_grid.py:15-18, 23-32, 38-51)
            //      or, behind the last stage, the new state, its output row, the AE input and the look-ahead rows
            const int s_ = e - 1;
            const bool final_stage = s_ + 1 == nstage;
            for (int idx = tid; idx < nx; idx += NT) {
                const int r = idx / TB, c = idx % TB;
                const float h = dts[c];
                const float x0 = xsrc[idx];
                const float ks = lds[f + qi(r, c)];
                float v;
                if (a.method == PSNODE_EULER) {
                    v = x0 + h * ks;
                } else if (a.method == PSNODE_MIDPOINT) {
                    v = s_ == 0 ? x0 + ks * (0.5f * h) : x0 + h * ks;
                } else {
                    if (s_ < 3) kbuf[s_ * nx + idx] = ks;
                    const float k1 = kbuf[idx];
                    if (s_ == 0) v = x0 + h * k1 * kOneThird;
                    else if (s_ == 1) v = x0 + h * (ks - k1 * kOneThird);
                    else if (s_ == 2) v = x0 + h * (k1 - kbuf[nx + idx] + ks);
                    else v = x0 + (k1 + 3.0f * (kbuf[nx + idx] + kbuf[2 * nx + idx]) + ks) * h * 0.125f;
                }
                if (!final_stage) {
                    put_x(r, c, v);
                } else {
                    xcur[idx] = v;
                    if (b0 + c < a.B) a.xo[((k + 1) * a.B + b0 + c) * xd + r] = v;
                    if constexpr (DAE) lds[inAE + qi(r, c)] = true_x ? a.x.p[(k + 1) * a.x.st + gb(c) * a.x.sb + r] : v;   // my_solvers.py:121
                }
            }
            if (final_stage) {
#pragma unroll
                for (int j = 0; j < PF; ++j) {
                    const int idx = tid + NT * j;
                    if (idx < nzv * TB) {
                        zvn[idx] = pz[j];
                        if constexpr (DAE) lds[inAE + qi(xd + idx / TB, idx % TB)] = pz[j];
                    }
                }
                for (int idx = tid + NT * PF; idx < nzv * TB; idx += NT) {
                    const float v = zv_at(k + 1, idx / TB, idx % TB);
                    zvn[idx] = v;
                    if constexpr (DAE) lds[inAE + qi(xd + idx / TB, idx % TB)] = v;
                }
            }
            lds_barrier();
            K0_PROF(3)
        }
    }
#ifdef PSNODE_K0_PROF
    if (blockIdx.x == 0 && tid == 0)
        printf("K0 phases (cycles of clock64, workgroup 0): step inputs %lld | mlp %lld | update %lld | AE head + output %lld | loop %lld\n", prof[0], prof[2],
               prof[3], prof[4], prof[5]);
#endif
#undef K0_PROF
}

}  // namespace

namespace {
struct PackArgs {
    int n;                       // layers in total (de then ae)
    int xd, nall, nzv;           // state dims, width of all_initial, z + v dims: the column orders of the two first layers
    int perm[2 * kMaxLayers];    // 0: columns as they are, 1: DE first layer (de_orig_col), 2: AE first layer (ae_orig_col),
                                 // 3: the TRANSPOSED matrix (tiles over the layer's inputs, contraction over its outputs; no bias)
    int K[2 * kMaxLayers], N[2 * kMaxLayers];
    const float* w[2 * kMaxLayers];
    const float* b[2 * kMaxLayers];
    float* img[2 * kMaxLayers];
};

// The MFMA image of every layer of both MLPs in one launch (blockIdx.y = layer): [tile nt][q][lane] f4, component c = W[16 nt + lane % 16]
// [16 q + 4 (lane / 16) + c] (zero outside the matrix), then the bias padded to 16 * tiles.
__global__ void pack_image_kernel(const PackArgs p) {
    const int l = blockIdx.y;
    const int K = p.K[l], N = p.N[l], perm = p.perm[l];
    if (perm == 3) {             // image of W^T: rows j < K, contraction kn < N
        const int S4 = (N + 15) >> 4, NTL = (K + 15) >> 4;
        const float* __restrict__ w = p.w[l];
        float* __restrict__ img = p.img[l];
        const int nw = NTL * S4 * 256;
        for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < nw + 16 * NTL; idx += gridDim.x * blockDim.x) {
            if (idx >= nw) { img[idx] = 0.0f; continue; }
            const int c = idx & 3, lane = (idx >> 2) & 63, q = (idx >> 8) % S4, nt = (idx >> 8) / S4;
            const int j = 16 * nt + (lane & 15), kn = 16 * q + 4 * (lane >> 4) + c;
            img[idx] = (j < K && kn < N) ? w[(size_t)kn * K + j] : 0.0f;
        }
        return;
    }
    const int K16 = perm == 1 ? de_k16(p.xd, p.nall) : (perm == 2 ? ae_k16(p.xd, p.nzv, p.nall) : up16(K));
    const int S4 = K16 >> 4, NTL = (N + 15) >> 4;
    const float* __restrict__ w = p.w[l];
    float* __restrict__ img = p.img[l];
    const int nw = NTL * S4 * 256;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < nw + 16 * NTL; idx += gridDim.x * blockDim.x) {
        if (idx >= nw) { const int j = idx - nw; img[idx] = j < N ? p.b[l][j] : 0.0f; continue; }
        const int c = idx & 3, lane = (idx >> 2) & 63, q = (idx >> 8) % S4, nt = (idx >> 8) / S4;
        const int j = 16 * nt + (lane & 15), kn = 16 * q + 4 * (lane >> 4) + c;
        const int k = perm == 1 ? de_orig_col(kn, p.xd, p.nall) : (perm == 2 ? ae_orig_col(kn, p.xd, p.nzv, p.nall) : (kn < K ? kn : -1));
        img[idx] = (j < N && k >= 0) ? w[(size_t)j * K + k] : 0.0f;
    }
}

void add_pack(PackArgs& p, const MlpDev& d, int perm0) {
    int k = d.in_dim;
    for (int l = 0; l < d.n_layers; ++l) {
        p.perm[p.n] = l ? 0 : perm0;
        p.K[p.n] = k;
        p.N[p.n] = d.out_dim[l];
        p.w[p.n] = d.w[l];
        p.b[p.n] = d.bias[l];
        p.img[p.n] = const_cast<float*>(d.wt[l]);
        ++p.n;
        k = d.out_dim[l];
    }
}

struct PackTArgs {
    int n;
    int K[2 * kMaxLayers], N[2 * kMaxLayers];
    const float* w[2 * kMaxLayers];
    float* wt[2 * kMaxLayers];
};

// wt[k][j] = w[j][k] for every layer of both MLPs in one launch (blockIdx.y = layer): the generic BACKWARD's weight layout.
__global__ void pack_transpose_kernel(const PackTArgs p) {
    const int l = blockIdx.y;
    const int K = p.K[l], N = p.N[l];
    const float* __restrict__ w = p.w[l];
    float* __restrict__ wt = p.wt[l];
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < K * N; idx += gridDim.x * blockDim.x) {
        const int k = idx / N, j = idx % N;
        wt[idx] = w[(size_t)j * K + k];
    }
}

void add_pack_t(PackTArgs& p, const MlpDev& d) {
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

// wt[k][j] = w[j][k] for every layer of one or two MLPs, one launch (generic backward)
hipError_t launch_pack_transpose(const MlpDev& de, const MlpDev* ae, hipStream_t stream) {
    PackTArgs p;
    p.n = 0;
    add_pack_t(p, de);
    if (ae) add_pack_t(p, *ae);
    hipLaunchKernelGGL(pack_transpose_kernel, dim3(8, p.n), dim3(256), 0, stream, p);
    return hipGetLastError();
}

// Plain (natural column order, with bias) and transposed images of one MLP's layers for the generic backward's register path: img[l] /
// imgT[l] must hold generic_image_floats(K_l, N_l) / generic_image_floats(N_l, K_l) floats.
hipError_t launch_pack_plain_images(const MlpDev& m, float* const* img, float* const* imgT, hipStream_t stream) {
    PackArgs p;
    p.n = 0; p.xd = 0; p.nall = 0; p.nzv = 0;
    int k = m.in_dim;
    for (int l = 0; l < m.n_layers; ++l) {
        for (int t = 0; t < 2; ++t) {
            p.perm[p.n] = t ? 3 : 0;
            p.K[p.n] = k; p.N[p.n] = m.out_dim[l];
            p.w[p.n] = m.w[l]; p.b[p.n] = m.bias[l];
            p.img[p.n] = t ? imgT[l] : img[l];
            ++p.n;
        }
        k = m.out_dim[l];
    }
    hipLaunchKernelGGL(pack_image_kernel, dim3(16, p.n), dim3(256), 0, stream, p);
    return hipGetLastError();
}

// floats of the generic forward's workspace segment of one layer (psnode_capi.hip carves the workspace with it)
size_t generic_image_floats(int K, int N) { return image_floats(K, N); }

// the MFMA images of one or two MLPs (d.wt[l] = the layer's segment, generic_image_floats each), one launch
hipError_t launch_pack_image(const MlpDev& de, const MlpDev* ae, int xd, int n, int nzv, hipStream_t stream) {
    PackArgs p;
    p.n = 0;
    p.xd = xd; p.nall = n; p.nzv = nzv;
    add_pack(p, de, 1);
    if (ae) add_pack(p, *ae, 2);
    hipLaunchKernelGGL(pack_image_kernel, dim3(16, p.n), dim3(256), 0, stream, p);
    return hipGetLastError();
}

// LDS of the kernel without any resident image: activations, state, biases
size_t generic_lds_bytes(const IntegrateDev& a, bool dae) {
    const int vd = dae ? a.vd : 0, id = dae ? a.id : 0;
    const int n = a.xd + a.zd + vd + id;
    const int nzv = a.zd + vd;
    // DE input, AE input, two layer-output buffers, a0, ext, xcur + xsrc, kbuf, icur, zvn, dts
    const size_t rows = (size_t)de_k16(a.xd, n) + (dae ? ae_k16(a.xd, nzv, n) : 0) + 2 * (size_t)up16(a.maxo) + n + (n - a.xd) + 2 * (size_t)a.xd +
                        4 * (size_t)a.xd + id + nzv + 1;
    return (rows * TB + generic_bias_floats(a, dae)) * sizeof(float);
}

// Which layers' images become resident: greedy in layer order, DE (evaluated once per stage) before AE (once per step).  Returns the
// launch's LDS bytes; `mask`: bit l = DE layer l, bit 8 + l = AE layer l.
// Register mode (mlp_reg): layers of at most four tiles (one per wave), the first layers' LEADING blocks (DE: 2 x_dim columns, AE: x | z | v)
// within 16 * QM columns -- the rest of a first layer is folded (fold0), so z / v / i may be as wide as LDS holds; at most four layers per
// MLP for the DAE (two MLPs: 160 operand registers at QM 8), eight for the ODE (144).  Returns QM (4 or 8), or 0.
int generic_reg_mode(const IntegrateDev& a, bool dae) {
    // the leading blocks of the two first layers (state columns / x | z | v): what the registers hold of layer 0
    const int lead_de = de_sx(a.xd), lead_ae = dae ? ae_sa(a.xd, a.zd + a.vd) : 0;
    if (lead_de > 128 || lead_ae > 128) return 0;
    for (int m = 0; m < (dae ? 2 : 1); ++m) {
        const MlpDev& d = m ? a.ae : a.de;
        if (d.n_layers > (dae ? 4 : kMaxLayers)) return 0;
        for (int l = 0; l < d.n_layers; ++l)
            if (d.out_dim[l] > 64) return 0;
    }
    return (lead_de <= 64 && lead_ae <= 64) ? 4 : 8;
}

// Wide register form (mlp_regw; ODE): 2 .. 4 layers, hidden layers within 128 units, every contraction within 128 columns, at most 64 outputs
bool generic_wide_mode(const IntegrateDev& a, bool dae) {
    if (dae || a.de.n_layers < 2 || a.de.n_layers > 4 || de_sx(a.xd) > 128) return false;
    for (int l = 0; l < a.de.n_layers; ++l)
        if (a.de.out_dim[l] > (l + 1 == a.de.n_layers ? 64 : 128)) return false;
    return true;
}

size_t generic_plan(const IntegrateDev& a, bool dae, unsigned& mask) {
    size_t bytes = generic_lds_bytes(a, dae);
    mask = 0;
    if (generic_reg_mode(a, dae) || generic_wide_mode(a, dae)) return bytes;
    const size_t limit = 160 * 1024;
    for (int m = 0; m < (dae ? 2 : 1); ++m) {
        const MlpDev& d = m ? a.ae : a.de;
        const int vd_ = dae ? a.vd : 0, n_ = a.xd + a.zd + vd_ + (dae ? a.id : 0);
        int K = m ? ae_k16(a.xd, a.zd + vd_, n_) : de_k16(a.xd, n_);
        for (int l = 0; l < d.n_layers; ++l) {
            const size_t img = (size_t)up16(d.out_dim[l]) * up16(K) * sizeof(float);
            if (bytes + img <= limit) { bytes += img; mask |= 1u << (8 * m + l); }
            K = d.out_dim[l];
        }
    }
    return bytes;
}

hipError_t launch_generic(const IntegrateDev& a_in, bool dae, hipStream_t stream_) {
    IntegrateDev a = a_in;
    const size_t lds = generic_plan(a, dae, a.k0_res);
    const unsigned grid = (unsigned)((a.B + TB - 1) / TB);
    unsigned all = (1u << a.de.n_layers) - 1u;
    if (dae) all |= ((1u << a.ae.n_layers) - 1u) << 8;
    const bool stream = a.k0_res != all;          // some layer's image does not fit LDS
    // Up to one workgroup per CU in the launch: ask for more than half a CU's LDS, so that the dispatcher cannot put two workgroups on one CU
    // (two waves per SIMD sharing the MFMA pipe) while other CUs stay empty.
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    const size_t lds_launch = (grid <= (unsigned)cus && lds < 81 * 1024) ? 81 * 1024 : lds;
    auto go = [&](auto kern) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_launch);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds_launch, stream_, a);
        return hipGetLastError();
    };
    const int qm = generic_reg_mode(a, dae);
    const bool deep = a.de.n_layers > 4 || (dae && a.ae.n_layers > 4);      // the layer loop is unrolled 4 or kMaxLayers times
    if (qm && deep) return qm == 4 ? go(&generic_kernel<false, 0, kMaxLayers, 4>) : go(&generic_kernel<false, 0, kMaxLayers, 8>);
    if (qm == 4) return dae ? go(&generic_kernel<true, 0, 4, 4>) : go(&generic_kernel<false, 0, 4, 4>);
    if (qm == 8) return dae ? go(&generic_kernel<true, 0, 4, 8>) : go(&generic_kernel<false, 0, 4, 8>);
    if (generic_wide_mode(a, dae)) return go(&generic_kernel<false, 3, 4>);
    if (deep) {
        if (dae) return stream ? go(&generic_kernel<true, 2, kMaxLayers>) : go(&generic_kernel<true, 1, kMaxLayers>);
        return stream ? go(&generic_kernel<false, 2, kMaxLayers>) : go(&generic_kernel<false, 1, kMaxLayers>);
    }
    if (dae) return stream ? go(&generic_kernel<true, 2, 4>) : go(&generic_kernel<true, 1, 4>);
    return stream ? go(&generic_kernel<false, 2, 4>) : go(&generic_kernel<false, 1, 4>);
}

}  // namespace psnode
