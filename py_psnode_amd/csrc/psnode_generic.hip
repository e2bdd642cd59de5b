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
// floats of one layer's image: N16 x K16 weights + N16 biases
__host__ __device__ constexpr size_t image_floats(int K, int N) { return (size_t)up16(N) * up16(K) + up16(N); }
// floats of the padded biases of every layer (LDS region behind the kernel's state)
__host__ __device__ inline int generic_bias_floats(const IntegrateDev& a, bool dae) {
    int tot = 0;
    for (int l = 0; l < a.de.n_layers; ++l) tot += up16(a.de.out_dim[l]);
    if (dae) for (int l = 0; l < a.ae.n_layers; ++l) tot += up16(a.ae.out_dim[l]);
    return tot;
}

// One MLP as the time loop sees it: wave-uniform scalars, built once per launch so that no kernel-argument load (a scalar-cache round trip,
// and an lgkmcnt wait that also drains the LDS queue) sits between two chunks.  The layer loop of mlp_eval is fully unrolled over
// kMaxLayers, which makes every index below a constant.
struct Tab {
    int L;
    unsigned dims[kMaxLayers];     // resident << 31 | quads of the contraction (ceil(K / 16)) << 16 | output tiles (ceil(N / 16))
    unsigned off[kMaxLayers];      // streamed layer: f4 offset of its image from `base`; resident layer: FLOAT offset of its copy in LDS
    unsigned boff[kMaxLayers];     // float offset of the padded bias in LDS
    const f4* base;                // image of layer 0 (workspace)
    unsigned first_off;            // this wave's first STREAMED chunk of an evaluation: f4 offset ...
    int first_q;                   // ... and the quads of that tile (0: the wave owns no tile in a streamed layer of this MLP)
};
__device__ __forceinline__ int tab_tiles(unsigned d) { return (int)(d & 0xffffu); }
__device__ __forceinline__ int tab_quads(unsigned d) { return (int)((d >> 16) & 0x7fffu); }
__device__ __forceinline__ bool tab_res(unsigned d) { return (d >> 31) != 0; }

// `res`: bit l = layer l's image is resident in LDS; `bias_at` / `img_at`: running float offsets of the LDS regions (advanced)
__device__ __forceinline__ Tab make_tab(const MlpDev& m, int w, unsigned res, unsigned& bias_at, unsigned& img_at) {
    Tab t;
    t.L = m.n_layers;
    t.base = reinterpret_cast<const f4*>(m.wt[0]);
    t.first_off = 0; t.first_q = 0;
#pragma unroll
    for (int l = 0; l < kMaxLayers; ++l) {
        const int K = l ? m.out_dim[l - 1] : m.in_dim, N = m.out_dim[l];
        const unsigned S4 = (K + 15) >> 4, NTL = (N + 15) >> 4;
        const bool on = l < m.n_layers, r = on && ((res >> l) & 1u);
        t.dims[l] = on ? ((r ? 1u << 31 : 0u) | S4 << 16 | NTL) : 0u;
        t.off[l] = !on ? 0u : (r ? img_at : (unsigned)((m.wt[l] - m.wt[0]) >> 2));
        t.boff[l] = bias_at;
        if (on) bias_at += 16u * NTL;
        if (r) img_at += 256u * NTL * S4;
    }
#pragma unroll
    for (int l = kMaxLayers - 1; l >= 0; --l)
        if (l < m.n_layers && !tab_res(t.dims[l]) && tab_tiles(t.dims[l]) > w) {
            t.first_off = t.off[l] + (unsigned)w * tab_quads(t.dims[l]) * 64u;
            t.first_q = tab_quads(t.dims[l]);
        }
    return t;
}

// copies the biases (always) and the resident images into LDS; no barrier
__device__ __forceinline__ void load_resident(const MlpDev& m, const Tab& t, float* lds) {
#pragma unroll
    for (int l = 0; l < kMaxLayers; ++l) {
        if (l >= t.L) break;
        const int NTL = tab_tiles(t.dims[l]), S4 = tab_quads(t.dims[l]);
        const float* __restrict__ src = m.wt[l];
        for (int i = threadIdx.x; i < 16 * NTL; i += NT) lds[t.boff[l] + i] = src[(size_t)NTL * S4 * 256 + i];
        if (tab_res(t.dims[l]))
            for (int i = threadIdx.x; i < NTL * S4 * 64; i += NT) reinterpret_cast<f4*>(lds + t.off[l])[i] = reinterpret_cast<const f4*>(src)[i];
    }
}

// A-operand prefetch carried from one MLP evaluation into the next: the first chunk (up to four f4 = 16 MFMA steps) of the wave's first tile
// in a streamed layer of the MLP whose image starts at `tag`.  A wrong guess only costs the latency of one L2 read.
struct Pref {
    f4 a[4];
    const f4* tag;
};

// MLP over the TB columns.  `in` / `out`: float offsets of the quad-row buffers in `lds`; returns the offset of the buffer that holds the
// last layer's output.  Ends with a barrier.  Resident layers read both MFMA operands from LDS.  Streamed layers: the A operands run one
// chunk ahead of the MFMAs that use them, across tile, layer and -- through `pf` -- evaluation boundaries: with one wave per SIMD nothing
// else hides the L2 latency.  `nx`: the MLP evaluated after this one.
__device__ __forceinline__ int mlp_eval(const Tab& T, float* lds, int in, int out, Pref& pf, const Tab& nx) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // chunk = quads q0 .. q0 + 3 of one tile, clamped inside the tile's run of the image (in bounds, unused beyond the tile's quads).  The
    // address comes out of scalar selects and the four loads are unconditional straight-line code: a load inside a conditional block gets
    // its result copied (and waited for) at the end of that block, in front of the MFMAs it should overlap.
    auto fetch = [&](const f4* base, unsigned off, int rem, f4 (&a)[4]) {
        const f4* __restrict__ A = base + off + lane;
#pragma unroll
        for (int c = 0; c < 4; ++c) a[c] = A[(c < rem ? c : rem - 1) * 64];
    };
    auto mfma_quad = [&](const f4 av, const f4 bv, f4& acc) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[e], acc, 0, 0, 0);
    };
    if (pf.tag != T.base && T.first_q > 0) fetch(T.base, T.first_off, T.first_q, pf.a);
#pragma unroll
    for (int l = 0; l < kMaxLayers; ++l) {
        if (l >= T.L) break;
        const int S4 = tab_quads(T.dims[l]), NTL = tab_tiles(T.dims[l]);
        const bool last = (l + 1 == T.L);
        const f4* bq = reinterpret_cast<const f4*>(lds + in) + lane;
        const f4* b16 = reinterpret_cast<const f4*>(lds + T.boff[l]) + (lane >> 4);
        if (tab_res(T.dims[l])) {
            const f4* aq = reinterpret_cast<const f4*>(lds + T.off[l]) + lane;
            for (int nt = w; nt < NTL; nt += 4) {
                const f4* at = aq + nt * S4 * 64;
                f4 acc = f4{0.f, 0.f, 0.f, 0.f};
                int q0 = 0;
                for (; q0 + 4 <= S4; q0 += 4) {
                    f4 av[4], bv[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) { av[c] = at[(q0 + c) * 64]; bv[c] = bq[(q0 + c) * 64]; }
                    __builtin_amdgcn_sched_barrier(0);      // all eight reads in flight before the first MFMA
#pragma unroll
                    for (int c = 0; c < 4; ++c) mfma_quad(av[c], bv[c], acc);
                }
                for (; q0 < S4; ++q0) mfma_quad(at[q0 * 64], bq[q0 * 64], acc);
                acc = acc + b16[4 * nt];
                if (!last) acc = elu_quad(acc);
                reinterpret_cast<f4*>(lds + out)[nt * 64 + lane] = acc;
            }
        } else {
            // where this wave's A stream continues behind its last chunk of layer l: its first tile of a later streamed layer, else of the
            // next evaluation
            const f4* tbase = nx.base;
            unsigned toff = nx.first_off;
            int tq = nx.first_q > 0 ? nx.first_q : 1;
#pragma unroll
            for (int nl = kMaxLayers - 1; nl > l; --nl)
                if (nl < T.L && !tab_res(T.dims[nl]) && tab_tiles(T.dims[nl]) > w) {
                    tbase = T.base; toff = T.off[nl] + (unsigned)w * tab_quads(T.dims[nl]) * 64u; tq = tab_quads(T.dims[nl]);
                }
            for (int nt = w; nt < NTL; nt += 4) {
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
                acc = acc + b16[4 * nt];
                if (!last) acc = elu_quad(acc);
                reinterpret_cast<f4*>(lds + out)[nt * 64 + lane] = acc;
            }
        }
        lds_barrier();
        const int tmp = in; in = out; out = tmp;
    }
    if (T.first_q > 0) pf.tag = nx.base;        // a wave without a streamed tile in T has fetched nothing
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

    constexpr int actA = 0;                    // quad-row activation buffers (float offsets into lds): the MLP inputs and every second layer
    const int actB = up16(a.maxw) * TB;        // ping-pong partner: holds layer outputs only -> maxo rows
    float* a0 = lds + actB + up16(a.maxo) * TB;    // [n][TB]
    float* ext = a0 + n * TB;          // [ne][TB] z | v | i fed to the DE stages of this step
    float* xcur = ext + ne * TB;       // [xd][TB] running state
    float* xsrc = xcur + xd * TB;      // [xd][TB] start of this step (xcur, or dataset x under teacher forcing)
    float* xst = xsrc + xd * TB;       // [xd][TB] stage argument
    float* kbuf = xst + xd * TB;       // [4][xd][TB]
    float* icur = kbuf + 4 * xd * TB;  // [id][TB]
    float* dts = icur + id * TB;       // [TB]

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
    const Tab tde = make_tab(a.de, wv, a.k0_res & 0xffu, bias_at, img_at);
    const Tab tae = DAE ? make_tab(a.ae, wv, (a.k0_res >> 8) & 0xffu, bias_at, img_at) : tde;
    load_resident(a.de, tde, lds);
    if constexpr (DAE) load_resident(a.ae, tae, lds);
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
    lds_barrier();

    // AE head g(x; z, v) -> icur.  jx >= 0: x from the dataset at grid point jx, else xcur.
    // jzv >= 0: z, v from the dataset at grid point jzv, else the (possibly jumped) rows of `ext`.
    auto ae_eval = [&](long long jx, long long jzv) {
        if constexpr (DAE) {
            const int m = n + xd + zd + vd;
            for (int idx = tid; idx < up16(m) * TB; idx += NT) {
                const int r = idx / TB, c = idx % TB;
                const long long b = gb(c);
                float v;
                if (r >= m) v = 0.0f;            // pad columns of the first layer's K
                else if (r < n) v = a0[idx];
                else if (r < n + xd) v = jx >= 0 ? a.x.p[jx * a.x.st + b * a.x.sb + (r - n)] : xcur[(r - n) * TB + c];
                else if (r < n + xd + zd) v = jzv >= 0 ? a.z.p[jzv * a.z.st + b * a.z.sb + (r - n - xd)] : ext[(r - n - xd) * TB + c];
                else v = jzv >= 0 ? a.v.p[jzv * a.v.st + b * a.v.sb + (r - n - xd - zd)] : ext[(r - n - xd) * TB + c];
                lds[actA + qi(r, c)] = v;
            }
            lds_barrier();
            const int out = mlp_eval(tae, lds, actA, actB, pf, tde);
            for (int idx = tid; idx < id * TB; idx += NT) icur[idx] = lds[out + qi(idx / TB, idx % TB)];
            lds_barrier();
        }
    };

    if constexpr (DAE) {
        ae_eval(true_x ? 0 : -1, 0);   // my_solvers.py:95
        for (int idx = tid; idx < id * TB; idx += NT) {
            const int r = idx / TB, c = idx % TB;
            if (b0 + c < a.B) a.io[(b0 + c) * id + r] = icur[idx];
        }
    }

    // DE right-hand side at xst with this step's frozen externals -> offset of the quad-row buffer holding [xd] columns
    auto de_eval = [&](bool next_ae) -> int {
        for (int idx = tid; idx < n * TB; idx += NT) {
            const int r = idx / TB, c = idx % TB;
            const float s = r < xd ? xst[idx] : ext[idx - nx];
            const float i0 = a0[idx];
            lds[actA + qi(r, c)] = i0;
            lds[actA + qi(n + r, c)] = s - i0;
            lds[actA + qi(2 * n + r, c)] = s;
        }
        for (int idx = 3 * n * TB + tid; idx < up16(3 * n) * TB; idx += NT) lds[actA + qi(idx / TB, idx % TB)] = 0.0f;   // pad columns
        lds_barrier();
        K0_PROF(1)
        const int r_ = next_ae ? mlp_eval(tde, lds, actA, actB, pf, tae) : mlp_eval(tde, lds, actA, actB, pf, tde);
        K0_PROF(2)
        return r_;
    };

    const int nstage = a.method == PSNODE_EULER ? 1 : (a.method == PSNODE_MIDPOINT ? 2 : 4);

    for (long long k = 0; k + 1 < a.T; ++k) {
        K0_PROF(5)
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
        lds_barrier();
        K0_PROF(0)
        if constexpr (DAE) {
            if (ev >= 0) ae_eval(-1, -1);   // my_solvers.py:110: i0 = i_func(x0, z0_jump, v0_jump)
            for (int idx = tid; idx < id * TB; idx += NT) {
                const int r = idx / TB, c = idx % TB;
                ext[(zd + vd) * TB + idx] = true_i ? a.i.p[k * a.i.st + gb(c) * a.i.sb + r] : icur[idx];
            }
            lds_barrier();
        }

        // ---- stages (my_fixed_grid.py:15-18, 23-32, 38-51)
        for (int s = 0; s < nstage; ++s) {
            const int f = de_eval(DAE && s + 1 == nstage);
            for (int idx = tid; idx < nx; idx += NT) {
                const float h = dts[idx % TB];
                const float x0 = xsrc[idx];
                const float ks = lds[f + qi(idx / TB, idx % TB)];
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
            lds_barrier();
            K0_PROF(3)
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
        K0_PROF(4)
    }
#ifdef PSNODE_K0_PROF
    if (blockIdx.x == 0 && tid == 0)
        printf("K0 phases (cycles of clock64, workgroup 0): step inputs %lld | build %lld | mlp %lld | update %lld | output %lld | loop %lld\n", prof[0], prof[1],
               prof[2], prof[3], prof[4], prof[5]);
#endif
#undef K0_PROF
}

}  // namespace

namespace {
struct PackArgs {
    int n;                       // layers in total (de then ae)
    int K[2 * kMaxLayers], N[2 * kMaxLayers];
    const float* w[2 * kMaxLayers];
    const float* b[2 * kMaxLayers];
    float* img[2 * kMaxLayers];
};

// The MFMA image of every layer of both MLPs in one launch (blockIdx.y = layer): [tile nt][q][lane] f4, component c = W[16 nt + lane % 16]
// [16 q + 4 (lane / 16) + c] (zero outside the matrix), then the bias padded to 16 * tiles.
__global__ void pack_image_kernel(const PackArgs p) {
    const int l = blockIdx.y;
    const int K = p.K[l], N = p.N[l];
    const int S4 = (K + 15) >> 4, NTL = (N + 15) >> 4;
    const float* __restrict__ w = p.w[l];
    float* __restrict__ img = p.img[l];
    const int nw = NTL * S4 * 256;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < nw + 16 * NTL; idx += gridDim.x * blockDim.x) {
        if (idx >= nw) { const int j = idx - nw; img[idx] = j < N ? p.b[l][j] : 0.0f; continue; }
        const int c = idx & 3, lane = (idx >> 2) & 63, q = (idx >> 8) % S4, nt = (idx >> 8) / S4;
        const int j = 16 * nt + (lane & 15), k = 16 * q + 4 * (lane >> 4) + c;
        img[idx] = (j < N && k < K) ? w[(size_t)j * K + k] : 0.0f;
    }
}

void add_pack(PackArgs& p, const MlpDev& d) {
    int k = d.in_dim;
    for (int l = 0; l < d.n_layers; ++l) {
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

// floats of the generic forward's workspace segment of one layer (psnode_capi.hip carves the workspace with it)
size_t generic_image_floats(int K, int N) { return image_floats(K, N); }

// the MFMA images of one or two MLPs (d.wt[l] = the layer's segment, generic_image_floats each), one launch
hipError_t launch_pack_image(const MlpDev& de, const MlpDev* ae, hipStream_t stream) {
    PackArgs p;
    p.n = 0;
    add_pack(p, de);
    if (ae) add_pack(p, *ae);
    hipLaunchKernelGGL(pack_image_kernel, dim3(16, p.n), dim3(256), 0, stream, p);
    return hipGetLastError();
}

// LDS of the kernel without any resident image: activations, state, biases
size_t generic_lds_bytes(const IntegrateDev& a, bool dae) {
    const int vd = dae ? a.vd : 0, id = dae ? a.id : 0;
    const int n = a.xd + a.zd + vd + id;
    const size_t rows = (size_t)up16(a.maxw) + up16(a.maxo) + n + (n - a.xd) + 3 * (size_t)a.xd + 4 * (size_t)a.xd + id + 1;
    return (rows * TB + generic_bias_floats(a, dae)) * sizeof(float);
}

// Which layers' images become resident: greedy in layer order, DE (evaluated once per stage) before AE (once per step).  Returns the
// launch's LDS bytes; `mask`: bit l = DE layer l, bit 8 + l = AE layer l.
size_t generic_plan(const IntegrateDev& a, bool dae, unsigned& mask) {
    size_t bytes = generic_lds_bytes(a, dae);
    mask = 0;
    const size_t limit = 160 * 1024;
    for (int m = 0; m < (dae ? 2 : 1); ++m) {
        const MlpDev& d = m ? a.ae : a.de;
        int K = d.in_dim;
        for (int l = 0; l < d.n_layers; ++l) {
            const size_t img = (size_t)up16(d.out_dim[l]) * up16(K) * sizeof(float);
            if (bytes + img <= limit) { bytes += img; mask |= 1u << (8 * m + l); }
            K = d.out_dim[l];
        }
    }
    return bytes;
}

hipError_t launch_generic(const IntegrateDev& a_in, bool dae, hipStream_t stream) {
    IntegrateDev a = a_in;
    const size_t lds = generic_plan(a, dae, a.k0_res);
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
