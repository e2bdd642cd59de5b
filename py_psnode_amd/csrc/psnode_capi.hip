// C ABI of libpsnode_hip.so (see include/psnode_hip.h): argument validation, workspace carving,
// weight packing and kernel dispatch.  Everything is enqueued on the caller's stream.
#include <stdio.h>
#include <string.h>

#include "psnode_common.h"

namespace psnode {
namespace {

constexpr size_t kAlignFloats = 64;   // 256-byte alignment of every workspace segment

inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

size_t generic_floats(const psnode_mlp_f32* m) {
    if (!m) return 0;
    size_t tot = 0;
    int k = m->in_dim;
    for (int l = 0; l < m->n_layers; ++l) {
        tot += round_up(generic_image_floats(k, m->out_dim[l]), kAlignFloats);
        k = m->out_dim[l];
    }
    return tot;
}

int check_mlp(const psnode_mlp_f32& m, int want_in, int want_out) {
    if (m.n_layers < 1 || m.n_layers > kMaxLayers) return PSNODE_ERR_DIMS;
    if (m.in_dim != want_in || m.in_dim < 1) return PSNODE_ERR_DIMS;
    if (m.in_dim > PSNODE_MAX_IN_WIDTH) return PSNODE_ERR_UNSUPPORTED;    // consistent, but wider than any kernel takes
    for (int l = 0; l < m.n_layers; ++l) {
        if (m.out_dim[l] < 1) return PSNODE_ERR_DIMS;
        if (m.out_dim[l] > PSNODE_MAX_WIDTH) return PSNODE_ERR_UNSUPPORTED;
        if (!m.weight[l] || !m.bias[l]) return PSNODE_ERR_NULL;
    }
    if (m.out_dim[m.n_layers - 1] != want_out) return PSNODE_ERR_DIMS;
    return PSNODE_OK;
}

// Fills `d` and assigns the transposed-weight segments; returns the next free float of the workspace.
float* bind_mlp(const psnode_mlp_f32& m, MlpDev& d, float* ws) {
    d.n_layers = m.n_layers;
    d.in_dim = m.in_dim;
    int k = m.in_dim;
    for (int l = 0; l < m.n_layers; ++l) {
        d.out_dim[l] = m.out_dim[l];
        d.w[l] = m.weight[l];
        d.bias[l] = m.bias[l];
        d.wt[l] = ws;
        ws += round_up(generic_image_floats(k, m.out_dim[l]), kAlignFloats);
        k = m.out_dim[l];
    }
    return ws;
}

int max_width(const psnode_mlp_f32& m) {
    int w = m.in_dim;
    for (int l = 0; l < m.n_layers; ++l) w = m.out_dim[l] > w ? m.out_dim[l] : w;
    return w;
}

__global__ void event_table_kernel(long long n_steps, const float* clock, long long stride_k, const float* ev_times,
                                   long long stride_e, int n_events, int* event_idx, int* dup_flag) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_steps) return;
    const float tk = clock[k * stride_k];
    int hit = -1, cnt = 0;
    for (int e = 0; e < n_events; ++e) {
        if (ev_times[e * stride_e] == tk) {   // exact equality, like Tensor.__contains__ (neural_base.py:54)
            if (hit < 0) hit = e;
            ++cnt;
        }
    }
    event_idx[k] = hit;
    if (cnt > 1 && dup_flag) *dup_flag = 1;
}

ViewDev view(const psnode_view_f32& v) { return ViewDev{v.ptr, v.stride_t, v.stride_b}; }

int dispatch(IntegrateDev& d, bool dae, int kernel, const psnode_mlp_f32* de, const psnode_mlp_f32* ae, void* workspace,
             size_t workspace_bytes, hipStream_t stream) {
    if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 255u)) return PSNODE_ERR_WORKSPACE;
    if (workspace_bytes < psnode_workspace_bytes(de, ae)) return PSNODE_ERR_WORKSPACE;
    float* ws = static_cast<float*>(workspace);
    ws = bind_mlp(*de, d.de, ws);
    if (dae) ws = bind_mlp(*ae, d.ae, ws);
    d.maxw = max_width(*de);
    if (dae && max_width(*ae) > d.maxw) d.maxw = max_width(*ae);
    d.maxo = 1;
    for (int l = 0; l < de->n_layers; ++l) d.maxo = de->out_dim[l] > d.maxo ? de->out_dim[l] : d.maxo;
    if (dae) for (int l = 0; l < ae->n_layers; ++l) d.maxo = ae->out_dim[l] > d.maxo ? ae->out_dim[l] : d.maxo;

    const bool has_mfma = dae ? mfma_dae_supported(d) : mfma_ode_supported(d);
    const bool want_mfma = kernel == PSNODE_KERNEL_MFMA || kernel == PSNODE_KERNEL_MFMA_TILE || kernel == PSNODE_KERNEL_MFMA_WAVE;
    if (want_mfma && !has_mfma) return PSNODE_ERR_UNSUPPORTED;
    if (kernel == PSNODE_KERNEL_MFMA_WAVE && !(dae ? mfma_x_dae_supported(d) : mfma_x_ode_supported(d))) return PSNODE_ERR_UNSUPPORTED;
    d.kern = kernel;
    const bool use_mfma = has_mfma && kernel != PSNODE_KERNEL_GENERIC;
    if (d.T < 1 || d.B < 1) return PSNODE_ERR_DIMS;

    hipError_t e;
    if (use_mfma) {
        e = launch_mfma(d, dae, ws, stream);
    } else {
        if (generic_lds_bytes(d, dae) > 160 * 1024) return PSNODE_ERR_UNSUPPORTED;
        e = launch_pack_image(d.de, dae ? &d.ae : nullptr, d.xd, d.xd + d.zd + (dae ? d.vd + d.id : 0), d.zd + (dae ? d.vd : 0), stream);
        if (e == hipSuccess) e = launch_generic(d, dae, stream);
    }
    return e == hipSuccess ? PSNODE_OK : PSNODE_ERR_HIP;
}

int fill_ode(const psnode_ode_args_f32* a, IntegrateDev& d) {
    if (!a) return PSNODE_ERR_NULL;
    if (a->method < PSNODE_EULER || a->method > PSNODE_RK4_38) return PSNODE_ERR_METHOD;
    if (a->x_dim < 1 || a->z_dim < 0 || a->T < 1 || a->B < 1) return PSNODE_ERR_DIMS;
    const int n = a->x_dim + a->z_dim;
    int rc = check_mlp(a->de, 3 * n, a->x_dim);
    if (rc) return rc;
    if (!a->t.ptr || !a->x.ptr || !a->all_initial || !a->x_out) return PSNODE_ERR_NULL;
    if (a->z_dim > 0 && !a->z.ptr) return PSNODE_ERR_NULL;
    if (a->event_idx && a->z_dim > 0 && !a->z_jump) return PSNODE_ERR_NULL;
    memset(&d, 0, sizeof(d));
    d.method = a->method;
    d.flags = a->flags & PSNODE_FLAG_INPUT_TRUE_X;
    d.xd = a->x_dim;
    d.zd = a->z_dim;
    d.T = a->T;
    d.B = a->B;
    d.t = view(a->t);
    d.x = view(a->x);
    d.z = view(a->z);
    d.a0 = a->all_initial;
    d.ev = a->event_idx;
    d.zj = a->z_jump;
    d.zjb = a->zj_stride_b;
    d.zje = a->zj_stride_e;
    d.xo = a->x_out;
    if ((a->save_act != nullptr) != (a->save_xstage != nullptr)) return PSNODE_ERR_NULL;
    d.sact = a->save_act;
    d.sxst = a->save_xstage;
    return PSNODE_OK;
}

int fill_dae(const psnode_dae_args_f32* a, IntegrateDev& d) {
    if (!a) return PSNODE_ERR_NULL;
    if (a->method < PSNODE_EULER || a->method > PSNODE_RK4_38) return PSNODE_ERR_METHOD;
    if (a->x_dim < 1 || a->z_dim < 0 || a->v_dim < 0 || a->i_dim < 1 || a->T < 1 || a->B < 1) return PSNODE_ERR_DIMS;
    const int n = a->x_dim + a->z_dim + a->v_dim + a->i_dim;
    int rc = check_mlp(a->de, 3 * n, a->x_dim);
    if (rc) return rc;
    rc = check_mlp(a->ae, n + a->x_dim + a->z_dim + a->v_dim, a->i_dim);
    if (rc) return rc;
    if (!a->t.ptr || !a->x_init || !a->all_initial || !a->x_out || !a->i_out) return PSNODE_ERR_NULL;
    if ((a->z_dim > 0 && !a->z.ptr) || (a->v_dim > 0 && !a->v.ptr)) return PSNODE_ERR_NULL;
    if ((a->flags & PSNODE_FLAG_INPUT_TRUE_X) && !a->x.ptr) return PSNODE_ERR_NULL;
    if ((a->flags & PSNODE_FLAG_INPUT_TRUE_I) && !a->i.ptr) return PSNODE_ERR_NULL;
    if (a->event_idx && ((a->z_dim > 0 && !a->z_jump) || (a->v_dim > 0 && !a->v_jump))) return PSNODE_ERR_NULL;
    memset(&d, 0, sizeof(d));
    d.method = a->method;
    d.flags = a->flags & (PSNODE_FLAG_INPUT_TRUE_X | PSNODE_FLAG_INPUT_TRUE_I);
    d.xd = a->x_dim;
    d.zd = a->z_dim;
    d.vd = a->v_dim;
    d.id = a->i_dim;
    d.T = a->T;
    d.B = a->B;
    d.t = view(a->t);
    d.x = view(a->x);
    d.z = view(a->z);
    d.v = view(a->v);
    d.i = view(a->i);
    d.x_init = a->x_init;
    d.a0 = a->all_initial;
    d.ev = a->event_idx;
    d.zj = a->z_jump;
    d.zjb = a->zj_stride_b;
    d.zje = a->zj_stride_e;
    d.vj = a->v_jump;
    d.vjb = a->vj_stride_b;
    d.vje = a->vj_stride_e;
    d.xo = a->x_out;
    d.io = a->i_out;
    const bool sv = a->save_act != nullptr;
    if ((a->save_xstage != nullptr) != sv || (a->save_ae_act != nullptr) != sv) return PSNODE_ERR_NULL;
    if (sv && a->event_idx && (!a->save_ev_act || !a->save_ev_i)) return PSNODE_ERR_NULL;
    d.sact = a->save_act;
    d.sxst = a->save_xstage;
    d.saeact = a->save_ae_act;
    d.sevact = a->save_ev_act;
    d.sevi = a->save_ev_i;
    return PSNODE_OK;
}

// dims-only view of the args for the *_kernel_for queries (pointers unused)
void bind_dims(const psnode_mlp_f32& m, MlpDev& d) {
    d.n_layers = m.n_layers;
    d.in_dim = m.in_dim;
    for (int l = 0; l < m.n_layers && l < kMaxLayers; ++l) d.out_dim[l] = m.out_dim[l];
}

}  // namespace
}  // namespace psnode

using namespace psnode;

extern "C" {

int32_t psnode_abi_version(void) { return PSNODE_ABI_VERSION; }

#define PSNODE_STR2(x) #x
#define PSNODE_STR(x) PSNODE_STR2(x)
const char* psnode_build_info(void) { return "psnode_hip abi " PSNODE_STR(PSNODE_ABI_VERSION) " gfx950 (generic + mfma kernels), built " __DATE__; }

const char* psnode_status_string(int32_t s) {
    switch (s) {
        case PSNODE_OK: return "ok";
        case PSNODE_ERR_NULL: return "required pointer is NULL";
        case PSNODE_ERR_DIMS: return "bad or inconsistent dimensions";
        case PSNODE_ERR_METHOD: return "unknown integration method";
        case PSNODE_ERR_WORKSPACE: return "workspace missing, misaligned (256 B) or too small";
        case PSNODE_ERR_UNSUPPORTED: return "shape not supported by the requested kernel";
        case PSNODE_ERR_HIP: return "HIP runtime error";
        default: return "unknown status";
    }
}

size_t psnode_workspace_bytes(const psnode_mlp_f32* de, const psnode_mlp_f32* ae) {
    if (!de) return 0;
    const size_t f = generic_floats(de) + generic_floats(ae) + mfma_pack_floats(de, ae) + kAlignFloats;
    return f * sizeof(float);
}

int32_t psnode_event_table_f32(int64_t n_steps, const float* clock, int64_t stride_k, const float* event_times,
                               int64_t stride_e, int32_t n_events, int32_t* event_idx, int32_t* dup_flag, void* stream) {
    if (n_steps <= 0) return PSNODE_OK;
    if (!clock || !event_idx || (n_events > 0 && !event_times)) return PSNODE_ERR_NULL;
    if (n_events < 0) return PSNODE_ERR_DIMS;
    const unsigned grid = (unsigned)((n_steps + 255) / 256);
    hipLaunchKernelGGL(event_table_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), (long long)n_steps, clock,
                       (long long)stride_k, event_times, (long long)stride_e, (int)n_events, event_idx, dup_flag);
    return hipGetLastError() == hipSuccess ? PSNODE_OK : PSNODE_ERR_HIP;
}

int32_t psnode_ode_save_hidden(const psnode_ode_args_f32* a) {
    if (!a) return 0;
    IntegrateDev d;
    memset(&d, 0, sizeof(d));
    d.method = a->method; d.flags = a->flags; d.xd = a->x_dim; d.zd = a->z_dim; d.T = a->T; d.B = a->B;
    bind_dims(a->de, d.de);
    return a->kernel == PSNODE_KERNEL_GENERIC ? 0 : mfma_ode_save_hidden(d);
}

int32_t psnode_ode_integrate_f32(const psnode_ode_args_f32* args, void* workspace, size_t workspace_bytes, void* stream) {
    IntegrateDev d;
    const int rc = fill_ode(args, d);
    if (rc) return rc;
    if (d.sact) {      // only K1 proper / K3c write the training side outputs (checked with the pointers in place: alignment counts)
        IntegrateDev q = d;
        bind_dims(args->de, q.de);
        if (args->kernel == PSNODE_KERNEL_GENERIC || !mfma_ode_save_hidden(q)) return PSNODE_ERR_UNSUPPORTED;
    }
    return dispatch(d, false, args->kernel, &args->de, nullptr, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

int32_t psnode_dae_save_hidden(const psnode_dae_args_f32* a) {
    if (!a) return 0;
    IntegrateDev d;
    memset(&d, 0, sizeof(d));
    d.method = a->method; d.flags = a->flags; d.xd = a->x_dim; d.zd = a->z_dim; d.vd = a->v_dim; d.id = a->i_dim; d.T = a->T; d.B = a->B;
    bind_dims(a->de, d.de);
    bind_dims(a->ae, d.ae);
    return a->kernel == PSNODE_KERNEL_GENERIC ? 0 : mfma_dae_save_hidden(d);
}

int32_t psnode_dae_integrate_f32(const psnode_dae_args_f32* args, void* workspace, size_t workspace_bytes, void* stream) {
    IntegrateDev d;
    const int rc = fill_dae(args, d);
    if (rc) return rc;
    if (d.sact) {      // only K2 proper writes the training side outputs
        IntegrateDev q = d;
        bind_dims(args->de, q.de);
        bind_dims(args->ae, q.ae);
        if (args->kernel == PSNODE_KERNEL_GENERIC || !mfma_dae_save_hidden(q)) return PSNODE_ERR_UNSUPPORTED;
    }
    return dispatch(d, true, args->kernel, &args->de, &args->ae, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

int32_t psnode_ode_kernel_for(const psnode_ode_args_f32* a) {
    if (!a) return PSNODE_ERR_NULL;
    IntegrateDev d;
    memset(&d, 0, sizeof(d));
    d.method = a->method; d.flags = a->flags; d.xd = a->x_dim; d.zd = a->z_dim; d.T = a->T; d.B = a->B;
    bind_dims(a->de, d.de);
    d.kern = a->kernel;                                              // a forced _TILE / _WAVE / GENERIC is what the call would run
    if (a->kernel == PSNODE_KERNEL_GENERIC || !mfma_ode_supported(d)) return PSNODE_KERNEL_GENERIC;
    d.sact = a->save_act;
    return mfma_x_ode_preferred(d) ? PSNODE_KERNEL_MFMA_WAVE : PSNODE_KERNEL_MFMA;      // (_WAVE: K1x -- the one-wave-per-4-trajectories integrator)
}

int32_t psnode_dae_kernel_for(const psnode_dae_args_f32* a) {
    if (!a) return PSNODE_ERR_NULL;
    IntegrateDev d;
    memset(&d, 0, sizeof(d));
    d.method = a->method; d.flags = a->flags; d.xd = a->x_dim; d.zd = a->z_dim; d.vd = a->v_dim; d.id = a->i_dim;
    d.T = a->T; d.B = a->B;
    bind_dims(a->de, d.de);
    bind_dims(a->ae, d.ae);
    d.kern = a->kernel;
    if (a->kernel == PSNODE_KERNEL_GENERIC || !mfma_dae_supported(d)) return PSNODE_KERNEL_GENERIC;
    d.sact = a->save_act;
    return mfma_x_dae_preferred(d) ? PSNODE_KERNEL_MFMA_WAVE : PSNODE_KERNEL_MFMA;      // (_WAVE: K2x)
}

}  // extern "C"

// ---- shared by every backward kernel: out[p] = sum over the per-workgroup partial vectors, in a fixed order (deterministic).
//      The first np_a entries go to out_a, the remaining np_b to out_b (out_b may be null when np_b == 0).
namespace psnode {
namespace {
// Two launches when there are many partial vectors (K8f: 1024 per-wave vectors of 1824 parameters = 44 us in one launch): the first sums
// each of kPartSlices slices INTO the slice's own first vector (in place -- the partials are the caller's scratch, exactly nparts vectors
// long, and a thread reads only its own column of its own slice), the second sums the slices' first vectors.
constexpr int kPartSlices = 16;
__device__ __forceinline__ float sum_parts(const float* __restrict__ part, const int np, const int pidx, const int q0, const int q1, const int qs) {
    // eight independent chains: one chain is (q1 - q0) / qs DEPENDENT loads; the order of the sum stays fixed
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int q = q0;
    for (; q + 8 * qs <= q1; q += 8 * qs) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += part[(size_t)(q + j * qs) * np + pidx];
    }
    for (int j = 0; q < q1; q += qs, ++j) acc[j] += part[(size_t)q * np + pidx];
    return ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
}
__global__ void reduce_partials_slices_kernel(float* __restrict__ part, int np, int nparts, int per) {
    const int pidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (pidx >= np) return;
    const int q0 = blockIdx.y * per, q1 = q0 + per < nparts ? q0 + per : nparts;
    if (q0 >= nparts) return;
    part[(size_t)q0 * np + pidx] = sum_parts(part, np, pidx, q0, q1, 1);
}
__global__ void reduce_partials_kernel(const float* __restrict__ part, float* __restrict__ out_a, float* __restrict__ out_b, int np_a,
                                       int np_b, int nparts, int stride) {
    const int pidx = blockIdx.x * blockDim.x + threadIdx.x, np = np_a + np_b;
    if (pidx >= np) return;
    const float total = sum_parts(part, np, pidx, 0, nparts, stride);
    if (pidx < np_a) out_a[pidx] = total;
    else out_b[pidx - np_a] = total;
}
}  // namespace
hipError_t launch_reduce_partials(float* part, float* out_a, float* out_b, int np_a, int np_b, int nparts, hipStream_t s) {
    const int np = np_a + np_b;
    int stride = 1;
    if (nparts >= 8 * kPartSlices) {       // many vectors: sum kPartSlices slices in place first (the partials are scratch: nothing reads them again)
        stride = (nparts + kPartSlices - 1) / kPartSlices;
        hipLaunchKernelGGL(reduce_partials_slices_kernel, dim3((np + 63) / 64, kPartSlices), dim3(64), 0, s, part, np, nparts, stride);
    }
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((np + 63) / 64), dim3(64), 0, s, part, out_a, out_b, np_a, np_b, nparts, stride);   // 64-wide: more CUs
    return hipGetLastError();
}
}  // namespace psnode
