// Packed weight images shared by the backward kernels that run at hidden 32 / 64 / 128 (K4w: psnode_backward_wide.hip, K4f:
// psnode_backward_fused.hip): the folded forward image of K1 (psnode_pack.h) and the transposed images of W2 / W3 for LDS.
#pragma once
#include "psnode_pack.h"

namespace psnode {
namespace {

typedef float wide_f4 __attribute__((ext_vector_type(4)));

// transposed images of W2 / W3 in the order the kernel's LDS array wants: [(layer * NWV + c) * NWV + w][lane] (f4):
//   reg r = W[16((w+c) % NWV) + 4g + r][16w + i]
struct PackWideT {
    int nw, hreal;
    const float *w2, *w3;
    wide_f4* out;
};
__global__ void pack_wide_t_kernel(const PackWideT p) {
    const int H = p.hreal, total = 2 * p.nw * p.nw * 64;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int lane = idx & 63, w = (idx >> 6) % p.nw, c = ((idx >> 6) / p.nw) % p.nw, layer = (idx >> 6) / (p.nw * p.nw);
        const int i = lane & 15, g = lane >> 4, ws = (w + c) & (p.nw - 1);
        const float* W = layer ? p.w3 : p.w2;
        wide_f4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * ws + 4 * g + r, col = 16 * w + i;
            v[r] = (row < H && col < H) ? W[(size_t)row * H + col] : 0.0f;
        }
        p.out[idx] = v;
    }
}

// forward images of W2 / W3 in the same slot order (K4f at 8 waves swaps them with the transposed ones in LDS):
//   reg r = W[16w + i][16((w+c) % NWV) + 4g + r]      -- the A-operand registers RD::W2 + 4c + r of psnode_pack.h, as one f4 per lane
__global__ void pack_wide_f_kernel(const PackWideT p) {
    const int H = p.hreal, total = 2 * p.nw * p.nw * 64;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int lane = idx & 63, w = (idx >> 6) % p.nw, c = ((idx >> 6) / p.nw) % p.nw, layer = (idx >> 6) / (p.nw * p.nw);
        const int i = lane & 15, g = lane >> 4, ws = (w + c) & (p.nw - 1);
        const float* W = layer ? p.w3 : p.w2;
        wide_f4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * w + i, col = 16 * ws + 4 * g + r;
            v[r] = (row < H && col < H) ? W[(size_t)row * H + col] : 0.0f;
        }
        p.out[idx] = v;
    }
}

size_t wide_fwd_floats(int nw, int n) { return (size_t)nw * (max_regs(nw) + (n + 3) / 4) * 64; }
size_t wide_t_floats(int nw) { return (size_t)2 * nw * nw * 64 * 4; }

__global__ void pack_wide_fwd_kernel(const PackMfma p) {
    const int R = pack_fwd_count(p);
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < p.nw * R * 64; idx += gridDim.x * blockDim.x)
        p.out[idx] = pack_fwd_value(p, (idx >> 6) / R, (idx >> 6) % R, idx & 63);
}


}  // namespace
}  // namespace psnode
