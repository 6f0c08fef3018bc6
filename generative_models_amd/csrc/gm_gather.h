// K1 gather: out[b,:] = data[idx[b],:]   (process_batch, ns_gan.py:222-226).
// One 3136-byte image row per wave: 196 float4 -> lanes issue coalesced 16-B loads.  Shared between
// its own launch (gm_gather_rows, 4 waves per workgroup) and the forward GEMM that can carry the
// gather workgroups in its grid (gm_linear_fwd_gather, 16 waves per workgroup).
#pragma once
#include "gm_common.h"

struct GatherP {
    const float* data; int64_t n_rows;
    const int64_t* idx; gm_slot idx_slot;
    float* out; int64_t ld_out;
    int B, row_elems, vec;
};

// bid: index among the gather workgroups; every workgroup has blockDim.x / 64 waves = rows.
static __device__ __forceinline__ void gather_body(const GatherP& p, int bid) {
    const int64_t* ix = p.idx + gm_slot_offset(p.idx_slot);
    const int wpb = blockDim.x >> 6;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = bid * wpb + wave;
    if (b >= p.B) return;
    int64_t r = ix[b];
    if (r < 0 || r >= p.n_rows) r = 0;     // never fault on a corrupt index; parity tests catch it
    const float* src = p.data + r * (int64_t)p.row_elems;
    float* dst = p.out + (int64_t)b * p.ld_out;
    if (p.vec) {
        const int n4 = p.row_elems >> 2;
        const float4* s4 = reinterpret_cast<const float4*>(src);
        float4* d4 = reinterpret_cast<float4*>(dst);
        for (int i = lane; i < n4; i += 64) d4[i] = s4[i];
    } else {
        for (int i = lane; i < p.row_elems; i += 64) dst[i] = src[i];
    }
}

static __global__ __launch_bounds__(256) void gather_rows_kernel(GatherP p) {
    gather_body(p, blockIdx.x);
}

static inline int gm_gather_blocks(const GatherP& p, int waves_per_block) {
    return (p.B + waves_per_block - 1) / waves_per_block;
}

static inline int gm_gather_fill(const float* data, int64_t n_rows, const int64_t* idx,
                                 gm_slot idx_slot, float* out, int64_t ld_out, int B, int row_elems,
                                 GatherP* g) {
    GM_CHECK_ARG(data && idx && out && B > 0 && row_elems > 0 && ld_out >= row_elems && n_rows > 0);
    g->data = data; g->n_rows = n_rows; g->idx = idx; g->idx_slot = idx_slot; g->out = out;
    g->ld_out = ld_out; g->B = B; g->row_elems = row_elems;
    g->vec = (row_elems % 4 == 0) && (ld_out % 4 == 0) &&
             ((reinterpret_cast<uintptr_t>(data) & 15) == 0) &&
             ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    return 0;
}
