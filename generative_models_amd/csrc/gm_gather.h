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
    // bit-packed resident dataset (utils.py:31 binarises MNIST, so a pixel is
    // one bit): row r = words[r * wpr .. ), pixel i = bit (i & 31) of word i >> 5.  98 B/row instead
    // of 3136 B: the one batch-proportional HBM read stream of the step shrinks 32x; the gather
    // expands to the fp32 rows the GEMMs consume.
    const uint32_t* bits; int wpr;
    // ... and, instead of expanding, the selected rows copied AS WORDS: out_bits[b * wpr ..) =
    // the dataset row's wpr words; the consumers (gm_linear_fwd_headpart_bits, gm_linear_bwd_dw_adam_head_fold_bits)
    // expand in registers.  100 B written per row instead of 3136.
    uint32_t* out_bits;
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
    float* dst = p.out + (int64_t)b * p.ld_out;
    if (p.bits) {
        const uint32_t* w = p.bits + r * (int64_t)p.wpr;
        if (p.out_bits) {
            for (int i = lane; i < p.wpr; i += 64) p.out_bits[(int64_t)b * p.wpr + i] = w[i];
            return;
        }
        if (p.vec) {                        // 4 pixels = 4 bits of one word (row_elems % 4 == 0)
            float4* d4 = reinterpret_cast<float4*>(dst);
            for (int q = lane; q < (p.row_elems >> 2); q += 64) {
                const uint32_t v = w[q >> 3] >> (4 * (q & 7));
                d4[q] = make_float4((float)(v & 1u), (float)((v >> 1) & 1u), (float)((v >> 2) & 1u),
                                    (float)((v >> 3) & 1u));
            }
        } else {
            for (int i = lane; i < p.row_elems; i += 64) dst[i] = (float)((w[i >> 5] >> (i & 31)) & 1u);
        }
        return;
    }
    const float* src = p.data + r * (int64_t)p.row_elems;
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
    g->bits = nullptr; g->wpr = 0; g->out_bits = nullptr;
    return 0;
}

// `data` is really the bit-packed dataset (see GatherP): words_per_row uint32 per row
static inline int gm_gather_fill_bits(const uint32_t* bits, int words_per_row, int64_t n_rows,
                                      const int64_t* idx, gm_slot idx_slot, float* out, int64_t ld_out,
                                      int B, int row_elems, GatherP* g) {
    GM_CHECK_ARG(bits && idx && out && B > 0 && row_elems > 0 && ld_out >= row_elems && n_rows > 0);
    GM_CHECK_ARG(words_per_row * 32 >= row_elems);
    g->data = nullptr; g->n_rows = n_rows; g->idx = idx; g->idx_slot = idx_slot; g->out = out;
    g->ld_out = ld_out; g->B = B; g->row_elems = row_elems;
    g->vec = (row_elems % 4 == 0) && (ld_out % 4 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    g->bits = bits; g->wpr = words_per_row; g->out_bits = nullptr;
    return 0;
}

// bit-packed dataset, rows copied as words (GatherP::out_bits)
static inline int gm_gather_fill_bits_packed(const uint32_t* bits, int words_per_row, int64_t n_rows,
                                             const int64_t* idx, gm_slot idx_slot, uint32_t* out_bits, int B,
                                             GatherP* g) {
    GM_CHECK_ARG(bits && idx && out_bits && B > 0 && words_per_row > 0 && n_rows > 0 && out_bits != bits);
    g->data = nullptr; g->n_rows = n_rows; g->idx = idx; g->idx_slot = idx_slot; g->out = nullptr;
    g->ld_out = 0; g->B = B; g->row_elems = 32 * words_per_row; g->vec = 0;
    g->bits = bits; g->wpr = words_per_row; g->out_bits = out_bits;
    return 0;
}
