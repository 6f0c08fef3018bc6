// gm_slab.h -- the "slab" GEMM family for gfx950 (fp32 MFMA 16x16x4): operand slabs arrive in LDS by
// LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write stream), 4-wave workgroups, the
// reduction split ACROSS workgroups with an ordered (deterministic) cross-workgroup sum.
//
// Replaces, for the launches it takes, the split-reduction kernels of gm_gemm.hip (autograd + Adam of
// ns_gan.py:138-139,155-156; vae.py:127-191):
//   DW (TN):  C[m][n] = sum_k A[k][m] * B[k][n]      weight gradient, both operands contiguous along the OUTPUT index
//   FWD (NT): C[m][n] = sum_k A[m][k] * B[n][k]      forward, both operands contiguous along k
//   DX (NN):  C[m][n] = sum_k A[m][k] * B[k][n]      input gradient
//
// Structure of one workgroup (256 threads = one wave per SIMD, one workgroup per CU):
//   * the reduction range of the workgroup is cut into UNITS of UR rows (DW) / UK columns (FWD, DX); a unit's A and
//     B parts are 1 KB "pieces" (64 lanes x 16 bytes) that the four waves issue round-robin as LDS-DMA into a ring
//     of NBUF unit buffers -- NBUF units are in flight ahead of the MFMAs, counted with s_waitcnt vmcnt(n) (never 0
//     inside the loop) and one raw s_barrier per unit;
//   * fragments are read one unit ahead of the MFMAs that consume them (ds_read_b128 / b32), so the matrix pipe
//     sees a back-to-back stream of independent accumulators;
//   * DW: the four waves split each unit's k-steps between them (same output tile, TMW x TNW sub-tiles of 16x16
//     each); operands contiguous along the output index are read as INTERLEAVED fragments (lane i takes 4 consecutive
//     outputs with one ds_read_b128 and feeds element j to sub-tile j) -- nothing crosses lanes;
//   * FWD / DX: the four waves split the tile's rows; the k-contiguous operand uses the "4 consecutive k per
//     lane" fragment (MFMA j consumes element j: an identical k-permutation on both operands);
//   * the workgroup's partial tile goes through LDS once (row-major image), then
//       S == 1: straight to the epilogue;
//       S  > 1: the tile's rows are cut into S slices; every workgroup publishes the S-1 slices it does not own
//               (write-through 16-byte stores), arrives on the tile's counter, waits for its S-1 siblings and sums
//               ITS slice over the k-ranges in range order -- the result does not depend on who arrives when.  The
//               epilogue (bias / activation / Adam ...) therefore also runs on all S workgroups, 1/S of the tile each.
//     All workgroups of a launch must be co-resident (tiles * S <= number of CUs; checked by the host side).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../generative_models_amd/csrc/gm_ldsdma.h"

namespace slab {

// 1: a unit's DMA pieces are spread between the accumulator rows of the MFMA block instead of issued back to back in
// front of it.  Measured SLOWER (dW 32x48 tiles at 2048 rows 26.3 -> 28.9 us, with a 2-deep ring 26.8 -> 34.1:
// profiles/r04_slab_probe.md): what the spread costs in prefetch distance outweighs the queueing it avoids.
#ifndef GM_SLAB_SPREAD_DMA
#define GM_SLAB_SPREAD_DMA 0
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

enum { DW = 0, FWD = 1, DX = 2 };

struct Split {
    int S;                    // k-ranges (= workgroups) per tile; 1: no cross-workgroup sum
    float* ws;                // [tiles][S slices][S sources][slice floats]
    unsigned* cnt;            // [tiles][2] arrivals / departures; zero between launches
    unsigned* err;            // set to 1 when a wait ran out (never hangs the GPU)
    unsigned long long* trace;  // probe only (else NULL): 8 s_memtime stamps per workgroup
};

struct CoreP {
    const float* A; const float* B;
    int M, N, K;              // C is M x N, reduction length K
    int lda, ldb;
    int tm, tn;               // tiles along m / n
    int xmap;                 // 1: workgroup b -> (split, tile) so that an XCD (b % 8) owns whole k-ranges
    Split sp;
};

__device__ __forceinline__ void stamp(const Split& sp, int i) {
    if (sp.trace && threadIdx.x == 0) sp.trace[blockIdx.x * 8 + i] = __builtin_readcyclecounter();
}

// wait until at most `units` * NP of this wave's DMA pieces are outstanding (units: wave-uniform, 0 .. MAXU)
template <int NP, int MAXU>
__device__ __forceinline__ void wait_units(int units) {
    static_assert(NP * MAXU <= 63, "vmcnt is a 6-bit field");
    if constexpr (MAXU >= 5) { if (units >= 5) { wait_vm<5 * NP>(); return; } }
    if constexpr (MAXU >= 4) { if (units == 4) { wait_vm<4 * NP>(); return; } }
    if constexpr (MAXU >= 3) { if (units == 3) { wait_vm<3 * NP>(); return; } }
    if constexpr (MAXU >= 2) { if (units == 2) { wait_vm<2 * NP>(); return; } }
    if constexpr (MAXU >= 1) { if (units == 1) { wait_vm<1 * NP>(); return; } }
    wait_vm<0>();                                   // units <= 0
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float4 as_f4(u32x4 v) {
    return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
}
__device__ __forceinline__ u32x4 as_u4(float4 v) {
    return u32x4{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// workgroup -> (tile, k-range).  xmap: S in {1, 2, 4, 8}: the 8/S XCDs of a k-range share its tiles, so an XCD's L2
// pulls 1/S of both operands over the fabric exactly once (workgroup b runs on XCD b % 8: observed, speed only).
__device__ __forceinline__ bool map_block(const CoreP& p, int& tile, int& s) {
    const int T = p.tm * p.tn, S = p.sp.S, b = blockIdx.x;
    if (p.xmap) {
        const int g = 8 / S, xcd = b & 7, j = b >> 3;
        s = xcd / g;
        tile = j * g + (xcd % g);
    } else {
        s = b % S;
        tile = b / S;
    }
    return tile < T;
}

// ------------------------------------------------------------------------------------------------------------------
// Finish: `red` holds NRED row-major images [BM][BN] of the workgroup's partial tile (DW: one per wave).  Sums them,
// runs the cross-workgroup protocol, calls epi(row, c4, v) for every float4 (row, 4*c4 ..) of the rows this
// workgroup owns.  epi sees tile-local coordinates.
// ------------------------------------------------------------------------------------------------------------------
template <int NRED, int BM, int BN, class Epi>
__device__ __forceinline__ void finish_tile(const Split& sp, int tile, int s, float* red, const Epi& epi) {
    constexpr int Q = BN / 4, NU = BM * Q;
    const int t = threadIdx.x, S = sp.S;
    float4* r4 = reinterpret_cast<float4*>(red);
    if (S == 1) {
        for (int u = t; u < NU; u += 256) {
            float4 v = r4[u];
#pragma unroll
            for (int r = 1; r < NRED; ++r) v = add4(v, r4[r * NU + u]);
            epi(u / Q, u % Q, v);
        }
        return;
    }
    const int R = (BM + S - 1) / S;                 // rows per slice
    const int SLU = R * Q;                          // float4 units per slice (the last slice may be shorter)
    const __amdgpu_buffer_rsrc_t wsr = rsrc_of(sp.ws);
    const uint32_t tbase = (uint32_t)tile * S * S * SLU;      // in float4 units
    for (int u = t; u < NU; u += 256) {
        float4 v = r4[u];
#pragma unroll
        for (int r = 1; r < NRED; ++r) v = add4(v, r4[r * NU + u]);
        const int q = (u / Q) / R;
        if (q == s) r4[u] = v;                      // own slice stays here (only this thread touches the slot)
        else __builtin_amdgcn_raw_buffer_store_b128(as_u4(v), wsr, (tbase + (q * S + s) * SLU + (u - q * SLU)) * 16u, 0, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    stamp(sp, 4);
    if (t == 0) {
        __hip_atomic_fetch_add(&sp.cnt[2 * tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(&sp.cnt[2 * tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)S) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 22)) { *sp.err = 1u; break; }
        }
    }
    __syncthreads();
    stamp(sp, 5);
    const int u0 = s * SLU, u1 = min(u0 + SLU, NU);
    for (int u = u0 + t; u < u1; u += 256) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        bool first = true;
        for (int q = 0; q < S; ++q) {               // k-range order, whoever arrived first
            const float4 x = (q == s) ? r4[u]
                : as_f4(__builtin_amdgcn_raw_buffer_load_b128(wsr, (tbase + (s * S + q) * SLU + (u - u0)) * 16u, 0, 16));
            v = first ? x : add4(v, x);
            first = false;
        }
        epi(u / Q, u % Q, v);
    }
    __syncthreads();
    stamp(sp, 6);
    if (t == 0) {
        const unsigned old = __hip_atomic_fetch_add(&sp.cnt[2 * tile + 1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (unsigned)(S - 1)) {             // everybody has read: re-arm for the next launch
            __hip_atomic_store(&sp.cnt[2 * tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sp.cnt[2 * tile + 1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// DW: C[m][n] = sum_k A[k][m] B[k][n].  Tile (16 TMW) x (16 TNW); unit = 16 KU reduction rows, wave w owns rows
// [4 KU w, 4 KU (w + 1)) of every unit (KU k-steps of 4).  K % 4 == 0, M % 4 == 0, N % 4 == 0, 16-byte aligned rows.
// Interleaved sub-tiles: sub-tile e < 4 (TMW / 4) holds tile rows 64 (e / 4) + 4 i + e % 4 (i = MFMA row), the
// TMW % 4 remaining ones rows 64 (TMW / 4) + 16 e' + i; columns likewise.
// ------------------------------------------------------------------------------------------------------------------
template <int TMW, int TNW, int NBUF, int KU>
struct DwCfg {
    static constexpr int BM = 16 * TMW, BN = 16 * TNW, UR = 16 * KU;
    static constexpr int UNIT = UR * (BM + BN);                      // floats per unit buffer
    // the four waves' partial tiles meet in LDS: four images when they fit, else two (waves 2, 3 hand theirs to
    // waves 0, 1 first)
    static constexpr int NRED = (4 * BM * BN * 4 <= 160 * 1024) ? 4 : 2;
    static constexpr int RING = NBUF * UNIT, RED = NRED * BM * BN;
    static constexpr int LDS_FLOATS = RING > RED ? RING : RED;
    static constexpr int PA = KU * TMW, PB = KU * TNW, P = PA + PB, NP = (P + 3) / 4;
};

template <int T> __device__ __forceinline__ int il_local(int e, int i) {    // tile-local index of (sub-tile e, MFMA index i)
    constexpr int a = T / 4;
    return (e < 4 * a) ? 64 * (e / 4) + 4 * i + (e % 4) : 64 * a + 16 * (e - 4 * a) + i;
}

template <int TMW, int TNW, int NBUF, int KU, class Epi, class AXf, int ABL = 0>
__device__ __forceinline__ void dw_tile(const CoreP& p, float* lds, int tile, int s, const Epi& epi, const AXf& axf) {
    using C = DwCfg<TMW, TNW, NBUF, KU>;
    constexpr int BM = C::BM, BN = C::BN, UR = C::UR, NP = C::NP;
    const int t = threadIdx.x, lane = t & 63, i16 = lane & 15, g4 = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tile_m = tile / p.tn, tile_n = tile % p.tn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const uint32_t lds_base = (uint32_t)(uintptr_t)lds;

    // this wave's DMA pieces of a unit: piece x covers float4 units [64 x, 64 x + 64) of the unit image
    // [UR][BM] ++ [UR][BN]; out-of-range columns are clamped into the row (they only feed outputs nobody stores)
    const float* pbase[NP]; int prow[NP], pld[NP]; uint32_t pdst[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int x = min(w + 4 * j, C::P - 1);                       // wave uniform (a repeated last piece is harmless)
        const bool isA = x < C::PA;
        const int idx = (isA ? x : x - C::PA) * 64 + lane;
        const int row = isA ? idx / (BM / 4) : idx / (BN / 4);
        const int c4 = isA ? idx % (BM / 4) : idx % (BN / 4);
        const int col = isA ? min(m0 + 4 * c4, p.M - 4) : min(n0 + 4 * c4, p.N - 4);
        pbase[j] = (isA ? p.A : p.B) + col;
        pld[j] = isA ? p.lda : p.ldb;
        prow[j] = row;
        pdst[j] = (uint32_t)x * 1024u;
    }
    // reduction range of this workgroup, in units
    const int units = (p.K + UR - 1) / UR;
    const int ub = (int)(((long)units * s) / p.sp.S), ue = (int)(((long)units * (s + 1)) / p.sp.S);
    const int nU = ue - ub;

    auto issue_piece = [&](int j, int u, int slot) {   // piece j of unit u (absolute) -> ring slot
        if constexpr (ABL == 1) return;               // probe: no operand traffic
        const int k = min(u * UR + prow[j], p.K - 1);
        glds16(pbase[j] + (int64_t)k * pld[j], lds_base + (uint32_t)slot * (C::UNIT * 4u) + pdst[j]);
    };
    auto issue = [&](int u, int slot) {
#pragma unroll
        for (int j = 0; j < NP; ++j) issue_piece(j, u, slot);
    };
    float fa[2][KU][TMW], fb[2][KU][TNW];
    auto frags = [&](int slot, int u, float (&a)[KU][TMW], float (&b)[KU][TNW]) {
        const float* ub_ = lds + slot * C::UNIT;
#pragma unroll
        for (int ks = 0; ks < KU; ++ks) {
            const int row = 4 * KU * w + 4 * ks + g4;
            const float* ar = ub_ + row * BM;
            const float* br = ub_ + UR * BM + row * BN;
#pragma unroll
            for (int q = 0; q < TMW / 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(ar + 64 * q + 4 * i16);
                a[ks][4 * q] = v.x; a[ks][4 * q + 1] = v.y; a[ks][4 * q + 2] = v.z; a[ks][4 * q + 3] = v.w;
            }
#pragma unroll
            for (int e = 4 * (TMW / 4); e < TMW; ++e) a[ks][e] = ar[64 * (TMW / 4) + 16 * (e - 4 * (TMW / 4)) + i16];
#pragma unroll
            for (int q = 0; q < TNW / 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(br + 64 * q + 4 * i16);
                b[ks][4 * q] = v.x; b[ks][4 * q + 1] = v.y; b[ks][4 * q + 2] = v.z; b[ks][4 * q + 3] = v.w;
            }
#pragma unroll
            for (int f = 4 * (TNW / 4); f < TNW; ++f) b[ks][f] = br[64 * (TNW / 4) + 16 * (f - 4 * (TNW / 4)) + i16];
        }
    };
    // k-steps past K (last unit only): the loads were clamped to row K - 1 (finite), the A side is zeroed; axf: the
    // caller's transform of an A element (folded head: dH from h)
    auto fix = [&](int u, const float (&a)[KU][TMW], float (&o)[KU][TMW]) {
#pragma unroll
        for (int ks = 0; ks < KU; ++ks) {
            const bool live = u * UR + 4 * KU * w + 4 * ks < p.K;
#pragma unroll
            for (int e = 0; e < TMW; ++e) o[ks][e] = live ? axf(a[ks][e], u * UR + 4 * KU * w + 4 * ks + g4, e, i16) : 0.f;
        }
    };
    f32x4 acc[TMW][TNW];
#pragma unroll
    for (int e = 0; e < TMW; ++e)
#pragma unroll
        for (int f = 0; f < TNW; ++f) acc[e][f] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto mfmas = [&](const float (&a)[KU][TMW], const float (&b)[KU][TNW]) {
        if constexpr (ABL == 2) {                       // probe: operands arrive and are read, no MFMA
#pragma unroll
            for (int ks = 0; ks < KU; ++ks) {
#pragma unroll
                for (int e = 0; e < TMW; ++e) asm volatile("" ::"v"(a[ks][e]));
#pragma unroll
                for (int f = 0; f < TNW; ++f) asm volatile("" ::"v"(b[ks][f]));
            }
            return;
        }
#pragma unroll
        for (int ks = 0; ks < KU; ++ks)
#pragma unroll
            for (int e = 0; e < TMW; ++e)
#pragma unroll
                for (int f = 0; f < TNW; ++f)
                    acc[e][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks][e], b[ks][f], acc[e][f], 0, 0, 0);
    };
    // The same MFMAs with the unit-after-next's DMA pieces SPREAD between the accumulator rows: a wave that issues
    // its pieces back to back sits in the memory pipeline's queue (one 1 KB piece per ~40 cycles per CU, measured)
    // and issues no MFMA meanwhile -- the fetch phase and the MFMA phase then add up instead of overlapping.
    auto mfmas_issue = [&](const float (&a)[KU][TMW], const float (&b)[KU][TNW], bool more, int u, int slot) {
        if constexpr (ABL != 0 || !GM_SLAB_SPREAD_DMA) { if (more) issue(u, slot); mfmas(a, b); return; }
        constexpr int G = KU * TMW;
#pragma unroll
        for (int ks = 0; ks < KU; ++ks)
#pragma unroll
            for (int e = 0; e < TMW; ++e) {
#pragma unroll
                for (int f = 0; f < TNW; ++f)
                    acc[e][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks][e], b[ks][f], acc[e][f], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NP; ++j)
                    if ((j * G) / NP == ks * TMW + e) { if (more) issue_piece(j, u, slot); }
            }
    };

    // prologue: NBUF units in flight, the first one's fragments in registers
    stamp(p.sp, 0);
#pragma unroll
    for (int j = 0; j < NBUF; ++j)
        if (j < nU) issue(ub + j, j);
    wait_units<NP, NBUF - 1>(min(NBUF - 1, nU - 1));
    sync_raw();
    stamp(p.sp, 1);
    frags(0, ub, fa[1], fb[0]);
    fix(ub, fa[1], fa[0]);
    // steady state: one unit per trip, branch-free around the MFMAs (a conditional second step made hipcc shuttle all
    // accumulators between VGPRs and AGPRs every trip).  Past the range's end the wait is a full one and the
    // fragment read a harmless read of a stale slot.
    int slot = 0;                                   // ring slot of unit ub + i
#pragma unroll 2
    for (int i = 0; i < nU; ++i) {
        const int nslot = (slot + 1 == NBUF) ? 0 : slot + 1;
        // units i+1 .. min(i+NBUF-1, nU-1) are in flight; unit i+1 must have landed
        wait_units<NP, NBUF - 2>(min(NBUF - 2, nU - 2 - i));
        wait_lgkm0();                               // this wave's reads of slot `slot` are complete
        sync_raw();                                 // ... everybody's; and unit i+1 landed for everybody
        frags(nslot, ub + i + 1, fa[1], fb[1]);     // next unit's fragment reads go out first ...
        __builtin_amdgcn_sched_barrier(0);
        mfmas_issue(fa[0], fb[0], i + NBUF < nU, ub + i + NBUF, slot);   // ... and land under this unit's MFMAs
        __builtin_amdgcn_sched_barrier(0);
        fix(ub + i + 1, fa[1], fa[0]);
#pragma unroll
        for (int ks = 0; ks < KU; ++ks)
#pragma unroll
            for (int f = 0; f < TNW; ++f) fb[0][ks][f] = fb[1][ks][f];
        slot = nslot;
    }
    // partial tiles of the four waves -> LDS, row-major images
    stamp(p.sp, 2);
    wait_vm<0>();
    __syncthreads();
    auto dump = [&](float* red, bool add) {
#pragma unroll
        for (int e = 0; e < TMW; ++e)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* rowp = red + il_local<TMW>(e, 4 * g4 + r) * BN;
#pragma unroll
                for (int q = 0; q < TNW / 4; ++q) {
                    float4* dst = reinterpret_cast<float4*>(rowp + 64 * q + 4 * i16);
                    float4 v = make_float4(acc[e][4 * q][r], acc[e][4 * q + 1][r], acc[e][4 * q + 2][r], acc[e][4 * q + 3][r]);
                    if (add) v = add4(v, *dst);
                    *dst = v;
                }
#pragma unroll
                for (int f = 4 * (TNW / 4); f < TNW; ++f) {
                    float* dst = rowp + 64 * (TNW / 4) + 16 * (f - 4 * (TNW / 4)) + i16;
                    *dst = add ? acc[e][f][r] + *dst : acc[e][f][r];
                }
            }
    };
    if constexpr (C::NRED == 4) {
        dump(lds + w * (BM * BN), false);
    } else {
        if (w >= 2) dump(lds + (w - 2) * (BM * BN), false);
        __syncthreads();
        if (w < 2) dump(lds + w * (BM * BN), true);           // image w = wave w + wave w + 2
    }
    __syncthreads();
    stamp(p.sp, 3);
    finish_tile<C::NRED, BM, BN>(p.sp, tile, s, lds, epi);
    stamp(p.sp, 7);
}

struct NoAXf { __device__ __forceinline__ float operator()(float v, int, int, int) const { return v; } };

// ------------------------------------------------------------------------------------------------------------------
// FWD: C[m][n] = sum_k A[m][k] B[n][k].  Tile (64 TMW) x (16 TNW): wave w owns rows [16 TMW w, 16 TMW (w+1)) and all
// TNW column sub-tiles; unit = UK = 32 reduction columns (full 128-byte lines per row and piece), two k-steps of 16.
// K % 16 == 0, 16-byte aligned rows.  LDS image of a part: [rows][8 x 16 bytes] with the 16-byte index XORed with
// (row >> 1) & 7 (applied on the SOURCE address of the DMA and on the read: conflict-free ds_read_b128).
// ------------------------------------------------------------------------------------------------------------------
template <int TMW, int TNW, int NBUF>
struct FwCfg {
    static constexpr int BM = 64 * TMW, BN = 16 * TNW, UK = 32;
    static constexpr int UNIT = UK * (BM + BN);
    static constexpr int RING = NBUF * UNIT, RED = BM * BN;
    static constexpr int LDS_FLOATS = RING > RED ? RING : RED;
    static constexpr int PA = BM / 8, PB = BN / 8, P = PA + PB, NP = (P + 3) / 4;   // a piece = 8 rows x 128 bytes
};

template <int TMW, int TNW, int NBUF, class Epi>
__device__ __forceinline__ void fwd_tile(const CoreP& p, float* lds, int tile, int s, const Epi& epi) {
    using C = FwCfg<TMW, TNW, NBUF>;
    constexpr int BM = C::BM, BN = C::BN, UK = C::UK, NP = C::NP;
    const int t = threadIdx.x, lane = t & 63, i16 = lane & 15, g4 = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tile_m = tile / p.tn, tile_n = tile % p.tn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const uint32_t lds_base = (uint32_t)(uintptr_t)lds;

    const float* pbase[NP]; int pk[NP]; uint32_t pdst[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int x = min(w + 4 * j, C::P - 1);
        const bool isA = x < C::PA;
        const int row = (isA ? x : x - C::PA) * 8 + (lane >> 3);      // row of the part
        const int up = lane & 7;                                      // 16-byte position inside the LDS row
        const int k4 = up ^ ((row >> 1) & 7);                         // ... holds this 16-byte unit of the source row
        const int grow = isA ? min(m0 + row, p.M - 1) : min(n0 + row, p.N - 1);
        pbase[j] = (isA ? p.A + (int64_t)grow * p.lda : p.B + (int64_t)grow * p.ldb);
        pk[j] = 4 * k4;
        pdst[j] = (uint32_t)x * 1024u;
    }
    // reduction range in k-steps of 16; a unit is two steps
    const int steps = p.K / 16;
    const int sb = (int)(((long)steps * s) / p.sp.S), se = (int)(((long)steps * (s + 1)) / p.sp.S);
    const int nU = (se - sb + 1) / 2;

    auto issue_piece = [&](int j, int i, int slot) { // piece j of unit i of this workgroup's range
        glds16(pbase[j] + min(16 * (sb + 2 * i) + pk[j], p.K - 4), lds_base + (uint32_t)slot * (C::UNIT * 4u) + pdst[j]);
    };
    auto issue = [&](int i, int slot) {
#pragma unroll
        for (int j = 0; j < NP; ++j) issue_piece(j, i, slot);
    };
    float4 fa[2][2][TMW], fb[2][2][TNW];
    auto frags = [&](int slot, float4 (&a)[2][TMW], float4 (&b)[2][TNW]) {
        const float* ua = lds + slot * C::UNIT;
        const float* ubp = ua + UK * BM;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int e = 0; e < TMW; ++e) {
                const int row = 16 * (TMW * w + e) + i16;
                a[h][e] = *reinterpret_cast<const float4*>(ua + row * UK + 4 * ((4 * h + g4) ^ ((row >> 1) & 7)));
            }
#pragma unroll
            for (int f = 0; f < TNW; ++f) {
                const int row = 16 * f + i16;
                b[h][f] = *reinterpret_cast<const float4*>(ubp + row * UK + 4 * ((4 * h + g4) ^ ((row >> 1) & 7)));
            }
        }
    };
    f32x4 acc[TMW][TNW];
#pragma unroll
    for (int e = 0; e < TMW; ++e)
#pragma unroll
        for (int f = 0; f < TNW; ++f) acc[e][f] = f32x4{0.f, 0.f, 0.f, 0.f};
    // MFMAs of one unit with the DMA pieces of unit i + NBUF spread between the accumulator rows (see dw_tile)
    auto mfmas_issue = [&](const float4 (&a)[2][TMW], const float4 (&b)[2][TNW], bool more, int i, int slot) {
        constexpr int G = 2 * TMW * TNW;
        if constexpr (!GM_SLAB_SPREAD_DMA) { if (more) issue(i, slot); }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < TMW; ++e)
#pragma unroll
                for (int f = 0; f < TNW; ++f) {
                    f32x4 c = acc[e][f];
                    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[h][e].x, b[h][f].x, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[h][e].y, b[h][f].y, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[h][e].z, b[h][f].z, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[h][e].w, b[h][f].w, c, 0, 0, 0);
                    acc[e][f] = c;
#pragma unroll
                    for (int j = 0; j < NP; ++j)
                        if (GM_SLAB_SPREAD_DMA && (j * G) / NP == (h * TMW + e) * TNW + f) { if (more) issue_piece(j, i, slot); }
                }
    };
    // an odd step count leaves the range's last unit one step short: its second step's A side is zeroed (the loads
    // were clamped into the operand: finite values)
    auto fix = [&](int i, float4 (&a)[2][TMW]) {
        const bool two = sb + 2 * i + 1 < se;
#pragma unroll
        for (int e = 0; e < TMW; ++e) a[1][e] = two ? a[1][e] : make_float4(0.f, 0.f, 0.f, 0.f);
    };
#pragma unroll
    for (int j = 0; j < NBUF; ++j)
        if (j < nU) issue(j, j);
    wait_units<NP, NBUF - 1>(min(NBUF - 1, nU - 1));
    sync_raw();
    frags(0, fa[0], fb[0]);
    fix(0, fa[0]);
    int slot = 0;
#pragma unroll 2
    for (int i = 0; i < nU; ++i) {
        const int nslot = (slot + 1 == NBUF) ? 0 : slot + 1;
        wait_units<NP, NBUF - 2>(min(NBUF - 2, nU - 2 - i));
        wait_lgkm0();
        sync_raw();
        frags(nslot, fa[1], fb[1]);
        __builtin_amdgcn_sched_barrier(0);
        mfmas_issue(fa[0], fb[0], i + NBUF < nU, i + NBUF, slot);
        __builtin_amdgcn_sched_barrier(0);
        fix(i + 1, fa[1]);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int e = 0; e < TMW; ++e) fa[0][h][e] = fa[1][h][e];
#pragma unroll
            for (int f = 0; f < TNW; ++f) fb[0][h][f] = fb[1][h][f];
        }
        slot = nslot;
    }
    wait_vm<0>();
    __syncthreads();
    // row-major image of the tile: C layout of the 16x16 forms: column = lane & 15, row = 4 (lane >> 4) + reg
#pragma unroll
    for (int e = 0; e < TMW; ++e)
#pragma unroll
        for (int f = 0; f < TNW; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) lds[(16 * (TMW * w + e) + 4 * g4 + r) * BN + 16 * f + i16] = acc[e][f][r];
    __syncthreads();
    finish_tile<1, BM, BN>(p.sp, tile, s, lds, epi);
}

}  // namespace slab
