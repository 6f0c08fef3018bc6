// gm_hostrng.cpp -- HOST side of the reference's RNG protocol (no device work).
//
// The reference draws everything a step needs from torch's GLOBAL CPU generator, per step:
//   process_batch  ns_gan.py:222-226  -> DataLoader iterator: base seed + sampler seed (two 64-bit
//                                        draws), randperm(N) on a private generator (first B used)
//   compute_noise  ns_gan.py:218-220  -> torch.randn(B, Z)
//   WGAN-GP eps    w_gp_gan.py:197    -> torch.rand(B, 1)
//   DRAGAN         dra_gan.py:200,205 -> torch.rand(B, 1), torch.rand(B, 784)
//   InfoGAN        info_gan.py:312-323-> randn(B, z) | randint(0, 10, (B,)) | randn(B, c)
//   VAE            vae.py:104         -> torch.randn(mu.shape)
// Round 1 replayed this with one torch call per draw from Python (63 us per NSGAN iteration, the
// wall behind the 71 us GPU step; DRAGAN 0.6 ms per critic step).  This file replays a whole
// sub-chunk of iterations in ONE C call from a serialized generator state: a 32-bit-state mt19937
// whose twist/temper loops vectorize (torch keeps 64-bit words and produces one output per call),
// ATen's float `uniform_` / `normal_` / `random_` / `randint` transformations restated exactly
// (aten/src/ATen/core/DistributionsHelper.h, native/cpu/DistributionTemplates.h: 24-bit
// uniforms; normal_fill's 16-wide Box-Muller over u[j], u[j+8] with libm logf/cosf/sinf), and a
// small persistent worker pool for the Box-Muller stage (the mt19937 stream itself is serial).
// Python verifies the restatement bit-for-bit against torch on a cloned generator before using it
// (engine.HostReplay.selfcheck) and falls back to the per-draw torch path if anything differs.
#include <immintrin.h>

#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

extern "C" void gm_set_error(const char* msg);     // gm_ops.hip
extern "C" int gm_randperm_prefix(uint64_t seed, int64_t n, int B, int64_t* out);   // below

namespace {

constexpr int GM_EINVAL = -10001;
constexpr int GM_EUNSUPPORTED = -10002;

// at::CPUGeneratorImplState (CPUGeneratorImpl.cpp): legacy POD + float normal cache
struct TorchCpuGenState {
    uint64_t the_initial_seed;
    int32_t left;
    int32_t seeded;
    uint64_t next;
    uint64_t state[624];
    double normal_x, normal_y, normal_rho;
    int32_t normal_is_valid;
    float next_float_normal_sample;
    bool is_next_float_normal_sample_valid;
};

// ---- mt19937 with 32-bit words ----------------------------------------------------------------
struct Mt {
    alignas(64) uint32_t s[624 + 16];
    int next;      // index of the next untempered word
    int remain;    // words left before the next twist
};

#define GM_TWIST_BODY                                                                              \
    for (int k = 0; k < 227; ++k) {                                                                \
        const uint32_t y = (s[k] & 0x80000000u) | (s[k + 1] & 0x7fffffffu);                        \
        s[k] = s[k + 397] ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);                            \
    }                                                                                              \
    for (int k = 227; k < 623; ++k) {                                                              \
        const uint32_t y = (s[k] & 0x80000000u) | (s[k + 1] & 0x7fffffffu);                        \
        s[k] = s[k - 227] ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);                            \
    }                                                                                              \
    {                                                                                              \
        const uint32_t y = (s[623] & 0x80000000u) | (s[0] & 0x7fffffffu);                          \
        s[623] = s[396] ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);                              \
    }
// dependence distances are 227 / 397 words and s[k+1] is read before it is written: any SIMD
// width up to 227 lanes is safe, which the compiler cannot prove -> ivdep
__attribute__((target("avx512f"))) void twist_avx512(uint32_t* __restrict__ s) {
#pragma GCC ivdep
    GM_TWIST_BODY
}
__attribute__((target("avx2"))) void twist_avx2(uint32_t* __restrict__ s) {
#pragma GCC ivdep
    GM_TWIST_BODY
}
void twist_base(uint32_t* __restrict__ s) {
#pragma GCC ivdep
    GM_TWIST_BODY
}

#define GM_TEMPER_BODY                                                                             \
    for (int i = 0; i < n; ++i) {                                                                  \
        uint32_t y = src[i];                                                                       \
        y ^= (y >> 11);                                                                            \
        y ^= (y << 7) & 0x9d2c5680u;                                                               \
        y ^= (y << 15) & 0xefc60000u;                                                              \
        y ^= (y >> 18);                                                                            \
        dst[i] = y;                                                                                \
    }
__attribute__((target("avx512f"))) void temper_avx512(const uint32_t* __restrict__ src,
                                                      uint32_t* __restrict__ dst, int n) { GM_TEMPER_BODY }
__attribute__((target("avx2"))) void temper_avx2(const uint32_t* __restrict__ src,
                                                 uint32_t* __restrict__ dst, int n) { GM_TEMPER_BODY }
void temper_base(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int n) { GM_TEMPER_BODY }

// tempered 24-bit uniforms in [0,1): (x & (2^24-1)) * 2^-24  (uniform_real_distribution<float>)
#define GM_UNIF_BODY                                                                               \
    for (int i = 0; i < n; ++i) {                                                                  \
        uint32_t y = src[i];                                                                       \
        y ^= (y >> 11);                                                                            \
        y ^= (y << 7) & 0x9d2c5680u;                                                               \
        y ^= (y << 15) & 0xefc60000u;                                                              \
        y ^= (y >> 18);                                                                            \
        dst[i] = (float)(int32_t)(y & 0x00ffffffu) * 5.9604644775390625e-8f;                       \
    }
__attribute__((target("avx512f"))) void unif_avx512(const uint32_t* __restrict__ src,
                                                    float* __restrict__ dst, int n) { GM_UNIF_BODY }
__attribute__((target("avx2"))) void unif_avx2(const uint32_t* __restrict__ src,
                                               float* __restrict__ dst, int n) { GM_UNIF_BODY }
void unif_base(const uint32_t* __restrict__ src, float* __restrict__ dst, int n) { GM_UNIF_BODY }

int cpu_level() {
    static const int lvl = __builtin_cpu_supports("avx512f") ? 2 : (__builtin_cpu_supports("avx2") ? 1 : 0);
    return lvl;
}
inline void twist(uint32_t* s) {
    const int l = cpu_level();
    if (l == 2) twist_avx512(s); else if (l == 1) twist_avx2(s); else twist_base(s);
}
inline void temper(const uint32_t* src, uint32_t* dst, int n) {
    const int l = cpu_level();
    if (l == 2) temper_avx512(src, dst, n); else if (l == 1) temper_avx2(src, dst, n); else temper_base(src, dst, n);
}
inline void unif(const uint32_t* src, float* dst, int n) {
    const int l = cpu_level();
    if (l == 2) unif_avx512(src, dst, n); else if (l == 1) unif_avx2(src, dst, n); else unif_base(src, dst, n);
}

int mt_load(Mt& m, const TorchCpuGenState* g) {
    if (g->seeded != 1 || g->left < 1 || g->left > 624 || g->next > 624) return GM_EINVAL;
    for (int i = 0; i < 624; ++i) m.s[i] = (uint32_t)g->state[i];
    // at::mt19937::operator(): if (--left == 0) next_state(); y = state[next++]
    m.remain = g->left - 1;
    m.next = (int)g->next;
    return 0;
}
void mt_store(const Mt& m, TorchCpuGenState* g) {
    for (int i = 0; i < 624; ++i) g->state[i] = m.s[i];
    g->left = m.remain + 1;
    g->next = (uint64_t)m.next;
}
// n tempered 32-bit outputs
void mt_raw(Mt& m, uint32_t* out, int64_t n) {
    while (n > 0) {
        if (m.remain == 0) { twist(m.s); m.next = 0; m.remain = 624; }
        const int take = (int)(n < m.remain ? n : m.remain);
        temper(m.s + m.next, out, take);
        m.next += take; m.remain -= take; out += take; n -= take;
    }
}
void mt_uniform(Mt& m, float* out, int64_t n) {
    while (n > 0) {
        if (m.remain == 0) { twist(m.s); m.next = 0; m.remain = 624; }
        const int take = (int)(n < m.remain ? n : m.remain);
        unif(m.s + m.next, out, take);
        m.next += take; m.remain -= take; out += take; n -= take;
    }
}
void mt_skip(Mt& m, int64_t n) {
    while (n > 0) {
        if (m.remain == 0) { twist(m.s); m.next = 0; m.remain = 624; }
        const int take = (int)(n < m.remain ? n : m.remain);
        m.next += take; m.remain -= take; n -= take;
    }
}
inline uint32_t mt_one(Mt& m) { uint32_t y; mt_raw(m, &y, 1); return y; }
// CPUGeneratorImpl::random64(): two engine outputs, the first is the high word
inline uint64_t mt_random64(Mt& m) {
    uint32_t y[2];
    mt_raw(m, y, 2);
    return ((uint64_t)y[0] << 32) | y[1];
}

// ---- normal_fill_16 (DistributionTemplates.h), float: in place on 16 uniforms ----------------
// ATen registers the distribution kernels without an AVX-512 variant, so on every x86 host with
// AVX2 `normal_()` on >= 16 contiguous floats runs normal_fill_AVX2: Box-Muller on 8 lanes with
// the cephes-style log256_ps / sincos256_ps of avx_mathfun.h.  Those are plain polynomial code;
// restated here lane by lane (the loops below vectorize to the same 8-wide operations).  The one
// thing the source does not pin down is whether the compiler that built torch contracted the
// mul+add pairs into FMAs; both forms exist (FMA = true/false) and the Python side picks the one
// that reproduces torch bit for bit on this host (engine.HostReplay.selfcheck).
// Flavour 0: scalar libm form (normal_fill_16<float>, hosts without AVX2).
inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

template <bool FMA> __attribute__((always_inline)) inline float madd(float a, float b, float c) {
    if (FMA) return __builtin_fmaf(a, b, c);
    const float t = a * b;
    return t + c;
}

template <bool FMA> __attribute__((always_inline)) inline float log256(float x) {
    const bool invalid = x <= 0.f;
    x = x > 1.17549435e-38f ? x : 1.17549435e-38f;                  // min_norm_pos
    int32_t imm0 = (int32_t)(f2u(x) >> 23);
    x = u2f((f2u(x) & ~0x7f800000u) | 0x3f000000u);                  // mantissa in [0.5, 1)
    imm0 -= 0x7f;
    float e = (float)imm0;
    e = e + 1.f;
    const bool mask = x < 0.707106781186547524f;                     // cephes_SQRTHF
    const float tmp0 = mask ? x : 0.f;
    x = x - 1.f;
    e = e - (mask ? 1.f : 0.f);
    x = x + tmp0;
    const float z = x * x;
    float y = 7.0376836292E-2f;
    y = madd<FMA>(y, x, -1.1514610310E-1f);
    y = madd<FMA>(y, x, 1.1676998740E-1f);
    y = madd<FMA>(y, x, -1.2420140846E-1f);
    y = madd<FMA>(y, x, +1.4249322787E-1f);
    y = madd<FMA>(y, x, -1.6668057665E-1f);
    y = madd<FMA>(y, x, +2.0000714765E-1f);
    y = madd<FMA>(y, x, -2.4999993993E-1f);
    y = madd<FMA>(y, x, +3.3333331174E-1f);
    y = y * x;
    // y = y*z + e*log_q1: the contraction pass fuses the FIRST multiplication (in statement order)
    // whose only use is the addition, i.e. y*z, with e*log_q1 rounded on its own
    y = madd<FMA>(y, z, e * -2.12194440e-4f);
    y = FMA ? __builtin_fmaf(-z, 0.5f, y) : (y - z * 0.5f);          // y -= z * 0.5
    x = x + y;
    x = madd<FMA>(e, 0.693359375f, x);                               // x += e * log_q2
    return invalid ? u2f(0xffffffffu) : x;
}

template <bool FMA> __attribute__((always_inline)) inline void sincos256(float xin, float* s, float* c) {
    uint32_t sign_bit_sin = f2u(xin) & 0x80000000u;
    float x = u2f(f2u(xin) & 0x7fffffffu);
    float y = x * 1.27323954473516f;                                 // 4/pi
    int32_t imm2 = (int32_t)y;                                       // cvttps
    imm2 = (imm2 + 1) & ~1;
    y = (float)imm2;
    int32_t imm4 = imm2;
    const uint32_t swap_sign_bit_sin = ((uint32_t)(imm2 & 4)) << 29;
    const bool poly_mask = (imm2 & 2) == 0;
    // extended precision modular arithmetic: x = ((x - y*DP1) - y*DP2) - y*DP3
    x = madd<FMA>(y, -0.78515625f, x);
    x = madd<FMA>(y, -2.4187564849853515625e-4f, x);
    x = madd<FMA>(y, -3.77489497744594108e-8f, x);
    imm4 = imm4 - 2;
    const uint32_t sign_bit_cos = ((uint32_t)(~imm4 & 4)) << 29;
    sign_bit_sin ^= swap_sign_bit_sin;
    const float z = x * x;
    float yc = 2.443315711809948E-005f;
    yc = madd<FMA>(yc, z, -1.388731625493765E-003f);
    yc = madd<FMA>(yc, z, 4.166664568298827E-002f);
    yc = yc * z;
    yc = FMA ? __builtin_fmaf(yc, z, -(z * 0.5f)) : (yc * z - z * 0.5f);   // same rule as in log256
    yc = yc + 1.f;
    float ys = -1.9515295891E-4f;
    ys = madd<FMA>(ys, z, 8.3321608736E-3f);
    ys = madd<FMA>(ys, z, -1.6666654611E-1f);
    ys = ys * z;
    ys = madd<FMA>(ys, x, x);
    // select: poly_mask lanes take the sine polynomial for sin and the cosine one for cos.
    // (the source forms  ysin1 + ysin2  and  (y - ysin1) + (y2 - ysin2): one addend is +0)
    const float ysin2 = poly_mask ? ys : 0.f, ysin1 = poly_mask ? 0.f : yc;
    const float y2r = ys - ysin2, yr = yc - ysin1;
    const float xs = ysin1 + ysin2, xc = yr + y2r;
    *s = u2f(f2u(xs) ^ sign_bit_sin);
    *c = u2f(f2u(xc) ^ sign_bit_cos);
}

template <bool FMA> __attribute__((always_inline)) inline void normal16_mathfun(float* data) {
    const float two_pi = (float)(2.0f * 3.14159265358979323846264338327950288);
    for (int j = 0; j < 8; ++j) {
        const float u1 = 1.f - data[j];
        const float u2 = data[j + 8];
        const float radius = __builtin_sqrtf(-2.f * log256<FMA>(u1));
        const float theta = two_pi * u2;
        float sn, cs;
        sincos256<FMA>(theta, &sn, &cs);
        const float n1 = radius * cs, n2 = radius * sn;
        data[j] = __builtin_fmaf(n1, 1.f, 0.f);                      // _mm256_fmadd_ps(n, std, mean)
        data[j + 8] = __builtin_fmaf(n2, 1.f, 0.f);
    }
}
// The same arithmetic on 8 lanes at a time (the scalar templates above are the readable statement
// and the reference for the CPU tests; gm_host_replay_flavour(3|4) selects them).
#define GM_AVX2 __attribute__((target("avx2,fma"), always_inline)) inline
template <bool FMA> GM_AVX2 __m256 vmadd(__m256 a, __m256 b, __m256 c) {
    if (FMA) return _mm256_fmadd_ps(a, b, c);
    return _mm256_add_ps(_mm256_mul_ps(a, b), c);
}
template <bool FMA> GM_AVX2 __m256 vlog256(__m256 x) {
    const __m256 one = _mm256_set1_ps(1.f);
    const __m256 invalid = _mm256_cmp_ps(x, _mm256_setzero_ps(), _CMP_LE_OS);
    x = _mm256_max_ps(x, _mm256_set1_ps(1.17549435e-38f));
    __m256i imm0 = _mm256_srli_epi32(_mm256_castps_si256(x), 23);
    x = _mm256_and_ps(x, _mm256_castsi256_ps(_mm256_set1_epi32(~0x7f800000)));
    x = _mm256_or_ps(x, _mm256_set1_ps(0.5f));
    imm0 = _mm256_sub_epi32(imm0, _mm256_set1_epi32(0x7f));
    __m256 e = _mm256_add_ps(_mm256_cvtepi32_ps(imm0), one);
    const __m256 mask = _mm256_cmp_ps(x, _mm256_set1_ps(0.707106781186547524f), _CMP_LT_OS);
    const __m256 tmp0 = _mm256_and_ps(x, mask);
    x = _mm256_sub_ps(x, one);
    e = _mm256_sub_ps(e, _mm256_and_ps(one, mask));
    x = _mm256_add_ps(x, tmp0);
    const __m256 z = _mm256_mul_ps(x, x);
    __m256 y = _mm256_set1_ps(7.0376836292E-2f);
    y = vmadd<FMA>(y, x, _mm256_set1_ps(-1.1514610310E-1f));
    y = vmadd<FMA>(y, x, _mm256_set1_ps(1.1676998740E-1f));
    y = vmadd<FMA>(y, x, _mm256_set1_ps(-1.2420140846E-1f));
    y = vmadd<FMA>(y, x, _mm256_set1_ps(+1.4249322787E-1f));
    y = vmadd<FMA>(y, x, _mm256_set1_ps(-1.6668057665E-1f));
    y = vmadd<FMA>(y, x, _mm256_set1_ps(+2.0000714765E-1f));
    y = vmadd<FMA>(y, x, _mm256_set1_ps(-2.4999993993E-1f));
    y = vmadd<FMA>(y, x, _mm256_set1_ps(+3.3333331174E-1f));
    y = _mm256_mul_ps(y, x);
    y = vmadd<FMA>(y, z, _mm256_mul_ps(e, _mm256_set1_ps(-2.12194440e-4f)));
    if (FMA) y = _mm256_fnmadd_ps(z, _mm256_set1_ps(0.5f), y);
    else y = _mm256_sub_ps(y, _mm256_mul_ps(z, _mm256_set1_ps(0.5f)));
    x = _mm256_add_ps(x, y);
    x = vmadd<FMA>(e, _mm256_set1_ps(0.693359375f), x);
    return _mm256_or_ps(x, invalid);
}
template <bool FMA> GM_AVX2 void vsincos256(__m256 xin, __m256* s, __m256* c) {
    const __m256 signmask = _mm256_castsi256_ps(_mm256_set1_epi32((int)0x80000000u));
    __m256 sign_bit_sin = _mm256_and_ps(xin, signmask);
    __m256 x = _mm256_andnot_ps(signmask, xin);
    __m256 y = _mm256_mul_ps(x, _mm256_set1_ps(1.27323954473516f));
    __m256i imm2 = _mm256_cvttps_epi32(y);
    imm2 = _mm256_and_si256(_mm256_add_epi32(imm2, _mm256_set1_epi32(1)), _mm256_set1_epi32(~1));
    y = _mm256_cvtepi32_ps(imm2);
    __m256i imm4 = imm2;
    const __m256 swap_sign_bit_sin =
        _mm256_castsi256_ps(_mm256_slli_epi32(_mm256_and_si256(imm2, _mm256_set1_epi32(4)), 29));
    const __m256 poly_mask = _mm256_castsi256_ps(
        _mm256_cmpeq_epi32(_mm256_and_si256(imm2, _mm256_set1_epi32(2)), _mm256_setzero_si256()));
    x = vmadd<FMA>(y, _mm256_set1_ps(-0.78515625f), x);
    x = vmadd<FMA>(y, _mm256_set1_ps(-2.4187564849853515625e-4f), x);
    x = vmadd<FMA>(y, _mm256_set1_ps(-3.77489497744594108e-8f), x);
    imm4 = _mm256_sub_epi32(imm4, _mm256_set1_epi32(2));
    imm4 = _mm256_slli_epi32(_mm256_andnot_si256(imm4, _mm256_set1_epi32(4)), 29);
    const __m256 sign_bit_cos = _mm256_castsi256_ps(imm4);
    sign_bit_sin = _mm256_xor_ps(sign_bit_sin, swap_sign_bit_sin);
    const __m256 z = _mm256_mul_ps(x, x);
    __m256 yc = _mm256_set1_ps(2.443315711809948E-005f);
    yc = vmadd<FMA>(yc, z, _mm256_set1_ps(-1.388731625493765E-003f));
    yc = vmadd<FMA>(yc, z, _mm256_set1_ps(4.166664568298827E-002f));
    yc = _mm256_mul_ps(yc, z);
    const __m256 hz = _mm256_mul_ps(z, _mm256_set1_ps(0.5f));
    if (FMA) yc = _mm256_fmsub_ps(yc, z, hz); else yc = _mm256_sub_ps(_mm256_mul_ps(yc, z), hz);
    yc = _mm256_add_ps(yc, _mm256_set1_ps(1.f));
    __m256 ys = _mm256_set1_ps(-1.9515295891E-4f);
    ys = vmadd<FMA>(ys, z, _mm256_set1_ps(8.3321608736E-3f));
    ys = vmadd<FMA>(ys, z, _mm256_set1_ps(-1.6666654611E-1f));
    ys = _mm256_mul_ps(ys, z);
    ys = vmadd<FMA>(ys, x, x);
    const __m256 ysin2 = _mm256_and_ps(poly_mask, ys), ysin1 = _mm256_andnot_ps(poly_mask, yc);
    const __m256 y2r = _mm256_sub_ps(ys, ysin2), yr = _mm256_sub_ps(yc, ysin1);
    *s = _mm256_xor_ps(_mm256_add_ps(ysin1, ysin2), sign_bit_sin);
    *c = _mm256_xor_ps(_mm256_add_ps(yr, y2r), sign_bit_cos);
}
template <bool FMA> GM_AVX2 void vnormal16(float* data) {
    const __m256 u1 = _mm256_sub_ps(_mm256_set1_ps(1.f), _mm256_loadu_ps(data));
    const __m256 u2 = _mm256_loadu_ps(data + 8);
    const __m256 radius = _mm256_sqrt_ps(_mm256_mul_ps(_mm256_set1_ps(-2.f), vlog256<FMA>(u1)));
    const __m256 theta = _mm256_mul_ps(_mm256_set1_ps((float)(2.0f * 3.14159265358979323846264338327950288)), u2);
    __m256 sn, cs;
    vsincos256<FMA>(theta, &sn, &cs);
    const __m256 one = _mm256_set1_ps(1.f), zero = _mm256_setzero_ps();
    _mm256_storeu_ps(data, _mm256_fmadd_ps(_mm256_mul_ps(radius, cs), one, zero));
    _mm256_storeu_ps(data + 8, _mm256_fmadd_ps(_mm256_mul_ps(radius, sn), one, zero));
}
__attribute__((target("avx2,fma"))) void normal_groups_fma(float* p, int64_t groups) {
    for (int64_t g = 0; g < groups; ++g) vnormal16<true>(p + 16 * g);
}
__attribute__((target("avx2,fma"))) void normal_groups_nofma(float* p, int64_t groups) {
    for (int64_t g = 0; g < groups; ++g) vnormal16<false>(p + 16 * g);
}
__attribute__((target("avx2,fma"))) void normal_groups_fma_scalar(float* p, int64_t groups) {
    for (int64_t g = 0; g < groups; ++g) normal16_mathfun<true>(p + 16 * g);
}
__attribute__((target("avx2,fma"))) void normal_groups_nofma_scalar(float* p, int64_t groups) {
    for (int64_t g = 0; g < groups; ++g) normal16_mathfun<false>(p + 16 * g);
}
void normal_groups_libm(float* p, int64_t groups) {
    for (int64_t g = 0; g < groups; ++g) {
        float* data = p + 16 * g;
        for (int j = 0; j < 8; ++j) {
            const float u1 = 1 - data[j];                 // [0,1) -> (0,1] for the log
            const float u2 = data[j + 8];
            const float radius = std::sqrt(-2 * std::log(u1));
            const float theta = (float)((double)(2.0f) * 3.14159265358979323846264338327950288 * (double)u2);
            data[j] = radius * std::cos(theta);           // * std (1) + mean (0): exact no-ops
            data[j + 8] = radius * std::sin(theta);
        }
    }
}
std::atomic<int> g_flavour{1};       // 0 libm, 1 avx_mathfun with FMA contraction, 2 without, 3/4: scalar forms of 1/2
inline void normal_groups(float* p, int64_t groups) {
    const int f = g_flavour.load(std::memory_order_relaxed);
    if (f == 1) normal_groups_fma(p, groups);
    else if (f == 2) normal_groups_nofma(p, groups);
    else if (f == 3) normal_groups_fma_scalar(p, groups);
    else if (f == 4) normal_groups_nofma_scalar(p, groups);
    else normal_groups_libm(p, groups);
}
inline void normal16(float* data) { normal_groups(data, 1); }

// ---- worker pool for the Box-Muller stage ----------------------------------------------------
class Pool {
  public:
    explicit Pool(int n) : stop_(false), gen_(0), pending_(0) {
        for (int i = 0; i < n; ++i) th_.emplace_back([this, i] { loop(i); });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> l(mu_); stop_ = true; ++gen_; }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    int size() const { return (int)th_.size(); }
    // run fn(part, nparts) on every worker and on the caller (part 0)
    template <class F> void run(F&& fn) {
        const int np = size() + 1;
        fn_ = [&](int part) { fn(part, np); };
        { std::lock_guard<std::mutex> l(mu_); pending_ = size(); ++gen_; }
        cv_.notify_all();
        fn(0, np);
        // the parts are tens of microseconds long: spin briefly, then sleep
        for (int spin = 0; spin < 20000 && pending_.load(std::memory_order_acquire) > 0; ++spin) {}
        std::unique_lock<std::mutex> l(mu_);
        done_.wait(l, [this] { return pending_.load() == 0; });
    }

  private:
    void loop(int i) {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> l(mu_);
                cv_.wait(l, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
            }
            fn_(i + 1);
            if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                std::lock_guard<std::mutex> l(mu_);
                done_.notify_all();
            }
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    bool stop_;
    uint64_t gen_;
    std::atomic<int> pending_;
    std::function<void(int)> fn_;
};

Pool* g_pool = nullptr;
std::mutex g_pool_mu;

struct Span { float* p; int64_t groups; };          // `groups` consecutive 16-float groups

}  // namespace

// One draw of the per-iteration program (include/gm_hip.h).
struct gm_draw_op {
    int32_t kind;         // GM_DRAW_*
    int32_t n;            // SAMPLER: B; NORMAL/UNIFORM: elements; INFO: B
    int64_t a;            // SAMPLER: dataset rows N; INFO: z_dim
    int32_t b, c;         // INFO: disc_dim, cont_dim
    void* dst;            // host destination of iteration 0
    int64_t iter_stride;  // bytes between consecutive iterations' destinations
    int64_t e0, e1;       // NORMAL/UNIFORM: element range to materialise ([0,n) = all); the stream
                          // always advances as for the whole tensor
};
enum { GM_DRAW_SAMPLER = 0, GM_DRAW_NORMAL = 1, GM_DRAW_UNIFORM = 2, GM_DRAW_INFO = 3 };

extern "C" int gm_host_replay_flavour(int flavour) {
    if (flavour < 0 || flavour > 4) return GM_EINVAL;
    if (flavour != 0 && !(__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma"))) return GM_EUNSUPPORTED;
    g_flavour.store(flavour);
    return 0;
}

extern "C" int gm_host_replay_threads(int n_threads) {
    std::lock_guard<std::mutex> l(g_pool_mu);
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 64) n_threads = 64;
    if (g_pool && g_pool->size() + 1 == n_threads) return 0;
    delete g_pool;
    g_pool = (n_threads > 1) ? new Pool(n_threads - 1) : nullptr;
    return 0;
}

namespace {

// uniforms of a normal_ call on n >= 16 contiguous floats, written to dst[0,n) (+ 16 scratch
// floats for the ragged tail, tail_u); Box-Muller groups are queued in `spans`.
int normal_uniforms(Mt& m, float* dst, int64_t n, int64_t e0, int64_t e1, float* tail_u,
                    std::vector<Span>& spans, bool* tail_pending) {
    *tail_pending = false;
    if (n < 16) return GM_EUNSUPPORTED;               // torch takes the scalar double path there
    const bool all = (e0 == 0 && e1 == n);
    if (all) {
        mt_uniform(m, dst, n);
        spans.push_back({dst, n / 16});
        if (n % 16 != 0) {
            // normal_fill: recompute the last 16 values from 16 NEW uniforms
            mt_uniform(m, tail_u, 16);
            *tail_pending = true;
        }
        return 0;
    }
    // partial materialisation (data-parallel rank): only 16-aligned ranges of a 16-multiple tensor
    if (n % 16 != 0 || e0 % 16 != 0 || e1 % 16 != 0 || e0 < 0 || e1 > n || e0 >= e1) return GM_EUNSUPPORTED;
    mt_skip(m, e0);
    mt_uniform(m, dst + e0, e1 - e0);
    mt_skip(m, n - e1);
    spans.push_back({dst + e0, (e1 - e0) / 16});
    return 0;
}

void run_spans(const std::vector<Span>& spans) {
    int64_t total = 0;
    for (const auto& s : spans) total += s.groups;
    if (total == 0) return;
    auto work = [&](int part, int nparts) {
        const int64_t lo = total * part / nparts, hi = total * (part + 1) / nparts;
        int64_t base = 0;
        for (const auto& s : spans) {
            const int64_t a = lo > base ? lo - base : 0;
            const int64_t b = (hi - base) < s.groups ? (hi - base) : s.groups;
            if (b > a) normal_groups(s.p + 16 * a, b - a);
            base += s.groups;
            if (base >= hi) break;
        }
    };
    Pool* pool;
    { std::lock_guard<std::mutex> l(g_pool_mu); pool = g_pool; }
    if (pool && total >= 256) pool->run(work); else work(0, 1);
}

}  // namespace

// ------------------------------------------------------------------------------------------
// O(B) prefix of torch.randperm(n) for a freshly seeded CPU generator (see gm_hip.h):
// at::native::randperm_cpu is a forward Fisher-Yates, r[i] <-> r[i + random() % (n - i)], so the
// first B outputs need B draws.  The permutation array is a thread-local dense int32 identity that
// survives between calls: the 2B entries a call touches are put back afterwards (an undo walk
// instead of a hash table: a data-parallel run draws the GLOBAL batch's indices on every rank, 8192
// per step at 8 x 1024 rows -- the table cost 20 ns per index, this costs ~3).  The draws come from
// the vectorised twist / temper above, one block of 624 at a time; random() % range is a 32-bit
// unsigned remainder (both operands < 2^32; torch computes it in 64 bits, same value).
// ------------------------------------------------------------------------------------------
extern "C" int gm_randperm_prefix(uint64_t seed, int64_t n, int B, int64_t* out) {
    if (!(out && n > 0 && B > 0 && B <= n && n < (int64_t)(0xffffffffu / 20))) {
        gm_set_error("bad argument: gm_randperm_prefix(seed, n, B, out)");
        return GM_EINVAL;
    }
    static thread_local std::vector<int32_t> perm;       // identity between calls
    if ((int64_t)perm.size() < n) {
        const size_t old = perm.size();
        perm.resize((size_t)n);
        for (size_t i = old; i < (size_t)n; ++i) perm[i] = (int32_t)i;
    }
    static thread_local std::vector<int32_t> touched;
    touched.clear();
    touched.reserve((size_t)B);
    // mt19937 seeded like at::mt19937(seed) (init_with_uint32), first use twists
    Mt m;
    m.s[0] = (uint32_t)(seed & 0xffffffffu);
    for (int j = 1; j < 624; ++j) m.s[j] = 1812433253u * (m.s[j - 1] ^ (m.s[j - 1] >> 30)) + (uint32_t)j;
    m.next = 624; m.remain = 0;
    uint32_t block[624];
    int have = 0, at = 0;
    int32_t* r = perm.data();
    const int64_t draws = (B < n) ? B : n - 1;            // the last element of a full permutation: no draw
    for (int64_t i = 0; i < draws; ++i) {
        if (at == have) {
            const int64_t left = draws - i;
            have = (int)(left < 624 ? left : 624);
            mt_raw(m, block, have);
            at = 0;
        }
        const uint32_t z = block[at++] % (uint32_t)(n - i);
        const int64_t j = i + (int64_t)z;
        const int32_t vi = r[i], vj = r[j];
        r[i] = vj; r[j] = vi;
        touched.push_back((int32_t)j);
        out[i] = (int64_t)vj;
    }
    if (draws < B) out[B - 1] = (int64_t)r[B - 1];
    // undo: every touched slot back to identity (positions i < draws and the recorded partners)
    for (int64_t i = 0; i < draws; ++i) r[i] = (int32_t)i;
    for (int32_t j : touched) r[j] = j;
    return 0;
}

// Replay `n_iters` iterations of the per-iteration draw program `ops[0..n_ops)` from the
// serialized torch CPU generator state (torch.get_rng_state(): 5056 bytes), advancing it exactly as
// the reference's calls would.  See include/gm_hip.h.
extern "C" int gm_host_replay(void* torch_cpu_rng_state, int64_t state_bytes, const gm_draw_op* ops,
                              int n_ops, int n_iters) {
    if (!torch_cpu_rng_state || state_bytes != (int64_t)sizeof(TorchCpuGenState) || !ops || n_ops < 0 ||
        n_iters < 0) {
        gm_set_error("gm_host_replay: bad arguments / unknown generator state layout");
        return GM_EINVAL;
    }
    TorchCpuGenState* g = reinterpret_cast<TorchCpuGenState*>(torch_cpu_rng_state);
    Mt m;
    if (mt_load(m, g)) { gm_set_error("gm_host_replay: generator state not understood"); return GM_EINVAL; }
    std::vector<Span> spans;
    struct Tail { float* dst; float u[16]; };
    std::vector<Tail> tails;
    tails.reserve((size_t)n_ops * (size_t)n_iters + 1);
    for (int it = 0; it < n_iters; ++it) {
        for (int k = 0; k < n_ops; ++k) {
            const gm_draw_op& op = ops[k];
            char* dst = reinterpret_cast<char*>(op.dst) + (int64_t)it * op.iter_stride;
            switch (op.kind) {
            case GM_DRAW_SAMPLER: {
                // _BaseDataLoaderIter.__init__: base seed (unused with 0 workers);
                // RandomSampler.__iter__: seed = int(torch.empty((), int64).random_().item())
                // random_() on int64: random64() % (2^63)   (uniform_int_distribution<int64_t>)
                (void)mt_random64(m);
                const uint64_t seed = mt_random64(m) & 0x7fffffffffffffffull;
                const int rc = gm_randperm_prefix(seed, op.a, op.n, reinterpret_cast<int64_t*>(dst));
                if (rc) return rc;
                break;
            }
            case GM_DRAW_NORMAL: {
                tails.push_back(Tail{});
                bool tp = false;
                const int rc = normal_uniforms(m, reinterpret_cast<float*>(dst), op.n, op.e0, op.e1,
                                               tails.back().u, spans, &tp);
                if (rc) { gm_set_error("gm_host_replay: normal_ shape outside the restated path"); return rc; }
                if (tp) tails.back().dst = reinterpret_cast<float*>(dst) + op.n - 16; else tails.pop_back();
                break;
            }
            case GM_DRAW_UNIFORM: {
                float* d = reinterpret_cast<float*>(dst);
                if (op.e0 == 0 && op.e1 == op.n) mt_uniform(m, d, op.n);
                else {
                    if (op.e0 < 0 || op.e1 > op.n || op.e0 > op.e1) return GM_EINVAL;
                    mt_skip(m, op.e0); mt_uniform(m, d + op.e0, op.e1 - op.e0); mt_skip(m, op.n - op.e1);
                }
                break;
            }
            case GM_DRAW_INFO: {
                // info_gan.py:312-323: z = randn(B, zd); cat = randint(0, nd, (B,)); c = randn(B, nc)
                // packed as rows [z | one_hot(cat) | c]
                const int B = op.n, zd = (int)op.a, nd = op.b, nc = op.c, W = zd + nd + nc;
                if ((int64_t)B * zd < 16 || (int64_t)B * nc < 16 || (B * zd) % 16 || (B * nc) % 16) {
                    gm_set_error("gm_host_replay: InfoGAN noise shape outside the restated path");
                    return GM_EUNSUPPORTED;
                }
                float* d = reinterpret_cast<float*>(dst);
                std::vector<float> zz((size_t)B * zd), cc((size_t)B * nc);
                std::vector<uint32_t> cat(B);
                mt_uniform(m, zz.data(), (int64_t)B * zd);
                mt_raw(m, cat.data(), B);             // random_from_to: random() % range + base
                mt_uniform(m, cc.data(), (int64_t)B * nc);
                normal_groups(zz.data(), (int64_t)B * zd / 16);
                normal_groups(cc.data(), (int64_t)B * nc / 16);
                for (int r = 0; r < B; ++r) {
                    float* row = d + (int64_t)r * W;
                    std::memcpy(row, zz.data() + (int64_t)r * zd, sizeof(float) * zd);
                    for (int j = 0; j < nd; ++j) row[zd + j] = 0.f;
                    row[zd + (int)(cat[r] % (uint32_t)nd)] = 1.f;
                    std::memcpy(row + zd + nd, cc.data() + (int64_t)r * nc, sizeof(float) * nc);
                }
                break;
            }
            default:
                gm_set_error("gm_host_replay: unknown draw kind");
                return GM_EINVAL;
            }
        }
    }
    run_spans(spans);
    for (auto& t : tails) {                           // ragged tails: after the main groups
        normal16(t.u);
        std::memcpy(t.dst, t.u, sizeof(t.u));
    }
    mt_store(m, g);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Fill worker: gm_host_replay jobs run by ONE persistent native thread, in submission order, on a
// generator state buffer the caller keeps alive -- the Python side only enqueues (microseconds, no
// interpreter lock hand-off between the thread that launches graphs and the one that draws).  After
// a job's ring slots are written the worker advances the fill gate (gm_stage_in_gated) with a release
// store.  A failed job is sticky: later jobs are not run (their gate never opens; the submitter sees
// the error at gm_fill_wait and opens the gates itself).
// ------------------------------------------------------------------------------------------------
namespace {

struct FillJob {
    int64_t id;
    void* state;
    int64_t state_bytes;
    std::vector<gm_draw_op> ops;
    int n_iters;
    int64_t* gate;
    int64_t gate_value;
};

struct FillWorker {
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    std::vector<FillJob> q;                    // FIFO (small: a handful of jobs in flight)
    size_t head = 0;
    int64_t submitted = 0;
    std::atomic<int64_t> completed{0};
    std::atomic<int> rc{0};
    std::thread th;

    FillWorker() : th([this] { loop(); }) { th.detach(); }

    void loop() {
        for (;;) {
            FillJob job;
            {
                std::unique_lock<std::mutex> l(mu);
                // a short spin before sleeping: consecutive sub-chunks arrive microseconds apart
                for (int spin = 0; head == q.size() && spin < 2000; ++spin) {
                    l.unlock();
                    _mm_pause();
                    l.lock();
                }
                cv_job.wait(l, [this] { return head < q.size(); });
                job = std::move(q[head++]);
                if (head == q.size()) { q.clear(); head = 0; }
            }
            if (rc.load() == 0) {
                const int r = gm_host_replay(job.state, job.state_bytes, job.ops.data(), (int)job.ops.size(),
                                             job.n_iters);
                if (r != 0) rc.store(r);
                else if (job.gate) __atomic_store_n(job.gate, job.gate_value, __ATOMIC_RELEASE);
            }
            {
                std::lock_guard<std::mutex> l(mu);
                completed.store(job.id);
            }
            cv_done.notify_all();
        }
    }
};

FillWorker* fill_worker() {
    static FillWorker* w = new FillWorker();   // never destroyed: the thread outlives static teardown
    return w;
}

}  // namespace

extern "C" int64_t gm_fill_submit(void* torch_cpu_rng_state, int64_t state_bytes, const gm_draw_op* ops,
                                  int n_ops, int n_iters, int64_t* gate, int64_t gate_value) {
    if (!torch_cpu_rng_state || !ops || n_ops <= 0 || n_iters <= 0) {
        gm_set_error("gm_fill_submit: bad arguments");
        return GM_EINVAL;
    }
    FillWorker* w = fill_worker();
    FillJob job;
    job.state = torch_cpu_rng_state; job.state_bytes = state_bytes;
    job.ops.assign(ops, ops + n_ops);
    job.n_iters = n_iters; job.gate = gate; job.gate_value = gate_value;
    int64_t id;
    {
        std::lock_guard<std::mutex> l(w->mu);
        id = job.id = ++w->submitted;
        w->q.push_back(std::move(job));
    }
    w->cv_job.notify_one();
    return id;
}

extern "C" int64_t gm_fill_completed(void) { return fill_worker()->completed.load(); }

extern "C" int gm_fill_wait(int64_t id) {
    FillWorker* w = fill_worker();
    if (w->completed.load() < id) {
        std::unique_lock<std::mutex> l(w->mu);
        w->cv_done.wait(l, [&] { return w->completed.load() >= id; });
    }
    return w->rc.load();
}

// Clears a sticky error once every submitted job has been retired (the run that hit it is over).
extern "C" int gm_fill_reset(void) {
    FillWorker* w = fill_worker();
    std::unique_lock<std::mutex> l(w->mu);
    w->cv_done.wait(l, [&] { return w->completed.load() >= w->submitted; });
    w->rc.store(0);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// numpy's LEGACY global generator (np.random.normal, what bir_vae.py:92-94 draws its noise from):
// RandomState = MT19937 + the polar Box-Muller of numpy/random/src/legacy/legacy-distributions.c
//     legacy_double: a = next32 >> 5, b = next32 >> 6, (a * 67108864.0 + b) / 9007199254740992.0
//     legacy_gauss : cached second value first; else candidates x1, x2 = 2 u - 1 until 0 < r2 < 1,
//                    f = sqrt(-2.0 * log(r2) / r2), keep f * x1, return f * x2
//     legacy_normal: loc + scale * legacy_gauss
// restated so that the 100 us of log / sqrt / divide per 10 240 samples can run on several threads: the
// uniform stream and the accept / reject decisions are sequential (phase 1, ~4 ns per candidate), the
// transform of the accepted pairs is not (phase 2, the pool).  Same libm, no FMA contraction (this file is
// built with -ffp-contract=off): bit-identical doubles, rounded to float as `.float()` does.
// state: key[624] / pos / has_gauss / gauss exactly as np.random.get_state(legacy=True) returns them.
// ------------------------------------------------------------------------------------------------
namespace {
Pool* g_np_pool = nullptr;
std::mutex g_np_mu;

// Eight candidates of the polar method (32 tempered words) at once: the same IEEE operations as the scalar loop
// (every step before r2 is exact in double; r2 = x1*x1 + x2*x2 as two products and a sum -- this file is built
// without FMA contraction), accepted ones appended in order.  Returns the number accepted.
__attribute__((target("avx512f"))) int np_candidates8_avx512(const uint32_t* w, double* x1o, double* x2o, double* r2o) {
    const __m512i A = _mm512_loadu_si512((const void*)w), B = _mm512_loadu_si512((const void*)(w + 16));
    const __m512i i0 = _mm512_setr_epi32(0, 4, 8, 12, 16, 20, 24, 28, 0, 0, 0, 0, 0, 0, 0, 0);
    const __m512i one_i = _mm512_set1_epi32(1);
    const __m512i i1 = _mm512_add_epi32(i0, one_i), i2 = _mm512_add_epi32(i1, one_i), i3 = _mm512_add_epi32(i2, one_i);
    const __m256i w0 = _mm512_castsi512_si256(_mm512_permutex2var_epi32(A, i0, B));
    const __m256i w1 = _mm512_castsi512_si256(_mm512_permutex2var_epi32(A, i1, B));
    const __m256i w2 = _mm512_castsi512_si256(_mm512_permutex2var_epi32(A, i2, B));
    const __m256i w3 = _mm512_castsi512_si256(_mm512_permutex2var_epi32(A, i3, B));
    const __m512d k26 = _mm512_set1_pd(67108864.0), k53 = _mm512_set1_pd(1.0 / 9007199254740992.0);
    const __m512d two = _mm512_set1_pd(2.0), one = _mm512_set1_pd(1.0);
    const __m512d a1 = _mm512_cvtepu32_pd(_mm256_srli_epi32(w0, 5)), b1 = _mm512_cvtepu32_pd(_mm256_srli_epi32(w1, 6));
    const __m512d a2 = _mm512_cvtepu32_pd(_mm256_srli_epi32(w2, 5)), b2 = _mm512_cvtepu32_pd(_mm256_srli_epi32(w3, 6));
    // (a * 2^26 + b) / 2^53: the product, the sum (< 2^53) and the scaling by a power of two are all exact
    const __m512d u1 = _mm512_mul_pd(_mm512_add_pd(_mm512_mul_pd(a1, k26), b1), k53);
    const __m512d u2 = _mm512_mul_pd(_mm512_add_pd(_mm512_mul_pd(a2, k26), b2), k53);
    const __m512d x1 = _mm512_sub_pd(_mm512_mul_pd(two, u1), one), x2 = _mm512_sub_pd(_mm512_mul_pd(two, u2), one);
    const __m512d r2 = _mm512_add_pd(_mm512_mul_pd(x1, x1), _mm512_mul_pd(x2, x2));
    const __mmask8 m = _mm512_cmp_pd_mask(r2, one, _CMP_LT_OQ) & _mm512_cmp_pd_mask(r2, _mm512_setzero_pd(), _CMP_NEQ_OQ);
    _mm512_mask_compressstoreu_pd(x1o, m, x1);
    _mm512_mask_compressstoreu_pd(x2o, m, x2);
    _mm512_mask_compressstoreu_pd(r2o, m, r2);
    return __builtin_popcount((unsigned)m);
}
}

extern "C" int gm_numpy_legacy_normal_f32(uint32_t* key, int32_t* pos_io, int32_t* has_gauss_io, double* gauss_io,
                                          double loc, double scale, int64_t n, float* out, int n_threads) {
    if (!key || !pos_io || !has_gauss_io || !gauss_io || !out || n < 0 || *pos_io < 0 || *pos_io > 624) {
        gm_set_error("gm_numpy_legacy_normal_f32: bad arguments");
        return GM_EINVAL;
    }
    if (n == 0) return 0;
    std::lock_guard<std::mutex> guard(g_np_mu);
    int64_t i0 = 0;
    if (*has_gauss_io) {                                   // the cached half of the previous pair comes first
        out[0] = (float)(loc + scale * *gauss_io);
        *has_gauss_io = 0; *gauss_io = 0.0;
        i0 = 1;
    }
    const int64_t m = n - i0;                              // values still to produce
    if (m == 0) return 0;
    const int64_t P = (m + 1) / 2;                         // accepted pairs needed
    // phase 1: tempered words -> candidates -> the first P accepted (x1, x2, r2).  Buffers persist across calls
    // (guarded by g_np_mu): a chunk of 32 batches is 2 MB of them, and fresh pages cost more than the arithmetic.
    alignas(64) uint32_t st[624 + 16];
    std::memcpy(st, key, 624 * sizeof(uint32_t));
    int pos = *pos_io;
    alignas(64) static uint32_t wb[624 + 16];              // tempered words of the current block (+ <= 3 carried over)
    int have = 0, rd = 0;
    static std::vector<double> X1, X2, R2;
    if ((int64_t)X1.size() < P) { X1.resize((size_t)P); X2.resize((size_t)P); R2.resize((size_t)P); }
    else if (X1.size() > ((size_t)1 << 20) && (size_t)P < X1.size() / 8) {
        // one very large request must not pin 3 x 8 bytes per pair for the life of the process
        std::vector<double>((size_t)P).swap(X1); std::vector<double>((size_t)P).swap(X2); std::vector<double>((size_t)P).swap(R2);
    }
    int64_t got = 0;
    static const bool wide = cpu_level() == 2 && !getenv("GM_NUMPY_SCALAR");
    while (got < P) {
        if (have - rd < 4) {                               // next block: carry the unread words to the front
            const int left = have - rd;
            for (int i = 0; i < left; ++i) wb[i] = wb[rd + i];
            if (pos >= 624) { twist(st); pos = 0; }
            temper(st + pos, wb + left, 624 - pos);
            have = left + (624 - pos); rd = 0; pos = 624;
            continue;
        }
        if (wide && have - rd >= 32 && P - got >= 8) {       // eight candidates at once, never past the P-th pair
            got += np_candidates8_avx512(wb + rd, X1.data() + got, X2.data() + got, R2.data() + got);
            rd += 32;
            continue;
        }
        const uint32_t* w = wb + rd;
        rd += 4;
        const double u1 = ((double)(w[0] >> 5) * 67108864.0 + (double)(w[1] >> 6)) / 9007199254740992.0;
        const double u2 = ((double)(w[2] >> 5) * 67108864.0 + (double)(w[3] >> 6)) / 9007199254740992.0;
        const double x1 = 2.0 * u1 - 1.0, x2 = 2.0 * u2 - 1.0;
        const double r2 = x1 * x1 + x2 * x2;
        if (r2 >= 1.0 || r2 == 0.0) continue;
        X1[(size_t)got] = x1; X2[(size_t)got] = x2; R2[(size_t)got] = r2;
        ++got;
    }
    // the generator's state after exactly the consumed words (the unread ones all belong to the last block)
    std::memcpy(key, st, 624 * sizeof(uint32_t));
    *pos_io = 624 - (have - rd);
    // phase 2: the transform of the accepted pairs
    float* o = out + i0;
    double last_x1f = 0.0;
    auto part = [&](int64_t lo, int64_t hi) {
        for (int64_t j = lo; j < hi; ++j) {
            const double r2 = R2[(size_t)j];
            const double f = std::sqrt(-2.0 * std::log(r2) / r2);
            const double a = f * X2[(size_t)j], b = f * X1[(size_t)j];
            o[2 * j] = (float)(loc + scale * a);
            if (2 * j + 1 < m) o[2 * j + 1] = (float)(loc + scale * b);
            else last_x1f = b;                             // (only the last pair, one writer)
        }
    };
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 16) n_threads = 16;
    if (n_threads > 1 && P >= 512) {
        // grow-only: two engines asking for different thread counts share the larger pool instead of rebuilding it on
        // every call (the partition of the pairs does not change a single output bit)
        if (!g_np_pool || g_np_pool->size() + 1 < n_threads) { delete g_np_pool; g_np_pool = new Pool(n_threads - 1); }
        g_np_pool->run([&](int k, int np) { part(P * k / np, P * (k + 1) / np); });
    } else {
        part(0, P);
    }
    if (m & 1) { *has_gauss_io = 1; *gauss_io = last_x1f; }
    return 0;
}
