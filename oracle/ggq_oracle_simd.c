/*
 * ggq_oracle_simd.c -- a THROUGHPUT-oriented CPU leg of the oracle, for bench.py's `cpu_baseline`.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as ggq_oracle.c: only tests/, smoke() and bench.py's
 * cpu_baseline leg may load it; the product never does).
 *
 * Why it exists: ggq_oracle.c evaluates every fp16 op in soft-float (exact double, one rounding) --
 * ideal as a checker, ~10x slower than the reference's own torch-CPU path, so timing IT would
 * understate what host cores can do.  This file restates the same block functions (default fp16
 * arithmetic, dequant.py:65-285) the way a tuned CPU port would: AVX2 + F16C, 8 elements per step,
 * OpenMP over blocks.  Each reference op is still one separately rounded fp16 op: the operands are
 * widened to fp32 (exact), combined with ONE fp32 op, and rounded back with VCVTPS2PH (RNE).  That
 * double rounding is innocuous for + - * when the wide format has p' >= 2p+2 significand bits
 * (24 >= 2*11+2), i.e. the result equals the correctly rounded fp16 result -- which is also how
 * torch's CPU half kernels compute.  tests/test_oracle.py holds it bit-exact to ggq_oracle.c (and through
 * it to the reference's golden vectors) on nominal, signed, adversarial and raw inputs.
 *
 * Structure: a per-format DECODE picks the 8 byte-sized integer fields of one chunk (8 consecutive
 * output elements) and the scale operands out of the packed block; one FINISH applies the op
 * sequence.  Every format is one of four shapes:
 *     K_D     rn(d * (q - bias))                          Q8_0 Q4_0 Q5_0 IQ4_NL
 *     K_DM    rn(rn(d * q) + m)                           Q4_1 Q5_1
 *     K_SCMN  rn(rn(rn(d * sc) * q) - rn(dmin * mn))      Q2_K Q4_K Q5_K
 *     K_SC    rn(rn(d * sc) * (q - bias))                 Q3_K Q6_K IQ4_XS
 *
 * Compiled with function-level target attributes (no global -march): ggq_oracle_simd_available()
 * reports at run time whether the host has AVX2 + F16C; without them the entry point returns -2.
 */
#include <stdint.h>
#include <string.h>
#include <immintrin.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define SIMD __attribute__((target("avx2,f16c")))

enum { K_D = 0, K_DM = 1, K_SCMN = 2, K_SC = 3 };

typedef struct {
    uint64_t q;        /* 8 unsigned byte fields, element i in byte i */
    uint16_t d, m;     /* fp16 bits: d, and m / dmin */
    int sc, mn;        /* integer sub-block scale (signed for K_SC) and min */
} fields_t;

static inline uint16_t ld16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }
static inline uint32_t ld32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t ld64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }

#define LO4 0x0F0F0F0F0F0F0F0Full
#define B01 0x0101010101010101ull
#define B03 0x0303030303030303ull

/* 8 bits of x (bit k -> bit 0 of byte k), four at a time: (n * 0x00204081) puts bit b of the nibble n at bit 8b */
static inline uint64_t bits8_to_bytes(uint32_t x)
{
    const uint64_t lo = ((x & 15u) * 0x00204081u) & 0x01010101u, hi = (((x >> 4) & 15u) * 0x00204081u) & 0x01010101u;
    return lo | (hi << 32);
}

/* legacy 32-element blocks: chunk j = 0,1 low nibbles of qs[8j..], j = 2,3 high nibbles of qs[8(j-2)..]  (dequant.py:121-122) */
static inline uint64_t legacy_nibbles(const uint8_t *qs, int j) { return (ld64(qs + 8 * (j & 1)) >> (4 * (j >> 1))) & LO4; }

static const int8_t KVALUES[16] = { -127, -104, -83, -65, -49, -35, -22, -10, 1, 13, 25, 38, 53, 69, 89, 113 };  /* dequant.py:241 */
static inline uint64_t kvalues8(uint64_t nib)   /* nibble bytes -> (KVALUES[n] + 128) bytes */
{
    uint64_t r = 0;
    for (int i = 0; i < 8; i++) r |= (uint64_t)(uint8_t)(KVALUES[(nib >> (8 * i)) & 15] + 128) << (8 * i);
    return r;
}

/* dequant.py:129-139 get_scale_min, sub-block sb of the 12 scale bytes s */
static inline void k_scale_min(const uint8_t *s, int sb, int *sc, int *mn)
{
    if (sb < 4) { *sc = s[sb] & 63; *mn = s[sb + 4] & 63; }
    else { *sc = (s[sb + 4] & 15) | ((s[sb - 4] >> 6) << 4); *mn = (s[sb + 4] >> 4) | ((s[sb] >> 6) << 4); }
}

/* ---- decoders: chunk j (elements 8j..8j+7) of the block at b */
static inline fields_t dec_q8_0(const uint8_t *b, int j)    /* dequant.py:65-69; bias 128 */
{ fields_t f = { ld64(b + 2 + 8 * j) ^ 0x8080808080808080ull, ld16(b), 0, 0, 0 }; return f; }
static inline fields_t dec_q4_0(const uint8_t *b, int j)    /* dequant.py:115-123; bias 8 */
{ fields_t f = { legacy_nibbles(b + 2, j), ld16(b), 0, 0, 0 }; return f; }
static inline fields_t dec_q4_1(const uint8_t *b, int j)    /* dequant.py:103-113 */
{ fields_t f = { legacy_nibbles(b + 4, j), ld16(b), ld16(b + 2), 0, 0 }; return f; }
static inline fields_t dec_q5_0(const uint8_t *b, int j)    /* dequant.py:87-101; bias 16 */
{ fields_t f = { legacy_nibbles(b + 6, j) | (bits8_to_bytes(ld32(b + 2) >> (8 * j)) << 4), ld16(b), 0, 0, 0 }; return f; }
static inline fields_t dec_q5_1(const uint8_t *b, int j)    /* dequant.py:71-85 */
{ fields_t f = { legacy_nibbles(b + 8, j) | (bits8_to_bytes(ld32(b + 4) >> (8 * j)) << 4), ld16(b), ld16(b + 2), 0, 0 }; return f; }
static inline fields_t dec_iq4_nl(const uint8_t *b, int j)  /* dequant.py:243-256; bias 128 */
{ fields_t f = { kvalues8(legacy_nibbles(b + 2, j)), ld16(b), 0, 0, 0 }; return f; }

static inline fields_t dec_q4_k(const uint8_t *b, int j)    /* dequant.py:180-195 */
{
    const int sb = j >> 2;
    fields_t f; f.d = ld16(b); f.m = ld16(b + 2);
    k_scale_min(b + 4, sb, &f.sc, &f.mn);
    f.q = (ld64(b + 16 + 32 * (sb >> 1) + 8 * (j & 3)) >> (4 * (sb & 1))) & LO4;
    return f;
}
static inline fields_t dec_q5_k(const uint8_t *b, int j)    /* dequant.py:159-178 */
{
    const int sb = j >> 2;
    fields_t f; f.d = ld16(b); f.m = ld16(b + 2);
    k_scale_min(b + 4, sb, &f.sc, &f.mn);
    f.q = ((ld64(b + 48 + 32 * (sb >> 1) + 8 * (j & 3)) >> (4 * (sb & 1))) & LO4) | (((ld64(b + 16 + 8 * (j & 3)) >> sb) & B01) << 4);
    return f;
}
static inline fields_t dec_q6_k(const uint8_t *b, int j)    /* dequant.py:141-157; bias 32 */
{
    const int half = j >> 4, k = (j >> 2) & 3, c4 = j & 3;
    fields_t f; f.d = ld16(b + 208); f.m = 0; f.mn = 0;
    f.sc = (int8_t)b[192 + (j >> 1)];
    f.q = ((ld64(b + 64 * half + 32 * (k & 1) + 8 * c4) >> (4 * (k >> 1))) & LO4) | (((ld64(b + 128 + 32 * half + 8 * c4) >> (2 * k)) & B03) << 4);
    return f;
}
static inline fields_t dec_q2_k(const uint8_t *b, int j)    /* dequant.py:221-238 */
{
    const int half = j >> 4, k = (j >> 2) & 3, c4 = j & 3;
    const int s = b[j >> 1];
    fields_t f; f.d = ld16(b + 80); f.m = ld16(b + 82); f.sc = s & 15; f.mn = s >> 4;
    f.q = (ld64(b + 16 + 32 * half + 8 * c4) >> (2 * k)) & B03;
    return f;
}
static inline fields_t dec_q3_k(const uint8_t *b, int j)    /* dequant.py:197-219; q = (ql | hb << 2) - 4 */
{
    const int half = j >> 4, k = (j >> 2) & 3, c4 = j & 3, jj = j >> 1;
    const int lo = (b[96 + (jj & 7)] >> (4 * (jj >> 3))) & 15, hi = (b[104 + (jj & 3)] >> (2 * (jj >> 2))) & 3;
    fields_t f; f.d = ld16(b + 108); f.m = 0; f.mn = 0;
    f.sc = (lo | (hi << 4)) - 32;
    f.q = ((ld64(b + 32 + 32 * half + 8 * c4) >> (2 * k)) & B03) | (((ld64(b + 8 * c4) >> (j >> 2)) & B01) << 2);
    return f;
}
static inline fields_t dec_iq4_xs(const uint8_t *b, int j)  /* dequant.py:258-285; bias 128 */
{
    const int g = j >> 2, c4 = j & 3;
    const int lo = (b[4 + (g >> 1)] >> (4 * (g & 1))) & 15, hi = (ld16(b + 2) >> (2 * g)) & 3;
    fields_t f; f.d = ld16(b); f.m = 0; f.mn = 0;
    f.sc = (lo | (hi << 4)) - 32;
    f.q = kvalues8((ld64(b + 8 + 16 * g + 8 * (c4 & 1)) >> (4 * (c4 >> 1))) & LO4);
    return f;
}

/* ---- the op sequence, 8 elements per step */
SIMD static inline __m256 rn16(__m256 x) { return _mm256_cvtph_ps(_mm256_cvtps_ph(x, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC)); }
SIMD static inline float rn16s(float x) { return _cvtsh_ss(_cvtss_sh(x, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC)); }

SIMD static inline void finish(int kind, int bias, const fields_t *f, uint16_t *out)
{
    const __m256 q = _mm256_sub_ps(_mm256_cvtepi32_ps(_mm256_cvtepu8_epi32(_mm_cvtsi64_si128((long long)f->q))), _mm256_set1_ps((float)bias));
    const float d = _cvtsh_ss(f->d);
    __m256 r;
    if (kind == K_D) {
        r = _mm256_mul_ps(_mm256_set1_ps(d), q);
    } else if (kind == K_DM) {
        r = _mm256_add_ps(rn16(_mm256_mul_ps(_mm256_set1_ps(d), q)), _mm256_set1_ps(_cvtsh_ss(f->m)));
    } else if (kind == K_SCMN) {
        const float dl = rn16s(d * (float)f->sc), ml = rn16s(_cvtsh_ss(f->m) * (float)f->mn);
        r = _mm256_sub_ps(rn16(_mm256_mul_ps(_mm256_set1_ps(dl), q)), _mm256_set1_ps(ml));
    } else {
        r = _mm256_mul_ps(_mm256_set1_ps(rn16s(d * (float)f->sc)), q);
    }
    _mm_storeu_si128((__m128i *)out, _mm256_cvtps_ph(r, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC));
}

#define DEF_RUN(NAME, DEC, KIND, BIAS, BS, TS)                                                         \
SIMD static void run_##NAME(const uint8_t *packed, int64_t n, uint16_t *out)                            \
{                                                                                                       \
    _Pragma("omp parallel for schedule(static)")                                                        \
    for (int64_t i = 0; i < n; i++) {                                                                   \
        const uint8_t *b = packed + (uint64_t)i * TS;                                                   \
        uint16_t *o = out + (uint64_t)i * BS;                                                           \
        for (int j = 0; j < BS / 8; j++) { const fields_t f = DEC(b, j); finish(KIND, BIAS, &f, o + 8 * j); } \
    }                                                                                                   \
}

DEF_RUN(q8_0, dec_q8_0, K_D, 128, 32, 34)
DEF_RUN(q4_0, dec_q4_0, K_D, 8, 32, 18)
DEF_RUN(q4_1, dec_q4_1, K_DM, 0, 32, 20)
DEF_RUN(q5_0, dec_q5_0, K_D, 16, 32, 22)
DEF_RUN(q5_1, dec_q5_1, K_DM, 0, 32, 24)
DEF_RUN(iq4_nl, dec_iq4_nl, K_D, 128, 32, 18)
DEF_RUN(q2_k, dec_q2_k, K_SCMN, 0, 256, 84)
DEF_RUN(q3_k, dec_q3_k, K_SC, 4, 256, 110)
DEF_RUN(q4_k, dec_q4_k, K_SCMN, 0, 256, 144)
DEF_RUN(q5_k, dec_q5_k, K_SCMN, 0, 256, 176)
DEF_RUN(q6_k, dec_q6_k, K_SC, 32, 256, 210)
DEF_RUN(iq4_xs, dec_iq4_xs, K_SC, 128, 256, 136)

int ggq_oracle_simd_available(void)
{
    __builtin_cpu_init();
    return __builtin_cpu_supports("avx2") && __builtin_cpu_supports("f16c");
}

/* default path (dequant_dtype=None): fp16 arithmetic, fp16 bit patterns out.  -1: unknown qtype, -2: no AVX2/F16C */
int ggq_oracle_simd_dequant_f16(int qtype, const uint8_t *packed, uint64_t n_blocks, uint16_t *out)
{
    if (!ggq_oracle_simd_available()) return -2;
    const int64_t n = (int64_t)n_blocks;
    switch (qtype) {
    case 8:  run_q8_0(packed, n, out); break;
    case 2:  run_q4_0(packed, n, out); break;
    case 3:  run_q4_1(packed, n, out); break;
    case 6:  run_q5_0(packed, n, out); break;
    case 7:  run_q5_1(packed, n, out); break;
    case 20: run_iq4_nl(packed, n, out); break;
    case 10: run_q2_k(packed, n, out); break;
    case 11: run_q3_k(packed, n, out); break;
    case 12: run_q4_k(packed, n, out); break;
    case 13: run_q5_k(packed, n, out); break;
    case 14: run_q6_k(packed, n, out); break;
    case 23: run_iq4_xs(packed, n, out); break;
    default: return -1;
    }
    return 0;
}
