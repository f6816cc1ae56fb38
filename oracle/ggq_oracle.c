/*
 * ggq_oracle.c -- CPU restatement of the reference GGUF block unpackers.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load this
 * library, and only as the checker / the CPU number reported beside the GPU one.
 * The shipped path (comfyui-gguf_amd/) never imports, links or calls it.
 *
 * Parity status: PINNED.  oracle/make_golden.py runs the reference's own
 * dequant.py *verbatim* (imported from /root/reference under a stub `gguf`
 * module) on seeded packed blocks and commits the input/output pairs under
 * tests/golden/; tests/test_oracle.py holds this file bit-exact to every one of
 * those vectors.  (The reference has no tests or golden vectors of its own,
 * SURVEY.md section 4.)
 *
 * What is restated (file:line into the reference, city96/ComfyUI-GGUF @ 2025-12-26):
 *   framing            dequant.py:30-44   packed bytes -> (n_blocks, type_size) -> (n_blocks, block_size)
 *   Q8_0               dequant.py:65-69
 *   Q5_1 / Q5_0        dequant.py:71-85 / 87-101
 *   Q4_1 / Q4_0        dequant.py:103-113 / 115-123
 *   get_scale_min      dequant.py:129-139
 *   Q6_K/Q5_K/Q4_K     dequant.py:141-157 / 159-178 / 180-195
 *   Q3_K / Q2_K        dequant.py:197-219 / 221-238
 *   IQ4_NL / IQ4_XS    dequant.py:243-256 / 258-285   (KVALUES dequant.py:241)
 *   BF16               dequant.py:61-62   (returns fp32 bits; separate entry point)
 *
 * Arithmetic model.  With the stock node dequant_dtype is None (nodes.py:152-153), so every
 * torch `*`, `+`, `-` in the block functions is an fp16 op that rounds once (round-to-nearest-
 * even).  Here each op is evaluated exactly in double (an fp16 x fp16 product needs 22 bits,
 * an fp16 +/- fp16 sum at most 51) and rounded once to fp16 by d2h(); that is the IEEE
 * correctly-rounded result, which is also what torch's CPU half kernels produce (they go
 * through float; 24 >= 2*11+2 makes that double rounding innocuous).  Integer -> fp16
 * conversions are exact (all fields are <= 8 bits).  No op is fused: d*q - dm is a rounded
 * multiply followed by a rounded subtract, never an FMA (finding 3).
 *
 * The `dequant_dtype` = float32 / bfloat16 modes of the Advanced loader (nodes.py:186) run
 * the same op sequence in that dtype; they are restated by the _f32 / _bf16 entry points.
 */
#include <stdint.h>
#include <string.h>
#include <math.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ggml type ids (public ggml enum; the values the reference receives through gguf.GGMLQuantizationType) */
enum {
    GGQ_F32 = 0, GGQ_F16 = 1, GGQ_Q4_0 = 2, GGQ_Q4_1 = 3, GGQ_Q5_0 = 6, GGQ_Q5_1 = 7, GGQ_Q8_0 = 8,
    GGQ_Q2_K = 10, GGQ_Q3_K = 11, GGQ_Q4_K = 12, GGQ_Q5_K = 13, GGQ_Q6_K = 14,
    GGQ_IQ4_NL = 20, GGQ_IQ4_XS = 23, GGQ_BF16 = 30
};

/* ---------------------------------------------------------------- fp16 / bf16 soft arithmetic */

static inline double h2d(uint16_t h)
{
    int s = h >> 15, e = (h >> 10) & 31, m = h & 1023;
    double v;
    if (e == 0)        v = ldexp((double)m, -24);
    else if (e == 31)  v = m ? NAN : INFINITY;
    else               v = ldexp((double)(m | 1024), e - 25);
    return s ? -v : v;
}

/* double -> fp16, round-to-nearest-even, one rounding. */
static inline uint16_t d2h(double x)
{
    union { double d; uint64_t u; } v; v.d = x;
    uint16_t sign = (uint16_t)((v.u >> 48) & 0x8000u);
    uint64_t a = v.u & 0x7FFFFFFFFFFFFFFFull;
    if (a > 0x7FF0000000000000ull) return (uint16_t)(sign | 0x7E00u);   /* NaN (canonical quiet) */
    if (a == 0x7FF0000000000000ull) return (uint16_t)(sign | 0x7C00u);
    if (a == 0) return sign;
    int e = (int)(a >> 52) - 1023;
    uint64_t m = a & ((1ull << 52) - 1);
    if (e >= 16) return (uint16_t)(sign | 0x7C00u);
    if (e >= -14) {
        uint32_t h = ((uint32_t)(e + 15) << 10) | (uint32_t)(m >> 42);
        uint64_t rem = m & ((1ull << 42) - 1), half = 1ull << 41;
        if (rem > half || (rem == half && (h & 1))) h++;      /* carry may run into the exponent, up to inf */
        return (uint16_t)(sign | h);
    }
    if (e < -25) return sign;                                  /* below half the smallest subnormal */
    {
        uint64_t full = (1ull << 52) | m;                      /* value = full * 2^(e-52); want value * 2^24 */
        int s = 28 - e;                                        /* 43..53 */
        uint32_t h = (uint32_t)(full >> s);
        uint64_t rem = full & ((1ull << s) - 1), half = 1ull << (s - 1);
        if (rem > half || (rem == half && (h & 1))) h++;
        return (uint16_t)(sign | h);
    }
}

static inline uint16_t hmul(uint16_t a, uint16_t b) { return d2h(h2d(a) * h2d(b)); }
static inline uint16_t hadd(uint16_t a, uint16_t b) { return d2h(h2d(a) + h2d(b)); }
static inline uint16_t hsub(uint16_t a, uint16_t b) { return d2h(h2d(a) - h2d(b)); }
static inline uint16_t i2h(int v) { return d2h((double)v); }

static inline float bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float h2f(uint16_t h) { return (float)h2d(h); }

/* float -> bf16 round-to-nearest-even (torch's c10::BFloat16 rounding) */
static inline uint16_t f2bf(float f)
{
    uint32_t u = f2bits(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return 0x7FC0;        /* NaN -> canonical, as torch */
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
static inline float bf2f(uint16_t b) { return bits2f((uint32_t)b << 16); }

/* Arithmetic "policies": the same op sequence in fp16 / fp32 / bf16 (dequant_dtype modes).
 * T is the storage type of a value in that mode. */
#define DEF_OPS(SUF, T, FROMH, FROMI, MUL, ADD, SUB)            \
    typedef T val_##SUF;                                         \
    static inline T fromh_##SUF(uint16_t h) { return FROMH; }    \
    static inline T fromi_##SUF(int i) { return FROMI; }         \
    static inline T mul_##SUF(T a, T b) { return MUL; }          \
    static inline T add_##SUF(T a, T b) { return ADD; }          \
    static inline T sub_##SUF(T a, T b) { return SUB; }

DEF_OPS(f16, uint16_t, h, i2h(i), hmul(a, b), hadd(a, b), hsub(a, b))
/* fp32 mode: d.view(f16).to(f32) is exact; each op is one IEEE float op (volatile-free: -ffp-contract=off in the Makefile) */
DEF_OPS(f32, float, h2f(h), (float)i, a * b, a + b, a - b)
/* bf16 mode: d.view(f16).to(bf16) rounds d to bf16; each op = float op then round to bf16 (torch CPU/GPU bf16 kernels) */
DEF_OPS(bf16, uint16_t, f2bf(h2f(h)), f2bf((float)i), f2bf(bf2f(a) * bf2f(b)), f2bf(bf2f(a) + bf2f(b)), f2bf(bf2f(a) - bf2f(b)))

static inline uint16_t ld16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static inline uint32_t ld32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

/* dequant.py:129-139 -- 12 packed bytes -> eight 6-bit scales and eight 6-bit mins */
static inline void get_scale_min(const uint8_t *s, uint8_t *sc, uint8_t *mn)
{
    for (int j = 0; j < 4; j++) {
        sc[j]     = s[j] & 63;
        mn[j]     = s[j + 4] & 63;
        sc[j + 4] = (uint8_t)((s[j + 8] & 15) | ((s[j] >> 6) << 4));
        mn[j + 4] = (uint8_t)((s[j + 8] >> 4) | ((s[j + 4] >> 6) << 4));
    }
}

static const int8_t KVALUES[16] = { -127, -104, -83, -65, -49, -35, -22, -10, 1, 13, 25, 38, 53, 69, 89, 113 };  /* dequant.py:241 */

/* One block of each format, generic over the arithmetic mode. */
#define DEF_BLOCKS(SUF)                                                                                   \
static void blk_q8_0_##SUF(const uint8_t *b, val_##SUF *o)   /* dequant.py:65-69 */                       \
{                                                                                                         \
    val_##SUF d = fromh_##SUF(ld16(b));                                                                   \
    for (int e = 0; e < 32; e++) o[e] = mul_##SUF(d, fromi_##SUF((int8_t)b[2 + e]));                       \
}                                                                                                         \
static void blk_q4_0_##SUF(const uint8_t *b, val_##SUF *o)   /* dequant.py:115-123 */                     \
{                                                                                                         \
    val_##SUF d = fromh_##SUF(ld16(b));                                                                   \
    for (int e = 0; e < 32; e++) {                                                                        \
        int q = e < 16 ? (b[2 + e] & 15) : (b[2 + e - 16] >> 4);                                          \
        o[e] = mul_##SUF(d, fromi_##SUF(q - 8));                                                          \
    }                                                                                                     \
}                                                                                                         \
static void blk_q4_1_##SUF(const uint8_t *b, val_##SUF *o)   /* dequant.py:103-113 */                     \
{                                                                                                         \
    val_##SUF d = fromh_##SUF(ld16(b)), m = fromh_##SUF(ld16(b + 2));                                     \
    for (int e = 0; e < 32; e++) {                                                                        \
        int q = e < 16 ? (b[4 + e] & 15) : (b[4 + e - 16] >> 4);                                          \
        o[e] = add_##SUF(mul_##SUF(d, fromi_##SUF(q)), m);                                                \
    }                                                                                                     \
}                                                                                                         \
static void blk_q5_0_##SUF(const uint8_t *b, val_##SUF *o)   /* dequant.py:87-101 */                      \
{                                                                                                         \
    val_##SUF d = fromh_##SUF(ld16(b));                                                                   \
    uint32_t qh = ld32(b + 2);                                                                            \
    for (int e = 0; e < 32; e++) {                                                                        \
        int q = e < 16 ? (b[6 + e] & 15) : (b[6 + e - 16] >> 4);                                          \
        q |= (int)((qh >> e) & 1) << 4;                                                                   \
        o[e] = mul_##SUF(d, fromi_##SUF(q - 16));                                                         \
    }                                                                                                     \
}                                                                                                         \
static void blk_q5_1_##SUF(const uint8_t *b, val_##SUF *o)   /* dequant.py:71-85 */                       \
{                                                                                                         \
    val_##SUF d = fromh_##SUF(ld16(b)), m = fromh_##SUF(ld16(b + 2));                                     \
    uint32_t qh = ld32(b + 4);                                                                            \
    for (int e = 0; e < 32; e++) {                                                                        \
        int q = e < 16 ? (b[8 + e] & 15) : (b[8 + e - 16] >> 4);                                          \
        q |= (int)((qh >> e) & 1) << 4;                                                                   \
        o[e] = add_##SUF(mul_##SUF(d, fromi_##SUF(q)), m);                                                \
    }                                                                                                     \
}                                                                                                         \
static void blk_q2_k_##SUF(const uint8_t *b, val_##SUF *o)   /* dequant.py:221-238 */                     \
{                                                                                                         \
    const uint8_t *scales = b, *qs = b + 16;                                                              \
    val_##SUF d = fromh_##SUF(ld16(b + 80)), dmin = fromh_##SUF(ld16(b + 82));                            \
    for (int e = 0; e < 256; e++) {                                                                       \
        int s = scales[e / 16];                                                                           \
        val_##SUF dl = mul_##SUF(d, fromi_##SUF(s & 15));                                                 \
        val_##SUF ml = mul_##SUF(dmin, fromi_##SUF(s >> 4));                                              \
        int q = (qs[32 * (e / 128) + e % 32] >> (2 * ((e % 128) / 32))) & 3;                              \
        o[e] = sub_##SUF(mul_##SUF(dl, fromi_##SUF(q)), ml);                                              \
    }                                                                                                     \
}                                                                                                         \
static void blk_q3_k_##SUF(const uint8_t *b, val_##SUF *o)   /* dequant.py:197-219 */                     \
{                                                                                                         \
    const uint8_t *hmask = b, *qs = b + 32, *sc = b + 96;                                                 \
    val_##SUF d = fromh_##SUF(ld16(b + 108));                                                             \
    for (int e = 0; e < 256; e++) {                                                                       \
        int j = e / 16;                                                                                   \
        int lo = j < 8 ? (sc[j] & 15) : (sc[j - 8] >> 4);                                                 \
        int hi = (sc[8 + j % 4] >> (2 * (j / 4))) & 3;                                                    \
        int scale = (int)(int8_t)(uint8_t)(lo | (hi << 4)) - 32;                                          \
        val_##SUF dl = mul_##SUF(d, fromi_##SUF(scale));                                                  \
        int ql = (qs[32 * (e / 128) + e % 32] >> (2 * ((e % 128) / 32))) & 3;                             \
        int hb = (hmask[e % 32] >> (e / 32)) & 1;                                                         \
        int q = ql - ((hb ^ 1) << 2);                                                                     \
        o[e] = mul_##SUF(dl, fromi_##SUF(q));                                                             \
    }                                                                                                     \
}                                                                                                         \
static void blk_q4_k_##SUF(const uint8_t *b, val_##SUF *o)   /* dequant.py:180-195 */                     \
{                                                                                                         \
    val_##SUF d = fromh_##SUF(ld16(b)), dmin = fromh_##SUF(ld16(b + 2));                                  \
    uint8_t sc[8], mn[8]; get_scale_min(b + 4, sc, mn);                                                   \
    const uint8_t *qs = b + 16;                                                                           \
    for (int e = 0; e < 256; e++) {                                                                       \
        int sb = e / 32;                                                                                  \
        val_##SUF dl = mul_##SUF(d, fromi_##SUF(sc[sb]));                                                 \
        val_##SUF ml = mul_##SUF(dmin, fromi_##SUF(mn[sb]));                                              \
        int byte = qs[32 * (sb / 2) + e % 32];                                                            \
        int q = (sb & 1) ? (byte >> 4) : (byte & 15);                                                     \
        o[e] = sub_##SUF(mul_##SUF(dl, fromi_##SUF(q)), ml);                                              \
    }                                                                                                     \
}                                                                                                         \
static void blk_q5_k_##SUF(const uint8_t *b, val_##SUF *o)   /* dequant.py:159-178 */                     \
{                                                                                                         \
    val_##SUF d = fromh_##SUF(ld16(b)), dmin = fromh_##SUF(ld16(b + 2));                                  \
    uint8_t sc[8], mn[8]; get_scale_min(b + 4, sc, mn);                                                   \
    const uint8_t *qh = b + 16, *qs = b + 48;                                                             \
    for (int e = 0; e < 256; e++) {                                                                       \
        int sb = e / 32;                                                                                  \
        val_##SUF dl = mul_##SUF(d, fromi_##SUF(sc[sb]));                                                 \
        val_##SUF ml = mul_##SUF(dmin, fromi_##SUF(mn[sb]));                                              \
        int byte = qs[32 * (sb / 2) + e % 32];                                                            \
        int q = (sb & 1) ? (byte >> 4) : (byte & 15);                                                     \
        q |= ((qh[e % 32] >> sb) & 1) << 4;                                                               \
        o[e] = sub_##SUF(mul_##SUF(dl, fromi_##SUF(q)), ml);                                              \
    }                                                                                                     \
}                                                                                                         \
static void blk_q6_k_##SUF(const uint8_t *b, val_##SUF *o)   /* dequant.py:141-157 */                     \
{                                                                                                         \
    const uint8_t *ql = b, *qh = b + 128; const int8_t *scales = (const int8_t *)(b + 192);               \
    val_##SUF d = fromh_##SUF(ld16(b + 208));                                                             \
    for (int e = 0; e < 256; e++) {                                                                       \
        int half = e / 128, k = (e % 128) / 32, l = e % 32;                                               \
        val_##SUF dl = mul_##SUF(d, fromi_##SUF(scales[e / 16]));                                         \
        int lo = k < 2 ? (ql[64 * half + 32 * k + l] & 15) : (ql[64 * half + 32 * (k - 2) + l] >> 4);     \
        int hi = (qh[32 * half + l] >> (2 * k)) & 3;                                                      \
        int q = (int)(int8_t)(uint8_t)(lo | (hi << 4)) - 32;                                              \
        o[e] = mul_##SUF(dl, fromi_##SUF(q));                                                             \
    }                                                                                                     \
}                                                                                                         \
static void blk_iq4_nl_##SUF(const uint8_t *b, val_##SUF *o) /* dequant.py:243-256 */                     \
{                                                                                                         \
    val_##SUF d = fromh_##SUF(ld16(b));                                                                   \
    for (int e = 0; e < 32; e++) {                                                                        \
        int q = e < 16 ? (b[2 + e] & 15) : (b[2 + e - 16] >> 4);                                          \
        o[e] = mul_##SUF(d, fromi_##SUF(KVALUES[q]));                                                     \
    }                                                                                                     \
}                                                                                                         \
static void blk_iq4_xs_##SUF(const uint8_t *b, val_##SUF *o) /* dequant.py:258-285 */                     \
{                                                                                                         \
    val_##SUF d = fromh_##SUF(ld16(b));                                                                   \
    uint32_t scales_h = ld16(b + 2);                                                                      \
    const uint8_t *scales_l = b + 4, *qs = b + 8;                                                         \
    for (int e = 0; e < 256; e++) {                                                                       \
        int g = e / 32, l = e % 32;                                                                       \
        int lo = (scales_l[g / 2] >> (4 * (g & 1))) & 15;                                                 \
        int hi = (int)(((scales_h >> (2 * g)) & 0xFF) & 3);                                               \
        int scale = (int)(int8_t)(uint8_t)(lo | (hi << 4)) - 32;                                          \
        val_##SUF dl = mul_##SUF(d, fromi_##SUF(scale));                                                  \
        int byte = qs[16 * g + l % 16];                                                                   \
        int q = l < 16 ? (byte & 15) : (byte >> 4);                                                       \
        o[e] = mul_##SUF(dl, fromi_##SUF(KVALUES[q]));                                                    \
    }                                                                                                     \
}                                                                                                         \
static int run_##SUF(int qtype, const uint8_t *packed, uint64_t n_blocks, val_##SUF *out)                 \
{                                                                                                         \
    void (*fn)(const uint8_t *, val_##SUF *) = 0; int bs = 0, ts = 0;                                     \
    switch (qtype) {                                                                                      \
    case GGQ_Q8_0:   fn = blk_q8_0_##SUF;   bs = 32;  ts = 34;  break;                                    \
    case GGQ_Q4_0:   fn = blk_q4_0_##SUF;   bs = 32;  ts = 18;  break;                                    \
    case GGQ_Q4_1:   fn = blk_q4_1_##SUF;   bs = 32;  ts = 20;  break;                                    \
    case GGQ_Q5_0:   fn = blk_q5_0_##SUF;   bs = 32;  ts = 22;  break;                                    \
    case GGQ_Q5_1:   fn = blk_q5_1_##SUF;   bs = 32;  ts = 24;  break;                                    \
    case GGQ_Q2_K:   fn = blk_q2_k_##SUF;   bs = 256; ts = 84;  break;                                    \
    case GGQ_Q3_K:   fn = blk_q3_k_##SUF;   bs = 256; ts = 110; break;                                    \
    case GGQ_Q4_K:   fn = blk_q4_k_##SUF;   bs = 256; ts = 144; break;                                    \
    case GGQ_Q5_K:   fn = blk_q5_k_##SUF;   bs = 256; ts = 176; break;                                    \
    case GGQ_Q6_K:   fn = blk_q6_k_##SUF;   bs = 256; ts = 210; break;                                    \
    case GGQ_IQ4_NL: fn = blk_iq4_nl_##SUF; bs = 32;  ts = 18;  break;                                    \
    case GGQ_IQ4_XS: fn = blk_iq4_xs_##SUF; bs = 256; ts = 136; break;                                    \
    default: return -1;                                                                                   \
    }                                                                                                     \
    int64_t n = (int64_t)n_blocks;                                                                        \
    _Pragma("omp parallel for schedule(static)")                                                          \
    for (int64_t i = 0; i < n; i++) fn(packed + (uint64_t)i * ts, out + (uint64_t)i * bs);                \
    return 0;                                                                                             \
}

DEF_BLOCKS(f16)
DEF_BLOCKS(f32)
DEF_BLOCKS(bf16)

/* ---------------------------------------------------------------- exported entry points */

int ggq_oracle_block_size(int qtype)
{
    switch (qtype) {
    case GGQ_Q8_0: case GGQ_Q4_0: case GGQ_Q4_1: case GGQ_Q5_0: case GGQ_Q5_1: case GGQ_IQ4_NL: return 32;
    case GGQ_Q2_K: case GGQ_Q3_K: case GGQ_Q4_K: case GGQ_Q5_K: case GGQ_Q6_K: case GGQ_IQ4_XS: return 256;
    case GGQ_BF16: case GGQ_F16: case GGQ_F32: return 1;
    default: return 0;
    }
}

int ggq_oracle_type_size(int qtype)
{
    switch (qtype) {
    case GGQ_Q8_0: return 34;  case GGQ_Q4_0: return 18;  case GGQ_Q4_1: return 20;  case GGQ_Q5_0: return 22;
    case GGQ_Q5_1: return 24;  case GGQ_Q2_K: return 84;  case GGQ_Q3_K: return 110; case GGQ_Q4_K: return 144;
    case GGQ_Q5_K: return 176; case GGQ_Q6_K: return 210; case GGQ_IQ4_NL: return 18; case GGQ_IQ4_XS: return 136;
    case GGQ_BF16: case GGQ_F16: return 2; case GGQ_F32: return 4;
    default: return 0;
    }
}

/* default path: dequant_dtype=None -> fp16 arithmetic, fp16 result (bit patterns) */
int ggq_oracle_dequant_f16(int qtype, const uint8_t *packed, uint64_t n_blocks, uint16_t *out)
{
    return run_f16(qtype, packed, n_blocks, out);
}

/* dequant_dtype=float32 -> fp32 arithmetic, fp32 result */
int ggq_oracle_dequant_f32(int qtype, const uint8_t *packed, uint64_t n_blocks, float *out)
{
    return run_f32(qtype, packed, n_blocks, out);
}

/* dequant_dtype=bfloat16 -> bf16 arithmetic, bf16 result (bit patterns) */
int ggq_oracle_dequant_bf16(int qtype, const uint8_t *packed, uint64_t n_blocks, uint16_t *out)
{
    return run_bf16(qtype, packed, n_blocks, out);
}

/* dequant.py:61-62 -- BF16 "blocks": int16 -> int32 << 16 -> view fp32.  Always fp32 out. */
int ggq_oracle_bf16_to_f32(const uint8_t *packed, uint64_t n, uint32_t *out_bits)
{
    for (uint64_t i = 0; i < n; i++) out_bits[i] = (uint32_t)ld16(packed + 2 * i) << 16;
    return 0;
}

/* the final `.to(dtype)` of dequantize_tensor (dequant.py:23) for an fp16 result */
void ggq_oracle_cast_f16_to_bf16(const uint16_t *in, uint64_t n, uint16_t *out)
{
    for (uint64_t i = 0; i < n; i++) out[i] = f2bf(h2f(in[i]));
}
void ggq_oracle_cast_f16_to_f32(const uint16_t *in, uint64_t n, float *out)
{
    for (uint64_t i = 0; i < n; i++) out[i] = h2f(in[i]);
}

/* ... and for an fp32 / bf16 result (dequant_dtype float32 / bfloat16): torch casts fp32 -> bf16 and
 * fp32 -> fp16 with round-to-nearest-even; bf16 -> anything goes through the exact bf16 -> fp32 widening */
void ggq_oracle_cast_f32_to_bf16(const float *in, uint64_t n, uint16_t *out)
{
    for (uint64_t i = 0; i < n; i++) out[i] = f2bf(in[i]);
}
void ggq_oracle_cast_f32_to_f16(const float *in, uint64_t n, uint16_t *out)
{
    for (uint64_t i = 0; i < n; i++) out[i] = d2h((double)in[i]);
}

/* exposed so tests can check the soft-float helpers against numpy */
uint16_t ggq_oracle_d2h(double x) { return d2h(x); }
double   ggq_oracle_h2d(uint16_t h) { return h2d(h); }
uint16_t ggq_oracle_hmul(uint16_t a, uint16_t b) { return hmul(a, b); }
uint16_t ggq_oracle_hadd(uint16_t a, uint16_t b) { return hadd(a, b); }
uint16_t ggq_oracle_hsub(uint16_t a, uint16_t b) { return hsub(a, b); }

int ggq_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void ggq_oracle_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
