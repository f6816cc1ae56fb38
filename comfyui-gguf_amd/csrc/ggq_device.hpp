// ggq_device.hpp -- gfx950 (MI355X / CDNA4) device code for GGUF block dequantization.
//
// What it computes: the reference's block unpackers (city96/ComfyUI-GGUF dequant.py:65-285),
// bit-exactly, in all three arithmetic modes its nodes can select (dequant_dtype None / float16,
// bfloat16, float32; nodes.py:152-153,186) and with the trailing `.to(dtype)` of dequantize_tensor
// (dequant.py:23) fused into the store: every `*`, `+`, `-` of the reference is ONE hardware op with
// ONE rounding here too (v_pk_mul_f16 / v_pk_add_f16; v_mul_f32 / v_sub_f32; the same + v_cvt_pk_bf16_f32).
// The build uses -ffp-contract=off so that d*q - dm never becomes an FMA -- a fused op differs from the
// reference by up to 1328 ULP on Q4_K, SURVEY.md section 0 finding 3.
//
// Shape of the work (HBM-bound byte unpack, no MFMA -- there is no contraction here):
//   * unit of work = a GROUP of G consecutive blocks, owned by a TEAM: one wavefront (default), or the
//     4 wavefronts of a workgroup together (COOP; most formats in the fp16 arithmetic mode -- the host picks the team
//     shape per (format, arithmetic, output dtype, launch size), ggq_capi.hip);
//   * the team copies the group's packed bytes HBM -> VGPR -> its LDS slice with coalesced
//     16 B/lane loads (a group starts 16-B aligned because 8*type_size % 16 == 0 for every ggml
//     block format);
//   * every lane then produces CHUNKS of 8 consecutive output elements: a per-format DECODE picks
//     the quant bytes / scale fields it needs out of LDS (broadcast reads, natural alignment only)
//     and widens sub-byte fields with v_perm_b32; a generic ARITHMETIC stage applies the reference's
//     rounding sequence; an OUTPUT stage converts and issues ONE 16-byte store;
//   * lane l of store s writes unit TEAM*s + l, so each store instruction of a wave covers
//     1 KiB of contiguous output (full 128-B lines, no partial-line writes) -- for fp32 output a
//     lane owns 4 elements instead of 8 to keep it so;
//   * teams never synchronise with each other (no atomics, no cross-workgroup state); inside a solo
//     team ordering is in-order LDS issue + a compiler fence, inside a COOP team one s_barrier;
//   * which workgroup takes which group is XCD-aware on large launches (Engine::run).
//
// No CUDA compatibility layer, no dual paths: this file is gfx950 code.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ggq {

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define GGQ_DEV __device__ __forceinline__

GGQ_DEV h2 as_h2(uint32_t u) { return __builtin_bit_cast(h2, u); }
GGQ_DEV uint32_t as_u32(h2 h) { return __builtin_bit_cast(uint32_t, h); }
GGQ_DEV h2 splat(_Float16 x) { return h2{x, x}; }
GGQ_DEV h2 splatf(float x) { return h2{(_Float16)x, (_Float16)x}; }
GGQ_DEV h2 bcast_lo(h2 v) { return __builtin_shufflevector(v, v, 0, 0); }
GGQ_DEV h2 bcast_hi(h2 v) { return __builtin_shufflevector(v, v, 1, 1); }

// fp16 1024.0 in both halves: (MAGIC | q) read as fp16 is exactly 1024 + q for 0 <= q < 1024,
// and (1024 + q) - (1024 + bias) is an exact fp16 subtraction -> int -> fp16 without v_cvt.
constexpr uint32_t MAGIC = 0x64006400u;

// bytes (b0,b1) of w -> 16-bit lanes [0x64:b0, 0x64:b1]; bytes (b2,b3) -> [0x64:b2, 0x64:b3]: the SAME v_perm_b32 that spreads the bytes also supplies the
// 0x64 exponent byte of the magic (its other source operand holds 0x64 in every byte; selector 4 = that operand's byte 0), so `spread | MAGIC` is one
// instruction, not two (round 5: 4 VALU fewer per chunk in every fp16-arithmetic kernel -- 46 -> 42 per 8 weights in ggq_linear_small, which is VALU-bound).
GGQ_DEV uint32_t spread_lo_magic(uint32_t w) { return __builtin_amdgcn_perm(0x64646464u, w, 0x04010400u); }
GGQ_DEV uint32_t spread_hi_magic(uint32_t w) { return __builtin_amdgcn_perm(0x64646464u, w, 0x04030402u); }

// two small unsigned ints (bits 0.. and 16..) -> h2 of (q - bias), exact
GGQ_DEV h2 ints_h2(uint32_t pair, float bias) { return as_h2(pair | MAGIC) - splatf(1024.0f + bias); }

// ---- LDS reads at the natural alignment the block layout guarantees (a misaligned ds_read_b64
// replays at 64 cycles, cdna_hip_programming.md Guideline 17, so 2-B-aligned formats read u16s)
GGQ_DEV uint32_t lds_u8(const uint8_t* p) { return *p; }
GGQ_DEV uint32_t lds_u16(const uint8_t* p) { return *reinterpret_cast<const uint16_t*>(p); }
GGQ_DEV _Float16 lds_h(const uint8_t* p) { return __builtin_bit_cast(_Float16, *reinterpret_cast<const uint16_t*>(p)); }

// A block whose size is 2 (mod 4) puts every other block at a 2-byte-aligned address.  hipcc would
// merge four u16 reads there into ONE misaligned ds_read_b64, which the LDS replays (measured:
// SQ_LDS_UNALIGNED_STALL 3.7e8-7.3e8 quad-cycles per launch on Q4_0 / Q8_0 / Q6_K / Q5_0, 0 on the
// aligned formats).  So, when the bytes are in LDS: read the enclosing ALIGNED dwords -- volatile
// on an explicit LDS-address-space pointer, so that they stay separate ds_read_b32 -- and
// funnel-shift them into place with v_alignbyte_b32.  (LDS = false: the same code pointed at
// global memory, where a misaligned load costs nothing; used only by the harness's no-LDS engine,
// tests/microbench/ggq_lab_engine.hpp.)
typedef const volatile __attribute__((address_space(3))) uint32_t* lds_dword_ptr;
GGQ_DEV uint32_t lds_dword(const uint8_t* p4) { return *(lds_dword_ptr)(p4); }

template <int ALIGN, bool LDS>
GGQ_DEV uint32_t lds_ld4(const uint8_t* p)
{
    if constexpr (ALIGN >= 4) {
        return *reinterpret_cast<const uint32_t*>(p);
    } else if constexpr (!LDS) {
        return lds_u16(p) | (lds_u16(p + 2) << 16);
    } else {
        const uint32_t sh = (uint32_t)reinterpret_cast<uintptr_t>(p) & 2u;
        const uint8_t* q = p - sh;
        const uint32_t d0 = lds_dword(q), d1 = lds_dword(q + 4);
        return __builtin_amdgcn_alignbyte(d1, d0, sh);
    }
}

template <int ALIGN, bool LDS>
GGQ_DEV u32x2 lds_ld8(const uint8_t* p)
{
    if constexpr (ALIGN >= 8) {
        return *reinterpret_cast<const u32x2*>(p);
    } else if constexpr (ALIGN >= 4 || !LDS) {
        return u32x2{lds_ld4<ALIGN, LDS>(p), lds_ld4<ALIGN, LDS>(p + 4)};
    } else {
        const uint32_t sh = (uint32_t)reinterpret_cast<uintptr_t>(p) & 2u;
        const uint8_t* q = p - sh;
        const uint32_t d0 = lds_dword(q), d1 = lds_dword(q + 4), d2 = lds_dword(q + 8);
        return u32x2{__builtin_amdgcn_alignbyte(d1, d0, sh), __builtin_amdgcn_alignbyte(d2, d1, sh)};
    }
}

// 4 byte-fields t (one per byte) -> two h2 of (field - bias), exact
struct H2x2 { h2 a, b; };
GGQ_DEV H2x2 fields_h2(uint32_t t, float bias)
{
    const h2 off = splatf(1024.0f + bias);
#ifdef GGQ_PERM_THEN_OR     /* A/B builds only: rounds 1-4 (zero-extend with v_perm_b32, then v_or_b32 the magic in) */
    return H2x2{as_h2(__builtin_amdgcn_perm(0u, t, 0x0C010C00u) | MAGIC) - off, as_h2(__builtin_amdgcn_perm(0u, t, 0x0C030C02u) | MAGIC) - off};
#else
    return H2x2{as_h2(spread_lo_magic(t)) - off, as_h2(spread_hi_magic(t)) - off};
#endif
}

// 4 bits of x (bit k -> bit 0 of byte k)
GGQ_DEV uint32_t bits4_to_bytes(uint32_t x) { return ((x & 15u) * 0x00204081u) & 0x01010101u; }

// ============================================================================ block formats
// Every ggml block format on this path has the same algebraic shape (dequant.py:65-285):
//
//     out[e] = (A * (field[e] - BIAS))  [+ m | - B]
//
// with A = d or rn(d * sc), B = rn(dmin * mn), `field` an unsigned byte-sized integer and d, m,
// dmin fp16 values stored in the block.  A format therefore only has to DECODE: pick the 8 byte
// fields of one chunk and the scale operands out of the packed bytes (pure integer work, shared by
// all arithmetic modes); the arithmetic itself -- the reference's op sequence with its roundings,
// in fp16 (default), bf16 or fp32 (the Advanced loader's dequant_dtype, nodes.py:186) -- is applied
// by finish_*() below.
//
//   KIND   K_D     rn(d * (q - BIAS))                          Q8_0 Q4_0 Q5_0 IQ4_NL
//          K_DM    rn(rn(d * q) + m)                           Q4_1 Q5_1
//          K_SCMN  rn(rn(rn(d * sc) * q) - rn(dmin * mn))      Q2_K Q4_K Q5_K
//          K_SC    rn(rn(d * sc) * (q - BIAS))                 Q3_K Q6_K IQ4_XS
enum : int { K_D = 0, K_DM = 1, K_SCMN = 2, K_SC = 3 };

// What one chunk (8 consecutive output elements [8j, 8j+8) of the block at `b`) decodes to.
struct Fields {
    uint32_t t0, t1;   // the 8 unsigned fields, one per byte: elements 0-3 in t0, 4-7 in t1
    uint32_t dm;       // fp16 bits: low half = d, high half = m / dmin (K_DM, K_SCMN)
    int32_t sc, mn;    // integer sub-block scale (K_SCMN, K_SC; signed for K_SC) and min (K_SCMN)
};

// dequant.py:65-69    [d f16][qs i8 x32]            out = rn(d * qs)
struct FmtQ8_0 {
    static constexpr int ID = 8, BS = 32, TS = 34, LDS_ALIGN = 2, KIND = K_D, BIAS = 128;
    template <bool LDS>
    GGQ_DEV static Fields fields(const uint8_t* b, int j)
    {
        const u32x2 w = lds_ld8<2, LDS>(b + 2 + 8 * j);
        // int8 x -> (x ^ 0x80) = x + 128 unsigned
        return Fields{w.x ^ 0x80808080u, w.y ^ 0x80808080u, lds_u16(b), 0, 0};
    }
};

// the 8 nibbles feeding chunk j of a 32-element legacy block whose 16 quant bytes start at qs:
// j = 0,1 -> low nibbles of qs[8j..], j = 2,3 -> high nibbles of qs[8(j-2)..]    (dequant.py:121-122)
template <int ALIGN, bool LDS>
GGQ_DEV u32x2 legacy_nibbles(const uint8_t* qs, int j)
{
    const u32x2 w = lds_ld8<ALIGN, LDS>(qs + 8 * (j & 1));
    const int sh = (j >> 1) * 4;
    return u32x2{(w.x >> sh) & 0x0F0F0F0Fu, (w.y >> sh) & 0x0F0F0F0Fu};
}

// dequant.py:115-123  [d][qs u8 x16]                out = rn(d * (q - 8))
struct FmtQ4_0 {
    static constexpr int ID = 2, BS = 32, TS = 18, LDS_ALIGN = 2, KIND = K_D, BIAS = 8;
    template <bool LDS>
    GGQ_DEV static Fields fields(const uint8_t* b, int j)
    {
        const u32x2 t = legacy_nibbles<2, LDS>(b + 2, j);
        return Fields{t.x, t.y, lds_u16(b), 0, 0};
    }
};

// dequant.py:103-113  [d][m][qs x16]                out = rn(rn(d * q) + m)
struct FmtQ4_1 {
    static constexpr int ID = 3, BS = 32, TS = 20, LDS_ALIGN = 4, KIND = K_DM, BIAS = 0;
    template <bool LDS>
    GGQ_DEV static Fields fields(const uint8_t* b, int j)
    {
        const u32x2 t = legacy_nibbles<4, LDS>(b + 4, j);
        return Fields{t.x, t.y, lds_ld4<4, LDS>(b), 0, 0};
    }
};

// dequant.py:87-101   [d][qh u32][qs x16]           q = nib | bit(qh, e) << 4;  out = rn(d * (q - 16))
struct FmtQ5_0 {
    static constexpr int ID = 6, BS = 32, TS = 22, LDS_ALIGN = 2, KIND = K_D, BIAS = 16;
    template <bool LDS>
    GGQ_DEV static Fields fields(const uint8_t* b, int j)
    {
        const uint32_t qh = lds_ld4<2, LDS>(b + 2) >> (8 * j);
        const u32x2 t = legacy_nibbles<2, LDS>(b + 6, j);
        return Fields{t.x | (bits4_to_bytes(qh) << 4), t.y | (bits4_to_bytes(qh >> 4) << 4), lds_u16(b), 0, 0};
    }
};

// dequant.py:71-85    [d][m][qh u32][qs x16]        out = rn(rn(d * q) + m)
struct FmtQ5_1 {
    static constexpr int ID = 7, BS = 32, TS = 24, LDS_ALIGN = 8, KIND = K_DM, BIAS = 0;
    template <bool LDS>
    GGQ_DEV static Fields fields(const uint8_t* b, int j)
    {
        const uint32_t qh = lds_ld4<4, LDS>(b + 4) >> (8 * j);
        const u32x2 t = legacy_nibbles<8, LDS>(b + 8, j);
        return Fields{t.x | (bits4_to_bytes(qh) << 4), t.y | (bits4_to_bytes(qh >> 4) << 4), lds_ld4<4, LDS>(b), 0, 0};
    }
};

// dequant.py:241 KVALUES as bytes; nibble -> int8 through two v_perm_b32 and a v_bfi_b32
GGQ_DEV uint32_t kvalues4(uint32_t t /* 4 nibble-bytes */)
{
    // { -127,-104,-83,-65, -49,-35,-22,-10,  1,13,25,38, 53,69,89,113 }
    constexpr uint32_t K0 = 0xBFAD9881u, K1 = 0xF6EADDCFu, K2 = 0x26190D01u, K3 = 0x71594535u;
    const uint32_t sel = t & 0x07070707u;
    const uint32_t lo = __builtin_amdgcn_perm(K1, K0, sel), hi = __builtin_amdgcn_perm(K3, K2, sel);
    const uint32_t m = ((t >> 3) & 0x01010101u) * 0xFFu;
    return (hi & m) | (lo & ~m);
}

// dequant.py:243-256  [d][qs x16]                   out = rn(d * KVALUES[q])
struct FmtIQ4_NL {
    static constexpr int ID = 20, BS = 32, TS = 18, LDS_ALIGN = 2, KIND = K_D, BIAS = 128;
    template <bool LDS>
    GGQ_DEV static Fields fields(const uint8_t* b, int j)
    {
        const u32x2 t = legacy_nibbles<2, LDS>(b + 2, j);
        return Fields{kvalues4(t.x) ^ 0x80808080u, kvalues4(t.y) ^ 0x80808080u, lds_u16(b), 0, 0};
    }
};

// ---- K-quants: 256-element super-blocks, 32 chunks each; chunk j covers elements 8j..8j+7.

// dequant.py:129-139 get_scale_min.  hdr = the first 16 bytes of a Q4_K / Q5_K super-block
// ([d][dmin][scales 12], one broadcast ds_read_b128); 6-bit (sc, mn) of sub-block sb, branch-free.
GGQ_DEV void k_scale_min(u32x4 hdr, int sb, int32_t& sc, int32_t& mn)
{
    const uint32_t sh = 8u * (uint32_t)(sb & 3);
    const uint32_t a = (hdr.y >> sh) & 0xFFu, bb = (hdr.z >> sh) & 0xFFu, c = (hdr.w >> sh) & 0xFFu;
    const uint32_t hi = 0u - (uint32_t)(sb >> 2);                    // all-ones for sub-blocks 4..7
    sc = (int32_t)(((a & 63u) & ~hi) | (((c & 15u) | ((a >> 2) & 0x30u)) & hi));
    mn = (int32_t)(((bb & 63u) & ~hi) | (((c >> 4) | ((bb >> 2) & 0x30u)) & hi));
}

// dequant.py:180-195  [d][dmin][scales 12][qs 128]  out = rn(rn(rn(d*sc) * q) - rn(dmin*mn))
struct FmtQ4_K {
    static constexpr int ID = 12, BS = 256, TS = 144, LDS_ALIGN = 16, KIND = K_SCMN, BIAS = 0;
    template <bool LDS>
    GGQ_DEV static Fields fields(const uint8_t* b, int j)
    {
        const int sb = j >> 2;
        const u32x4 hdr = *reinterpret_cast<const u32x4*>(b);
        Fields f;
        k_scale_min(hdr, sb, f.sc, f.mn);
        const u32x2 w = lds_ld8<8, LDS>(b + 16 + 32 * (sb >> 1) + 8 * (j & 3));
        const int sh = (sb & 1) * 4;
        f.t0 = (w.x >> sh) & 0x0F0F0F0Fu;
        f.t1 = (w.y >> sh) & 0x0F0F0F0Fu;
        f.dm = hdr.x;
        return f;
    }
};

// dequant.py:159-178  [d][dmin][scales 12][qh 32][qs 128]   q = nib | bit(qh[l], sb) << 4
struct FmtQ5_K {
    static constexpr int ID = 13, BS = 256, TS = 176, LDS_ALIGN = 16, KIND = K_SCMN, BIAS = 0;
    template <bool LDS>
    GGQ_DEV static Fields fields(const uint8_t* b, int j)
    {
        const int sb = j >> 2;
        const u32x4 hdr = *reinterpret_cast<const u32x4*>(b);
        Fields f;
        k_scale_min(hdr, sb, f.sc, f.mn);
        const u32x2 h = lds_ld8<8, LDS>(b + 16 + 8 * (j & 3));
        const u32x2 w = lds_ld8<8, LDS>(b + 48 + 32 * (sb >> 1) + 8 * (j & 3));
        const int sh = (sb & 1) * 4;
        f.t0 = ((w.x >> sh) & 0x0F0F0F0Fu) | (((h.x >> sb) & 0x01010101u) << 4);
        f.t1 = ((w.y >> sh) & 0x0F0F0F0Fu) | (((h.y >> sb) & 0x01010101u) << 4);
        f.dm = hdr.x;
        return f;
    }
};

// dequant.py:141-157  [ql 128][qh 64][scales i8 x16][d]     out = rn(rn(d*scale) * (q - 32))
struct FmtQ6_K {
    static constexpr int ID = 14, BS = 256, TS = 210, LDS_ALIGN = 2, KIND = K_SC, BIAS = 32;
    template <bool LDS>
    GGQ_DEV static Fields fields(const uint8_t* b, int j)
    {
        const int half = j >> 4, k = (j >> 2) & 3, c4 = j & 3;
        Fields f;
        f.sc = (int32_t)(int8_t)lds_u8(b + 192 + (j >> 1));
        f.mn = 0;
        f.dm = lds_u16(b + 208);
        const u32x2 w = lds_ld8<2, LDS>(b + 64 * half + 32 * (k & 1) + 8 * c4);
        const u32x2 h = lds_ld8<2, LDS>(b + 128 + 32 * half + 8 * c4);
        const int sh = (k >> 1) * 4;
        f.t0 = ((w.x >> sh) & 0x0F0F0F0Fu) | (((h.x >> (2 * k)) & 0x03030303u) << 4);
        f.t1 = ((w.y >> sh) & 0x0F0F0F0Fu) | (((h.y >> (2 * k)) & 0x03030303u) << 4);
        return f;
    }
};

// dequant.py:221-238  [scales 16][qs 64][d][dmin]   out = rn(rn(rn(d*(s&15)) * q) - rn(dmin*(s>>4)))
struct FmtQ2_K {
    static constexpr int ID = 10, BS = 256, TS = 84, LDS_ALIGN = 4, KIND = K_SCMN, BIAS = 0;
    template <bool LDS>
    GGQ_DEV static Fields fields(const uint8_t* b, int j)
    {
        const int half = j >> 4, k = (j >> 2) & 3, c4 = j & 3;
        const uint32_t s = lds_u8(b + (j >> 1));
        const u32x2 w = lds_ld8<4, LDS>(b + 16 + 32 * half + 8 * c4);
        return Fields{(w.x >> (2 * k)) & 0x03030303u, (w.y >> (2 * k)) & 0x03030303u, lds_ld4<4, LDS>(b + 80),
                      (int32_t)(s & 15u), (int32_t)(s >> 4)};
    }
};

// dequant.py:197-219  [hmask 32][qs 64][scales 12][d]   out = rn(rn(d*(scale-32)) * (ql - (hb ? 0 : 4)))
struct FmtQ3_K {
    static constexpr int ID = 11, BS = 256, TS = 110, LDS_ALIGN = 2, KIND = K_SC, BIAS = 4;
    template <bool LDS>
    GGQ_DEV static Fields fields(const uint8_t* b, int j)
    {
        const int half = j >> 4, k = (j >> 2) & 3, c4 = j & 3, jj = j >> 1;
        const uint32_t lo = (lds_u8(b + 96 + (jj & 7)) >> (4 * (jj >> 3))) & 15u;
        const uint32_t hi = (lds_u8(b + 104 + (jj & 3)) >> (2 * (jj >> 2))) & 3u;
        Fields f;
        f.sc = (int32_t)(lo | (hi << 4)) - 32;
        f.mn = 0;
        f.dm = lds_u16(b + 108);
        const u32x2 w = lds_ld8<2, LDS>(b + 32 + 32 * half + 8 * c4);
        const u32x2 hm = lds_ld8<2, LDS>(b + 8 * c4);
        const int hs = j >> 2;
        // q = ql - 4*(1-hb) = (ql | hb << 2) - 4
        f.t0 = ((w.x >> (2 * k)) & 0x03030303u) | (((hm.x >> hs) & 0x01010101u) << 2);
        f.t1 = ((w.y >> (2 * k)) & 0x03030303u) | (((hm.y >> hs) & 0x01010101u) << 2);
        return f;
    }
};

// dequant.py:258-285  [d][scales_h u16][scales_l 4][qs 128]   out = rn(rn(d*(scale-32)) * KVALUES[q])
struct FmtIQ4_XS {
    static constexpr int ID = 23, BS = 256, TS = 136, LDS_ALIGN = 8, KIND = K_SC, BIAS = 128;
    template <bool LDS>
    GGQ_DEV static Fields fields(const uint8_t* b, int j)
    {
        const int g = j >> 2, c4 = j & 3;
        const uint32_t lo = (lds_u8(b + 4 + (g >> 1)) >> (4 * (g & 1))) & 15u;
        const uint32_t hi = (lds_u16(b + 2) >> (2 * g)) & 3u;
        Fields f;
        f.sc = (int32_t)(lo | (hi << 4)) - 32;
        f.mn = 0;
        f.dm = lds_u16(b);
        const u32x2 w = lds_ld8<8, LDS>(b + 8 + 16 * g + 8 * (c4 & 1));
        const int sh = (c4 >> 1) * 4;
        f.t0 = kvalues4((w.x >> sh) & 0x0F0F0F0Fu) ^ 0x80808080u;
        f.t1 = kvalues4((w.y >> sh) & 0x0F0F0F0Fu) ^ 0x80808080u;
        return f;
    }
};

// ============================================================================ arithmetic modes
// The reference's `dtype` argument of the block functions (its dequant_dtype): None / float16 is the
// stock path; float32 and bfloat16 cast d, m, dmin first and run the SAME op sequence in that dtype
// (SURVEY.md section 8a "Precision modes").  Every reference op is one correctly rounded op here.
enum : int { AR_F16 = 0, AR_BF16 = 1, AR_F32 = 2 };

GGQ_DEV _Float16 h_of(uint32_t bits) { return __builtin_bit_cast(_Float16, (uint16_t)bits); }

#define GGQ_EMIT4(EXPR_A0, EXPR_B0, EXPR_A1, EXPR_B1) \
    u32x4 { as_u32(EXPR_A0), as_u32(EXPR_B0), as_u32(EXPR_A1), as_u32(EXPR_B1) }

// All arithmetic works on QUADS: 4 consecutive elements = the 4 byte fields of one dword t.  A chunk
// is two quads (t0, t1); the scale operands are common to both (the compiler shares them).

// fp16 arithmetic: packed v_pk_mul_f16 / v_pk_add_f16, two elements per instruction; int -> fp16
// through the exact 0x6400 trick.  Result: 4 fp16 values as 2 dwords.
template <int KIND, int BIAS>
GGQ_DEV u32x2 quad_f16(const Fields& f, uint32_t t)
{
    const H2x2 q = fields_h2(t, (float)BIAS);
    if constexpr (KIND == K_D) {
        const h2 d = bcast_lo(as_h2(f.dm));
        return u32x2{as_u32(d * q.a), as_u32(d * q.b)};
    } else if constexpr (KIND == K_DM) {
        const h2 dm = as_h2(f.dm);
        const h2 d = bcast_lo(dm), m = bcast_hi(dm);
        return u32x2{as_u32(d * q.a + m), as_u32(d * q.b + m)};
    } else if constexpr (KIND == K_SCMN) {
        const h2 dlml = as_h2(f.dm) * ints_h2((uint32_t)f.sc | ((uint32_t)f.mn << 16), 0.0f);   // (d*sc, dmin*mn)
        const h2 dl = bcast_lo(dlml), ml = bcast_hi(dlml);
        return u32x2{as_u32(dl * q.a - ml), as_u32(dl * q.b - ml)};
    } else {
        const h2 dl = splat(h_of(f.dm) * (_Float16)(int16_t)f.sc);                               // v_cvt_f16_i16: exact
        return u32x2{as_u32(dl * q.a), as_u32(dl * q.b)};
    }
}

// fp32 / bf16 arithmetic share one body: values live in fp32 registers; RB = true rounds every
// result to bf16 (RNE) -- exactly what torch's bf16 kernels do (fp32 op, then round).  Integers
// (|q| <= 255) are exact in both dtypes.  Roundings go through the hardware converter two at a time:
// v_cvt_pk_bf16_f32 packs (a, b) as bf16, and a bf16 is the top half of its fp32 value.
typedef __bf16 bf16_t;
typedef bf16_t bf2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
GGQ_DEV uint32_t pack_bf16(float a, float b) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{a, b}, bf2)); }
GGQ_DEV float bits_f32(uint32_t u) { return __builtin_bit_cast(float, u); }

template <bool RB> GGQ_DEV float rnd(float x)
{
    if constexpr (RB) return (float)(bf16_t)x;
    else return x;
}
template <bool RB> GGQ_DEV void rnd2(float& a, float& b)
{
    if constexpr (RB) {
        const uint32_t p = pack_bf16(a, b);
        a = bits_f32(p << 16);
        b = bits_f32(p & 0xFFFF0000u);
    }
}

template <bool RB> GGQ_DEV f32x4 rnd4(f32x4 v)
{
    float a = v.x, b = v.y, c = v.z, d = v.w;
    rnd2<RB>(a, b);
    rnd2<RB>(c, d);
    return f32x4{a, b, c, d};
}

// Returns the results of the LAST op of the sequence, not yet rounded to bf16 when RB: the output
// stage applies that final rounding together with the `.to(dtype)` conversion (for a bf16 result the
// two coincide in one v_cvt_pk_bf16_f32).
// Two elements at a time (f2): the multiplies / adds become v_pk_mul_f32 / v_pk_add_f32 and every bf16 rounding is one
// v_cvt_pk_bf16_f32 plus a shift and a mask to re-widen the pair.
template <bool RB> GGQ_DEV f2 rnd_pair(f2 v)
{
    if constexpr (RB) {
        const uint32_t p = pack_bf16(v.x, v.y);
        return f2{bits_f32(p << 16), bits_f32(p & 0xFFFF0000u)};
    } else {
        return v;
    }
}

template <int KIND, int BIAS, bool RB>
GGQ_DEV f32x4 quad_f32(const Fields& f, uint32_t t)
{
    const f2 bias = f2{(float)BIAS, (float)BIAS};
    const f2 q01 = f2{(float)(t & 0xFFu), (float)((t >> 8) & 0xFFu)} - bias;          // v_cvt_f32_ubyteN, exact
    const f2 q23 = f2{(float)((t >> 16) & 0xFFu), (float)(t >> 24)} - bias;
    const float d = rnd<RB>((float)h_of(f.dm));
    f2 r01, r23;
    if constexpr (KIND == K_D) {
        const f2 dd{d, d};
        r01 = dd * q01; r23 = dd * q23;
    } else if constexpr (KIND == K_DM) {
        const float m = rnd<RB>((float)h_of(f.dm >> 16));
        const f2 dd{d, d}, mm{m, m};
        r01 = rnd_pair<RB>(dd * q01) + mm; r23 = rnd_pair<RB>(dd * q23) + mm;
    } else if constexpr (KIND == K_SCMN) {
        const f2 dlml = rnd_pair<RB>(f2{d, rnd<RB>((float)h_of(f.dm >> 16))} * f2{(float)f.sc, (float)f.mn});   // (d*sc, dmin*mn)
        const f2 dl{dlml.x, dlml.x}, ml{dlml.y, dlml.y};
        r01 = rnd_pair<RB>(dl * q01) - ml; r23 = rnd_pair<RB>(dl * q23) - ml;
    } else {
        const float dl1 = rnd<RB>(d * (float)f.sc);
        const f2 dl{dl1, dl1};
        r01 = dl * q01; r23 = dl * q23;
    }
    return f32x4{r01.x, r01.y, r23.x, r23.y};
}

// ============================================================================ output stage
// The reference's dequantize_tensor ends with `.to(dtype)` (dequant.py:23): one cast of the fp16
// result.  OUT = 0 keeps fp16; 1 / 2 fuse that cast (fp16 -> bf16 RNE, fp16 -> fp32 exact) into
// the store so the dense tensor is written once instead of written, re-read and re-written.
enum : int { OUT_F16 = 0, OUT_BF16 = 1, OUT_F32 = 2 };

template <int OUT> struct OutBytes { static constexpr int V = 2; };
template <> struct OutBytes<OUT_F32> { static constexpr int V = 4; };

// Device-memory pointers are typed as such (address space 1): a pointer that comes out of a
// descriptor table has no address space the compiler could infer, and every access through it
// would become a flat_* instruction -- which also occupies lgkmcnt, the counter the LDS reads wait on.
#define GGQ_GLOBAL __attribute__((address_space(1)))
typedef GGQ_GLOBAL const uint8_t* gcptr;
typedef GGQ_GLOBAL uint8_t* gptr;

// The stores that are NOT non-temporal -- the single-tensor launches, whose output the next kernel reads (ggq_capi.hip) -- carry the sc1 cache
// policy: written through to memory and dropped from the XCD's L2, yet still found in the Infinity Cache by the GEMM that reads the weight next.
// Measured against plain, sc0, sc0 sc1 and non-temporal stores on one box (profiles/r03_layer_store_cache_policy.json): standalone 3072x3072 Q4_K ->
// bf16 5.25 us (nt 5.1, plain 7.15), cost of the dequant path per emulated FLUX step 1.7-2.0 ms (plain 2.2, nt 5.0).  GGQ_PLAIN_STORE_POLICY
// (A/B builds): 0 = plain, 2 = sc1 (shipped), 3 = sc0 sc1, 6 = sc0.
//
// hipcc has no cache-policy argument on a plain global store, but it has one on the raw BUFFER store builtin, and that one the compiler MODELS
// (data-register hazards, vmcnt): rounds 2-3 issued these stores as an inline-asm `global_store_dwordx4 ... sc1` + a hand-placed `s_nop 1`,
// whose correctness rested on the exact wait-state count of one architecture.  Now: a Window = a buffer resource over the
// bytes a team is about to write (base wave-uniform, 32-bit per-lane offsets -- one VGPR of address instead of two), and
// `buffer_store_dwordx4 v[data], v_off, s[rsrc], 0 offen sc1` from `__builtin_amdgcn_raw_buffer_store_b128(..., aux)`; aux bits on gfx94x/gfx950:
// 1 = sc0, 2 = nt, 16 = sc1.  No asm statement is left in the store path.
#ifndef GGQ_PLAIN_STORE_POLICY
#define GGQ_PLAIN_STORE_POLICY 2
#endif
constexpr int STORE_AUX = GGQ_PLAIN_STORE_POLICY == 2 ? 16 : (GGQ_PLAIN_STORE_POLICY == 3 ? 17 : (GGQ_PLAIN_STORE_POLICY == 6 ? 1 : 0));

template <bool NT, class T>
GGQ_DEV void gstore(gptr p, T v)
{
    if constexpr (NT) __builtin_nontemporal_store(v, (GGQ_GLOBAL T*)p);
    else *(GGQ_GLOBAL T*)p = v;
}

// `bytes` of output starting at the WAVE-UNIFORM address `base` (stores outside [0, bytes) are dropped by the hardware's range check; the
// callers mask them anyway).  word 3 = 0x00020000: raw buffer, 32-bit data format (what every dwordx4 buffer access on gfx9 uses).
struct Window { __amdgpu_buffer_rsrc_t rsrc; };
GGQ_DEV Window window(gptr base, uint32_t bytes)
{
    return Window{__builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)base, (short)0, (int)bytes, 0x00020000)};
}
template <int AUX = STORE_AUX, class T>
GGQ_DEV void wstore(const Window& w, uint32_t byte_off, T v)
{
    static_assert(sizeof(T) == 16, "16-byte stores");
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), w.rsrc, (int)byte_off, 0, AUX);
}

template <bool NT>
GGQ_DEV u32x4 gload16(gcptr p)
{
    if constexpr (NT) return __builtin_nontemporal_load((GGQ_GLOBAL const u32x4*)p);
    else return *(GGQ_GLOBAL const u32x4*)p;
}

// The final `.to(dtype)` (dequant.py:23) is one RNE conversion, done in registers by the hardware
// converters (v_cvt_f32_f16 is exact; v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 round to nearest even).
GGQ_DEV uint32_t pack_f16(float a, float b) { return as_u32(h2{(_Float16)a, (_Float16)b}); }
GGQ_DEV uint32_t h2_to_bf16x2(uint32_t hh) { const h2 v = as_h2(hh); return pack_bf16((float)v.x, (float)v.y); }

// How a wave lays its stores out.  Every store instruction must cover 1 KiB of CONTIGUOUS output
// (64 lanes x 16 B: full 128-B lines; two half-line stores to the same lines measured 2x slower):
//   2-byte outputs: a lane owns a whole chunk (8 elements = 16 B)            PIECES = 1
//   fp32 output:    a lane owns a QUAD (4 elements = 16 B); the two lanes of a pair decode the
//                   same chunk and each keeps one half                        PIECES = 2
template <int OUT> struct Layout { static constexpr int PIECES = (OUT == OUT_F32) ? 2 : 1, ELEMS = 8 / PIECES; };

// decode -> arithmetic (ARITH) -> cast (OUT) -> one 16-byte value handed to `store` (a non-temporal global store, or a Window store with the
// write-through policy: the caller knows where); `piece` selects the quad for fp32 output
template <class F, int ARITH, int OUT, class Store>
GGQ_DEV void emit_to(const Fields& f, int piece, Store&& store)
{
    constexpr int KIND = F::KIND, BIAS = F::BIAS;
    constexpr bool RB = ARITH == AR_BF16;
    if constexpr (OUT == OUT_F32) {
        const uint32_t t = piece ? f.t1 : f.t0;
        if constexpr (ARITH == AR_F16) {
            const u32x2 v = quad_f16<KIND, BIAS>(f, t);
            const h2 a = as_h2(v.x), b = as_h2(v.y);
            store(f32x4{(float)a.x, (float)a.y, (float)b.x, (float)b.y});
        } else {
            store(rnd4<RB>(quad_f32<KIND, BIAS, RB>(f, t)));
        }
    } else if constexpr (ARITH == AR_F16) {
        const u32x2 lo = quad_f16<KIND, BIAS>(f, f.t0), hi = quad_f16<KIND, BIAS>(f, f.t1);
        if constexpr (OUT == OUT_F16) store(u32x4{lo.x, lo.y, hi.x, hi.y});
        else store(u32x4{h2_to_bf16x2(lo.x), h2_to_bf16x2(lo.y), h2_to_bf16x2(hi.x), h2_to_bf16x2(hi.y)});
    } else {
        f32x4 lo = quad_f32<KIND, BIAS, RB>(f, f.t0), hi = quad_f32<KIND, BIAS, RB>(f, f.t1);
        if constexpr (OUT == OUT_F16) {          // bf16 arithmetic, fp16 result: round to bf16 FIRST, then to fp16
            lo = rnd4<RB>(lo);
            hi = rnd4<RB>(hi);
        }
        // bf16 result: RNE of the fp32 op result == the op's own bf16 rounding (RB) or the cast of an fp32 value
        if constexpr (OUT == OUT_BF16) store(u32x4{pack_bf16(lo.x, lo.y), pack_bf16(lo.z, lo.w), pack_bf16(hi.x, hi.y), pack_bf16(hi.z, hi.w)});
        else store(u32x4{pack_f16(lo.x, lo.y), pack_f16(lo.z, lo.w), pack_f16(hi.x, hi.y), pack_f16(hi.z, hi.w)});
    }
}

// ... to element `elem` of the dense tensor at `out`, through a 64-bit address (non-temporal when NT, else a plain store: the harnesses)
template <class F, int ARITH, int OUT, bool NT>
GGQ_DEV void emit(const Fields& f, int piece, gptr out, uint64_t elem)
{
    emit_to<F, ARITH, OUT>(f, piece, [&](auto v) { gstore<NT>(out + elem * (uint64_t)OutBytes<OUT>::V, v); });
}

// fp32 output WITHOUT decoding every chunk twice.  A store instruction must cover 1 KiB of contiguous output, so for 4-byte results a lane
// can only write a QUAD (16 B) of a chunk per store; the first three rounds therefore had BOTH lanes of a pair decode the same chunk
// (LDS reads, field extraction, scale products) and keep one half each (Layout<OUT_F32>::PIECES = 2).  Here each lane of a pair decodes
// its OWN chunk -- the even lane chunk c of the wave's 64, the odd lane chunk c + 32 -- computes both of its quads, and the two swap the
// halves the other one stores (DPP quad_perm [1,0,3,2]: one v_mov_dpp per dword, or folded into the v_cndmask that selects):
//     store 1 (first KiB):  even lane <- its own quad 0,  odd lane <- the even lane's quad 1      -> chunk c, contiguous
//     store 2 (second KiB): even lane <- the odd lane's quad 0,  odd lane <- its own quad 1       -> chunk c + 32
// With fp16 arithmetic the halves cross as packed fp16 pairs (2 dwords) and are widened afterwards (v_cvt_f32_f16, exact).
// Measured (tools/mode_table.py --arith --outs f32, five builds alternated twice on one box, profiles/r05_mode_table_f32out_pairing_and_teams.json): in the
// workgroup teams the swap is worth +2...+8 % on the formats with a real decode (Q4_0, Q5_0, the K-quants, IQ4_*), level on Q4_1 / Q8_0; in the one-wave
// teams (8 store rows per wave, no partner wave to hide the exchange behind) it LOSES 3-10 % -- so: workgroup teams swap, one-wave teams keep decoding twice.
#ifndef GGQ_F32_PAIR_MODE       /* A/B builds: 0 = never (the rounds 1-4 layout), 1 = workgroup teams only (shipped), 2 = every team */
#define GGQ_F32_PAIR_MODE 1
#endif

GGQ_DEV uint32_t swap_pair(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true); }

template <class F, int ARITH>
GGQ_DEV void pair_f32(const Fields& f, bool odd, f32x4& first, f32x4& second)
{
    constexpr int KIND = F::KIND, BIAS = F::BIAS;
    constexpr bool RB = ARITH == AR_BF16;
    if constexpr (ARITH == AR_F16) {
        const u32x2 lo = quad_f16<KIND, BIAS>(f, f.t0), hi = quad_f16<KIND, BIAS>(f, f.t1);
        const u32x2 give = odd ? lo : hi;
        const u32x2 got{swap_pair(give.x), swap_pair(give.y)};
        const u32x2 a = odd ? got : lo, b = odd ? hi : got;
        const h2 a0 = as_h2(a.x), a1 = as_h2(a.y), b0 = as_h2(b.x), b1 = as_h2(b.y);
        first = f32x4{(float)a0.x, (float)a0.y, (float)a1.x, (float)a1.y};
        second = f32x4{(float)b0.x, (float)b0.y, (float)b1.x, (float)b1.y};
    } else {
        const u32x4 lo = __builtin_bit_cast(u32x4, rnd4<RB>(quad_f32<KIND, BIAS, RB>(f, f.t0)));
        const u32x4 hi = __builtin_bit_cast(u32x4, rnd4<RB>(quad_f32<KIND, BIAS, RB>(f, f.t1)));
        const u32x4 give = odd ? lo : hi;
        const u32x4 got{swap_pair(give.x), swap_pair(give.y), swap_pair(give.z), swap_pair(give.w)};
        first = __builtin_bit_cast(f32x4, odd ? got : lo);
        second = __builtin_bit_cast(f32x4, odd ? hi : got);
    }
}

// ============================================================================ the engine

// One tensor (or one contiguous run of blocks) to dequantize.  first_group = number of groups in
// all earlier descriptors of a table (exclusive prefix sum), filled by the host.
struct Desc {
    const uint8_t* packed;
    uint8_t* out;
    uint64_t n_blocks;
    uint64_t first_group;
};

struct Work {
    gcptr packed;
    gptr out;
    uint64_t n_blocks;
    uint64_t lg;   // group index inside the tensor
};

GGQ_DEV void wave_sync()
{
    // LDS traffic of one wave is issued and serviced in order; all that is needed between the
    // slice fill and the cross-lane reads is that the COMPILER keeps them in program order.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Launch shape: one team = one group, start to finish, then the waves retire and the hardware dispatcher starts the next
// workgroup -- no grid-stride loop.  (Measured on MI355X: a persistent loop with a register prefetch of the next group is
// 10-15 % SLOWER, because its s_waitcnt vmcnt(0) in front of the LDS fill also waits for the previous group's stores to be
// acknowledged -- gfx950 counts loads and stores in the one vmcnt; a fully pipelined persistent engine is 17-25 % slower still;
// profiles/r01_microbench_a/r.)  The experiment knobs those measurements needed (compile-time XCD mappings, a no-LDS engine, a store
// throttle, several groups per wave, buffer loads with explicit cache policies) live in tests/microbench/ggq_lab_engine.hpp, not here.
//   F      block format            G        blocks per group
//   OUT    output dtype            NTL/NTS  non-temporal loads / stores
//   WAVES  wavefronts per workgroup         ARITH  arithmetic mode (AR_*)
//   COOP   the workgroup's waves own ONE group together (else: one group per wave, the waves share nothing but the LDS allocation)
//   SKEW   the tensor's base pointer itself may be only 2-byte aligned (a row inside a packed table): the misalignment of every
//          group start is then taken from the ADDRESS, not from the offset inside the tensor.
template <class F, int G, int OUT, bool NTL, bool NTS, int WAVES, int ARITH = AR_F16, bool COOP = false, bool SKEW = false>
struct Engine {
    static constexpr int TS = F::TS, BS = F::BS;
    static constexpr int CPB = BS / 8;                 // chunks per block
    static constexpr int GROUP_BYTES = G * TS;
    // A group starts 16-B aligned when GROUP_BYTES % 16 == 0 (any G that is a multiple of 8);
    // otherwise its start is only 2-byte aligned and the team loads from the aligned
    // address below it, keeping the same byte offset inside its LDS slice.
    static constexpr bool ALIGNED = GROUP_BYTES % 16 == 0 && !SKEW;
    static constexpr int UNITS = (GROUP_BYTES + (ALIGNED ? 0 : 14) + 15) / 16;   // 16-B load units per group (max)
    // TEAM = the threads that own one group: a wavefront, or (COOP) the whole workgroup -- then every wave stores fewer
    // 1-KiB rows back to back, which the memory system takes 6 % faster (tests/microbench `fillrows`).
    static constexpr int TEAM = COOP ? WAVES * 64 : 64;
    static constexpr int NU = (UNITS + TEAM - 1) / TEAM;   // loads per thread
    static constexpr int CHUNKS = G * CPB;
    static constexpr int PIECES = Layout<OUT>::PIECES;  // lanes per chunk (2 for fp32 output: one quad each)
    static constexpr int NCH = CHUNKS * PIECES / TEAM; // stores per thread per group
    static constexpr bool PAIRED = OUT == OUT_F32 && (GGQ_F32_PAIR_MODE == 2 || (GGQ_F32_PAIR_MODE == 1 && COOP));   // fp32 output: one decode per chunk, halves swapped inside lane pairs (pair_f32)
    static constexpr int NIT = CHUNKS / TEAM;           // PAIRED: chunks per thread per group (two stores each)
    static constexpr int SLICE = NU * TEAM * 16;       // LDS bytes per team
    static constexpr int THREADS = WAVES * 64;
    static_assert(GROUP_BYTES % 2 == 0, "block formats are 2-byte aligned");
    static_assert(ALIGNED || (GROUP_BYTES % F::LDS_ALIGN == 0), "group start must keep the format's LDS read alignment");
    static_assert(!SKEW || F::TS % F::LDS_ALIGN == 0, "a row start is a multiple of the block size only");
    static_assert((CHUNKS * PIECES) % TEAM == 0, "a group must be a whole number of 1 KiB store rows per wave");
    static_assert(!PAIRED || CHUNKS % TEAM == 0, "fp32 output: every wave takes 64 chunks (two 1 KiB store rows) at a time");

    GGQ_DEV static void team_sync()
    {
        if constexpr (COOP) __syncthreads();             // s_barrier over the workgroup's waves
        else wave_sync();
    }

    // FULL = the whole group lies inside the tensor (wave-uniform): no per-lane bounds checks,
    // so the compiler batches the LDS reads of all NCH chunks.
    template <bool FULL>
    GGQ_DEV static void body(uint8_t* slice, const Work& w, int lane)      // lane = index inside the team
    {
        const uint64_t off = w.lg * (uint64_t)GROUP_BYTES;
        const uint32_t a = ALIGNED ? 0u : ((uint32_t)((SKEW ? (uint64_t)w.packed : 0ull) + off) & 15u);   // wave-uniform
        const gcptr base = w.packed + off - a;
        uint32_t valid = a + (uint32_t)GROUP_BYTES;
        if constexpr (!FULL) {
            const uint64_t left = w.n_blocks * (uint64_t)TS - off;           // > 0 by construction
            if (left < (uint64_t)GROUP_BYTES) valid = a + (uint32_t)left;
        }
        u32x4 pf[NU];
#pragma unroll
        for (int u = 0; u < NU; u++) {
            const uint32_t o = (uint32_t)(lane + TEAM * u) * 16u;
            if (FULL && ALIGNED && (u + 1) * TEAM <= UNITS) {
                pf[u] = gload16<NTL>(base + o);
            } else {
                // the last unit of a tensor may straddle its end: an aligned 16-B read never crosses
                // a page, so the over-read (< 16 B, inside the same aligned unit) cannot fault.
                pf[u] = (o < valid) ? gload16<NTL>(base + o) : u32x4{0, 0, 0, 0};
            }
        }
#pragma unroll
        for (int u = 0; u < NU; u++) *reinterpret_cast<u32x4*>(slice + (lane + TEAM * u) * 16) = pf[u];
        team_sync();
        const uint64_t b0 = w.lg * (uint64_t)G;
        // write-through launches: a Window over this group's slice of the output (base wave-uniform; the per-lane part of the address is 32 bits)
        constexpr uint32_t OB = (uint32_t)OutBytes<OUT>::V;
        [[maybe_unused]] const Window win = window(w.out + b0 * (uint64_t)(BS * OB), (uint32_t)(G * BS) * OB);
        if constexpr (PAIRED) {
            const int ll = lane & 63, half = ll >> 1, oddi = ll & 1;
            const bool odd = oddi != 0;
            const int wbase = lane - ll;                               // first lane of this wave inside the team
#pragma unroll
            for (int s = 0; s < NIT; s++) {
                const int c0 = TEAM * s + wbase;                       // the wave's 64 chunks of this pass: c0 .. c0 + 63 (2 KiB of fp32)
                const int chunk = c0 + half + 32 * oddi;               // the one THIS lane decodes
                const int bl = chunk / CPB, j = chunk % CPB;
                // (a lane whose own chunk lies past the tensor's end still decodes -- zero-filled LDS -- because its partner needs the swap)
                const Fields f = F::template fields<true>(slice + a + bl * TS, j);
                f32x4 v1, v2;
                pair_f32<F, ARITH>(f, odd, v1, v2);
                const int c1 = c0 + half, c2 = c1 + 32;                // the chunks this lane STORES a quad of
                const uint32_t o1 = (uint32_t)c0 * 32u + (uint32_t)ll * 16u, o2 = o1 + 1024u;   // byte offsets inside the group's output
                if (FULL || b0 + (uint64_t)(c1 / CPB) < w.n_blocks) {
                    if constexpr (NTS) gstore<true>(w.out + b0 * (uint64_t)(BS * OB) + o1, v1);
                    else wstore(win, o1, v1);
                }
                if (FULL || b0 + (uint64_t)(c2 / CPB) < w.n_blocks) {
                    if constexpr (NTS) gstore<true>(w.out + b0 * (uint64_t)(BS * OB) + o2, v2);
                    else wstore(win, o2, v2);
                }
            }
            return;
        }
#pragma unroll
        for (int s = 0; s < NCH; s++) {
            const int unit = lane + TEAM * s;
            const int chunk = unit / PIECES, piece = unit % PIECES;
            const int bl = chunk / CPB, j = chunk % CPB;
            const uint64_t gb = b0 + (uint64_t)bl;
            if (FULL || gb < w.n_blocks) {
                const Fields f = F::template fields<true>(slice + a + bl * TS, j);
                if constexpr (NTS) emit<F, ARITH, OUT, true>(f, piece, w.out, gb * (uint64_t)BS + (uint64_t)(j * 8 + piece * Layout<OUT>::ELEMS));
                else emit_to<F, ARITH, OUT>(f, piece, [&](auto v) { wstore(win, (uint32_t)(bl * BS + j * 8 + piece * Layout<OUT>::ELEMS) * OB, v); });
            }
        }
    }

    // xrun_log2 (wave-uniform kernel argument, 0 = identity): the XCD-aware workgroup -> group mapping, chosen per launch by the
    // host (it pays on large launches only; profiles/r01_microbench_l_*).
    template <class Locate>
    GGQ_DEV static void run(uint64_t total_groups, uint32_t xrun_log2, Locate locate)
    {
        // (the host may add untouched DYNAMIC LDS to a launch: it only caps how many workgroups a CU holds at once)
        __shared__ __attribute__((aligned(16))) uint8_t smem[(COOP ? 1 : WAVES) * SLICE];
        const int wave = COOP ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        const int lane = COOP ? (int)threadIdx.x : (int)(threadIdx.x & 63);
        uint32_t bid = blockIdx.x;
        // Workgroup b runs on XCD b % 8 (observed dispatch order; a speed hint only, never correctness).  Runs: inside every tile of
        // 8 << xrun_log2 consecutive workgroups, XCD x takes (1 << xrun_log2) CONSECUTIVE groups, so the 128-B line two neighbouring
        // groups share (formats whose group size is not a multiple of 128 B) is fetched into ONE L2 instead of two, while the tile as a
        // whole still streams its output together.  What matters (profiles/r01_microbench_l/m/n): every XCD keeps the SAME slot of
        // every tile (rotating the slot with the tile index loses 10 %; which XCD gets which slot is irrelevant, as is the walk
        // direction).  The last, partial tile keeps the identity mapping.
        if (xrun_log2 != 0) {
            const uint32_t tl = xrun_log2 + 3u, tile = bid >> tl, in = bid & ((1u << tl) - 1u);
            if (((tile + 1) << tl) <= gridDim.x) bid = (tile << tl) + ((in & 7u) << xrun_log2) + (in >> 3);
        }
        const uint64_t g = COOP ? (uint64_t)bid : (uint64_t)bid * WAVES + (uint64_t)wave;
        if (g >= total_groups) return;
        const Work w = locate(g);
        uint8_t* slice = smem + wave * SLICE;             // COOP: wave == 0, one slice per workgroup
        if ((w.lg + 1) * (uint64_t)G <= w.n_blocks) body<true>(slice, w, lane);
        else body<false>(slice, w, lane);
    }
};

// one tensor, descriptor by value
template <class F, int G, int OUT, bool NTL, bool NTS, int WAVES, int ARITH = AR_F16, bool COOP = false>
__global__ __launch_bounds__(WAVES * 64) void dequant_one(Desc d, uint64_t total_groups, uint32_t xrun_log2)
{
    Engine<F, G, OUT, NTL, NTS, WAVES, ARITH, COOP>::run(total_groups, xrun_log2, [&](uint64_t g) { return Work{(gcptr)d.packed, (gptr)d.out, d.n_blocks, g}; });
}

// many tensors of one format: table in device memory, sorted by first_group.  Finding the tensor of group g is a chain
// of DEPENDENT scalar loads at the head of every wave: `coarse[c]` (optional) = the entry that holds group c << coarse_shift,
// which leaves a 1-2 step forward scan instead of a log2(n)-step binary search -- a team holds its wave slots idle during
// that chain, which costs the multi-wave (COOP) teams most (tests/microbench `ablocate`).
template <class F, int G, int OUT, bool NTL, bool NTS, int WAVES, int ARITH = AR_F16, bool COOP = false>
__global__ __launch_bounds__(WAVES * 64) void dequant_many(const Desc* __restrict__ table, uint32_t n, uint64_t total_groups, uint32_t xrun_log2,
                                                           const uint32_t* __restrict__ coarse, uint32_t coarse_shift)
{
    Engine<F, G, OUT, NTL, NTS, WAVES, ARITH, COOP>::run(total_groups, xrun_log2, [&](uint64_t g) {
        uint32_t lo = 0;                                // last entry with first_group <= g
        if (coarse != nullptr) {
            lo = coarse[g >> coarse_shift];
            while (lo + 1 < n && table[lo + 1].first_group <= g) lo++;
        } else {
            uint32_t hi = n;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (table[mid].first_group <= g) lo = mid; else hi = mid;
            }
        }
        const Desc d = table[lo];
        return Work{(gcptr)d.packed, (gptr)d.out, d.n_blocks, g - d.first_group};
    });
}

// rows of ONE packed table picked by an index vector (the embedding lookup, reference ops.py:251-260): output row t = table row
// indices[t].  The same engine with a different locate -- grid.y walks the output rows (no integer division on the device: its
// expansion goes through fp32 multiply-adds, which the build's FMA guard rightly refuses), grid.x the groups of one row; the
// index is a wave-uniform scalar load; indices outside [0, n_rows) are clamped (F.embedding asserts on them).  A row starts
// wherever its blocks do (2-byte aligned at worst): the SKEW engine loads from the aligned address below it.
template <class F, int G, int OUT, bool NTL, bool NTS, int WAVES, int ARITH = AR_F16, bool COOP = false>
__global__ __launch_bounds__(WAVES * 64) void dequant_rows(const uint8_t* __restrict__ packed, const int64_t* __restrict__ indices, uint8_t* __restrict__ out,
                                                           uint64_t n_rows, uint32_t row_blocks, uint64_t groups_per_row)
{
    constexpr uint64_t OUT_BYTES = (OUT == OUT_F32) ? 4 : 2;
    const uint64_t t = blockIdx.y;
    Engine<F, G, OUT, NTL, NTS, WAVES, ARITH, COOP, true>::run(groups_per_row, 0u, [&](uint64_t g) {
        int64_t row = indices[t];
        row = row < 0 ? 0 : (row >= (int64_t)n_rows ? (int64_t)n_rows - 1 : row);
        return Work{(gcptr)packed + (uint64_t)row * row_blocks * (uint64_t)F::TS, (gptr)out + t * row_blocks * (uint64_t)F::BS * OUT_BYTES,
                    (uint64_t)row_blocks, g};
    });
}

}  // namespace ggq
