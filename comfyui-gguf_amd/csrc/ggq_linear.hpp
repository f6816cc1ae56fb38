// ggq_linear.hpp -- gfx950 device code: y = x @ dequant(W)^T (+ bias) for a FEW rows of x (m <= 4), straight from the
// packed GGUF blocks, without ever writing the dense weight.
//
// Where it sits: GGMLOps.Linear.forward (reference ops.py:242-244) dequantizes the whole
// weight and then calls F.linear.  When x has one to four rows -- FLUX's modulation layers act on the conditioning vector:
// 76 linears, 27 % of the model's weights, m = batch -- that is 0.56 B/element read + 2 B written + 2 B read again for
// a matrix-vector product; fused, only the packed bytes move.  Still HBM-bound, still no MFMA: with m <= 4 there is no
// tile to feed a matrix core, the work is one pass over the packed bytes and m dot products per row.
//
// Numerics.  The WEIGHTS are exactly the reference's: decode + quad_f16 produce the same fp16 values bit for bit as the
// dequant kernels, then the `.to(dtype)` of dequantize_tensor (fp16 -> bf16 RNE / exact) is applied -- so W is the very
// tensor F.linear would have been given.  The contraction accumulates in fp32 (v_dot2c_f32_f16 / v_dot2c_f32_bf16 for 16-bit
// activations, separate fp32 multiplies and adds for fp32 ones), chunk by chunk per lane, and reduces the 64 lanes by a butterfly:
// like any GEMV, the result differs from rocBLAS's by the order of fp32 additions only.  Parity is therefore stated
// against an fp32 reference of the same op with a tolerance (tests/test_gpu_linear.py), not bit for bit -- which is why
// install(exact=True) keeps dequantize + F.linear; the default install uses this path since round 5, after tools/fused_error.py measured it no further from an fp64
// product than F.linear on every FLUX / SD3.5 / T5 layer shape (profiles/r05_fused_error.json).
//
// Shape of the work: one WAVE per output row, rows dealt round-robin to a persistent grid.  x (m x cols, 16- or 32-bit)
// is staged once per workgroup in LDS; per row the wave copies the row's packed bytes (cols/block_size * type_size, e.g.
// 1728 B for Q4_K x 3072) into its private LDS slice with 16 B/lane loads -- the NEXT row's bytes are already in flight in
// registers while the current row is decoded -- lane l takes chunks l, l+64, ... (8 weights each), and lane 0 stores y.
#pragma once

#include "ggq_device.hpp"

namespace ggq {

template <int OUT> struct XBytes { static constexpr int V = (OUT == OUT_F32) ? 4 : 2; };

typedef short s16x2 __attribute__((ext_vector_type(2)));

// sum over the 8 elements of a chunk of w[k] * x[mm][e + k], added to acc.  w = the chunk's weights as the reference's fp16
// values (4 x h2).  16-bit activations use the packed dot instructions (v_dot2c_f32_f16 / v_dot2c_f32_bf16: two products and
// the fp32 accumulate per lane-op); for bf16 the weights first take the `.to(bfloat16)` rounding of dequantize_tensor.
template <int OUT>
GGQ_DEV float dot8(const uint32_t (&w)[4], const uint8_t* xs, uint32_t cols, int mm, uint32_t e, float acc)
{
    if constexpr (OUT == OUT_F32) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(xs + ((size_t)mm * cols + e) * 4);
        const f32x4 b = *reinterpret_cast<const f32x4*>(xs + ((size_t)mm * cols + e) * 4 + 16);
        const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const h2 h = as_h2(w[k]);
            s += (float)h.x * x[2 * k];
            s += (float)h.y * x[2 * k + 1];
        }
        return acc + s;
    } else {
        const u32x4 v = *reinterpret_cast<const u32x4*>(xs + ((size_t)mm * cols + e) * 2);
        const uint32_t x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if constexpr (OUT == OUT_F16) acc = __builtin_amdgcn_fdot2(as_h2(w[k]), as_h2(x[k]), acc, false);
            else acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(s16x2, w[k]), __builtin_bit_cast(s16x2, x[k]), acc, false);
        }
        return acc;
    }
}

// the 8 weights of a chunk as what dequantize_tensor(..., dtype) would hold: fp16 pairs (OUT_F16 / OUT_F32: widened exactly
// later) or bf16 pairs (OUT_BF16: RNE from the fp16 values)
template <class F, int OUT>
GGQ_DEV void weights8(const Fields& f, uint32_t (&w)[4])
{
    const u32x2 lo = quad_f16<F::KIND, F::BIAS>(f, f.t0), hi = quad_f16<F::KIND, F::BIAS>(f, f.t1);
    w[0] = lo.x; w[1] = lo.y; w[2] = hi.x; w[3] = hi.y;
    if constexpr (OUT == OUT_BF16) {
#pragma unroll
        for (int k = 0; k < 4; k++) w[k] = h2_to_bf16x2(w[k]);
    }
}

// Formats whose sub-block of 32 elements (= 4 chunks = 4 consecutive lanes) has ONE (scale, min) pair decoded from the block header (get_scale_min,
// dequant.py:129-139): the pair's two products rn(d*sc), rn(dmin*mn) are the same for the sub-block's four chunks.
template <class F> struct SharedScale { static constexpr bool V = false; };
template <> struct SharedScale<FmtQ4_K> { static constexpr bool V = true; };
template <> struct SharedScale<FmtQ5_K> { static constexpr bool V = true; };
#ifndef GGQ_LIN_SHARED_SCALE
#define GGQ_LIN_SHARED_SCALE 1  /* Q4_K / Q5_K: lane l decodes the pair of sub-block l of a pass of 256 chunks ONCE (instead of every lane for every chunk: 22 of 62 VALU per
                                   trip) and the chunk's lane fetches it by ds_bpermute -- the very expression of quad_f16<K_SCMN>, the same bits; 0 = generic loop; A/B builds */
#endif

#ifndef GGQ_LIN_UNROLL2
#define GGQ_LIN_UNROLL2 0      /* A/B builds */
#endif
// Sum over the 64 lanes of a wave, result in LANE 63 (and nowhere else): six v_add_f32 with DPP operands -- quad_perm xor 1, xor 2, row_half_mirror, row_mirror
// (every lane of a row of 16 then holds its row's sum), row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3 -- instead of a butterfly of six
// ds_bpermute_b32 round trips through the LDS crossbar (what __shfl_xor compiles to), each a dependent ~100-cycle wait at the end of every row.
template <int CTRL, int ROWS>
GGQ_DEV float dpp_f(float v)        // lanes of rows outside ROWS read 0
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWS, 0xF, false));
}
#define GGQ_DPP_ADD(V, CTRL, ROWS) V += dpp_f<CTRL, ROWS>(V)
GGQ_DEV float wave_sum_lane63(float v)
{
    GGQ_DPP_ADD(v, 0xB1, 0xF);       // quad_perm [1,0,3,2]
    GGQ_DPP_ADD(v, 0x4E, 0xF);       // quad_perm [2,3,0,1]
    GGQ_DPP_ADD(v, 0x141, 0xF);      // row_half_mirror
    GGQ_DPP_ADD(v, 0x140, 0xF);      // row_mirror
    GGQ_DPP_ADD(v, 0x142, 0xA);      // row_bcast:15 -> rows 1, 3 (the other rows add 0)
    GGQ_DPP_ADD(v, 0x143, 0xC);      // row_bcast:31 -> rows 2, 3
    return v;
}
#undef GGQ_DPP_ADD

constexpr int LIN_NU_MAX = 6;                    // 16-byte load units per lane per row: rows of up to 6128 packed bytes
constexpr int LIN_SLICE = LIN_NU_MAX * 64 * 16;  // the LARGEST LDS slice a wave can need for one row (+ up to 15 bytes of leading misalignment)
constexpr int LIN_WAVES = 4;

// LDS bytes per wave for rows of `row_bytes` packed bytes: whole 64-lane x 16-byte load units (a row may start up to 15 bytes into its first
// unit).  Sized to the row, not to LIN_SLICE: a 3072-column Q4_K row takes 2 KiB, so eight workgroups fit a CU instead of four.
constexpr uint32_t lin_slice_bytes(uint32_t row_bytes) { return (row_bytes + 15u + 1023u) & ~1023u; }

#if GGQ_LIN_UNROLL2
#define GGQ_LIN_OCC __attribute__((amdgpu_waves_per_eu(4, 6)))     /* let the scheduler spend registers on hoisted LDS reads: 6 waves per SIMD instead of 8 */
#else
#define GGQ_LIN_OCC
#endif
template <class F, int OUT, int M>
__global__ __launch_bounds__(LIN_WAVES * 64) GGQ_LIN_OCC void linear_small(const uint8_t* __restrict__ packed_, const uint8_t* __restrict__ x_,
                                                               const uint8_t* __restrict__ bias_, uint8_t* __restrict__ y_,
                                                               uint32_t rows, uint32_t cols, uint32_t slice_bytes)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const gcptr packed = (gcptr)packed_;
    constexpr int XB = XBytes<OUT>::V;
    uint8_t* xs = lds;                                                    // m x cols values of x
    const uint32_t x_bytes = (uint32_t)M * cols * XB;
    uint8_t* slice = lds + ((x_bytes + 15u) & ~15u) + (threadIdx.x >> 6) * slice_bytes;
    for (uint32_t o = threadIdx.x * 16u; o < x_bytes; o += LIN_WAVES * 64 * 16)
        *reinterpret_cast<u32x4*>(xs + o) = *reinterpret_cast<const u32x4*>(x_ + o);
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * LIN_WAVES + (threadIdx.x >> 6));
    const uint32_t n_waves = gridDim.x * LIN_WAVES;
    const uint32_t row_bytes = cols / F::BS * F::TS;
    const uint32_t chunks = cols / 8;                                     // per row
    constexpr int CPB = F::BS / 8;

    auto fetch = [&](uint32_t r, u32x4 (&pf)[LIN_NU_MAX]) {
        const uint64_t off = (uint64_t)r * row_bytes;
        const uint32_t a = (uint32_t)off & 15u, valid = a + row_bytes;
        const gcptr base = packed + (off - a);
#pragma unroll
        for (int u = 0; u < LIN_NU_MAX; u++) {
            const uint32_t o = (uint32_t)(lane + 64 * u) * 16u;
            pf[u] = (o < valid) ? gload16<true>(base + o) : u32x4{0, 0, 0, 0};
        }
    };

    u32x4 pf[LIN_NU_MAX];
    if (wave < rows) fetch(wave, pf);
    for (uint32_t r = wave; r < rows; r += n_waves) {
        const uint32_t a = (uint32_t)((uint64_t)r * row_bytes) & 15u;
#pragma unroll
        for (int u = 0; u < LIN_NU_MAX; u++)
            if ((uint32_t)(64 * u) * 16u < a + row_bytes) *reinterpret_cast<u32x4*>(slice + (lane + 64 * u) * 16) = pf[u];
        wave_sync();
        if (r + n_waves < rows) fetch(r + n_waves, pf);                  // next row's bytes fly while this row is decoded
        float acc[M];
#pragma unroll
        for (int mm = 0; mm < M; mm++) acc[mm] = 0.0f;
        uint32_t j = lane;
        if constexpr (GGQ_LIN_SHARED_SCALE && SharedScale<F>::V) {
            const uint32_t n_sub = cols / 32u;                               // sub-blocks per row; 8 per super-block
            for (uint32_t c0 = 0; c0 < chunks; c0 += 256u) {                 // one pass = 256 chunks = 64 sub-blocks: lane l decodes the pair of sub-block c0 / 4 + l
                const uint32_t sbi = (c0 >> 2) + (uint32_t)lane, sbc = sbi < n_sub ? sbi : n_sub - 1u;
                const u32x4 hdr = *reinterpret_cast<const u32x4*>(slice + a + (sbc >> 3) * F::TS);       // [d][dmin][scales 12]
                int32_t sc, mn;
                k_scale_min(hdr, (int)(sbc & 7u), sc, mn);
                const uint32_t pre = as_u32(as_h2(hdr.x) * ints_h2((uint32_t)sc | ((uint32_t)mn << 16), 0.0f));    // (rn(d*sc), rn(dmin*mn)): quad_f16<K_SCMN>'s own expression
#pragma unroll
                for (uint32_t it = 0; it < 4u; it++) {
                    if (c0 + 64u * it >= chunks) break;                      // wave-uniform
                    // chunk jj belongs to sub-block jj / 4 = c0 / 4 + 16 it + lane / 4: its pair sits in lane 16 it + lane / 4 (every lane takes part in the exchange)
                    const h2 dlml = as_h2((uint32_t)__builtin_amdgcn_ds_bpermute((int)((16u * it + ((uint32_t)lane >> 2)) << 2), (int)pre));
                    const uint32_t jj = c0 + 64u * it + (uint32_t)lane;
                    if (jj < chunks) {
                        const h2 dl = bcast_lo(dlml), ml = bcast_hi(dlml);
                        const Fields f = F::template fields<true>(slice + a + (jj / CPB) * F::TS, (int)(jj % CPB));   // only the quants are used: its scale decode is dead code
                        const H2x2 q0 = fields_h2(f.t0, (float)F::BIAS), q1 = fields_h2(f.t1, (float)F::BIAS);
                        uint32_t w[4] = {as_u32(dl * q0.a - ml), as_u32(dl * q0.b - ml), as_u32(dl * q1.a - ml), as_u32(dl * q1.b - ml)};
                        if constexpr (OUT == OUT_BF16) {
#pragma unroll
                            for (int k = 0; k < 4; k++) w[k] = h2_to_bf16x2(w[k]);
                        }
#pragma unroll
                        for (int mm = 0; mm < M; mm++) acc[mm] = dot8<OUT>(w, xs, cols, mm, jj * 8, acc[mm]);
                    }
                }
            }
            j = chunks;                                                      // nothing left for the generic loop
        }
#if GGQ_LIN_UNROLL2
        // two chunks per trip, every LDS read of both issued before the arithmetic of either (A/B builds; EXPERIMENTS.md R4-6: mixed, not taken)
        for (; j + 64 < chunks; j += 128) {
            const Fields f0 = F::template fields<true>(slice + a + (j / CPB) * F::TS, (int)(j % CPB));
            const Fields f1 = F::template fields<true>(slice + a + ((j + 64) / CPB) * F::TS, (int)((j + 64) % CPB));
            uint32_t w0[4], w1[4];
            weights8<F, OUT>(f0, w0);
            weights8<F, OUT>(f1, w1);
#pragma unroll
            for (int mm = 0; mm < M; mm++) {
                acc[mm] = dot8<OUT>(w0, xs, cols, mm, j * 8, acc[mm]);
                acc[mm] = dot8<OUT>(w1, xs, cols, mm, (j + 64) * 8, acc[mm]);
            }
        }
#endif
        for (; j < chunks; j += 64) {
            const Fields f = F::template fields<true>(slice + a + (j / CPB) * F::TS, (int)(j % CPB));
            uint32_t w[4];
            weights8<F, OUT>(f, w);
#pragma unroll
            for (int mm = 0; mm < M; mm++) acc[mm] = dot8<OUT>(w, xs, cols, mm, j * 8, acc[mm]);
        }
        wave_sync();                                                      // the slice is rewritten at the top of the loop
#pragma unroll
        for (int mm = 0; mm < M; mm++) {
#ifdef GGQ_LIN_REDUCE_BPERMUTE      /* A/B builds only: the rounds 2-4 butterfly */
            float v = acc[mm];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
#else
            const float v0 = wave_sum_lane63(acc[mm]);
            float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v0), 63));      // wave-uniform: lane 0 stores it
#endif
            if (lane == 0) {
                if (bias_ != nullptr) {
                    float b;
                    if constexpr (OUT == OUT_F32) b = *reinterpret_cast<const float*>(bias_ + (size_t)r * 4);
                    else if constexpr (OUT == OUT_F16) b = (float)__builtin_bit_cast(_Float16, *reinterpret_cast<const uint16_t*>(bias_ + (size_t)r * 2));
                    else b = bits_f32((uint32_t)*reinterpret_cast<const uint16_t*>(bias_ + (size_t)r * 2) << 16);
                    v += b;
                }
                uint8_t* dst = y_ + ((size_t)mm * rows + r) * XB;
                if constexpr (OUT == OUT_F32) *reinterpret_cast<float*>(dst) = v;
                else if constexpr (OUT == OUT_F16) *reinterpret_cast<uint16_t*>(dst) = __builtin_bit_cast(uint16_t, (_Float16)v);
                else *reinterpret_cast<uint16_t*>(dst) = (uint16_t)(pack_bf16(v, 0.0f) & 0xFFFFu);
            }
        }
    }
}

}  // namespace ggq
