// ggq_mfma16.hpp -- gfx950 device code: y = x @ dequant(W)^T (+ bias) for ONE TO 32 rows of x, straight from the packed GGUF blocks, on
// v_mfma_f32_16x16x32_{f16,bf16} -- the small-batch end of GGMLOps.Linear.forward_ggml_cast_weights (reference ops.py:242-244).  Round 6.
//
// Why a second MFMA kernel.  Below ~16 rows of x neither round-5 kernel fits: the 32x32x16 kernel of ggq_mfma.hpp launches 384 workgroups x 4 waves on 1024
// SIMDs (12288 x 3072) and decodes 32 rows of W per wave for at most 8 useful rows of x; the one-wave-per-row GEMV of ggq_linear.hpp spends a third of
// its instructions on per-row work (fetch predicates, the scale pass, the cross-lane sum: 61 VALU per 8 weights against 40 in the chunk loop itself) and 4
// v_dot2c per chunk and row of x, so its time grows with m (11.3 us at one row, 17.1 at four).  tools/probes/valu_rates.hip (profiles/r06_valu_issue_rates.json)
// measured what the decode's instructions cost: v_and / v_lshrrev / v_fma_f32 issue at 2 cycles per wave, EVERYTHING else the decode is made of (v_perm_b32,
// v_pk_*_f16, v_cvt_*, v_dot2c_*, three-operand integer ops) at 4.  So: the contraction goes to the matrix core (one MFMA per chunk whatever m is), the
// scale decode is shared by four chunks, and there are enough waves per SIMD that one wave's memory wait is another's issue slot.
//
// What it reaches and what bounds it (Q4_K, bf16, graph replay; profiles/r06_mfma16_*.json): 12288 x 3072 at 1 / 4 / 8 rows of x 11.0 / 11.0 / 11.9 us
// (GEMV 11.3 / 17.1 / -, 32-row kernel 13.6 / 13.5 / 13.5); 3072 x 12288 at 1..16 rows 12.3..18.5 us (32-row kernel 20.7..28: it has 96 workgroups there).
// Taken apart in lab builds: the load -> LDS -> MFMA -> reduce skeleton alone runs 6.5 us (21 MB: 3.3 us of HBM time + launch + one memory latency + the
// reduction), the decode adds 2.6 at one row (of 4.8 us of VALU issue: the rest hides), the x loads 0.8 at one row -- but 5.3 at 16 rows and 12.8 at 32: every
// workgroup reads all of x (75 / 150 MB through L2 -> L1 against 21 MB of weights, at ~13 TB/s), which is why the host hands 9+ rows on tall weights to the
// 32-row kernel.  Tried against that wall and not kept (EXPERIMENTS.md R6-3): an x-contiguous k map (one 64-byte sector per row and instruction: level), two
// blocks of W per wave (half the x traffic, 105 registers: level to slower), blocks of W side by side meeting in L1 (no reuse: slower), x staged once per
// K-split group in LDS behind two barriers per span (level with the 32-row kernel at 32 rows, slower below), two spans of packed bytes in flight (slower).
//
// Shape of the work:
//   * a workgroup owns 16 rows of W (= 16 output columns) x MB*16 rows of x; its KW waves (2..16, chosen per launch from the number of 256-element
//     spans and of workgroups) split K: wave w takes spans w, w + KW, ...  768 workgroups x 6 waves for 12288 x 3072 = 4.5 waves per SIMD;
//   * A operand = W: lane (r = lane % 16, c = lane / 16) holds A[row r][k = 8c .. 8c+7] of a 32-wide k-step.  The contraction index may be permuted
//     as long as both operands agree, so inside a GROUP of 128 elements lane (r, c) takes the 32 CONSECUTIVE elements 32c .. 32c+31 of row r, one
//     chunk of 8 per k-step t = 0..3.  For the K-quants that is one whole sub-block per lane and group: its (scale, min) pair is decoded once per
//     four chunks by the ordinary F::fields (the compiler hoists it out of the unrolled t loop), its quant bytes are 32 contiguous bytes; for the
//     32-element formats it is one whole block;
//   * B operand = x: lane (r, c) reads the same 32 elements of row r of x as 64 contiguous bytes straight from global memory (x is at most
//     32 x cols x 2 bytes and every workgroup reads all of it: L2- and mostly L1-resident).  The x loads of a span are issued BEFORE the next span's
//     weight prefetch, so the s_waitcnt in front of the first MFMA waits for them and not for the HBM stream behind them;
//   * per span a wave copies the packed bytes of its 16 rows into a private LDS slice (16 B / lane, the next span in flight in registers; row pitch
//     an ODD number of 16-byte units: the 16 rows of a ds_read_b128 lane group fall on 16 distinct bank quads);
//   * D[row = 4c + i][col = r] = y[x row r][W row 4c + i]: after the K-partials are summed through LDS in wave order (deterministic), a lane owns
//     four consecutive outputs of one row of y -- one 8-byte store.
//
// Numerics: as ggq_mfma.hpp -- the weights are the reference's values bit for bit (same decode, same fp16 op sequence, then the `.to(dtype)` of
// dequantize_tensor), products exact in fp32, accumulation in fp32 in the MFMA's order, K-partials added in wave order.
#pragma once

#include "ggq_mfma.hpp"

namespace ggq {

template <int OUT>
GGQ_DEV f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c)
{
    if constexpr (OUT == OUT_F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

constexpr int MF16_MAX_WAVES = 16;

// what the lanes of a short last span read in place of x elements past the row's end
__device__ const u32x4 MF16_ZEROS[4] = {};

template <class F> struct Mf16Geom {
    static constexpr int SPAN_BYTES = MF_SPAN / F::BS * F::TS;                     // packed bytes of one row's span
    // a row's span may start at any 2-byte boundary: block sizes that are not multiples of 16 (Q6_K 210 B, Q3_K 110 B, ...), and -- 32-element
    // formats only -- rows that are not a whole number of spans (SD3.5's 2432 columns: a Q5_0 row is 1672 bytes).  Such formats always load from the
    // aligned address below the span and keep the misalignment as an offset inside the LDS row (the 32x32 kernel assumed aligned
    // rows for every format whose SPAN is a multiple of 16 bytes).
    static constexpr bool SKEW = SPAN_BYTES % 16 != 0 || F::BS == 32;
    static constexpr int U = (SPAN_BYTES + (SKEW ? 14 : 0) + 15) / 16;            // 16-byte load units per row
    static constexpr int ROW_STRIDE = (U | 1) * 16;                                // LDS pitch: an odd number of units
    static constexpr int NUW = (16 * U + 63) / 64;                                 // load units per lane
    static constexpr int SLICE = 16 * ROW_STRIDE;                                  // LDS bytes per wave
};

// LDS bytes of a workgroup of `kw` waves: the weight slices during the main loop, the K-partials (64 lanes x 16 B per wave and block of x) after it
template <class F, int MB> constexpr uint32_t mf16_lds_bytes(uint32_t kw)
{
    const uint32_t a = kw * (uint32_t)Mf16Geom<F>::SLICE, b = kw * (uint32_t)MB * 1024u;
    return a > b ? a : b;
}

#ifndef GGQ_MF16_ABLATE
#define GGQ_MF16_ABLATE 0       /* A/B builds only (WRONG results): 1 = no decode (the MFMA eats raw LDS bytes), 2 = no x loads, 3 = both: the floor of the load -> LDS -> MFMA -> reduce skeleton */
#endif
// occupancy the register allocator is asked for: 5 waves per SIMD with one block of x (<= 96 registers), 4 with two (<= 128) -- where that fits.  The formats
// whose rows start at any 2-byte offset carry more state (4-5 load units per lane, the skew, the tail masks) and SPILLED under those caps (Q8_0: 72-108 bytes of
// scratch per lane, Q5_0 28-52, Q4_0 / IQ4_NL 12-44, IQ4_XS 8-36; Q6_K and Q5_1 with two blocks of x only): a spill is a vector-memory access in front of the
// in-order weight stream, and Q8_0 ran at 25 us where the one-row GEMV takes 12 (profiles/r06_format_sweep.json).  They get one wave per SIMD less.
template <class F> struct Mf16Occ { static constexpr int MB1 = 5, MB2 = 4; };
#ifndef GGQ_MF16_UNIFORM_OCC   /* A/B builds only: the same caps for every format (round 6 before the format sweep) */
template <> struct Mf16Occ<FmtQ4_0> { static constexpr int MB1 = 4, MB2 = 3; };
template <> struct Mf16Occ<FmtQ5_0> { static constexpr int MB1 = 4, MB2 = 3; };
template <> struct Mf16Occ<FmtQ8_0> { static constexpr int MB1 = 4, MB2 = 3; };
template <> struct Mf16Occ<FmtIQ4_NL> { static constexpr int MB1 = 4, MB2 = 3; };
template <> struct Mf16Occ<FmtIQ4_XS> { static constexpr int MB1 = 4, MB2 = 3; };
template <> struct Mf16Occ<FmtQ6_K> { static constexpr int MB1 = 5, MB2 = 3; };
template <> struct Mf16Occ<FmtQ5_1> { static constexpr int MB1 = 5, MB2 = 3; };
#endif
// the widest workgroup of an instantiation: 16 waves cap the allocator at 128 registers (4 waves per SIMD); the ones asked for 3 per SIMD get 12 (170 registers)
template <class F, int MB> constexpr int mf16_max_waves() { return (MB == 2 && Mf16Occ<F>::MB2 < 4) ? 12 : MF16_MAX_WAVES; }
#define GGQ_MF16_OCC(F, MB) __attribute__((amdgpu_waves_per_eu((MB) == 1 ? Mf16Occ<F>::MB1 : Mf16Occ<F>::MB2, 8)))

template <class F, int OUT, int MB>
__global__ __launch_bounds__((mf16_max_waves<F, MB>()) * 64) GGQ_MF16_OCC(F, MB) void linear_mfma16(const uint8_t* __restrict__ packed_, const uint8_t* __restrict__ x_,
                                                                      const uint8_t* __restrict__ bias_, uint8_t* __restrict__ y_,
                                                                      uint32_t m, uint32_t n_rows, uint32_t cols)
{
    using G = Mf16Geom<F>;
    static_assert(OUT == OUT_F16 || OUT == OUT_BF16, "16-bit activations only");
    constexpr int CPB = F::BS / 8;                                                 // chunks per block
    extern __shared__ __attribute__((aligned(16))) uint8_t smem16[];

    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t kw = blockDim.x >> 6;
    const int lane = (int)(threadIdx.x & 63);
    const int r = lane & 15, c = lane >> 4;
    const uint32_t n0 = blockIdx.x * 16u, m0 = blockIdx.y * (uint32_t)(MB * 16);
    const gcptr packed = (gcptr)packed_;
    const uint64_t row_bytes = (uint64_t)(cols / F::BS) * F::TS;
    const uint32_t n_spans = (cols + MF_SPAN - 1) / MF_SPAN;
    const uint32_t tail_len = cols - (n_spans - 1) * (uint32_t)MF_SPAN;           // elements of the LAST span: 256, or a multiple of 64 below it (32-element blocks only)
    uint8_t* slice = smem16 + wave * G::SLICE;

    // the weight row this lane decodes (clamped at the edge: the stores are masked instead)
    const uint32_t wrow = (n0 + (uint32_t)r < n_rows) ? n0 + (uint32_t)r : n_rows - 1;
    const uint64_t wrow_off = (uint64_t)wrow * row_bytes;

    // copy of one span's packed bytes for the 16 rows: unit = (row, 16-byte piece); lane takes units lane, lane + 64, ...
    // A short last span (32-element formats only) is handled by DATA, not by control flow: the bytes past the span's end are zeroed on their way
    // into LDS, so the blocks that do not exist decode to d = 0 -> weights 0.0 exactly, and the lanes that own them read their x fragments from
    // 64 bytes of zeros (MF16_ZEROS) instead of from past the row's end.  The k-steps themselves are the same code for every span.
    auto fetch = [&](uint32_t span, u32x4 (&pf)[G::NUW]) {
        const bool tail = F::BS == 32 && span + 1 == n_spans && tail_len != (uint32_t)MF_SPAN;       // wave-uniform
        const uint32_t span_bytes = (tail ? tail_len : (uint32_t)MF_SPAN) / (uint32_t)F::BS * (uint32_t)F::TS;
#pragma unroll
        for (int u = 0; u < G::NUW; u++) {
            const uint32_t unit = (uint32_t)(lane + 64 * u), ur = unit / (uint32_t)G::U, uu = unit % (uint32_t)G::U;
            const uint32_t rr = (n0 + ur < n_rows) ? n0 + ur : n_rows - 1;
            const uint64_t off = (uint64_t)rr * row_bytes + (uint64_t)span * G::SPAN_BYTES;
            const uint32_t a = G::SKEW ? ((uint32_t)off & 15u) : 0u;
            // every load starts at a 16-byte aligned address, so the bytes it reads past the span (< 16, in its last unit) are in the same aligned
            // 16-byte unit as bytes of the tensor: never a page the tensor does not touch
            pf[u] = (ur < 16u && uu * 16u < a + span_bytes) ? gload16<true>(packed + (off - a) + uu * 16u) : u32x4{0, 0, 0, 0};
            if (tail) {                                                            // zero what lies past the span inside its last unit (offsets are even)
                const int32_t valid = (int32_t)(a + span_bytes) - (int32_t)(uu * 16u);
                uint32_t d[4] = {pf[u].x, pf[u].y, pf[u].z, pf[u].w};
#pragma unroll
                for (int i = 0; i < 4; i++) d[i] &= valid >= 4 * i + 4 ? 0xFFFFFFFFu : (valid == 4 * i + 2 ? 0x0000FFFFu : 0u);
                pf[u] = u32x4{d[0], d[1], d[2], d[3]};
            }
        }
    };
    auto stash = [&](const u32x4 (&pf)[G::NUW]) {
#pragma unroll
        for (int u = 0; u < G::NUW; u++) {
            const uint32_t unit = (uint32_t)(lane + 64 * u), ur = unit / (uint32_t)G::U, uu = unit % (uint32_t)G::U;
            if (ur < 16u) *reinterpret_cast<u32x4*>(slice + ur * G::ROW_STRIDE + uu * 16u) = pf[u];
        }
    };

    // x: at k-step t of group g lane (r, c) holds elements 128 g + 32 c + 8 t .. + 7 of row (block mb, r): 64 contiguous bytes per lane and group
    const GGQ_GLOBAL uint8_t* xrow[MB];
#pragma unroll
    for (int mb = 0; mb < MB; mb++) {
        const uint32_t mr = m0 + (uint32_t)(mb * 16 + r);
        xrow[mb] = (GGQ_GLOBAL const uint8_t*)x_ + (uint64_t)(mr < m ? mr : m - 1) * cols * 2 + (uint32_t)(c * 64);
    }

    f32x4 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; mb++) acc[mb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    u32x4 pf[G::NUW];
    if ((uint32_t)wave < n_spans) fetch((uint32_t)wave, pf);
    for (uint32_t span = (uint32_t)wave; span < n_spans; span += kw) {
        // this span's fragments of x first, the next span's weights behind them (in-order vmcnt: waiting for x does not wait for the prefetch)
        u32x4 xf[MB][8];
        const uint32_t kb = span * (uint32_t)(MF_SPAN * 2);
#pragma unroll
        for (int g = 0; g < 2; g++) {
#pragma unroll
            for (int mb = 0; mb < MB; mb++) {
                const GGQ_GLOBAL uint8_t* src = xrow[mb] + kb + (uint32_t)(g * 256);
                if constexpr (F::BS == 32) {                                       // past the row's end (a short last span): zeros
                    if (span * (uint32_t)MF_SPAN + (uint32_t)(128 * g + 32 * c) >= cols) src = (GGQ_GLOBAL const uint8_t*)MF16_ZEROS;
                }
#pragma unroll
                for (int t = 0; t < 4; t++) {
#if GGQ_MF16_ABLATE & 2
                    xf[mb][4 * g + t] = u32x4{(uint32_t)lane, span, (uint32_t)t, (uint32_t)g};
#else
                    xf[mb][4 * g + t] = *(GGQ_GLOBAL const u32x4*)(src + t * 16);
#endif
                }
            }
        }
        stash(pf);
        wave_sync();
        if (span + kw < n_spans) fetch(span + kw, pf);                             // the next span's bytes fly while this one is decoded
        const uint32_t a = G::SKEW ? ((uint32_t)(wrow_off + (uint64_t)span * G::SPAN_BYTES) & 15u) : 0u;
        const uint8_t* wspan = slice + r * G::ROW_STRIDE + a;
#pragma unroll
        for (int g = 0; g < 2; g++) {
            if (g) __builtin_amdgcn_sched_barrier(0);                              // keep the second group's LDS reads and decode out of the first group's registers
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int j = 16 * g + 4 * c + t;                                  // chunk of the span: elements 128 g + 32 c + 8 t .. + 7
#if GGQ_MF16_ABLATE & 1
                const u32x4 wa = *reinterpret_cast<const u32x4*>(slice + (r * G::ROW_STRIDE + (j % (G::U - 1)) * 16));
#else
                const Fields f = F::template fields<true>(wspan + (j / CPB) * F::TS, j % CPB);
                uint32_t w[4];
                weights8<F, OUT>(f, w);
                const u32x4 wa{w[0], w[1], w[2], w[3]};
#endif
#pragma unroll
                for (int mb = 0; mb < MB; mb++) acc[mb] = mfma16<OUT>(wa, xf[mb][4 * g + t], acc[mb]);
            }
        }
        wave_sync();                                                               // the slice is rewritten at the top of the next span
    }

    // ---- sum the KW partials through LDS in wave order, then bias, cast, store.  Wave mb % KW finishes block mb of x.
    __syncthreads();                                                               // every wave is done with its slice
    f32x4* red = reinterpret_cast<f32x4*>(smem16);
#pragma unroll
    for (int mb = 0; mb < MB; mb++) red[((uint32_t)wave * MB + (uint32_t)mb) * 64 + lane] = acc[mb];
    __syncthreads();
    for (uint32_t mb = (uint32_t)wave; mb < (uint32_t)MB; mb += kw) {
        f32x4 v = red[mb * 64 + lane];
        for (uint32_t w = 1; w < kw; w++) {
            const f32x4 p = red[(w * MB + mb) * 64 + lane];
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        // D[row = 4c + i][col = r] = y[x row r][W row 4c + i]
        const uint32_t mr = m0 + mb * 16u + (uint32_t)r, nc = n0 + (uint32_t)(4 * c);     // row of y, first of this lane's four output columns
        if (mr >= m || nc >= n_rows) continue;
        float o[4] = {v.x, v.y, v.z, v.w};
        uint16_t h[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (bias_ != nullptr && nc + i < n_rows) {
                const uint16_t bb = *reinterpret_cast<const uint16_t*>(bias_ + (size_t)(nc + i) * 2);
                if constexpr (OUT == OUT_F16) o[i] += (float)__builtin_bit_cast(_Float16, bb);
                else o[i] += bits_f32((uint32_t)bb << 16);
            }
            if constexpr (OUT == OUT_F16) h[i] = __builtin_bit_cast(uint16_t, (_Float16)o[i]);
            else h[i] = (uint16_t)(pack_bf16(o[i], 0.0f) & 0xFFFFu);
        }
        uint8_t* dst = y_ + ((size_t)mr * n_rows + nc) * 2;
        if (nc + 4 <= n_rows && (n_rows & 3u) == 0 && ((uintptr_t)y_ & 7u) == 0) {
            *reinterpret_cast<u32x2*>(dst) = u32x2{(uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16)};
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (nc + i < n_rows) *reinterpret_cast<uint16_t*>(dst + 2 * i) = h[i];
        }
    }
}

}  // namespace ggq
