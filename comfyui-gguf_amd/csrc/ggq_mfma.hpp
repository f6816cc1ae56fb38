// ggq_mfma.hpp -- gfx950 device code: y = x @ dequant(W)^T (+ bias) for MANY rows of x, straight from the packed GGUF blocks,
// on the matrix cores -- the dense weight is never written to memory.
//
// Where it sits (the large-m end): GGMLOps.Linear.forward_ggml_cast_weights (reference ops.py:242-244)
// dequantizes the whole weight (0.56 B read + 2 B written per weight for Q4_K) and hands it to F.linear (2 B read again).  Here a
// wave decodes exactly the 8 consecutive weights it needs, in registers, into the operand of one MFMA instruction.
//
// Why this fits the hardware so directly: v_mfma_f32_32x32x16_{f16,bf16} wants from lane l of the B operand the 8 values
// B[k = 8*(l/32) .. +7][j = l%32] -- eight CONSECUTIVE k of ONE column j.  With B = W^T that is eight consecutive elements of one
// weight row: exactly one CHUNK of the dequant kernels (ggq_device.hpp), which a lane already decodes on its own from the packed
// bytes.  So `F::fields` + `quad_f16` (+ the `.to(dtype)` rounding) produce the MFMA operand in place; no dense tile in LDS, no
// transposes.  The A operand (x, i = row of x) has the same shape: eight consecutive k of one row, one 16-byte load per lane.
// The contraction index may be permuted freely as long as both operands use the same permutation, so inside a 64-element span of
// k a lane reads its four A fragments as 64 contiguous bytes.
//
// Shape of the work:
//   * a workgroup owns a tile of MB*32 rows of x  x  32 rows of W (= 32 output columns); its 4 waves split K: wave w takes the
//     256-element spans w, w+4, ... of the contraction (K % 256 == 0; for the 32-element legacy blocks also K % 64 == 0 with a shorter
//     LAST span -- SD3.5's 2432-column layers -- round 5) and all of the tile, so every weight of the tile is
//     decoded ONCE per workgroup; at the end the four partial accumulators are summed through LDS in a fixed order (deterministic);
//   * per span a wave copies the span's packed bytes of its 32 weight rows (32 x 144 B for Q4_K) into its private LDS slice with
//     16 B/lane loads -- the next span's bytes are already in flight in registers -- then runs 16 k-steps: decode one chunk
//     (~35-45 VALU lane-ops), MB MFMAs (32 cycles each) against MB A fragments (see linear_mfma below for how they are fetched);
//   * no __syncthreads in the main loop (waves share nothing until the reduction).
//
// Numerics: the WEIGHTS are the reference's values bit for bit (same decode, same fp16 op sequence, then the `.to(dtype)` of
// dequantize_tensor); products are exact in fp32, accumulation is fp32 in the MFMA's order.  Like any GEMM against another GEMM the
// result differs from hipBLASLt's by summation order: parity is a tolerance against an fp64 evaluation on the oracle's weights
// (tests/test_gpu_mfma.py); part of install()'s default for up to 256 rows of x since round 5 (profiles/r05_fused_error.json), off under `exact`.
#pragma once

#include "ggq_linear.hpp"

#ifndef GGQ_MF_ABLATE
#define GGQ_MF_ABLATE 0      /* A/B builds only (WRONG results): 1 = no decode (the MFMA eats raw LDS bytes), 2 = no global loads of x, 4 = (64+ rows of x) no LDS staging of x either */
#endif
#ifndef GGQ_MF_XLDS_MIN_MB
#define GGQ_MF_XLDS_MIN_MB 2   /* blocks of 32 rows of x from which a wave stages its pieces of x through LDS (coalesced 64-byte loads) instead of loading fragments straight from global memory */
#endif
#ifndef GGQ_MF_SETPRIO
#define GGQ_MF_SETPRIO 0     /* s_setprio 1 around the MFMAs of a k-step: 4-20 % SLOWER here (1-4 MFMAs per toggle; EXPERIMENTS A2c), unlike the shared-tile kernel; A/B builds */
#endif

namespace ggq {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int OUT>
GGQ_DEV f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c)
{
    if constexpr (OUT == OUT_F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

constexpr int MF_WAVES = 4;       // waves per workgroup = K-split factor of rounds 2-5; since round 6 the launch picks 4..16 (blockDim.x / 64): see mf_lds_bytes / ggq_linear.hip
constexpr int mf_max_waves(int mb) { return mb == 1 ? 12 : (mb == 2 ? 8 : 4); }   // the register budget of a workgroup's waves: 170 / 256 / 512 per lane
constexpr int MF_SPAN = 256;      // contraction elements per span (one K-quant super-block, 8 legacy blocks)

template <class F> struct MfmaGeom {
    static constexpr int SPAN_BYTES = MF_SPAN / F::BS * F::TS;                     // packed bytes of one row's span
    // a span may start at any 2-byte boundary (Q6_K 210 B, Q3_K 110 B, ...): every row keeps its own leading misalignment.  The 32-element formats count as
    // unaligned too: with a short last span (cols % 256 != 0: SD3.5's 2432 columns) a ROW is no longer a multiple of 16 bytes (Q5_0: 1672), so the rows of a
    // tile start at different offsets mod 16 although every span is 176 bytes (treating them as aligned made the 16-byte loads of odd rows
    // start unaligned and the last one reach up to 12 bytes past the row -- past the tensor on its last row)
#ifdef GGQ_MF_LEGACY_ALIGNED   /* A/B builds only: rounds 2-5 (32-element formats treated as 16-byte aligned rows: wrong for cols % 256 != 0) */
    static constexpr bool ALIGNED = SPAN_BYTES % 16 == 0;
#else
    static constexpr bool ALIGNED = SPAN_BYTES % 16 == 0 && F::BS != 32;
#endif
    static constexpr int U = (SPAN_BYTES + (ALIGNED ? 0 : 14) + 15) / 16;         // 16-byte load units per row
    // LDS pitch of a row: an ODD number of 16-byte units.  Lane r decodes row r, so a pitch of 12 units (Q5_0 once its rows count as unaligned: 192 B = 48 banks)
    // puts rows r and r + 4 on the same banks -- measured 35-55 % slower than 11 units on SD3.5's Q5_0 layers (profiles/r06_lds_row_pitch.json); Q3_K's 8 units
    // (128 B) was a 16-way conflict since round 2.  With an odd pitch only rows 16 apart share banks, whatever the format.
    static constexpr int PITCH_U = U | 1;
    static constexpr int ROW_STRIDE = PITCH_U * 16;
    static constexpr int NUW = (32 * U + 63) / 64;                                 // load units per lane
    static constexpr int SLICE = ((NUW * 64 + U - 1) / U) * ROW_STRIDE;            // LDS bytes per wave: every unit a lane holds has a place, the ones past row 31 too
};

// MB = 32-row blocks of x per workgroup tile (1, 2, 4, 8).
// How a wave gets its A fragments (x): with ONE block of rows (MB = 1) straight from global memory -- lane (r, h) reads its 64
// contiguous bytes of a 64-element span, the 2 KiB a wave touches per step stay in the vector L1.  With MB >= 2 that pattern
// (32 B of each 128-B line per instruction) overflows the L1 -- 4 waves x MB x 4 KiB per step -- and every line is fetched from L2
// several times (measured: a 128-row tile ran 2.5x SLOWER than four 32-row tiles).  So for MB >= 2 the wave first copies a
// (MB*32 rows x 32 elements) piece of x into its own LDS slice with coalesced 16 B/lane loads (4 lanes per row: whole 64-B
// sectors, each fetched once; the next TWO pieces are already in flight in registers) and reads the fragments back with ds_read_b128.
// LDS layout of a piece: 64 bytes per row, the 16-byte column XOR-swizzled by (row >> 3) & 3.  gfx950 serves a ds_read_b128 in four
// NON-contiguous 16-lane groups ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32; MI355X_MICROARCH.md, LDS) against 16 slots of
// 16 bytes: with one row per lane at 64 B per row, the four rows of a group that share (row mod 4) must sit in four different columns, and
// their (row >> 3) are 0, 1, 2, 3 in every group -- conflict-free; ds_write_b128 goes in contiguous 8-lane groups (2 rows x 4 pieces) against 8
// slots and is conflict-free under any XOR of the column.  (First version: rows padded to 80 bytes, SQ_LDS_BANK_CONFLICT = 33 % of the LDS
// cycles and, with one piece in flight, 45 % of wave cycles in s_waitcnt; second: an XOR derived from contiguous lane groups, still 19 % --
// profiles/r02_mfma_kernel_counters.txt.)
constexpr int MF_XPITCH = 64;
GGQ_DEV uint32_t mf_swz(uint32_t row) { return (row >> 3) & 3u; }

// LDS bytes of a workgroup of `kw` waves: per wave its slice of packed bytes (+ its piece of x when MB >= 2) during the main loop, one accumulator block
// (16 registers x 64 lanes x 4 B) per wave for the reduction after it
template <class F, int MB> constexpr uint32_t mf_lds_bytes(uint32_t kw)
{
    const uint32_t per_wave = (uint32_t)MfmaGeom<F>::SLICE + (MB >= GGQ_MF_XLDS_MIN_MB ? (uint32_t)(MB * 32 * 64) : 0u), red = 16u * 64u * 4u;
    return kw * (per_wave > red ? per_wave : red);
}

template <class F, int OUT, int MB>
__global__ __launch_bounds__(mf_max_waves(MB) * 64) void linear_mfma(const uint8_t* __restrict__ packed_, const uint8_t* __restrict__ x_,
                                                             const uint8_t* __restrict__ bias_, uint8_t* __restrict__ y_,
                                                             uint32_t m, uint32_t n_rows, uint32_t cols, float* __restrict__ partial_)
{
    // partial_ != nullptr: K is ALSO split across workgroups (gridDim.z slices of whole spans, round 6): this workgroup contracts its slice only and stores its fp32
    // partial sums to partial_[blockIdx.z][m][n_rows]; splitk_reduce adds the slices in order, the bias, and casts -- deterministic, like the in-workgroup sum.
    // (Adding the slices in the last workgroup of each tile instead -- ticket counters, one launch -- was built twice: with device-scope fences it ran 6-9x SLOWER (an L2
    // write-back + invalidate per workgroup on a part with eight L2s); fence-free, through device-scope relaxed atomics, it is correct and level with this.  And requesting a span's x fragments BEFORE the next span's weight prefetch, so that the in-order memory pipe does not make
    // them wait for it, 10-45 % slower than this order: EXPERIMENTS.md R6-11, profiles/r06_mfma32_ablations.json.)
    using G = MfmaGeom<F>;
    static_assert(OUT == OUT_F16 || OUT == OUT_BF16, "16-bit activations only (an fp32 MFMA runs at 1/16 of the rate)");
    constexpr int CPB = F::BS / 8;                                                 // chunks per block
    constexpr int RED = 16 * 64 * 4;                                               // one accumulator block of one wave, bytes
    constexpr bool XLDS = MB >= GGQ_MF_XLDS_MIN_MB;
    constexpr int XS = XLDS ? MB * 32 * MF_XPITCH : 0;                             // LDS bytes per wave for its piece of x
    constexpr int PER_WAVE = G::SLICE + XS;
    static_assert(RED == 16 * 64 * 4 && (!XLDS || XS == MB * 32 * 64), "mf_lds_bytes() restates these");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];                 // mf_lds_bytes<F, MB>(kw)

    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t kw = blockDim.x >> 6;                                           // waves of this workgroup = K-split factor (4..16, the launch's choice)
    const int lane = (int)(threadIdx.x & 63);
    const int r = lane & 31, h = lane >> 5;
    const uint32_t n0 = blockIdx.x * 32u, m0 = blockIdx.y * (uint32_t)(MB * 32);
    const gcptr packed = (gcptr)packed_;
    const uint64_t row_bytes = (uint64_t)(cols / F::BS) * F::TS;
    const uint32_t n_spans = (cols + MF_SPAN - 1) / MF_SPAN;
    const uint32_t tail_len = cols - (n_spans - 1) * (uint32_t)MF_SPAN;           // elements of the LAST span: 256, or a multiple of 64 below it (32-element blocks only)
    auto span_len = [&](uint32_t span) { return span + 1 == n_spans ? tail_len : (uint32_t)MF_SPAN; };
    // this workgroup's slice of the spans: [lo, hi)
    const uint32_t per_z = (n_spans + gridDim.z - 1) / gridDim.z;
    const uint32_t lo = blockIdx.z * per_z, hi = (lo + per_z < n_spans) ? lo + per_z : n_spans;
    const uint32_t first = lo + (uint32_t)wave;                                    // this wave's first span
    uint8_t* slice = smem + wave * PER_WAVE;
    uint8_t* xs = slice + G::SLICE;

    // the weight row this lane decodes (clamped at the edge: the stores are masked instead)
    const uint32_t wrow = (n0 + (uint32_t)r < n_rows) ? n0 + (uint32_t)r : n_rows - 1;
    const uint64_t wrow_off = (uint64_t)wrow * row_bytes;

    // copy of one span's packed bytes for the 32 rows: unit = (row, 16-byte piece); lane takes units lane, lane + 64, ...
    auto fetch = [&](uint32_t span, u32x4 (&pf)[G::NUW]) {
#pragma unroll
        for (int u = 0; u < G::NUW; u++) {
            const uint32_t unit = (uint32_t)(lane + 64 * u), ur = unit / (uint32_t)G::U, uu = unit % (uint32_t)G::U;
            const uint32_t rr = (n0 + ur < n_rows) ? n0 + ur : n_rows - 1;
            const uint64_t off = (uint64_t)rr * row_bytes + (uint64_t)span * G::SPAN_BYTES;
            const uint32_t a = G::ALIGNED ? 0u : ((uint32_t)off & 15u);
            const uint32_t span_bytes = span_len(span) / (uint32_t)F::BS * (uint32_t)F::TS;       // == SPAN_BYTES but for a short last span
            // every load starts 16-byte aligned (off - a), so what it reads past the row's span (< 16 bytes) lies inside an aligned unit that also holds bytes of the tensor
            pf[u] = (ur < 32u && uu * 16u < a + span_bytes) ? gload16<false>(packed + (off - a) + uu * 16u) : u32x4{0, 0, 0, 0};
        }
    };

    f32x16 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; mb++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[mb][i] = 0.0f;

    // one k-step: decode chunk j of the span (8 weights of row r) and run it against the MB fragments of x
    auto step = [&](const uint8_t* wspan, int j, const u32x4 (&xa)[MB]) {
#if GGQ_MF_ABLATE & 1
        const u32x4 wb = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint8_t*>(reinterpret_cast<uintptr_t>(wspan) & ~(uintptr_t)15) + (j % (G::U - 1)) * 16);
#else
        const Fields f = F::template fields<true>(wspan + (j / CPB) * F::TS, j % CPB);
        uint32_t w[4];
        weights8<F, OUT>(f, w);
        const u32x4 wb{w[0], w[1], w[2], w[3]};
#endif
#if GGQ_MF_SETPRIO
        __builtin_amdgcn_s_setprio(1);             // A/B builds: a wave with MFMAs to issue outranks the ones that decode
#endif
#pragma unroll
        for (int mb = 0; mb < MB; mb++) acc[mb] = mfma32<OUT>(xa[mb], wb, acc[mb]);
#if GGQ_MF_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    };

    u32x4 pf[G::NUW];
    if (first < hi) fetch(first, pf);
    uint32_t stash_at[G::NUW];                                                     // where this lane's units go in the slice: row * pitch + piece * 16
#pragma unroll
    for (int u = 0; u < G::NUW; u++) {
        const uint32_t unit = (uint32_t)(lane + 64 * u);
        if constexpr (G::PITCH_U == G::U) stash_at[u] = unit * 16u;               // (one register + immediates)
        else stash_at[u] = unit / (uint32_t)G::U * (uint32_t)G::ROW_STRIDE + unit % (uint32_t)G::U * 16u;
    }

    if constexpr (!XLDS) {
        const uint32_t mr = m0 + (uint32_t)r;
        const GGQ_GLOBAL uint8_t* xrow = (GGQ_GLOBAL const uint8_t*)x_ + (uint64_t)(mr < m ? mr : m - 1) * cols * 2 + (uint32_t)(h * 64);
        for (uint32_t span = first; span < hi; span += kw) {
#pragma unroll
            for (int u = 0; u < G::NUW; u++) *reinterpret_cast<u32x4*>(slice + stash_at[u]) = pf[u];
            wave_sync();
            if (span + kw < hi) fetch(span + kw, pf);                              // the next span's bytes fly while this one is decoded
            const uint32_t a = G::ALIGNED ? 0u : ((uint32_t)(wrow_off + (uint64_t)span * G::SPAN_BYTES) & 15u);
            const uint8_t* wspan = slice + r * G::ROW_STRIDE + a;
            const uint32_t kbyte = span * (uint32_t)(MF_SPAN * 2);
            const uint32_t len = span_len(span);
#pragma unroll
            for (int q = 0; q < 4; q++) {                                          // 64 contraction elements per q: k = 64 q + 32 h + 8 s .. + 7
                if ((uint32_t)(q * 64) >= len) break;                              // a short last span (wave-uniform)
                u32x4 xq[4];
#pragma unroll
                for (int s4 = 0; s4 < 4; s4++) {
#if GGQ_MF_ABLATE & 2
                    xq[s4] = u32x4{(uint32_t)lane, span, (uint32_t)q, (uint32_t)s4};
#else
                    xq[s4] = *(GGQ_GLOBAL const u32x4*)(xrow + kbyte + (uint32_t)(q * 128 + s4 * 16));
#endif
                }
#pragma unroll
                for (int s4 = 0; s4 < 4; s4++) {
                    const u32x4 xa[MB] = {xq[s4]};
                    step(wspan, 8 * q + 4 * h + s4, xa);
                }
            }
            wave_sync();                                                           // the slice is rewritten at the top of the loop
        }
    } else {
        // pieces of x: (MB*32 rows) x (32 elements = 64 bytes); lane -> (row lane/4 + 16 i, 16-byte piece lane%4), MB*2 loads per piece
        constexpr int NX = MB * 2;
        const int lrow = lane >> 2, lpc = lane & 3;
        const GGQ_GLOBAL uint8_t* xsrc[NX];
        uint32_t xdst[NX];
#pragma unroll
        for (int i = 0; i < NX; i++) {
            const uint32_t row = (uint32_t)(lrow + 16 * i), mr = m0 + row;
            xsrc[i] = (GGQ_GLOBAL const uint8_t*)x_ + (uint64_t)(mr < m ? mr : m - 1) * cols * 2 + (uint32_t)(lpc * 16);
            xdst[i] = row * (uint32_t)MF_XPITCH + (((uint32_t)lpc ^ mf_swz(row)) * 16u);
        }
        const uint32_t swr = mf_swz((uint32_t)r);                                  // rows 32 mb + r swizzle like row r
        u32x4 ring[2][NX];                                                         // pieces g and g+1 in flight while piece g-1 is consumed
        auto xfetch = [&](uint32_t piece, u32x4 (&dst)[NX]) {                      // piece = index over the wave's own sequence of 32-element pieces
            // the wave's p-th piece: span = first + kw (p / 8), t = p % 8
            const uint32_t kb = ((first + kw * (piece >> 3)) * (uint32_t)(MF_SPAN * 2)) + (piece & 7u) * 64u;
#pragma unroll
            for (int i = 0; i < NX; i++) {
#if GGQ_MF_ABLATE & 2
                dst[i] = u32x4{(uint32_t)lane, kb, (uint32_t)i, piece};
#else
                dst[i] = *(GGQ_GLOBAL const u32x4*)(xsrc[i] + kb);
#endif
            }
        };
        const uint32_t my_spans = (first < hi) ? (hi - first + kw - 1) / kw : 0u;
        // 8 pieces of 32 elements per span; the wave that owns the LAST span has fewer in it when that span is short
        const bool owns_last = my_spans > 0 && first + kw * (my_spans - 1) == n_spans - 1;
        const uint32_t my_pieces = my_spans * 8u - (owns_last ? (uint32_t)(MF_SPAN - tail_len) / 32u : 0u);
        if (my_pieces > 0) xfetch(0u, ring[0]);
        if (my_pieces > 1) xfetch(1u, ring[1]);
        uint32_t piece = 0;
        for (uint32_t span = first; span < hi; span += kw) {
#pragma unroll
            for (int u = 0; u < G::NUW; u++) *reinterpret_cast<u32x4*>(slice + stash_at[u]) = pf[u];
            wave_sync();
            if (span + kw < hi) fetch(span + kw, pf);
            const uint32_t a = G::ALIGNED ? 0u : ((uint32_t)(wrow_off + (uint64_t)span * G::SPAN_BYTES) & 15u);
            const uint8_t* wspan = slice + r * G::ROW_STRIDE + a;
            const uint32_t len = span_len(span);
#pragma unroll
            for (int t = 0; t < 8; t++, piece++) {                                 // 32 contraction elements per t: k = 32 t + 16 h + 8 s .. + 7
                if ((uint32_t)(t * 32) >= len) break;                              // a short last span (wave-uniform; it is the wave's last one)
#if !(GGQ_MF_ABLATE & 4)
#pragma unroll
                for (int i = 0; i < NX; i++) *reinterpret_cast<u32x4*>(xs + xdst[i]) = ring[t & 1][i];
                wave_sync();
#endif
                if (piece + 2 < my_pieces) xfetch(piece + 2, ring[t & 1]);
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++) {
                    u32x4 xa[MB];
                    const uint32_t col = (((uint32_t)(2 * h + s2)) ^ swr) * 16u;
#pragma unroll
                    for (int mb = 0; mb < MB; mb++) {
#if GGQ_MF_ABLATE & 4
                        xa[mb] = u32x4{ring[t & 1][mb].x, col, (uint32_t)lane, piece};
#else
                        xa[mb] = *reinterpret_cast<const u32x4*>(xs + (mb * 32 + r) * MF_XPITCH + col);
#endif
                    }
                    step(wspan, 4 * t + 2 * h + s2, xa);
                }
                wave_sync();                                                       // xs is rewritten at the top of the t loop
            }
        }
    }

    // ---- sum the kw K-partials through LDS, fixed order (wave 0 + 1 + ... ), then bias, cast, store.
    // C/D layout of the 32x32 MFMA: register i of lane l holds D[row = (i & 3) + 8 (i >> 2) + 4 (l >> 5)][col = l & 31]; here row = row
    // of x inside its 32-block, col = output column.  In round mb every wave parks its accumulator block; wave w then finishes registers
    // w, w + kw, ... of the block for all lanes.
    float bias = 0.0f;
    const uint32_t ncol = n0 + (uint32_t)r;
    if (bias_ != nullptr && ncol < n_rows) {
        const uint16_t b = *reinterpret_cast<const uint16_t*>(bias_ + (size_t)ncol * 2);
        if constexpr (OUT == OUT_F16) bias = (float)__builtin_bit_cast(_Float16, b);
        else bias = bits_f32((uint32_t)b << 16);
    }
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int mb = 0; mb < MB; mb++) {
        __syncthreads();                                                           // main loop / previous round done with smem
#pragma unroll
        for (int i = 0; i < 16; i++) red[(wave * 16 + i) * 64 + lane] = acc[mb][i];
        __syncthreads();
        for (uint32_t i = (uint32_t)wave; i < 16u; i += kw) {
            float v = red[i * 64 + lane];
            for (uint32_t w = 1; w < kw; w++) v += red[((w * 16u + i) * 64u) + lane];
            const uint32_t mr = m0 + (uint32_t)(mb * 32) + (i & 3u) + 8u * (i >> 2) + 4u * (uint32_t)h;
            if (partial_ != nullptr) {
                if (mr < m && ncol < n_rows) partial_[((size_t)blockIdx.z * m + mr) * n_rows + ncol] = v;
                continue;
            }
            v += bias;
            if (mr < m && ncol < n_rows) {
                uint16_t o;
                if constexpr (OUT == OUT_F16) o = __builtin_bit_cast(uint16_t, (_Float16)v);
                else o = (uint16_t)(pack_bf16(v, 0.0f) & 0xFFFFu);
                *reinterpret_cast<uint16_t*>(y_ + ((size_t)mr * n_rows + ncol) * 2) = o;
            }
        }
    }
}

// y[mr][n] = cast(sum over the gridDim.z slices of partial[z][mr][n], in slice order, + bias[n]): the second pass of a launch whose K was split across workgroups
template <int OUT>
__global__ __launch_bounds__(256) void splitk_reduce(const float* __restrict__ partial, const uint8_t* __restrict__ bias_, uint8_t* __restrict__ y_, uint32_t m, uint32_t n_rows, uint32_t zs)
{
    const uint32_t n = blockIdx.x * 256u + threadIdx.x, mr = blockIdx.y;         // (no integer division on the device: its expansion goes through fused multiply-adds)
    if (n >= n_rows) return;
    const uint64_t total = (uint64_t)m * n_rows, i = (uint64_t)mr * n_rows + n;
    float v = partial[i];
    for (uint32_t z = 1; z < zs; z++) v += partial[(uint64_t)z * total + i];
    if (bias_ != nullptr) {
        const uint16_t b = *reinterpret_cast<const uint16_t*>(bias_ + (size_t)n * 2);
        if constexpr (OUT == OUT_F16) v += (float)__builtin_bit_cast(_Float16, b);
        else v += bits_f32((uint32_t)b << 16);
    }
    uint16_t o;
    if constexpr (OUT == OUT_F16) o = __builtin_bit_cast(uint16_t, (_Float16)v);
    else o = (uint16_t)(pack_bf16(v, 0.0f) & 0xFFFFu);
    *reinterpret_cast<uint16_t*>(y_ + i * 2) = o;
}

}  // namespace ggq
