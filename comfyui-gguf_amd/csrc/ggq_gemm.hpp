// ggq_gemm.hpp -- gfx950 device code: y = x @ dequant(W)^T (+ bias) for MANY rows of x (hundreds to thousands: the token-side
// linears of a diffusion step), straight from the packed GGUF blocks, on the matrix cores -- the dense weight is never written
// to memory.  The many-rows end of SURVEY.md section 8f item 4 (reference ops.py:242-244: dequantize the whole weight, F.linear).
//
// Why a second MFMA kernel.  ggq_mfma.hpp lets every lane decode the eight weights that are its own MFMA operand: no dense tile
// anywhere, but every 32..128-row block of x re-decodes every weight it meets, so its cost grows with rows/64 and it loses to
// "unpack once + hipBLASLt" beyond ~256 rows.  Here the decode is amortised the way a GEMM amortises its operand
// loads: a workgroup owns a 256 x 256 output tile; per K-step of 32 it decodes its 256 x 32 weight tile ONCE -- 16 weights per
// thread, the same `fields` / `quad_f16` (+ `.to(dtype)`) code as the dequant kernels, so the weights are the reference's values
// bit for bit -- into LDS as fp16 / bf16, next to the 256 x 32 tile of x; all 16 waves then read MFMA fragments of both with
// ds_read_b128.  One decoded weight feeds 256 rows of x instead of 32..128.
//
// Shape of the work (16 waves = 1024 threads, four waves per SIMD, one workgroup per CU):
//   * LDS (GemmGeom below, per format): X[XR] and W[2] tiles of 256 rows x 64 B (32 K-elements) = 16 KiB each, plus STAGING = the packed bytes of the
//     tile's 256 weight rows for one span of K (one K-quant super-block = 256 elements; 4 blocks = 128 elements for the legacy formats) rounded up to
//     whole 16-byte units per thread: Q4_K (144 B per row-span -> 9 units) and Q8_0 (136 B, 2-byte aligned -> 10 units) 48 KiB, Q6_K (210 B -> 14) 64 KiB.
//     Two staging buffers where 4 tiles + 2 x STAGING <= 160 KiB (Q4_K, Q8_0: exactly 160 KiB, XR = 2); one buffer, refilled at the span boundary,
//     and a THIRD x tile otherwise (Q6_K: 5 x 16 + 64 = 144 KiB, XR = 3).  Everything arrives by LDS-DMA (global_load_lds_dwordx4: no staging
//     registers, no ds_write pass): packed spans one span ahead, x tiles one (XR = 2) or two (XR = 3) K-steps ahead, waited for with a COUNTED
//     s_waitcnt vmcnt and a raw s_barrier;
//   * tile rows are 64 B; the 16-byte column is XOR-swizzled with ((row >> 3) & 3) ^ ((row >> 1) & 1) (for the DMA'd x tile on the SOURCE
//     address, the LDS image of a DMA being lane-linear): ds_read_b128 fragment reads (one row per lane, gfx950's four non-contiguous 16-lane
//     groups, MI355X_MICROARCH.md LDS) and the decode's ds_write_b128 (4 lanes per row) are bank-conflict-free;
//   * K-step t: [DMA: x tile t+2] [decode weight tile t+1 -> W[next]: one chunk of 8 weights per thread] [MFMAs on X[t], W[t]: 2 k-slices of 16,
//     per wave 2 (n) x 2 (m) tiles of 32 x 32 = 8 v_mfma_f32_32x32x16] one s_barrier.  The waves of a SIMD alternate the order of decode and
//     MFMAs (ping-pong), so that the VALU and the matrix pipe of a SIMD have work at the same time;
//   * MFMA operands: A = weights (i = output column n), B = x (j = row m): a lane's accumulator registers then hold FOUR CONSECUTIVE
//     n of one row m, so the epilogue packs 8-byte pieces, transposes through LDS (XOR-swizzled, wave-private) and stores full
//     128-byte lines of y with 16 B per lane;
//   * workgroup -> tile mapping: XCD x (workgroup b runs on XCD b % 8) takes a contiguous eighth of the tiles in column-major
//     order, so the tiles that run together in one XCD share weight panels (read from HBM once) and x panels in that XCD's L2.
//
// Where the time goes, and why this kernel is FROZEN (round 4; EXPERIMENTS.md A2c / R4-2, profiles/r04_gemm_skeleton_sweep.json, r04_gemm_tile_skeleton_ablation.json):
// at 12288 x 3072 x 4608 rows the full kernel takes 382 us (0.91 PFLOP/s) on a box where unpack + hipBLASLt takes 311-323; without the decode 330, without
// decode AND LDS-DMA 269, MFMAs alone 213.  A standalone harness (tests/microbench/gemm_skel.hip) ran the no-decode skeleton as a real dense GEMM in
// every geometry the verdict named -- 16 waves of 64 x 64, 8 of 128 x 64, 4 of 128 x 128 (accumulators in AGPRs), BK 32 / 64, LDS rings 2-4 deep, three
// tile orders, fragments prefetched across the barrier: 0.86-1.01 PFLOP/s with real loads, 1.21-1.29 with NO loads at all, against the 1.35 the stop
// rule asked for.  The reason is power, not scheduling: with the operands coming out of LDS the matrix pipes are busy 91-96 % of the cycles -- at a
// shader clock of 1.70-1.78 GHz; an MFMA-only loop gets 2.2-2.4 GHz with near-zero operands and 1.87-1.91 GHz (1.60-1.67 PFLOP/s on this grid of
// 3.375 tile rounds) with random ones.  Every joule the decode's VALU work and the LDS traffic add comes out of the clock of the matrix pipe, so a fused
// kernel cannot beat a memory-bound unpack followed by a GEMM that already sits at that wall.  Hence: opt-in, capped at 256 rows of x by the callers.
//
// Numerics: weights = the reference's values bit for bit; products exact in fp32; fp32 accumulation in k order inside the MFMA,
// K-steps in order, no K split (deterministic).  Like any GEMM against another GEMM the result differs from hipBLASLt's by
// summation order: parity is a tolerance against an fp64 evaluation on the oracle's weights plus exact-arithmetic cases
// (tests/test_gpu_mfma.py), hence OPT-IN.
#pragma once

#include "ggq_mfma.hpp"

#include <type_traits>

namespace ggq {

constexpr int GT_BM = 256, GT_BN = 256, GT_BK = 32;
constexpr int GT_WAVES = 8, GT_THREADS = GT_WAVES * 64;
constexpr int GT_PITCH = GT_BK * 2;                      // bytes per tile row
constexpr int GT_TILE = 256 * GT_PITCH;                  // one X or W tile: 16 KiB

#ifndef GGQ_GT_CROSS
#define GGQ_GT_CROSS 0x100  /* what may be scheduled ACROSS the line between the two halves of a K-step: LDS reads (so the second half's operands
                               are already on their way while the first half runs); 0 = nothing; A/B builds */
#endif
#ifndef GGQ_GT_ABLATE
#define GGQ_GT_ABLATE 0     /* timing ablations for EXPERIMENTS.md (wrong results!): 1 = K-steps without the decode, 2 = without the MFMAs;
                               skeleton decomposition (round 4), all without the decode: 3 = MFMAs on loop-invariant registers (no fragment reads; DMA + barriers
                               stay), 4 = no LDS-DMA of x / packed spans (fragment reads + MFMAs + barriers), 5 = MFMAs only (no reads, no DMA, no barriers):
                               the ceiling of this grid -- tile quantisation and the clock under matrix load */
#endif
#ifndef GGQ_GT_XRING
#define GGQ_GT_XRING 1      /* a third x buffer where it fits: x tiles requested two K-steps ahead (0 = one step ahead); A/B builds */
#endif
#ifndef GGQ_GT_SETPRIO
#define GGQ_GT_SETPRIO 1    /* s_setprio around one half of a K-step: 1 = the MFMA half (fragment reads + MFMAs) runs at priority 1 -- 3-5 % faster at every
                               shape, profiles/r03_gemm_tile_setprio.json; 2 = the decode half (level); 0 = none; A/B builds */
#endif
#ifndef GGQ_GT_PRIO_LEVEL
#define GGQ_GT_PRIO_LEVEL 1  /* the priority (1..3) of the MFMA half; A/B builds */
#endif
#ifndef GGQ_GT_SHARED_SCALE
#define GGQ_GT_SHARED_SCALE 1 /* Q4_K / Q5_K: the (d*sc, dmin*mn) pair of a sub-block is computed once per SPAN by one of the row's four lanes and handed round
                                 by ds_bpermute, instead of by every lane in every K-step (same ops, same bits; -10 VALU of 55 per K-step); 0 = generic path; A/B builds */
#endif
#ifndef GGQ_GT_PONG_MASK
#define GGQ_GT_PONG_MASK 0x9 /* bit s: the wave in slot s of its SIMD (wave >> 2) runs the MFMAs of a K-step BEFORE its decode.  Slots {0, 3} measured best
                               (3 % over the alternating 0xA, profiles/r03_gemm_tile_wave_order.json); A/B builds */
#endif
#ifndef GGQ_GT_PINGPONG
#define GGQ_GT_PINGPONG 1   /* the waves of a SIMD alternate the order of decode and MFMAs (0 = same order in all waves); A/B builds */
#endif

GGQ_DEV uint32_t gt_swz(uint32_t row) { return ((row >> 3) & 3u) ^ ((row >> 1) & 1u); }

// How much of K is staged at a time: one super-block (256 elements) for the K-quants -- `fields` wants the whole block -- and 4 blocks (128
// elements) for the 32-element legacy formats, whose staging then fits twice beside the operand tiles (Q8_0: 2 x 40 KiB) so that they too are
// filled by LDS-DMA instead of through registers.
// formats whose K-step of 32 is exactly one sub-block with one (scale, min) pair: the pair can be precomputed per span and shared by the row's lanes
template <class F> struct SpanScales { static constexpr bool V = false; };
template <> struct SpanScales<FmtQ4_K> { static constexpr bool V = true; };
template <> struct SpanScales<FmtQ5_K> { static constexpr bool V = true; };

template <class F> struct TileSpan {
    static constexpr int SPAN = (F::BS == 32) ? 128 : 256;
    static constexpr int STEPS = SPAN / GT_BK;                                    // K-steps per staged span
    static constexpr int SPAN_BYTES = SPAN / F::BS * F::TS;
    // a span may start at any 2-byte boundary (Q6_K 210 B, Q3_K 110 B, 4 x 34 B, ...): every row keeps its own leading misalignment
    static constexpr bool ALIGNED = SPAN_BYTES % 16 == 0;
    // 16-byte load units per row, made ODD: the staging is filled linearly by LDS-DMA (unit u at u * 16), so its row pitch IS U units, and the decode reads 16 rows per
    // wave -- at an even pitch rows 8 apart (Q3_K's 8 units: EVERY row) fall on the same banks.  The extra unit of a row is never loaded (masked like the units past
    // the span) and never read.
    // Measured against the even-pitch build (profiles/r06_lds_row_pitch.json): Q3_K (8 -> 9 units) 34-40 % faster, Q6_K (14 -> 15) up to 8 %, Q8_0 / IQ4_XS (10 -> 11)
    // 0-1 %; the 6-unit rows (Q4_0, IQ4_NL, Q5_1: rows 8 apart share banks, a 2-way conflict) ran 1-3 % SLOWER at 7 and stay at 6.
    static constexpr int U0 = (SPAN_BYTES + (ALIGNED ? 0 : 14) + 15) / 16;
#ifdef GGQ_GT_EVEN_PITCH   /* A/B builds only: rounds 2-5 */
    static constexpr int U = U0;
#else
    static constexpr int U = U0 == 6 ? U0 : (U0 | 1);
#endif
    static constexpr int ROW_STRIDE = U * 16;
};

template <class F, int WM = 4> struct GemmGeom {
    using G = TileSpan<F>;
    static constexpr int THREADS = WM * 4 * 64;                                  // WM x 4 waves
    static constexpr int UNITS = GT_BN * G::U;                                   // 16-byte units of one staged span
    static constexpr int NUW = (UNITS + THREADS - 1) / THREADS;                  // units per thread
    static constexpr int STAGING = NUW * THREADS * 16;                           // LDS bytes (>= 256 * ROW_STRIDE)
    // XR = x tiles in flight: with room for a THIRD x buffer the x tile of step t + 2 is requested at step t and only the one of step t + 1 is
    // waited for.  DBUF: the packed span after the current one lands in a second staging buffer while this one is decoded; where two do not
    // fit (Q6_K: 2 x 56 KiB) the single buffer is refilled at the span boundary and the workgroup waits for it once per 8 K-steps.
    static constexpr bool DBUF = 4 * GT_TILE + 2 * STAGING <= 160 * 1024;
    static constexpr int XR = (GGQ_GT_XRING && 5 * GT_TILE + (DBUF ? 2 : 1) * STAGING <= 160 * 1024) ? 3 : 2;
    static constexpr int LDS_BYTES = (2 + XR) * GT_TILE + (DBUF ? 2 : 1) * STAGING;
};

// one 16-byte piece per lane, global -> LDS without passing through registers; the LDS address is wave-uniform base + 16 * lane
GGQ_DEV void dma16(const GGQ_GLOBAL uint8_t* src, uint8_t* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const GGQ_GLOBAL void*)src, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// WM = waves along the rows of x: 2 (8 waves, each 128 x 64 of the tile: 128 accumulator registers, 2 waves per SIMD) or 4 (16 waves, each
// 64 x 64: 64 accumulator registers, 128 registers per wave, FOUR waves per SIMD -- twice the wavefronts to hide each other's LDS and
// barrier latency at 33 % more fragment reads per MFMA).
template <class F, int OUT, int WM = 4>
__global__ __launch_bounds__(WM * 256) void linear_tile(const uint8_t* __restrict__ packed_, const uint8_t* __restrict__ x_,
                                                        const uint8_t* __restrict__ bias_, uint8_t* __restrict__ y_,
                                                        uint32_t m, uint32_t n_rows, uint32_t cols, uint32_t tiles_m, uint32_t tiles_n)
{
    using G = TileSpan<F>;
    using GG = GemmGeom<F, WM>;
    static_assert(OUT == OUT_F16 || OUT == OUT_BF16, "16-bit activations only");
    static_assert(WM == 2 || WM == 4, "8 or 16 waves");
    constexpr int THREADS = GG::THREADS;
    constexpr int MT = 8 / WM;                       // 32-row blocks of x per wave
    constexpr int TPR = THREADS / 256;               // decoding threads per weight row
    constexpr int CPT = 4 / TPR;                     // chunks (8 weights) per thread per K-step
    constexpr int XU = 1024 / THREADS;               // 16-byte units of the x tile per thread per K-step
    constexpr int CPB = F::BS / 8;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int XR = GG::XR;
    uint8_t* const xt = smem;                        // X[0 .. XR)
    uint8_t* const wt = smem + XR * GT_TILE;         // W[0], W[1]
    uint8_t* const stg = smem + (XR + 2) * GT_TILE;

    const uint32_t t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int)(t >> 6));
    const uint32_t lane = t & 63u;

    // ---- which tile: XCD x takes a contiguous eighth of the column-major tile list (tm fastest)
    const uint32_t n_tiles = tiles_m * tiles_n;
    uint32_t tile = blockIdx.x;
    {
        const uint32_t q = n_tiles >> 3, r = n_tiles & 7u, xcd = tile & 7u, idx = tile >> 3;
        tile = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + idx;       // bijective for any n_tiles
    }
    const uint32_t tn = tile / tiles_m, tm = tile - tn * tiles_m;
    const uint32_t m0 = tm * GT_BM, n0 = tn * GT_BN;

    const gcptr packed = (gcptr)packed_;
    const uint64_t row_bytes = (uint64_t)(cols / F::BS) * F::TS;
    constexpr uint32_t STEPS = (uint32_t)G::STEPS;
    const uint32_t n_spans = cols / (uint32_t)G::SPAN, n_steps = n_spans * STEPS;

    // ---- staging fill by LDS-DMA: unit u of the tile's span = (row u / U, 16-byte piece u % U); thread takes units t, t + THREADS, ...
    // (lanes whose unit lies outside the span are masked: those bytes are never read; the last unit of a row may reach past the row's span by
    // < 16 bytes inside its aligned 16-byte unit: same page, never faults)
    auto dma_span = [&](uint32_t span, uint32_t buf) {
#pragma unroll
        for (int u = 0; u < GG::NUW; u++) {
            const uint32_t unit = t + (uint32_t)(THREADS * u), ur = unit / (uint32_t)G::U, uu = unit - ur * (uint32_t)G::U;
            const uint32_t rr = (n0 + ur < n_rows) ? n0 + ur : n_rows - 1;
            const uint64_t off = (uint64_t)rr * row_bytes + (uint64_t)span * G::SPAN_BYTES;
            const uint32_t a = G::ALIGNED ? 0u : ((uint32_t)off & 15u);
            if (ur < (uint32_t)GT_BN && uu * 16u < a + (uint32_t)G::SPAN_BYTES)
                dma16(packed + (off - a) + uu * 16u, stg + buf * (uint32_t)GG::STAGING + ((uint32_t)wave * 64u + (uint32_t)(THREADS * u)) * 16u);
        }
    };
    // ---- weight decode: thread -> (row t / TPR, CPT consecutive chunks of the K-step's four)
    const uint32_t drow = t / (uint32_t)TPR, dc0 = (t % (uint32_t)TPR) * (uint32_t)CPT;
    const uint32_t wrow = (n0 + drow < n_rows) ? n0 + drow : n_rows - 1;
    const uint64_t wrow_off = (uint64_t)wrow * row_bytes;
    const uint32_t dswz = gt_swz(drow);
    // Q4_K / Q5_K with one chunk per thread: lane c of a row's four holds the (d*sc, dmin*mn) pairs of sub-blocks c and c + 4 of the current span
    constexpr bool SHARED = GGQ_GT_SHARED_SCALE && SpanScales<F>::V && CPT == 1 && G::SPAN == 256;
    uint32_t pre0 = 0, pre1 = 0;
    const uint32_t quad_lane0 = ((uint32_t)lane & ~3u) << 2;                        // ds_bpermute address of the row's first lane
    auto prescale = [&](uint32_t span) {
        if constexpr (SHARED) {
            const uint32_t a = G::ALIGNED ? 0u : ((uint32_t)(wrow_off + (uint64_t)span * G::SPAN_BYTES) & 15u);
            const uint8_t* blk = stg + (GG::DBUF ? (span & 1u) * (uint32_t)GG::STAGING : 0u) + drow * (uint32_t)G::ROW_STRIDE + a;
            const u32x4 hdr = *reinterpret_cast<const u32x4*>(blk);                 // [d][dmin][scales 12]
            int32_t sc, mn;
            k_scale_min(hdr, (int)dc0, sc, mn);
            pre0 = as_u32(as_h2(hdr.x) * ints_h2((uint32_t)sc | ((uint32_t)mn << 16), 0.0f));      // the very expression of quad_f16<K_SCMN>
            k_scale_min(hdr, (int)dc0 + 4, sc, mn);
            pre1 = as_u32(as_h2(hdr.x) * ints_h2((uint32_t)sc | ((uint32_t)mn << 16), 0.0f));
        }
    };
    auto decode = [&](uint32_t step, uint8_t* wdst) {
#if GGQ_GT_SETPRIO == 2            /* A/B builds: the reverse -- the decoding wave outranks the one that feeds the matrix pipe */
        __builtin_amdgcn_s_setprio(1);
#endif
        const uint32_t span = step / STEPS, ks = step % STEPS;
        const uint32_t a = G::ALIGNED ? 0u : ((uint32_t)(wrow_off + (uint64_t)span * G::SPAN_BYTES) & 15u);
        const uint8_t* wspan = stg + (GG::DBUF ? (span & 1u) * (uint32_t)GG::STAGING : 0u) + drow * (uint32_t)G::ROW_STRIDE + a;
        if constexpr (SHARED) {
            // sub-block ks of the span: its pair sits in lane ks % 4 of the row, slot ks / 4
            const uint32_t mine = (ks & 4u) ? pre1 : pre0;
            const h2 dlml = as_h2((uint32_t)__builtin_amdgcn_ds_bpermute((int)(quad_lane0 | ((ks & 3u) << 2)), (int)mine));
            const h2 dl = bcast_lo(dlml), ml = bcast_hi(dlml);
            const uint32_t j = ks * 4u + dc0;
            const Fields f = F::template fields<true>(wspan, (int)j);             // only the quants are used: the scale decode in there is dead code
            const H2x2 q0 = fields_h2(f.t0, (float)F::BIAS), q1 = fields_h2(f.t1, (float)F::BIAS);
            uint32_t w[4] = {as_u32(dl * q0.a - ml), as_u32(dl * q0.b - ml), as_u32(dl * q1.a - ml), as_u32(dl * q1.b - ml)};
            if constexpr (OUT == OUT_BF16) {
#pragma unroll
                for (int k = 0; k < 4; k++) w[k] = h2_to_bf16x2(w[k]);
            }
            *reinterpret_cast<u32x4*>(wdst + drow * GT_PITCH + ((dc0 ^ dswz) * 16u)) = u32x4{w[0], w[1], w[2], w[3]};
        } else {
#pragma unroll
            for (int s = 0; s < CPT; s++) {
                const uint32_t c = dc0 + (uint32_t)s, j = ks * 4u + c;                 // chunk of the 256-element span
                const Fields f = F::template fields<true>(wspan + (j / CPB) * F::TS, (int)(j % CPB));
                uint32_t w[4];
                weights8<F, OUT>(f, w);
                *reinterpret_cast<u32x4*>(wdst + drow * GT_PITCH + ((c ^ dswz) * 16u)) = u32x4{w[0], w[1], w[2], w[3]};
            }
        }
#if GGQ_GT_SETPRIO == 2
        __builtin_amdgcn_s_setprio(0);
#endif
    };

    // ---- x tile by LDS-DMA: unit u = t + THREADS i -> (row u / 4, 16-byte piece u % 4); the LDS image is lane-linear, so the XOR swizzle goes
    // on the SOURCE piece
    const GGQ_GLOBAL uint8_t* xsrc[XU];
#pragma unroll
    for (int i = 0; i < XU; i++) {
        const uint32_t row = (t >> 2) + (uint32_t)(THREADS / 4 * i), mr = m0 + row;
        xsrc[i] = (GGQ_GLOBAL const uint8_t*)x_ + (uint64_t)(mr < m ? mr : m - 1) * cols * 2 + (((t & 3u) ^ gt_swz(row)) * 16u);
    }
    auto xdma = [&](uint32_t step, uint8_t* xd) {
#pragma unroll
        for (int i = 0; i < XU; i++) dma16(xsrc[i] + (uint64_t)step * GT_PITCH, xd + ((uint32_t)wave * 64u + (uint32_t)(THREADS * i)) * 16u);
    };

    // ---- MFMA roles: wave -> (wm = wave / 4: rows of x [32 MT wm, +32 MT), wn = wave % 4: output columns [64 wn, +64))
    const uint32_t wm = (uint32_t)wave >> 2, wn = (uint32_t)wave & 3u;
    const uint32_t r32 = lane & 31u, hk = lane >> 5, fswz = gt_swz(r32);
    f32x16 acc[2][MT];
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[nt][mt][i] = 0.0f;

    // one k-slice of 16: the wave's 2 weight fragments and MT x fragments (one ds_read_b128 each), then its 2 MT MFMAs
    auto mma = [&](const uint8_t* xs, const uint8_t* ws) {
#if GGQ_GT_SETPRIO == 1            /* the wave that feeds the matrix pipe outranks the ones that decode */
        __builtin_amdgcn_s_setprio(GGQ_GT_PRIO_LEVEL);
#endif
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            const uint32_t col = (((uint32_t)(2 * kk) + hk) ^ fswz) * 16u;
            u32x4 wa[2], xb[MT];
#if GGQ_GT_ABLATE == 3 || GGQ_GT_ABLATE == 5   /* operands from registers: the matrix pipe alone */
#pragma unroll
            for (int nt = 0; nt < 2; nt++) wa[nt] = u32x4{lane + col, lane, (uint32_t)nt, 0x3c003c00u};
#pragma unroll
            for (int mt = 0; mt < MT; mt++) xb[mt] = u32x4{lane, lane ^ col, (uint32_t)mt, 0x3c003c00u};
            (void)xs; (void)ws;
#else
#pragma unroll
            for (int nt = 0; nt < 2; nt++) wa[nt] = *reinterpret_cast<const u32x4*>(ws + (64u * wn + 32u * (uint32_t)nt + r32) * GT_PITCH + col);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) xb[mt] = *reinterpret_cast<const u32x4*>(xs + ((uint32_t)(32 * MT) * wm + 32u * (uint32_t)mt + r32) * GT_PITCH + col);
#endif
#if GGQ_GT_SETPRIO == 3            /* A/B builds: priority 1 only while the MFMAs issue, not for the fragment reads */
            __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int mt = 0; mt < MT; mt++) acc[nt][mt] = mfma32<OUT>(wa[nt], xb[mt], acc[nt][mt]);
#if GGQ_GT_SETPRIO == 3
            __builtin_amdgcn_s_setprio(0);
#endif
        }
#if GGQ_GT_SETPRIO == 1
        __builtin_amdgcn_s_setprio(0);
#endif
    };

    // ---- prologue: span 0 staged, x tiles 0 and 1 requested, weight tile 0 decoded
    auto dma_fence = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every LDS-DMA of this wave has landed; the barrier then publishes it
        __syncthreads();
    };
    // the fence of a K-step with XR = 3: the x tile requested THIS step (the XU newest DMAs) may stay in flight; everything older -- the x tile
    // the next step reads, a packed span -- has landed.  Raw s_barrier: __syncthreads() would make hipcc drain vmcnt to 0 behind our back.
    auto ring_fence = [&]() {
        if constexpr (XU == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    dma_span(0u, 0u);
    xdma(0u, xt);
    dma_fence();
    if (GG::DBUF && n_spans > 1) dma_span(1u, 1u);
    prescale(0u);
    decode(0u, wt);
    xdma(n_steps > 1 ? 1u : 0u, xt + GT_TILE);
    dma_fence();

    // ---- main loop.  One K-step = [decode weight tile t+1 -> W[next]] + [MFMAs on X[cur], W[cur]] + [x tile t+1 -> X[next]] + one s_barrier.
    // decode(t + 1) and mma(t) are independent, so a wave may run them in either order: the waves that share a SIMD (w, w + 4, w + 8, ...: a
    // workgroup's waves are dealt to the four SIMDs cyclically) take ALTERNATING orders (PONG) -- while one feeds the matrix pipe another keeps
    // the VALU busy with the decode, then they swap; sched_barrier keeps the compiler from mixing the halves back together.  Each order is its
    // own copy of the loop (a wave-uniform branch around the whole loop, not inside it: the accumulator registers never meet in a phi); steps
    // come in pairs so that the buffer parity is a compile-time constant.
    auto kstep = [&](uint32_t step, auto parity_tag, auto pong_tag, auto decode_tag) {
        constexpr int P = decltype(parity_tag)::value;
        constexpr bool PONG = decltype(pong_tag)::value, DECODE = decltype(decode_tag)::value;
        uint8_t* const xcur = xt + (XR == 3 ? step % 3u : (uint32_t)P) * GT_TILE;
        uint8_t* const wcur = wt + P * GT_TILE;
        uint8_t* const xnxt = xt + (XR == 3 ? (step + 2u) % 3u : (uint32_t)(P ^ 1)) * GT_TILE;    // XR = 3: the buffer of step + 2 (read last at step - 1)
        uint8_t* const wnxt = wt + (P ^ 1) * GT_TILE;
        if (GGQ_GT_ABLATE < 4 && DECODE && (step + 1) % STEPS == 0) {
            // the next K-step opens a new span: every decode of the old one finished before the previous barrier
            const uint32_t span = (step + 1) / STEPS;
            if constexpr (GG::DBUF) {
                // span `span` landed in its buffer steps ago; the OTHER buffer (span - 1) is free: start filling it with span + 1
                if (span + 1 < n_spans) dma_span(span + 1, (span + 1) & 1u);
            } else {
                dma_span(span, 0u);                     // one buffer: refill it now and wait for it (once per span)
                dma_fence();
            }
            prescale(span);                             // the new span's bytes are in LDS (fenced steps ago, or just now)
        }
        if constexpr (DECODE && GGQ_GT_ABLATE < 4) {
            // x tile of step + 2 (XR = 3: two K-steps to land) or of step + 1; clamped at the end: a harmless re-read into a buffer nobody reads
            if constexpr (XR == 3) xdma(step + 2 < n_steps ? step + 2 : n_steps - 1, xnxt);
            else xdma(step + 1, xnxt);
        }
#if GGQ_GT_ABLATE == 1 || GGQ_GT_ABLATE >= 3   /* timing ablation (wrong results): no decode */
        mma(xcur, wcur);
        if constexpr (false)
#elif GGQ_GT_ABLATE == 2           /* timing ablation (wrong results): no MFMAs */
        if constexpr (DECODE) decode(step + 1, wnxt);
        if constexpr (false)
#endif
        if constexpr (!DECODE) {
            mma(xcur, wcur);
        } else if constexpr (PONG && GGQ_GT_PINGPONG) {
            mma(xcur, wcur);
            __builtin_amdgcn_sched_barrier(GGQ_GT_CROSS);
            decode(step + 1, wnxt);
        } else {
            decode(step + 1, wnxt);
#if GGQ_GT_PINGPONG
            __builtin_amdgcn_sched_barrier(GGQ_GT_CROSS);
#endif
            mma(xcur, wcur);
        }
#if GGQ_GT_ABLATE == 5
        asm volatile("" ::: "memory");
#else
        if constexpr (XR == 3) ring_fence();
        else dma_fence();
#endif
    };
    auto main_loop = [&](auto pong_tag) {
        using T0 = std::integral_constant<int, 0>;
        using T1 = std::integral_constant<int, 1>;
        uint32_t step = 0;
        for (; step + 2 < n_steps; step += 2) {                     // n_steps is a multiple of 8 (cols % 256 == 0)
            kstep(step, T0{}, pong_tag, std::true_type{});
            kstep(step + 1, T1{}, pong_tag, std::true_type{});
        }
        kstep(step, T0{}, pong_tag, std::true_type{});
        kstep(step + 1, T1{}, pong_tag, std::false_type{});        // the last step has nothing left to decode
    };
    // which of a SIMD's waves (slots wave >> 2 = 0 .. WM*4/4-1) run their MFMAs first: GGQ_GT_PONG_MASK bit s = slot s does (0x9: slots 0 and 3)
    if ((GGQ_GT_PONG_MASK >> (wave >> 2)) & 1) main_loop(std::true_type{});
    else main_loop(std::false_type{});
    if constexpr (XR == 3) dma_fence();                      // nothing may still be landing in LDS when the epilogue reuses it

    // ---- epilogue: bias, cast, transpose through LDS (wave-private 4 KiB: 32 rows of x  x  64 columns), full-line stores.
    // C/D layout of the 32x32 MFMA: register i of lane l holds D[row = (i & 3) + 8 (i >> 2) + 4 (l >> 5)][col = l & 31]; here
    // row = output column n inside its 32-block, col = row of x inside its 32-block.
    uint8_t* const ep = smem + wave * 4096;
    const Window ywin = window((gptr)y_ + ((uint64_t)m0 * n_rows + n0) * 2, 0xFFFFFFFFu);            // this tile of y: offsets < 256 rows x n_rows x 2 B
    const uint32_t nbase = n0 + 64u * wn, mbase = m0 + (uint32_t)(32 * MT) * wm;
    float bias[2][4][4];
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t n = nbase + 32u * (uint32_t)nt + 8u * (uint32_t)q + 4u * hk;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                float b = 0.0f;
                if (bias_ != nullptr && n + (uint32_t)k < n_rows) {
                    const uint16_t bb = *reinterpret_cast<const uint16_t*>(bias_ + (size_t)(n + (uint32_t)k) * 2);
                    if constexpr (OUT == OUT_F16) b = (float)__builtin_bit_cast(_Float16, bb);
                    else b = bits_f32((uint32_t)bb << 16);
                }
                bias[nt][q][k] = b;
            }
        }
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float v0 = acc[nt][mt][4 * q + 0] + bias[nt][q][0], v1 = acc[nt][mt][4 * q + 1] + bias[nt][q][1];
                const float v2 = acc[nt][mt][4 * q + 2] + bias[nt][q][2], v3 = acc[nt][mt][4 * q + 3] + bias[nt][q][3];
                u32x2 o;
                if constexpr (OUT == OUT_F16) o = u32x2{pack_f16(v0, v1), pack_f16(v2, v3)};
                else o = u32x2{pack_bf16(v0, v1), pack_bf16(v2, v3)};
                const uint32_t p = 4u * (uint32_t)nt + (uint32_t)q;              // 16-byte piece = columns 8p .. 8p+7
                *reinterpret_cast<u32x2*>(ep + r32 * 128u + ((p ^ (r32 & 7u)) * 16u) + 8u * hk) = o;
            }
        wave_sync();
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t row = (lane >> 3) + 8u * (uint32_t)i, p = lane & 7u;
            const u32x4 v = *reinterpret_cast<const u32x4*>(ep + row * 128u + ((p ^ (row & 7u)) * 16u));
            const uint32_t mr = mbase + 32u * (uint32_t)mt + row, nc = nbase + 8u * p;
            if (mr < m && nc < n_rows) wstore(ywin, ((mr - m0) * n_rows + (nc - n0)) * 2u, v);              // n_rows % 8 == 0, n_rows <= 4 M (host)
        }
        wave_sync();
    }
}

}  // namespace ggq
