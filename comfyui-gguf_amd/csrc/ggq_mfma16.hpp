// ggq_mfma16.hpp -- gfx950 device code: y = x @ dequant(W)^T (+ bias) for ONE TO 32 rows of x, straight from the packed GGUF blocks, on
// v_mfma_f32_16x16x32_{f16,bf16} -- the small-batch end of GGMLOps.Linear.forward_ggml_cast_weights (reference ops.py:242-244).  Round 6.
//
// Why a second MFMA kernel.  At <= 32 rows of x the 32x32x16 kernel of ggq_mfma.hpp is not bound by anything the hardware is good at: 384 workgroups
// x 4 waves on 1024 SIMDs (12288 x 3072), every wave parked in s_waitcnt for 56 % of its life (profiles/r02_mfma_kernel_counters.txt) -- three spans
// per wave, each one a dependent chain HBM -> registers -> LDS -> decode with nothing else resident on the SIMD to run meanwhile -- and the one-wave-
// per-row GEMV of ggq_linear.hpp spends a third of its instructions on per-row work (fetch predicates, the scale pass, the cross-lane sum: 61 VALU
// per 8 weights against 40 in the chunk loop itself) and 4 half-rate v_dot2c per chunk and row of x.  tools/probes/valu_rates.hip (profiles/
// r06_valu_issue_rates.json) measured what the decode's instructions cost: v_and / v_lshrrev / v_fma_f32 issue at 2 cycles per wave, EVERYTHING
// else the decode is made of (v_perm_b32, v_pk_*_f16, v_cvt_*, v_dot2c_*, three-operand integer ops) at 4.  So: fewer instructions per weight,
// and enough waves per SIMD that one wave's memory wait is another's issue slot.
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

// NT = 16-row blocks of W a wave decodes per span (1 or 2: the workgroup's tile is NT*16 output columns)
template <class F, int NT> struct Mf16Geom {
    static constexpr int ROWS = NT * 16;
    static constexpr int SPAN_BYTES = MF_SPAN / F::BS * F::TS;                     // packed bytes of one row's span
    // a row's span may start at any 2-byte boundary: block sizes that are not multiples of 16 (Q6_K 210 B, Q3_K 110 B, ...), and -- 32-element
    // formats only -- rows that are not a whole number of spans (SD3.5's 2432 columns: a Q5_0 row is 1672 bytes).  Such formats always load from the
    // aligned address below the span and keep the misalignment as an offset inside the LDS row (the 32x32 kernel assumed aligned
    // rows for every format whose SPAN is a multiple of 16 bytes).
    static constexpr bool SKEW = SPAN_BYTES % 16 != 0 || F::BS == 32;
    static constexpr int U = (SPAN_BYTES + (SKEW ? 14 : 0) + 15) / 16;            // 16-byte load units per row
    static constexpr int ROW_STRIDE = (U | 1) * 16;                                // LDS pitch: an odd number of units
    static constexpr int NUW = (ROWS * U + 63) / 64;                               // load units per lane
    static constexpr int SLICE = ROWS * ROW_STRIDE;                                // LDS bytes per wave
};

// LDS bytes of a workgroup of `kw` waves: the weight slices during the main loop, the K-partials (64 lanes x 16 B per wave, block of W and block of x) after it
template <class F, int MB, int NT> constexpr uint32_t mf16_lds_bytes(uint32_t kw)
{
    const uint32_t a = kw * (uint32_t)Mf16Geom<F, NT>::SLICE, b = kw * (uint32_t)(MB * NT) * 1024u;
    return a > b ? a : b;
}

#ifndef GGQ_MF16_WPE
#define GGQ_MF16_WPE 1          /* occupancy the register allocator is asked for: 5 waves per SIMD with one block of x and of W, 4 otherwise (A/B builds: 0 = whatever 1024 threads allow) */
#endif
#if GGQ_MF16_WPE
#define GGQ_MF16_OCC(MB, NT) __attribute__((amdgpu_waves_per_eu((MB) * (NT) == 1 ? 5 : 4, 8)))
#else
#define GGQ_MF16_OCC(MB, NT)
#endif
template <class F, int OUT, int MB, bool XC = false, int NT = 1>
__global__ __launch_bounds__(MF16_MAX_WAVES * 64) GGQ_MF16_OCC(MB, NT) void linear_mfma16(const uint8_t* __restrict__ packed_, const uint8_t* __restrict__ x_,
                                                                      const uint8_t* __restrict__ bias_, uint8_t* __restrict__ y_,
                                                                      uint32_t m, uint32_t n_rows, uint32_t cols)
{
    using G = Mf16Geom<F, NT>;
    static_assert(OUT == OUT_F16 || OUT == OUT_BF16, "16-bit activations only");
    constexpr int CPB = F::BS / 8;                                                 // chunks per block
    extern __shared__ __attribute__((aligned(16))) uint8_t smem16[];

    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t kw = blockDim.x >> 6;
    const int lane = (int)(threadIdx.x & 63);
    const int r = lane & 15, c = lane >> 4;
    const uint32_t n0 = blockIdx.x * (uint32_t)G::ROWS, m0 = blockIdx.y * (uint32_t)(MB * 16);
    const gcptr packed = (gcptr)packed_;
    const uint64_t row_bytes = (uint64_t)(cols / F::BS) * F::TS;
    const uint32_t n_spans = (cols + MF_SPAN - 1) / MF_SPAN;
    const uint32_t tail_len = cols - (n_spans - 1) * (uint32_t)MF_SPAN;           // elements of the LAST span: 256, or a multiple of 64 below it (32-element blocks only)
    uint8_t* slice = smem16 + wave * G::SLICE;

    // the weight rows this lane decodes (clamped at the edge: the stores are masked instead): row nt*16 + r of the tile
    uint64_t wrow_off[NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        const uint32_t wr = n0 + (uint32_t)(nt * 16 + r);
        wrow_off[nt] = (uint64_t)(wr < n_rows ? wr : n_rows - 1) * row_bytes;
    }

    // copy of one span's packed bytes for the tile's rows: unit = (row, 16-byte piece); lane takes units lane, lane + 64, ...
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
            pf[u] = (ur < (uint32_t)G::ROWS && uu * 16u < a + span_bytes) ? gload16<true>(packed + (off - a) + uu * 16u) : u32x4{0, 0, 0, 0};
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
            if (ur < (uint32_t)G::ROWS) *reinterpret_cast<u32x4*>(slice + ur * G::ROW_STRIDE + uu * 16u) = pf[u];
        }
    };

    // x: which 8 elements of a group of 128 lane (r, c) holds at k-step t (both operands use the same map):
    //   XC = false: 32 c + 8 t -- the lane owns 32 CONSECUTIVE elements (one K-quant sub-block: its scale pair is decoded once per group) and reads
    //               them as 64 contiguous bytes; but one load instruction then takes 16 bytes out of 64 different 64-byte sectors: fine while the
    //               rows of x are few (<= 8: the lanes of the clamped rows share their sectors), a wall when they are 16 (+ 4 us at 12288 x 3072);
    //   XC = true:  32 t + 8 c -- the four lanes of a row read 64 CONTIGUOUS bytes per instruction (16 whole sectors per wave instruction); every
    //               k-step is then another sub-block, the same one for the whole wave: eight scale decodes per span and lane instead of two.
    const GGQ_GLOBAL uint8_t* xrow[MB];
#pragma unroll
    for (int mb = 0; mb < MB; mb++) {
        const uint32_t mr = m0 + (uint32_t)(mb * 16 + r);
        xrow[mb] = (GGQ_GLOBAL const uint8_t*)x_ + (uint64_t)(mr < m ? mr : m - 1) * cols * 2 + (uint32_t)(c * (XC ? 16 : 64));
    }

    f32x4 acc[NT][MB];
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int mb = 0; mb < MB; mb++) acc[nt][mb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

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
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const GGQ_GLOBAL uint8_t* src = xrow[mb] + kb + (uint32_t)(g * 256 + t * (XC ? 64 : 16));
                    if constexpr (F::BS == 32) {                                   // past the row's end (a short last span): zeros
                        const uint32_t k0 = span * (uint32_t)MF_SPAN + (uint32_t)(128 * g) + (XC ? (uint32_t)(32 * t) : (uint32_t)(32 * c));
                        if (k0 >= cols) src = (GGQ_GLOBAL const uint8_t*)MF16_ZEROS;
                    }
                    xf[mb][4 * g + t] = *(GGQ_GLOBAL const u32x4*)src;
                }
            }
        }
        stash(pf);
        wave_sync();
        if (span + kw < n_spans) fetch(span + kw, pf);                             // the next span's bytes fly while this one is decoded
#pragma unroll
        for (int g = 0; g < 2; g++) {
            if (g) __builtin_amdgcn_sched_barrier(0);                              // keep the second group's LDS reads and decode out of the first group's registers
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                const uint32_t a = G::SKEW ? ((uint32_t)(wrow_off[nt] + (uint64_t)span * G::SPAN_BYTES) & 15u) : 0u;
                const uint8_t* wspan = slice + (nt * 16 + r) * G::ROW_STRIDE + a;
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int j = 16 * g + (XC ? 4 * t + c : 4 * c + t);           // chunk of the span: elements 128 g + 32 c + 8 t .. + 7 (XC: 32 t + 8 c)
                    const Fields f = F::template fields<true>(wspan + (j / CPB) * F::TS, j % CPB);
                    uint32_t w[4];
                    weights8<F, OUT>(f, w);
                    const u32x4 wa{w[0], w[1], w[2], w[3]};
#pragma unroll
                    for (int mb = 0; mb < MB; mb++) acc[nt][mb] = mfma16<OUT>(wa, xf[mb][4 * g + t], acc[nt][mb]);
                }
            }
        }
        wave_sync();                                                               // the slice is rewritten at the top of the next span
    }

    // ---- sum the KW partials through LDS in wave order, then bias, cast, store.  Block b = (nt, mb) of the tile is finished by wave b % KW.
    constexpr int NB = NT * MB;
    __syncthreads();                                                               // every wave is done with its slice
    f32x4* red = reinterpret_cast<f32x4*>(smem16);
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int mb = 0; mb < MB; mb++) red[((uint32_t)wave * NB + (uint32_t)(nt * MB + mb)) * 64 + lane] = acc[nt][mb];
    __syncthreads();
    for (uint32_t b = (uint32_t)wave; b < (uint32_t)NB; b += kw) {
        const uint32_t nt = b / (uint32_t)MB, mb = b % (uint32_t)MB;
        f32x4 v = red[b * 64 + lane];
        for (uint32_t w = 1; w < kw; w++) {
            const f32x4 p = red[(w * NB + b) * 64 + lane];
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        // D[row = 4c + i][col = r] = y[x row r][W row 4c + i]
        const uint32_t mr = m0 + mb * 16u + (uint32_t)r, nc = n0 + nt * 16u + (uint32_t)(4 * c);     // row of y, first of this lane's four output columns
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
