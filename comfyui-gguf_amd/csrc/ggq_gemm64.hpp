// ggq_gemm64.hpp -- EXPERIMENT (compiled only with -DGGQ_WITH_TILE64; EXPERIMENTS.md A2b): the shared-tile fused GEMM of ggq_gemm.hpp with
// K-steps of 64 instead of 32, for the formats whose packed bytes can be staged PER K-STEP (Step64<F>): half as many barriers, LDS latency
// exposures and fragment-read restarts per MFMA.  Bit-identical weights, passes the same tests -- and measured 13 % SLOWER (Q4_K 12288x3072,
// 4608 rows: 465 vs 410 us; one tile alone 109 vs 94 us): with 32 MFMAs and a 222-instruction decode per wave between barriers, each wave's own
// serial chain (decode, then four fragment-read / 8-MFMA rounds) is what the step takes, and only two waves per SIMD are there to overlap it.
//
// Why it needs its own staging.  ggq_gemm.hpp decodes through the dequant kernels' generic `fields(block, chunk)`, which wants a whole
// 256-element super-block of every row in LDS: 36 KiB for Q4_K next to 4 x 16 KiB of operand tiles.  With K-steps of 64 the operand tiles
// are 4 x 32 KiB and that staging no longer fits (164 > 160 KiB).  But one K-step of 64 elements of a Q4_K row needs only 48 of the
// super-block's 144 bytes -- the 16-byte header and the 32 quant bytes of one sub-block pair -- and those are whole, 16-byte-aligned
// units of the GGUF layout: they go straight from global memory into a COMPACT staging row by LDS-DMA (12 KiB per K-step, double
// buffered).  The decode reads them with three ds_read_b128 and runs the very same k_scale_min / quad_f16 / `.to(dtype)` code on them
// (Step64<F>::decode32 builds the same `Fields` the generic path builds), so the weights stay the reference's values bit for bit
// (tests/test_gpu_mfma.py holds both kernels to the same fp64 / exact-arithmetic checks).
//
// Shape of the work: as ggq_gemm.hpp -- 256 x 256 output tile, 8 waves, ping-pong order of the two waves of a SIMD, LDS-transposed
// epilogue -- with: tile rows of 128 B (64 K-elements), 16-byte column XOR-swizzled by (row >> 1) & 7 (conflict-free for the
// ds_read_b128 fragment groups of gfx950 and, with the decode's row order 0,2,4,6,1,3,5,7 per 8 lanes, for its ds_write_b128);
// per K-step and wave 32 MFMAs (4 k-slices of 16) behind ONE barrier; a thread decodes one sub-block (32 weights) per K-step.
// LDS: X 2 x 32 + W 2 x 32 + staging 2 x 12 = 152 KiB.
#pragma once

#include "ggq_gemm.hpp"

namespace ggq {

constexpr int G64_BK = 64;
constexpr int G64_PITCH = G64_BK * 2;                    // bytes per tile row
constexpr int G64_TILE = 256 * G64_PITCH;                // one X or W tile: 32 KiB

GGQ_DEV uint32_t g64_swz(uint32_t row) { return (row >> 1) & 7u; }

// What one K-step of 64 elements needs from a row's packed bytes, as whole 16-byte units of the GGUF layout, and how to decode it.
template <class F> struct Step64 { static constexpr bool OK = false; };

// Q4_K (dequant.py:180-195): [d][dmin][scales 12] = unit 0; the 32 quant bytes of sub-block pair pp (elements 64 pp .. 64 pp + 63:
// low nibbles = sub-block 2 pp, high nibbles = sub-block 2 pp + 1) = units 1, 2.
template <> struct Step64<FmtQ4_K> {
    static constexpr bool OK = true;
    static constexpr int UNITS = 3, ROW = UNITS * 16;
    GGQ_DEV static uint32_t src(uint32_t u, uint32_t pp) { return u == 0u ? 0u : 32u * pp + 16u * u; }      // byte offset inside the super-block
    // the 32 weights of sub-block 2 pp + par (4 chunks of 8), as dequantize_tensor(.., dtype) would hold them
    template <int OUT>
    GGQ_DEV static void decode32(const uint8_t* row, uint32_t pp, uint32_t par, u32x4 (&w)[4])
    {
        const u32x4 hdr = *reinterpret_cast<const u32x4*>(row);
        const u32x4 q0 = *reinterpret_cast<const u32x4*>(row + 16), q1 = *reinterpret_cast<const u32x4*>(row + 32);
        Fields f;
        k_scale_min(hdr, (int)(2u * pp + par), f.sc, f.mn);                 // same header decode as FmtQ4_K::fields
        f.dm = hdr.x;
        const uint32_t sh = 4u * par;
        const uint32_t qw[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
        for (int c = 0; c < 4; c++) {
            f.t0 = (qw[2 * c] >> sh) & 0x0F0F0F0Fu;
            f.t1 = (qw[2 * c + 1] >> sh) & 0x0F0F0F0Fu;
            uint32_t o[4];
            weights8<FmtQ4_K, OUT>(f, o);
            w[c] = u32x4{o[0], o[1], o[2], o[3]};
        }
    }
};

template <class F> struct Gemm64Geom {
    using S = Step64<F>;
    static constexpr int UNITS = GT_BN * S::UNITS;                               // 16-byte units staged per K-step
    static constexpr int NUW = (UNITS + GT_THREADS - 1) / GT_THREADS;
    static constexpr int STAGING = GT_BN * S::ROW;
    static constexpr int LDS_BYTES = 4 * G64_TILE + 2 * STAGING;
};

template <class F, int OUT>
__global__ __launch_bounds__(GT_THREADS) void linear_tile64(const uint8_t* __restrict__ packed_, const uint8_t* __restrict__ x_,
                                                            const uint8_t* __restrict__ bias_, uint8_t* __restrict__ y_,
                                                            uint32_t m, uint32_t n_rows, uint32_t cols, uint32_t tiles_m, uint32_t tiles_n)
{
    using S = Step64<F>;
    using GG = Gemm64Geom<F>;
    static_assert(S::OK && F::BS == 256, "per-K-step staging is defined for this format");
    static_assert(OUT == OUT_F16 || OUT == OUT_BF16, "16-bit activations only");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* const xt = smem;                        // X[0], X[1]
    uint8_t* const wt = smem + 2 * G64_TILE;         // W[0], W[1]
    uint8_t* const stg = smem + 4 * G64_TILE;        // S[0], S[1]

    const uint32_t t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int)(t >> 6));
    const uint32_t lane = t & 63u;

    const uint32_t n_tiles = tiles_m * tiles_n;
    uint32_t tile = blockIdx.x;
    {
        const uint32_t q = n_tiles >> 3, r = n_tiles & 7u, xcd = tile & 7u, idx = tile >> 3;
        tile = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + idx;
    }
    const uint32_t tn = tile / tiles_m, tm = tile - tn * tiles_m;
    const uint32_t m0 = tm * GT_BM, n0 = tn * GT_BN;

    const gcptr packed = (gcptr)packed_;
    const uint64_t row_bytes = (uint64_t)(cols / F::BS) * F::TS;
    const uint32_t n_steps = cols / G64_BK;

    // ---- staging of K-step `step` into buffer `buf`: unit u = (row u / UNITS, piece u % UNITS), thread takes units t, t + 512, ...
    const GGQ_GLOBAL uint8_t* ssrc[GG::NUW];
    uint32_t spiece[GG::NUW];
    bool sok[GG::NUW];
#pragma unroll
    for (int k = 0; k < GG::NUW; k++) {
        const uint32_t unit = t + (uint32_t)(GT_THREADS * k), ur = unit / (uint32_t)S::UNITS;
        spiece[k] = unit - ur * (uint32_t)S::UNITS;
        sok[k] = unit < (uint32_t)GG::UNITS;
        const uint32_t rr = (n0 + ur < n_rows) ? n0 + ur : n_rows - 1;
        ssrc[k] = packed + (uint64_t)rr * row_bytes;
    }
    auto sdma = [&](uint32_t step, uint32_t buf) {
        const uint32_t span = step >> 2, pp = step & 3u;
#pragma unroll
        for (int k = 0; k < GG::NUW; k++)
            if (sok[k]) dma16(ssrc[k] + (uint64_t)span * F::TS + S::src(spiece[k], pp), stg + buf * (uint32_t)GG::STAGING + ((uint32_t)wave * 64u + (uint32_t)(GT_THREADS * k)) * 16u);
    };

    // ---- x tile by LDS-DMA: unit u = (row u / 8, piece u % 8), four per thread; the XOR swizzle goes on the SOURCE piece
    const GGQ_GLOBAL uint8_t* xsrc[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t unit = t + (uint32_t)(GT_THREADS * i), row = unit >> 3, pc = unit & 7u, mr = m0 + row;
        xsrc[i] = (GGQ_GLOBAL const uint8_t*)x_ + (uint64_t)(mr < m ? mr : m - 1) * cols * 2 + ((pc ^ g64_swz(row)) * 16u);
    }
    auto xdma = [&](uint32_t step, uint8_t* xd) {
#pragma unroll
        for (int i = 0; i < 4; i++) dma16(xsrc[i] + (uint64_t)step * G64_PITCH, xd + ((uint32_t)wave * 64u + (uint32_t)(GT_THREADS * i)) * 16u);
    };

    // ---- weight decode: lane pair i = t / 2 -> row (8 consecutive lanes take rows r, r+2, r+4, r+6: four different swizzles, so their two
    // 16-byte pieces each land in eight different LDS slots), t & 1 -> which sub-block of the pair
    const uint32_t di = t >> 1, par = t & 1u;
    const uint32_t drow = 8u * (di >> 3) + 2u * (di & 3u) + ((di >> 2) & 1u), dswz = g64_swz(drow);
    auto decode = [&](uint32_t step, uint8_t* wdst) {
        u32x4 w[4];
        S::template decode32<OUT>(stg + (step & 1u) * (uint32_t)GG::STAGING + drow * (uint32_t)S::ROW, step & 3u, par, w);
#pragma unroll
        for (int c = 0; c < 4; c++) *reinterpret_cast<u32x4*>(wdst + drow * G64_PITCH + (((4u * par + (uint32_t)c) ^ dswz) * 16u)) = w[c];
    };

    // ---- MFMA roles as in ggq_gemm.hpp: wave -> (wm = wave / 4: rows of x [128 wm, +128), wn = wave % 4: output columns [64 wn, +64))
    const uint32_t wm = (uint32_t)wave >> 2, wn = (uint32_t)wave & 3u;
    const uint32_t r32 = lane & 31u, hk = lane >> 5, fswz = g64_swz(r32);
    f32x16 acc[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
        for (int mt = 0; mt < 4; mt++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[nt][mt][i] = 0.0f;

    auto mma = [&](const uint8_t* xs, const uint8_t* ws) {
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const uint32_t col = (((uint32_t)(2 * kk) + hk) ^ fswz) * 16u;
            u32x4 wa[2], xb[4];
#pragma unroll
            for (int nt = 0; nt < 2; nt++) wa[nt] = *reinterpret_cast<const u32x4*>(ws + (64u * wn + 32u * (uint32_t)nt + r32) * G64_PITCH + col);
#pragma unroll
            for (int mt = 0; mt < 4; mt++) xb[mt] = *reinterpret_cast<const u32x4*>(xs + (128u * wm + 32u * (uint32_t)mt + r32) * G64_PITCH + col);
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int mt = 0; mt < 4; mt++) acc[nt][mt] = mfma32<OUT>(wa[nt], xb[mt], acc[nt][mt]);
        }
    };

    auto dma_fence = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every LDS-DMA of this wave has landed; the barrier then publishes it
        __syncthreads();
    };

    // ---- prologue
    sdma(0u, 0u);
    xdma(0u, xt);
    dma_fence();
    sdma(n_steps > 1 ? 1u : 0u, 1u);
    decode(0u, wt);
    dma_fence();

    // ---- main loop: K-step t = [DMA: staging of t + 2, x tile of t + 1] + [decode weight tile t + 1] + [32 MFMAs on tile t] + one barrier;
    // the two waves of a SIMD (w, w + 4) run decode and MFMAs in opposite order (see ggq_gemm.hpp), each order its own copy of the loop
    auto kstep = [&](uint32_t step, auto parity_tag, auto pong_tag, auto decode_tag) {
        constexpr int P = decltype(parity_tag)::value;
        constexpr bool PONG = decltype(pong_tag)::value, DECODE = decltype(decode_tag)::value;
        uint8_t* const xcur = xt + P * G64_TILE;
        uint8_t* const wcur = wt + P * G64_TILE;
        uint8_t* const xnxt = xt + (P ^ 1) * G64_TILE;
        uint8_t* const wnxt = wt + (P ^ 1) * G64_TILE;
        if constexpr (DECODE) {
            // (indices clamped at the end: a harmless re-read into a buffer nobody reads any more)
            sdma(step + 2 < n_steps ? step + 2 : n_steps - 1, (uint32_t)P);          // S[P] held step `step`: decoded before the previous barrier
            xdma(step + 1, xnxt);
        }
        if constexpr (!DECODE) {
            mma(xcur, wcur);
        } else if constexpr (PONG) {
            mma(xcur, wcur);
            __builtin_amdgcn_sched_barrier(GGQ_GT_CROSS);
            decode(step + 1, wnxt);
        } else {
            decode(step + 1, wnxt);
            __builtin_amdgcn_sched_barrier(GGQ_GT_CROSS);
            mma(xcur, wcur);
        }
        dma_fence();
    };
    auto main_loop = [&](auto pong_tag) {
        using T0 = std::integral_constant<int, 0>;
        using T1 = std::integral_constant<int, 1>;
        uint32_t step = 0;
        for (; step + 2 < n_steps; step += 2) {                     // n_steps is a multiple of 4
            kstep(step, T0{}, pong_tag, std::true_type{});
            kstep(step + 1, T1{}, pong_tag, std::true_type{});
        }
        kstep(step, T0{}, pong_tag, std::true_type{});
        kstep(step + 1, T1{}, pong_tag, std::false_type{});
    };
    if (wave >= 4) main_loop(std::true_type{});
    else main_loop(std::false_type{});

    // ---- epilogue (identical to ggq_gemm.hpp): bias, cast, transpose through wave-private LDS, full-line stores
    uint8_t* const ep = smem + wave * 8192;
    const uint32_t nbase = n0 + 64u * wn, mbase = m0 + 128u * wm;
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
    for (int rd = 0; rd < 2; rd++) {
#pragma unroll
        for (int mh = 0; mh < 2; mh++) {
            const int mt = 2 * rd + mh;
            const uint32_t ml = 32u * (uint32_t)mh + r32;
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float v0 = acc[nt][mt][4 * q + 0] + bias[nt][q][0], v1 = acc[nt][mt][4 * q + 1] + bias[nt][q][1];
                    const float v2 = acc[nt][mt][4 * q + 2] + bias[nt][q][2], v3 = acc[nt][mt][4 * q + 3] + bias[nt][q][3];
                    u32x2 o;
                    if constexpr (OUT == OUT_F16) o = u32x2{pack_f16(v0, v1), pack_f16(v2, v3)};
                    else o = u32x2{pack_bf16(v0, v1), pack_bf16(v2, v3)};
                    const uint32_t p = 4u * (uint32_t)nt + (uint32_t)q;
                    *reinterpret_cast<u32x2*>(ep + ml * 128u + ((p ^ (ml & 7u)) * 16u) + 8u * hk) = o;
                }
        }
        wave_sync();
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t row = (lane >> 3) + 8u * (uint32_t)i, p = lane & 7u;
            const u32x4 v = *reinterpret_cast<const u32x4*>(ep + row * 128u + ((p ^ (row & 7u)) * 16u));
            const uint32_t mr = mbase + 64u * (uint32_t)rd + row, nc = nbase + 8u * p;
            if (mr < m && nc < n_rows) gstore<false>((gptr)y_ + ((uint64_t)mr * n_rows + nc) * 2, v);
        }
        wave_sync();
    }
}

}  // namespace ggq
