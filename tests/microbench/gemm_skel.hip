// gemm_skel.hip -- LAB harness (test infrastructure, not product): how fast can the LDS-tiled MFMA SKELETON of the shared-tile fused GEMM
// (csrc/ggq_gemm.hpp) go once the decode is taken out?  VERDICT round 3, Next #2: "fix the skeleton first, with a stop rule" -- the no-decode
// skeleton must reach >= 1.35 PFLOP/s at 12288 x 3072 weights x 4608 rows before the decode is put back, else the kernel is frozen.
//
// The skeleton here is a REAL dense GEMM, y[m][n] = sum_k x[m][k] * w[n][k] (bf16 in, fp32 accumulate, bf16 out: the weight tile arrives the way
// the x tile does instead of being decoded), so every variant is checked against a CPU evaluation before it is timed.  One workgroup per
// 256 x 256 output tile, XCD-contiguous tile order, operands staged in LDS with a 16-byte-column XOR swizzle, v_mfma_f32_32x32x16_bf16.
// Template knobs = the geometries the verdict names:
//     WM x WN waves (rows of x  x  output columns): 4x4 = 16 waves of 64 x 64 (the shipped shape), 2x4 = 8 waves of 128 x 64, 2x2 = 4 waves of
//     128 x 128 (256 accumulator registers, one wave per SIMD, as hipBLASLt's MT256x256x64); BK = 32 or 64; operands by LDS-DMA (global_load_lds_dwordx4)
//     into a ring of STAGES tiles; tile order inside an XCD; and two timing-only modes (no loads; a thinned weight stream).  The first run of
//     this harness (profiles/r04_gemm_skeleton_first_sweep.jsonl) also had loads through registers (8-12 % slower than LDS-DMA) and hand-written
//     fragment double buffering (no effect: hipcc already hoists the reads).
// Also timed: an MFMA-ONLY loop with the same grid (no LDS, no loads, no barriers): the ceiling the grid itself allows (tile quantisation of
// 864 tiles on 256 CUs + the clock under matrix load) -- once with operands that barely toggle the multipliers and once with random normal values,
// because the clock the power management grants depends on the switching activity; every kernel reports the shader clock it actually got
// (s_memtime / s_memrealtime around workgroup 0's K loop) and the fraction of those cycles its matrix pipes were busy.
//
//     ./gemm_skel [m=4608] [n=12288] [k=3072] [reps=20]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define DEV __device__ __forceinline__
#define GLOBAL __attribute__((address_space(1)))

DEV f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0); }
DEV void dma16(const GLOBAL uint8_t* src, uint8_t* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const GLOBAL void*)src, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
DEV uint32_t pack_bf16(float a, float b)
{
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{a, b}, bf2));
}

// shader clock actually delivered: s_memtime counts shader cycles, s_memrealtime a constant 100 MHz -- workgroup 0 brackets its K loop with both
__device__ uint64_t g_clk[4];
DEV void clk_mark(int slot)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        g_clk[slot] = __builtin_readcyclecounter();
        g_clk[slot + 1] = __builtin_amdgcn_s_memrealtime();
    }
}

// 16-byte column swizzle: rows of 64 B (BK = 32: 4 columns) and of 128 B (BK = 64: 8 columns); conflict-free for ds_read_b128 with one row per lane
// in gfx950's 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md, LDS) and for the lane-linear DMA image
template <int BK> DEV uint32_t swz(uint32_t row)
{
    if constexpr (BK == 32) return ((row >> 3) & 3u) ^ ((row >> 1) & 1u);
    else return (row >> 1) & 7u;
}

// ORDER: how an XCD's contiguous eighth of the tile list walks the tile grid -- 0 = column-major (tm fastest: the shipped kernel: concurrent tiles share
// weight panels), 1 = row-major (tn fastest: concurrent tiles share x panels), 2 = blocks of 4 (m) x 8 (n) tiles.
// STAGES: LDS ring depth; the tile of step t + STAGES - 1 is requested at step t (counted s_waitcnt vmcnt, raw s_barrier).
// PREF = 1 (needs STAGES >= 3): the ring runs one stage further ahead, so that the tile of step t + 1 is already visible to every wave DURING step t --
// a wave then reads the first k-slice's fragments of step t + 1 before the barrier that ends step t, and the matrix pipe does not drain at the barrier.
// NOLOAD (timing only, wrong results): no loads inside the K loop.  WDIV (timing only): the W tile is fetched only every WDIV-th step -- what the
// fused kernel's packed weight stream costs (Q4_K: 4.5 bits per weight = 1 / 3.56 of the dense bytes).
template <int WM, int WN, int BK, int STAGES, int ORDER, int NOLOAD, int WDIV, int PREF>
__global__ __launch_bounds__(WM * WN * 64) void skel(const uint8_t* __restrict__ x_, const uint8_t* __restrict__ w_, uint8_t* __restrict__ y_,
                                                     uint32_t m, uint32_t n, uint32_t k, uint32_t tiles_m, uint32_t tiles_n)
{
    constexpr int THREADS = WM * WN * 64;
    constexpr int MT = 256 / WM / 32, NT = 256 / WN / 32;          // 32 x 32 MFMA tiles per wave
    constexpr int PITCH = BK * 2, CPR = PITCH / 16;                // bytes per tile row, 16-byte columns per row
    constexpr int TILE = 256 * PITCH;
    constexpr int UNITS = 256 * CPR / THREADS;                     // 16-byte units per thread per operand per K-step
    constexpr int KS = BK / 16;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* const xt = smem;                                       // X[STAGES]
    uint8_t* const wt = smem + STAGES * TILE;                       // W[STAGES]
    const uint32_t t = threadIdx.x, lane = t & 63u;
    const int wave = __builtin_amdgcn_readfirstlane((int)(t >> 6));

    uint32_t tile = blockIdx.x;
    {
        const uint32_t n_tiles = tiles_m * tiles_n, q = n_tiles >> 3, r = n_tiles & 7u, xcd = tile & 7u, idx = tile >> 3;
        tile = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + idx;
    }
    uint32_t tn, tm;
    if constexpr (ORDER == 0) { tn = tile / tiles_m; tm = tile - tn * tiles_m; }
    else if constexpr (ORDER == 1) { tm = tile / tiles_n; tn = tile - tm * tiles_n; }
    else {
        // blocks of 4 x 8 tiles, blocks row-major, tiles inside a block n-fastest; ragged edges fall back to row-major order of the remainder
        const uint32_t bm = tiles_m / 4u, bn = tiles_n / 8u, inblk = bm * bn * 32u;
        if (tile < inblk) {
            const uint32_t blk = tile >> 5, in = tile & 31u, bi = blk / bn, bj = blk - bi * bn;
            tm = bi * 4u + (in >> 3);
            tn = bj * 8u + (in & 7u);
        } else {
            // the tiles outside the blocked region, enumerated row-major: rows [0, 4 bm) x columns [8 bn, tiles_n), then rows [4 bm, tiles_m) x all columns
            uint32_t rest = tile - inblk;
            const uint32_t right = 4u * bm * (tiles_n - 8u * bn);
            if (rest < right) { const uint32_t wdt = tiles_n - 8u * bn; tm = rest / wdt; tn = 8u * bn + (rest - tm * wdt); }
            else { rest -= right; tm = 4u * bm + rest / tiles_n; tn = rest % tiles_n; }
        }
    }
    const uint32_t m0 = tm * 256u, n0 = tn * 256u;
    const uint32_t n_steps = k / BK;

    // unit u = t + THREADS i -> (row u / CPR, column u % CPR); LDS image lane-linear (DMA), so the swizzle goes on the SOURCE column
    const GLOBAL uint8_t* xsrc[UNITS];
    const GLOBAL uint8_t* wsrc[UNITS];
#pragma unroll
    for (int i = 0; i < UNITS; i++) {
        const uint32_t u = t + (uint32_t)(THREADS * i), row = u / CPR, col = u % CPR;
        const uint32_t mr = m0 + row < m ? m0 + row : m - 1, nr = n0 + row < n ? n0 + row : n - 1;
        xsrc[i] = (const GLOBAL uint8_t*)x_ + (uint64_t)mr * k * 2 + ((col ^ swz<BK>(row)) * 16u);
        wsrc[i] = (const GLOBAL uint8_t*)w_ + (uint64_t)nr * k * 2 + ((col ^ swz<BK>(row)) * 16u);
    }
    auto fetch = [&](uint32_t step, uint32_t buf) {
#pragma unroll
        for (int i = 0; i < UNITS; i++) {
            dma16(xsrc[i] + (uint64_t)step * PITCH, xt + buf * TILE + ((uint32_t)wave * 64u + (uint32_t)(THREADS * i)) * 16u);
            // (WDIV > 1: the skipped W fetches are replaced by nothing; the vmcnt bookkeeping below counts x only then)
            if (WDIV == 1) dma16(wsrc[i] + (uint64_t)step * PITCH, wt + buf * TILE + ((uint32_t)wave * 64u + (uint32_t)(THREADS * i)) * 16u);
        }
    };
    auto fetch_w = [&](uint32_t step, uint32_t buf) {
#pragma unroll
        for (int i = 0; i < UNITS; i++) dma16(wsrc[i] + (uint64_t)step * PITCH, wt + buf * TILE + ((uint32_t)wave * 64u + (uint32_t)(THREADS * i)) * 16u);
    };

    const uint32_t wm = (uint32_t)wave / WN, wn = (uint32_t)wave % WN;
    const uint32_t r32 = lane & 31u, hk = lane >> 5, fs = swz<BK>(r32);          // rows 32 apart swizzle alike (both formulas are periodic in 32)
    f32x16 acc[NT][MT];
#pragma unroll
    for (int a = 0; a < NT; a++)
#pragma unroll
        for (int b = 0; b < MT; b++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[a][b][i] = 0.0f;

    auto frags = [&](const uint8_t* xs, const uint8_t* ws, int kk, u32x4* wa, u32x4* xb) {
        const uint32_t col = (((uint32_t)(2 * kk) + hk) ^ fs) * 16u;
#pragma unroll
        for (int a = 0; a < NT; a++) wa[a] = *reinterpret_cast<const u32x4*>(ws + ((uint32_t)(32 * NT) * wn + 32u * (uint32_t)a + r32) * PITCH + col);
#pragma unroll
        for (int b = 0; b < MT; b++) xb[b] = *reinterpret_cast<const u32x4*>(xs + ((uint32_t)(32 * MT) * wm + 32u * (uint32_t)b + r32) * PITCH + col);
    };
    u32x4 pwa[NT], pxb[MT];                                         // PREF: slice 0 of the next step
    auto compute = [&](const uint8_t* xs, const uint8_t* ws, const uint8_t* xn, const uint8_t* wnx) {
#pragma unroll
        for (int kk = 0; kk < KS; kk++) {
            u32x4 wa[NT], xb[MT];
            if (PREF && kk == 0) {
#pragma unroll
                for (int a = 0; a < NT; a++) wa[a] = pwa[a];
#pragma unroll
                for (int b = 0; b < MT; b++) xb[b] = pxb[b];
            } else {
                frags(xs, ws, kk, wa, xb);
            }
            if (PREF && kk == KS - 1) frags(xn, wnx, 0, pwa, pxb);  // the next step's first slice, requested before this slice's MFMAs and the barrier
#pragma unroll
            for (int a = 0; a < NT; a++)
#pragma unroll
                for (int b = 0; b < MT; b++) acc[a][b] = mfma(wa[a], xb[b], acc[a][b]);
        }
    };

    // prologue: tiles 0 .. STAGES-2 requested, tile 0 landed
    constexpr int PER_STAGE = UNITS * (WDIV == 1 ? 2 : 1);         // DMA instructions per thread per fetched stage
#pragma unroll
    for (int s = 0; s < STAGES - 1; s++) fetch((uint32_t)s < n_steps ? (uint32_t)s : n_steps - 1, (uint32_t)s);
    if (WDIV != 1) { fetch_w(0u, 0u); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    // all but the newest STAGES - 2 (PREF: STAGES - 3) stages have landed
    static_assert(!PREF || (STAGES >= 3 && WDIV == 1 && !NOLOAD), "PREF needs a ring of 3+");
    auto ring_wait = [&]() {
        constexpr int N = (STAGES - 2 - PREF) * PER_STAGE;
        static_assert(N <= 63, "vmcnt is 6 bits");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    ring_wait();
    clk_mark(0);
    if (PREF) frags(xt, wt, 0, pwa, pxb);
    uint32_t cur = 0, nxt = STAGES - 1;                             // ring positions of step and step + STAGES - 1
    for (uint32_t step = 0; step < n_steps; step++) {
        if (!NOLOAD) {
            const uint32_t ahead = step + (uint32_t)(STAGES - 1);
            fetch(ahead < n_steps ? ahead : n_steps - 1, nxt);      // the buffer read at step - 1 (clamped at the end: a harmless re-read)
            if (WDIV != 1 && step % WDIV == 0) { /* timing only: one W fetch per WDIV steps, waited for at once (it is rare) */
                fetch_w(step, cur);
            }
        }
        const uint32_t c1 = cur + 1 == STAGES ? 0 : cur + 1;
        compute(xt + cur * TILE, wt + (WDIV == 1 ? cur : 0u) * TILE, xt + c1 * TILE, wt + c1 * TILE);
        if (NOLOAD) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
        else if (WDIV != 1 && step % WDIV == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
        else ring_wait();
        cur = cur + 1 == STAGES ? 0 : cur + 1;
        nxt = nxt + 1 == STAGES ? 0 : nxt + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    clk_mark(2);

    // epilogue (plain: 8-byte pieces straight from the accumulator layout; register i of lane l = D[n = (i & 3) + 8 (i >> 2) + 4 (l >> 5)][m = l & 31])
#pragma unroll
    for (int a = 0; a < NT; a++)
#pragma unroll
        for (int b = 0; b < MT; b++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t nn = n0 + (uint32_t)(32 * NT) * wn + 32u * (uint32_t)a + 8u * (uint32_t)q + 4u * hk;
                const uint32_t mm = m0 + (uint32_t)(32 * MT) * wm + 32u * (uint32_t)b + r32;
                if (mm < m && nn < n)
                    *reinterpret_cast<u32x2*>(y_ + ((uint64_t)mm * n + nn) * 2) =
                        u32x2{pack_bf16(acc[a][b][4 * q], acc[a][b][4 * q + 1]), pack_bf16(acc[a][b][4 * q + 2], acc[a][b][4 * q + 3])};
            }
}

// the grid's own ceiling: the same 864 workgroups, the same number of MFMAs per wave, nothing else
template <int WM, int WN, int RANDOM>
__global__ __launch_bounds__(WM * WN * 64) void mfma_only(uint8_t* __restrict__ y_, uint32_t m, uint32_t n, uint32_t k, uint32_t tiles_m)
{
    constexpr int MT = 256 / WM / 32, NT = 256 / WN / 32;
    const uint32_t lane = threadIdx.x & 63u;
    f32x16 acc[NT][MT];
#pragma unroll
    for (int a = 0; a < NT; a++)
#pragma unroll
        for (int b = 0; b < MT; b++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[a][b][i] = 0.0f;
    u32x4 wa[NT], xb[MT];
    // RANDOM = 0: operands that are mostly zeros / denormals (the multipliers barely toggle); 1: random NORMAL bf16 values in [0.5, 2) with random
    // signs, different in every lane and register -- the switching activity of real data, which is what the power management sees
    auto rnd = [&](uint32_t i) {
        uint32_t h = (lane * 0x9E3779B1u) ^ (i * 0x85EBCA77u) ^ (threadIdx.x >> 6) * 0xC2B2AE3Du;
        h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
        return RANDOM ? ((h & 0x80FF80FFu) | 0x3F003F00u) : (i & 1u ? 0x3c003c00u : (lane + i));
    };
#pragma unroll
    for (int a = 0; a < NT; a++) wa[a] = u32x4{rnd(8u * a), rnd(8u * a + 1), rnd(8u * a + 2), rnd(8u * a + 3)};
#pragma unroll
    for (int b = 0; b < MT; b++) xb[b] = u32x4{rnd(8u * b + 4), rnd(8u * b + 5), rnd(8u * b + 6), rnd(8u * b + 7)};
    clk_mark(0);
    for (uint32_t s = 0; s < k / 16; s++) {
#pragma unroll
        for (int a = 0; a < NT; a++)
#pragma unroll
            for (int b = 0; b < MT; b++) acc[a][b] = mfma(wa[a], xb[b], acc[a][b]);
        asm volatile("" ::: "memory");
    }
    clk_mark(2);
    float sum = 0.0f;
#pragma unroll
    for (int a = 0; a < NT; a++)
#pragma unroll
        for (int b = 0; b < MT; b++)
#pragma unroll
            for (int i = 0; i < 16; i++) sum += acc[a][b][i];
    if (sum == 123.456f) y_[blockIdx.x] = 1;                        // keeps the loop alive
    (void)m; (void)n; (void)tiles_m;
}

static uint16_t f2bf(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h)
{
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

struct Problem {
    uint32_t m, n, k;
    uint8_t *x, *w, *y;
    std::vector<uint16_t> hx, hw;
    int reps;
};

template <class Launch>
static double time_us(Launch&& launch, int reps)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) launch();
    CK(hipDeviceSynchronize());
    double best = 1e30, sum = 0;
    for (int r = 0; r < 3; r++) {
        CK(hipEventRecord(a));
        for (int i = 0; i < reps; i++) launch();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        const double us = ms * 1e3 / reps;
        best = us < best ? us : best;
        sum += us;
    }
    (void)sum;
    return best;
}

// MHz of the shader clock over workgroup 0's K loop in the LAST launch (after `reps` back-to-back launches: the steady state)
static double shader_mhz()
{
    uint64_t c[4];
    CK(hipMemcpyFromSymbol(c, HIP_SYMBOL(g_clk), sizeof(c)));
    const double cycles = (double)(c[2] - c[0]), us = (double)(c[3] - c[1]) / 100.0;
    return us > 0 ? cycles / us : 0.0;
}

static int check(const Problem& p, const char* name)
{
    std::vector<uint16_t> hy((size_t)p.m * p.n);
    CK(hipMemcpy(hy.data(), p.y, hy.size() * 2, hipMemcpyDeviceToHost));
    int bad = 0;
    uint64_t s = 88172645463325252ull;
    for (int i = 0; i < 4000; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        uint32_t mm = (uint32_t)(s % p.m), nn = (uint32_t)((s >> 32) % p.n);
        if (i < 8) { mm = (i & 1) ? p.m - 1 : 0; nn = (i & 2) ? p.n - 1 : 0; if (i & 4) { mm = p.m / 2 + 131; nn = p.n / 2 + 77; } }
        double ref = 0, mag = 0;
        for (uint32_t kk = 0; kk < p.k; kk++) {
            const double a = bf2f(p.hx[(size_t)mm * p.k + kk]), b = bf2f(p.hw[(size_t)nn * p.k + kk]);
            ref += a * b;
            mag += fabs(a * b);
        }
        const double got = bf2f(hy[(size_t)mm * p.n + nn]);
        if (fabs(got - ref) > 1e-3 * mag + fabs(ref) / 128.0 + 1e-6) {
            if (bad < 3) fprintf(stderr, "%s: y[%u][%u] = %g, expected %g\n", name, mm, nn, got, ref);
            bad++;
        }
    }
    return bad;
}

template <int WM, int WN, int BK, int STAGES, int ORDER, int NOLOAD = 0, int WDIV = 1, int PREF = 0>
static void run(const Problem& p, const char* name)
{
    constexpr int LDS = 2 * STAGES * 256 * BK * 2;
    static_assert(LDS <= 160 * 1024, "LDS");
    const uint32_t tiles_m = (p.m + 255) / 256, tiles_n = (p.n + 255) / 256;
    auto* fn = &skel<WM, WN, BK, STAGES, ORDER, NOLOAD, WDIV, PREF>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipMemset(p.y, 0xFF, (size_t)p.m * p.n * 2));
    auto launch = [&]() { hipLaunchKernelGGL(fn, dim3(tiles_m * tiles_n), dim3(WM * WN * 64), LDS, 0, p.x, p.w, p.y, p.m, p.n, p.k, tiles_m, tiles_n); };
    launch();
    CK(hipDeviceSynchronize());
    const bool exact = !NOLOAD && WDIV == 1;
    const int bad = exact ? check(p, name) : 0;
    const double us = time_us(launch, p.reps);
    const double mhz = shader_mhz();
    uint64_t c[4];
    CK(hipMemcpyFromSymbol(c, HIP_SYMBOL(g_clk), sizeof(c)));
    const double busy = (double)(p.k / 16) * 16.0 * 32.0 / (double)(c[2] - c[0]);
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(fn)));
    static const char* const orders[] = {"column-major (tm fastest)", "row-major (tn fastest)", "4x8 blocks"};
    printf("{\"variant\": \"%s\", \"waves\": %d, \"wave_tile\": \"%dx%d\", \"BK\": %d, \"stages\": %d, \"tile_order\": \"%s\", \"no_loads\": %d, \"w_fetch_every\": %d, \"fragments_prefetched_across_barrier\": %d, "
           "\"us\": %.1f, \"PFLOPs\": %.3f, \"shader_MHz\": %.0f, \"matrix_pipe_busy\": %.3f, \"vgprs\": %d, \"lds_bytes\": %d, \"check\": \"%s\"}\n",
           name, WM * WN, 256 / WM, 256 / WN, BK, STAGES, orders[ORDER], NOLOAD, WDIV, PREF, us, 2.0 * p.m * p.n * p.k / us / 1e9, mhz,
           /* fraction of workgroup 0's K-loop cycles its SIMDs' matrix pipes were busy: (k/16 slices x 16 MFMAs per SIMD x 32 cycles) / measured cycles */ busy, fa.numRegs, LDS,
           !exact ? "timing only (results wrong by construction)" : (bad ? "MISMATCH" : "ok (4000 sampled outputs vs fp64)"));
    fflush(stdout);
}

template <int WM, int WN, int RANDOM>
static void run_ceiling(const Problem& p)
{
    const uint32_t tiles_m = (p.m + 255) / 256, tiles_n = (p.n + 255) / 256;
    auto launch = [&]() { hipLaunchKernelGGL((mfma_only<WM, WN, RANDOM>), dim3(tiles_m * tiles_n), dim3(WM * WN * 64), 0, 0, p.y, p.m, p.n, p.k, tiles_m); };
    const double us = time_us(launch, p.reps);
    const double mhz = shader_mhz();
    printf("{\"variant\": \"mfma_only_%dx%d_%s\", \"shader_MHz\": %.0f, \"waves\": %d, \"what\": \"same grid and MFMA count, no LDS / loads / barriers: the ceiling of the grid (tile rounds + clock under matrix load)\", "
           "\"us\": %.1f, \"PFLOPs\": %.3f, \"tiles\": %u, \"rounds_on_256_CUs\": %.3f}\n",
           WM, WN, RANDOM ? "random_operands" : "near_zero_operands", mhz, WM * WN, us, 2.0 * p.m * p.n * p.k / us / 1e9, tiles_m * tiles_n, tiles_m * tiles_n / 256.0);
    fflush(stdout);
}

int main(int argc, char** argv)
{
    Problem p;
    p.m = argc > 1 ? (uint32_t)atoi(argv[1]) : 4608;
    p.n = argc > 2 ? (uint32_t)atoi(argv[2]) : 12288;
    p.k = argc > 3 ? (uint32_t)atoi(argv[3]) : 3072;
    p.reps = argc > 4 ? atoi(argv[4]) : 20;
    if (p.k % 64 || p.n % 8) { fprintf(stderr, "k %% 64 == 0 and n %% 8 == 0 required\n"); return 2; }
    p.hx.resize((size_t)p.m * p.k);
    p.hw.resize((size_t)p.n * p.k);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((double)(s >> 11) / 9007199254740992.0 - 0.5); };
    for (auto& v : p.hx) v = f2bf(rnd());
    for (auto& v : p.hw) v = f2bf(rnd() * 0.1f);
    CK(hipMalloc(&p.x, p.hx.size() * 2));
    CK(hipMalloc(&p.w, p.hw.size() * 2));
    CK(hipMalloc(&p.y, (size_t)p.m * p.n * 2));
    CK(hipMemcpy(p.x, p.hx.data(), p.hx.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(p.w, p.hw.data(), p.hw.size() * 2, hipMemcpyHostToDevice));
    printf("{\"problem\": \"y[%u x %u] = x[%u x %u] @ w[%u x %u]^T, bf16, %.2f GFLOP\", \"reps\": %d}\n", p.m, p.n, p.m, p.k, p.n, p.k, 2.0 * p.m * p.n * p.k / 1e9, p.reps);
    run_ceiling<4, 4, 0>(p);
    run_ceiling<2, 2, 0>(p);
    run_ceiling<4, 4, 1>(p);
    run_ceiling<2, 2, 1>(p);
    //  WM WN BK STAGES ORDER [NOLOAD WDIV]
    run<4, 4, 32, 2, 0>(p, "16w_bk32_s2_col");                // the geometry of the shipped kernel (its x ring is 3 deep)
    run<4, 4, 32, 3, 0>(p, "16w_bk32_s3_col");
    run<4, 4, 32, 4, 0>(p, "16w_bk32_s4_col");
    run<4, 4, 32, 4, 1>(p, "16w_bk32_s4_row");
    run<4, 4, 32, 4, 2>(p, "16w_bk32_s4_blk");
    run<4, 4, 64, 2, 0>(p, "16w_bk64_s2_col");
    run<4, 4, 64, 2, 1>(p, "16w_bk64_s2_row");
    run<4, 4, 64, 2, 2>(p, "16w_bk64_s2_blk");
    run<2, 4, 32, 4, 2>(p, "8w_128x64_bk32_s4_blk");
    run<2, 4, 64, 2, 2>(p, "8w_128x64_bk64_s2_blk");
    run<2, 2, 32, 4, 2>(p, "4w_128x128_bk32_s4_blk");
    run<2, 2, 64, 2, 2>(p, "4w_128x128_bk64_s2_blk");
    // fragments of the next step's first slice read BEFORE the barrier (ring one stage further ahead)
    run<4, 4, 32, 4, 2, 0, 1, 1>(p, "16w_bk32_s4_blk_pref");
    run<4, 4, 32, 3, 2, 0, 1, 1>(p, "16w_bk32_s3_blk_pref");
    run<2, 4, 32, 4, 2, 0, 1, 1>(p, "8w_128x64_bk32_s4_blk_pref");
    run<2, 2, 32, 4, 2, 0, 1, 1>(p, "4w_128x128_bk32_s4_blk_pref");
    // no loads at all inside the K loop: fragment reads + MFMAs + one barrier per step -- what each geometry could do with free operands
    run<4, 4, 32, 2, 0, 1>(p, "16w_bk32_noload");
    run<4, 4, 64, 2, 0, 1>(p, "16w_bk64_noload");
    run<2, 4, 32, 2, 0, 1>(p, "8w_128x64_bk32_noload");
    run<2, 4, 64, 2, 0, 1>(p, "8w_128x64_bk64_noload");
    run<2, 2, 32, 2, 0, 1>(p, "4w_128x128_bk32_noload");
    run<2, 2, 64, 2, 0, 1>(p, "4w_128x128_bk64_noload");
    // the fused kernel's traffic: x dense, weights as a packed stream of 1 / 3.56 the bytes (here: one dense W tile per 4 steps)
    run<4, 4, 32, 4, 0, 0, 4>(p, "16w_bk32_s4_col_wdiv4");
    run<4, 4, 32, 4, 1, 0, 4>(p, "16w_bk32_s4_row_wdiv4");
    run<4, 4, 32, 4, 2, 0, 4>(p, "16w_bk32_s4_blk_wdiv4");
    run<4, 4, 64, 2, 1, 0, 4>(p, "16w_bk64_s2_row_wdiv4");
    return 0;
}
