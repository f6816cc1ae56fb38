// ggq_linear.hip -- C ABI over ggq_linear.hpp: y = x @ dequant(W)^T + bias for m <= 4 rows of x, from the packed blocks.
#include "ggq_linear.hpp"
#include "ggq_mfma.hpp"
#include "ggq_mfma16.hpp"
#include "ggq_gemm.hpp"
#include "ggq_host.hpp"
#include "../../include/ggq.h"

#include <atomic>
#include <cstdlib>

namespace {

using namespace ggq;

typedef hipError_t (*lin_fn)(const void*, const void*, const void*, void*, uint32_t, uint32_t, hipStream_t);

constexpr int MAX_DEVICES = 64;
#ifndef GGQ_TILE_WM_DEFAULT
#define GGQ_TILE_WM_DEFAULT 4
#endif

// compute units of the current device, asked once per device (the launch path runs per layer per step)
uint32_t compute_units(int dev)
{
    static std::atomic<int> cached[MAX_DEVICES];
    if (dev < 0 || dev >= MAX_DEVICES) return 256;
    int cus = cached[dev].load(std::memory_order_relaxed);
    if (cus == 0) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        cached[dev].store(cus, std::memory_order_relaxed);
    }
    return (uint32_t)cus;
}

template <class F, int OUT, int M>
hipError_t launch(const void* packed, const void* x, const void* bias, void* y, uint32_t rows, uint32_t cols, hipStream_t s)
{
    const uint32_t x_bytes = ((uint32_t)M * cols * XBytes<OUT>::V + 15u) & ~15u;
    const uint32_t slice = lin_slice_bytes(cols / F::BS * F::TS);
    const uint32_t lds = x_bytes + LIN_WAVES * slice;
    // persistent grid: enough waves to keep every CU's slots full, never more workgroups than rows need
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint32_t cus = compute_units(dev);
    const uint32_t need = (rows + LIN_WAVES - 1) / LIN_WAVES;
    uint32_t per_cu = (152u * 1024u) / lds;             // workgroups whose LDS fits one CU (160 KiB, some left to the allocator's granularity) ...
    // ... up to the wave slots this instantiation's register count leaves: 512 VGPRs per SIMD lane, allocated in eights; a workgroup puts one
    // wave on each SIMD, so workgroups per CU = waves per SIMD (asked once per instantiation)
    static std::atomic<uint32_t> by_regs{0};
    uint32_t wps = by_regs.load(std::memory_order_relaxed);
    if (wps == 0) {
        hipFuncAttributes fa;
        wps = 8;
        if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&linear_small<F, OUT, M>)) == hipSuccess && fa.numRegs > 0) {
            const uint32_t regs = ((uint32_t)fa.numRegs + 7u) & ~7u;
            wps = 512u / regs;
            wps = wps > 8u ? 8u : (wps < 1u ? 1u : wps);
        }
        by_regs.store(wps, std::memory_order_relaxed);
    }
    per_cu = per_cu > wps ? wps : (per_cu < 1u ? 1u : per_cu);
    static const int lab_per_cu = lab_int("GGQ_LIN_PER_CU", 1, 8);   // lab builds only, read once (-1 in the shipped library)
    if (lab_per_cu >= 1 && (uint32_t)lab_per_cu < per_cu) per_cu = (uint32_t)lab_per_cu;
    // (Equal shares -- every wave the same number of rows, fewer waves than the chip holds -- measured 5-10 % SLOWER than filling every wave slot and letting a
    // quarter of the waves run one row more: the extra waves hide more latency than the uneven tail costs.  profiles/r05_fused_linear_small_grid_shares.json)
    const uint32_t grid = need < cus * per_cu ? need : cus * per_cu;
    if (lds > 64 * 1024) {                      // beyond the default dynamic-LDS limit: raise it once per device for this instantiation
        static std::atomic<uint64_t> raised{0};
        const uint64_t bit = 1ull << (dev & (MAX_DEVICES - 1));
        if (!(raised.load(std::memory_order_relaxed) & bit)) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_small<F, OUT, M>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            raised.fetch_or(bit, std::memory_order_relaxed);
        }
    }
    hipLaunchKernelGGL((linear_small<F, OUT, M>), dim3(grid), dim3(LIN_WAVES * 64), lds, s, static_cast<const uint8_t*>(packed),
                       static_cast<const uint8_t*>(x), static_cast<const uint8_t*>(bias), static_cast<uint8_t*>(y), rows, cols, slice);
    return hipGetLastError();
}

struct LinEntry { int qtype, block_size, type_size; lin_fn fn[3][4]; };   // [dtype][m - 1]

#define GGQ_LIN_ROW(F, OUT) {launch<F, OUT, 1>, launch<F, OUT, 2>, launch<F, OUT, 3>, launch<F, OUT, 4>}
#define GGQ_LIN(F) LinEntry { F::ID, F::BS, F::TS, {GGQ_LIN_ROW(F, OUT_F16), GGQ_LIN_ROW(F, OUT_BF16), GGQ_LIN_ROW(F, OUT_F32)} }

const LinEntry LINEAR[] = {
    GGQ_LIN(FmtQ4_0), GGQ_LIN(FmtQ4_1), GGQ_LIN(FmtQ5_0), GGQ_LIN(FmtQ5_1), GGQ_LIN(FmtQ8_0),
    GGQ_LIN(FmtQ2_K), GGQ_LIN(FmtQ3_K), GGQ_LIN(FmtQ4_K), GGQ_LIN(FmtQ5_K), GGQ_LIN(FmtQ6_K),
    GGQ_LIN(FmtIQ4_NL), GGQ_LIN(FmtIQ4_XS),
};

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- the MFMA kernel (ggq_mfma.hpp): tile = MB*32 rows of x  x  32 output columns; MB picked from m
typedef hipError_t (*mfma_fn)(const void*, const void*, const void*, void*, uint32_t, uint32_t, uint32_t, void*, uint64_t, hipStream_t);

// K-split width of the 32-row kernel: a launch parameter since round 6 (rounds 2-5: always 4 waves per workgroup).
uint32_t mf_waves(uint32_t tiles, uint32_t n_spans, uint32_t slots, uint32_t min_kw, uint32_t max_kw);

// K split ACROSS workgroups (round 6): a weight with few rows gives the 32-row kernel few workgroups -- FLUX's 3072 x 12288 `mlp.2` is 96 of them, each walking 48 spans,
// on 256 CUs: 37 us at 64 rows of x where 12288 x 3072 (the same bytes, 384 workgroups) takes 23.  With a caller-provided workspace the launch gets gridDim.z slices of whole
// spans, every workgroup stores fp32 partial sums of its slice, and splitk_reduce adds them in slice order (deterministic), the bias, and casts.  zs = 1: no split.
uint32_t mf_splitk(uint32_t workgroups, uint32_t n_spans, uint32_t m, uint32_t rows, uint64_t workspace_bytes)
{
    static const int lab_zs = lab_int("GGQ_MF32_ZS", 1, 64);                         // lab builds only, read once (-1 in the shipped library)
    uint32_t zs = 1;
    if (lab_zs >= 1) zs = (uint32_t)lab_zs;
    else if (workgroups <= 128u && n_spans >= 16u) zs = workgroups <= 48u ? 4u : 2u;
    // (profiles/r06_splitk_across_workgroups.json, slices 1 / 2 / 4 / 6 / 8 / 12 / 16: on 3072 x 12288 two slices take 30.2 -> 20.8 us at 32 rows of x and 38.0 -> 25.5 at 64, more slices only add
    // partial-sum traffic; with two tiles of x -- 192 workgroups -- or 9216+ rows of W every split costs 5-60 %)
    if (zs > n_spans) zs = n_spans;
    if (zs < 2u || (uint64_t)zs * m * rows * 4u > workspace_bytes) return 1u;
    return zs;
}

template <class F, int OUT, int MB>
hipError_t launch_mfma(const void* packed, const void* x, const void* bias, void* y, uint32_t m, uint32_t rows, uint32_t cols, void* workspace, uint64_t workspace_bytes,
                       hipStream_t s)
{
    dim3 grid((rows + 31u) / 32u, (m + (uint32_t)(MB * 32) - 1u) / (uint32_t)(MB * 32));
    int dev = 0;
    (void)hipGetDevice(&dev);
    static const int lab_kw = lab_int("GGQ_MF32_KW", 1, 16);                         // lab builds only, read once (-1 in the shipped library)
    constexpr uint32_t cap = (uint32_t)mf_max_waves(MB);
    const uint32_t n_spans = (cols + (uint32_t)MF_SPAN - 1u) / (uint32_t)MF_SPAN;
    const uint32_t zs = workspace != nullptr ? mf_splitk(grid.x * grid.y, n_spans, m, rows, workspace_bytes) : 1u;
    grid.z = zs;
    // Measured (profiles/r06_mfma32_ksplit_width_sweep.json, widths 4 / 6 / 8 / 12 at 32..256 rows of x): 4 waves stay best wherever the launch has a workgroup
    // per CU or more -- the kernel is bound by its x loads through L2, not by latency -- and 8 win 8-12 % when it has fewer than 256 workgroups (3072-row weights)
    uint32_t kw = lab_kw >= 1 ? (uint32_t)lab_kw : (grid.x * grid.y * zs < 256u ? 8u : 4u);
    const uint32_t per_z = (n_spans + zs - 1u) / zs;
    if (kw > cap) kw = cap;
    if (kw > per_z) kw = per_z;
    const uint32_t lds = mf_lds_bytes<F, MB>(kw);
    if (lds > 64 * 1024) {                      // beyond the default dynamic-LDS limit: raise it once per device for this instantiation
        static std::atomic<uint64_t> raised{0};
        const uint64_t bit = 1ull << (dev & (MAX_DEVICES - 1));
        if (!(raised.load(std::memory_order_relaxed) & bit)) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_mfma<F, OUT, MB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            raised.fetch_or(bit, std::memory_order_relaxed);
        }
    }
    hipLaunchKernelGGL((linear_mfma<F, OUT, MB>), grid, dim3(kw * 64u), lds, s, static_cast<const uint8_t*>(packed), static_cast<const uint8_t*>(x),
                       static_cast<const uint8_t*>(bias), static_cast<uint8_t*>(y), m, rows, cols, zs > 1u ? static_cast<float*>(workspace) : nullptr);
    hipError_t err = hipGetLastError();
    if (err == hipSuccess && zs > 1u) {
        hipLaunchKernelGGL((splitk_reduce<OUT>), dim3((rows + 255u) / 256u, m), dim3(256), 0, s, static_cast<const float*>(workspace), static_cast<const uint8_t*>(bias),
                           static_cast<uint8_t*>(y), m, rows, zs);
        err = hipGetLastError();
    }
    return err;
}

// ---- the 16-row MFMA kernel (ggq_mfma16.hpp): tile = MB*16 rows of x  x  16 output columns, K split over KW waves of the workgroup.
// KW: enough waves that every SIMD holds several (one wave's memory wait is another's issue slot), never more than the 256-element spans of a row,
// and -- among the candidates -- the one whose slowest wave has the fewest spans; ties go to the smaller workgroup (less to sum at the end).
uint32_t mf_waves(uint32_t tiles, uint32_t n_spans, uint32_t slots, uint32_t min_kw, uint32_t max_kw)
{
    const uint32_t cap = n_spans < max_kw ? n_spans : max_kw;
    uint32_t best = min_kw, best_cost = ~0u;
    for (uint32_t kw = min_kw; kw <= (cap > min_kw ? cap : min_kw); kw++) {
        const uint32_t per_wave = (n_spans + kw - 1) / kw, rounds = (tiles * kw + slots - 1) / slots;
        // a wave's life ~ one memory latency + its spans; a second round of workgroups costs another whole life
        const uint32_t cost = rounds * (2u + per_wave);
        if (cost < best_cost) { best_cost = cost; best = kw; }
    }
    return best;
}

template <class F, int OUT, int MB>
hipError_t launch_mfma16(const void* packed, const void* x, const void* bias, void* y, uint32_t m, uint32_t rows, uint32_t cols, void*, uint64_t, hipStream_t s)
{
    const dim3 grid((rows + 15u) / 16u, (m + (uint32_t)(MB * 16) - 1u) / (uint32_t)(MB * 16));
    int dev = 0;
    (void)hipGetDevice(&dev);
    static const int lab_kw = lab_int("GGQ_MF16_KW", 1, MF16_MAX_WAVES);             // lab builds only, read once (-1 in the shipped library)
    const uint32_t n_spans = (cols + (uint32_t)MF_SPAN - 1u) / (uint32_t)MF_SPAN;
    // waves the chip holds at this instantiation's register count (per SIMD: 6 / 4 measured best for the formats that fit 96 / 128 registers; the ones that asked
    // for more registers hold 4 / 3 -- with 6 assumed they were launched 4.5 waves per SIMD wide and ran a second round: 12-15 % slower, profiles/r06_mfma16_spills.json)
    constexpr uint32_t per_simd = MB == 1 ? (Mf16Occ<F>::MB1 >= 5 ? 6u : 4u) : (Mf16Occ<F>::MB2 >= 4 ? 4u : 3u);
    const uint32_t slots = compute_units(dev) * 4u * per_simd;
    constexpr uint32_t cap = (uint32_t)mf16_max_waves<F, MB>();
    uint32_t kw = lab_kw >= 1 ? (uint32_t)lab_kw : mf_waves(grid.x * grid.y, n_spans, slots, 2u, cap);
    if (kw > cap) kw = cap;
    const uint32_t lds = mf16_lds_bytes<F, MB>(kw);
    hipLaunchKernelGGL((linear_mfma16<F, OUT, MB>), grid, dim3(kw * 64u), lds, s, static_cast<const uint8_t*>(packed),
                       static_cast<const uint8_t*>(x), static_cast<const uint8_t*>(bias), static_cast<uint8_t*>(y), m, rows, cols);
    return hipGetLastError();
}

// ---- the shared-tile kernel (ggq_gemm.hpp): 256 rows of x  x  256 output columns per workgroup, weights decoded once per workgroup.
// WM = 4 (shipped): 16 waves, 64 x 64 outputs each, 4 waves per SIMD; WM = 2: 8 waves, 128 x 64 each, 2 waves per SIMD -- 2-6 % slower
// (profiles/r03_gemm_tile_16_waves_and_xring.json), compiled only into A/B builds (-DGGQ_TILE_WM_AB; GGQ_TILE_WM=2 in the environment then picks it).
template <class F, int OUT, int WM>
hipError_t launch_tile_wm(const void* packed, const void* x, const void* bias, void* y, uint32_t m, uint32_t rows, uint32_t cols, hipStream_t s)
{
    constexpr uint32_t lds = (uint32_t)GemmGeom<F, WM>::LDS_BYTES;
    static_assert(lds <= 160 * 1024, "one workgroup's LDS");
    int dev = 0;
    (void)hipGetDevice(&dev);
    static std::atomic<uint64_t> raised{0};          // > 64 KiB of dynamic LDS: raise the limit once per device for this instantiation
    const uint64_t bit = 1ull << (dev & (MAX_DEVICES - 1));
    if (!(raised.load(std::memory_order_relaxed) & bit)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_tile<F, OUT, WM>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        raised.fetch_or(bit, std::memory_order_relaxed);
    }
    const uint32_t tiles_m = (m + GT_BM - 1) / GT_BM, tiles_n = (rows + GT_BN - 1) / GT_BN;
    hipLaunchKernelGGL((linear_tile<F, OUT, WM>), dim3(tiles_m * tiles_n), dim3(GemmGeom<F, WM>::THREADS), lds, s, static_cast<const uint8_t*>(packed),
                       static_cast<const uint8_t*>(x), static_cast<const uint8_t*>(bias), static_cast<uint8_t*>(y), m, rows, cols, tiles_m, tiles_n);
    return hipGetLastError();
}

template <class F, int OUT>
hipError_t launch_tile(const void* packed, const void* x, const void* bias, void* y, uint32_t m, uint32_t rows, uint32_t cols, void*, uint64_t, hipStream_t s)
{
#ifdef GGQ_TILE_WM_AB      /* A/B builds (with -DGGQ_LAB) carry both shapes; the shipped library only the default one */
    static const int wm = lab_int("GGQ_TILE_WM", 2, 4);
    if ((wm == 2 || wm == 4) && wm != GGQ_TILE_WM_DEFAULT) return launch_tile_wm<F, OUT, (GGQ_TILE_WM_DEFAULT == 4 ? 2 : 4)>(packed, x, bias, y, m, rows, cols, s);
#endif
    return launch_tile_wm<F, OUT, GGQ_TILE_WM_DEFAULT>(packed, x, bias, y, m, rows, cols, s);
}

constexpr int MFMA_SHAPES = 6;                   // MB = 1, 2, 4 (K-split kernel), the 256 x 256 shared-tile kernel, then the 16-row kernel with one / two blocks of x
struct MfmaEntry { int qtype, block_size, type_size; mfma_fn fn[2][MFMA_SHAPES]; };   // [dtype f16 / bf16][shape]
#define GGQ_MF_ROW(F, OUT) {launch_mfma<F, OUT, 1>, launch_mfma<F, OUT, 2>, launch_mfma<F, OUT, 4>, launch_tile<F, OUT>, launch_mfma16<F, OUT, 1>, launch_mfma16<F, OUT, 2>}
#define GGQ_MF(F) MfmaEntry { F::ID, F::BS, F::TS, {GGQ_MF_ROW(F, OUT_F16), GGQ_MF_ROW(F, OUT_BF16)} }
const MfmaEntry MFMA[] = {
    GGQ_MF(FmtQ4_0), GGQ_MF(FmtQ4_1), GGQ_MF(FmtQ5_0), GGQ_MF(FmtQ5_1), GGQ_MF(FmtQ8_0),
    GGQ_MF(FmtQ2_K), GGQ_MF(FmtQ3_K), GGQ_MF(FmtQ4_K), GGQ_MF(FmtQ5_K), GGQ_MF(FmtQ6_K),
    GGQ_MF(FmtIQ4_NL), GGQ_MF(FmtIQ4_XS),
};

// rows of x from which the shared-tile kernel takes over (lab builds: knob GGQ_TILE_MIN_M, read once)
uint32_t tile_min_m()
{
    static const int v = lab_int("GGQ_TILE_MIN_M", 1, 1 << 20);
    return v > 0 ? (uint32_t)v : 192u;
}

}  // namespace

namespace {

// which of the MFMA_SHAPES a request runs as; -1 = GGQ_ERR_ARG
int mfma_shape(const MfmaEntry* e, uint32_t rows, uint32_t cols, uint32_t m, int tile_rows, const void* y);

int linear_mfma_impl(int qtype, const void* packed, uint32_t rows, uint32_t cols, const void* x, uint32_t m, const void* bias,
                     void* y, int dtype, int tile_rows, void* workspace, uint64_t workspace_bytes, void* hip_stream)
{
    const MfmaEntry* e = nullptr;
    for (const MfmaEntry& c : MFMA)
        if (c.qtype == qtype) e = &c;
    if (!e) return GGQ_ERR_QTYPE;
    // the contraction length: whole 256-element spans, or -- 32-element legacy blocks only -- a last span that is a multiple of 64 (SD3.5's 2432 columns)
    const bool k_tail = cols % (uint32_t)MF_SPAN != 0;
    if ((dtype != GGQ_F16 && dtype != GGQ_BF16) || cols == 0 || (k_tail && (e->block_size != 32 || cols % 64u != 0))) return GGQ_ERR_ARG;   // caller: dequantize + GEMM
    if (rows == 0 || m == 0) return GGQ_OK;
    if (!packed || !x || !y) return GGQ_ERR_ARG;
    if (!aligned16(packed) || !aligned16(x)) return GGQ_ERR_ALIGN;
    // tile_rows: rows of x per workgroup tile.  32 / 64 / 128 = the K-split kernel (ggq_mfma.hpp: every lane decodes its own MFMA operand,
    // one decode per 32 / 64 / 128 rows of x); 256 = the shared-tile kernel (ggq_gemm.hpp: 256 x 256 output tile, weights decoded once per
    // workgroup into LDS; needs rows % 8 == 0); 0 = pick from m: one 32-row block up to m = 32, 64-row tiles to m < GGQ_TILE_MIN_M, the
    // shared-tile kernel from there on (profiles/r03_gemm_tile_bench.json).
    const int shape = mfma_shape(e, rows, cols, m, tile_rows, y);
    if (shape < 0) return GGQ_ERR_ARG;
    if (shape == 3 && !aligned16(y)) return GGQ_ERR_ALIGN;                                // the shared-tile epilogue stores 16-byte vectors
    if (workspace != nullptr && !aligned16(workspace)) return GGQ_ERR_ALIGN;
    const hipError_t err = e->fn[dtype][shape](packed, x, bias, y, m, rows, cols, workspace, workspace_bytes, static_cast<hipStream_t>(hip_stream));
    return err == hipSuccess ? GGQ_OK : hip_fail(err);
}

int mfma_shape(const MfmaEntry* e, uint32_t rows, uint32_t cols, uint32_t m, int tile_rows, const void* y)
{
    (void)e;
    const bool k_tail = cols % (uint32_t)MF_SPAN != 0;
    int shape;
    if (tile_rows == 0) {
        // the fastest FUSED shape for (m, rows).  The shared-tile kernel needs enough 256 x 256 tiles to fill the chip (one workgroup per CU, 256 CUs):
        // measured crossover against the K-split kernel's 32-column workgroups (profiles/r03_gemm_tile_bench.json): 48 tiles K-split wins by 35-80 %,
        // 84 tiles (21504 x 3072 at 256 rows) the shared tile by 13 %, 192+ tiles by 2x.  Whether ANY fused shape beats unpack + hipBLASLt is the
        // caller's question (fused.linear_mfma declines above 256 rows of x unless a tile is forced: profiles/r04_gemm_skeleton_sweep.json).
        const uint32_t n_tiles = ((m + GT_BM - 1) / GT_BM) * ((rows + GT_BN - 1) / GT_BN);
        const bool tile_ok = !k_tail && m >= tile_min_m() && rows % 8u == 0 && rows <= (1u << 22) && aligned16(y) && n_tiles >= 80u;
        // the 16-row kernel (ggq_mfma16.hpp, round 6) where it measured faster than the 32-row one: up to 8 rows of x on every weight, up to 16 rows on weights
        // of at most 4096 rows (few 32-row tiles: 96 workgroups for FLUX's 3072 x 12288).  Beyond that its x loads (every workgroup reads all of x for 16
        // output columns) cost more than the 32-row tile's wasted MFMA rows: profiles/r06_mfma16_variants_kmap_and_blocks_per_wave.json
        const bool use16 = m <= 8 || (m <= 16 && rows <= 4096u);
        shape = use16 ? 4 : (m <= 32 ? 0 : (tile_ok ? 3 : (m < 384 ? 1 : 2)));
    }
    else if (tile_rows == 16) shape = m <= 16 ? 4 : 5;              // the 16-row kernel (ggq_mfma16.hpp): one block of 16 rows of x per tile, else two (grid.y tiles beyond 32 rows)
    else if (tile_rows == 32 || tile_rows == 64 || tile_rows == 128 || tile_rows == 256) shape = tile_rows == 32 ? 0 : (tile_rows == 64 ? 1 : (tile_rows == 128 ? 2 : 3));
    else return -1;
    if (shape == 3 && (k_tail || rows % 8u != 0 || rows > (1u << 22))) return -1;   // whole spans only; 16-byte pieces of y; 32-bit offsets inside a tile's rows of y
    return shape;
}

}  // namespace

extern "C" int ggq_linear_mfma(int qtype, const void* packed, uint32_t rows, uint32_t cols, const void* x, uint32_t m, const void* bias,
                               void* y, int dtype, int tile_rows, void* hip_stream)
{
    return linear_mfma_impl(qtype, packed, rows, cols, x, m, bias, y, dtype, tile_rows, nullptr, 0, hip_stream);
}

extern "C" int ggq_linear_mfma_ws(int qtype, const void* packed, uint32_t rows, uint32_t cols, const void* x, uint32_t m, const void* bias,
                                  void* y, int dtype, int tile_rows, void* workspace, uint64_t workspace_bytes, void* hip_stream)
{
    return linear_mfma_impl(qtype, packed, rows, cols, x, m, bias, y, dtype, tile_rows, workspace, workspace_bytes, hip_stream);
}

extern "C" uint64_t ggq_linear_mfma_workspace(int qtype, uint32_t rows, uint32_t cols, uint32_t m, int tile_rows)
{
    const MfmaEntry* e = nullptr;
    for (const MfmaEntry& c : MFMA)
        if (c.qtype == qtype) e = &c;
    if (!e || rows == 0 || cols == 0 || m == 0) return 0;
    const int shape = mfma_shape(e, rows, cols, m, tile_rows, nullptr);
    if (shape < 0 || shape > 2) return 0;                                                 // only the 32-row K-split kernel splits K across workgroups
    const uint32_t mb = shape == 0 ? 1u : (shape == 1 ? 2u : 4u);
    const uint32_t workgroups = ((rows + 31u) / 32u) * ((m + mb * 32u - 1u) / (mb * 32u)), n_spans = (cols + (uint32_t)MF_SPAN - 1u) / (uint32_t)MF_SPAN;
    const uint32_t zs = mf_splitk(workgroups, n_spans, m, rows, ~0ull);
    return zs > 1u ? (uint64_t)zs * m * rows * 4u : 0u;
}

extern "C" int ggq_linear_small(int qtype, const void* packed, uint32_t rows, uint32_t cols, const void* x, uint32_t m, const void* bias,
                                void* y, int dtype, void* hip_stream)
{
    const LinEntry* e = nullptr;
    for (const LinEntry& c : LINEAR)
        if (c.qtype == qtype) e = &c;
    if (!e) return GGQ_ERR_QTYPE;
    if (dtype < 0 || dtype > 2 || m < 1 || m > 4) return GGQ_ERR_ARG;
    if (rows == 0) return GGQ_OK;
    if (cols == 0 || cols % (uint32_t)e->block_size != 0) return GGQ_ERR_ARG;
    const uint64_t row_bytes = (uint64_t)cols / (uint32_t)e->block_size * (uint32_t)e->type_size;
    const uint64_t x_bytes = (uint64_t)m * cols * (dtype == GGQ_F32 ? 4 : 2);
    if (row_bytes + 15 > (uint64_t)LIN_SLICE || x_bytes + (uint64_t)LIN_WAVES * lin_slice_bytes((uint32_t)row_bytes) > 150 * 1024) return GGQ_ERR_ARG;   // caller: dequantize + GEMM
    if (!packed || !x || !y) return GGQ_ERR_ARG;
    if (!aligned16(packed) || !aligned16(x)) return GGQ_ERR_ALIGN;
    const hipError_t err = e->fn[dtype][m - 1](packed, x, bias, y, rows, cols, static_cast<hipStream_t>(hip_stream));
    return err == hipSuccess ? GGQ_OK : hip_fail(err);
}
