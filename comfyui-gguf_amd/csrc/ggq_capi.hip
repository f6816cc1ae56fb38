// ggq_capi.hip -- the C ABI of include/ggq.h over the kernels of ggq_device.hpp.
// Host side is plain C++ on the HIP runtime; nothing here knows about torch.
#include "ggq_device.hpp"
#include "ggq_host.hpp"
#include "../../include/ggq.h"
#include "../../include/ggq_gguf.h"

#include <cstdlib>
#include <type_traits>
#include <new>
#include <vector>
#include <algorithm>

namespace {

using namespace ggq;

// Launch geometry.  Every choice below comes from alternating two settings on ONE MI355X box (box-to-box spread is +-3 %):
// the torch-free harness (tests/microbench, profiles/r01_microbench_*) proposes, bench.py -- as the headline pool AND
// inside its per-format table, whose tensors are separate 2 MiB-aligned torch allocations -- decides.  Pools: 3 G
// elements per format, packed bytes alone 4-12x the 256 MiB Infinity Cache.
//   * TEAM: who owns a group.  "coop" (default) = the 4 waves of a workgroup own 4096 output elements together: one
//     shared LDS slice, the load spread over 256 threads, one s_barrier, every wave stores 2 rows of 1 KiB (pure fills run
//     6 % faster with fewer rows per wave).  "solo" = one wavefront owns 2048 elements from first load to last store, one
//     wave per workgroup.  Coop vs solo at bench.py level (headline / per-format table): Q8_0 +6.3 %, Q5_0 +1.2 / +3.7 %,
//     Q5_K +3.3 / +0.7 %, IQ4_XS +3.2 / +1.3 %, Q4_1 +2.0 %, Q4_0 +0.6 / +3.7 %, IQ4_NL +1.4 / +1.1 %, Q4_K +0.4 / +1.5 %;
//     Q2_K +2.7 / 0 %, Q5_1 +4.5 %; Q6_K is level with its occupancy-capped solo shape and keeps it; Q3_K loses 1-5 %
//     and stays solo.  One row per wave (4 waves x 2048) halves the rate -- the
//     per-wave fixed cost dominates -- and a team holds its wave slots idle while it finds its tensor, which is why the
//     coop teams get the coarse index (run_many below): before it, coop LOST up to 3.5 % on Q5_K / Q5_0 / Q6_K.
//   * non-temporal stores (+3.4 %; the other cache-policy bits make no difference); non-temporal loads are a wash for
//     the 4/5/8-bit formats and cost 2-4 % on Q3_K / Q6_K, so those two use plain loads (Q2_K too in rounds 1-4; re-measured in round 5).
//   * XCD-aware workgroup -> group mapping on LARGE launches (profiles/r01_microbench_l_xcd_run_mapping.txt): inside each
//     tile every XCD takes a run of consecutive groups covering 256 KiB of fp16 output (64 solo groups, 32 coop groups)
//     instead of every eighth group: +1.2...+2.2 % at bench.py level; -4 % for Q2_K / Q3_K and -1.5 % for Q5_1 (identity
//     mapping for those); on 47 M-element launches it costs 1-3 %, so launches below XRUN_MIN_ELEMENTS keep the identity
//     mapping too.  Shorter runs lose 5-10 %; rotating an XCD's slot from tile to tile loses 10 %.
//   * LDS_PAD: untouched dynamic LDS added to a format's launches -- it only caps how many teams a CU holds at once.
//     Q6_K (solo) runs 4.2 % faster at 26 teams per CU; the other formats are flat up to 4 KiB and lose beyond.
// The no-LDS DIRECT engine only wins when the packed pool fits the Infinity Cache (a benchmark artefact) and is not used.
// Measurement knobs (environment, read once): GGQ_XRUN_LOG2, GGQ_LDS_PAD force one value for every format.
template <class F> struct PadOf { static constexpr uint32_t V = 0; };
template <> struct PadOf<FmtQ6_K> { static constexpr uint32_t V = 4096; };
template <class F> struct PlainLoads { static constexpr bool V = false; };    // non-temporal loads cost these 2-4 %
#ifdef GGQ_Q2K_PLAIN_LOADS  /* A/B builds only: rounds 1-4.  With the buffer-store path non-temporal loads are +1.0...1.5 % for Q2_K in both alternations
                               (profiles/r05_mode_table_q2k_shapes.json), so it follows the general rule now */
template <> struct PlainLoads<FmtQ2_K> { static constexpr bool V = true; };
#endif
template <> struct PlainLoads<FmtQ3_K> { static constexpr bool V = true; };
template <> struct PlainLoads<FmtQ6_K> { static constexpr bool V = true; };
template <class F> struct TuneSolo {             // one-wave teams x 2048 elements, runs of 64 groups
    static constexpr int G = (F::BS == 256) ? 8 : 64;
    static constexpr bool COOP = false, NTL = !PlainLoads<F>::V, NTS = true;
    static constexpr int WAVES = 1;
    static constexpr uint32_t XRUN_LOG2 = 6;
};
template <class F> struct TuneCoop {             // teams of 4 waves x 4096 elements, runs of 32 groups
    static constexpr int G = (F::BS == 256) ? 16 : 128;
    static constexpr bool COOP = true, NTL = !PlainLoads<F>::V, NTS = true;
    static constexpr int WAVES = 4;
    static constexpr uint32_t XRUN_LOG2 = 5;
};
// Single tensors between one and four dispatch rounds of the shipped shapes (8.4 M ... 33.5 M elements: FLUX 3072x3072 and
// 9216x3072, T5 4096x4096, SD3.5 7296x2432): teams of 4 waves x 8192 elements -- 4 store rows per wave, HALF as many waves, so a
// 3072x3072 tensor is ONE round of resident workgroups instead of 1.125 (the 0.125 is a whole second latency chain on an almost
// idle chip).  tests/microbench `ablayer`, one launch per tensor over pools > 2x the Infinity Cache
// (profiles/r02_microbench_layer_sized_launch_shapes.txt): Q4_K 9.4 M elements 4323 -> 5110 GB/s, 16.8 M 4954 -> 5934, 28.3 M 5273 -> 5968;
// Q5_0 17.7 M 4747 -> 5311; Q8_0 9.4 M 4717 -> 5267; level from 37.7 M up, and 3-4 % SLOWER below 8 M (one round either way).
template <class F> struct TuneMid {
    static constexpr int G = (F::BS == 256) ? 32 : 256;
    static constexpr bool COOP = true, NTL = !PlainLoads<F>::V, NTS = true;
    static constexpr int WAVES = 4;
    static constexpr uint32_t XRUN_LOG2 = 0;
};
constexpr uint64_t MID_MIN_ELEMENTS = 1ull << 23, MID_MAX_ELEMENTS = 1ull << 25;
// ... for every format, the one-wave-team formats included (`ablayer2`: Q3_K 9.4 M 3635 -> 4802 GB/s, 16.8 M 4469 -> 5688, 28.3 M 4592 -> 5821;
// Q6_K 4756 -> 5082 / 4690 -> 5462 / 5172 -> 5629; Q2_K and IQ4_XS like Q4_K).
template <class F> struct MidShape { static constexpr bool V = true; };

// Tune<F> = the shape for the stock fp16 arithmetic and fp16 output; TuneFor<F, ARITH, OUT> below picks per mode.
#ifdef GGQ_SOLO_ONLY      /* A/B builds only: one-wave teams for every format */
template <class F> struct Tune : TuneSolo<F> {};
#else
template <class F> struct Tune : TuneCoop<F> {};
#endif
#define GGQ_TUNE(F, G_, COOP_, WAVES_, XRUN_)                                                    \
    template <> struct Tune<F> {                                                                 \
        static constexpr int G = G_, WAVES = WAVES_;                                             \
        static constexpr bool COOP = COOP_, NTL = !PlainLoads<F>::V, NTS = true;                 \
        static constexpr uint32_t XRUN_LOG2 = XRUN_;                                             \
    }
//       format      G   coop  waves  log2(run)
GGQ_TUNE(FmtQ3_K,    8,  false, 1,    0);
GGQ_TUNE(FmtQ6_K,    8,  false, 1,    6);
#undef GGQ_TUNE

// Team shape per (format, arithmetic, output) -- the table after the two-box pruning of round 6 (Next #7).
// The GENERAL RULE: workgroup teams for the stock fp16 arithmetic and for every fp32 OUTPUT (one decode per chunk, halves swapped inside lane pairs: ggq_device.hpp
// pair_f32), one-wave teams for the bf16 / fp32 arithmetic modes of the Advanced loader (2-4x the VALU work per element: with only two chunks per thread the workgroup teams
// lose 3-8 % there).  An EXCEPTION stays listed only if an all-workgroup-team build against an all-one-wave-team build (tools/mode_table.py --arith, two alternations each)
// says so by >= 2 % with the same sign on TWO different boxes (tools/mode_prune.py -> profiles/r06_mode_table_two_boxes.json).  Seven of the twelve of round 5 did:
//   Q8_0 in every arithmetic mode: workgroup teams +2.4...+9.8 % (the lightest arithmetic per packed byte);
//   Q4_1, Q5_1 in fp32 arithmetic with 2-byte outputs: workgroup teams +2.3...+6.7 %;
//   Q4_K with bf16 output (the production case: FLUX computes in bf16): one-wave teams +3.8 / +4.4 %;
// one is new -- Q2_K bf16 arithmetic -> fp32 output: one-wave teams +3.5 / +5.6 % -- and five are gone: bf16 output of Q2_K / Q5_K / IQ4_NL / IQ4_XS in one-wave teams
// (-0.3...+2.8 %: sign flips or < 2 % on a box), Q5_1 bf16 -> bf16 in workgroup teams (+0.7 / +1.9 %), and Q4_1's "every mode" entry with its two bf16-arithmetic overrides
// (now just the fp32-arithmetic cells).
#ifdef GGQ_COOP_ALL_MODES     /* A/B builds only: the workgroup teams in every arithmetic mode and for every output dtype */
template <class F> struct CoopInAllModes { static constexpr bool V = true; };
#else
template <class F> struct CoopInAllModes { static constexpr bool V = false; };
template <> struct CoopInAllModes<FmtQ8_0> { static constexpr bool V = true; };
#endif
#if defined(GGQ_SOLO_CAST_OUT)      /* A/B builds only: one-wave teams whenever the output is not fp16 */
template <class F, int OUT> struct CoopForOut { static constexpr bool V = OUT == OUT_F16; };
#elif defined(GGQ_COOP_ALL_MODES)
template <class F, int OUT> struct CoopForOut { static constexpr bool V = true; };
#else
template <class F, int OUT> struct CoopForOut { static constexpr bool V = true; };
template <> struct CoopForOut<FmtQ4_K, OUT_BF16> { static constexpr bool V = false; };
#endif
template <class F, int ARITH, int OUT> struct UseCoop {
#ifdef GGQ_F32_TEAMS_R4     /* A/B builds only: fp32 output follows the arithmetic's team shape, as in rounds 1-4 (minus that table's per-cell exceptions) */
    static constexpr bool V = (ARITH == AR_F16 || CoopInAllModes<F>::V) && CoopForOut<F, OUT>::V;
#else
    static constexpr bool V = OUT == OUT_F32 || ((ARITH == AR_F16 || CoopInAllModes<F>::V) && CoopForOut<F, OUT>::V);
#endif
};
#if !defined(GGQ_SOLO_CAST_OUT) && !defined(GGQ_COOP_ALL_MODES)
#define GGQ_TEAM(F, AR, OUT_, COOP_) template <> struct UseCoop<F, AR, OUT_> { static constexpr bool V = COOP_; }
//       format   arithmetic  output   workgroup teams?     all-workgroup / all-one-wave build on box A, box B
GGQ_TEAM(FmtQ4_1, AR_F32,  OUT_F16,  true);              // 1.046  1.039
GGQ_TEAM(FmtQ4_1, AR_F32,  OUT_BF16, true);              // 1.037  1.035
GGQ_TEAM(FmtQ5_1, AR_F32,  OUT_F16,  true);              // 1.067  1.030
GGQ_TEAM(FmtQ5_1, AR_F32,  OUT_BF16, true);              // 1.058  1.023
GGQ_TEAM(FmtQ2_K, AR_BF16, OUT_F32,  false);             // 0.966  0.947
#undef GGQ_TEAM
#endif
// fp32 output writes twice the bytes per element: a workgroup team's 4096 elements would be FOUR 1-KiB store rows per wave, and a pure fill already runs
// 5 % slower with four rows per wave than with two (profiles/r01_microbench_q_stream_ceilings_rows_policy_mapping.txt).  So teams of 2048 elements there -- two
// rows per wave, as for the 2-byte outputs -- wherever the per-chunk arithmetic is light enough that one chunk per thread still hides its latency: the fp16
// arithmetic (packed ops; the stock loader) of every format, and the 32-element legacy blocks in every arithmetic.  Same-box alternation against the
// 4096-element teams (profiles/r05_mode_table_f32out_half_group.json): f16->f32 Q4_0 +6 %, Q4_1 +8 %, Q5_1 +6 %, Q8_0 +5 %, Q4_K / Q2_K +3 %, level for Q5_K /
// IQ4_NL; f32->f32 legacy blocks +3...+6 %; the K-quants in bf16 / fp32 arithmetic lose 1-10 % with it and keep 4096.
template <class T> struct HalfGroup : T { static constexpr int G = T::G / 2; };
#if defined(GGQ_F32_FULL_GROUP)      /* A/B builds only */
template <class F, int ARITH, int OUT> struct HalfForF32 { static constexpr bool V = false; };
#else
template <class F, int ARITH, int OUT> struct HalfForF32 { static constexpr bool V = OUT == OUT_F32 && (ARITH == AR_F16 || F::BS == 32); };
#endif
template <class F, int ARITH, int OUT> struct CoopShape : std::conditional<HalfForF32<F, ARITH, OUT>::V, HalfGroup<Tune<F>>, Tune<F>>::type {};
// The two formats whose 2-byte outputs run in one-wave teams: with fp32 output (8 store rows per wave there) Q6_K gains 5-8 % in workgroup teams of 2048 elements in every
// arithmetic, Q3_K loses 2-6 % and stays (same-box alternation, profiles/r05_mode_table_f32out_q3k_q6k_workgroup_teams.json).
template <class F> struct CoopForF32Only { static constexpr bool V = false; };
#ifndef GGQ_Q6K_F32_SOLO             /* A/B builds define it */
template <> struct CoopForF32Only<FmtQ6_K> { static constexpr bool V = true; };
#endif
template <class F, int ARITH, int OUT> struct TuneFor
    : std::conditional<!Tune<F>::COOP, typename std::conditional<OUT == OUT_F32 && CoopForF32Only<F>::V, HalfGroup<TuneCoop<F>>, Tune<F>>::type,
                       typename std::conditional<UseCoop<F, ARITH, OUT>::V, CoopShape<F, ARITH, OUT>, TuneSolo<F>>::type>::type {};

constexpr uint64_t XRUN_MIN_ELEMENTS = 1ull << 27;   // 134 M elements: whole-model plans, not single FLUX layers (<= 66 M)
// ... for the one-wave teams.  The workgroup teams need the run mapping more (identity costs them 8 % on the 3 G-element pool)
// and still gain 1-2 % from it on single layers (rocprof kernel times, 3072x3072: 7.33 vs 7.45 us, 3072x12288: 17.36 vs 17.66 us).
constexpr uint64_t XRUN_MIN_ELEMENTS_COOP = 1ull << 23;

// Measurement knobs of LAB builds only (ggq_host.hpp lab_int; the shipped library reads no environment): GGQ_XRUN_LOG2 forces the run
// length of the XCD mapping for every format and size (0 = identity mapping everywhere), GGQ_LDS_PAD the occupancy-capping LDS pad.

template <class F> uint32_t lds_pad_for()
{
    static const int o = lab_int("GGQ_LDS_PAD", 0, 64 * 1024);
    return o >= 0 ? (uint32_t)o : PadOf<F>::V;
}

template <class T, class F> uint32_t xrun_of(uint64_t groups)      // T = the team shape being launched
{
    static const int o = lab_int("GGQ_XRUN_LOG2", 0, 16);
    if (o >= 0) return (uint32_t)o;
    return groups * (uint64_t)(T::G * F::BS) >= (T::COOP ? XRUN_MIN_ELEMENTS_COOP : XRUN_MIN_ELEMENTS) ? T::XRUN_LOG2 : 0u;
}

thread_local int t_last_hip = 0;
constexpr uint64_t MAX_GRID = 0x7FFFFFFFull;

typedef hipError_t (*one_fn)(const Desc&, hipStream_t, bool);
typedef hipError_t (*many_fn)(const Desc*, uint32_t, uint64_t, const uint32_t*, uint32_t, hipStream_t);
typedef hipError_t (*rows_fn)(const void*, const int64_t*, void*, uint64_t, uint32_t, uint64_t, hipStream_t);

// Single tensors of layer size (per-layer calls, ops.py:177; rocprof kernel times of 3072x3072 / 3072x12288 tensors, solo vs coop
// builds): the coop shape is 3-8 % FASTER for the 4/5-bit formats (Q4_K 6.10 vs 6.61 us, 17.7 vs 18.1 us) but 3-8 % SLOWER for
// Q8_0 (6.52 vs 6.03 us, 20.6 vs 20.0 us), whose coop win only shows on whole-model launches: Q8_0 tensors below
// XRUN_MIN_ELEMENTS take the solo shape.
template <class F> struct SoloWhenSmall { static constexpr bool V = false; };
template <> struct SoloWhenSmall<FmtQ8_0> { static constexpr bool V = true; };

// Stores of SINGLE-TENSOR launches are WRITE-THROUGH (sc1), not non-temporal.  A whole-weight-set launch streams gigabytes nobody reads back
// soon: there non-temporal stores are worth +3.4 % (above).  A per-layer launch is the opposite case: the reference's next call is the GEMM
// that READS the weight just written (ops.py:242-244), and 19-132 MB of dense weight that were not stored non-temporally are still in the
// 256 MiB Infinity Cache when it starts.  Emulated FLUX.1-dev step (304 linears, 4608 tokens, bf16), same box, alternating: cost of the dequant
// path per step 5.0 ms with non-temporal stores, 2.2 ms with plain stores, 1.7-2.0 ms with sc1 -- and standalone (nothing reads the result) the sc1
// launch is within 3 % of the non-temporal one where the plain one is 8-38 % slower (3072x3072 Q4_K -> bf16: 5.25 / 5.1 / 7.15 us;
// profiles/r03_layer_store_cache_policy.json, r03_flux_forward_emulation_store_policy.json).  Both instantiations ship: ggq_dequant stores sc1 (its caller
// is a layer), ggq_dequant_stream non-temporal (results nobody reads back soon).  In lab builds (-DGGQ_LAB) GGQ_LAYER_NT_STORES=1 makes ggq_dequant stream too (A/B).
bool layer_nt_default()
{
    static const int o = lab_int("GGQ_LAYER_NT_STORES", 0, 1);
    return o == 1;
}

template <class T, class F, int ARITH, int OUT>
hipError_t launch_one(const Desc& d, hipStream_t s, bool nt)
{
    const uint64_t groups = (d.n_blocks + T::G - 1) / T::G;
    if (groups == 0) return hipSuccess;
    const uint64_t blocks = T::COOP ? groups : (groups + T::WAVES - 1) / T::WAVES;
    if (blocks > MAX_GRID) return hipErrorInvalidConfiguration;
    if (nt && T::NTS) hipLaunchKernelGGL((dequant_one<F, T::G, OUT, T::NTL, true, T::WAVES, ARITH, T::COOP>), dim3((uint32_t)blocks), dim3(T::WAVES * 64), lds_pad_for<F>(), s, d, groups, xrun_of<T, F>(groups));
    else hipLaunchKernelGGL((dequant_one<F, T::G, OUT, T::NTL, false, T::WAVES, ARITH, T::COOP>), dim3((uint32_t)blocks), dim3(T::WAVES * 64), lds_pad_for<F>(), s, d, groups, xrun_of<T, F>(groups));
    return hipGetLastError();
}

template <class F, int ARITH, int OUT>
hipError_t run_one(const Desc& d, hipStream_t s, bool nt)
{
    using Big = TuneFor<F, ARITH, OUT>;                              // the shape of whole-model launches
    using Fp16Out = TuneFor<F, ARITH, OUT_F16>;
    const uint64_t elements = d.n_blocks * (uint64_t)F::BS;
    if constexpr (MidShape<F>::V && OUT != OUT_F32) {                // (fp32 output already stores twice the rows per wave)
        if (elements > MID_MIN_ELEMENTS && elements < MID_MAX_ELEMENTS) return launch_one<TuneMid<F>, F, ARITH, OUT>(d, s, nt);
    }
    const bool layer_sized = elements < XRUN_MIN_ELEMENTS;
    if constexpr (SoloWhenSmall<F>::V && Big::COOP) {
        if (layer_sized) return launch_one<TuneSolo<F>, F, ARITH, OUT>(d, s, nt);
    }
    // one-wave teams chosen for the output cast only (CoopForOut): a whole-model effect too.  At layer size, bf16 output, rocprof
    // kernel times coop vs solo: 3072x3072 Q4_K 6.09 vs 6.52 us, Q5_K 6.21 vs 6.76, IQ4_XS 6.19 vs 6.80; 3072x12288 Q5_K 18.1 vs 18.9,
    // IQ4_XS 17.1 vs 17.9, Q4_K level (profiles/r01_layer_kernel_times_bf16_out_coop_vs_solo.json): single layers stay coop
#ifndef GGQ_CAST_SOLO_AT_LAYER_SIZE      /* A/B builds define it */
    if constexpr (!Big::COOP && Fp16Out::COOP) {
        if (layer_sized) return launch_one<Fp16Out, F, ARITH, OUT>(d, s, nt);
    }
#endif
    return launch_one<Big, F, ARITH, OUT>(d, s, nt);
}

// The coarse index replaces the binary search only for the COOP teams: bench.py, alternating builds on one box, measured
// Q4_1 +2.8 %, Q5_1 +1.3 %, Q8_0 +0.2 % with it and -0.4...-1.5 % for the solo formats (whose waves overlap the search with
// each other's memory traffic anyway, and whose binary search mostly hits the scalar cache).
template <class F, int ARITH, int OUT>
hipError_t run_many(const Desc* table, uint32_t n, uint64_t groups, const uint32_t* coarse, uint32_t coarse_shift, hipStream_t s)
{
    using T = TuneFor<F, ARITH, OUT>;
    if (groups == 0) return hipSuccess;
    const uint64_t blocks = T::COOP ? groups : (groups + T::WAVES - 1) / T::WAVES;
    if (blocks > MAX_GRID) return hipErrorInvalidConfiguration;
    hipLaunchKernelGGL((dequant_many<F, T::G, OUT, T::NTL, T::NTS, T::WAVES, ARITH, T::COOP>), dim3((uint32_t)blocks), dim3(T::WAVES * 64), lds_pad_for<F>(), s, table, n, groups, xrun_of<T, F>(groups), T::COOP ? coarse : nullptr, coarse_shift);
    return hipGetLastError();
}

// the embedding lookup: a few hundred rows of one table per call -- one-wave teams, plain loads (a token may repeat)
template <class F, int ARITH, int OUT>
hipError_t run_rows(const void* packed, const int64_t* indices, void* out, uint64_t n_rows, uint32_t row_blocks, uint64_t n_indices, hipStream_t s)
{
    using T = TuneSolo<F>;
    constexpr uint64_t MAX_GRID_Y = 65535, OUT_BYTES = (OUT == OUT_F32) ? 4 : 2;
    const uint32_t groups_per_row = (row_blocks + T::G - 1) / T::G;
    for (uint64_t first = 0; first < n_indices; first += MAX_GRID_Y) {          // grid.y = output rows
        const uint64_t rows = std::min(MAX_GRID_Y, n_indices - first);
        hipLaunchKernelGGL((dequant_rows<F, T::G, OUT, false, T::NTS, 1, ARITH, false>), dim3(groups_per_row, (uint32_t)rows), dim3(64), 0, s,
                           static_cast<const uint8_t*>(packed), indices + first, static_cast<uint8_t*>(out) + first * row_blocks * (uint64_t)F::BS * OUT_BYTES,
                           n_rows, row_blocks, (uint64_t)groups_per_row);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

struct FormatEntry {
    int qtype, block_size, type_size;
    int group[3][3];       // blocks per group, [compute dtype][out dtype] (the team shape may differ per mode)
    one_fn one[3][3];      // [compute dtype][out dtype]
    many_fn many[3][3];
    rows_fn rows[3][3];
};

#define GGQ_ROW(FN, F, AR) {FN<F, AR, OUT_F16>, FN<F, AR, OUT_BF16>, FN<F, AR, OUT_F32>}
#define GGQ_GROUPS(F, AR) {TuneFor<F, AR, OUT_F16>::G, TuneFor<F, AR, OUT_BF16>::G, TuneFor<F, AR, OUT_F32>::G}
#define GGQ_FORMAT(F)                                                                          \
    FormatEntry {                                                                              \
        F::ID, F::BS, F::TS, {GGQ_GROUPS(F, AR_F16), GGQ_GROUPS(F, AR_BF16), GGQ_GROUPS(F, AR_F32)},     \
        {GGQ_ROW(run_one, F, AR_F16), GGQ_ROW(run_one, F, AR_BF16), GGQ_ROW(run_one, F, AR_F32)},      \
        {GGQ_ROW(run_many, F, AR_F16), GGQ_ROW(run_many, F, AR_BF16), GGQ_ROW(run_many, F, AR_F32)},   \
        {GGQ_ROW(run_rows, F, AR_F16), GGQ_ROW(run_rows, F, AR_BF16), GGQ_ROW(run_rows, F, AR_F32)}    \
    }

const FormatEntry FORMATS[] = {
    GGQ_FORMAT(FmtQ4_0), GGQ_FORMAT(FmtQ4_1), GGQ_FORMAT(FmtQ5_0), GGQ_FORMAT(FmtQ5_1), GGQ_FORMAT(FmtQ8_0),
    GGQ_FORMAT(FmtQ2_K), GGQ_FORMAT(FmtQ3_K), GGQ_FORMAT(FmtQ4_K), GGQ_FORMAT(FmtQ5_K), GGQ_FORMAT(FmtQ6_K),
    GGQ_FORMAT(FmtIQ4_NL), GGQ_FORMAT(FmtIQ4_XS),
};
constexpr int N_FORMATS = (int)(sizeof(FORMATS) / sizeof(FORMATS[0]));

const FormatEntry* find_format(int qtype)
{
    for (int i = 0; i < N_FORMATS; i++)
        if (FORMATS[i].qtype == qtype) return &FORMATS[i];
    return nullptr;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int check_tensor(const FormatEntry* f, const void* packed, const void* out, uint64_t n_blocks, int compute_dtype, int out_dtype)
{
    if (!f) return GGQ_ERR_QTYPE;
    if (out_dtype < 0 || out_dtype > 2 || compute_dtype < 0 || compute_dtype > 2) return GGQ_ERR_ARG;
    if (n_blocks == 0) return GGQ_OK;
    if (!packed || !out) return GGQ_ERR_ARG;
    if (!aligned16(packed) || !aligned16(out)) return GGQ_ERR_ALIGN;
    return GGQ_OK;
}

struct Segment {
    const FormatEntry* fmt;
    int compute_dtype, out_dtype;
    uint32_t first, count;     // slice of the device table
    uint64_t groups;
    uint64_t coarse_first;     // slice of the device coarse index (entries), and its granularity
    uint32_t coarse_shift;
};


// ---- ggq_calibrate: what THIS memory system gives a stream with no arithmetic at all, measured by the product library itself so that bench.py
// can put the figure in the same JSON line as the dequant kernels' (roofline.measured_*).  The shape is the fastest stream the round-1 sweeps found on
// this part (profiles/r01_microbench_q_stream_ceilings_rows_policy_mapping.txt): one-shot workgroups of 4 waves, ONE 16-byte access per lane -- each wave
// instruction covers 1 KiB of contiguous memory, a workgroup 4 KiB -- with the dequant engine's XCD run mapping at 2^6 (every XCD writes runs of 256 KiB);
// a grid-stride loop is 10-30 % slower for fills (a wave's next store row waits behind vmcnt for the previous one's acknowledgement).
// KIND: 0 fill, 1 fill with non-temporal stores, 2 copy, 3 copy with non-temporal loads and stores, 4 read-only (the loaded words are folded into
// one value that is stored only if it equals a constant -- once in 2^32 threads on random data, into the caller's scratch).
template <int KIND>
__global__ __launch_bounds__(256) void calibrate_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, uint64_t n16)
{
    uint32_t bid = blockIdx.x;
    constexpr uint32_t XR = 6, TL = XR + 3u;
    const uint32_t tile = bid >> TL, inside = bid & ((1u << TL) - 1u);
    if (((tile + 1) << TL) <= gridDim.x) bid = (tile << TL) + ((inside & 7u) << XR) + (inside >> 3);
    const uint64_t i = (uint64_t)bid * 256ull + threadIdx.x;
    if (i >= n16) return;
    if constexpr (KIND == 0) out[i] = u32x4{(uint32_t)i, 1u, 2u, 3u};
    else if constexpr (KIND == 1) __builtin_nontemporal_store(u32x4{(uint32_t)i, 1u, 2u, 3u}, out + i);
    else if constexpr (KIND == 2) out[i] = in[i];
    else if constexpr (KIND == 3) __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
    else {
        const u32x4 v = __builtin_nontemporal_load(in + i);
        if ((v.x ^ v.y ^ v.z ^ v.w) == 0x9E3779B9u) out[threadIdx.x] = v;
    }
}

}  // namespace

int ggq::hip_fail(hipError_t e)
{
    t_last_hip = (int)e;
    return GGQ_ERR_HIP;
}

struct ggq_plan {
    std::vector<Segment> segments;
    Desc* dev_table = nullptr;
    uint32_t* dev_coarse = nullptr;    // per segment: entry holding group (c << shift), relative to the segment's first entry
    uint64_t bytes = 0;
    int device = 0;
};

extern "C" {

int ggq_abi_version(void) { return 11; }

#ifndef GGQ_BUILD_ID
#define GGQ_BUILD_ID "unstamped"
#endif
const char* ggq_build_id(void) { return GGQ_BUILD_ID; }

int ggq_supported(int qtype) { return find_format(qtype) ? 1 : 0; }

int ggq_block_size(int qtype)
{
    const FormatEntry* f = find_format(qtype);
    return f ? f->block_size : 0;
}

int ggq_type_size(int qtype)
{
    const FormatEntry* f = find_format(qtype);
    return f ? f->type_size : 0;
}

const char* ggq_strerror(int status)
{
    switch (status) {
    case GGQ_OK: return "ok";
    case GGQ_ERR_QTYPE: return "quantization type has no HIP unpacker";
    case GGQ_ERR_ALIGN: return "packed/out pointer is not 16-byte aligned";
    case GGQ_ERR_ARG: return "invalid argument";
    case GGQ_ERR_HIP: return "HIP runtime error (see ggq_last_hip_error)";
    case GGQ_ERR_NOMEM: return "out of memory (host, pinned or device allocation)";
    case GGQ_ERR_IO: return "file I/O error (open / mmap / pread)";
    case GGQ_ERR_FORMAT: return "not a GGUF v2/v3 file, or truncated / inconsistent";
    default: return "unknown ggq status";
    }
}

int ggq_last_hip_error(void) { return t_last_hip; }

int ggq_dequant(int qtype, const void* packed, uint64_t n_blocks, void* out, int compute_dtype, int out_dtype, void* hip_stream)
{
    const FormatEntry* f = find_format(qtype);
    const int rc = check_tensor(f, packed, out, n_blocks, compute_dtype, out_dtype);
    if (rc != GGQ_OK || n_blocks == 0) return rc;
    const Desc d{static_cast<const uint8_t*>(packed), static_cast<uint8_t*>(out), n_blocks, 0};
    const hipError_t e = f->one[compute_dtype][out_dtype](d, static_cast<hipStream_t>(hip_stream), layer_nt_default());
    return e == hipSuccess ? GGQ_OK : hip_fail(e);
}

int ggq_dequant_stream(int qtype, const void* packed, uint64_t n_blocks, void* out, int compute_dtype, int out_dtype, void* hip_stream)
{
    const FormatEntry* f = find_format(qtype);
    const int rc = check_tensor(f, packed, out, n_blocks, compute_dtype, out_dtype);
    if (rc != GGQ_OK || n_blocks == 0) return rc;
    const Desc d{static_cast<const uint8_t*>(packed), static_cast<uint8_t*>(out), n_blocks, 0};
    const hipError_t e = f->one[compute_dtype][out_dtype](d, static_cast<hipStream_t>(hip_stream), true);
    return e == hipSuccess ? GGQ_OK : hip_fail(e);
}

int ggq_dequant_f16(int qtype, const void* packed, uint64_t n_blocks, void* out_f16, void* hip_stream)
{
    return ggq_dequant(qtype, packed, n_blocks, out_f16, GGQ_F16, GGQ_F16, hip_stream);
}

int ggq_dequant_rows(int qtype, const void* packed, uint64_t n_rows, uint32_t row_blocks, const int64_t* indices, uint64_t n_indices, void* out,
                     int compute_dtype, int out_dtype, void* hip_stream)
{
    const FormatEntry* f = find_format(qtype);
    if (!f) return GGQ_ERR_QTYPE;
    if (out_dtype < 0 || out_dtype > 2 || compute_dtype < 0 || compute_dtype > 2) return GGQ_ERR_ARG;
    if (n_indices == 0 || row_blocks == 0) return GGQ_OK;
    if (n_rows == 0 || !packed || !indices || !out) return GGQ_ERR_ARG;
    if (!aligned16(packed) || !aligned16(out)) return GGQ_ERR_ALIGN;        // the table and the result; rows start wherever their blocks do
    const hipError_t e = f->rows[compute_dtype][out_dtype](packed, indices, out, n_rows, row_blocks, n_indices, static_cast<hipStream_t>(hip_stream));
    return e == hipSuccess ? GGQ_OK : hip_fail(e);
}

int ggq_calibrate(int kind, const void* src, void* dst, uint64_t bytes, void* hip_stream)
{
    if (kind < GGQ_CAL_FILL || kind > GGQ_CAL_READ) return GGQ_ERR_ARG;
    if (bytes == 0) return GGQ_OK;
    if (!dst || (kind >= GGQ_CAL_COPY && !src) || (bytes & 15u) != 0 || (kind == GGQ_CAL_READ && bytes < 4096)) return GGQ_ERR_ARG;
    if (!aligned16(dst) || (src && !aligned16(src))) return GGQ_ERR_ALIGN;
    const uint64_t n16 = bytes / 16;
    const uint64_t blocks = (n16 + 255) / 256;
    if (blocks > MAX_GRID) return GGQ_ERR_ARG;
    const uint32_t grid = (uint32_t)blocks;
    const u32x4* in = static_cast<const u32x4*>(src);
    u32x4* out = static_cast<u32x4*>(dst);
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    switch (kind) {
    case GGQ_CAL_FILL: hipLaunchKernelGGL(calibrate_kernel<0>, dim3(grid), dim3(256), 0, s, in, out, n16); break;
    case GGQ_CAL_FILL_NT: hipLaunchKernelGGL(calibrate_kernel<1>, dim3(grid), dim3(256), 0, s, in, out, n16); break;
    case GGQ_CAL_COPY: hipLaunchKernelGGL(calibrate_kernel<2>, dim3(grid), dim3(256), 0, s, in, out, n16); break;
    case GGQ_CAL_COPY_NT: hipLaunchKernelGGL(calibrate_kernel<3>, dim3(grid), dim3(256), 0, s, in, out, n16); break;
    default: hipLaunchKernelGGL(calibrate_kernel<4>, dim3(grid), dim3(256), 0, s, in, out, n16); break;
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? GGQ_OK : hip_fail(e);
}

int ggq_plan_create(const ggq_desc* descs, uint32_t n, ggq_plan** plan_out)
{
    if (!plan_out || (n > 0 && !descs)) return GGQ_ERR_ARG;
    *plan_out = nullptr;
    for (uint32_t i = 0; i < n; i++) {
        const int rc = check_tensor(find_format(descs[i].qtype), descs[i].packed, descs[i].out, descs[i].n_blocks, descs[i].compute_dtype, descs[i].out_dtype);
        if (rc != GGQ_OK) return rc;
    }
    ggq_plan* plan = new (std::nothrow) ggq_plan();
    if (!plan) return GGQ_ERR_NOMEM;
    std::vector<Desc> table;
    std::vector<uint32_t> coarse;
    try {
        table.reserve(n);
        for (int fi = 0; fi < N_FORMATS; fi++) {
            for (int cd_od = 0; cd_od < 9; cd_od++) {
                const int cd = cd_od / 3, od = cd_od % 3;
                Segment seg{&FORMATS[fi], cd, od, (uint32_t)table.size(), 0, 0, 0, 0};
                for (uint32_t i = 0; i < n; i++) {
                    const ggq_desc& d = descs[i];
                    if (d.qtype != FORMATS[fi].qtype || d.compute_dtype != cd || d.out_dtype != od || d.n_blocks == 0) continue;
                    table.push_back(Desc{static_cast<const uint8_t*>(d.packed), static_cast<uint8_t*>(d.out), d.n_blocks, seg.groups});
                    seg.groups += (d.n_blocks + FORMATS[fi].group[cd][od] - 1) / FORMATS[fi].group[cd][od];
                    seg.count++;
                    plan->bytes += d.n_blocks * ((uint64_t)FORMATS[fi].type_size + (uint64_t)FORMATS[fi].block_size * (od == GGQ_F32 ? 4 : 2));
                }
                if (seg.count) {
                    // coarse index: at most 65536 entries per segment, at least 64 groups per entry
                    seg.coarse_shift = 6;
                    while ((seg.groups >> seg.coarse_shift) + 1 > 65536) seg.coarse_shift++;
                    seg.coarse_first = coarse.size();
                    const uint64_t entries = (seg.groups >> seg.coarse_shift) + 1;
                    uint32_t idx = 0;
                    for (uint64_t c = 0; c < entries; c++) {
                        const uint64_t g0 = c << seg.coarse_shift;
                        while (idx + 1 < seg.count && table[seg.first + idx + 1].first_group <= g0) idx++;
                        coarse.push_back(idx);
                    }
                    plan->segments.push_back(seg);
                }
            }
        }
    } catch (const std::bad_alloc&) {
        delete plan;
        return GGQ_ERR_NOMEM;
    }
    (void)hipGetDevice(&plan->device);
    if (!table.empty()) {
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&plan->dev_table), table.size() * sizeof(Desc));
        if (e == hipSuccess) e = hipMemcpy(plan->dev_table, table.data(), table.size() * sizeof(Desc), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&plan->dev_coarse), coarse.size() * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemcpy(plan->dev_coarse, coarse.data(), coarse.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            if (plan->dev_coarse) (void)hipFree(plan->dev_coarse);
            if (plan->dev_table) (void)hipFree(plan->dev_table);
            delete plan;
            return e == hipErrorOutOfMemory ? GGQ_ERR_NOMEM : hip_fail(e);
        }
    }
    *plan_out = plan;
    return GGQ_OK;
}

int ggq_plan_launch(const ggq_plan* plan, void* hip_stream)
{
    if (!plan) return GGQ_ERR_ARG;
    int device = plan->device;
    if (!plan->segments.empty() && (hipGetDevice(&device) != hipSuccess || device != plan->device)) return GGQ_ERR_ARG;   // the table lives on the plan's device
    for (const Segment& seg : plan->segments) {
        const hipError_t e = seg.fmt->many[seg.compute_dtype][seg.out_dtype](plan->dev_table + seg.first, seg.count, seg.groups, plan->dev_coarse + seg.coarse_first,
                                                                               seg.coarse_shift, static_cast<hipStream_t>(hip_stream));
        if (e != hipSuccess) return hip_fail(e);
    }
    return GGQ_OK;
}

uint64_t ggq_plan_bytes(const ggq_plan* plan) { return plan ? plan->bytes : 0; }
uint32_t ggq_plan_kernels(const ggq_plan* plan) { return plan ? (uint32_t)plan->segments.size() : 0; }

void ggq_plan_destroy(ggq_plan* plan)
{
    if (!plan) return;
    if (plan->dev_coarse) (void)hipFree(plan->dev_coarse);
    if (plan->dev_table) (void)hipFree(plan->dev_table);
    delete plan;
}

}  // extern "C"
