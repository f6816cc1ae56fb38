// ggq_microbench.hip -- standalone MI355X harness (test infrastructure, lives under tests/):
//   1. parity of every HIP unpacker (through the C ABI of libggq_hip.so) against the CPU oracle
//      (oracle/ggq_oracle.c, linked in) on seeded blocks incl. adversarial scale fields and ragged sizes;
//   2. HBM ceilings for this traffic mix (fill / copy / "0.5625 B read + 2 B write" streams);
//   3. timing of kernel variants (group size G, non-temporal access, grid cap) instantiated
//      straight from ggq_device.hpp, over a working set >> the 256 MiB Infinity Cache.
// Build: see tests/microbench/Makefile.   Run: ./ggq_microbench [parity|ceil|variants|formats|all]
#include "../../comfyui-gguf_amd/csrc/ggq_device.hpp"
#include "ggq_lab_engine.hpp"
#include <chrono>      // the engine with its experiment knobs (XCD / DIRECT / THR / R / LPOL): ggq::lab
#include "ggq_stream.hpp"
#include "../../include/ggq.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern "C" {
int ggq_oracle_dequant_f16(int qtype, const uint8_t* packed, uint64_t n_blocks, uint16_t* out);
void ggq_oracle_cast_f16_to_bf16(const uint16_t* in, uint64_t n, uint16_t* out);
void ggq_oracle_cast_f16_to_f32(const uint16_t* in, uint64_t n, float* out);
int ggq_oracle_block_size(int qtype);
int ggq_oracle_type_size(int qtype);
}

#define HIP_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static inline uint64_t rng() { uint64_t x = rng_state; x ^= x << 13; x ^= x >> 7; x ^= x << 17; return rng_state = x; }

struct QT { int id; const char* name; int scale_off[2]; };
static const QT QTS[] = {
    {2, "Q4_0", {0, -1}}, {3, "Q4_1", {0, 2}}, {6, "Q5_0", {0, -1}}, {7, "Q5_1", {0, 2}}, {8, "Q8_0", {0, -1}},
    {10, "Q2_K", {80, 82}}, {11, "Q3_K", {108, -1}}, {12, "Q4_K", {0, 2}}, {13, "Q5_K", {0, 2}}, {14, "Q6_K", {208, -1}},
    {20, "IQ4_NL", {0, -1}}, {23, "IQ4_XS", {0, -1}},
};

static uint16_t f2h_pos(float f)  // positive normal floats only (nominal scales)
{
    uint32_t u; memcpy(&u, &f, 4);
    int e = (int)((u >> 23) & 255) - 127 + 15;
    uint32_t m = (u >> 13) & 1023;
    if (e <= 0) return 0x0001;
    return (uint16_t)((e << 10) | m);
}

static const uint16_t HARD[] = {0x0000, 0x8000, 0x0001, 0x8001, 0x03FF, 0x83FF, 0x0400, 0x8400, 0x7BFF, 0xFBFF, 0x7C00, 0xFC00,
                                0x7E00, 0x3C00, 0xBC00, 0x3C01, 0x3555, 0x2E66, 0x1400, 0x1001, 0x5640, 0x6400, 0x7000, 0xB555, 0x9400, 0xD640};

static void make_blocks(const QT& q, uint64_t n, int mode, std::vector<uint8_t>& out)
{
    const int ts = ggq_oracle_type_size(q.id);
    out.resize(n * ts + 64);
    uint64_t* p = reinterpret_cast<uint64_t*>(out.data());
    for (size_t i = 0; i < out.size() / 8; i++) p[i] = rng();
    const bool legacy = ggq_oracle_block_size(q.id) == 32 && q.id != 20;
    for (uint64_t b = 0; b < n; b++) {
        for (int k = 0; k < 2; k++) {
            if (q.scale_off[k] < 0) continue;
            uint16_t h;
            const uint64_t r = rng();
            if (mode == 0) {   // nominal (signed on odd blocks)
                const float lo = legacy ? 1e-3f : 1e-4f, hi = legacy ? 2.1e-2f : 2e-3f;
                h = f2h_pos(lo + (hi - lo) * (float)((r >> 11) * (1.0 / 9007199254740992.0)));
                if ((b & 1) && (r & 1)) h |= 0x8000;
            } else {           // adversarial
                const unsigned pick = (unsigned)(r % (sizeof(HARD) / 2 + 6));
                h = pick < sizeof(HARD) / 2 ? HARD[pick] : (uint16_t)(r >> 20);
            }
            out[b * ts + q.scale_off[k]] = (uint8_t)(h & 255);
            out[b * ts + q.scale_off[k] + 1] = (uint8_t)(h >> 8);
        }
    }
}

static inline uint16_t canon16(uint16_t h) { return (h & 0x7FFF) > 0x7C00 ? 0x7E00 : h; }
static inline uint16_t canonbf(uint16_t h) { return (h & 0x7FFF) > 0x7F80 ? 0x7FC0 : h; }

static int parity()
{
    int failures = 0;
    const uint64_t sizes_legacy[] = {1, 63, 64, 65, 4097, 300007};
    const uint64_t sizes_k[] = {1, 7, 8, 9, 513, 40003};
    for (const QT& q : QTS) {
        const int bs = ggq_oracle_block_size(q.id), ts = ggq_oracle_type_size(q.id);
        if (!ggq_supported(q.id) || ggq_block_size(q.id) != bs || ggq_type_size(q.id) != ts) { printf("PARITY %s: geometry/support mismatch\n", q.name); failures++; continue; }
        for (int mode = 0; mode < 2; mode++) {
            for (int si = 0; si < 6; si++) {
                const uint64_t n = bs == 32 ? sizes_legacy[si] : sizes_k[si];
                std::vector<uint8_t> packed;
                make_blocks(q, n, mode, packed);
                std::vector<uint16_t> want(n * bs), got(n * bs), wantbf(n * bs), gotbf(n * bs);
                std::vector<float> want32(n * bs), got32(n * bs);
                ggq_oracle_dequant_f16(q.id, packed.data(), n, want.data());
                ggq_oracle_cast_f16_to_bf16(want.data(), n * bs, wantbf.data());
                ggq_oracle_cast_f16_to_f32(want.data(), n * bs, want32.data());
                uint8_t* dp; uint8_t* dout;
                HIP_CHECK(hipMalloc(&dp, packed.size()));
                HIP_CHECK(hipMalloc(&dout, n * bs * 4 + 256));
                HIP_CHECK(hipMemcpy(dp, packed.data(), packed.size(), hipMemcpyHostToDevice));
                uint64_t bad[3] = {0, 0, 0};
                for (int od = 0; od < 3; od++) {
                    HIP_CHECK(hipMemset(dout, 0xCD, n * bs * 4 + 256));
                    const int rc = ggq_dequant(q.id, dp, n, dout, GGQ_F16, od, nullptr);
                    if (rc) { printf("PARITY %s: ggq_dequant rc=%d (%s)\n", q.name, rc, ggq_strerror(rc)); failures++; continue; }
                    HIP_CHECK(hipDeviceSynchronize());
                    uint8_t guard[256];
                    const size_t ob = (size_t)n * bs * (od == 2 ? 4 : 2);
                    HIP_CHECK(hipMemcpy(guard, dout + ob, 256, hipMemcpyDeviceToHost));
                    for (int g = 0; g < 256; g++) if (guard[g] != 0xCD) { bad[od]++; break; }   // wrote past the end
                    if (od == 0) {
                        HIP_CHECK(hipMemcpy(got.data(), dout, ob, hipMemcpyDeviceToHost));
                        for (uint64_t i = 0; i < n * bs; i++) bad[0] += canon16(got[i]) != canon16(want[i]);
                    } else if (od == 1) {
                        HIP_CHECK(hipMemcpy(gotbf.data(), dout, ob, hipMemcpyDeviceToHost));
                        for (uint64_t i = 0; i < n * bs; i++) bad[1] += canonbf(gotbf[i]) != canonbf(wantbf[i]);
                    } else {
                        HIP_CHECK(hipMemcpy(got32.data(), dout, ob, hipMemcpyDeviceToHost));
                        for (uint64_t i = 0; i < n * bs; i++) {
                            uint32_t a, b; memcpy(&a, &got32[i], 4); memcpy(&b, &want32[i], 4);
                            const bool an = (a & 0x7FFFFFFF) > 0x7F800000, bn = (b & 0x7FFFFFFF) > 0x7F800000;
                            bad[2] += (an || bn) ? (an != bn) : (a != b);
                        }
                    }
                }
                if (bad[0] | bad[1] | bad[2]) {
                    failures++;
                    printf("PARITY %-6s mode=%d n_blocks=%-7llu MISMATCH f16=%llu bf16=%llu f32=%llu\n", q.name, mode, (unsigned long long)n,
                           (unsigned long long)bad[0], (unsigned long long)bad[1], (unsigned long long)bad[2]);
                    if (bad[0]) for (uint64_t i = 0, shown = 0; i < n * bs && shown < 6; i++) if (canon16(got[i]) != canon16(want[i])) { printf("    elem %llu (blk %llu el %llu): got %04x want %04x\n", (unsigned long long)i, (unsigned long long)(i / bs), (unsigned long long)(i % bs), got[i], want[i]); shown++; }
                }
                HIP_CHECK(hipFree(dp)); HIP_CHECK(hipFree(dout));
            }
        }
        printf("PARITY %-6s %s\n", q.name, failures ? "(see above)" : "bit-exact (f16, bf16, f32 outputs; 6 sizes x 2 modes)");
        fflush(stdout);
    }
    printf("PARITY total failures: %d\n", failures);
    return failures;
}

// ------------------------------------------------------------------------------------------ timing
struct Timer {
    hipEvent_t a, b;
    Timer() { HIP_CHECK(hipEventCreate(&a)); HIP_CHECK(hipEventCreate(&b)); }
    template <class Fn> void run(Fn fn, int warm, int reps, double& med_ms, double& min_ms)
    {
        for (int i = 0; i < warm; i++) fn();
        std::vector<float> t(reps);
        for (int i = 0; i < reps; i++) {
            HIP_CHECK(hipEventRecord(a, nullptr));
            fn();
            HIP_CHECK(hipEventRecord(b, nullptr));
            HIP_CHECK(hipEventSynchronize(b));
            HIP_CHECK(hipEventElapsedTime(&t[i], a, b));
        }
        std::sort(t.begin(), t.end());
        med_ms = t[reps / 2]; min_ms = t[0];
    }
};

__global__ void k_fill_rand(uint64_t* p, uint64_t n, uint64_t seed)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t x = (i + 1) * 0x9E3779B97F4A7C15ull ^ seed; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        p[i] = x;
    }
}
__global__ void k_fix_scales(uint8_t* p, uint64_t n_blocks, int ts, int off0, int off1, int legacy)
{
    for (uint64_t b = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; b < n_blocks; b += (uint64_t)gridDim.x * blockDim.x) {
        // fp16 in [2^-10, 2^-9) legacy / [2^-13, 2^-12) K : exponent fixed, mantissa from the random bytes
        const uint16_t e = legacy ? 0x1400 : 0x0800;
        for (int k = 0; k < 2; k++) { const int off = k ? off1 : off0; if (off < 0) continue; uint8_t* s = p + b * ts + off; const uint16_t h = e | ((s[0] | (s[1] << 8)) & 0x03FF); s[0] = h & 255; s[1] = h >> 8; }
    }
}

// streams with no arithmetic: the ceilings this traffic mix can reach
__global__ __launch_bounds__(256) void k_fill16(ggq::u32x4* out, uint64_t n16)
{
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += gridDim.x * 256ull) out[i] = ggq::u32x4{(uint32_t)i, 1, 2, 3};
}
__global__ __launch_bounds__(256) void k_copy16(const ggq::u32x4* in, ggq::u32x4* out, uint64_t n16)
{
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += gridDim.x * 256ull) out[i] = in[i];
}
__global__ __launch_bounds__(256) void k_fill16nt(ggq::u32x4* out, uint64_t n16)
{
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += gridDim.x * 256ull) __builtin_nontemporal_store(ggq::u32x4{(uint32_t)i, 1, 2, 3}, out + i);
}
__global__ __launch_bounds__(256) void k_copy16nt(const ggq::u32x4* in, ggq::u32x4* out, uint64_t n16)
{
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += gridDim.x * 256ull) __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
}
// per 9 units read, 32 units written (0.5625 B in : 2 B out, the Q4_0 / Q4_K mix)
template <bool NT>
__global__ __launch_bounds__(256) void k_mix(const ggq::u32x4* in, ggq::u32x4* out, uint64_t n_tiles)
{
    // tile = 256 threads: 72 units in (threads 0..71 load), 256 units out
    for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        ggq::u32x4 v{1, 2, 3, 4};
        if (threadIdx.x < 72) v = NT ? __builtin_nontemporal_load(in + t * 72 + threadIdx.x) : in[t * 72 + threadIdx.x];
        v.x += __shfl(v.y, threadIdx.x & 7);
        if (NT) __builtin_nontemporal_store(v, out + t * 256 + threadIdx.x); else out[t * 256 + threadIdx.x] = v;
    }
}

// the same streams as one-shot 4 KiB workgroups (256 threads x 16 B) with the engine's XCD run mapping (xr = log2 run, 0 = off)
__device__ __forceinline__ uint32_t xmap(uint32_t bid, uint32_t xr)
{
    if (xr == 0) return bid;
    const uint32_t tl = xr + 3u, tile = bid >> tl, in = bid & ((1u << tl) - 1u);
    return (((tile + 1) << tl) <= gridDim.x) ? (tile << tl) + ((in & 7u) << xr) + (in >> 3) : bid;
}
template <int MODE>   // 0 fill, 1 copy, 2 mix 9:32 (72 of 256 threads load, everyone stores after a cross-lane dependency)
__global__ __launch_bounds__(256) void k_stream_x(const ggq::u32x4* in, ggq::u32x4* out, uint32_t xr)
{
    const uint64_t t = xmap(blockIdx.x, xr);
    ggq::u32x4 v{1, 2, 3, (uint32_t)t};
    if (MODE == 1) v = __builtin_nontemporal_load(in + t * 256 + threadIdx.x);
    if (MODE == 2) {
        if (threadIdx.x < 72) v = __builtin_nontemporal_load(in + t * 72 + threadIdx.x);
        v.x += __shfl(v.y, threadIdx.x & 7);
    }
    __builtin_nontemporal_store(v, out + t * 256 + threadIdx.x);
}

// pure fill, every 4 KiB piece written by ONE workgroup of W waves, each wave writing R = 4 / W rows of 1 KiB
// (W = 1: the dequant kernels' pattern, one wave stores 4 rows back to back; W = 4: one row per wave)
template <int W>
__global__ __launch_bounds__(W * 64) void k_fill_rows(ggq::u32x4* out, uint32_t xr)
{
    const uint64_t t = xmap(blockIdx.x, xr);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int R = 4 / W;
#pragma unroll
    for (int r = 0; r < R; r++)
        __builtin_nontemporal_store(ggq::u32x4{1, 2, (uint32_t)r, (uint32_t)t}, out + t * 256 + (wave * R + r) * 64 + lane);
}

// store cache-policy bits (gfx942/950: sc0, sc1, nt) on a pure fill, 4 waves x 1 row, run mapping 2^6
template <int POL>
__global__ __launch_bounds__(256) void k_fill_policy(ggq::u32x4* out, uint32_t xr)
{
    const uint64_t t = xmap(blockIdx.x, xr);
    ggq::u32x4 v{1, 2, 3, (uint32_t)t};
    ggq::u32x4* p = out + t * 256 + threadIdx.x;
    if (POL == 0) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
    if (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
    if (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    if (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    if (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(p), "v"(v) : "memory");
    if (POL == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
    if (POL == 6) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
    if (POL == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" :: "v"(p), "v"(v) : "memory");
}

static void fill_policy()
{
    Timer T; double med, mn;
    const uint64_t bytes = 6ull << 30;
    ggq::u32x4* b;
    HIP_CHECK(hipMalloc(&b, bytes));
    const uint32_t grid = (uint32_t)(bytes / 4096);
    const char* names[] = {"plain", "nt", "sc1", "sc0 sc1", "sc1 nt", "sc0 sc1 nt", "sc0", "sc0 nt"};
    for (int rep = 0; rep < 2; rep++) {
#define POLICY(P) T.run([&] { k_fill_policy<P><<<grid, 256>>>(b, 6); }, 2, 9, med, mn); printf("FILLPOL %-11s %8.1f GB/s (median) %8.1f (best)\n", names[P], bytes / med / 1e6, bytes / mn / 1e6);
        POLICY(0) POLICY(1) POLICY(2) POLICY(3) POLICY(4) POLICY(5) POLICY(6) POLICY(7)
#undef POLICY
        fflush(stdout);
    }
    HIP_CHECK(hipFree(b));
}

static void fill_rows()
{
    Timer T; double med, mn;
    const uint64_t bytes = 6ull << 30;
    ggq::u32x4* b;
    HIP_CHECK(hipMalloc(&b, bytes));
    const uint32_t grid = (uint32_t)(bytes / 4096);
    for (int rep = 0; rep < 2; rep++)
        for (uint32_t xr : {0u, 6u}) {
            T.run([&] { k_fill_rows<1><<<grid, 64>>>(b, xr); }, 2, 9, med, mn);
            printf("FILLROWS waves=1 rows/wave=4 run=2^%u %8.1f GB/s (median) %8.1f (best)\n", xr, bytes / med / 1e6, bytes / mn / 1e6);
            T.run([&] { k_fill_rows<2><<<grid, 128>>>(b, xr); }, 2, 9, med, mn);
            printf("FILLROWS waves=2 rows/wave=2 run=2^%u %8.1f GB/s (median) %8.1f (best)\n", xr, bytes / med / 1e6, bytes / mn / 1e6);
            T.run([&] { k_fill_rows<4><<<grid, 256>>>(b, xr); }, 2, 9, med, mn);
            printf("FILLROWS waves=4 rows/wave=1 run=2^%u %8.1f GB/s (median) %8.1f (best)\n", xr, bytes / med / 1e6, bytes / mn / 1e6);
            fflush(stdout);
        }
    HIP_CHECK(hipFree(b));
}

static void ceilings_x()
{
    Timer T; double med, mn;
    const uint64_t bytes = 6ull << 30;   // 6 GiB written per launch, like the bench pool's output
    ggq::u32x4 *a, *b;
    HIP_CHECK(hipMalloc(&a, bytes)); HIP_CHECK(hipMalloc(&b, bytes));
    k_fill_rand<<<4096, 256>>>(reinterpret_cast<uint64_t*>(a), bytes / 8, 1);
    HIP_CHECK(hipDeviceSynchronize());
    const uint32_t grid = (uint32_t)(bytes / 4096);
    for (int rep = 0; rep < 2; rep++)
        for (uint32_t xr : {0u, 4u, 5u, 6u, 7u, 8u, 10u}) {
            T.run([&] { k_stream_x<0><<<grid, 256>>>(a, b, xr); }, 2, 9, med, mn);
            printf("CEILX fill  run=2^%-2u %8.1f GB/s (median) %8.1f (best)\n", xr, bytes / med / 1e6, bytes / mn / 1e6);
            T.run([&] { k_stream_x<1><<<grid, 256>>>(a, b, xr); }, 2, 9, med, mn);
            printf("CEILX copy  run=2^%-2u %8.1f GB/s (median) %8.1f (best)   [read+write]\n", xr, 2.0 * bytes / med / 1e6, 2.0 * bytes / mn / 1e6);
            const double mixbytes = (double)grid * (72 + 256) * 16;
            T.run([&] { k_stream_x<2><<<grid, 256>>>(a, b, xr); }, 2, 9, med, mn);
            printf("CEILX mix   run=2^%-2u %8.1f GB/s (median) %8.1f (best)   [9:32 read:write]\n", xr, mixbytes / med / 1e6, mixbytes / mn / 1e6);
            fflush(stdout);
        }
    HIP_CHECK(hipFree(a)); HIP_CHECK(hipFree(b));
}

static void ceilings()
{
    Timer T; double med, mn;
    const uint64_t bytes = 2ull << 30;   // 2 GiB buffers
    ggq::u32x4 *a, *b;
    HIP_CHECK(hipMalloc(&a, bytes)); HIP_CHECK(hipMalloc(&b, bytes));
    k_fill_rand<<<4096, 256>>>(reinterpret_cast<uint64_t*>(a), bytes / 8, 1);
    HIP_CHECK(hipDeviceSynchronize());
    const uint64_t n16 = bytes / 16;
    for (int grid : {2048, 65536, 524288}) {
        T.run([&] { k_fill16nt<<<grid, 256>>>(b, n16); }, 3, 15, med, mn);
        printf("CEIL fill16nt grid=%-6d  %8.1f GB/s (median)  %8.1f GB/s (best)\n", grid, bytes / med / 1e6, bytes / mn / 1e6);
        T.run([&] { k_copy16nt<<<grid, 256>>>(a, b, n16); }, 3, 15, med, mn);
        printf("CEIL copy16nt grid=%-6d  %8.1f GB/s (median)  %8.1f GB/s (best)   [read+write]\n", grid, 2.0 * bytes / med / 1e6, 2.0 * bytes / mn / 1e6);
        T.run([&] { k_fill16<<<grid, 256>>>(b, n16); }, 3, 15, med, mn);
        printf("CEIL fill16   grid=%-6d  %8.1f GB/s (median)  %8.1f GB/s (best)\n", grid, bytes / med / 1e6, bytes / mn / 1e6);
        T.run([&] { k_copy16<<<grid, 256>>>(a, b, n16); }, 3, 15, med, mn);
        printf("CEIL copy16   grid=%-6d  %8.1f GB/s (median)  %8.1f GB/s (best)   [read+write]\n", grid, 2.0 * bytes / med / 1e6, 2.0 * bytes / mn / 1e6);
        const uint64_t tiles = bytes / 4096;   // writes all of b, reads 72/256 of a
        const double mixbytes = (double)tiles * (72 + 256) * 16;
        T.run([&] { k_mix<false><<<grid, 256>>>(a, b, tiles); }, 3, 15, med, mn);
        printf("CEIL mix9:32  grid=%-6d  %8.1f GB/s (median)  %8.1f GB/s (best)\n", grid, mixbytes / med / 1e6, mixbytes / mn / 1e6);
        T.run([&] { k_mix<true><<<grid, 256>>>(a, b, tiles); }, 3, 15, med, mn);
        printf("CEIL mix9:32nt grid=%-6d %8.1f GB/s (median)  %8.1f GB/s (best)\n", grid, mixbytes / med / 1e6, mixbytes / mn / 1e6);
        fflush(stdout);
    }
    HIP_CHECK(hipFree(a)); HIP_CHECK(hipFree(b));
}

// A pool of FLUX.1-dev-shaped linears (3072x3072 and 3072x12288), `pairs` of each, one format.
struct Pool {
    std::vector<ggq::Desc> descs;     // host copy (first_group filled per G at launch time)
    std::vector<uint64_t> nblk;
    uint8_t* packed = nullptr; uint8_t* out = nullptr;
    uint64_t packed_bytes = 0, out_bytes = 0, elements = 0;
    int bs = 0, ts = 0;
};

// GGQ_POOL_ALIGN (bytes, default 256): start alignment of every tensor's packed bytes inside the pool.  torch gives
// each tensor its own 2 MiB-aligned allocation (GGQ_POOL_ALIGN=2097152 mimics that); a GGUF arena packs them at 32 B.
static uint64_t pool_align()
{
    const char* e = getenv("GGQ_POOL_ALIGN");
    const uint64_t a = e ? strtoull(e, nullptr, 10) : 256;
    return a ? a : 256;
}

static Pool make_pool(const QT& q, int pairs, uint64_t only_elements = 0)
{
    const uint64_t A = pool_align();
    Pool P; P.bs = ggq_oracle_block_size(q.id); P.ts = ggq_oracle_type_size(q.id);
    const uint64_t shapes[2] = {3072ull * 3072, 3072ull * 12288};
    if (only_elements) { for (int i = 0; i < pairs; i++) { P.nblk.push_back(only_elements / P.bs); P.elements += only_elements; } }
    else for (int i = 0; i < pairs; i++) for (uint64_t el : shapes) { P.nblk.push_back(el / P.bs); P.elements += el; }
    for (uint64_t nb : P.nblk) { P.packed_bytes += (nb * P.ts + A - 1) / A * A; }
    P.out_bytes = P.elements * 2;
    HIP_CHECK(hipMalloc(&P.packed, P.packed_bytes)); HIP_CHECK(hipMalloc(&P.out, P.out_bytes));
    k_fill_rand<<<4096, 256>>>(reinterpret_cast<uint64_t*>(P.packed), P.packed_bytes / 8, q.id);
    uint64_t po = 0, oo = 0;
    const bool legacy = P.bs == 32 && q.id != 20;
    for (uint64_t nb : P.nblk) {
        k_fix_scales<<<2048, 256>>>(P.packed + po, nb, P.ts, q.scale_off[0], q.scale_off[1], legacy ? 1 : 0);
        P.descs.push_back(ggq::Desc{P.packed + po, P.out + oo, nb, 0});
        po += (nb * P.ts + A - 1) / A * A; oo += nb * P.bs * 2;
    }
    HIP_CHECK(hipDeviceSynchronize());
    return P;
}
static void free_pool(Pool& P) { HIP_CHECK(hipFree(P.packed)); HIP_CHECK(hipFree(P.out)); }

// parity of an arbitrary instantiation (variants other than the shipped one) vs the oracle
template <class F, int G, bool NTL, bool NTS, int WAVES, int XCD, bool DIRECT = false, int THR = -1, int R = 1, bool COOP = false>
static bool check_variant()
{
    const QT* q = nullptr;
    for (const QT& x : QTS) if (x.id == F::ID) q = &x;
    bool ok = true;
    for (uint64_t n : {(uint64_t)1, (uint64_t)(G - 1 > 0 ? G - 1 : 1), (uint64_t)G, (uint64_t)(3 * G + 1), (uint64_t)(G * R), (uint64_t)(G * R + 1), (uint64_t)(F::BS == 32 ? 20011 : 2503)}) {
        std::vector<uint8_t> packed; make_blocks(*q, n, (int)(n & 1), packed);
        std::vector<uint16_t> want(n * F::BS), got(n * F::BS);
        ggq_oracle_dequant_f16(F::ID, packed.data(), n, want.data());
        uint8_t *dp, *dout;
        HIP_CHECK(hipMalloc(&dp, packed.size())); HIP_CHECK(hipMalloc(&dout, n * F::BS * 2 + 256));
        HIP_CHECK(hipMemcpy(dp, packed.data(), packed.size(), hipMemcpyHostToDevice));
        HIP_CHECK(hipMemset(dout, 0xCD, n * F::BS * 2 + 256));
        const uint64_t groups = (n + G * R - 1) / (G * R);
        hipLaunchKernelGGL((ggq::lab::dequant_one<F, G, ggq::OUT_F16, NTL, NTS, WAVES, XCD, DIRECT, THR, R, ggq::AR_F16, COOP>), dim3((uint32_t)(COOP ? groups : (groups + WAVES - 1) / WAVES)), dim3(WAVES * 64), 0, nullptr,
                           ggq::Desc{dp, dout, n, 0}, groups, 0u);
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipMemcpy(got.data(), dout, n * F::BS * 2, hipMemcpyDeviceToHost));
        uint8_t guard[256]; HIP_CHECK(hipMemcpy(guard, dout + n * F::BS * 2, 256, hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < n * F::BS; i++) ok &= canon16(got[i]) == canon16(want[i]);
        for (int g = 0; g < 256; g++) ok &= guard[g] == 0xCD;
        HIP_CHECK(hipFree(dp)); HIP_CHECK(hipFree(dout));
    }
    return ok;
}

template <class F, int G, bool NTL, bool NTS, int WAVES, int XCD, bool DIRECT = false, int THR = -1>
static void time_variant(const char* name, Pool& P, Timer& T)
{
    const bool ok = check_variant<F, G, NTL, NTS, WAVES, XCD, DIRECT, THR>();
    std::vector<ggq::Desc> d = P.descs;
    uint64_t groups = 0;
    for (auto& x : d) { x.first_group = groups; groups += (x.n_blocks + G - 1) / G; }
    ggq::Desc* dt; HIP_CHECK(hipMalloc(&dt, d.size() * sizeof(ggq::Desc)));
    HIP_CHECK(hipMemcpy(dt, d.data(), d.size() * sizeof(ggq::Desc), hipMemcpyHostToDevice));
    const uint64_t blocks = (groups + WAVES - 1) / WAVES;
    double med, mn;
    T.run([&] { hipLaunchKernelGGL((ggq::lab::dequant_many<F, G, ggq::OUT_F16, NTL, NTS, WAVES, XCD, DIRECT, THR>), dim3((uint32_t)blocks), dim3(WAVES * 64), 0, nullptr, dt, (uint32_t)d.size(), groups, 0u, nullptr, 0u); }, 3, 31, med, mn);
    const double bytes = (double)P.elements * (2.0 + (double)P.ts / P.bs);
    printf("VAR %-6s %s thr=%-2d G=%-3d ntl=%d nts=%d waves=%-2d xcd=%d grid=%-7llu  %7.3f ms  %8.1f GB/s median (%.1f%% of 8 TB/s)  best %8.1f GB/s  parity=%s\n", name, DIRECT ? "direct" : "lds   ", THR, G, (int)NTL, (int)NTS, WAVES, (int)XCD,
           (unsigned long long)blocks, med, bytes / med / 1e6, bytes / med / 1e6 / 80.0, bytes / mn / 1e6, ok ? "ok" : "MISMATCH");
    fflush(stdout);
    HIP_CHECK(hipFree(dt));
}

template <class F, int G>
static void sweep_thr(const char* name, int qi)
{
    Timer T;
    Pool P = make_pool(QTS[qi], 12);
    time_variant<F, G, true, true, 4, false, false, -1>(name, P, T);
    time_variant<F, G, true, true, 4, false, false, 0>(name, P, T);
    time_variant<F, G, true, true, 4, false, false, 1>(name, P, T);
    time_variant<F, G, true, true, 4, false, false, 2>(name, P, T);
    time_variant<F, 2 * G, true, true, 4, false, false, 0>(name, P, T);
    time_variant<F, 2 * G, true, true, 4, false, false, 1>(name, P, T);
    time_variant<F, 4 * G, true, true, 4, false, false, 0>(name, P, T);
    time_variant<F, G, true, true, 8, false, false, 0>(name, P, T);
    time_variant<F, G, true, true, 2, false, false, 0>(name, P, T);
    time_variant<F, G, true, false, 4, false, false, 0>(name, P, T);
    time_variant<F, G, true, true, 4, false, false, -1>(name, P, T);
    free_pool(P);
}

static void variants()
{
    sweep_thr<ggq::FmtQ4_K, 8>("Q4_K", 7);
    sweep_thr<ggq::FmtQ2_K, 8>("Q2_K", 5);
    sweep_thr<ggq::FmtQ6_K, 8>("Q6_K", 9);
    sweep_thr<ggq::FmtQ4_0, 64>("Q4_0", 0);
    sweep_thr<ggq::FmtQ8_0, 64>("Q8_0", 4);
}

// ---- traffic skeletons: the dequant kernels' exact memory shape with no LDS and no arithmetic.
// One wave = LOAD_UNITS 16-B loads (contiguous, like a group's packed bytes) then NST 1-KiB store rows.
// DEP: 0 = no loads at all, 1 = the stores depend on the loads (as in the real kernel),
//      2 = loads issued but the stores do not wait for them.
template <int NST, int LOAD_UNITS, int DEP, bool NT, int WAVES, int THR = -1>
__global__ __launch_bounds__(WAVES * 64) void k_skel(const ggq::u32x4* __restrict__ in, ggq::u32x4* __restrict__ out, uint64_t n_waves)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const uint64_t w = (uint64_t)blockIdx.x * WAVES + wave;
    if (w >= n_waves) return;
    ggq::u32x4 acc{(uint32_t)lane, 1, 2, 3};
    constexpr int NU = (LOAD_UNITS + 63) / 64;
    ggq::u32x4 ld[NU > 0 ? NU : 1];
    if (DEP) {
#pragma unroll
        for (int u = 0; u < NU; u++) {
            const int idx = lane + 64 * u;
            ld[u] = ggq::u32x4{0, 0, 0, 0};
            if (idx < LOAD_UNITS) ld[u] = NT ? __builtin_nontemporal_load(in + w * LOAD_UNITS + idx) : in[w * LOAD_UNITS + idx];
        }
    }
    if (DEP == 1) {
#pragma unroll
        for (int u = 0; u < NU; u++) acc.x ^= (uint32_t)__builtin_amdgcn_readfirstlane((int)ld[u].y) + ld[u].x;
    }
#pragma unroll
    for (int s = 0; s < NST; s++) {
        ggq::u32x4 v = acc; v.y += s;
        if (NT) __builtin_nontemporal_store(v, out + (w * NST + s) * 64 + lane); else out[(w * NST + s) * 64 + lane] = v;
        if (s + 1 < NST) ggq::lab::store_throttle<THR>();
    }
    if (DEP == 2) {
#pragma unroll
        for (int u = 0; u < NU; u++) asm volatile("" ::"v"(ld[u].x));
    }
}

template <int NST, int LOAD_UNITS, int DEP, bool NT, int WAVES, int THR = -1>
static void run_skel(Timer& T, ggq::u32x4* in, ggq::u32x4* out, uint64_t out_bytes)
{
    const uint64_t n_waves = out_bytes / (NST * 1024ull);
    const uint64_t blocks = (n_waves + WAVES - 1) / WAVES;
    double med, mn;
    T.run([&] { hipLaunchKernelGGL((k_skel<NST, LOAD_UNITS, DEP, NT, WAVES, THR>), dim3((uint32_t)blocks), dim3(WAVES * 64), 0, nullptr, in, out, n_waves); }, 3, 21, med, mn);
    const double bytes = (double)n_waves * (NST * 1024.0 + (DEP ? LOAD_UNITS * 16.0 : 0.0));
    printf("SKEL thr=%-2d store_rows=%d load_units=%-3d dep=%d nt=%d waves=%d  out %.2f GB  %7.3f ms  %8.1f GB/s median  best %8.1f   (writes alone %.1f GB/s)\n", THR, NST, DEP ? LOAD_UNITS : 0, DEP, (int)NT, WAVES,
           out_bytes / 1e9, med, bytes / med / 1e6, bytes / mn / 1e6, n_waves * NST * 1024.0 / med / 1e6);
    fflush(stdout);
}

static void skeletons()
{
    Timer T;
    const uint64_t out_bytes = 1132462080ull;             // the 12-pair pool's fp16 output
    ggq::u32x4 *in, *out;
    HIP_CHECK(hipMalloc(&in, out_bytes)); HIP_CHECK(hipMalloc(&out, out_bytes));   // in: >= 288/1024 of out for every shape below
    k_fill_rand<<<4096, 256>>>(reinterpret_cast<uint64_t*>(in), out_bytes / 8, 1);
    HIP_CHECK(hipDeviceSynchronize());
    // pure writes: rows per wave x throttle
    run_skel<1, 0, 0, true, 4>(T, in, out, out_bytes);
    run_skel<4, 0, 0, true, 4>(T, in, out, out_bytes);      run_skel<4, 0, 0, true, 4, 0>(T, in, out, out_bytes);
    run_skel<4, 0, 0, true, 4, 1>(T, in, out, out_bytes);   run_skel<16, 0, 0, true, 4, 0>(T, in, out, out_bytes);
    run_skel<16, 0, 0, true, 4, 1>(T, in, out, out_bytes);  run_skel<64, 0, 0, true, 4, 0>(T, in, out, out_bytes);
    // Q4_K-shaped: 72 units (1152 B) in, 4 rows (4 KiB) out
    run_skel<4, 72, 1, true, 4>(T, in, out, out_bytes);     run_skel<4, 72, 1, true, 4, 0>(T, in, out, out_bytes);
    run_skel<4, 72, 1, true, 4, 1>(T, in, out, out_bytes);  run_skel<8, 144, 1, true, 4, 0>(T, in, out, out_bytes);
    run_skel<16, 288, 1, true, 4, 0>(T, in, out, out_bytes);
    // Q2_K-shaped (42 units), Q6_K-shaped (105)
    run_skel<4, 42, 1, true, 4, 0>(T, in, out, out_bytes);  run_skel<4, 105, 1, true, 4, 0>(T, in, out, out_bytes);
    run_skel<4, 72, 1, true, 4>(T, in, out, out_bytes);
    HIP_CHECK(hipFree(in)); HIP_CHECK(hipFree(out));
}

// Round 4 (Next #5, "record the ceiling experiment"): what a LAYER-SIZED launch can reach at all -- the Q4_K traffic skeleton (72 load
// units in, 4 KiB out per wave, no LDS, no arithmetic) launched once per "tensor" over a rotating pool far beyond the Infinity Cache, exactly like
// the per-layer dequant launches, for the FLUX layer sizes.  The gap between this and the whole-pool rate is ramp + drain of a 5-20 us kernel.
template <int NST, int WAVES>
static void layer_ceiling_one(Timer& T, ggq::u32x4* in, ggq::u32x4* out, uint64_t pool_out_bytes, uint64_t elements)
{
    const uint64_t out_bytes = elements * 2, n_waves = out_bytes / (NST * 1024ull), blocks = (n_waves + WAVES - 1) / WAVES;
    const uint64_t in_units = n_waves * 18ull * NST;                       // 72 units per 4 KiB of output = Q4_K's 0.5625 B per element
    const int n_t = (int)(pool_out_bytes / out_bytes);
    double med, mn;
    T.run([&] {
        for (int t = 0; t < n_t; t++)
            hipLaunchKernelGGL((k_skel<NST, 18 * NST, 1, true, WAVES, -1>), dim3((uint32_t)blocks), dim3(WAVES * 64), 0, nullptr, in + (uint64_t)t * in_units,
                               out + (uint64_t)t * (out_bytes / 16), n_waves);
    }, 3, 15, med, mn);
    const double bytes = (double)n_t * ((double)out_bytes + (double)in_units * 16.0);
    printf("LAYER-CEILING %5.1f M elements  %d launches per pass  %d waves x %d KiB rows per workgroup  %7.3f us per launch  %8.1f GB/s median (%.1f%% of 8 TB/s)  best %8.1f\n", elements / 1e6, n_t, WAVES,
           NST, med * 1e3 / n_t, bytes / med / 1e6, bytes / med / 1e6 / 80.0, bytes / mn / 1e6);
    fflush(stdout);
}

static void layer_ceiling()
{
    Timer T;
    const uint64_t pool = 1200ull << 20;
    ggq::u32x4 *in, *out;
    HIP_CHECK(hipMalloc(&in, pool)); HIP_CHECK(hipMalloc(&out, pool));
    k_fill_rand<<<4096, 256>>>(reinterpret_cast<uint64_t*>(in), pool / 8, 1);
    HIP_CHECK(hipDeviceSynchronize());
    for (uint64_t el : {3072ull * 3072, 9216ull * 3072, 12288ull * 3072, 21504ull * 3072, 600ull << 20 >> 1}) {
        layer_ceiling_one<4, 4>(T, in, out, pool, el);                     // one-wave teams x 4 KiB
        layer_ceiling_one<8, 4>(T, in, out, pool, el);
        layer_ceiling_one<2, 4>(T, in, out, pool, el);
    }
    HIP_CHECK(hipFree(in)); HIP_CHECK(hipFree(out));
}

// ---- counter study: a fixed sequence of kernel variants for rocprofv3 --pmc passes
template <class F, int G, bool NTL, bool NTS, int WAVES, int XCD, bool DIRECT, int THR>
static void launch3(Pool& P)
{
    std::vector<ggq::Desc> d = P.descs;
    uint64_t groups = 0;
    for (auto& x : d) { x.first_group = groups; groups += (x.n_blocks + G - 1) / G; }
    ggq::Desc* dt; HIP_CHECK(hipMalloc(&dt, d.size() * sizeof(ggq::Desc)));
    HIP_CHECK(hipMemcpy(dt, d.data(), d.size() * sizeof(ggq::Desc), hipMemcpyHostToDevice));
    const uint64_t blocks = (groups + WAVES - 1) / WAVES;
    for (int i = 0; i < 3; i++)
        hipLaunchKernelGGL((ggq::lab::dequant_many<F, G, ggq::OUT_F16, NTL, NTS, WAVES, XCD, DIRECT, THR>), dim3((uint32_t)blocks), dim3(WAVES * 64), 0, nullptr, dt, (uint32_t)d.size(), groups, 0u, nullptr, 0u);
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipFree(dt));
}

static void pmc2_sequence()
{
    { Pool P = make_pool(QTS[5], 12);
      launch3<ggq::FmtQ2_K, 8, true, true, 4, false, false, -1>(P);
      launch3<ggq::FmtQ2_K, 8, true, true, 4, false, true, -1>(P);
      free_pool(P); }
    { Pool P = make_pool(QTS[7], 12);
      launch3<ggq::FmtQ4_K, 8, true, true, 4, false, false, -1>(P);
      launch3<ggq::FmtQ4_K, 8, true, true, 4, false, true, -1>(P);
      free_pool(P); }
    const uint64_t out_bytes = 1132462080ull;
    ggq::u32x4 *in, *out;
    HIP_CHECK(hipMalloc(&in, out_bytes)); HIP_CHECK(hipMalloc(&out, out_bytes));
    for (int i = 0; i < 3; i++) {
        hipLaunchKernelGGL((k_skel<1, 0, 0, true, 4, -1>), dim3((uint32_t)(out_bytes / 1024 / 4)), dim3(256), 0, nullptr, in, out, out_bytes / 1024);
        hipLaunchKernelGGL((k_skel<4, 0, 0, true, 4, -1>), dim3((uint32_t)(out_bytes / 4096 / 4)), dim3(256), 0, nullptr, in, out, out_bytes / 4096);
        hipLaunchKernelGGL((k_skel<4, 72, 1, true, 4, -1>), dim3((uint32_t)(out_bytes / 4096 / 4)), dim3(256), 0, nullptr, in, out, out_bytes / 4096);
    }
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipFree(in)); HIP_CHECK(hipFree(out));
}

// ---- interleaved A/B: N variants x R rounds in one process, round-robin, so clock / thermal drift
// hits every variant equally (cdna_hip_programming.md 5.4 rule 24).  Reports the median over rounds.
#include <functional>
struct ABVariant { std::string name; std::function<void()> launch; double bytes; std::vector<double> ms; bool ok; };
struct AB {
    std::vector<ABVariant> v;
    std::vector<void*> to_free;
    void run(int rounds, int launches)
    {
        Timer T;
        for (int w = 0; w < 40; w++) for (auto& x : v) x.launch();        // sustained warm-up (~50-100 ms)
        HIP_CHECK(hipDeviceSynchronize());
        for (int r = 0; r < rounds; r++) {
            for (size_t k = 0; k < v.size(); k++) {
                auto& x = v[(k + r) % v.size()];                           // rotate the order every round
                HIP_CHECK(hipEventRecord(T.a, nullptr));
                for (int i = 0; i < launches; i++) x.launch();
                HIP_CHECK(hipEventRecord(T.b, nullptr));
                HIP_CHECK(hipEventSynchronize(T.b));
                float t; HIP_CHECK(hipEventElapsedTime(&t, T.a, T.b));
                x.ms.push_back(t / launches);
            }
        }
        for (auto& x : v) {
            std::sort(x.ms.begin(), x.ms.end());
            const double med = x.ms[x.ms.size() / 2], lo = x.ms[x.ms.size() / 10], hi = x.ms[x.ms.size() * 9 / 10];
            printf("AB %-44s %7.4f ms  %8.1f GB/s median (%.1f%% of 8 TB/s)   p10..p90 %8.1f .. %8.1f GB/s  parity=%s\n", x.name.c_str(), med, x.bytes / med / 1e6,
                   x.bytes / med / 1e6 / 80.0, x.bytes / hi / 1e6, x.bytes / lo / 1e6, x.ok ? "ok" : "MISMATCH");
        }
        fflush(stdout);
        for (void* p : to_free) HIP_CHECK(hipFree(p));
    }
};

template <class F, int G, bool NTL, bool NTS, int WAVES, int XCD, bool DIRECT, int THR, int R = 1, bool COOP = false>
static void ab_add(AB& ab, const char* name, Pool& P, int dyn_lds = 0, uint32_t xrun = 0)
{
    std::vector<ggq::Desc> d = P.descs;
    uint64_t groups = 0;
    for (auto& x : d) { x.first_group = groups; groups += (x.n_blocks + G * R - 1) / (G * R); }
    ggq::Desc* dt; HIP_CHECK(hipMalloc(&dt, d.size() * sizeof(ggq::Desc)));
    HIP_CHECK(hipMemcpy(dt, d.data(), d.size() * sizeof(ggq::Desc), hipMemcpyHostToDevice));
    ab.to_free.push_back(dt);
    const uint32_t blocks = (uint32_t)(COOP ? groups : (groups + WAVES - 1) / WAVES), n = (uint32_t)d.size();
    char buf[160];
    snprintf(buf, sizeof buf, "%s %s G=%dx%d ntl=%d nts=%d waves=%d xcd=%d xrun=%u thr=%d dynlds=%dK", name, COOP ? "coop" : (DIRECT ? "direct" : "lds"), G, R, (int)NTL, (int)NTS, WAVES, XCD, xrun, THR, dyn_lds / 1024);
    if (dyn_lds > 0) HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&ggq::lab::dequant_many<F, G, ggq::OUT_F16, NTL, NTS, WAVES, XCD, DIRECT, THR, R, ggq::AR_F16, COOP>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 16384));
    ab.v.push_back(ABVariant{buf, [=] { hipLaunchKernelGGL((ggq::lab::dequant_many<F, G, ggq::OUT_F16, NTL, NTS, WAVES, XCD, DIRECT, THR, R, ggq::AR_F16, COOP>), dim3(blocks), dim3(WAVES * 64), dyn_lds, nullptr, dt, n, groups, xrun, nullptr, 0u); },
                             (double)P.elements * (2.0 + (double)P.ts / P.bs), {}, check_variant<F, G, NTL, NTS, WAVES, XCD, DIRECT, THR, R, COOP>()});
}

// Round 2: LAYER-SIZED launches -- one dequant_one per tensor, tensor after tensor, as ComfyUI issues them (ops.py:177) -- where a launch
// is 1-5 dispatch rounds of workgroups and ramp / tail / drain are a third of the time.  One "launch" of the AB = one pass over the pool.
template <class F, int G, int WAVES, bool COOP, bool NTL = true, int OUT = ggq::OUT_F16, bool NTS = true>
static void ab_add_layer(AB& ab, const char* name, Pool& P, uint32_t xrun, int dyn_lds = 0)
{
    std::vector<ggq::Desc> d = P.descs;
    char buf[160];
    snprintf(buf, sizeof buf, "%s layer %s G=%d waves=%d ntl=%d xrun=%u %s %s", name, COOP ? "coop" : "solo", G, WAVES, (int)NTL, xrun, OUT == ggq::OUT_BF16 ? "bf16" : "f16",
             NTS ? "nt" : "sc1");
    ab.v.push_back(ABVariant{buf, [=] {
                                 for (const ggq::Desc& t : d) {
                                     const uint64_t groups = (t.n_blocks + G - 1) / G;
                                     hipLaunchKernelGGL((ggq::lab::dequant_one<F, G, OUT, NTL, NTS, WAVES, 0, false, -1, 1, ggq::AR_F16, COOP>),
                                                        dim3((uint32_t)(COOP ? groups : (groups + WAVES - 1) / WAVES)), dim3(WAVES * 64), dyn_lds, nullptr, t, groups, xrun);
                                 }
                             },
                             (double)P.elements * (2.0 + (double)P.ts / P.bs), {}, check_variant<F, G, NTL, NTS, WAVES, 0, false, -1, 1, COOP>()});   // parity of the shape (fp16 leg)
}

// Round 3: the same sweep for what the node really launches -- bf16 result, write-through (sc1) stores (Next #2 candidate (c))
template <class F, int GB>
static void ab_layer_bf16(const char* name, int qi, std::initializer_list<uint64_t> sizes)
{
    constexpr int B = ggq::OUT_BF16;
    for (uint64_t el : sizes) {
        const int n_t = (int)std::max<uint64_t>(6, std::min<uint64_t>(48, (600ull << 20) / (el * 2)));
        Pool Q = make_pool(QTS[qi], n_t, el);
        printf("LAYER-BF16 %s: %d tensors of %llu elements (%.1f M), one launch each, bf16 result, sc1 stores\n", name, n_t, (unsigned long long)el, el / 1e6);
        AB ab;
        ab_add_layer<F, 2 * GB, 4, true, true, B, false>(ab, name, Q, 0);      // 4 waves x 4096 (whole-model coop shape)
        ab_add_layer<F, 2 * GB, 4, true, true, B, false>(ab, name, Q, 5);
        ab_add_layer<F, 4 * GB, 4, true, true, B, false>(ab, name, Q, 0);      // 4 waves x 8192 (TuneMid)
        ab_add_layer<F, 4 * GB, 2, true, true, B, false>(ab, name, Q, 0);      // 2 waves x 8192
        ab_add_layer<F, 8 * GB, 4, true, true, B, false>(ab, name, Q, 0);      // 4 waves x 16384
        ab_add_layer<F, GB, 4, false, true, B, false>(ab, name, Q, 0);         // one-wave teams x 2048, 4 per workgroup
        ab_add_layer<F, GB, 1, false, true, B, false>(ab, name, Q, 0);         // one-wave teams x 2048
        ab_add_layer<F, 4 * GB, 4, true, true, B, true>(ab, name, Q, 0);       // TuneMid with non-temporal stores (reference point)
        ab.run(12, 4);
        free_pool(Q);
    }
}

// Round 4 (Next #5): group sizes that make the number of workgroups a whole multiple of the 256 CUs for the FLUX layer sizes --
// 3072 x 3072 Q4_K = 36864 super-blocks: 24 per team = 1536 workgroups = 6 per CU exactly, where the shipped 32 (TuneMid) gives 1152 = 4.5 per CU.
template <class F, int GB>
static void ab_layer_balance(const char* name, int qi, std::initializer_list<uint64_t> sizes)
{
    constexpr int B = ggq::OUT_BF16;
    for (uint64_t el : sizes) {
        const int n_t = (int)std::max<uint64_t>(6, std::min<uint64_t>(48, (600ull << 20) / (el * 2)));
        Pool Q = make_pool(QTS[qi], n_t, el);
        printf("LAYER-BALANCE %s: %d tensors of %llu elements (%.1f M), one launch each, bf16 result, sc1 stores; workgroups per CU at 4096 / 6144 / 8192 / 12288 / 16384 elements: %.2f %.2f %.2f %.2f %.2f\n",
               name, n_t, (unsigned long long)el, el / 1e6, el / 4096.0 / 256, el / 6144.0 / 256, el / 8192.0 / 256, el / 12288.0 / 256, el / 16384.0 / 256);
        AB ab;
        ab_add_layer<F, 2 * GB, 4, true, true, B, false>(ab, name, Q, 0);      // 4 waves x 4096
        ab_add_layer<F, 3 * GB, 4, true, true, B, false>(ab, name, Q, 0);      // 4 waves x 6144
        ab_add_layer<F, 4 * GB, 4, true, true, B, false>(ab, name, Q, 0);      // 4 waves x 8192 (TuneMid)
        ab_add_layer<F, 6 * GB, 4, true, true, B, false>(ab, name, Q, 0);      // 4 waves x 12288
        ab_add_layer<F, 8 * GB, 4, true, true, B, false>(ab, name, Q, 0);      // 4 waves x 16384
        ab_add_layer<F, 3 * GB, 2, true, true, B, false>(ab, name, Q, 0);      // 2 waves x 6144
        ab_add_layer<F, 6 * GB, 8, true, true, B, false>(ab, name, Q, 0);      // 8 waves x 12288
        ab.run(12, 4);
        free_pool(Q);
    }
}

template <class F, int GB /* blocks per 2048 elements */>
static void ab_layer(const char* name, int qi, std::initializer_list<uint64_t> sizes)
{
    for (uint64_t el : sizes) {
        const int n_t = (int)std::max<uint64_t>(6, std::min<uint64_t>(48, (600ull << 20) / (el * 2)));   // dense bytes of the pool > 2x the Infinity Cache
        Pool Q = make_pool(QTS[qi], n_t, el);
        printf("LAYER %s: %d tensors of %llu elements (%.1f M), one launch each\n", name, n_t, (unsigned long long)el, el / 1e6);
        AB ab;
        ab_add_layer<F, 2 * GB, 4, true>(ab, name, Q, 0);          // shipped: 4 waves own 4096 elements (2 store rows per wave)
        ab_add_layer<F, 2 * GB, 4, true>(ab, name, Q, 5);
        ab_add_layer<F, 2 * GB, 2, true>(ab, name, Q, 0);          // 2 waves own 4096: 4 rows per wave, half as many waves
        ab_add_layer<F, 2 * GB, 2, true>(ab, name, Q, 5);
        ab_add_layer<F, 4 * GB, 4, true>(ab, name, Q, 0);          // 4 waves own 8192: 4 rows per wave
        ab_add_layer<F, 4 * GB, 2, true>(ab, name, Q, 0);          // 2 waves own 8192: 8 rows per wave
        ab_add_layer<F, GB, 4, false>(ab, name, Q, 0);             // one-wave teams x 2048, 4 per workgroup
        ab_add_layer<F, GB, 1, false>(ab, name, Q, 0);             // one-wave teams x 2048
        ab.run(12, 4);
        free_pool(Q);
    }
}

template <class F, int G>
static void ab_format(const char* name, int qi)
{
    Pool P = make_pool(QTS[qi], 12);
    AB ab;
    ab_add<F, G, true, true, 4, false, false, -1>(ab, name, P);
    ab_add<F, G, true, true, 4, false, true, -1>(ab, name, P);
    ab_add<F, G, false, true, 4, false, true, -1>(ab, name, P);
    ab_add<F, G, true, true, 4, false, false, 0>(ab, name, P);
    ab_add<F, G, true, true, 2, false, false, -1>(ab, name, P);
    ab_add<F, G, true, true, 1, false, false, -1>(ab, name, P);
    ab.run(20, 8);
    free_pool(P);
}

template <class F, int G>
static void ab_occupancy(const char* name, int qi)
{
    Pool P = make_pool(QTS[qi], 12);
    AB ab;
    // dynamic LDS that is never touched: it only caps the workgroups (x4 waves) resident per CU
    ab_add<F, G, true, true, 4, false, false, -1>(ab, name, P, 0);            // 8 WG = 32 waves / CU
    ab_add<F, G, true, true, 4, false, false, -1>(ab, name, P, 18 * 1024);    // 6 WG = 24 waves
    ab_add<F, G, true, true, 4, false, false, -1>(ab, name, P, 30 * 1024);    // 4 WG = 16 waves
    ab_add<F, G, true, true, 4, false, false, -1>(ab, name, P, 44 * 1024);    // 3 WG = 12 waves
    ab_add<F, G, true, true, 4, false, false, -1>(ab, name, P, 70 * 1024);    // 2 WG =  8 waves
    ab_add<F, G, true, true, 4, false, false, -1>(ab, name, P, 140 * 1024);   // 1 WG =  4 waves
    ab_add<F, G, true, true, 4, false, false, 0>(ab, name, P, 30 * 1024);
    ab_add<F, 2 * G, true, true, 4, false, false, -1>(ab, name, P, 70 * 1024);
    ab.run(15, 8);
    free_pool(P);
}

// Pool size matters: the 256 MiB Infinity Cache keeps a packed pool that fits in it resident across
// launches (the nt stores do not evict it), which makes reads nearly free and flatters small formats.
// PAIRS = 64 -> 3.0 G elements: packed 0.99 GB (Q2_K) .. 3.2 GB (Q8_0), fp16 out 6.0 GB.
template <class F, int G>
static void ab_big(const char* name, int qi, int pairs)
{
    Pool P = make_pool(QTS[qi], pairs);
    printf("POOL %s pairs=%d packed %.2f GB out %.2f GB\n", name, pairs, P.packed_bytes / 1e9, P.out_bytes / 1e9);
    AB ab;
    ab_add<F, G, true, true, 4, false, false, -1>(ab, name, P);
    ab_add<F, G, true, true, 1, false, false, -1>(ab, name, P);
    ab_add<F, G, false, true, 4, false, false, -1>(ab, name, P);
    ab_add<F, G, false, true, 1, false, false, -1>(ab, name, P);
    ab_add<F, G, true, false, 1, false, false, -1>(ab, name, P);
    ab_add<F, G, true, true, 4, false, true, -1>(ab, name, P);
    ab_add<F, G, false, true, 4, false, true, -1>(ab, name, P);
    ab_add<F, G, false, true, 1, false, true, -1>(ab, name, P);
    ab.run(9, 3);
    free_pool(P);
}

// XCD-aware run mapping (runtime xrun argument = log2 of the run length; template XCD = 1: one eighth per XCD)
// against the identity mapping, shipped launch shape otherwise.  profiles/r01_microbench_l (template runs of
// 48/64/96/128), _m (slot rotation, reversed walk, 2 waves) and _n (slot permutations) were taken with
// experimental variants of this function that have since been removed from the engine.
template <class F, int G, bool NTL>
static void ab_xcd(const char* name, int qi, int pairs = 64)
{
    Pool P = make_pool(QTS[qi], pairs);
    printf("POOL %s pairs=%d\n", name, pairs);
    AB ab;
    for (uint32_t xr : {0u, 5u, 6u, 7u, 8u}) ab_add<F, G, NTL, true, 1, 0, false, -1>(ab, name, P, 0, xr);
    ab_add<F, G, NTL, true, 1, 1, false, -1>(ab, name, P);
    ab_add<F, G, NTL, true, 2, 0, false, -1>(ab, name, P, 0, 5u);
    ab.run(9, 3);
    free_pool(P);
}

// COOP (the workgroup's waves share ONE group; every wave stores fewer rows) against the shipped one-wave teams
template <class F, int G, bool NTL>
static void ab_coop(const char* name, int qi, int pairs, uint32_t xr)
{
    Pool P = make_pool(QTS[qi], pairs);
    printf("POOL %s pairs=%d\n", name, pairs);
    AB ab;
    ab_add<F, G, NTL, true, 1, 0, false, -1>(ab, name, P, 0, xr);
    ab_add<F, G, NTL, true, 2, 0, false, -1, 1, true>(ab, name, P, 0, xr);
    ab_add<F, 2 * G, NTL, true, 4, 0, false, -1, 1, true>(ab, name, P, 0, xr > 0 ? xr - 1 : 0);
    ab_add<F, 2 * G, NTL, true, 2, 0, false, -1, 1, true>(ab, name, P, 0, xr > 0 ? xr - 1 : 0);
    ab_add<F, 4 * G, NTL, true, 4, 0, false, -1, 1, true>(ab, name, P, 0, xr > 1 ? xr - 2 : 0);
    ab.run(9, 3);
    free_pool(P);
}

// Q2_K / Q3_K: their 672 / 880-byte groups share 128-B lines with their neighbours (12-14 % extra reads under the
// identity mapping); shapes whose groups are whole lines (Q2_K G=32) or that keep neighbours on one XCD
template <class F, bool NTL>
static void ab_small_groups(const char* name, int qi)
{
    Pool P = make_pool(QTS[qi], 64);
    printf("POOL %s pairs=64\n", name);
    AB ab;
    ab_add<F, 8, NTL, true, 1, 0, false, -1>(ab, name, P, 0, 0);
    ab_add<F, 8, NTL, true, 1, 0, false, -1>(ab, name, P, 0, 6);
    ab_add<F, 8, !NTL, true, 1, 0, false, -1>(ab, name, P, 0, 6);
    ab_add<F, 8, NTL, true, 2, 0, false, -1>(ab, name, P, 0, 0);
    ab_add<F, 8, NTL, true, 4, 0, false, -1>(ab, name, P, 0, 0);
    ab_add<F, 16, NTL, true, 1, 0, false, -1>(ab, name, P, 0, 0);
    ab_add<F, 32, NTL, true, 4, 0, false, -1, 1, true>(ab, name, P, 0, 0);
    ab_add<F, 32, NTL, true, 4, 0, false, -1, 1, true>(ab, name, P, 0, 4);
    ab_add<F, 32, !NTL, true, 4, 0, false, -1, 1, true>(ab, name, P, 0, 4);
    ab_add<F, 16, NTL, true, 4, 0, false, -1, 1, true>(ab, name, P, 0, 5);
    ab.run(9, 3);
    free_pool(P);
}

static void ab_coop_all()       // profiles/r01_microbench_o_coop_teams.txt
{
    printf("pool alignment %llu\n", (unsigned long long)pool_align());
    for (int pairs : {64, 2}) {
        const uint32_t xr = pairs == 64 ? 6 : 0;
        ab_coop<ggq::FmtQ4_0, 64, true>("Q4_0", 0, pairs, xr);
        ab_coop<ggq::FmtQ4_1, 64, true>("Q4_1", 1, pairs, xr);
        ab_coop<ggq::FmtQ5_0, 64, true>("Q5_0", 2, pairs, xr);
        ab_coop<ggq::FmtQ5_1, 64, true>("Q5_1", 3, pairs, 0);
        ab_coop<ggq::FmtQ8_0, 64, true>("Q8_0", 4, pairs, xr);
        ab_coop<ggq::FmtQ4_K, 8, true>("Q4_K", 7, pairs, xr);
        ab_coop<ggq::FmtQ5_K, 8, true>("Q5_K", 8, pairs, xr);
        ab_coop<ggq::FmtQ6_K, 8, false>("Q6_K", 9, pairs, xr);
        ab_coop<ggq::FmtIQ4_NL, 64, true>("IQ4_NL", 10, pairs, xr);
        ab_coop<ggq::FmtIQ4_XS, 8, true>("IQ4_XS", 11, pairs, xr);
    }
}

// how much does finding the tensor cost?  the same Q4_K pool as 128 descriptors (7-step binary search per wave) and as ONE
// descriptor covering the whole (contiguous) pool (no search)
static void ab_locate()
{
    Pool P = make_pool(QTS[7], 64);
    Pool M = P;                                  // shares the device buffers
    uint64_t nb = 0;
    for (auto& d : P.descs) nb += d.n_blocks;
    M.descs.assign(1, ggq::Desc{P.descs[0].packed, P.descs[0].out, nb, 0});
    AB ab;
    ab_add<ggq::FmtQ4_K, 8, true, true, 1, 0, false, -1>(ab, "Q4_K table128", P, 0, 6);
    ab_add<ggq::FmtQ4_K, 8, true, true, 1, 0, false, -1>(ab, "Q4_K merged1", M, 0, 6);
    ab_add<ggq::FmtQ4_K, 8, true, true, 4, 0, false, -1, 1, true>(ab, "Q4_K table128", P, 0, 6);
    ab_add<ggq::FmtQ4_K, 8, true, true, 4, 0, false, -1, 1, true>(ab, "Q4_K merged1", M, 0, 6);
    ab_add<ggq::FmtQ4_K, 8, true, true, 2, 0, false, -1, 1, true>(ab, "Q4_K table128", P, 0, 6);
    ab_add<ggq::FmtQ4_K, 8, true, true, 2, 0, false, -1, 1, true>(ab, "Q4_K merged1", M, 0, 6);
    ab.run(9, 3);
    free_pool(P);
}

// resident waves per CU, capped by an untouched dynamic-LDS allocation (160 KiB per CU / (static slice + dyn) workgroups)
template <class F, int G, bool NTL>
static void ab_occ1(const char* name, int qi, uint32_t xr)
{
    Pool P = make_pool(QTS[qi], 64);
    AB ab;
    for (int kb : {0, 2, 3, 4, 5, 6, 8, 12, 18})
        ab_add<F, G, NTL, true, 1, 0, false, -1>(ab, name, P, kb * 1024, xr);
    ab.run(9, 3);
    free_pool(P);
}
static void ab_occ1_all()
{
    ab_occ1<ggq::FmtQ4_K, 8, true>("Q4_K", 7, 6);
    ab_occ1<ggq::FmtQ2_K, 8, false>("Q2_K", 5, 0);
    ab_occ1<ggq::FmtQ6_K, 8, false>("Q6_K", 9, 6);
}

// Round 3 (Next #5): Q3_K reads 1.127x its packed bytes because neighbouring 880-byte groups share a 128-byte line that two
// XCDs then both fetch.  A group of 64 super-blocks is 7040 bytes = 55 lines EXACTLY: no shared line under any mapping.  Does that shape win?
static void ab_q3k_line_exact()
{
    using F = ggq::FmtQ3_K;
    Pool P = make_pool(QTS[6], 64);
    AB ab;
    ab_add<F, 8, false, true, 1, 0, false, -1, 1, false>(ab, "Q3_K", P);                 // shipped: one-wave teams, identity mapping
    ab_add<F, 64, false, true, 4, 0, false, -1, 1, true>(ab, "Q3_K", P);                 // 4 waves share 64 super-blocks (16 store rows per wave)
    ab_add<F, 64, false, true, 4, 0, false, -1, 1, true>(ab, "Q3_K", P, 0, 3);           // ... XCD runs of 8 groups
    ab_add<F, 64, false, true, 8, 0, false, -1, 1, true>(ab, "Q3_K", P);                 // 8 waves share them (8 rows per wave)
    ab_add<F, 64, false, true, 16, 0, false, -1, 1, true>(ab, "Q3_K", P);                // 16 waves (4 rows per wave)
    ab_add<F, 64, false, true, 16, 0, false, -1, 1, true>(ab, "Q3_K", P, 0, 3);
    ab_add<F, 32, false, true, 4, 0, false, -1, 1, true>(ab, "Q3_K", P, 0, 4);           // round 1's best workgroup-team shape, for reference
    ab.run(12, 4);
    free_pool(P);
}

// Round 4 (Next #8): bytes two lanes need -- by DPP from the neighbour lane instead of a second LDS read -- against the shipped
// decode, at the shipped launch shapes (workgroup teams x 4096 elements with XCD runs; one-wave teams beside them), 3.0 G-element pool.
static void ab_dpp()
{
    { Pool P = make_pool(QTS[7], 64);
      printf("POOL Q4_K pairs=64 packed %.2f GB out %.2f GB\n", P.packed_bytes / 1e9, P.out_bytes / 1e9);
      AB ab;
      ab_add<ggq::FmtQ4_K, 16, true, true, 4, 0, false, -1, 1, true>(ab, "Q4_K shipped (LDS broadcast reads + v_perm)", P, 0, 5);
      ab_add<ggq::lab::FmtQ4_K_DPP, 16, true, true, 4, 0, false, -1, 1, true>(ab, "Q4_K DPP row_shr:4 for the high-nibble lanes", P, 0, 5);
      ab_add<ggq::FmtQ4_K, 8, true, true, 1, 0, false, -1, 1, false>(ab, "Q4_K shipped, one-wave teams", P, 0, 6);
      ab_add<ggq::lab::FmtQ4_K_DPP, 8, true, true, 1, 0, false, -1, 1, false>(ab, "Q4_K DPP, one-wave teams", P, 0, 6);
      ab.run(12, 4);
      free_pool(P); }
    { Pool P = make_pool(QTS[4], 64);
      printf("POOL Q8_0 pairs=64 packed %.2f GB out %.2f GB\n", P.packed_bytes / 1e9, P.out_bytes / 1e9);
      AB ab;
      ab_add<ggq::FmtQ8_0, 128, true, true, 4, 0, false, -1, 1, true>(ab, "Q8_0 shipped (3 aligned dwords + v_alignbyte)", P, 0, 5);
      ab_add<ggq::lab::FmtQ8_0_DPP, 128, true, true, 4, 0, false, -1, 1, true>(ab, "Q8_0 DPP row_shl:1 for the third dword", P, 0, 5);
      ab_add<ggq::FmtQ8_0, 64, true, true, 1, 0, false, -1, 1, false>(ab, "Q8_0 shipped, one-wave teams", P, 0, 6);
      ab_add<ggq::lab::FmtQ8_0_DPP, 64, true, true, 1, 0, false, -1, 1, false>(ab, "Q8_0 DPP, one-wave teams", P, 0, 6);
      ab.run(12, 4);
      free_pool(P); }
}

static void ab_small_all()      // profiles/r01_microbench_p_q2k_q3k_shapes.txt
{
    ab_small_groups<ggq::FmtQ2_K, false>("Q2_K", 5);
    ab_small_groups<ggq::FmtQ3_K, false>("Q3_K", 6);
}

static void ab_nt_all()         // non-temporal vs plain loads / stores under the run mapping (NT stores: +3.4 % on Q4_K)
{
    {
        Pool P = make_pool(QTS[7], 64); AB ab;
        ab_add<ggq::FmtQ4_K, 8, true, true, 1, 0, false, -1>(ab, "Q4_K", P, 0, 6);
        ab_add<ggq::FmtQ4_K, 8, true, false, 1, 0, false, -1>(ab, "Q4_K", P, 0, 6);
        ab_add<ggq::FmtQ4_K, 8, false, true, 1, 0, false, -1>(ab, "Q4_K", P, 0, 6);
        ab_add<ggq::FmtQ4_K, 8, false, false, 1, 0, false, -1>(ab, "Q4_K", P, 0, 6);
        ab.run(9, 3); free_pool(P);
    }
    {
        Pool P = make_pool(QTS[4], 64); AB ab;
        ab_add<ggq::FmtQ8_0, 128, true, true, 4, 0, false, -1, 1, true>(ab, "Q8_0", P, 0, 5);
        ab_add<ggq::FmtQ8_0, 128, true, false, 4, 0, false, -1, 1, true>(ab, "Q8_0", P, 0, 5);
        ab_add<ggq::FmtQ8_0, 128, false, true, 4, 0, false, -1, 1, true>(ab, "Q8_0", P, 0, 5);
        ab.run(9, 3); free_pool(P);
    }
    {
        Pool P = make_pool(QTS[9], 64); AB ab;
        ab_add<ggq::FmtQ6_K, 8, false, true, 1, 0, false, -1>(ab, "Q6_K", P, 0, 6);
        ab_add<ggq::FmtQ6_K, 8, false, false, 1, 0, false, -1>(ab, "Q6_K", P, 0, 6);
        ab.run(9, 3); free_pool(P);
    }
}

static void ab_xcd_all()
{
    for (int pairs : {64, 2}) {
        ab_xcd<ggq::FmtQ4_0, 64, true>("Q4_0", 0, pairs);
        ab_xcd<ggq::FmtQ8_0, 64, true>("Q8_0", 4, pairs);
        ab_xcd<ggq::FmtQ2_K, 8, false>("Q2_K", 5, pairs);
        ab_xcd<ggq::FmtQ3_K, 8, false>("Q3_K", 6, pairs);
        ab_xcd<ggq::FmtQ4_K, 8, true>("Q4_K", 7, pairs);
        ab_xcd<ggq::FmtQ6_K, 8, false>("Q6_K", 9, pairs);
    }
}

static void ab_all()
{
    ab_big<ggq::FmtQ2_K, 8>("Q2_K", 5, 64);
    ab_big<ggq::FmtQ2_K, 8>("Q2_K", 5, 12);
    ab_big<ggq::FmtQ3_K, 8>("Q3_K", 6, 64);
    ab_big<ggq::FmtQ4_K, 8>("Q4_K", 7, 64);
    ab_big<ggq::FmtQ4_K, 8>("Q4_K", 7, 8);
    ab_big<ggq::FmtQ6_K, 8>("Q6_K", 9, 64);
    ab_big<ggq::FmtQ5_0, 64>("Q5_0", 2, 64);
    ab_big<ggq::FmtQ4_0, 64>("Q4_0", 0, 64);
    ab_big<ggq::FmtQ8_0, 64>("Q8_0", 4, 64);
}

// every format through the shipped library (plan API), as bench.py drives it
static void formats()
{
    Timer T;
    for (const QT& q : QTS) {
        Pool P = make_pool(q, 64);
        std::vector<ggq_desc> descs;
        for (auto& d : P.descs) descs.push_back(ggq_desc{q.id, GGQ_F16, d.packed, d.out, d.n_blocks});
        ggq_plan* plan = nullptr;
        int rc = ggq_plan_create(descs.data(), (uint32_t)descs.size(), &plan);
        if (rc) { printf("FMT %s: plan_create rc=%d\n", q.name, rc); continue; }
        double med, mn;
        T.run([&] { ggq_plan_launch(plan, nullptr); }, 3, 21, med, mn);
        const double bytes = (double)ggq_plan_bytes(plan);
        printf("FMT %-6s %6.2f GB/launch  %7.3f ms  %8.1f GB/s median (%.1f%% of 8 TB/s)  best %8.1f GB/s\n", q.name, bytes / 1e9, med, bytes / med / 1e6, bytes / med / 1e6 / 80.0, bytes / mn / 1e6);
        fflush(stdout);
        // single-tensor entry point on the largest tensor, back-to-back over the pool
        T.run([&] { for (auto& d : P.descs) ggq_dequant_f16(q.id, d.packed, d.n_blocks, d.out, nullptr); }, 2, 11, med, mn);
        printf("FMT %-6s per-tensor launches (%zu launches)  %7.3f ms  %8.1f GB/s median\n", q.name, P.descs.size(), med, bytes / med / 1e6);
        ggq_plan_destroy(plan);
        free_pool(P);
    }
}

// A short, fixed sequence for rocprofv3 --pmc passes: streams of KNOWN size (to calibrate
// FETCH_SIZE / WRITE_SIZE on this access pattern) followed by the shipped kernels on the bench pool.
// Round 2: which queue is full between the kernels' 80 % and the no-arithmetic streams' 86-88 %?  The SHIPPED Q4_K and Q2_K kernels on
// the bench pool (64 pairs, through ggq_plan_launch) next to the three no-arithmetic streams of `ceilx` at the same output size and
// under the same XCD run mapping: pure fill, copy, and the 9:32 read:write mix.  Driven by tests/microbench/pmc3.sh.
static void pmc3_sequence()
{
    for (int qi : {7, 5}) {
        Pool P = make_pool(QTS[qi], 64);
        std::vector<ggq_desc> descs;
        for (auto& d : P.descs) descs.push_back(ggq_desc{QTS[qi].id, GGQ_F16, d.packed, d.out, d.n_blocks});
        ggq_plan* plan = nullptr;
        if (ggq_plan_create(descs.data(), (uint32_t)descs.size(), &plan)) { printf("plan_create failed\n"); continue; }
        for (int i = 0; i < 4; i++) ggq_plan_launch(plan, nullptr);
        HIP_CHECK(hipDeviceSynchronize());
        ggq_plan_destroy(plan);
        free_pool(P);
    }
    const uint64_t bytes = 6ull << 30;
    ggq::u32x4 *a, *b;
    HIP_CHECK(hipMalloc(&a, bytes)); HIP_CHECK(hipMalloc(&b, bytes));
    k_fill_rand<<<4096, 256>>>(reinterpret_cast<uint64_t*>(a), bytes / 8, 1);
    HIP_CHECK(hipDeviceSynchronize());
    const uint32_t grid = (uint32_t)(bytes / 4096);
    for (int i = 0; i < 4; i++) {
        k_stream_x<0><<<grid, 256>>>(a, b, 5u);
        k_stream_x<1><<<grid, 256>>>(a, b, 5u);
        k_stream_x<2><<<grid, 256>>>(a, b, 5u);
    }
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipFree(a)); HIP_CHECK(hipFree(b));
}

// Host cost of ONE kernel launch (the floor under dequantize_tensor's per-call cost): hipLaunchKernelGGL as the library issues it, against
// hipModuleLaunchKernel on a hipFunction_t resolved once (hipGetFuncBySymbol) -- same kernel, same arguments, a 16-block tensor.
static void launch_cost()
{
    using K = ggq::FmtQ4_K;
    const uint64_t n = 16;
    uint8_t *dp, *dout;
    HIP_CHECK(hipMalloc(&dp, n * K::TS)); HIP_CHECK(hipMalloc(&dout, n * K::BS * 2));
    HIP_CHECK(hipMemset(dp, 0, n * K::TS));
    ggq::Desc d{dp, dout, n, 0};
    uint64_t groups = 1; uint32_t xrun = 0;
    auto kern = ggq::dequant_one<K, 16, ggq::OUT_F16, true, true, 4, ggq::AR_F16, true>;
    hipFunction_t fn = nullptr;
    const hipError_t ge = hipGetFuncBySymbol(&fn, reinterpret_cast<const void*>(kern));
    printf("LAUNCH hipGetFuncBySymbol: %s\n", hipGetErrorString(ge));
    for (int rep = 0; rep < 3; rep++) {
        const int N = 20000;
        HIP_CHECK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; i++) hipLaunchKernelGGL(kern, dim3(1), dim3(256), 0, nullptr, d, groups, xrun);
        auto t1 = std::chrono::steady_clock::now();
        HIP_CHECK(hipDeviceSynchronize());
        double a = std::chrono::duration<double, std::micro>(t1 - t0).count() / N;
        double b = -1;
        if (ge == hipSuccess) {
            void* args[] = {&d, &groups, &xrun};
            t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; i++) (void)hipModuleLaunchKernel(fn, 1, 1, 1, 256, 1, 1, 0, nullptr, args, nullptr);
            t1 = std::chrono::steady_clock::now();
            HIP_CHECK(hipDeviceSynchronize());
            b = std::chrono::duration<double, std::micro>(t1 - t0).count() / N;
        }
        printf("LAUNCH host us per launch: hipLaunchKernelGGL %.3f   hipModuleLaunchKernel %.3f\n", a, b);
    }
    HIP_CHECK(hipFree(dp)); HIP_CHECK(hipFree(dout));
}

static void pmc_sequence()
{
    const uint64_t bytes = 1ull << 30;
    ggq::u32x4 *a, *b;
    HIP_CHECK(hipMalloc(&a, bytes)); HIP_CHECK(hipMalloc(&b, bytes));
    k_fill_rand<<<4096, 256>>>(reinterpret_cast<uint64_t*>(a), bytes / 8, 1);
    HIP_CHECK(hipDeviceSynchronize());
    const uint64_t n16 = bytes / 16;
    printf("PMC known sizes: k_copy16/k_copy16nt read %llu B + write %llu B; k_fill16/k_fill16nt write %llu B\n", (unsigned long long)bytes, (unsigned long long)bytes, (unsigned long long)bytes);
    for (int i = 0; i < 3; i++) { k_copy16<<<262144, 256>>>(a, b, n16); k_copy16nt<<<262144, 256>>>(a, b, n16); k_fill16<<<262144, 256>>>(b, n16); k_fill16nt<<<262144, 256>>>(b, n16); }
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipFree(a)); HIP_CHECK(hipFree(b));
    for (int qi : {7, 0, 9, 4, 2, 5, 6}) {
        Pool P = make_pool(QTS[qi], 64);
        std::vector<ggq_desc> descs;
        for (auto& d : P.descs) descs.push_back(ggq_desc{QTS[qi].id, GGQ_F16, d.packed, d.out, d.n_blocks});
        ggq_plan* plan = nullptr;
        if (ggq_plan_create(descs.data(), (uint32_t)descs.size(), &plan)) { printf("plan_create failed\n"); continue; }
        printf("PMC %s pool: packed read %llu B + fp16 write %llu B = %llu B per launch\n", QTS[qi].name, (unsigned long long)(P.elements / P.bs * P.ts),
               (unsigned long long)P.out_bytes, (unsigned long long)ggq_plan_bytes(plan));
        for (int i = 0; i < 5; i++) ggq_plan_launch(plan, nullptr);
        HIP_CHECK(hipDeviceSynchronize());
        ggq_plan_destroy(plan);
        if (QTS[qi].id == 12) {
            // the other (compute dtype -> out dtype) modes of the headline format on the same packed pool
            uint8_t* out32 = nullptr;
            HIP_CHECK(hipMalloc(&out32, P.elements * 4));
            const int modes[][2] = {{GGQ_F16, GGQ_BF16}, {GGQ_BF16, GGQ_BF16}, {GGQ_F32, GGQ_F32}, {GGQ_F16, GGQ_F32}};
            for (const auto& m : modes) {
                std::vector<ggq_desc> dm;
                uint64_t oo = 0;
                for (auto& d : P.descs) {
                    dm.push_back(ggq_desc{QTS[qi].id, m[1], d.packed, out32 + oo, d.n_blocks, m[0], 0});
                    oo += d.n_blocks * P.bs * (m[1] == GGQ_F32 ? 4 : 2);
                }
                ggq_plan* pm = nullptr;
                if (ggq_plan_create(dm.data(), (uint32_t)dm.size(), &pm)) { printf("plan_create failed\n"); continue; }
                printf("PMC %s compute=%d out=%d: %llu B per launch\n", QTS[qi].name, m[0], m[1], (unsigned long long)ggq_plan_bytes(pm));
                for (int i = 0; i < 5; i++) ggq_plan_launch(pm, nullptr);
                HIP_CHECK(hipDeviceSynchronize());
                ggq_plan_destroy(pm);
            }
            HIP_CHECK(hipFree(out32));
        }
        free_pool(P);
    }
}

// ---- persistent, software-pipelined engine (ggq_stream.hpp) against the shipped one-shot shapes ----------------------
static uint8_t* stream_trash()
{
    static uint8_t* t = nullptr;
    if (!t) HIP_CHECK(hipMalloc(&t, 4096));
    return t;
}

template <class F, int G, bool NTL, int D>
static bool check_stream()
{
    const QT* q = nullptr;
    for (const QT& x : QTS) if (x.id == F::ID) q = &x;
    bool ok = true;
    // two tensors per launch (ragged first one), grids from "fewer groups than waves" to "many turns of the ring"
    for (uint64_t n : {(uint64_t)1, (uint64_t)(3 * G + 1), (uint64_t)(G * 40), (uint64_t)(G * 333 + 5)}) {
        for (uint32_t grid : {8u, 64u}) {
            for (uint32_t xr : {0u, 2u}) {
                const uint64_t n2 = (uint64_t)G * 17;
                std::vector<uint8_t> p1, p2; make_blocks(*q, n, (int)(n & 1), p1); make_blocks(*q, n2, 0, p2);
                std::vector<uint16_t> want((n + n2) * F::BS), got((n + n2) * F::BS);
                ggq_oracle_dequant_f16(F::ID, p1.data(), n, want.data());
                ggq_oracle_dequant_f16(F::ID, p2.data(), n2, want.data() + n * F::BS);
                const uint64_t p1pad = (p1.size() + 255) / 256 * 256;
                uint8_t *dp, *dout;
                HIP_CHECK(hipMalloc(&dp, p1pad + p2.size())); HIP_CHECK(hipMalloc(&dout, (n + n2) * F::BS * 2 + 256));
                HIP_CHECK(hipMemcpy(dp, p1.data(), p1.size(), hipMemcpyHostToDevice));
                HIP_CHECK(hipMemcpy(dp + p1pad, p2.data(), p2.size(), hipMemcpyHostToDevice));
                HIP_CHECK(hipMemset(dout, 0xCD, (n + n2) * F::BS * 2 + 256));
                const uint64_t g1 = (n + G - 1) / G, g2 = (n2 + G - 1) / G;
                const ggq::Desc tab[2] = {{dp, dout, n, 0}, {dp + p1pad, dout + n * F::BS * 2, n2, g1}};
                ggq::Desc* dt; HIP_CHECK(hipMalloc(&dt, sizeof tab)); HIP_CHECK(hipMemcpy(dt, tab, sizeof tab, hipMemcpyHostToDevice));
                hipLaunchKernelGGL((ggq::dequant_many_stream<F, G, ggq::OUT_F16, NTL, true, D>), dim3(grid), dim3(64), 0, nullptr, dt, 2u, g1 + g2, xr, nullptr, 0u, stream_trash());
                HIP_CHECK(hipDeviceSynchronize());
                HIP_CHECK(hipMemcpy(got.data(), dout, (n + n2) * F::BS * 2, hipMemcpyDeviceToHost));
                uint8_t guard[256]; HIP_CHECK(hipMemcpy(guard, dout + (n + n2) * F::BS * 2, 256, hipMemcpyDeviceToHost));
                for (uint64_t i = 0; i < (n + n2) * F::BS; i++) ok &= canon16(got[i]) == canon16(want[i]);
                for (int g = 0; g < 256; g++) ok &= guard[g] == 0xCD;
                HIP_CHECK(hipFree(dp)); HIP_CHECK(hipFree(dout)); HIP_CHECK(hipFree(dt));
            }
        }
    }
    return ok;
}

template <class F, int G, bool NTL, int D>
static void ab_add_stream(AB& ab, const char* name, Pool& P, uint32_t waves_per_cu, uint32_t xrun)
{
    std::vector<ggq::Desc> d = P.descs;
    uint64_t groups = 0;
    for (auto& x : d) { x.first_group = groups; groups += (x.n_blocks + G - 1) / G; }
    ggq::Desc* dt; HIP_CHECK(hipMalloc(&dt, d.size() * sizeof(ggq::Desc)));
    HIP_CHECK(hipMemcpy(dt, d.data(), d.size() * sizeof(ggq::Desc), hipMemcpyHostToDevice));
    ab.to_free.push_back(dt);
    const uint32_t grid = 256u * waves_per_cu, n = (uint32_t)d.size();
    uint8_t* trash = stream_trash();
    char buf[160];
    snprintf(buf, sizeof buf, "%s stream G=%d ntl=%d D=%d waves/CU=%u xrun=%u", name, G, (int)NTL, D, waves_per_cu, xrun);
    ab.v.push_back(ABVariant{buf, [=] { hipLaunchKernelGGL((ggq::dequant_many_stream<F, G, ggq::OUT_F16, NTL, true, D>), dim3(grid), dim3(64), 0, nullptr, dt, n, groups, xrun, nullptr, 0u, trash); },
                             (double)P.elements * (2.0 + (double)P.ts / P.bs), {}, check_stream<F, G, NTL, D>()});
}

template <class F, int G, bool NTL, bool COOP_BASE>
static void ab_stream(const char* name, int qi, int pairs)
{
    Pool P = make_pool(QTS[qi], pairs);
    printf("POOL %s pairs=%d\n", name, pairs);
    AB ab;
    if (COOP_BASE) ab_add<F, 2 * G, NTL, true, 4, 0, false, -1, 1, true>(ab, name, P, 0, 5);      // the shipped team shape
    else ab_add<F, G, NTL, true, 1, 0, false, -1>(ab, name, P, F::ID == 14 ? 4096 : 0, 6);      // Q6_K ships with the 4 KiB occupancy pad
    const char* only = getenv("GGQ_STREAM_ONLY");                                                   // e.g. "D4" to cut the sweep
    if (!only || strstr(only, "D2")) { ab_add_stream<F, G, NTL, 2>(ab, name, P, 8, 6); ab_add_stream<F, G, NTL, 2>(ab, name, P, 16, 6); ab_add_stream<F, G, NTL, 2>(ab, name, P, 24, 6); }
    if (!only || strstr(only, "D4")) { ab_add_stream<F, G, NTL, 4>(ab, name, P, 4, 6); ab_add_stream<F, G, NTL, 4>(ab, name, P, 8, 6); ab_add_stream<F, G, NTL, 4>(ab, name, P, 12, 6); ab_add_stream<F, G, NTL, 4>(ab, name, P, 16, 6); }
    if (!only || strstr(only, "D6")) { ab_add_stream<F, G, NTL, 6>(ab, name, P, 4, 6); ab_add_stream<F, G, NTL, 6>(ab, name, P, 8, 6); }
    ab.run(9, 3);
    free_pool(P);
}

static void ab_stream_all()
{
    ab_stream<ggq::FmtQ4_K, 8, true, true>("Q4_K", 7, 64);
    ab_stream<ggq::FmtQ2_K, 8, false, true>("Q2_K", 5, 64);
    ab_stream<ggq::FmtQ4_0, 64, true, true>("Q4_0", 0, 64);
    ab_stream<ggq::FmtQ8_0, 64, true, true>("Q8_0", 4, 64);
    ab_stream<ggq::FmtQ6_K, 8, false, false>("Q6_K", 9, 64);
}

// ---- load cache policy: buffer loads with sc0 / sc1 / nt against the shipped global loads ---------------------------
template <class F, int G, int WAVES, bool COOP, int LPOL>
static bool check_lpol()
{
    const QT* q = nullptr;
    for (const QT& x : QTS) if (x.id == F::ID) q = &x;
    bool ok = true;
    for (uint64_t n : {(uint64_t)1, (uint64_t)(G - 1), (uint64_t)G, (uint64_t)(3 * G + 1), (uint64_t)(F::BS == 32 ? 20011 : 2503)}) {
        std::vector<uint8_t> packed; make_blocks(*q, n, (int)(n & 1), packed);
        std::vector<uint16_t> want(n * F::BS), got(n * F::BS);
        ggq_oracle_dequant_f16(F::ID, packed.data(), n, want.data());
        uint8_t *dp, *dout;
        HIP_CHECK(hipMalloc(&dp, packed.size())); HIP_CHECK(hipMalloc(&dout, n * F::BS * 2 + 256));
        HIP_CHECK(hipMemcpy(dp, packed.data(), packed.size(), hipMemcpyHostToDevice));
        HIP_CHECK(hipMemset(dout, 0xCD, n * F::BS * 2 + 256));
        const uint64_t groups = (n + G - 1) / G;
        const ggq::Desc tab[1] = {{dp, dout, n, 0}};
        ggq::Desc* dt; HIP_CHECK(hipMalloc(&dt, sizeof tab)); HIP_CHECK(hipMemcpy(dt, tab, sizeof tab, hipMemcpyHostToDevice));
        hipLaunchKernelGGL((ggq::lab::dequant_many<F, G, ggq::OUT_F16, false, true, WAVES, 0, false, -1, 1, ggq::AR_F16, COOP, LPOL>), dim3((uint32_t)(COOP ? groups : (groups + WAVES - 1) / WAVES)), dim3(WAVES * 64), 0, nullptr,
                           dt, 1u, groups, 0u, nullptr, 0u);
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipMemcpy(got.data(), dout, n * F::BS * 2, hipMemcpyDeviceToHost));
        uint8_t guard[256]; HIP_CHECK(hipMemcpy(guard, dout + n * F::BS * 2, 256, hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < n * F::BS; i++) ok &= canon16(got[i]) == canon16(want[i]);
        for (int g = 0; g < 256; g++) ok &= guard[g] == 0xCD;
        HIP_CHECK(hipFree(dp)); HIP_CHECK(hipFree(dout)); HIP_CHECK(hipFree(dt));
    }
    return ok;
}

template <class F, int G, int WAVES, bool COOP, int LPOL>
static void ab_add_lpol(AB& ab, const char* name, Pool& P, uint32_t xrun)
{
    std::vector<ggq::Desc> d = P.descs;
    uint64_t groups = 0;
    for (auto& x : d) { x.first_group = groups; groups += (x.n_blocks + G - 1) / G; }
    ggq::Desc* dt; HIP_CHECK(hipMalloc(&dt, d.size() * sizeof(ggq::Desc)));
    HIP_CHECK(hipMemcpy(dt, d.data(), d.size() * sizeof(ggq::Desc), hipMemcpyHostToDevice));
    ab.to_free.push_back(dt);
    const uint32_t blocks = (uint32_t)(COOP ? groups : (groups + WAVES - 1) / WAVES), n = (uint32_t)d.size();
    char buf[160];
    snprintf(buf, sizeof buf, "%s %s G=%d buffer loads%s%s%s xrun=%u", name, COOP ? "coop" : "solo", G, (LPOL & 1) ? " sc0" : "", (LPOL & 16) ? " sc1" : "", (LPOL & 2) ? " nt" : "", xrun);
    ab.v.push_back(ABVariant{buf, [=] { hipLaunchKernelGGL((ggq::lab::dequant_many<F, G, ggq::OUT_F16, false, true, WAVES, 0, false, -1, 1, ggq::AR_F16, COOP, LPOL>), dim3(blocks), dim3(WAVES * 64), 0, nullptr, dt, n, groups, xrun, nullptr, 0u); },
                             (double)P.elements * (2.0 + (double)P.ts / P.bs), {}, check_lpol<F, G, WAVES, COOP, LPOL>()});
}

template <class F, int G, bool NTL, bool COOP>
static void ab_load_policy(const char* name, int qi, uint32_t xr)
{
    Pool P = make_pool(QTS[qi], 64);
    printf("POOL %s pairs=64\n", name);
    AB ab;
    constexpr int W = COOP ? 4 : 1;
    if (COOP) ab_add<F, G, NTL, true, 4, 0, false, -1, 1, true>(ab, name, P, 0, xr);                  // shipped: global loads
    else ab_add<F, G, NTL, true, 1, 0, false, -1>(ab, name, P, F::ID == 14 ? 4096 : 0, xr);
    ab_add_lpol<F, G, W, COOP, 0>(ab, name, P, xr);
    ab_add_lpol<F, G, W, COOP, 2>(ab, name, P, xr);
    ab_add_lpol<F, G, W, COOP, 1>(ab, name, P, xr);
    ab_add_lpol<F, G, W, COOP, 16>(ab, name, P, xr);
    ab_add_lpol<F, G, W, COOP, 17>(ab, name, P, xr);
    ab_add_lpol<F, G, W, COOP, 18>(ab, name, P, xr);
    ab.run(9, 3);
    free_pool(P);
}

// ---- store cache policy on WHOLE-POOL launches (round 4): buffer stores with every aux combination against the shipped non-temporal global stores.
// (Round 3 swept the policies for layer-sized launches only; the buffer-store builtin makes the sweep one template argument.)
template <class F, int G, int WAVES, bool COOP, bool NTL, int SPOL>
static void ab_add_spol(AB& ab, const char* name, Pool& P, uint32_t xrun)
{
    std::vector<ggq::Desc> d = P.descs;
    uint64_t groups = 0;
    for (auto& x : d) { x.first_group = groups; groups += (x.n_blocks + G - 1) / G; }
    ggq::Desc* dt; HIP_CHECK(hipMalloc(&dt, d.size() * sizeof(ggq::Desc)));
    HIP_CHECK(hipMemcpy(dt, d.data(), d.size() * sizeof(ggq::Desc), hipMemcpyHostToDevice));
    ab.to_free.push_back(dt);
    const uint32_t blocks = (uint32_t)(COOP ? groups : (groups + WAVES - 1) / WAVES), n = (uint32_t)d.size();
    char buf[160];
    snprintf(buf, sizeof buf, "%s %s G=%d buffer STORES aux=%d:%s%s%s%s xrun=%u", name, COOP ? "coop" : "solo", G, SPOL, (SPOL & 1) ? " sc0" : "", (SPOL & 16) ? " sc1" : "", (SPOL & 2) ? " nt" : "",
             SPOL ? "" : " plain", xrun);
    ab.v.push_back(ABVariant{buf, [=] { hipLaunchKernelGGL((ggq::lab::dequant_many<F, G, ggq::OUT_F16, NTL, true, WAVES, 0, false, -1, 1, ggq::AR_F16, COOP, -1, SPOL>), dim3(blocks), dim3(WAVES * 64), 0, nullptr, dt, n, groups, xrun, nullptr, 0u); },
                             (double)P.elements * (2.0 + (double)P.ts / P.bs), {}, true});
}

template <class F, int G, bool NTL>
static void ab_store_policy(const char* name, int qi, uint32_t xr)
{
    Pool P = make_pool(QTS[qi], 64);
    printf("POOL %s pairs=64\n", name);
    AB ab;
    ab_add<F, G, NTL, true, 4, 0, false, -1, 1, true>(ab, name, P, 0, xr);                  // shipped: non-temporal GLOBAL stores
    ab_add_spol<F, G, 4, true, NTL, 2>(ab, name, P, xr);                                    // the same policy through the buffer path (one address VGPR)
    ab_add_spol<F, G, 4, true, NTL, 0>(ab, name, P, xr);
    ab_add_spol<F, G, 4, true, NTL, 1>(ab, name, P, xr);
    ab_add_spol<F, G, 4, true, NTL, 16>(ab, name, P, xr);
    ab_add_spol<F, G, 4, true, NTL, 17>(ab, name, P, xr);
    ab_add_spol<F, G, 4, true, NTL, 18>(ab, name, P, xr);
    ab_add_spol<F, G, 4, true, NTL, 3>(ab, name, P, xr);
    ab_add_spol<F, G, 4, true, NTL, 19>(ab, name, P, xr);
    ab.run(9, 3);
    free_pool(P);
}

// ---- round 4: the group's packed bytes by LDS-DMA (global_load_lds_dwordx4) instead of global_load -> VGPR -> ds_write_b128
template <class F, int G, int WAVES, bool COOP, bool NTL>
static void ab_add_dma(AB& ab, const char* name, Pool& P, uint32_t xrun)
{
    std::vector<ggq::Desc> d = P.descs;
    uint64_t groups = 0;
    for (auto& x : d) { x.first_group = groups; groups += (x.n_blocks + G - 1) / G; }
    ggq::Desc* dt; HIP_CHECK(hipMalloc(&dt, d.size() * sizeof(ggq::Desc)));
    HIP_CHECK(hipMemcpy(dt, d.data(), d.size() * sizeof(ggq::Desc), hipMemcpyHostToDevice));
    ab.to_free.push_back(dt);
    const uint32_t blocks = (uint32_t)(COOP ? groups : (groups + WAVES - 1) / WAVES), n = (uint32_t)d.size();
    // parity of this path on a small tensor with a ragged tail, against the shipped engine's output
    bool ok = true;
    {
        const uint64_t nb = (uint64_t)G * 37 + 5;
        uint8_t *dp, *o1, *o2;
        HIP_CHECK(hipMalloc(&dp, nb * F::TS + 64)); HIP_CHECK(hipMalloc(&o1, nb * F::BS * 2)); HIP_CHECK(hipMalloc(&o2, nb * F::BS * 2));
        HIP_CHECK(hipMemcpy(dp, P.descs[0].packed, nb * F::TS, hipMemcpyDeviceToDevice));
        ggq::Desc one{dp, o1, nb, 0}, two{dp, o2, nb, 0};
        ggq::Desc* t2; HIP_CHECK(hipMalloc(&t2, 2 * sizeof(ggq::Desc)));
        HIP_CHECK(hipMemcpy(t2, &one, sizeof one, hipMemcpyHostToDevice)); HIP_CHECK(hipMemcpy(t2 + 1, &two, sizeof two, hipMemcpyHostToDevice));
        const uint64_t g1 = (nb + G - 1) / G;
        const uint32_t b1 = (uint32_t)(COOP ? g1 : (g1 + WAVES - 1) / WAVES);
        hipLaunchKernelGGL((ggq::lab::dequant_many<F, G, ggq::OUT_F16, NTL, true, WAVES, 0, false, -1, 1, ggq::AR_F16, COOP>), dim3(b1), dim3(WAVES * 64), 0, nullptr, t2, 1u, g1, 0u, nullptr, 0u);
        hipLaunchKernelGGL((ggq::lab::dequant_many<F, G, ggq::OUT_F16, NTL, true, WAVES, 0, false, -1, 1, ggq::AR_F16, COOP, -1, -1, true>), dim3(b1), dim3(WAVES * 64), 0, nullptr, t2 + 1, 1u, g1, 0u, nullptr, 0u);
        HIP_CHECK(hipDeviceSynchronize());
        std::vector<uint16_t> a(nb * F::BS), b(nb * F::BS);
        HIP_CHECK(hipMemcpy(a.data(), o1, a.size() * 2, hipMemcpyDeviceToHost)); HIP_CHECK(hipMemcpy(b.data(), o2, b.size() * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < a.size(); i++) ok &= a[i] == b[i];
        HIP_CHECK(hipFree(dp)); HIP_CHECK(hipFree(o1)); HIP_CHECK(hipFree(o2)); HIP_CHECK(hipFree(t2));
    }
    char buf[160];
    snprintf(buf, sizeof buf, "%s %s G=%d waves=%d LDS-DMA staging ntl=%d xrun=%u", name, COOP ? "coop" : "solo", G, WAVES, (int)NTL, xrun);
    ab.v.push_back(ABVariant{buf, [=] { hipLaunchKernelGGL((ggq::lab::dequant_many<F, G, ggq::OUT_F16, NTL, true, WAVES, 0, false, -1, 1, ggq::AR_F16, COOP, -1, -1, true>), dim3(blocks), dim3(WAVES * 64), 0, nullptr, dt, n, groups, xrun, nullptr, 0u); },
                             (double)P.elements * (2.0 + (double)P.ts / P.bs), {}, ok});
}

static void ab_dma()
{
    { Pool P = make_pool(QTS[7], 64); printf("POOL Q4_K pairs=64\n"); AB ab;
      ab_add<ggq::FmtQ4_K, 16, true, true, 4, 0, false, -1, 1, true>(ab, "Q4_K", P, 0, 5);
      ab_add_dma<ggq::FmtQ4_K, 16, 4, true, true>(ab, "Q4_K", P, 5);
      ab_add_dma<ggq::FmtQ4_K, 16, 4, true, false>(ab, "Q4_K", P, 5);
      ab_add<ggq::FmtQ4_K, 8, true, true, 1, 0, false, -1, 1, false>(ab, "Q4_K", P, 0, 6);
      ab_add_dma<ggq::FmtQ4_K, 8, 1, false, true>(ab, "Q4_K", P, 6);
      ab.run(12, 4); free_pool(P); }
    { Pool P = make_pool(QTS[4], 64); printf("POOL Q8_0 pairs=64\n"); AB ab;
      ab_add<ggq::FmtQ8_0, 128, true, true, 4, 0, false, -1, 1, true>(ab, "Q8_0", P, 0, 5);
      ab_add_dma<ggq::FmtQ8_0, 128, 4, true, true>(ab, "Q8_0", P, 5);
      ab_add_dma<ggq::FmtQ8_0, 128, 4, true, false>(ab, "Q8_0", P, 5);
      ab.run(12, 4); free_pool(P); }
    { Pool P = make_pool(QTS[9], 64); printf("POOL Q6_K pairs=64\n"); AB ab;
      ab_add<ggq::FmtQ6_K, 8, false, true, 1, 0, false, -1>(ab, "Q6_K", P, 4096, 6);
      ab_add_dma<ggq::FmtQ6_K, 8, 1, false, false>(ab, "Q6_K", P, 6);
      ab.run(12, 4); free_pool(P); }
}

static void ab_load_policy_all()
{
    ab_load_policy<ggq::FmtQ4_K, 16, true, true>("Q4_K", 7, 5);
    ab_load_policy<ggq::FmtQ2_K, 16, false, true>("Q2_K", 5, 5);
    ab_load_policy<ggq::FmtQ3_K, 8, false, false>("Q3_K", 6, 0);
    ab_load_policy<ggq::FmtQ6_K, 8, false, false>("Q6_K", 9, 6);
    ab_load_policy<ggq::FmtQ8_0, 128, true, true>("Q8_0", 4, 5);
}

int main(int argc, char** argv)
{
    const std::string what = argc > 1 ? argv[1] : "all";
    hipDeviceProp_t prop; HIP_CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  CUs=%d  clock=%d MHz  mem=%.0f GB\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000, prop.totalGlobalMem / 1e9);
    int rc = 0;
    if (what == "parity" || what == "all") rc |= parity();
    if (what == "ceil" || what == "all") ceilings();
    if (what == "formats" || what == "all") formats();
    if (what == "variants" || what == "all") variants();
    if (what == "pmc") pmc_sequence();
    if (what == "skel") skeletons();
    if (what == "pmc2") pmc2_sequence();
    if (what == "pmc3") pmc3_sequence();
    if (what == "launchcost") launch_cost();
    if (what == "ablayer") {
        // FLUX 3072x3072 / 9216x3072 / 12288x3072 / 21504x3072, T5 4096x4096, SD3.5 2432x2432 (as 23 x 256 x 1024) and 7296x2432, a small one
        ab_layer<ggq::FmtQ4_K, 8>("Q4_K", 7, {1ull << 21, 23ull * 256 * 1024, 3072ull * 3072, 4096ull * 4096, 9216ull * 3072, 12288ull * 3072, 21504ull * 3072});
        ab_layer<ggq::FmtQ5_0, 64>("Q5_0", 2, {2432ull * 2432, 7296ull * 2432, 3072ull * 3072});
        ab_layer<ggq::FmtQ8_0, 64>("Q8_0", 4, {3072ull * 3072, 4096ull * 4096, 12288ull * 3072});
    }
    if (what == "ablayer2") {     // the formats whose whole-model shape is not the 4-wave team: do they want the layer-sized shape too?
        ab_layer<ggq::FmtQ3_K, 8>("Q3_K", 6, {3072ull * 3072, 4096ull * 4096, 9216ull * 3072});
        ab_layer<ggq::FmtQ6_K, 8>("Q6_K", 9, {3072ull * 3072, 4096ull * 4096, 9216ull * 3072});
        ab_layer<ggq::FmtQ2_K, 8>("Q2_K", 5, {3072ull * 3072, 4096ull * 4096});
        ab_layer<ggq::FmtIQ4_XS, 8>("IQ4_XS", 11, {3072ull * 3072, 4096ull * 4096});
    }
    if (what == "ablayer3") {     // round 3: layer-sized launches with bf16 result and write-through stores: FLUX, T5 and SD3.5 layer sizes
        ab_layer_bf16<ggq::FmtQ4_K, 8>("Q4_K", 7, {3072ull * 3072, 4096ull * 4096, 9216ull * 3072, 12288ull * 3072, 10240ull * 4096, 3072ull * 15360, 18432ull * 3072, 21504ull * 3072});
        ab_layer_bf16<ggq::FmtQ5_0, 64>("Q5_0", 2, {2432ull * 2432, 7296ull * 2432, 9728ull * 2432, 14592ull * 2432});
    }
    if (what == "abq3k") ab_q3k_line_exact();
    if (what == "abdpp") ab_dpp();
    if (what == "abdma") ab_dma();
    if (what == "abstore") { ab_store_policy<ggq::FmtQ4_K, 16, true>("Q4_K", 7, 5); ab_store_policy<ggq::FmtQ8_0, 128, true>("Q8_0", 4, 5); }
    if (what == "ceillayer") layer_ceiling();
    if (what == "ablayer4") {     // round 4: workgroup counts that divide evenly over the CUs
        ab_layer_balance<ggq::FmtQ4_K, 8>("Q4_K", 7, {3072ull * 3072, 9216ull * 3072, 12288ull * 3072, 3072ull * 15360, 18432ull * 3072, 21504ull * 3072});
        ab_layer_balance<ggq::FmtQ5_K, 8>("Q5_K", 8, {9216ull * 3072});
    }
    if (what == "ablds") {          // round 3: LDS-staged vs no-LDS ("direct", zero bank conflicts by construction) for the 2-byte-aligned legacy formats
        ab_big<ggq::FmtQ4_0, 64>("Q4_0", 0, 64);
        ab_big<ggq::FmtQ8_0, 64>("Q8_0", 4, 64);
    }
    if (what == "ab") ab_all();
    if (what == "abxcd") ab_xcd_all();
    if (what == "ceilx") ceilings_x();
    if (what == "abcoop") ab_coop_all();
    if (what == "absmall") ab_small_all();
    if (what == "abnt") ab_nt_all();
    if (what == "ablocate") ab_locate();
    if (what == "abocc") ab_occ1_all();
    if (what == "fillrows") fill_rows();
    if (what == "fillpol") fill_policy();
    if (what == "abstream") ab_stream_all();
    if (what == "abload") ab_load_policy_all();
    return rc ? 1 : 0;
}
