// ggq_stream.hpp -- EXPERIMENT (harness only, not part of libggq_hip.so): the PERSISTENT form of the dequant engine.  A wave
// never retires; it walks its share of the groups with the packed bytes of the next D groups already in flight in registers.
//
// The question it answers ("What the persistent engine showed"): every format that moves <= 2.56 B per
// element runs at the same ~2.5 T elements/s whatever its read volume (Q2_K 2.33 B/el ... Q4_0 2.56 B/el) -- is that rate set
// by the one-shot engine's per-wave latency chain (load -> LDS -> decode -> store -> wait for the acknowledgement -> retire ->
// dispatch the next workgroup) times the number of wave slots?  Here the chain is cut:
//   * loads for group i+D are issued before group i is decoded (D register sets per wave; gfx950 counts loads and stores
//     in one in-order vmcnt, so waiting for the OLDEST load leaves the younger loads and the stores of the last D-1
//     groups in flight -- the compiler derives the s_waitcnt vmcnt(N) from the fully unrolled ring),
//   * a wave's stores are never waited for, and no wave slot sits idle through a store acknowledgement or a dispatch,
//   * the tensor lookup happens at issue time, D groups ahead of its use.
// The workgroup -> group mapping is the one-shot engine's, applied to VIRTUAL workgroup ids v = blockIdx.x + i * gridDim.x
// (gridDim.x a multiple of 8, so every id a workgroup takes falls on its own XCD's slot of the run mapping).
// ANSWER (profiles/r01_microbench_r_persistent_stream_engine.txt): no.  Bit-exact, and 17-25 % SLOWER than the shipped
// one-shot shapes at every ring depth (2 / 4 / 6) and every occupancy (4-24 waves per CU): 4.7-5.3 TB/s against 6.2-6.4;
// fewest waves fastest, depth irrelevant.  The rate is not a latency x occupancy product -- the dispatcher-ordered stream of
// short-lived workgroups, whose frontier advances in address order, is what the memory system likes.
#pragma once

#include "../../comfyui-gguf_amd/csrc/ggq_device.hpp"

#include <utility>

namespace ggq {

// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): a ring walk with STATIC slot indices (register sets)
template <class Fn, int... Is>
GGQ_DEV void static_for_impl(Fn&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class Fn>
GGQ_DEV void static_for(Fn&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// Every path through the steady-state loop issues EXACTLY NU loads and NCH stores per group -- no exec-masked branch around a
// memory instruction -- because the s_waitcnt pass takes the minimum over paths: one path with fewer operations would turn
// the vmcnt(N) in front of the LDS fill into a full drain.  So lanes past the end of a tensor load from a clamped (valid)
// address and store to `trash` (>= 4 KiB of device memory nobody reads) instead of being masked off.
template <class F, int G, int OUT, bool NTL, bool NTS, int D, int ARITH = AR_F16>
struct StreamEngine {
    using E = Engine<F, G, OUT, NTL, NTS, 1, ARITH, false>;
    static constexpr int TS = F::TS, BS = F::BS, CPB = E::CPB, NU = E::NU, NCH = E::NCH, SLICE = E::SLICE, PIECES = E::PIECES;
    static constexpr int GROUP_BYTES = E::GROUP_BYTES;
    static constexpr bool ALIGNED = E::ALIGNED;
    static_assert(D >= 1 && D * (NU + NCH) <= 56, "the ring must fit the 6-bit vmcnt");

    struct Slot {                    // one group in flight: its packed bytes (per lane) and where it goes (wave-uniform)
        u32x4 pf[NU];
        gptr out;
        uint64_t n_blocks, lg;
        uint32_t a;
    };

    template <class Locate>
    GGQ_DEV static void run(uint64_t total_groups, uint32_t xrun_log2, gptr trash, Locate locate)
    {
        __shared__ __attribute__((aligned(16))) uint8_t slice[SLICE];
        const int lane = (int)threadIdx.x;
        const uint64_t nv = total_groups;                                   // virtual workgroups = the one-shot grid (1 wave each)
        const uint64_t stride = gridDim.x;
        if (blockIdx.x >= nv) return;
        const uint64_t n_mine = (nv - blockIdx.x + stride - 1) / stride;    // groups this wave owns
        uint64_t v = blockIdx.x;

        auto issue = [&](Slot& s) {
            uint64_t g = v;
            if (xrun_log2 != 0) {
                const uint32_t tl = xrun_log2 + 3u;
                const uint64_t tile = g >> tl, in = g & ((1ull << tl) - 1ull);
                if (((tile + 1) << tl) <= nv) g = (tile << tl) + ((in & 7ull) << xrun_log2) + (in >> 3);
            }
            v += stride;
            const Work w = locate(g);
            const uint64_t off = w.lg * (uint64_t)GROUP_BYTES;
            const uint32_t a = ALIGNED ? 0u : ((uint32_t)off & 15u);
            const gcptr base = w.packed + off - a;
            const uint64_t left = w.n_blocks * (uint64_t)TS - off;
            const uint32_t valid = a + (left < (uint64_t)GROUP_BYTES ? (uint32_t)left : (uint32_t)GROUP_BYTES);
#pragma unroll
            for (int u = 0; u < NU; u++) {
                const uint32_t o = (uint32_t)(lane + 64 * u) * 16u;
                s.pf[u] = gload16<NTL>(base + (o < valid ? o : 0u));       // past the end: a valid address, bytes never used
            }
            s.out = w.out; s.n_blocks = w.n_blocks; s.lg = w.lg; s.a = a;
        };

        auto consume = [&](const Slot& s) {
#pragma unroll
            for (int u = 0; u < NU; u++) *reinterpret_cast<u32x4*>(slice + (lane + 64 * u) * 16) = s.pf[u];
            wave_sync();
            const uint64_t b0 = s.lg * (uint64_t)G;
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                const int unit = lane + 64 * c;
                const int chunk = unit / PIECES, piece = unit % PIECES;
                const int bl = chunk / CPB, j = chunk % CPB;
                const uint64_t gb = b0 + (uint64_t)bl;
                const bool inside = gb < s.n_blocks;
                const Fields f = F::template fields<true>(slice + s.a + bl * TS, j);
                emit<F, ARITH, OUT, NTS>(f, piece, inside ? s.out : trash,
                                          inside ? gb * (uint64_t)BS + (uint64_t)(j * 8 + piece * Layout<OUT>::ELEMS) : (uint64_t)(lane * Layout<OUT>::ELEMS));
            }
            wave_sync();                                                    // the slice is refilled by the next group
        };

        Slot ring[D];
        uint64_t issued = 0, consumed = 0;
        // steady state: whole turns of the ring in which every consume is followed by a live issue.  The first turn is
        // peeled so that the loop is entered with the same operations in flight as its back edge carries: the wait in front
        // of a slot's LDS fill then becomes vmcnt((D-1) * (NU + NCH)) instead of a drain.
        const uint64_t turns = n_mine > (uint64_t)D ? (n_mine - (uint64_t)D) / (uint64_t)D : 0;
        if (turns > 0) {
            static_for<D>([&](auto d) { issue(ring[d]); });
            static_for<D>([&](auto d) { consume(ring[d]); issue(ring[d]); });
            for (uint64_t t = 1; t < turns; t++)
                static_for<D>([&](auto d) { consume(ring[d]); issue(ring[d]); });
            issued = (turns + 1) * D; consumed = turns * D;
        } else {
            static_for<D>([&](auto d) { if (issued < n_mine) { issue(ring[d]); issued++; } });
        }
        // drain (fewer than 2 turns): same walk, issue only while ids remain
        while (consumed < n_mine) {
            static_for<D>([&](auto d) {
                if (consumed < n_mine) {
                    consume(ring[d]); consumed++;
                    if (issued < n_mine) { issue(ring[d]); issued++; }
                }
            });
        }
    }
};

template <class F, int G, int OUT, bool NTL, bool NTS, int D, int ARITH = AR_F16>
__global__ __launch_bounds__(64) void dequant_many_stream(const Desc* __restrict__ table, uint32_t n, uint64_t total_groups, uint32_t xrun_log2,
                                                          const uint32_t* __restrict__ coarse, uint32_t coarse_shift, uint8_t* __restrict__ trash)
{
    StreamEngine<F, G, OUT, NTL, NTS, D, ARITH>::run(total_groups, xrun_log2, (gptr)trash, [&](uint64_t g) {
        uint32_t lo = 0;
        if (coarse != nullptr) {
            lo = coarse[g >> coarse_shift];
            while (lo + 1 < n && table[lo + 1].first_group <= g) lo++;
        } else {
            uint32_t hi = n;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (table[mid].first_group <= g) lo = mid; else hi = mid;
            }
        }
        const Desc d = table[lo];
        return Work{(gcptr)d.packed, (gptr)d.out, d.n_blocks, g - d.first_group};
    });
}

}  // namespace ggq
