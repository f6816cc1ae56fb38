// ggq_lab_engine.hpp -- HARNESS ONLY (tests/microbench): the dequant engine with its experiment knobs.
//
// The shipped engine (comfyui-gguf_amd/csrc/ggq_device.hpp, `ggq::Engine`) carries only what the product launches.  The A/B
// experiments recorded under profiles/r01_microbench_* needed more: compile-time XCD mappings (XCD), the no-LDS engine (DIRECT),
// a store throttle (THR), several groups per wave walked strictly in sequence (R) and buffer loads with an explicit cache policy
// (LPOL).  They live here, in namespace ggq::lab, over the SAME formats / arithmetic / emit code of the product header -- so a
// variant measured here computes exactly the shipped values (each is parity-checked against the oracle by the harness).
#pragma once

#include "../../comfyui-gguf_amd/csrc/ggq_device.hpp"

namespace ggq {
namespace lab {

// Launch shape: one wavefront = one group, start to finish, then the wave retires and the
// hardware dispatcher starts the next workgroup -- no grid-stride loop.  (Measured on MI355X: a
// persistent loop with a register prefetch of the next group is 10-15 % SLOWER, because its
// s_waitcnt vmcnt(0) in front of the LDS fill also waits for the previous group's stores to be
// acknowledged -- gfx950 counts loads and stores in the one vmcnt; profiles/r01_*.)
//   F      block format            G      blocks per group
//   OUT    output dtype            NTL/NTS  non-temporal loads / stores
//   WAVES  wavefronts per workgroup (they share nothing but the LDS allocation)
// at most THR+1 store rows of a wave in flight (THR < 0: no throttle)
template <int THR>
GGQ_DEV void store_throttle()
{
    // s_waitcnt simm16 on gfx9: vmcnt = [3:0] | [15:14], expcnt = [6:4], lgkmcnt = [11:8]
    if constexpr (THR >= 0) __builtin_amdgcn_s_waitcnt((THR & 15) | ((THR >> 4) << 14) | 0x0F70);
}

// SKEW: the tensor's base pointer itself may be only 2-byte aligned (a row inside a packed table): the misalignment of every
// group start is then taken from the ADDRESS, not from the offset inside the tensor.
// LPOL >= 0 (harness only): the group's bytes are fetched with BUFFER loads carrying that cache policy (aux bits: 1 = sc0, 2 = nt,
// 16 = sc1) from a wave-uniform resource whose range ends at the tensor's last byte, instead of global loads with NTL.
// SPOL >= 0 (harness only): every store is a Window (buffer) store carrying that cache policy (same aux bits), whatever NTS says.
template <class F, int G, int OUT, bool NTL, bool NTS, int WAVES, int XCD = 0, bool DIRECT = false, int THR = -1, int R = 1, int ARITH = AR_F16, bool COOP = false,
          bool SKEW = false, int LPOL = -1, int SPOL = -1, bool DMA = false>
struct Engine {
    static constexpr int TS = F::TS, BS = F::BS;
    static constexpr int CPB = BS / 8;                 // chunks per block
    static constexpr int GROUP_BYTES = G * TS;
    // A group starts 16-B aligned when GROUP_BYTES % 16 == 0 (any G that is a multiple of 8);
    // otherwise its start is only MISALIGN_STEP-aligned and the wave loads from the aligned
    // address below it, keeping the same byte offset inside its LDS slice.
    static constexpr bool ALIGNED = GROUP_BYTES % 16 == 0 && !SKEW;
    static constexpr int UNITS = (GROUP_BYTES + (ALIGNED ? 0 : 14) + 15) / 16;   // 16-B load units per group (max)
    // TEAM = the threads that own one group: a wavefront, or (COOP) the whole workgroup -- then every wave stores ONE
    // 1-KiB row instead of four back to back, which the memory system takes 6 % faster (tests/microbench `fillrows`).
    static constexpr int TEAM = COOP ? WAVES * 64 : 64;
    static constexpr int NU = (UNITS + TEAM - 1) / TEAM;   // loads per thread
    static constexpr int CHUNKS = G * CPB;
    static constexpr int PIECES = Layout<OUT>::PIECES;  // lanes per chunk (2 for fp32 output: one quad each)
    static constexpr int NCH = CHUNKS * PIECES / TEAM; // stores per thread per group
    static constexpr int SLICE = NU * TEAM * 16;       // LDS bytes per team
    static constexpr int THREADS = WAVES * 64;
    static_assert(GROUP_BYTES % 2 == 0, "block formats are 2-byte aligned");
    static_assert(ALIGNED || (GROUP_BYTES % F::LDS_ALIGN == 0), "group start must keep the format's LDS read alignment");
    static_assert(!SKEW || (F::TS % F::LDS_ALIGN == 0 && !DIRECT), "a row start is a multiple of the block size only");
    static_assert((CHUNKS * PIECES) % TEAM == 0, "a group must be a whole number of 1 KiB store rows per wave");
    static_assert(!(COOP && DIRECT) && !(COOP && R != 1), "COOP is the LDS-staged single-pass engine");

    GGQ_DEV static void team_sync()
    {
        if constexpr (COOP) __syncthreads();             // s_barrier over the workgroup's waves
        else wave_sync();
    }

    // FULL = the whole group lies inside the tensor (wave-uniform): no per-lane bounds checks,
    // so the compiler batches the LDS reads of all NCH chunks.
    template <bool FULL>
    GGQ_DEV static void body(uint8_t* slice, const Work& w, int lane)      // lane = index inside the team
    {
        const uint64_t off = w.lg * (uint64_t)GROUP_BYTES;
        const uint32_t a = ALIGNED ? 0u : ((uint32_t)((SKEW ? (uint64_t)w.packed : 0ull) + off) & 15u);   // wave-uniform
        const gcptr base = w.packed + off - a;
        uint32_t valid = a + (uint32_t)GROUP_BYTES;
        if constexpr (!FULL) {
            const uint64_t left = w.n_blocks * (uint64_t)TS - off;           // > 0 by construction
            if (left < (uint64_t)GROUP_BYTES) valid = a + (uint32_t)left;
        }
        u32x4 pf[NU];
        if constexpr (DMA) {
            // round 4: the group's bytes go global -> LDS without passing through registers (global_load_lds_dwordx4: the LDS image of one wave-instruction
            // is lane-linear from a wave-uniform base, which is exactly the slice layout); masked lanes leave stale bytes nobody decodes
            const uint32_t wbase = (COOP ? ((uint32_t)threadIdx.x >> 6) * 64u : 0u);
#pragma unroll
            for (int u = 0; u < NU; u++) {
                const uint32_t o = (uint32_t)(lane + TEAM * u) * 16u;
                if ((FULL && ALIGNED && (u + 1) * TEAM <= UNITS) || o < valid)
                    __builtin_amdgcn_global_load_lds((const GGQ_GLOBAL void*)(base + o), (__attribute__((address_space(3))) void*)(slice + (wbase + (uint32_t)(TEAM * u)) * 16u), 16, 0, NTL ? 2 : 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if constexpr (LPOL >= 0) {
            // raw buffer: units past the last one that holds valid bytes read as zero, so neither a bounds check nor a 64-bit address
            // per lane (the range is rounded up to whole units: the same < 16-byte over-read inside an aligned unit as below)
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)((valid + 15u) & ~15u), 0x00020000);
#pragma unroll
            for (int u = 0; u < NU; u++) pf[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((uint32_t)(lane + TEAM * u) * 16u), 0, LPOL);
        } else
#pragma unroll
        for (int u = 0; u < NU; u++) {
            const uint32_t o = (uint32_t)(lane + TEAM * u) * 16u;
            if (FULL && ALIGNED && (u + 1) * TEAM <= UNITS) {
                pf[u] = gload16<NTL>(base + o);
            } else {
                // the last unit of a tensor may straddle its end: an aligned 16-B read never crosses
                // a page, so the over-read (< 16 B, inside the same aligned unit) cannot fault.
                pf[u] = (o < valid) ? gload16<NTL>(base + o) : u32x4{0, 0, 0, 0};
            }
        }
        if constexpr (!DMA) {
#pragma unroll
            for (int u = 0; u < NU; u++) *reinterpret_cast<u32x4*>(slice + (lane + TEAM * u) * 16) = pf[u];
        }
        team_sync();
        const uint64_t b0 = w.lg * (uint64_t)G;
        // (NTS = false: the product's write-through Window stores, ggq_device.hpp)
        constexpr uint32_t OB = (uint32_t)OutBytes<OUT>::V;
        [[maybe_unused]] const Window win = window(w.out + b0 * (uint64_t)(BS * OB), (uint32_t)(G * BS) * OB);
#pragma unroll
        for (int s = 0; s < NCH; s++) {
            const int unit = lane + TEAM * s;
            const int chunk = unit / PIECES, piece = unit % PIECES;
            const int bl = chunk / CPB, j = chunk % CPB;
            const uint64_t gb = b0 + (uint64_t)bl;
            if (FULL || gb < w.n_blocks) {
                const Fields f = F::template fields<true>(slice + a + bl * TS, j);
                if constexpr (SPOL >= 0) emit_to<F, ARITH, OUT>(f, piece, [&](auto v) { wstore<SPOL>(win, (uint32_t)(bl * BS + j * 8 + piece * Layout<OUT>::ELEMS) * OB, v); });
                else if constexpr (NTS) emit<F, ARITH, OUT, true>(f, piece, w.out, gb * (uint64_t)BS + (uint64_t)(j * 8 + piece * Layout<OUT>::ELEMS));
                else emit_to<F, ARITH, OUT>(f, piece, [&](auto v) { wstore(win, (uint32_t)(bl * BS + j * 8 + piece * Layout<OUT>::ELEMS) * OB, v); });
            }
            if (s + 1 < NCH) store_throttle<THR>();
        }
    }

    // DIRECT: no LDS staging -- every lane reads the few bytes its chunk needs straight from
    // global memory (the same F::fields decode, pointed at the packed bytes).  A wave-row of 64 chunks
    // touches 3-5 cache lines; neighbouring lanes share them through the vector L1.
    template <bool FULL>
    GGQ_DEV static void body_direct(const Work& w, int lane)
    {
        const uint64_t b0 = w.lg * (uint64_t)G;
#pragma unroll
        for (int s = 0; s < NCH; s++) {
            const int unit = lane + 64 * s;
            const int chunk = unit / PIECES, piece = unit % PIECES;
            const int bl = chunk / CPB, j = chunk % CPB;
            const uint64_t gb = b0 + (uint64_t)bl;
            if (FULL || gb < w.n_blocks) {
                const Fields f = F::template fields<false>((const uint8_t*)(w.packed + gb * (uint64_t)TS), j);
                emit<F, ARITH, OUT, NTS>(f, piece, w.out, gb * (uint64_t)BS + (uint64_t)(j * 8 + piece * Layout<OUT>::ELEMS));
            }
        }
    }

    // xrun_log2 (wave-uniform kernel argument, 0 = off): the run-length form of the XCD >= 2 mapping below,
    // chosen per launch by the host (it pays on large launches only; profiles/r01_microbench_l_*).
    template <class Locate>
    GGQ_DEV static void run(uint64_t total_groups, uint32_t xrun_log2, Locate locate)
    {
        // (the host may add untouched DYNAMIC LDS to a launch: it only caps how many workgroups a CU holds at once)
        __shared__ __attribute__((aligned(16))) uint8_t smem[DIRECT ? 16 : (COOP ? 1 : WAVES) * SLICE];
        const int wave = COOP ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        const int lane = COOP ? (int)threadIdx.x : (int)(threadIdx.x & 63);
        uint32_t bid = blockIdx.x;
        // workgroup b runs on XCD b % 8 (observed dispatch order; a speed hint only, never correctness)
        if constexpr (XCD == 1) {
            // give each XCD one contiguous eighth of the work instead of every eighth workgroup
            const uint32_t nb = gridDim.x, q = nb >> 3, r = nb & 7u, x = bid & 7u;
            bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
        } else if constexpr (XCD >= 2) {
            // runs: inside every tile of 8*XCD consecutive workgroups, XCD x takes XCD CONSECUTIVE groups, so
            // the 128-B line two neighbouring groups share (formats whose group size is not a multiple of
            // 128 B) is fetched into ONE L2 instead of two; the tile as a whole still streams 8*XCD*4 KiB
            // of output together.  The last, partial tile keeps the identity mapping.
            constexpr uint32_t T = 8u * (uint32_t)XCD;
            const uint32_t tile = bid / T, in = bid % T;
            if ((tile + 1) * T <= gridDim.x) bid = tile * T + (in & 7u) * (uint32_t)XCD + (in >> 3);
        } else if (xrun_log2 != 0) {
            // What matters (profiles/r01_microbench_l/m/n): every XCD keeps the SAME slot of every tile (rotating the slot
            // with the tile index loses 10 %; which XCD gets which slot is irrelevant, as is the walk direction).
            const uint32_t tl = xrun_log2 + 3u, tile = bid >> tl, in = bid & ((1u << tl) - 1u);
            if (((tile + 1) << tl) <= gridDim.x) bid = (tile << tl) + ((in & 7u) << xrun_log2) + (in >> 3);
        }
        const uint64_t g = COOP ? (uint64_t)bid : (uint64_t)bid * WAVES + (uint64_t)wave;
        if (g >= total_groups) return;
        const Work w0 = locate(g);                       // lg counts units of R*G blocks
        uint8_t* slice = smem + (DIRECT ? 0 : wave * SLICE);          // COOP: wave == 0, one slice per workgroup
        // R > 1: the wave walks R consecutive groups strictly one after the other -- load, unpack,
        // store, wait for the store -- so it never has more than one group's traffic in flight.
#pragma unroll 1
        for (int r = 0; r < R; r++) {
            Work w = w0;
            w.lg = w0.lg * (uint64_t)R + (uint64_t)r;
            if (r > 0 && w.lg * (uint64_t)G >= w.n_blocks) break;
            const bool full = (w.lg + 1) * (uint64_t)G <= w.n_blocks;
            if constexpr (DIRECT) {
                if (full) body_direct<true>(w, lane); else body_direct<false>(w, lane);
            } else {
                if (full) body<true>(slice, w, lane); else body<false>(slice, w, lane);
            }
            if (r + 1 < R) {
                store_throttle<0>();
                wave_sync();
            }
        }
    }
};

// one tensor, descriptor by value
template <class F, int G, int OUT, bool NTL, bool NTS, int WAVES, int XCD = 0, bool DIRECT = false, int THR = -1, int R = 1, int ARITH = AR_F16, bool COOP = false>
__global__ __launch_bounds__(WAVES * 64) void dequant_one(Desc d, uint64_t total_groups, uint32_t xrun_log2)
{
    Engine<F, G, OUT, NTL, NTS, WAVES, XCD, DIRECT, THR, R, ARITH, COOP>::run(total_groups, xrun_log2, [&](uint64_t g) { return Work{(gcptr)d.packed, (gptr)d.out, d.n_blocks, g}; });
}

// many tensors of one format: table in device memory, sorted by first_group.  Finding the tensor of group g is a chain
// of DEPENDENT scalar loads at the head of every wave: `coarse[c]` (optional) = the entry that holds group c << coarse_shift,
// which leaves a 1-2 step forward scan instead of a log2(n)-step binary search -- a team holds its wave slots idle during
// that chain, which costs the multi-wave (COOP) teams most (tests/microbench `ablocate`).
template <class F, int G, int OUT, bool NTL, bool NTS, int WAVES, int XCD = 0, bool DIRECT = false, int THR = -1, int R = 1, int ARITH = AR_F16, bool COOP = false,
          int LPOL = -1, int SPOL = -1, bool DMA = false>
__global__ __launch_bounds__(WAVES * 64) void dequant_many(const Desc* __restrict__ table, uint32_t n, uint64_t total_groups, uint32_t xrun_log2,
                                                           const uint32_t* __restrict__ coarse, uint32_t coarse_shift)
{
    Engine<F, G, OUT, NTL, NTS, WAVES, XCD, DIRECT, THR, R, ARITH, COOP, false, LPOL, SPOL, DMA>::run(total_groups, xrun_log2, [&](uint64_t g) {
        uint32_t lo = 0;                                // last entry with first_group <= g
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

// ---- round 4: lane-to-lane exchange (DPP) instead of LDS reads for bytes two lanes need (Next #8; north_star: "sub-byte
// nibble extraction via DPP/permlane").  Same values as the shipped formats (parity-checked by the harness); only where the bytes come from differs.
// The engine deals chunks to lanes in order (chunk = lane + 64 s), which is what makes fixed DPP patterns line up.

// Q4_K: chunks j and j + 4 unpack the LOW and the HIGH nibbles of the same 8 bytes (dequant.py:189-192).  Shipped: both lanes read them from LDS (a
// broadcast read).  Here: only the lanes of the even sub-blocks read; the lane four up gets the two dwords by `row_shr:4` and shifts its nibbles down.
struct FmtQ4_K_DPP : FmtQ4_K {
    template <bool LDS>
    GGQ_DEV static Fields fields(const uint8_t* b, int j)
    {
        const int sb = j >> 2;
        const u32x4 hdr = *reinterpret_cast<const u32x4*>(b);
        Fields f;
        k_scale_min(hdr, sb, f.sc, f.mn);
        u32x2 w{0u, 0u};
        if ((sb & 1) == 0) w = lds_ld8<8, LDS>(b + 16 + 32 * (sb >> 1) + 8 * (j & 3));
        const uint32_t sx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w.x, 0x114 /* row_shr:4 */, 0xF, 0xF, false);
        const uint32_t sy = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w.y, 0x114, 0xF, 0xF, false);
        const bool odd = (sb & 1) != 0;
        f.t0 = ((odd ? sx : w.x) >> (odd ? 4 : 0)) & 0x0F0F0F0Fu;
        f.t1 = ((odd ? sy : w.y) >> (odd ? 4 : 0)) & 0x0F0F0F0Fu;
        f.dm = hdr.x;
        return f;
    }
};

// Q8_0: a chunk's 8 bytes start 2 bytes into a 34-byte block, i.e. at a 2-byte-aligned address in every other block: the shipped decode reads the
// three enclosing aligned dwords and funnel-shifts.  The third dword is the first dword of the NEXT chunk's read whenever that chunk is in the same
// block: here it comes from the lane one up (`row_shl:1`), and only the last chunk of a block (whose neighbour starts a block of the other
// alignment) reads its own third dword.
struct FmtQ8_0_DPP : FmtQ8_0 {
    template <bool LDS>
    GGQ_DEV static Fields fields(const uint8_t* b, int j)
    {
        const uint8_t* p = b + 2 + 8 * j;
        const uint32_t sh = (uint32_t)reinterpret_cast<uintptr_t>(p) & 2u;
        const uint8_t* q = p - sh;
        const uint32_t d0 = lds_dword(q), d1 = lds_dword(q + 4);
        uint32_t d2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)d0, 0x101 /* row_shl:1: from the lane one up */, 0xF, 0xF, false);
        if (j == 3 && sh) d2 = lds_dword(q + 8);                   // (lane 15 of a DPP row always holds a j == 3 chunk: no row-edge case left)
        const uint32_t x = __builtin_amdgcn_alignbyte(d1, d0, sh), y = __builtin_amdgcn_alignbyte(d2, d1, sh);
        return Fields{x ^ 0x80808080u, y ^ 0x80808080u, lds_u16(b), 0, 0};
    }
};

}  // namespace lab
}  // namespace ggq
