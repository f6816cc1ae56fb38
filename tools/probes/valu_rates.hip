// valu_rates.hip -- issue rate of the vector instructions the fused-linear decode is made of, on gfx950, per SIMD, at 1 / 2 / 4 / 8 waves per SIMD.
//
// Why: MI355X_MICROARCH.md prices v_fma_f32 at 2 cycles per wave-instruction (SIMD-32), while the round-4 counter reading of ggq_linear_small
// assumed 4 (one quad-cycle) for every VALU instruction.  Which of v_perm_b32 / v_pk_*_f16 / v_cvt_* / v_dot2c_* run at the full rate decides
// what the VALU floor of the decode is and which ops are worth removing.
//
// Method: every wave runs N trips of 32 instructions of ONE kind over 8 independent register chains (a dependent instruction is 8 issues away);
// a workgroup is 256 x W threads (W waves on each SIMD of its CU), one workgroup per CU (W = 8: two of 1024).  Cycles = s_memtime ticks of
// wave 0 of workgroup 0 scaled by (shader clock / memtime clock) measured from the v_fma_f32 arm... no scaling is assumed: reported is
// ticks per instruction per SIMD AND the ratio to v_fma_f32 at the same W, which is what matters.
//
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates tools/probes/valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));

enum Op : int { FMA_F32, AND_B32, LSHR_B32, PERM_B32, AND_OR_B32, BFI_B32, PK_MUL_F16, PK_ADD_F16, PK_FMA_F16, CVT_F32_F16, CVT_PK_BF16_F32, DOT2_F16, DOT2_BF16,
                PK_MUL_F32, MOV_DPP, CNDMASK, CVT_F16_U16_SDWA, CVT_F32_UBYTE, PK_ADD_U16, LSHL_OR, N_OPS };
static const char* NAMES[N_OPS] = {"v_fma_f32", "v_and_b32", "v_lshrrev_b32", "v_perm_b32", "v_and_or_b32", "v_bfi_b32", "v_pk_mul_f16", "v_pk_add_f16", "v_pk_fma_f16",
                                   "v_cvt_f32_f16", "v_cvt_pk_bf16_f32", "v_dot2c_f32_f16", "v_dot2c_f32_bf16", "v_pk_mul_f32", "v_mov_b32_dpp", "v_cndmask_b32",
                                   "v_cvt_f16_u16_sdwa", "v_cvt_f32_ubyte1", "v_pk_add_u16", "v_lshl_or_b32"};

template <int OP>
__device__ __forceinline__ void one(uint32_t& a, uint32_t b, uint32_t c, f2& p)
{
    if constexpr (OP == FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    else if constexpr (OP == AND_B32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a) : "v"(b));
    else if constexpr (OP == LSHR_B32) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a));
    else if constexpr (OP == PERM_B32) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    else if constexpr (OP == AND_OR_B32) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    else if constexpr (OP == BFI_B32) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    else if constexpr (OP == PK_MUL_F16) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(a) : "v"(b));
    else if constexpr (OP == PK_ADD_F16) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(a) : "v"(b));
    else if constexpr (OP == PK_FMA_F16) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    else if constexpr (OP == CVT_F32_F16) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(a));
    else if constexpr (OP == CVT_PK_BF16_F32) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a) : "v"(b));
    else if constexpr (OP == DOT2_F16) a = __builtin_bit_cast(uint32_t, __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, b), __builtin_bit_cast(h2, c), __builtin_bit_cast(float, a), false));
    else if constexpr (OP == DOT2_BF16) a = __builtin_bit_cast(uint32_t, __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(s16x2, b), __builtin_bit_cast(s16x2, c), __builtin_bit_cast(float, a), false));
    else if constexpr (OP == PK_MUL_F32) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(p));
    else if constexpr (OP == MOV_DPP) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a));
    else if constexpr (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b) : );
    else if constexpr (OP == CVT_F16_U16_SDWA) asm volatile("v_cvt_f16_u16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2" : "+v"(a) : "v"(b));
    else if constexpr (OP == CVT_F32_UBYTE) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(a));
    else if constexpr (OP == PK_ADD_U16) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a) : "v"(b));
    else if constexpr (OP == LSHL_OR) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a) : "v"(b));
}

template <int OP>
__global__ __launch_bounds__(1024) void rate(uint32_t* out, uint64_t* ticks, int trips, uint32_t seed)
{
    uint32_t r[8];
    f2 p[8];
    for (int i = 0; i < 8; i++) {
        r[i] = (threadIdx.x * 2654435761u + i * 40503u + seed) & 0x3BFF3BFFu;      // finite fp16 pairs / small fp32
        p[i] = f2{1.0f + i, 1.0f};
    }
    const uint32_t b = (seed & 0x03FF03FFu) | 0x3C003C00u, c = 0x07060504u ^ (seed & 0x01010101u);
    const uint64_t t0 = wall_clock64();
    const uint64_t c0 = __builtin_readcyclecounter();
    for (int t = 0; t < trips; t++) {
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int i = 0; i < 8; i++) one<OP>(r[i], b, c, p[i]);
    }
    const uint64_t c1 = __builtin_readcyclecounter();
    const uint64_t t1 = wall_clock64();
    uint32_t acc = 0;
    for (int i = 0; i < 8; i++) acc ^= r[i] ^ __builtin_bit_cast(uint32_t, p[i].x);
    if (acc == 0x12345678u) out[threadIdx.x] = acc;                                   // keep the chains alive
    if (threadIdx.x == 0 && blockIdx.x == 0) { ticks[0] = t1 - t0; ticks[1] = c1 - c0; }
}

typedef void (*kern_t)(uint32_t*, uint64_t*, int, uint32_t);
template <int OP> static kern_t pick() { return rate<OP>; }

int main()
{
    kern_t K[N_OPS] = {pick<0>(), pick<1>(), pick<2>(), pick<3>(), pick<4>(), pick<5>(), pick<6>(), pick<7>(), pick<8>(), pick<9>(), pick<10>(), pick<11>(), pick<12>(),
                       pick<13>(), pick<14>(), pick<15>(), pick<16>(), pick<17>(), pick<18>(), pick<19>()};
    uint32_t* out;
    uint64_t* ticks;
    hipMalloc(&out, 4096);
    hipMalloc(&ticks, 16);
    int cus = 256, wall_khz = 0, clk_khz = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    const int trips = 4000;
    printf("{\"cus\": %d, \"wall_clock_khz\": %d, \"shader_clock_khz\": %d, \"trips\": %d, \"insts_per_trip\": 32, \"rows\": [\n", cus, wall_khz, clk_khz, trips);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    bool first = true;
    for (int W : {1, 2, 4, 8}) {
        const int threads = W >= 4 ? 1024 : 256 * W, per_cu = W == 8 ? 2 : 1;
        double base_ns = 0.0;
        for (int op = 0; op < N_OPS; op++) {
            for (int rep = 0; rep < 2; rep++) {                                       // rep 0 warms the code up
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(K[op], dim3(cus * per_cu), dim3(threads), 0, 0, out, ticks, trips, 12345u + op);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
            }
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            uint64_t h[2];
            hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
            const double insts_per_simd = (double)trips * 32.0 * W;
            const double ns_per_inst = (double)ms * 1e6 / insts_per_simd;            // per SIMD, launch overhead included (~1 %)
            if (op == FMA_F32) base_ns = ns_per_inst;
            printf("%s{\"waves_per_simd\": %d, \"op\": \"%s\", \"ns_per_inst_per_simd\": %.4f, \"vs_fma_f32\": %.3f, \"wave0_shader_cycles_per_own_inst\": %.3f, \"wave0_wall_ticks\": %llu}",
                   first ? "" : ",\n", W, NAMES[op], ns_per_inst, ns_per_inst / base_ns, (double)h[1] / ((double)trips * 32.0), (unsigned long long)h[0]);
            first = false;
        }
    }
    printf("\n]}\n");
    return 0;
}
