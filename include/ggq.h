/*
 * ggq.h -- C ABI of libggq_hip.so: MI355X (gfx950) GGUF block dequantization.
 *
 * This is the drop-in boundary for the one hot path of city96/ComfyUI-GGUF, dequant.py.
 * The reference has no FFI of its own (it is eager torch code); each entry point below names the
 * reference interface it stands in for, and INTEGRATION.md shows the ctypes binding a maintainer
 * adds on the reference side.  Plain pointers and sizes only -- no torch types cross this line.
 *
 * Conventions
 *   - every function returns a ggq_status (0 = GGQ_OK) and never throws, aborts or syncs the device;
 *   - `packed` / `out` are DEVICE pointers on the current HIP device; `packed` is borrowed and
 *     read-only, `out` is caller-allocated (torch's caching allocator owns it) and only filled;
 *   - `hip_stream` is a hipStream_t passed as void* (NULL = the legacy default stream); kernels are
 *     enqueued on it and the call returns immediately -- pass torch's CURRENT stream so the launch
 *     orders after the H2D copy before it and before the F.linear after it (ops.py:209-210,244);
 *   - re-entrant and thread-safe: no global mutable state except a per-device property cache;
 *   - `packed` and `out` must be 16-byte aligned (GGQ_ERR_ALIGN otherwise; torch allocations are);
 *   - `qtype` is ggml's public type id (== int(gguf.GGMLQuantizationType.X)):
 *       Q4_0=2 Q4_1=3 Q5_0=6 Q5_1=7 Q8_0=8 Q2_K=10 Q3_K=11 Q4_K=12 Q5_K=13 Q6_K=14 IQ4_NL=20 IQ4_XS=23.
 *
 * Numerics: bit-identical to the reference's eager op sequence -- in the default fp16 mode
 * (dequant_dtype=None, nodes.py:152-153) and in the float32 / bfloat16 modes of the Advanced loader
 * (nodes.py:186): every torch op of a block function is one correctly rounded op of that dtype here too.
 */
#ifndef GGQ_H
#define GGQ_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum ggq_status {
    GGQ_OK = 0,
    GGQ_ERR_QTYPE = 1,   /* qtype has no HIP unpacker (caller keeps the reference path, dequant.py:24-28) */
    GGQ_ERR_ALIGN = 2,   /* packed or out not 16-byte aligned */
    GGQ_ERR_ARG = 3,     /* NULL pointer with n_blocks > 0, bad compute/out dtype, bad descriptor table */
    GGQ_ERR_HIP = 4,     /* a HIP runtime call failed; ggq_last_hip_error() has the hipError_t */
    GGQ_ERR_NOMEM = 5    /* host or device allocation for a plan failed */
    /* 6, 7: GGQ_ERR_IO, GGQ_ERR_FORMAT of the GGUF reader (ggq_gguf.h) */
} ggq_status;

/* Floating-point dtypes, used for two independent choices of dequantize_tensor (dequant.py:15-23):
 *   compute_dtype  the reference's `dequant_dtype` (Advanced loader, nodes.py:186): the dtype the block
 *                  function's op sequence runs in.  F16 = the stock path (dequant_dtype None); BF16 /
 *                  F32 first cast d, m, dmin to that dtype and run the SAME ops in it (dequant.py:67,
 *                  75-76, ...), every op rounded once in that dtype.  "target" = pass out_dtype.
 *   out_dtype      the dtype of the dense result: the single `.to(dtype)` cast dequantize_tensor applies
 *                  to the block function's result (dequant.py:23), fused into the store -- same values
 *                  as dequantize(..., dtype=compute_dtype).to(out_dtype), one pass over memory. */
typedef enum ggq_dtype { GGQ_F16 = 0, GGQ_BF16 = 1, GGQ_F32 = 2 } ggq_dtype;

/* ---- queries ------------------------------------------------------------------------------- */

/* 1 if `qtype` has a HIP unpacker.  Replaces: `qtype in dequantize_functions` (dequant.py:21,287-301). */
int ggq_supported(int qtype);

/* Elements / bytes per block, 0 for an unknown qtype.  Replaces: gguf.GGML_QUANT_SIZES[qtype] (dequant.py:34). */
int ggq_block_size(int qtype);
int ggq_type_size(int qtype);

/* Static message for a ggq_status. */
const char* ggq_strerror(int status);

/* hipError_t (as int) of the most recent failing HIP call on this thread, 0 if none. */
int ggq_last_hip_error(void);

/* Library ABI version (bumped on any signature change). */
int ggq_abi_version(void);

/* Identity of the dequant kernels this binary holds: the first 16 hex digits of the sha256 over the compiler flags and the
 * sources of the dequant device code and its launch geometry (stamped by the build; "unstamped" for a hand build).  bench.py compares it with the id
 * recorded next to the committed PMC traffic figures, so a roofline.traffic number can never outlive the kernels it was
 * measured on. */
const char* ggq_build_id(void);

/* Memory-system probes with no arithmetic: the MEASURED ceiling bench.py prints beside the spec peak (roofline.measured_fill_GBps / _copy_GBps /
 * _read_GBps, and the read/write-weighted blend of them for each kernel's own traffic mix).  One kernel per call over `bytes` bytes (a multiple of
 * 16), 16 B per lane, every wave instruction covering 1 KiB of contiguous memory like the dequant kernels' accesses: FILL writes dst; COPY reads src
 * and writes dst; READ reads src only (dst = at least 4 KiB of scratch that is practically never written); the _NT kinds use non-temporal accesses.
 * Replaces: nothing in the reference -- measurement only ("6.29 TB/s measured copy ceiling" is this figure, per box). */
typedef enum ggq_cal_kind { GGQ_CAL_FILL = 0, GGQ_CAL_FILL_NT = 1, GGQ_CAL_COPY = 2, GGQ_CAL_COPY_NT = 3, GGQ_CAL_READ = 4 } ggq_cal_kind;
int ggq_calibrate(int kind, const void* src, void* dst, uint64_t bytes, void* hip_stream);

/* ---- one tensor ---------------------------------------------------------------------------- */

/* Dequantize n_blocks consecutive blocks: packed[n_blocks * type_size] -> out[n_blocks * block_size].
 * Replaces: dequantize(data, qtype, oshape, dtype=None) (dequant.py:30-44) -- the framing
 * (n_blocks = numel // type_size, output element b*block_size+j = block b element j) is the
 * caller's reshape; and dequantize_functions[qtype](blocks, block_size, type_size) (dequant.py:43).
 * With compute_dtype / out_dtype it also covers dequantize_tensor(tensor, dtype, dequant_dtype)
 * (dequant.py:15-23): dequantize(..., dtype=compute_dtype).to(out_dtype).
 * n_blocks == 0 is a no-op that returns GGQ_OK. */
int ggq_dequant(int qtype, const void* packed, uint64_t n_blocks, void* out, int compute_dtype, int out_dtype, void* hip_stream);

/* The same kernels with NON-TEMPORAL stores.  ggq_dequant stores write-through (sc1): its caller is a layer whose very next kernel reads
 * the weight (ops.py:242-244) and then finds it in the Infinity Cache (the emulated FLUX.1-dev step is 3 ms faster for it).  For a
 * result nobody reads back soon -- a tensor dequantized at load time, a benchmark of the unpack alone -- non-temporal stores are a
 * few per cent faster.  Same arguments, same values. */
int ggq_dequant_stream(int qtype, const void* packed, uint64_t n_blocks, void* out, int compute_dtype, int out_dtype, void* hip_stream);

/* Same with compute_dtype = out_dtype = GGQ_F16 (the stock node: dequantize(data, qtype, oshape)). */
int ggq_dequant_f16(int qtype, const void* packed, uint64_t n_blocks, void* out_f16, void* hip_stream);

/* ---- rows of one table (the embedding lookup) ------------------------------------------------------------------- */

/* out[i, :] = dequantized row indices[i] of a packed (n_rows x cols) table, cols = row_blocks * block_size; out is
 * (n_indices x cols) of out_dtype.  Replaces: GGMLOps.Embedding.forward_ggml_cast_weights (ops.py:251-260), which
 * dequantizes the WHOLE table (dequantize_tensor, ops.py:177) and then gathers with F.embedding: the values are the same bit
 * for bit, but only the rows asked for are unpacked -- a 152 k x 3584 table costs 1.1 GB of transient dense weight the
 * reference's way.  indices: device, int64, n_indices of them; an index outside [0, n_rows) is clamped (F.embedding asserts).
 * packed and out 16-byte aligned; the rows themselves may start at any block boundary (row_blocks * type_size need not be a
 * multiple of 16). */
int ggq_dequant_rows(int qtype, const void* packed, uint64_t n_rows, uint32_t row_blocks, const int64_t* indices, uint64_t n_indices,
                     void* out, int compute_dtype, int out_dtype, void* hip_stream);

/* ---- many tensors, one call (the weight set of a model) ------------------------------------- */

/* One entry per tensor.  Replaces: one dequantize_tensor() call per layer (ops.py:177). */
typedef struct ggq_desc {
    int32_t qtype;
    int32_t out_dtype;      /* ggq_dtype of the dense result */
    const void* packed;     /* device, 16-B aligned */
    void* out;              /* device, 16-B aligned */
    uint64_t n_blocks;
    int32_t compute_dtype;  /* ggq_dtype the arithmetic runs in (GGQ_F16 = stock path) */
    int32_t reserved;       /* must be 0 */
} ggq_desc;

typedef struct ggq_plan ggq_plan;

/* Build a launch plan for `n` tensors on the current device: descriptors are grouped by
 * (qtype, compute_dtype, out_dtype), their work is prefix-summed and the tables are copied to device memory
 * once (synchronous, load-time).  Pointers are captured, not the bytes: the plan stays valid
 * while the tensors keep their addresses (packed weights resident in HBM). */
int ggq_plan_create(const ggq_desc* descs, uint32_t n, ggq_plan** plan_out);

/* Enqueue the whole plan on `hip_stream`: one kernel per (qtype, compute_dtype, out_dtype) present.  The calling
 * thread's current device must be the one the plan was created on (its tables live there): GGQ_ERR_ARG otherwise. */
int ggq_plan_launch(const ggq_plan* plan, void* hip_stream);

/* Algorithmic bytes one launch moves (packed read + dense write), and its kernel count. */
uint64_t ggq_plan_bytes(const ggq_plan* plan);
uint32_t ggq_plan_kernels(const ggq_plan* plan);

void ggq_plan_destroy(ggq_plan* plan);

/* ---- layer i+1 while layer i computes (side-stream prefetch) --------------------------------------------------------------- */

/* The reference dequantizes a layer's weight and then runs its GEMM, one after the other on one stream (ops.py:242-244), and in
 * low-VRAM mode first copies the packed bytes host->device on that same stream (ops.py:209).  The unpack is HBM-bound, the copy
 * PCIe-bound, the GEMM MFMA-bound: a ggq_overlap owns a COPY stream and an UNPACK stream (and per-slot events) on which the next
 * layers' copies and the NEXT layer's unpack run while the current layer's GEMM occupies the matrix cores.  Values are the same
 * kernels' output: bit-identical.  Slots are the caller's buffers; the object only orders the work on them:
 *   ggq_overlap_create     two streams + 4 events per slot on the current device; n_slots in [1, 16] (dense slots and staging slots
 *                          are numbered independently, both < n_slots).
 *   ggq_overlap_copy       enqueue on the copy streams (two: weights of 4 MB and more go as two halves, which keeps the link fuller -- 45 -> 51
 *                          GB/s measured): packed_bytes host_packed -> dev_packed (staging slot `staging_slot`), after
 *                          the unpack that last read that staging slot.  Not ordered against the caller's stream at all: copies run
 *                          as far ahead as there are staging slots.  host_packed should be pinned memory.
 *   ggq_overlap_prefetch   enqueue on the unpack stream, ordered after everything `main_stream` holds at this moment (so the dense
 *                          slot's previous consumer has finished) and, if staging_slot >= 0, after that slot's copy:
 *                          ggq_dequant(qtype, dev_packed, n_blocks, out, compute_dtype, out_dtype).
 *   ggq_overlap_wait       make `main_stream` wait for the dense slot's last prefetch (no host sync).
 * The caller owns every buffer and keeps it alive until the consumer has run; one thread drives a ggq_overlap at a time. */
typedef struct ggq_overlap ggq_overlap;
int ggq_overlap_create(int n_slots, ggq_overlap** out);
int ggq_overlap_copy(ggq_overlap* ov, int staging_slot, const void* host_packed, void* dev_packed, uint64_t packed_bytes);
int ggq_overlap_prefetch(ggq_overlap* ov, int slot, int staging_slot, int qtype, const void* dev_packed, uint64_t n_blocks, void* out,
                         int compute_dtype, int out_dtype, void* main_stream);
int ggq_overlap_wait(ggq_overlap* ov, int slot, void* main_stream);
void ggq_overlap_destroy(ggq_overlap* ov);

/* ---- fused dequantize + linear for a few rows of x -------------------------------------- */

/* y[m, rows] = x[m, cols] @ W^T (+ bias[rows]), W = dequantize_tensor(packed, dtype) of logical shape (rows, cols), 1 <= m <= 4,
 * computed straight from the packed blocks: the dense weight is never written.  x, bias, y are of `dtype` (ggq_dtype) and
 * contiguous; x must be 16-byte aligned.  The weights are the reference's values bit for bit; the contraction accumulates in
 * fp32, so y equals F.linear(x, dequantize_tensor(...), bias) up to the order of fp32 additions (as any two GEMV kernels differ).
 * Replaces, for m <= 4: GGMLOps.Linear.forward_ggml_cast_weights (ops.py:242-244) = get_weight + F.linear.
 * GGQ_ERR_ARG if the shape is outside what the kernel stages in LDS (rows of more than 6128 packed bytes, m*cols too large):
 * the caller keeps dequantize + F.linear. */
int ggq_linear_small(int qtype, const void* packed, uint32_t rows, uint32_t cols, const void* x, uint32_t m, const void* bias,
                     void* y, int dtype, void* hip_stream);

/* ---- fused dequantize + GEMM on the matrix cores for many rows of x (large-m end) ---------------------- */

/* y[m, rows] = x[m, cols] @ W^T (+ bias[rows]), W = dequantize_tensor(packed, dtype) of logical shape (rows, cols), any m >= 1, computed
 * from the packed blocks on v_mfma_f32_32x32x16_{f16,bf16}: each lane decodes the 8 consecutive weights that ARE its MFMA operand, the
 * dense weight never exists in memory.  x, bias, y of `dtype` in {GGQ_F16, GGQ_BF16}, contiguous, x 16-byte aligned; cols % 256 == 0 -- for the 32-element
 * block formats cols % 64 == 0 suffices (a shorter last span; not with tile_rows 256) -- GGQ_ERR_ARG otherwise: the caller keeps dequantize + GEMM.
 * tile_rows = rows of x per workgroup tile: 16 (the 16-row kernel on v_mfma_f32_16x16x32, csrc/ggq_mfma16.hpp: up to 32 rows of x per tile) / 32 / 64 / 128 / 256, 0 = auto
 * (16-row kernel up to 8 rows of x -- 16 on weights of at most 4096 rows --, 32-row tiles to 32 rows, 64-row tiles above).
 * Weights bit-identical to the reference's, fp32 accumulation in the MFMA's order, K split over 4 waves and summed in a fixed order
 * (deterministic).  Replaces (install()'s default for up to 256 rows of x; `exact` turns it off): GGMLOps.Linear.forward_ggml_cast_weights
 * (ops.py:242-244) = get_weight + F.linear. */
int ggq_linear_mfma(int qtype, const void* packed, uint32_t rows, uint32_t cols, const void* x, uint32_t m, const void* bias,
                    void* y, int dtype, int tile_rows, void* hip_stream);

/* The same with a caller-provided scratch buffer (device memory, 16-byte aligned, `workspace_bytes` long; torch's allocator owns it, the call only
 * uses it until the kernels it enqueued have run): a weight with few rows and long rows (FLUX `mlp.2` 3072 x 12288: 96 workgroups of the 32-row kernel on 256 CUs)
 * then has its K ALSO split across workgroups -- every workgroup stores fp32 partial sums of its slice of whole spans, a second small kernel adds the slices in
 * order (deterministic), the bias, and casts.  ggq_linear_mfma_workspace() = the bytes the library would use for that request (0: it would not split; a smaller or
 * NULL workspace is never an error, the launch just keeps K inside the workgroups).  Same numerics statement as ggq_linear_mfma; same reference interface
 * (ops.py:242-244). */
int ggq_linear_mfma_ws(int qtype, const void* packed, uint32_t rows, uint32_t cols, const void* x, uint32_t m, const void* bias,
                       void* y, int dtype, int tile_rows, void* workspace, uint64_t workspace_bytes, void* hip_stream);
uint64_t ggq_linear_mfma_workspace(int qtype, uint32_t rows, uint32_t cols, uint32_t m, int tile_rows);

#ifdef __cplusplus
}
#endif
#endif /* GGQ_H */
