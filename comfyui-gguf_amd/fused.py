"""``F.linear`` on a quantized weight straight from the packed blocks -- the dense weight is never written.  (Part of install()'s default since round 5.)
Two kernels: ``linear_small`` for ONE TO FOUR rows of input (include/ggq.h ``ggq_linear_small``, csrc/ggq_linear.hpp: HBM-bound,
no matrix cores) and ``linear_mfma`` for many rows (``ggq_linear_mfma``, csrc/ggq_mfma.hpp: each lane decodes the eight
consecutive weights that are its MFMA operand).

The reference's ``GGMLOps.Linear.forward_ggml_cast_weights`` (ops.py:242-244) dequantizes the whole weight and calls
``F.linear`` whatever the input looks like.  FLUX's modulation linears see the conditioning vector (batch rows): 76 layers,
27 % of the model's weights, one row each.  For them the fused kernel moves 0.56 B per weight instead of 0.56 + 2 + 2.

The weights are the reference's values bit for bit (same decode, same fp16 op sequence, same ``.to(dtype)``); the dot products
accumulate in fp32 in this kernel's own order, so the result equals ``F.linear(x, dequantize_tensor(w, x.dtype), bias)`` up to
fp32 summation order -- parity is a tolerance against an fp32 reference (tests/test_gpu_linear.py), not bit-exact; measured against an fp64 product the fused
results are no further from exact than F.linear's (tools/fused_error.py, profiles/r05_fused_error.json), which is why install() turns them on by default (``exact=True`` off):
``GGMLLinear.fuse_small_m = True`` for the stand-in, ``install(..., fused_small_m=True)`` for a ComfyUI-GGUF checkout.
"""
import torch

from . import _native
from .dequant import (GGQUnsupported, _DEVICE_OK, _HIP_TABLE, _OUT_CODE, _as_bytes, _cur_device, _dequantize_op, _device_served, _is_compiling, _qtype_key, _raw_stream,
                      dequantize_tensor, is_quantized)

MAX_ROWS = 4
_F16, _BF16 = torch.float16, torch.bfloat16
_small_call = None        # ggq_linear_small / ggq_linear_mfma(_ws): the CPython binding (csrc/ggq_pyfast.c) when built, else ctypes -- same C entry points
_mfma_call = None
_mfma_ws_call = None
_ws_bytes = {}            # (qtype id, rows, cols, m, tile_rows) -> bytes of scratch ggq_linear_mfma_workspace() asks for (0 for most shapes): asked once per shape


def _bind():
    global _small_call, _mfma_call, _mfma_ws_call
    fast = _native.fast()
    L = _native.lib()
    _small_call = fast.linear_small if fast is not None else L.ggq_linear_small
    _mfma_call = fast.linear_mfma if fast is not None else L.ggq_linear_mfma
    _mfma_ws_call = fast.linear_mfma_ws if fast is not None else L.ggq_linear_mfma_ws


def _prepare(x, weight, bias, dequant_dtype, what, dtypes, need_cols_256):
    """The checks both fused kernels share; returns (qid, rows, cols, m, x as (m, cols), bias tensor or None).  Raises GGQUnsupported for
    anything the kernels do not take.  Kept flat: these calls sit on the per-layer hot path of small-batch / text-encoder runs, where the
    host's issue rate is the limit."""
    if dequant_dtype is not None and dequant_dtype is not _F16:
        raise GGQUnsupported(f"{what} computes the stock fp16 weight values only")
    if torch.is_grad_enabled() and (x.requires_grad or (bias is not None and bias.requires_grad)):
        # the result of a raw kernel launch has no grad_fn; the reference's F.linear carries the gradient back to the input and the bias
        # (a LoRA-training node, gradient-based guidance): those calls keep dequantize + F.linear
        raise GGQUnsupported(f"{what}: autograd is recording -- dequantize + F.linear builds the graph")
    if getattr(weight, "patches", None) or (bias is not None and getattr(bias, "patches", None)):
        # get_weight applies LoRA patches to the weight AND to the bias (ops.py:183-190 via ops.py:205-206)
        raise GGQUnsupported("LoRA-patched weight or bias: needs the reference's get_weight")
    qtype = getattr(weight, "tensor_type", None)
    ent = _HIP_TABLE.get(qtype)
    if ent is None:
        ent = _HIP_TABLE.get(_qtype_key(qtype))
        if ent is None:
            raise GGQUnsupported(f"no {what} for qtype {getattr(qtype, 'name', qtype)!r}")
    shape = getattr(weight, "tensor_shape", ())
    if len(shape) != 2:
        raise GGQUnsupported(f"{what}: 2-D weight")
    rows, cols = int(shape[0]), int(shape[1])
    # the MFMA kernels contract whole 256-element spans; for the 32-element legacy blocks the LAST span may be a shorter multiple of 64 (SD3.5: 2432 columns)
    k_ok = not need_cols_256 or cols % 256 == 0 or (ent[1] == 32 and cols % 64 == 0)
    if (not x.is_cuda or x.dtype not in dtypes or x.shape[-1] != cols or cols == 0 or not k_ok or x.numel() == 0):
        raise GGQUnsupported(f"{what}: GPU tensors, {'fp16 / bf16' if need_cols_256 else 'fp16 / bf16 / fp32'} input of matching width"
                             + (", cols % 256 == 0 (32-element blocks: cols % 64 == 0)" if need_cols_256 else ""))
    m = x.numel() // cols
    xf = x if x.dim() == 2 else x.reshape(m, cols)
    if not xf.is_contiguous() or xf.data_ptr() & 15:
        xf = xf.contiguous().clone()
    if bias is not None:
        if is_quantized(bias):
            bias = dequantize_tensor(bias, x.dtype)
        if type(bias) is not torch.Tensor:
            bias = bias.as_subclass(torch.Tensor)
        if bias.device != x.device or bias.dtype is not x.dtype or not bias.is_contiguous():
            bias = bias.to(device=x.device, dtype=x.dtype).contiguous()
        if bias.numel() != rows:
            raise GGQUnsupported("bias does not match the weight's rows")
    return ent[0], rows, cols, m, xf, bias


def _run(call, name, qid, weight, rows, cols, xf, m, bias, x, extra, weight_to=None):
    if weight_to is not None and weight.device != x.device:
        # low-VRAM mode (CPU-resident packed weight, reference ops.py:209): the host -> device copy is made HERE, after every
        # eligibility check has passed -- a request the kernel declines never pays for a copy the reference's method then repeats
        weight = weight.to(weight_to)
    with torch._C.DisableTorchFunctionSubclass():
        if not weight.is_cuda or weight.dtype is not torch.uint8 or not weight.is_contiguous() or weight.data_ptr() & 15:
            raise GGQUnsupported("packed weight must be a contiguous, 16-byte aligned byte tensor on the GPU")
        index = x.device.index
        if not (_DEVICE_OK.get(index) or _device_served(index)):
            raise GGQUnsupported(f"cuda:{index} is not a gfx950 device")
        y = torch.empty((m, rows), dtype=x.dtype, device=x.device)
        args = (qid, weight.data_ptr(), rows, cols, xf.data_ptr(), m, None if bias is None else bias.data_ptr(), y.data_ptr(), _OUT_CODE[x.dtype]) + extra
        if _cur_device() != index:
            with torch.cuda.device(index):
                rc = call(*args, _raw_stream(index))
        else:
            rc = call(*args, _raw_stream(index))
    if rc == _native.GGQ_ERR_ARG:
        raise GGQUnsupported(f"shape outside what {name} takes")
    if rc:
        _native.check(rc, name)
    return y if x.dim() == 2 else y.reshape(*x.shape[:-1], rows)


_LIN_SLICE, _LIN_WAVES = 6 * 64 * 16, 4        # csrc/ggq_linear.hpp LIN_SLICE / LIN_WAVES: one row's packed bytes must fit a wave's LDS slice


def linear_small(x, weight, bias=None, dequant_dtype=None, weight_to=None):
    """x: (..., cols) on the GPU with at most MAX_ROWS rows in total; weight: GGMLTensor of logical shape (rows, cols).
    ``weight_to``: device to move a CPU-resident packed weight to once the request is known to be served (install.py).
    Raises GGQUnsupported for anything the kernel does not take (the caller keeps dequantize + F.linear)."""
    qid, rows, cols, m, xf, bias = _prepare(x, weight, bias, dequant_dtype, "fused linear", _OUT_CODE, False)
    if not 1 <= m <= MAX_ROWS:
        raise GGQUnsupported(f"fused linear takes 1..{MAX_ROWS} input rows, got {m}")
    _, block_size, type_size = _HIP_TABLE[_qtype_key(qid)]
    row_bytes = cols // block_size * type_size
    slice_bytes = (row_bytes + 15 + 1023) & ~1023          # csrc/ggq_linear.hpp lin_slice_bytes(): whole 64-lane x 16-byte load units per wave
    if cols % block_size or row_bytes + 15 > _LIN_SLICE or m * cols * x.element_size() + _LIN_WAVES * slice_bytes > 150 * 1024:
        raise GGQUnsupported("fused linear: a row's packed bytes (or x) exceed the kernel's LDS staging")     # what ggq_linear_small answers GGQ_ERR_ARG to
    if _small_call is None:
        _bind()
    return _run(_small_call, "ggq_linear_small", qid, weight, rows, cols, xf, m, bias, x, (), weight_to)


AUTO_MAX_ROWS = 256      # rows of x up to which a fused MFMA shape beats dequantize + F.linear on FLUX / SD3.5 / T5 layer shapes (profiles/r03_gemm_tile_bench.json;
                         # above it the shared-tile kernel runs at 0.9-1.0 PFLOP/s against hipBLASLt's 1.2-1.3 on the freshly written, cache-hot dense weight,
                         # and profiles/r04_gemm_skeleton_sweep.json shows why that gap does not close: DESIGN.md section 4c)


# ... and, above 128 rows of x, the multiply-accumulates (rows of x) x (rows of W) x (columns) beyond which the K-split kernel's repeated decode (once per 64 rows of x) costs more
# than one unpack + hipBLASLt.  Measured per format on FLUX / SD3.5 / T5 shapes at 64 ... 256 rows, two alternations (profiles/r06_fused_vs_unpack_row_threshold.json, us fused vs
# unpack + F.linear): Q4_K 12288 x 3072 at 192 rows 46-48 vs 51-52 (7.2 G: stays), at 256 rows 60-61 vs 54-55 (9.7 G: declines), 4096 x 4096 at 256 rows 27 vs 32; the formats with
# the dearer decode or the wider blocks cross earlier -- 12288 x 3072 at 192 rows: Q8_0 58 vs 50, Q6_K 61 vs 49, Q3_K 52 vs 49; Q5_0 9728 x 2432 at 256 rows (6.1 G) 49-51 vs 43,
# at 192 rows (4.5 G) level; 4096 x 4096 at 256 rows (4.3 G) still ahead or level for all four.  Up to 128 rows every format is ahead or level on every shape measured.
# (Rounds 5-6 used rows of x x rows of W > 3.6 M plus flat row limits of 128 / 64 for the 32-element-block formats: those limits had been measured on a 32-row kernel whose LDS
# row pitch put every fourth row of a Q5_0 tile on the same banks -- EXPERIMENTS.md R6-10.)
AUTO_MAX_MACS = 8.0e9
AUTO_MAX_MACS_DEAR_DECODE = 5.0e9
_DEAR_DECODE = (6, 8, 11, 14)          # ggml type ids of Q5_0, Q8_0, Q3_K, Q6_K


def _auto_declines(qid, rows, cols, m, max_rows=AUTO_MAX_ROWS):
    """True where `tile_rows=0` (auto) hands the call back to dequantize + F.linear."""
    return m > max_rows or (m > 128 and m * rows * cols > (AUTO_MAX_MACS_DEAR_DECODE if qid in _DEAR_DECODE else AUTO_MAX_MACS))


def linear_mfma(x, weight, bias=None, dequant_dtype=None, tile_rows=0, weight_to=None, auto_max_rows=AUTO_MAX_ROWS):
    """``F.linear(x, dequantize_tensor(weight, x.dtype), bias)`` on the matrix cores, from the packed blocks (include/ggq.h
    ``ggq_linear_mfma``).  x: (..., cols) fp16 / bf16 on the GPU; weight: GGMLTensor of logical shape (rows, cols) with
    cols % 256 == 0.  Weights bit-identical to the reference's; fp32 accumulation in the kernel's own order (tolerance parity,
    tests/test_gpu_mfma.py).  Raises GGQUnsupported for anything the kernel does not take.

    ``tile_rows=0`` (auto) never picks something slower than the default path: the library chooses the fastest fused shape for
    (rows of x, rows of the weight), and inputs of more than ``auto_max_rows`` rows -- or of more than 128 rows on a weight so large that
    rows of x times its elements exceeds ``AUTO_MAX_MACS`` (``AUTO_MAX_MACS_DEAR_DECODE`` for Q5_0 / Q8_0 / Q3_K / Q6_K) -- are DECLINED
    (GGQUnsupported: the caller keeps dequantize + F.linear, which is faster there).  An explicit ``tile_rows`` (32 / 64 / 128 = K-split kernel, 256 = shared-tile
    kernel) or ``auto_max_rows=None`` forces the fused kernel at any size."""
    qid, rows, cols, m, xf, bias = _prepare(x, weight, bias, dequant_dtype, "fused GEMM", (_F16, _BF16), True)
    if tile_rows not in (0, 16, 32, 64, 128, 256) or (tile_rows == 256 and (rows % 8 or cols % 256)):
        raise GGQUnsupported("fused GEMM: tile_rows is 0 (auto), 16, 32, 64, 128 or 256 (the shared-tile kernel: rows % 8 == 0, cols % 256 == 0)")
    if tile_rows == 0 and auto_max_rows is not None and _auto_declines(qid, rows, cols, m, auto_max_rows):
        raise GGQUnsupported(f"fused GEMM (auto): {m} rows of x on a {rows} x {cols} weight -- dequantize + F.linear is the faster path there (above {auto_max_rows} rows; above 128 "
                             f"rows when rows of x * rows * columns exceeds {AUTO_MAX_MACS:.1e}, {AUTO_MAX_MACS_DEAR_DECODE:.1e} for Q5_0 / Q8_0 / Q3_K / Q6_K); pass tile_rows= to "
                             f"force a fused shape")
    if _mfma_call is None:
        _bind()
    # K split across workgroups for weights with few, long rows (include/ggq.h ggq_linear_mfma_ws): the library says how much scratch it would use for this shape
    # (0 for most: asked once per shape), torch's caching allocator provides it, stream-ordered like y
    key = (qid, rows, cols, m, tile_rows)
    ws = _ws_bytes.get(key)
    if ws is None:
        ws = _ws_bytes[key] = int(_native.lib().ggq_linear_mfma_workspace(qid, rows, cols, m, int(tile_rows)))
    if ws:
        scratch = torch.empty(ws, dtype=torch.uint8, device=x.device)
        return _run(_mfma_ws_call, "ggq_linear_mfma_ws", qid, weight, rows, cols, xf, m, bias, x, (int(tile_rows), scratch.data_ptr(), ws), weight_to)
    return _run(_mfma_call, "ggq_linear_mfma", qid, weight, rows, cols, xf, m, bias, x, (int(tile_rows),), weight_to)


SMALL_M_TALL_ROWS = 16384     # one row of x: from this many output columns on the 16-row MFMA kernel is ahead of the GEMV


def _tall(weight):
    shape = getattr(weight, "tensor_shape", ())
    return len(shape) == 2 and shape[0] >= SMALL_M_TALL_ROWS


def linear_auto(x, weight, bias=None, dequant_dtype=None, weight_to=None, small_m=True, mfma_max_m=AUTO_MAX_ROWS):
    """The policy of install()'s default in one place (install._fuse_linear, bench.py's fused workloads, the forward emulations): which fused kernel
    takes ``F.linear(x, dequantize_tensor(weight, x.dtype), bias)`` for this many rows of x, or GGQUnsupported when none does (the caller keeps
    dequantize + F.linear).  One row -> ``linear_small`` when ``small_m`` (shapes it declines, and weights of 16384+ rows, go to the MFMA kernels); 2..``mfma_max_m``
    rows -> ``linear_mfma`` with the library's choice of kernel and tile (16-row kernel up to 8 rows of x, 32-row kernel above; ggq_linear.hip)."""
    cols = x.shape[-1]
    m = x.numel() // cols if cols else 0
    # one row: the GEMV (level with the 16-row MFMA kernel up to ~12 k output columns, ahead on the shorter weights: 9.2 vs 10.2 us at 9216 x 3072; also the only
    # one that takes fp32 activations); two to four rows, and one row on the tallest weights: the MFMA kernel, whose time does not grow with m
    # (12288 x 3072: 11.0 us at 4 rows against the GEMV's 17.1; 18432 x 3072 at one row 14.5 against 15.6 -- profiles/r06_mfma16_variants_kmap_and_blocks_per_wave.json)
    if small_m and m <= MAX_ROWS and not (mfma_max_m and x.dtype in (_F16, _BF16) and (m > 1 or _tall(weight))):
        try:
            return linear_small(x, weight, bias, dequant_dtype, weight_to=weight_to)
        except GGQUnsupported:
            if not mfma_max_m:
                raise
    if m <= mfma_max_m:
        return linear_mfma(x, weight, bias, dequant_dtype, weight_to=weight_to, auto_max_rows=mfma_max_m)
    raise GGQUnsupported(f"fused linears: {m} rows of x is above the {mfma_max_m} the default fuses")


# ---- torch.compile: the fused launches as opaque custom ops ------------------------------------------------------------------------------
# The reference lets torch >= 2.8 compile straight through GGMLOps.Linear.forward_ggml_cast_weights (ops.py:11-42, 242-244).  A ctypes call is nothing
# Dynamo can put in a graph, so while it traces, install()'s wrapper goes through `linear_traced` below: the same eligibility rules as the eager
# wrappers, stated as plain conditions on what is static under tracing (dtypes, shapes, the GGMLTensor's attributes) -- no exceptions, no pointer
# values -- and then ONE custom op per layer (`ggq::linear_small` / `ggq::linear_mfma`, torch.library, with a fake implementation for shape
# propagation).  Eager calls keep the direct binding (no dispatcher on the per-layer hot loop).  So a compiled model runs the SAME kernels as the
# eager default install -- bit for bit, the kernels are deterministic -- instead of silently falling back to unpack + F.linear (Missing #1).
def _op_body(call_name, x, packed, bias, qtype, rows, cols, extra):
    """What both ops do at run time (real tensors): make x / packed / bias launchable, launch, and -- a graph cannot decline at run time -- compute the
    layer as unpack + F.linear right here if the library answers "not this shape" after all."""
    m = x.numel() // cols
    xf = x.reshape(m, cols)
    if not xf.is_contiguous() or xf.data_ptr() & 15:
        xf = xf.contiguous().clone()
    if packed.dtype is not torch.uint8 or not packed.is_contiguous() or packed.data_ptr() & 15:
        packed = _as_bytes(packed)
    if bias is not None and (bias.dtype is not x.dtype or not bias.is_contiguous()):
        bias = bias.to(x.dtype).contiguous()
    index = x.device.index
    if not (_DEVICE_OK.get(index) or _device_served(index)):
        raise _native.GGQNativeError(f"cuda:{index} is not a gfx950 device: the compiled graph holds a gfx950 kernel")
    if _small_call is None:
        _bind()
    y = torch.empty((m, rows), dtype=x.dtype, device=x.device)
    args = (qtype, packed.data_ptr(), rows, cols, xf.data_ptr(), m, None if bias is None else bias.data_ptr(), y.data_ptr(), _OUT_CODE[x.dtype]) + extra
    call = _small_call if call_name == "ggq_linear_small" else _mfma_call
    if call_name == "ggq_linear_mfma":
        key = (qtype, rows, cols, m, extra[0])
        ws = _ws_bytes.get(key)
        if ws is None:
            ws = _ws_bytes[key] = int(_native.lib().ggq_linear_mfma_workspace(qtype, rows, cols, m, extra[0]))
        if ws:
            scratch = torch.empty(ws, dtype=torch.uint8, device=x.device)
            call, args = _mfma_ws_call, args + (scratch.data_ptr(), ws)
    with torch.cuda.device(index):
        rc = call(*args, _raw_stream(index))
    if rc == _native.GGQ_ERR_ARG:
        w = _dequantize_op(packed, qtype, _native.F16, _OUT_CODE[x.dtype]).reshape(rows, cols)
        return torch.nn.functional.linear(x, w, bias)
    if rc:
        _native.check(rc, call_name)
    return y.reshape(*x.shape[:-1], rows)


def _linear_small_impl(x: torch.Tensor, packed: torch.Tensor, bias: "torch.Tensor | None", qtype: int, rows: int, cols: int) -> torch.Tensor:
    return _op_body("ggq_linear_small", x, packed, bias, qtype, rows, cols, ())


def _linear_mfma_impl(x: torch.Tensor, packed: torch.Tensor, bias: "torch.Tensor | None", qtype: int, rows: int, cols: int, tile_rows: int) -> torch.Tensor:
    return _op_body("ggq_linear_mfma", x, packed, bias, qtype, rows, cols, (tile_rows,))


def _linear_fake(x, packed, bias, qtype, rows, cols, tile_rows=0):
    return x.new_empty(tuple(x.shape[:-1]) + (rows,))


try:
    from typing import Optional
    _linear_small_impl.__annotations__["bias"] = Optional[torch.Tensor]
    _linear_mfma_impl.__annotations__["bias"] = Optional[torch.Tensor]
    _linear_small_op = torch.library.custom_op("ggq::linear_small", _linear_small_impl, mutates_args=(), device_types="cuda")
    _linear_small_op.register_fake(_linear_fake)
    _linear_mfma_op = torch.library.custom_op("ggq::linear_mfma", _linear_mfma_impl, mutates_args=(), device_types="cuda")
    _linear_mfma_op.register_fake(_linear_fake)
except (AttributeError, RuntimeError):       # torch without torch.library.custom_op: compiled graphs keep the reference's method
    _linear_small_op = _linear_mfma_op = None


_TRACE_ANY_DEVICE = False      # tests only: lets the CPU suite trace the default install over stub CPU kernels of the custom ops (graph-break count without a GPU)


def linear_traced(layer, x, small_m, mfma_max_m):
    """install()'s Linear wrapper while torch.compile traces it: the fused custom op for this layer and input, or None when the eager wrapper
    would have handed the call to the reference's method (the caller then traces that).  Mirrors `_prepare` + `linear_small` / `linear_mfma(auto)`;
    every condition here is on trace-time constants."""
    weight, bias = layer.weight, layer.bias
    if _linear_small_op is None or weight is None or not (x.is_cuda or _TRACE_ANY_DEVICE):
        return None
    dd = layer.dequant_dtype
    if dd is not None and dd is not _F16:
        return None
    if torch.is_grad_enabled() and (x.requires_grad or (bias is not None and bias.requires_grad)):
        return None
    if getattr(weight, "patches", None) or (bias is not None and getattr(bias, "patches", None)):
        return None
    ent = _HIP_TABLE.get(_qtype_key(getattr(weight, "tensor_type", None)))
    shape = getattr(weight, "tensor_shape", ())
    if ent is None or len(shape) != 2:
        return None
    qid, block_size, type_size = ent
    rows, cols = int(shape[0]), int(shape[1])
    if cols == 0 or x.shape[-1] != cols or x.numel() == 0 or cols % block_size:
        return None
    m = x.numel() // cols
    if bias is not None and bias.numel() != rows:
        return None
    small = small_m and m <= MAX_ROWS and x.dtype in _OUT_CODE and not (mfma_max_m and x.dtype in (_F16, _BF16) and (m > 1 or rows >= SMALL_M_TALL_ROWS))
    if small:
        row_bytes = cols // block_size * type_size
        slice_bytes = (row_bytes + 15 + 1023) & ~1023
        small = row_bytes + 15 <= _LIN_SLICE and m * cols * x.element_size() + _LIN_WAVES * slice_bytes <= 150 * 1024
    mfma = False
    if not small:
        k_ok = cols % 256 == 0 or (block_size == 32 and cols % 64 == 0)
        mfma = x.dtype in (_F16, _BF16) and k_ok and m <= mfma_max_m and not _auto_declines(qid, rows, cols, m)
        if not mfma:
            return None
    if weight.device != x.device:
        weight = weight.to(x.device)                                     # low-VRAM mode: the copy the reference's method makes (ops.py:209)
    if bias is not None and is_quantized(bias):
        bias = dequantize_tensor(bias, x.dtype)                          # (traced: the ggq::dequantize custom op)
    with torch._C.DisableTorchFunctionSubclass():                        # the GGMLTensor operands must not claim the result (and Dynamo cannot trace as_subclass)
        packed = _as_bytes(weight, align=False)
        if bias is not None:
            bias = bias.to(device=x.device, dtype=x.dtype)
        if small:
            return _linear_small_op(x, packed, bias, qid, rows, cols)
        return _linear_mfma_op(x, packed, bias, qid, rows, cols, 0)
