"""Host side of the MI355X dequant path: same names, arguments and results as the reference's
dequant.py (city96/ComfyUI-GGUF), backed by the hand-written HIP kernels behind include/ggq.h.

Mirrored surface (reference file:line):
    TORCH_COMPATIBLE_QTYPES / is_torch_compatible / is_quantized      dequant.py:7-13
    dequantize_tensor(tensor, dtype=None, dequant_dtype=None)          dequant.py:15-28
    dequantize(data, qtype, oshape, dtype=None)                        dequant.py:30-44
    dequantize_functions  {qtype: fn(blocks, block_size, type_size, dtype=None)}   dequant.py:287-301

What runs where
    * packed bytes on an AMD GPU: the HIP kernels -- bit-identical to the reference's eager op
      sequence in all three arithmetic modes the nodes can select (dequant_dtype None / float16,
      bfloat16, float32, or "target"; nodes.py:186).  The final ``.to(dtype)`` of
      dequantize_tensor (dequant.py:23) is fused into the store for dtype in {float16, bfloat16,
      float32}; any other dtype gets the same single torch cast.
    * BF16 "blocks" (dequant.py:61-62) are a pure bit reinterpretation -> one torch op on device.
    * anything else -- CPU tensors, other dequant_dtype values, unknown qtypes --
      is NOT served here: :class:`GGQUnsupported` is raised.  There is deliberately no CPU or torch
      re-implementation in this package; ``install()`` (install.py) wires this module in front of
      the reference's own functions, which keep handling those cases.
"""
import torch

from . import _native
from .qtypes import GGMLQuantizationType, GGML_QUANT_SIZES, HIP_QTYPES

Q = GGMLQuantizationType

TORCH_COMPATIBLE_QTYPES = (None, Q.F32, Q.F16)

_OUT_CODE = {torch.float16: _native.F16, torch.bfloat16: _native.BF16, torch.float32: _native.F32}
# the reference's arithmetic dtype (`dtype` of the block functions = dequant_dtype) -> ggq_dtype
_COMPUTE_CODE = {None: _native.F16, torch.float16: _native.F16, torch.bfloat16: _native.BF16, torch.float32: _native.F32}
_COMPUTE_TORCH = {None: torch.float16, torch.float16: torch.float16, torch.bfloat16: torch.bfloat16, torch.float32: torch.float32}


class GGQUnsupported(NotImplementedError):
    """The request is outside what the HIP path serves (see module docstring)."""


def is_torch_compatible(tensor):
    return tensor is None or getattr(tensor, "tensor_type", None) in TORCH_COMPATIBLE_QTYPES


def is_quantized(tensor):
    return not is_torch_compatible(tensor)


def _qtype_key(qtype):
    try:
        return Q(int(qtype))
    except (ValueError, TypeError):
        return None


def hip_supported(qtype):
    return _qtype_key(qtype) in HIP_QTYPES


def _is_compiling():
    c = getattr(torch, "compiler", None)
    return bool(c and hasattr(c, "is_compiling") and c.is_compiling())


def _as_bytes(data, align=True):
    """dequant.py:37-39: rows = data.reshape((-1, data.shape[-1])).view(torch.uint8), flattened."""
    if type(data) is not torch.Tensor and not _is_compiling():
        # strip GGMLTensor: plain byte buffer from here on.  (Not while torch.compile traces: Dynamo cannot trace `as_subclass(torch.Tensor)` -- a graph break per
        # layer in rounds 2-5 -- and does not need it: the subclass rides through reshape / view into the custom op, whose result the caller strips.)
        data = data.as_subclass(torch.Tensor)
    if data.dtype != torch.uint8:
        data = data.reshape((-1, data.shape[-1])).view(torch.uint8)
    data = data.reshape(-1)
    if not data.is_contiguous():
        data = data.contiguous()
    if align and data.data_ptr() % 16:
        data = data.clone()                            # fresh allocations are >= 256-B aligned
    return data


# The per-layer hot loop calls this once per quantized layer per forward (ops.py:177) and the kernel
# itself takes a few microseconds, so the host side is kept lean: raw stream / device queries, one
# ctypes call, no torch-function dispatch on the GGMLTensor subclass (each costs microseconds).
try:
    _raw_stream = torch._C._cuda_getCurrentRawStream       # (device_index) -> hipStream_t as int
    _cur_device = torch._C._cuda_getDevice
except AttributeError:                                     # torch built without the GPU runtime bindings
    def _raw_stream(index):
        return torch.cuda.current_stream(index).cuda_stream

    def _cur_device():
        return torch.cuda.current_device()

_NoTorchFunction = torch._C.DisableTorchFunctionSubclass
_ggq_dequant = None     # ggq_dequant(qtype, packed, n_blocks, out, compute, out_dtype, stream) -> status: _ggq_fast.dequant (CPython
                        # binding, csrc/ggq_pyfast.c) when it has been built, else the ctypes function -- the same C entry point
_fast = None


def _bind():
    global _ggq_dequant, _fast
    _fast = _native.fast()
    _ggq_dequant = _fast.dequant if _fast is not None else _native.lib().ggq_dequant
    return _ggq_dequant

# libggq_hip.so holds gfx950 code objects only.  A GPU of any other architecture (another AMD part, or an NVIDIA device in a
# CUDA build of torch) is "not served here", exactly like a CPU tensor: GGQUnsupported, so that under install() the reference's
# own torch path keeps working on it instead of a failing kernel launch.  Checked once per device index.
_DEVICE_OK = {}


def _device_served(index):
    ok = _DEVICE_OK.get(index)
    if ok is None:
        arch = getattr(torch.cuda.get_device_properties(index), "gcnArchName", "") or ""
        ok = _DEVICE_OK[index] = arch.split(":")[0] == "gfx950"
        if not ok:
            import logging
            logging.warning(f"comfyui-gguf_amd: cuda:{index} is {arch or 'not an AMD GPU'}; the HIP kernels are gfx950 (MI355X) only -- "
                            "requests on that device keep the reference's torch path")
    return ok


def _launch(qtype, data, n_blocks, out, compute_code, out_code, entry=None):
    """Enqueue on torch's CURRENT stream of data's device: orders after the H2D copy, before F.linear.
    ``entry``: the C entry point to call (None = ggq_dequant through the fastest binding there is)."""
    index = data.device.index
    if not (_DEVICE_OK.get(index) or _device_served(index)):
        raise GGQUnsupported(f"cuda:{index} is not a gfx950 device")
    if _cur_device() != index:
        with torch.cuda.device(index):
            return _launch(qtype, data, n_blocks, out, compute_code, out_code, entry)
    rc = (entry or _ggq_dequant or _bind())(qtype, data.data_ptr(), n_blocks, out.data_ptr(), compute_code, out_code, _raw_stream(index))
    if rc:
        _native.check(rc, f"ggq_dequant({Q(int(qtype)).name})")


def _check_compute(dtype):
    try:
        return _COMPUTE_CODE[dtype]
    except (KeyError, TypeError):
        raise GGQUnsupported(f"dequant_dtype={dtype}: the HIP kernels compute in float16, bfloat16 or float32") from None


# ---- torch.compile: the launch as an opaque custom op ------------------------------------------------
# The reference lets torch >= 2.8 compile straight through its forward (ops.py:20-42).  A ctypes call is
# not traceable, so while Dynamo is tracing the launch goes through `ggq::dequantize`, a
# torch.library custom op with a fake (meta) implementation; eager calls keep the direct ctypes path
# (no dispatcher overhead on the per-layer hot loop).
_TORCH_OF_CODE = {v: k for k, v in _OUT_CODE.items()}


def _dequantize_op_impl(data: torch.Tensor, qtype: int, compute: int, out: int) -> torch.Tensor:
    block_size, type_size = GGML_QUANT_SIZES[Q(qtype)]
    n_blocks = data.numel() // type_size
    res = torch.empty(n_blocks * block_size, dtype=_TORCH_OF_CODE[out], device=data.device)
    if n_blocks:
        if not data.is_contiguous() or data.data_ptr() % 16:
            data = data.contiguous().clone()
        _launch(qtype, data, n_blocks, res, compute, out)
    return res


def _dequantize_op_fake(data, qtype, compute, out):
    block_size, type_size = GGML_QUANT_SIZES[Q(qtype)]
    return data.new_empty((data.numel() // type_size) * block_size, dtype=_TORCH_OF_CODE[out])


try:
    _dequantize_op = torch.library.custom_op("ggq::dequantize", _dequantize_op_impl, mutates_args=(), device_types="cuda")
    _dequantize_op.register_fake(_dequantize_op_fake)
except (AttributeError, RuntimeError):       # torch without torch.library.custom_op: eager path only
    _dequantize_op = None


# qtype (IntEnum member of this package, of the gguf package, or a plain int: all hash alike) ->
# (ggml type id, block_size, type_size)
_HIP_TABLE = {k: (int(k),) + tuple(GGML_QUANT_SIZES[k]) for k in HIP_QTYPES}


def _dequant_hip(data, qtype, out_dtype, compute=None, oshape=None, entry=None):
    """Packed device bytes -> dense tensor of ``out_dtype``: the block function's op sequence in the
    ``compute`` dtype (None = fp16), then one cast to ``out_dtype``, in one kernel.  Result is flat, or
    of ``oshape`` when given (must hold exactly the dequantized elements, as the reference's reshape)."""
    ent = _HIP_TABLE.get(qtype)
    if ent is None:
        key = _qtype_key(qtype)
        if key not in _HIP_TABLE:
            raise GGQUnsupported(f"no HIP unpacker for qtype {getattr(qtype, 'name', qtype)!r}")
        ent = _HIP_TABLE[key]
    qid, block_size, type_size = ent
    compute_code = _check_compute(compute)
    out_code = _OUT_CODE[out_dtype]
    if _dequantize_op is not None and _is_compiling():
        with _NoTorchFunction():                            # (the GGMLTensor subclass must not claim the result: a plain dense tensor comes back)
            res = _dequantize_op(_as_bytes(data, align=False), qid, compute_code, out_code)
            return res if oshape is None else res.reshape(oshape)
    with _NoTorchFunction():
        if not data.is_cuda:
            raise GGQUnsupported(f"packed data is on {data.device}; the HIP path serves GPU-resident weights only")
        if data.dtype is not torch.uint8 or not data.is_contiguous() or data.data_ptr() & 15:
            data = _as_bytes(data)                          # views, other storage dtypes, misaligned starts
        n_blocks = data.numel() // type_size                # dequant.py:41
        n = n_blocks * block_size
        if oshape is None:
            out = torch.empty(n, dtype=out_dtype, device=data.device)
        else:
            out = torch.empty(oshape, dtype=out_dtype, device=data.device)
            if out.numel() != n:
                raise RuntimeError(f"shape '{list(oshape)}' is invalid for input of size {n}")   # what .reshape(oshape) raises
        if n_blocks:
            _launch(qid, data, n_blocks, out, compute_code, out_code, entry)
    return out


def dequantize(data, qtype, oshape, dtype=None):
    """Dequantize tensor back to usable shape/dtype (dequant.py:30-44).

    ``dtype`` is the reference's *arithmetic* dtype (its ``dequant_dtype``): None / float16 is the
    stock fp16 path and returns float16; bfloat16 / float32 run the same op sequence in that dtype and
    return it.
    """
    key = _qtype_key(qtype)
    if key == Q.BF16:
        return dequantize_blocks_BF16(_as_bytes(data), 1, 2, dtype).reshape(oshape)
    _check_compute(dtype)
    return _dequant_hip(data, qtype, _COMPUTE_TORCH[dtype], compute=dtype, oshape=oshape)


def dequantize_tensor(tensor, dtype=None, dequant_dtype=None, _entry=None):
    """dequant.py:15-28, same argument meaning and result.  (``_entry``, private: another C entry point with ggq_dequant's
    signature for the launch -- how dequantize_tensor_streaming asks for non-temporal stores without touching any shared state.)"""
    qtype = getattr(tensor, "tensor_type", None)
    try:
        ent = _HIP_TABLE.get(qtype)
    except TypeError:                                   # an unhashable "qtype": the general path names it in its error
        ent = None
    if ent is not None and dtype in _OUT_CODE and not _is_compiling():
        # ---- the per-layer hot loop (ops.py:177): a quantized GGMLTensor on the GPU, a result dtype the kernels emit.  Everything
        # the general path below does, flattened into one frame: each avoided call / lookup is ~0.1 us of a ~7 us call.
        cd = dtype if dequant_dtype == "target" else dequant_dtype
        try:
            compute_code = _COMPUTE_CODE[cd]
        except (KeyError, TypeError):
            raise GGQUnsupported(f"dequant_dtype={cd}: the HIP kernels compute in float16, bfloat16 or float32") from None
        oshape = getattr(tensor, "tensor_shape", None)
        if oshape is not None and isinstance(tensor, torch.Tensor):
            qid, block_size, type_size = ent
            with _NoTorchFunction():
                data = tensor
                if not data.is_cuda:
                    raise GGQUnsupported(f"packed data is on {data.device}; the HIP path serves GPU-resident weights only")
                if data.dtype is not torch.uint8 or not data.is_contiguous() or data.data_ptr() & 15:
                    data = _as_bytes(data)                      # views, other storage dtypes, misaligned starts
                n_blocks = data.numel() // type_size            # dequant.py:41
                device = data.device
                out = torch.empty(oshape, dtype=dtype, device=device)
                if out.numel() != n_blocks * block_size:
                    raise RuntimeError(f"shape '{list(oshape)}' is invalid for input of size {n_blocks * block_size}")   # what .reshape(oshape) raises
                if n_blocks:
                    index = device.index
                    if not (_DEVICE_OK.get(index) or _device_served(index)):
                        raise GGQUnsupported(f"cuda:{index} is not a gfx950 device")
                    if _cur_device() != index:
                        _launch(qid, data, n_blocks, out, compute_code, _OUT_CODE[dtype], _entry)
                    else:
                        rc = (_entry or _ggq_dequant or _bind())(qid, data.data_ptr(), n_blocks, out.data_ptr(), compute_code, _OUT_CODE[dtype], _raw_stream(index))
                        if rc:
                            _native.check(rc, f"ggq_dequant({Q(qid).name})")
            return out
    return _dequantize_tensor_general(tensor, dtype, dequant_dtype, qtype, _entry)


def dequantize_tensor_streaming(tensor, dtype=None, dequant_dtype=None):
    """``dequantize_tensor`` for a result that is NOT read back soon (a tensor unpacked at load time, a measurement of the unpack
    alone): same kernels and values, non-temporal stores (include/ggq.h ``ggq_dequant_stream``).  The per-layer path stores write-through (sc1),
    because the layer's GEMM reads the weight next and finds it in cache.  Re-entrant and thread-safe: the entry point travels down the
    call as an argument, no module state changes."""
    return dequantize_tensor(tensor, dtype, dequant_dtype, _native.lib().ggq_dequant_stream)


def _dequantize_tensor_general(tensor, dtype, dequant_dtype, qtype, entry=None):
    """dequant.py:15-28 for everything the hot path above does not take: passthrough types, plain-int qtypes, result dtypes the
    kernels do not emit, carriers that are not tensors, BF16, tracing under torch.compile, unknown qtypes."""
    oshape = getattr(tensor, "tensor_shape", None)
    if oshape is None:
        oshape = tensor.shape

    if qtype in TORCH_COMPATIBLE_QTYPES:
        return tensor.to(dtype)
    key = qtype if qtype in _HIP_TABLE else _qtype_key(qtype)
    if key in _HIP_TABLE:
        dequant_dtype = dtype if dequant_dtype == "target" else dequant_dtype
        _check_compute(dequant_dtype)
        # the packed bytes are read straight off the GGMLTensor (the reference's `tensor.data`, dequant.py:23;
        # `.data` on a Tensor subclass is one more torch-function round trip for the same storage)
        data = tensor if isinstance(tensor, torch.Tensor) else tensor.data
        if dtype in _OUT_CODE:
            # dequantize(..., dtype=dequant_dtype).to(dtype) with the cast fused into the kernel's store
            return _dequant_hip(data, key, dtype, compute=dequant_dtype, oshape=oshape, entry=entry)
        return _dequant_hip(data, key, _COMPUTE_TORCH[dequant_dtype], compute=dequant_dtype, oshape=oshape, entry=entry).to(dtype)
    if key == Q.BF16:
        return dequantize(tensor.data, key, oshape, dtype=dequant_dtype).to(dtype)
    raise GGQUnsupported(f"no HIP unpacker for qtype {getattr(qtype, 'name', qtype)!r} "
                         "(the reference falls back to gguf's numpy dequantize here, dequant.py:24-28)")


def dequantize_tensor_via_gpu(tensor, dtype=None, dequant_dtype=None, device=None):
    """``dequantize_tensor`` for a CPU-RESIDENT quantized tensor, computed on the GPU: upload the packed bytes, unpack there,
    copy the dense result back -- the result lives on the CPU, as the reference's does, and holds the same bits.  For the
    load-time callers that dequantize big CPU tensors (token_embd / mmproj, reference loader.py:253-254,270,386,397): a T5-xxl
    embedding table is 131 M elements, ~0.3-0.6 s of torch-CPU ops against tens of milliseconds of PCIe traffic.
    Raises GGQUnsupported for anything that is not a CPU tensor of a type with a kernel (the caller keeps the reference's path)."""
    qtype = getattr(tensor, "tensor_type", None)
    key = qtype if qtype in _HIP_TABLE else _qtype_key(qtype)
    if key not in _HIP_TABLE or not isinstance(tensor, torch.Tensor):
        raise GGQUnsupported("GPU route: a quantized tensor of a type with a HIP unpacker")
    if dtype not in _OUT_CODE and dtype is not None:
        raise GGQUnsupported("GPU route: fp16 / bf16 / fp32 results")
    oshape = getattr(tensor, "tensor_shape", None)
    with _NoTorchFunction():
        if tensor.device.type != "cpu":
            raise GGQUnsupported("GPU route is for CPU-resident tensors")
        if device is None:
            if not torch.cuda.is_available():
                raise GGQUnsupported("no GPU to route through")
            device = torch.device("cuda", torch.cuda.current_device())
        data = _as_bytes(tensor, align=False).to(device, non_blocking=False)
    from .ops import GGMLTensor
    carrier = GGMLTensor(data, tensor_type=key, tensor_shape=oshape if oshape is not None else tensor.shape)
    return dequantize_tensor(carrier, dtype, dequant_dtype).cpu()


import os as _os


def _check_indices_default(dropin):
    """GGQ_CHECK_INDICES: 1 / 0 forces / forbids the asynchronous device-side assert on token ids outside the table.  Unset: ON when the call stands in for
    ``F.embedding`` under install() (the reference fails on such an id -- a tokenizer / vocabulary mismatch -- and so must the drop-in), OFF for direct calls
    of this function (the kernel clamps)."""
    v = _os.environ.get("GGQ_CHECK_INDICES")
    return dropin if v is None or v == "" else v != "0"


def dequantize_rows(tensor, indices, dtype=None, dequant_dtype=None, check_indices=None):
    """``F.embedding(indices, dequantize_tensor(tensor, dtype, dequant_dtype))`` without unpacking the whole table: only the
    rows ``indices`` names are dequantized (include/ggq.h ``ggq_dequant_rows``), bit-identical values.  What
    ``GGMLOps.Embedding.forward_ggml_cast_weights`` computes (ops.py:251-260) in two steps.  ``tensor``: quantized, logical
    shape (n_rows, cols), GPU-resident; ``indices``: integer tensor on the same device; result: indices.shape + (cols,).
    Raises GGQUnsupported for anything else (the caller keeps dequantize_tensor + F.embedding).  Ids outside the table are CLAMPED by the kernel;
    ``check_indices`` adds the device-side assert ``F.embedding`` would raise (default: GGQ_CHECK_INDICES, else off here / on under install())."""
    qtype = getattr(tensor, "tensor_type", None)
    key = qtype if qtype in _HIP_TABLE else _qtype_key(qtype)
    if key not in _HIP_TABLE:
        raise GGQUnsupported(f"no HIP unpacker for qtype {getattr(qtype, 'name', qtype)!r}")
    shape = tuple(getattr(tensor, "tensor_shape", ()))
    dequant_dtype = dtype if dequant_dtype == "target" else dequant_dtype
    compute_code = _check_compute(dequant_dtype)
    out_dtype = _COMPUTE_TORCH[dequant_dtype] if dtype is None else dtype
    if len(shape) != 2 or out_dtype not in _OUT_CODE:
        raise GGQUnsupported("row lookup: a 2-D table and an fp16 / bf16 / fp32 result")
    qid, block_size, type_size = _HIP_TABLE[key]
    n_rows, cols = shape
    if cols % block_size:
        raise GGQUnsupported("row lookup: every row must be whole blocks")
    data = tensor if isinstance(tensor, torch.Tensor) else tensor.data
    with _NoTorchFunction():
        if not data.is_cuda or not indices.is_cuda or indices.dtype not in (torch.int64, torch.int32):
            raise GGQUnsupported("row lookup: GPU-resident table and int32 / int64 indices on the GPU")
        if data.dtype is not torch.uint8 or not data.is_contiguous() or data.data_ptr() & 15:
            data = _as_bytes(data)
        if data.numel() != n_rows * (cols // block_size) * type_size:
            raise GGQUnsupported("row lookup: packed bytes do not match the logical shape")
        idx = indices if (indices.dtype is torch.int64 and indices.is_contiguous()) else indices.to(torch.int64).contiguous()
        if _check_indices_default(False) if check_indices is None else check_indices:
            # F.embedding raises a device-side assert on an id outside the table; the kernel clamps.  The check restores the failure
            # (asynchronously, like torch's own assert): on by default under install(), where this call stands in for F.embedding.
            torch._assert_async(((idx >= 0) & (idx < n_rows)).all(), "ggq: token id outside the embedding table")
        out = torch.empty(tuple(indices.shape) + (cols,), dtype=out_dtype, device=data.device)
        n_idx = idx.numel()
        if n_idx:
            index = data.device.index
            if not (_DEVICE_OK.get(index) or _device_served(index)):
                raise GGQUnsupported(f"cuda:{index} is not a gfx950 device")
            args = (qid, data.data_ptr(), n_rows, cols // block_size, idx.data_ptr(), n_idx, out.data_ptr(), compute_code, _OUT_CODE[out_dtype])
            if _cur_device() != index:
                with torch.cuda.device(index):
                    rc = _native.lib().ggq_dequant_rows(*args, _raw_stream(index))
            else:
                rc = _native.lib().ggq_dequant_rows(*args, _raw_stream(index))
            if rc:
                _native.check(rc, "ggq_dequant_rows")
    return out


# ---- torch.compile: the row lookup as an opaque custom op (the Embedding wrapper of install(); same reasons as ggq::dequantize above)
def _dequantize_rows_op_impl(packed: torch.Tensor, indices: torch.Tensor, qtype: int, n_rows: int, cols: int, compute: int, out: int, check: bool) -> torch.Tensor:
    from .ops import GGMLTensor
    table = GGMLTensor(packed, tensor_type=Q(qtype), tensor_shape=(n_rows, cols))
    cd = {v: k for k, v in _COMPUTE_CODE.items() if k is not None}[compute]
    return dequantize_rows(table, indices, _TORCH_OF_CODE[out], cd, check_indices=check)


def _dequantize_rows_op_fake(packed, indices, qtype, n_rows, cols, compute, out, check):
    return packed.new_empty(tuple(indices.shape) + (cols,), dtype=_TORCH_OF_CODE[out])


try:
    _dequantize_rows_op = torch.library.custom_op("ggq::dequantize_rows", _dequantize_rows_op_impl, mutates_args=(), device_types="cuda")
    _dequantize_rows_op.register_fake(_dequantize_rows_op_fake)
except (AttributeError, RuntimeError):
    _dequantize_rows_op = None


_TRACE_ANY_DEVICE = False      # tests only (see fused._TRACE_ANY_DEVICE)


def dequantize_rows_traced(tensor, indices, dtype, dequant_dtype, check_indices):
    """``dequantize_rows`` while torch.compile traces install()'s Embedding wrapper: the custom op, or None when the eager call would have raised
    GGQUnsupported (the caller then traces the reference's method).  Conditions on trace-time constants only."""
    if _dequantize_rows_op is None:
        return None
    key = _qtype_key(getattr(tensor, "tensor_type", None))
    shape = tuple(getattr(tensor, "tensor_shape", ()))
    cd = dtype if dequant_dtype == "target" else dequant_dtype
    if key not in _HIP_TABLE or len(shape) != 2 or cd not in _COMPUTE_CODE:
        return None
    out_dtype = _COMPUTE_TORCH[cd] if dtype is None else dtype
    qid, block_size, type_size = _HIP_TABLE[key]
    n_rows, cols = int(shape[0]), int(shape[1])
    if out_dtype not in _OUT_CODE or cols % block_size or not ((tensor.is_cuda and indices.is_cuda) or _TRACE_ANY_DEVICE) or indices.dtype not in (torch.int64, torch.int32):
        return None
    with _NoTorchFunction():
        packed = _as_bytes(tensor, align=False)
        if packed.numel() != n_rows * (cols // block_size) * type_size:
            return None
        return _dequantize_rows_op(packed, indices, qid, n_rows, cols, _COMPUTE_CODE[cd], _OUT_CODE[out_dtype], bool(check_indices))


# ---- dequantize_functions: the reference's per-format block functions (dequant.py:287-301) --------

def dequantize_blocks_BF16(blocks, block_size, type_size, dtype=None):
    """dequant.py:61-62: (int16 -> int32 << 16) viewed as fp32 == an exact bf16 -> fp32 widening."""
    return blocks.reshape(-1).view(torch.uint8).view(torch.bfloat16).to(torch.float32).reshape((-1, 1))


def _make_block_fn(key):
    def fn(blocks, block_size, type_size, dtype=None):
        """(n_blocks, type_size) uint8 -> (n_blocks, block_size) of ``dtype`` (None: float16), on the GPU."""
        _check_compute(dtype)
        bs, ts = GGML_QUANT_SIZES[key]
        if (block_size, type_size) != (bs, ts):
            raise ValueError(f"{key.name}: expected block geometry {(bs, ts)}, got {(block_size, type_size)}")
        return _dequant_hip(blocks, key, _COMPUTE_TORCH[dtype], compute=dtype).reshape((-1, bs))
    fn.__name__ = fn.__qualname__ = f"dequantize_blocks_{key.name}"
    return fn


dequantize_functions = {Q.BF16: dequantize_blocks_BF16}
dequantize_functions.update({k: _make_block_fn(k) for k in HIP_QTYPES})
for _k in HIP_QTYPES:
    globals()[f"dequantize_blocks_{_k.name}"] = dequantize_functions[_k]
