"""Opt-in: ``F.linear`` on a quantized weight straight from the packed blocks -- the dense weight is never written.
Two kernels: ``linear_small`` for ONE TO FOUR rows of input (include/ggq.h ``ggq_linear_small``, csrc/ggq_linear.hpp: HBM-bound,
no matrix cores) and ``linear_mfma`` for many rows (``ggq_linear_mfma``, csrc/ggq_mfma.hpp: each lane decodes the eight
consecutive weights that are its MFMA operand).

The reference's ``GGMLOps.Linear.forward_ggml_cast_weights`` (ops.py:242-244) dequantizes the whole weight and calls
``F.linear`` whatever the input looks like.  FLUX's modulation linears see the conditioning vector (batch rows): 76 layers,
27 % of the model's weights, one row each.  For them the fused kernel moves 0.56 B per weight instead of 0.56 + 2 + 2.

The weights are the reference's values bit for bit (same decode, same fp16 op sequence, same ``.to(dtype)``); the dot products
accumulate in fp32 in this kernel's own order, so the result equals ``F.linear(x, dequantize_tensor(w, x.dtype), bias)`` up to
fp32 summation order -- parity is a tolerance against an fp32 reference (tests/test_gpu_linear.py), not bit-exact, hence opt-in:
``GGMLLinear.fuse_small_m = True`` for the stand-in, ``install(..., fused_small_m=True)`` for a ComfyUI-GGUF checkout.
"""
import torch

from . import _native
from .dequant import GGQUnsupported, _DEVICE_OK, _HIP_TABLE, _OUT_CODE, _device_served, _qtype_key, _raw_stream, dequantize_tensor, is_quantized

MAX_ROWS = 4


def linear_small(x, weight, bias=None, dequant_dtype=None):
    """x: (..., cols) on the GPU with at most MAX_ROWS rows in total; weight: GGMLTensor of logical shape (rows, cols).
    Raises GGQUnsupported for anything the kernel does not take (the caller keeps dequantize + F.linear)."""
    if dequant_dtype not in (None, torch.float16):
        raise GGQUnsupported("the fused linear computes the stock fp16 weight values only")
    if getattr(weight, "patches", None) or getattr(bias, "patches", None):
        # get_weight applies LoRA patches to the weight AND to the bias (ops.py:183-190 via ops.py:205-206)
        raise GGQUnsupported("LoRA-patched weight or bias: needs the reference's get_weight")
    qtype = getattr(weight, "tensor_type", None)
    key = qtype if qtype in _HIP_TABLE else _qtype_key(qtype)
    if key not in _HIP_TABLE:
        raise GGQUnsupported(f"no fused linear for qtype {getattr(qtype, 'name', qtype)!r}")
    shape = tuple(getattr(weight, "tensor_shape", ()))
    if len(shape) != 2 or not x.is_cuda or not weight.is_cuda or x.dtype not in _OUT_CODE or x.shape[-1] != shape[1]:
        raise GGQUnsupported("fused linear: 2-D weight, GPU tensors, fp16 / bf16 / fp32 input of matching width")
    rows, cols = shape
    m = x.numel() // cols if cols else 0
    if not 1 <= m <= MAX_ROWS:
        raise GGQUnsupported(f"fused linear takes 1..{MAX_ROWS} input rows, got {m}")
    xf = x.reshape(m, cols)
    if not xf.is_contiguous() or xf.data_ptr() & 15:
        xf = xf.contiguous().clone()
    if bias is not None:
        if is_quantized(bias):
            bias = dequantize_tensor(bias, x.dtype)
        bias = bias.to(device=x.device, dtype=x.dtype).contiguous()
        if bias.numel() != rows:
            raise GGQUnsupported("bias does not match the weight's rows")
    with torch._C.DisableTorchFunctionSubclass():
        if weight.dtype is not torch.uint8 or not weight.is_contiguous() or weight.data_ptr() & 15:
            raise GGQUnsupported("packed weight must be a contiguous, 16-byte aligned byte tensor")
        index = x.device.index
        if not (_DEVICE_OK.get(index) or _device_served(index)):
            raise GGQUnsupported(f"cuda:{index} is not a gfx950 device")
        y = torch.empty((m, rows), dtype=x.dtype, device=x.device)
        with torch.cuda.device(index):
            rc = _native.lib().ggq_linear_small(_HIP_TABLE[key][0], weight.data_ptr(), rows, cols, xf.data_ptr(), m,
                                                None if bias is None else bias.data_ptr(), y.data_ptr(), _OUT_CODE[x.dtype], _raw_stream(index))
    if rc == _native.GGQ_ERR_ARG:
        raise GGQUnsupported("shape outside what the fused kernel stages in LDS")
    _native.check(rc, "ggq_linear_small")
    return y.reshape(*x.shape[:-1], rows)


def linear_mfma(x, weight, bias=None, dequant_dtype=None, tile_rows=0):
    """``F.linear(x, dequantize_tensor(weight, x.dtype), bias)`` for any number of rows of x, on the matrix cores, from the packed
    blocks (include/ggq.h ``ggq_linear_mfma``).  x: (..., cols) fp16 / bf16 on the GPU; weight: GGMLTensor of logical shape
    (rows, cols) with cols % 256 == 0.  Weights bit-identical to the reference's; fp32 accumulation in the kernel's own order
    (tolerance parity, tests/test_gpu_mfma.py).  Raises GGQUnsupported for anything the kernel does not take."""
    if dequant_dtype not in (None, torch.float16):
        raise GGQUnsupported("the fused GEMM computes the stock fp16 weight values only")
    if getattr(weight, "patches", None) or getattr(bias, "patches", None):
        raise GGQUnsupported("LoRA-patched weight or bias: needs the reference's get_weight")
    qtype = getattr(weight, "tensor_type", None)
    key = qtype if qtype in _HIP_TABLE else _qtype_key(qtype)
    if key not in _HIP_TABLE:
        raise GGQUnsupported(f"no fused GEMM for qtype {getattr(qtype, 'name', qtype)!r}")
    shape = tuple(getattr(weight, "tensor_shape", ()))
    if (len(shape) != 2 or not x.is_cuda or not weight.is_cuda or x.dtype not in (torch.float16, torch.bfloat16)
            or x.shape[-1] != shape[1] or shape[1] % 256 or x.numel() == 0):
        raise GGQUnsupported("fused GEMM: 2-D weight with cols % 256 == 0, GPU tensors, fp16 / bf16 input of matching width")
    rows, cols = shape
    m = x.numel() // cols
    xf = x.reshape(m, cols)
    if not xf.is_contiguous() or xf.data_ptr() & 15:
        xf = xf.contiguous().clone()
    if bias is not None:
        if is_quantized(bias):
            bias = dequantize_tensor(bias, x.dtype)
        bias = bias.to(device=x.device, dtype=x.dtype).contiguous()
        if bias.numel() != rows:
            raise GGQUnsupported("bias does not match the weight's rows")
    with torch._C.DisableTorchFunctionSubclass():
        if weight.dtype is not torch.uint8 or not weight.is_contiguous() or weight.data_ptr() & 15:
            raise GGQUnsupported("packed weight must be a contiguous, 16-byte aligned byte tensor")
        index = x.device.index
        if not (_DEVICE_OK.get(index) or _device_served(index)):
            raise GGQUnsupported(f"cuda:{index} is not a gfx950 device")
        y = torch.empty((m, rows), dtype=x.dtype, device=x.device)
        with torch.cuda.device(index):
            rc = _native.lib().ggq_linear_mfma(_HIP_TABLE[key][0], weight.data_ptr(), rows, cols, xf.data_ptr(), m,
                                               None if bias is None else bias.data_ptr(), y.data_ptr(), _OUT_CODE[x.dtype], int(tile_rows), _raw_stream(index))
    if rc == _native.GGQ_ERR_ARG:
        raise GGQUnsupported("shape outside what the fused GEMM takes")
    _native.check(rc, "ggq_linear_mfma")
    return y.reshape(*x.shape[:-1], rows)
