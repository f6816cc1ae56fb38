"""The boundary types the hot path is called through, in the smallest form that needs no
ComfyUI import: ``GGMLTensor`` (reference ops.py:44-91), the ``get_weight`` / ``cast_bias_weight``
call chain of ``GGMLLayer`` (reference ops.py:166-211) and the five layer types that call it
(``GGMLOps.Linear / Conv2d / Embedding / LayerNorm / GroupNorm``, reference ops.py:227-271).

In a real ComfyUI install the reference's own classes stay in place (they are "reused as-is",
SURVEY.md section 2 rows 7-8) and only the dequant functions underneath them are replaced by
install.py.  These stand-ins exist so the path can be driven end to end -- tests, smoke, bench --
on a machine that has neither ComfyUI nor the reference checked out (the GPU box).  LoRA patches,
state-dict hooks and VRAM accounting belong to the reference's control plane and are not
re-implemented here.
"""
import torch

from .dequant import dequantize_tensor, is_quantized


class GGMLTensor(torch.Tensor):
    """Packed GGUF bytes + the logical view: ``tensor_type`` (ggml type id), ``tensor_shape``."""

    def __new__(cls, data, *, tensor_type, tensor_shape, patches=None):
        t = torch.Tensor._make_subclass(cls, data, False)
        t.tensor_type = tensor_type
        t.tensor_shape = torch.Size(tensor_shape)
        t.patches = list(patches or [])
        return t

    def _carry(self, new):
        new.tensor_type = getattr(self, "tensor_type", None)
        new.tensor_shape = getattr(self, "tensor_shape", new.size())
        new.patches = list(getattr(self, "patches", []))
        return new

    def to(self, *args, **kwargs):                 # ops.py:57-62: the attrs survive a device move
        return self._carry(super().to(*args, **kwargs))

    def clone(self, *args, **kwargs):              # ops.py:64-68
        return self

    def detach(self, *args, **kwargs):
        return self

    def copy_(self, *args, **kwargs):              # ops.py:70-75: a failing in-place copy is logged, not raised
        try:
            return super().copy_(*args, **kwargs)
        except Exception as e:                     # noqa: BLE001 -- the reference swallows everything here
            import logging
            logging.warning(f"ignoring 'copy_' on tensor: {e}")

    def new_empty(self, size, *args, **kwargs):    # ops.py:77-85: same type and attrs, logical shape = the new size
        return GGMLTensor(super().new_empty(size, *args, **kwargs), tensor_type=getattr(self, "tensor_type", None),
                          tensor_shape=size, patches=getattr(self, "patches", []))

    @property
    def shape(self):                               # ops.py:87-91: the LOGICAL shape
        return getattr(self, "tensor_shape", self.size())


class GGMLLayer(torch.nn.Module):
    """What the reference's ``GGMLLayer`` does on the hot path (ops.py:93-211): keep the packed weight / bias, and on
    every forward re-dequantize them (no caching, ops.py:166-191) on the device the input lives on.  LoRA patches,
    state-dict hooks and VRAM accounting are the reference's control plane and are not restated."""

    dequant_dtype = None                           # ops.py:98; set by the loader nodes (nodes.py:152-157)
    _dequantize = staticmethod(dequantize_tensor)  # what get_weight calls; resident.DenseCache may be put here

    def __init__(self, weight, bias=None):
        super().__init__()
        self.weight = weight
        self.bias = bias

    def is_ggml_quantized(self):                   # ops.py:103-108
        return is_quantized(self.weight) or is_quantized(self.bias)

    def get_weight(self, tensor, dtype):           # ops.py:166-181 without the LoRA branch
        if tensor is None:
            return None
        weight = self._dequantize(tensor, dtype, self.dequant_dtype)
        if isinstance(weight, GGMLTensor):         # ops.py:180-181: never hand the subclass on
            weight = weight.as_subclass(torch.Tensor)
        return weight

    def cast_bias_weight(self, input=None, dtype=None, device=None, bias_dtype=None):   # ops.py:194-211
        if input is not None:
            if dtype is None:
                dtype = getattr(input, "dtype", torch.float32)
            if bias_dtype is None:
                bias_dtype = dtype
            if device is None:
                device = input.device
        bias = None
        if self.bias is not None:
            bias = self.get_weight(self.bias.to(device), dtype)
            bias = bias.to(device=device, dtype=bias_dtype)
        weight = self.get_weight(self.weight.to(device), dtype)
        weight = weight.to(device=device, dtype=dtype)
        return weight, bias


class GGMLLinear(GGMLLayer):
    """``GGMLOps.Linear`` (ops.py:227-244).  ``fuse_small_m`` / ``fuse_mfma_max_m`` (off in this stand-in; install()'s default over the reference's class, fused.py): inputs of at most four rows /
    of at most that many rows go through a fused dequantize + linear kernel instead of dequantize-then-F.linear."""

    fuse_small_m = False
    fuse_mfma_max_m = 0                            # opt-in: inputs of up to this many rows go through fused.linear_mfma

    def forward(self, input):
        if (self.fuse_small_m or self.fuse_mfma_max_m) and is_quantized(self.weight):
            from .fused import linear_auto                 # the policy of install()'s default (fused.py): GEMV / 16-row / 32-row MFMA kernel by rows of x
            from .dequant import GGQUnsupported
            try:
                return linear_auto(input, self.weight, self.bias, self.dequant_dtype, input.device, self.fuse_small_m, self.fuse_mfma_max_m)
            except GGQUnsupported:
                pass
        if not self.is_ggml_quantized():
            return torch.nn.functional.linear(input, self.weight.to(input.dtype), None if self.bias is None else self.bias.to(input.dtype))
        weight, bias = self.cast_bias_weight(input)
        return torch.nn.functional.linear(input, weight, bias)


class GGMLEmbedding(GGMLLayer):
    """``GGMLOps.Embedding`` (ops.py:251-260): the (possibly quantized) table is dequantized whole, then indexed.  As in the
    reference, a table stored in fp16 / bf16 is used in its own dtype and the result cast to ``out_dtype``."""

    def __init__(self, weight, padding_idx=None):
        super().__init__(weight, None)
        self.padding_idx = padding_idx

    gather_rows = True                             # quantized table on the GPU: unpack only the rows asked for (bit-identical)

    def forward(self, input, out_dtype=None):
        output_dtype = out_dtype
        if self.weight.dtype == torch.float16 or self.weight.dtype == torch.bfloat16:
            out_dtype = None
        if self.gather_rows and is_quantized(self.weight) and input.is_cuda and not getattr(self.weight, "patches", None):
            from .dequant import GGQUnsupported, dequantize_rows
            try:
                # cast_bias_weight(self, dtype=None) falls back to getattr(self, "dtype", float32) (ops.py:196-197): same here
                table_dtype = out_dtype if out_dtype is not None else getattr(self, "dtype", torch.float32)
                rows = dequantize_rows(self.weight.to(input.device), input, table_dtype, self.dequant_dtype)
                return rows.to(dtype=output_dtype)
            except GGQUnsupported:
                pass
        weight, _ = self.cast_bias_weight(self, device=input.device, dtype=out_dtype)
        return torch.nn.functional.embedding(input, weight, self.padding_idx).to(dtype=output_dtype)


class GGMLConv2d(GGMLLayer):
    """``GGMLOps.Conv2d`` (ops.py:246-249); the packed weight's logical shape is (out, in, kh, kw)."""

    def __init__(self, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
        super().__init__(weight, bias)
        self.stride, self.padding, self.dilation, self.groups = stride, padding, dilation, groups

    def forward(self, input):
        weight, bias = self.cast_bias_weight(input)
        return torch.nn.functional.conv2d(input, weight, bias, self.stride, self.padding, self.dilation, self.groups)


class GGMLLayerNorm(GGMLLayer):
    """``GGMLOps.LayerNorm`` (ops.py:262-266)."""

    def __init__(self, normalized_shape, weight, bias=None, eps=1e-5):
        super().__init__(weight, bias)
        self.normalized_shape, self.eps = tuple(normalized_shape), eps

    def forward(self, input):
        if self.weight is None:
            return torch.nn.functional.layer_norm(input, self.normalized_shape, None, None, self.eps)
        weight, bias = self.cast_bias_weight(input)
        return torch.nn.functional.layer_norm(input, self.normalized_shape, weight, bias, self.eps)


class GGMLGroupNorm(GGMLLayer):
    """``GGMLOps.GroupNorm`` (ops.py:268-271)."""

    def __init__(self, num_groups, weight, bias=None, eps=1e-5):
        super().__init__(weight, bias)
        self.num_groups, self.eps = num_groups, eps

    def forward(self, input):
        weight, bias = self.cast_bias_weight(input)
        return torch.nn.functional.group_norm(input, self.num_groups, weight, bias, self.eps)
