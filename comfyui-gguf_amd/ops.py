"""The two boundary types the hot path is called through, in the smallest form that needs no
ComfyUI import: ``GGMLTensor`` (reference ops.py:44-91) and the ``get_weight`` -> ``F.linear``
call chain of ``GGMLOps.Linear`` (reference ops.py:166-191, 242-244).

In a real ComfyUI install the reference's own classes stay in place (they are "reused as-is",
SURVEY.md section 2 rows 7-8) and only the dequant functions underneath them are replaced by
install.py.  These stand-ins exist so the path can be driven end to end -- tests, smoke, bench --
on a machine that has neither ComfyUI nor the reference checked out (the GPU box).  LoRA patches,
state-dict hooks and the other layer types belong to the reference's control plane and are not
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


class GGMLLinear(torch.nn.Module):
    """``GGMLOps.Linear`` reduced to the hot path: every forward re-dequantizes the weight
    (no caching, ops.py:166-191) on the device the packed bytes live on, then ``F.linear``."""

    dequant_dtype = None                           # GGMLLayer.dequant_dtype, ops.py:98

    def __init__(self, weight, bias=None):
        super().__init__()
        self.weight = weight
        self.bias = bias

    def get_weight(self, tensor, dtype):           # ops.py:166-181 without the LoRA branch
        if tensor is None:
            return None
        weight = dequantize_tensor(tensor, dtype, self.dequant_dtype)
        if isinstance(weight, GGMLTensor):
            weight = weight.as_subclass(torch.Tensor)
        return weight

    def cast_bias_weight(self, input):             # ops.py:194-211
        device, dtype = input.device, input.dtype
        bias = self.get_weight(self.bias.to(device), dtype) if self.bias is not None else None
        weight = self.get_weight(self.weight.to(device), dtype)
        return weight, bias

    def forward(self, input):                      # ops.py:242-244
        if not (is_quantized(self.weight) or is_quantized(self.bias)):
            return torch.nn.functional.linear(input, self.weight.to(input.dtype), None if self.bias is None else self.bias.to(input.dtype))
        weight, bias = self.cast_bias_weight(input)
        return torch.nn.functional.linear(input, weight, bias)
