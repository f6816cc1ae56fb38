"""The ``comfy`` symbols the reference's ops.py touches, as a stand-in package.

TEST INFRASTRUCTURE.  ComfyUI itself is not in this image (and is control plane, out of scope); the
reference's ``ops.py`` needs exactly:

    comfy.ops.manual_cast.{Linear, Conv2d, Embedding, LayerNorm, GroupNorm}   base classes with
        ``forward`` -> ``forward_comfy_cast_weights`` (ops.py:213-271)
    comfy.ops.cast_to(t, dtype, device, non_blocking=, copy=)                 (ops.py:207,210)
    comfy.model_management.device_supports_non_blocking(device)               (ops.py:204)
    comfy.lora.calculate_weight(patches, weight, key[, intermediate_dtype])   (ops.py:186,190)

``calculate_weight`` here applies "diff"-style patches the way ComfyUI does -- IN PLACE on the weight it is
handed (``weight += strength * diff``): that in-place update is what makes a shared dense cache unsafe for
patched tensors, so the fake has to do it too.  A patch is ``(strength, diff_tensor)``.
"""
import types

import torch


def build():
    """A fresh {module name: module} dict to put into sys.modules (monkeypatch.setitem in tests)."""
    comfy = types.ModuleType("comfy")
    ops = types.ModuleType("comfy.ops")

    class CastWeightBiasOp:
        comfy_cast_weights = False
        weight_function = []
        bias_function = []

    class _Dispatch:
        def forward(self, *args, **kwargs):           # comfy.ops: cast path when comfy_cast_weights is set, else the torch forward
            if self.comfy_cast_weights:
                return self.forward_comfy_cast_weights(*args, **kwargs)
            return super().forward(*args, **kwargs)

    class manual_cast:
        class Linear(_Dispatch, torch.nn.Linear, CastWeightBiasOp):
            def forward_comfy_cast_weights(self, x):
                return torch.nn.functional.linear(x, self.weight.to(x.dtype), None if self.bias is None else self.bias.to(x.dtype))

        class Conv2d(_Dispatch, torch.nn.Conv2d, CastWeightBiasOp):
            def forward_comfy_cast_weights(self, x):
                return self._conv_forward(x, self.weight.to(x.dtype), None if self.bias is None else self.bias.to(x.dtype))

        class Embedding(_Dispatch, torch.nn.Embedding, CastWeightBiasOp):
            bias = None                                # as comfy.ops.disable_weight_init.Embedding

            def forward_comfy_cast_weights(self, input, out_dtype=None):
                return torch.nn.functional.embedding(input, self.weight.to(out_dtype), self.padding_idx)

        class LayerNorm(_Dispatch, torch.nn.LayerNorm, CastWeightBiasOp):
            def forward_comfy_cast_weights(self, x):
                w = None if self.weight is None else self.weight.to(x.dtype)
                b = None if self.bias is None else self.bias.to(x.dtype)
                return torch.nn.functional.layer_norm(x, self.normalized_shape, w, b, self.eps)

        class GroupNorm(_Dispatch, torch.nn.GroupNorm, CastWeightBiasOp):
            def forward_comfy_cast_weights(self, x):
                return torch.nn.functional.group_norm(x, self.num_groups, self.weight.to(x.dtype), self.bias.to(x.dtype), self.eps)

    def cast_to(t, dtype=None, device=None, non_blocking=False, copy=False):
        if t is None:
            return None
        return t.to(device=device, dtype=dtype, non_blocking=non_blocking, copy=copy)

    def calculate_weight(patches, weight, key, intermediate_dtype=torch.float32, original_weights=None):
        for strength, diff in patches:
            weight += (float(strength) * diff.to(device=weight.device, dtype=intermediate_dtype)).to(weight.dtype)   # in place, as ComfyUI
        return weight

    ops.manual_cast = manual_cast
    ops.CastWeightBiasOp = CastWeightBiasOp
    ops.cast_to = cast_to
    mm = types.ModuleType("comfy.model_management")
    mm.device_supports_non_blocking = lambda device: False
    lora = types.ModuleType("comfy.lora")
    lora.calculate_weight = calculate_weight
    comfy.ops, comfy.model_management, comfy.lora = ops, mm, lora
    return {"comfy": comfy, "comfy.ops": ops, "comfy.model_management": mm, "comfy.lora": lora}
