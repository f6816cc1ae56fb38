"""Optional: keep dequantized weights RESIDENT in HBM instead of re-dequantizing them on every forward.

The reference re-dequantizes every quantized layer on every forward because the GPUs it targets cannot
hold the dense weights (ops.py:166-191 -- no caching).  An MI355X has 288 GB: FLUX.1-dev's 304 quantized
linears are 6.8 GB packed and 23.7 GB dense.  With a byte budget, ``DenseCache`` memoises
``dequantize_tensor(tensor, dtype, dequant_dtype)`` per packed tensor object, so the dequant kernels run
once per (tensor, mode) instead of once per layer per denoising step -- the 5.3 ms per FLUX step the path
costs in context (profiles/r01_flux_forward_emulation.json) go away.  The values are the kernels' own
output, so results stay bit-identical.

It is OPT-IN (``install(..., dense_cache_gb=N)`` or ``GGQ_DENSE_CACHE_GB=N``) because it changes two things a
ComfyUI user may rely on: VRAM use (dense weights stay allocated until evicted / ``clear()``), and aliasing
-- the same dense tensor object is handed out on every call, so it must be treated as read-only.  The one
caller in the reference that writes into the dequantized weight, the LoRA branch of ``get_weight``
(``comfy.lora.calculate_weight`` patches in place, ops.py:183-190), is detected through the tensor's
``patches`` attribute and always gets a fresh tensor.

An entry is valid for exactly one packed tensor OBJECT in one state: it is keyed by ``id(tensor)``, holds a
weak reference to it (a recycled id never matches, and the entry dies with the tensor) and records the
tensor's in-place version counter (a ``copy_`` into the packed bytes invalidates it).
"""
import collections
import weakref

import torch


class DenseCache:
    def __init__(self, budget_bytes, dequantize_tensor, require_gpu=True):
        self.budget = int(budget_bytes)
        self._fn = dequantize_tensor
        self._require_gpu = require_gpu               # False only in the CPU unit tests of the bookkeeping
        self._entries = collections.OrderedDict()     # key -> (weakref to packed tensor, version, dense)
        self.bytes = 0
        self.hits = self.misses = self.bypassed = 0

    @staticmethod
    def _nbytes(t):
        return t.numel() * t.element_size()

    def _drop(self, key):
        ent = self._entries.pop(key, None)
        if ent is not None:
            self.bytes -= self._nbytes(ent[2])

    def clear(self):
        self._entries.clear()
        self.bytes = 0

    def __call__(self, tensor, dtype=None, dequant_dtype=None):
        # not cacheable: anything that is not a quantized GGML tensor living on a GPU, or one with LoRA patches
        if (not isinstance(tensor, torch.Tensor) or getattr(tensor, "tensor_type", None) is None
                or getattr(tensor, "patches", None) or (self._require_gpu and not tensor.is_cuda)):
            self.bypassed += 1
            return self._fn(tensor, dtype, dequant_dtype)
        key = (id(tensor), dtype, dequant_dtype)
        ent = self._entries.get(key)
        if ent is not None:
            if ent[0]() is tensor and ent[1] == tensor._version:
                self._entries.move_to_end(key)
                self.hits += 1
                return ent[2]
            self._drop(key)
        dense = self._fn(tensor, dtype, dequant_dtype)
        self.misses += 1
        if not isinstance(dense, torch.Tensor) or dense.data_ptr() == tensor.data_ptr():
            return dense                               # passthrough types (F16 / F32 weights): nothing was produced
        size = self._nbytes(dense)
        if size > self.budget:
            return dense
        while self.bytes + size > self.budget and self._entries:
            self._drop(next(iter(self._entries)))      # least recently used first
        ref = weakref.ref(tensor, lambda _r, k=key: self._drop(k))
        self._entries[key] = (ref, tensor._version, dense)
        self.bytes += size
        return dense

    def stats(self):
        return {"entries": len(self._entries), "bytes": self.bytes, "hits": self.hits, "misses": self.misses, "bypassed": self.bypassed}
