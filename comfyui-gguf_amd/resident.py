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

Low-VRAM mode (weights kept on the CPU, ``s.weight.to(device)`` on every forward, ops.py:209): ``GGMLTensor.to`` hands out
a NEW tensor object per forward (ops.py:57-62), so nothing can ever hit -- every entry dies, unused, with the temporary it
was keyed on, within the very call chain that created it.  The cache notices (``EPHEMERAL_STREAK`` entries in a row that died
without a single hit AND within ``EPHEMERAL_WINDOW`` cache calls of their insertion) and stands aside: it stops inserting and
only probes every ``PROBE_EVERY``-th call, so that mode pays one integer compare per call instead of the bookkeeping.  Any hit
ends the streak.  A RESIDENT model that runs one forward and is then freed (a text encoder that encodes once and is unloaded)
also dies without hits, but many calls after its entries were made: those deaths do not count, so the next model is cached from
its first layer.  ``stats()["standing_aside"]`` reports the state.  (A packed byte address is not a usable key for the
temporaries: the caching allocator recycles their addresses between layers.)
"""
import collections
import weakref

import torch


EPHEMERAL_STREAK = 16      # entries in a row that died without a hit -> the callers hand in per-call temporaries
EPHEMERAL_WINDOW = 8        # ... counted only when the entry died within this many cache calls of its insertion (a per-call temporary)
PROBE_EVERY = 64           # ... then only every 64th cacheable call is inserted, to notice when that changes


class DenseCache:
    def __init__(self, budget_bytes, dequantize_tensor, require_gpu=True):
        self.budget = int(budget_bytes)
        self._fn = dequantize_tensor
        self._require_gpu = require_gpu               # False only in the CPU unit tests of the bookkeeping
        self._entries = collections.OrderedDict()     # key -> [weakref to packed tensor, version, dense, hits, call number of the insertion]
        self._calls = 0                               # cacheable calls so far (the clock of EPHEMERAL_WINDOW)
        self.bytes = 0
        self.hits = self.misses = self.bypassed = self.ephemeral_bypassed = 0
        self._streak = 0                              # consecutive entries that died with their tensor, never hit
        self._calls_while_aside = 0

    @staticmethod
    def _nbytes(t):
        return t.numel() * t.element_size()

    def _drop(self, key):
        ent = self._entries.pop(key, None)
        if ent is not None:
            self.bytes -= self._nbytes(ent[2])

    def _tensor_died(self, key):
        ent = self._entries.get(key)
        if ent is not None:
            if ent[3] > 0:
                self._streak = 0
            elif self._calls - ent[4] <= EPHEMERAL_WINDOW:
                self._streak += 1                      # a per-call temporary (low-VRAM mode); an unused entry that lived longer says nothing
        self._drop(key)

    def clear(self):
        self._entries.clear()
        self.bytes = 0
        self._streak = 0

    def __call__(self, tensor, dtype=None, dequant_dtype=None):
        # not cacheable: anything that is not a quantized GGML tensor living on a GPU, or one with LoRA patches
        if (not isinstance(tensor, torch.Tensor) or getattr(tensor, "tensor_type", None) is None
                or getattr(tensor, "patches", None) or (self._require_gpu and not tensor.is_cuda)):
            self.bypassed += 1
            return self._fn(tensor, dtype, dequant_dtype)
        self._calls += 1
        key = (id(tensor), dtype, dequant_dtype)
        ent = self._entries.get(key)
        if ent is not None:
            if ent[0]() is tensor and ent[1] == tensor._version:
                self._entries.move_to_end(key)
                self.hits += 1
                ent[3] += 1
                self._streak = 0
                return ent[2]
            self._drop(key)
        if self._streak >= EPHEMERAL_STREAK:           # low-VRAM mode: a fresh tensor object per forward, nothing to hit
            self._calls_while_aside += 1
            if self._calls_while_aside % PROBE_EVERY:
                self.ephemeral_bypassed += 1
                return self._fn(tensor, dtype, dequant_dtype)
        dense = self._fn(tensor, dtype, dequant_dtype)
        self.misses += 1
        if not isinstance(dense, torch.Tensor) or dense.data_ptr() == tensor.data_ptr():
            return dense                               # passthrough types (F16 / F32 weights): nothing was produced
        size = self._nbytes(dense)
        if size > self.budget:
            return dense
        while self.bytes + size > self.budget and self._entries:
            self._drop(next(iter(self._entries)))      # least recently used first
        ref = weakref.ref(tensor, lambda _r, k=key: self._tensor_died(k))
        self._entries[key] = [ref, tensor._version, dense, 0, self._calls]
        self.bytes += size
        return dense

    def stats(self):
        return {"entries": len(self._entries), "bytes": self.bytes, "hits": self.hits, "misses": self.misses, "bypassed": self.bypassed,
                "ephemeral_bypassed": self.ephemeral_bypassed, "standing_aside": self._streak >= EPHEMERAL_STREAK}

    def scratch_bytes(self):
        """Device memory the cache holds that the reference's VRAM estimate knows nothing about."""
        return self.bytes
