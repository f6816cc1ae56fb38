"""Opt-in: unpack the next few layers TOGETHER with the one that was asked for -- one kernel launch instead of K.

The reference calls ``dequantize_tensor`` once per quantized layer per forward (ops.py:177), so a FLUX.1-dev step is 304 separate
unpack launches of 5-18 us each.  On MI355X a dependent kernel boundary costs ~1.3 us whatever the kernels are (measured:
tools/probes/anyorder_probe.hip -- and ``hipExtAnyOrderLaunch``, which would let an unpack start while its predecessor still runs, is
not honoured on gfx950), and a single-tensor launch reads all of its packed bytes before its first store: per layer the path reaches
57-66 % of the HBM peak where the same kernels reach 80 % on a whole weight set.  ``DequantAhead`` learns the order in which
weights are asked for (it repeats every denoising step) and, on a call for weight A that it has no result for, unpacks A AND the
``depth - 1`` weights it expects next in ONE launch per format (include/ggq.h ``ggq_dequant_batch``: descriptors by value, nothing to
build or keep); the following calls are handed their tensors without a launch.

Measured (round 3, profiles/r03_flux_forward_emulation_fused_and_lookahead.json; bench.py ``workloads.per_layer``): with nothing between
the unpacks it helps (304 launches -> 76: 6151 -> 6326 GB/s eager), but inside a forward it LOSES -- an emulated FLUX.1-dev step is
+0.8 ms at depth 2 and +3.9 ms at depth 4 -- because a weight that was unpacked three layers early has left the Infinity Cache by the time
its GEMM reads it, and cache residency is what makes the default per-layer path nearly free (DESIGN.md section 4a, sc1 stores).  It stays
in the tree as a documented negative result and for callers that unpack without a GEMM in between.

What it changes, and why it is opt-in (``install(..., lookahead=K)`` / ``GGQ_LOOKAHEAD=K``):
  * values: nothing -- the same kernels, the same bits; every result is a FRESH tensor from torch's allocator that nobody else holds
    (so the LoRA branch of ``get_weight``, which patches the dequantized weight in place, ops.py:183-190, works unchanged);
  * memory: up to ``depth - 1`` dense weights are alive ahead of their use (3 x 130 MB for FLUX.1-dev at depth 4) -- VRAM the
    reference's estimate does not know about (``scratch_bytes()``);
  * a result is only handed out if the packed tensor is the same OBJECT in the same in-place version, asked for in the same
    (dtype, dequant_dtype) on the same stream as predicted; anything else (a changed order, another dtype, a weight that was written
    to) is recomputed on the spot, the stale result dropped.  Weights that live on the CPU (low-VRAM mode: a new GGMLTensor per forward,
    ops.py:209) are never predicted -- that mode is what overlap.py is for.  Tracing under torch.compile, HIP-graph capture and calls
    from another thread than the first one take the plain path.
"""
import threading
import weakref

import torch

from . import _native
from . import dequant as _dq


class _Entry:
    __slots__ = ("ref", "nxt", "mode", "pending", "__weakref__")

    def __init__(self, ref):
        self.ref = ref            # weak reference to the packed tensor (an id() is only trusted while ref() is that very object)
        self.nxt = None           # weak reference to the tensor that was asked for after this one last time
        self.mode = None          # (dtype, compute dtype, device index) of its last call
        self.pending = None       # (version, mode, raw stream, dense result) unpacked ahead of its call


class DequantAhead:
    """``ahead(tensor, dtype, dequant_dtype)`` == ``dequantize_tensor(tensor, dtype, dequant_dtype)`` (same values, same kind of
    result), with the launches of up to ``depth`` consecutive layers coalesced.  ``dequantize_tensor``: the HIP path's function
    (serves everything this class does not take; its GGQUnsupported passes straight through)."""

    def __init__(self, depth, dequantize_tensor, launch_batch=None):
        self.depth = max(1, min(int(depth), _native.BATCH_MAX))
        self._fn = dequantize_tensor
        self._launch_batch = launch_batch or self._native_batch        # (items, raw stream, device index) -> list of dense tensors
        self._entries = {}            # id(tensor) -> _Entry
        self._prev = None             # weak reference to the tensor of the previous call
        self._owner_thread = None
        self.hits = self.launches = self.batched = self.stale = self.bypassed = 0

    # ---- bookkeeping
    def _entry(self, tensor, create=True):
        key = id(tensor)
        e = self._entries.get(key)
        if e is not None and e.ref() is tensor:
            return e
        if not create:
            return None
        e = _Entry(weakref.ref(tensor, lambda _r, k=key: self._died(k, _r)))
        self._entries[key] = e
        return e

    def _died(self, key, ref):
        e = self._entries.get(key)
        if e is not None and e.ref is ref:
            del self._entries[key]           # its pending result (if any) goes back to the allocator with it

    def clear(self):
        self._entries.clear()
        self._prev = None

    def scratch_bytes(self):
        """Bytes of dense weights currently held ahead of their use (device memory the reference's VRAM estimate does not see)."""
        return sum(e.pending[3].numel() * e.pending[3].element_size() for e in list(self._entries.values()) if e.pending is not None)

    def stats(self):
        return {"depth": self.depth, "hits": self.hits, "launches": self.launches, "tensors_in_batches": self.batched, "stale_dropped": self.stale,
                "bypassed": self.bypassed, "tracked": len(self._entries), "held_bytes": self.scratch_bytes()}

    # ---- what qualifies
    @staticmethod
    def _packed_ok(t):
        return t.is_cuda and t.dtype is torch.uint8 and t.is_contiguous() and not (t.data_ptr() & 15)

    def _mode_of(self, tensor, dtype, dequant_dtype):
        """(dtype, compute dtype, device index) if this call can be served / predicted here, else None."""
        if dtype not in _dq._OUT_CODE or not isinstance(tensor, torch.Tensor):
            return None
        qtype = getattr(tensor, "tensor_type", None)
        try:
            if qtype not in _dq._HIP_TABLE:
                return None
        except TypeError:
            return None
        compute = dtype if dequant_dtype == "target" else dequant_dtype
        if compute not in _dq._COMPUTE_CODE or getattr(tensor, "tensor_shape", None) is None:
            return None
        with _dq._NoTorchFunction():
            if not self._packed_ok(tensor):
                return None
            index = tensor.device.index
        if not (_dq._DEVICE_OK.get(index) or _dq._device_served(index)) or _dq._cur_device() != index:
            return None
        return (dtype, compute, index)

    # ---- the launch
    @staticmethod
    def _native_batch(items, stream, index):
        """items: [(tensor, (dtype, compute, index))]; returns the dense results (fresh tensors), ONE ggq_dequant_batch call."""
        n = len(items)
        descs = (_native.ggq_desc * n)()
        outs = []
        with _dq._NoTorchFunction():
            for i, (t, (dtype, compute, _)) in enumerate(items):
                qid, block_size, type_size = _dq._HIP_TABLE[t.tensor_type]
                n_blocks = t.numel() // type_size
                out = torch.empty(tuple(t.tensor_shape), dtype=dtype, device=t.device)
                if out.numel() != n_blocks * block_size:
                    raise RuntimeError(f"shape '{list(t.tensor_shape)}' is invalid for input of size {n_blocks * block_size}")
                descs[i] = _native.ggq_desc(qid, _dq._OUT_CODE[dtype], t.data_ptr(), out.data_ptr(), n_blocks, _dq._COMPUTE_CODE[compute], 0)
                outs.append(out)
        _native.check(_native.lib().ggq_dequant_batch(descs, n, stream), "ggq_dequant_batch")
        return outs

    # ---- the call
    def __call__(self, tensor, dtype=None, dequant_dtype=None):
        mode = None
        tid = threading.get_ident()
        if self._owner_thread is None:
            self._owner_thread = tid
        if tid == self._owner_thread and not _dq._is_compiling():
            mode = self._mode_of(tensor, dtype, dequant_dtype)
            if mode is not None and torch.cuda.is_current_stream_capturing():
                mode = None
        if mode is None:
            # not served here (an F32 bias, a CPU tensor, tracing ...): the plain path, and NO effect on the learnt order -- the reference's
            # cast_bias_weight asks for the bias right before the weight (ops.py:205-207), which must not cut the chain of weights in two
            self.bypassed += 1
            return self._fn(tensor, dtype, dequant_dtype)
        stream = _dq._raw_stream(mode[2])
        e = self._entry(tensor)
        pend, e.pending = e.pending, None
        out = None
        if pend is not None:
            if pend[0] == tensor._version and pend[1] == mode and pend[2] == stream:
                out = pend[3]
                self.hits += 1
            else:
                self.stale += 1
        # learn the order
        prev = self._prev() if self._prev is not None else None
        if prev is not None and prev is not tensor:
            pe = self._entry(prev, create=False)
            if pe is not None:
                pe.nxt = e.ref
        self._prev = e.ref
        e.mode = mode
        if out is not None:
            return out
        # nothing ready: this tensor and the ones expected next, in one launch
        items, seen, cur = [(tensor, mode)], {id(tensor)}, e
        while len(items) < self.depth and cur.nxt is not None:
            nt = cur.nxt()
            if nt is None or id(nt) in seen:
                break
            ne = self._entry(nt, create=False)
            if ne is None or ne.mode is None or ne.mode[2] != mode[2] or ne.pending is not None:
                break
            with _dq._NoTorchFunction():
                if not self._packed_ok(nt):
                    break
            items.append((nt, ne.mode))
            seen.add(id(nt))
            cur = ne
        if len(items) == 1:
            self.launches += 1
            return self._fn(tensor, dtype, dequant_dtype)
        outs = self._launch_batch(items, stream, mode[2])
        self.launches += 1
        self.batched += len(items)
        for (t, m), o in zip(items[1:], outs[1:]):
            self._entry(t).pending = (t._version, m, stream, o)
        return outs[0]
