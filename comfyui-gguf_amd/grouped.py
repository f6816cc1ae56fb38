"""Many tensors, one call: a launch plan over the weight set of a model (include/ggq.h
``ggq_plan_*``).  The reference dequantizes one layer per ``dequantize_tensor`` call
(ops.py:177); for streaming a whole GGUF weight set through the GPU (BASELINE.json configs 3-4)
and for roofline measurement over a working set far beyond the 256 MiB Infinity Cache, the
tensors are described once -- device pointers, block counts -- and each ``launch()`` enqueues one
kernel per (quant type, output dtype) present, with no per-launch host->device traffic.

``ShardedPlan`` is the same thing over SEVERAL GPUs from ONE process (ComfyUI is single-process): the tensor list is partitioned
with ``sharding.partition`` -- the very split the one-process-per-GPU runs use -- one ``DequantPlan`` per device, every launch
enqueued by the calling thread (launches are asynchronous: at ~1 ms of GPU work per launch one thread keeps eight devices fed).
No collective, no peer access, no cross-device ordering: every tensor is independent.
"""
import ctypes

import torch

from . import _native
from .dequant import _OUT_CODE, _as_bytes, _check_compute, GGQUnsupported
from .qtypes import GGML_QUANT_SIZES, GGMLQuantizationType, HIP_QTYPES


class DequantPlan:
    """``items``: iterable of (packed_bytes_tensor, qtype, logical_shape[, out_dtype[, dequant_dtype]]).
    ``dequant_dtype`` is the reference's arithmetic dtype (None = fp16, or bfloat16 / float32).

    All tensors must live on one GPU.  Outputs are allocated here (``self.outputs``, same order,
    logical shapes) unless ``outputs`` supplies pre-allocated dense tensors.
    """

    def __init__(self, items, out_dtype=torch.float16, outputs=None, dequant_dtype=None):
        items = [tuple(it) for it in items]
        if not items:
            raise ValueError("empty plan")
        self._keep = []                      # packed buffers the plan points into
        self.outputs = []
        descs = (_native.ggq_desc * len(items))()
        device = None
        for i, it in enumerate(items):
            data, qtype, shape = it[:3]
            odt = it[3] if len(it) > 3 else out_dtype
            cdt = it[4] if len(it) > 4 else dequant_dtype
            cdt = odt if cdt == "target" else cdt
            key = GGMLQuantizationType(int(qtype))
            if key not in HIP_QTYPES:
                raise GGQUnsupported(f"no HIP unpacker for {key.name}")
            if not data.is_cuda:
                raise GGQUnsupported("plan tensors must be GPU-resident")
            device = device or data.device
            if data.device != device:
                raise ValueError("all tensors of a plan must live on one device")
            bs, ts = GGML_QUANT_SIZES[key]
            data = _as_bytes(data)
            n_blocks = data.numel() // ts
            n_el = 1
            for s in shape:
                n_el *= int(s)
            if n_el != n_blocks * bs:
                raise ValueError(f"shape {tuple(shape)} has {n_el} elements, packed data holds {n_blocks * bs}")
            out = outputs[i] if outputs is not None else torch.empty(tuple(shape), dtype=odt, device=device)
            if out.dtype != odt or out.numel() != n_el or not out.is_contiguous() or out.device != device:
                raise ValueError("pre-allocated output does not match (dtype, numel, contiguity, device)")
            self._keep.append(data)
            self.outputs.append(out)
            descs[i] = _native.ggq_desc(int(key), _OUT_CODE[odt], data.data_ptr(), out.data_ptr(), n_blocks, _check_compute(cdt), 0)
        self.device = device
        self._plan = ctypes.c_void_p()
        with torch.cuda.device(device):
            _native.check(_native.lib().ggq_plan_create(descs, len(items), ctypes.byref(self._plan)), "ggq_plan_create")
        self.bytes = int(_native.lib().ggq_plan_bytes(self._plan))
        self.kernels = int(_native.lib().ggq_plan_kernels(self._plan))

    def launch(self, stream=None):
        """Enqueue on ``stream`` (default: torch's current stream for the plan's device)."""
        if self._plan is None:
            raise RuntimeError("plan was closed")
        if torch.cuda.current_device() != self.device.index:
            with torch.cuda.device(self.device):
                return self.launch(stream)
        s = (stream or torch.cuda.current_stream(self.device)).cuda_stream
        _native.check(_native.lib().ggq_plan_launch(self._plan, s), "ggq_plan_launch")
        return self.outputs

    def close(self):
        if getattr(self, "_plan", None):
            _native.lib().ggq_plan_destroy(self._plan)
            self._plan = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _explicit(device):
    """torch.device with an explicit index: ``torch.device("cuda") != torch.device("cuda:0")``, and tensors always carry the index."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None and torch.cuda.is_available():
        device = torch.device("cuda", torch.cuda.current_device())
    return device


class ShardedPlan:
    """One ``DequantPlan`` per shard, shards on several devices (or, for tests on a one-GPU box, several shards on one device).

    ``shards``: list of (device, items) -- items as for ``DequantPlan``, already resident on that device.  Use
    :meth:`place` to get there from a flat tensor list.  ``own_streams=True`` gives every shard a stream of its own (ordered after
    its device's current stream at every launch); the default enqueues on each device's CURRENT stream, like ``DequantPlan``.
    """

    def __init__(self, shards, out_dtype=torch.float16, dequant_dtype=None, own_streams=False, indices=None):
        shards = [(_explicit(d), list(items)) for d, items in shards]
        if not shards or any(not items for _, items in shards):
            raise ValueError("every shard needs at least one tensor")
        self.devices = [d for d, _ in shards]
        self.plans = [DequantPlan(items, out_dtype=out_dtype, dequant_dtype=dequant_dtype) for _, items in shards]
        for d, p in zip(self.devices, self.plans):
            if p.device != d:
                raise ValueError(f"shard declared on {d} holds tensors on {p.device}")
        self.streams = [torch.cuda.Stream(d) for d in self.devices] if own_streams else None
        self.indices = indices                      # indices[s][k] = position of shard s's k-th tensor in the flat list (place())
        self.bytes = sum(p.bytes for p in self.plans)
        self.kernels = sum(p.kernels for p in self.plans)

    @staticmethod
    def assignment(entries, devices):
        """[(device, [indices into entries])] -- ``sharding.partition`` over ``len(devices)`` shards, shard r on ``devices[r]``; shards
        that got nothing (fewer tensors than devices) drop out.  ``entries``: (anything, qtype, shape) triples.  Pure: no GPU needed."""
        from .sharding import partition
        devices = [_explicit(d) for d in devices]
        if not devices:
            raise ValueError("no devices")
        parts = partition([(None, e[1], e[2]) for e in entries], len(devices))
        return [(d, p) for d, p in zip(devices, parts) if p]

    @classmethod
    def place(cls, items, devices, out_dtype=torch.float16, dequant_dtype=None, own_streams=None):
        """Partition a flat list of (packed, qtype, shape) over ``devices`` (sharding.partition: LPT on read+write bytes) and move
        every tensor's packed bytes to its device (no copy when it is already there).  ``devices`` may name one device several times."""
        items = [tuple(it) for it in items]
        used = cls.assignment(items, devices)
        shards = [(d, [(_as_bytes(items[i][0]).to(d, non_blocking=True),) + items[i][1:] for i in p]) for d, p in used]
        if own_streams is None:
            own_streams = len({d for d, _ in used}) < len(used)              # several shards on one device: only separate streams overlap them
        return cls(shards, out_dtype=out_dtype, dequant_dtype=dequant_dtype, own_streams=own_streams, indices=[p for _, p in used])

    @property
    def outputs(self):
        """Dense tensors per shard; see :meth:`outputs_in_order` for the flat list's order."""
        return [p.outputs for p in self.plans]

    def outputs_in_order(self):
        """The dense tensors in the order of the flat list given to :meth:`place` (each on its shard's device)."""
        if self.indices is None:
            raise ValueError("only a plan built by place() knows the flat order")
        flat = [None] * sum(len(ix) for ix in self.indices)
        for ix, p in zip(self.indices, self.plans):
            for i, o in zip(ix, p.outputs):
                flat[i] = o
        return flat

    def launch(self, join=True):
        """Enqueue every shard's kernels from the calling thread; returns at once (the host waits for nothing).  With ``own_streams`` every
        shard runs on its own stream -- ordered after its device's current stream, and (``join=True``) that current stream is then made to
        wait for the shard (a GPU-side dependency): the outputs may be consumed on the current stream as after ``DequantPlan.launch``, while
        shards that share a device still overlap each other.  The join also means that step k + 1 of EVERY shard of a device waits for step k
        of all of them; ``join=False`` leaves the shard streams free-running (back-to-back launches of a benchmark; the caller then orders
        consumers itself: ``synchronize()``, or ``current_stream.wait_stream(plan.streams[k])``) and skips the wait on the current stream."""
        for k, p in enumerate(self.plans):
            if self.streams is None:
                p.launch()
            else:
                s = self.streams[k]
                if join:
                    s.wait_stream(torch.cuda.current_stream(p.device))
                p.launch(s)
        if self.streams is not None and join:
            for s, p in zip(self.streams, self.plans):
                torch.cuda.current_stream(p.device).wait_stream(s)
        return self.outputs

    def synchronize(self):
        for k, p in enumerate(self.plans):
            (self.streams[k] if self.streams is not None else torch.cuda.current_stream(p.device)).synchronize()

    def close(self):
        for p in self.plans:
            p.close()
