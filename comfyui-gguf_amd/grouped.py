"""Many tensors, one call: a launch plan over the weight set of a model (include/ggq.h
``ggq_plan_*``).  The reference dequantizes one layer per ``dequantize_tensor`` call
(ops.py:177); for streaming a whole GGUF weight set through the GPU (BASELINE.json configs 3-4)
and for roofline measurement over a working set far beyond the 256 MiB Infinity Cache, the
tensors are described once -- device pointers, block counts -- and each ``launch()`` enqueues one
kernel per (quant type, output dtype) present, with no per-launch host->device traffic.
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
