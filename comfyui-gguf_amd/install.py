"""Drop the HIP path in behind an unmodified ComfyUI-GGUF checkout.

    import ComfyUI_GGUF.dequant as ref_dequant, ComfyUI_GGUF.ops as ref_ops
    install(ref_dequant, ref_ops)

replaces ``dequantize`` and ``dequantize_tensor`` (reference dequant.py:15,30) -- and the name
``ops.py`` imported at load time (reference ops.py:9) -- by wrappers that send GPU-resident
requests (any dequant_dtype the nodes offer) to the HIP kernels and hand EVERYTHING ELSE to the
reference's own original functions: CPU tensors at load time (loader.py:124,253,...), qtypes
without a kernel.  With ``exact=True`` (``GGQ_EXACT=1``) nothing above ``dequantize_tensor`` changes:
``GGMLTensor``, ``GGMLOps``, the loader and the nodes keep running the reference's code, so the
"Unet Loader (GGUF)" node works unchanged and every output is bit-equal to the reference's own on the same GPU.
The DEFAULT (``ref_ops`` given, no ``exact``) additionally turns on ``fast`` = ``fused_small_m`` + ``fused_mfma`` +
``gather_embedding`` below (``DEFAULT_FAST``: measured no further from an fp64 product than F.linear, profiles/r05_fused_error.json).
``uninstall()`` restores the originals.

``dense_cache_gb`` (or the environment variable ``GGQ_DENSE_CACHE_GB``) additionally keeps dequantized weights resident
in HBM up to that budget (resident.py: opt-in, off by default -- it trades VRAM for the per-step dequant work).

``fused_small_m`` (or ``GGQ_FUSED_SMALL_M=1``; needs ``ref_ops``) wraps ``GGMLOps.Linear.forward_ggml_cast_weights``
(reference ops.py:242-244): inputs of one to four rows (FLUX's modulation layers) go through the fused dequantize +
linear kernel (fused.py) when weight and input qualify, everything else -- LoRA-patched weights included -- through the
reference's method.  The result matches F.linear up to fp32 summation order, not bit for bit (same weights bit for bit): part of the default since round 5, off under ``exact``.

``fused_mfma`` (or ``GGQ_FUSED_MFMA=1``; needs ``ref_ops``) wraps the same method for inputs of up to ``fused_mfma_max_m`` rows
(default 256, ``GGQ_FUSED_MFMA_MAX_M``): fused dequantize + GEMM on the matrix cores (fused.linear_mfma), 1.2-3x faster than
dequantize + hipBLASLt in that range on FLUX / T5 layer shapes; above it -- and above 128 rows on the largest weights (fused.AUTO_MAX_MACS) -- hipBLASLt on
the dense weight wins and keeps the job.  Same numerics statement; part of the default, off under ``exact``.

``gather_embedding`` (or ``GGQ_GATHER_EMBEDDING=1``; needs ``ref_ops``) wraps ``GGMLOps.Embedding.forward_ggml_cast_weights``
(reference ops.py:251-260): instead of dequantizing the whole table and then gathering, only the rows the token ids name are
unpacked (dequant.dequantize_rows) -- bit-identical values, no transient dense table (a 152 k x 3584 vocabulary is 1.1 GB).
Tables the kernel does not take (CPU, F16 / F32 storage, ``max_norm`` set, LoRA patches) keep the
reference's method.  Ids outside the table fail as F.embedding's do (asynchronous device-side assert; ``GGQ_CHECK_INDICES=0``: the kernel's clamp).  Part of the default.

``cpu_route_mb`` (or ``GGQ_CPU_ROUTE_MB=N``): CPU-resident quantized tensors of at least N MB of packed bytes -- the load-time
callers, token_embd / mmproj tables (reference loader.py:253-254,270,386,397) -- are uploaded, unpacked on the GPU and copied back
(dequant.dequantize_tensor_via_gpu) instead of running the reference's torch-CPU ops: same bits, CPU result.  Off by default (it
touches the GPU at load time, before ComfyUI's model management has placed anything).

``native_reader`` (or ``GGQ_NATIVE_READER=1``; needs ``ref_loader``): the GGUF container behind "Unet Loader (GGUF)" is parsed by the native reader
(include/ggq_gguf.h) instead of gguf-py's ``GGUFReader`` -- the reference's ``gguf_sd_loader`` runs unchanged over an adapter that yields what
loader.py:55-106 reads (gguf_adapter.py: CPU mmap views, same ownership and semantics).  Opt-in.

``fast`` (or ``GGQ_FAST=1``): ``fused_small_m`` + ``fused_mfma`` + ``gather_embedding`` in one switch -- the options that hold no VRAM and measured faster
wherever they apply.  The default when ``ref_ops`` is given; ``exact`` (``GGQ_EXACT=1``) is its opposite.

``overlap`` (or ``GGQ_OVERLAP=1``; needs ``ref_ops``) wraps ``GGMLLayer.cast_bias_weight`` (reference ops.py:194-211) with
overlap.LayerPrefetcher: for CPU-resident packed weights (low-VRAM mode, ops.py:209) the NEXT layer's bytes are copied host->device
and unpacked on a side stream while the current layer's GEMM runs (weights already in HBM are left alone: prefetching those measured slower in
every run on record, EXPERIMENTS.md Part R4).  Bit-identical values; opt-in because the dense weight handed out is a view into a scratch buffer that is
reused two layers later (see overlap.py).

VRAM accounting: whenever an option that HOLDS device memory is on (``dense_cache_gb``, ``overlap``), ``install()`` also wraps
``GGMLLayer.ggml_save_to_state_dict`` (reference ops.py:145-160) -- the fake state dict ComfyUI's model management sizes a GGUF model
by -- so that the layer marked ``largest_layer`` reports one more meta tensor, ``temp.ggq_scratch``, of the options' worst case
(``scratch_reservation``): nobody has to edit the reference for ComfyUI to know.  A default install adds nothing.
"""
import os

import torch

from . import dequant as _hip

_installed = {}

# The default of install() / the drop-in when neither ``fast`` / ``exact`` nor one of the three options' own switches says otherwise.
# True since round 5: the fused kernels use the SAME weights bit for bit and differ from ``F.linear`` only in the order of an fp32
# summation that is hipBLASLt's implementation detail, not a contract of the reference (ops.py:242-244 calls F.linear and takes what
# the BLAS library does).  Measured against an fp64 evaluation on the oracle's weights on every FLUX.1-dev / SD3.5-large / T5-xxl linear
# shape, 1 / 4 / 64 / 256 rows, bf16 and fp16 (tools/fused_error.py -> profiles/r05_fused_error.json): the fused results are no further
# from the exact product than F.linear's (RMS within 2 %, max within one output rounding step) and reproduce run to run; the default
# install otherwise costs 20-48 % of a FLUX step at <= 1024 tokens.  ``exact=True`` / ``GGQ_EXACT=1`` keeps
# every linear on unpack + F.linear (bit-equal to the reference's own output on the same GPU).
DEFAULT_FAST = True


def _env_flag(name):
    return os.environ.get(name, "0") not in ("", "0")


def install(ref_dequant, ref_ops=None, ref_loader=None, dense_cache_gb=None, fused_small_m=None, gather_embedding=None, overlap=None,
            fused_mfma=None, fused_mfma_max_m=None, cpu_route_mb=None, fast=None, exact=None, native_reader=None):
    """Patch the reference modules in place; returns the dict of original functions.

    ``fast`` (or ``GGQ_FAST=1``; needs ``ref_ops``): ``fused_small_m`` + ``fused_mfma`` (same weights, results as close to the exact product as
    F.linear's, in another fp32 summation order) + ``gather_embedding`` (bit-identical) -- the options that hold no VRAM and measured faster
    wherever they apply.  It is the DEFAULT when ``ref_ops`` is given (``DEFAULT_FAST``); ``exact=True`` (or ``GGQ_EXACT=1``) turns it off: every
    linear then runs unpack + F.linear, bit-equal to the reference on the same GPU.  An option given explicitly (argument or its own
    environment variable) wins over both switches."""
    if id(ref_dequant) in _installed:
        return _installed[id(ref_dequant)]["orig"]
    if exact is None:
        exact = _env_flag("GGQ_EXACT")
    asked_fast = fast if fast is not None else (_env_flag("GGQ_FAST") if "GGQ_FAST" in os.environ else None)
    if asked_fast and exact:
        raise ValueError("fast and exact (GGQ_FAST / GGQ_EXACT) contradict each other")
    if asked_fast and ref_ops is None:
        raise ValueError("fast patches GGMLOps.Linear / GGMLOps.Embedding: pass ref_ops")
    fast = bool(asked_fast) if asked_fast is not None else (DEFAULT_FAST and not exact and ref_ops is not None)
    if fast:
        fused_small_m = True if fused_small_m is None and "GGQ_FUSED_SMALL_M" not in os.environ else fused_small_m
        fused_mfma = True if fused_mfma is None and "GGQ_FUSED_MFMA" not in os.environ else fused_mfma
        gather_embedding = True if gather_embedding is None and "GGQ_GATHER_EMBEDDING" not in os.environ else gather_embedding
    from . import _native
    _native.lib()                                   # fail now, loudly, if the extension is absent
    orig = {"dequantize": ref_dequant.dequantize, "dequantize_tensor": ref_dequant.dequantize_tensor}
    unsupported = _hip.GGQUnsupported
    hip_dequantize, hip_dequantize_tensor = _hip.dequantize, _hip.dequantize_tensor
    if dense_cache_gb is None and os.environ.get("GGQ_DENSE_CACHE_GB"):
        dense_cache_gb = float(os.environ["GGQ_DENSE_CACHE_GB"])
    cache = None
    if dense_cache_gb:
        from .resident import DenseCache
        cache = DenseCache(dense_cache_gb * 1e9, hip_dequantize_tensor)
        hip_dequantize_tensor = cache               # GGQUnsupported from the wrapped function passes straight through
    orig_dequantize, orig_dequantize_tensor = orig["dequantize"], orig["dequantize_tensor"]

    # Try the HIP path first; a request it does not serve (CPU-resident bytes at load time, a qtype
    # without a kernel, an exotic dequant_dtype) raises GGQUnsupported BEFORE anything is launched and
    # is handed to the reference's own function.  No eligibility pre-checks: on the per-layer hot loop
    # every attribute probe of a Tensor subclass costs about as much as the kernel launch itself.
    def dequantize(data, qtype, oshape, dtype=None):
        try:
            return hip_dequantize(data, qtype, oshape, dtype=dtype)
        except unsupported:
            return orig_dequantize(data, qtype, oshape, dtype=dtype)

    if cpu_route_mb is None and os.environ.get("GGQ_CPU_ROUTE_MB"):
        cpu_route_mb = float(os.environ["GGQ_CPU_ROUTE_MB"])
    route_bytes = int(cpu_route_mb * 1e6) if cpu_route_mb else 0
    via_gpu = _hip.dequantize_tensor_via_gpu

    def dequantize_tensor(tensor, dtype=None, dequant_dtype=None):
        try:
            return hip_dequantize_tensor(tensor, dtype, dequant_dtype)
        except unsupported:
            if route_bytes and isinstance(tensor, torch.Tensor) and tensor.device.type == "cpu" and tensor.numel() * tensor.element_size() >= route_bytes:
                try:
                    return via_gpu(tensor, dtype, dequant_dtype)
                except unsupported:
                    pass
            return orig_dequantize_tensor(tensor, dtype, dequant_dtype)

    dequantize.__wrapped__ = orig["dequantize"]
    dequantize_tensor.__wrapped__ = orig["dequantize_tensor"]
    if fused_small_m is None:
        fused_small_m = _env_flag("GGQ_FUSED_SMALL_M")
    if fused_mfma is None:
        fused_mfma = _env_flag("GGQ_FUSED_MFMA")
    if fused_mfma_max_m is None:
        fused_mfma_max_m = int(os.environ.get("GGQ_FUSED_MFMA_MAX_M", "256"))
    if gather_embedding is None:
        gather_embedding = _env_flag("GGQ_GATHER_EMBEDDING")
    if overlap is None:
        overlap = _env_flag("GGQ_OVERLAP")
    if native_reader is None:
        native_reader = _env_flag("GGQ_NATIVE_READER")
    if native_reader and (ref_loader is None or not hasattr(ref_loader, "gguf")):
        raise ValueError("native_reader rebinds the name `gguf` inside the reference's loader module (loader.py:5,55): pass ref_loader")
    # Everything an option needs is looked up BEFORE the first attribute is replaced, and the replacing itself is undone if any step fails: a
    # ComfyUI-GGUF checkout that lacks one of the classes must not be left half-patched with `_installed` unwritten (uninstall() could then restore
    # nothing).  Options that only came from the DEFAULT (nobody asked for them by name) are skipped when their class is absent.
    ggml_ops = getattr(ref_ops, "GGMLOps", None) if ref_ops is not None else None
    explicit = asked_fast is not None                 # `fast` given by argument or GGQ_FAST: its parts are then requests, not defaults
    if fused_small_m or fused_mfma:
        if ref_ops is None:
            raise ValueError("fused_small_m / fused_mfma patch GGMLOps.Linear: pass ref_ops")
        if not hasattr(getattr(ggml_ops, "Linear", None), "forward_ggml_cast_weights"):
            if explicit or not fast:
                raise AttributeError("this ComfyUI-GGUF checkout has no GGMLOps.Linear.forward_ggml_cast_weights to wrap (fused_small_m / fused_mfma)")
            fused_small_m = fused_mfma = False
    if gather_embedding:
        if ref_ops is None:
            raise ValueError("gather_embedding patches GGMLOps.Embedding: pass ref_ops")
        if not hasattr(getattr(ggml_ops, "Embedding", None), "forward_ggml_cast_weights"):
            if explicit or not fast:
                raise AttributeError("this ComfyUI-GGUF checkout has no GGMLOps.Embedding.forward_ggml_cast_weights to wrap (gather_embedding)")
            gather_embedding = False
    if overlap and ref_ops is None:
        raise ValueError("overlap patches GGMLLayer.cast_bias_weight: pass ref_ops")
    patched = []
    prefetcher = None

    def put(owner, name, new):
        patched.append((owner, name, getattr(owner, name)))
        setattr(owner, name, new)

    try:
        put(ref_dequant, "dequantize", dequantize)
        put(ref_dequant, "dequantize_tensor", dequantize_tensor)
        for mod in (ref_ops, ref_loader):               # `from .dequant import dequantize_tensor` bound the old object
            if mod is not None and getattr(mod, "dequantize_tensor", None) is orig["dequantize_tensor"]:
                put(mod, "dequantize_tensor", dequantize_tensor)
        if fused_small_m or fused_mfma:
            patched.append(_fuse_linear(ggml_ops.Linear, unsupported, bool(fused_small_m), fused_mfma_max_m if fused_mfma else 0))
        if gather_embedding:
            patched.append(_gather_embedding(ggml_ops.Embedding, unsupported))
        if overlap:
            from .overlap import attach
            record, prefetcher = attach(ref_ops.GGMLLayer)
            patched.append(record)
        if native_reader:
            # "Unet Loader (GGUF)" -> loader.gguf_sd_loader -> gguf.GGUFReader(path) (loader.py:55): the container is parsed by the C++ reader of
            # include/ggq_gguf.h, handed to the reference's UNCHANGED loader in gguf-py's attribute layout (CPU mmap views, same ownership)
            from .gguf_adapter import GGUFModuleProxy
            put(ref_loader, "gguf", GGUFModuleProxy(ref_loader.gguf))
    except BaseException:
        for owner, name, fn in reversed(patched):
            setattr(owner, name, fn)
        if prefetcher is not None:
            prefetcher.close()
        raise
    options = {"dense_cache_gb": dense_cache_gb or None, "fused_small_m": bool(fused_small_m) or None, "fused_mfma": (fused_mfma_max_m if fused_mfma else None),
               "gather_embedding": bool(gather_embedding) or None, "overlap": bool(overlap) or None, "cpu_route_mb": cpu_route_mb or None,
               "exact": bool(exact) or None, "native_reader": bool(native_reader) or None}
    rec = {"orig": orig, "patched": patched, "cache": cache, "prefetcher": prefetcher,
           "options": {k: v for k, v in options.items() if v is not None}}
    if (dense_cache_gb or overlap) and ref_ops is not None and hasattr(getattr(ref_ops, "GGMLLayer", None), "ggml_save_to_state_dict"):
        patched.append(_account_scratch(ref_ops.GGMLLayer, rec))
    _installed[id(ref_dequant)] = rec
    return orig


def describe(ref_dequant):
    """'; options: ...' of an installation, for the one log line the drop-in prints (autoinstall.py)."""
    rec = _installed.get(id(ref_dequant)) or {}
    names = sorted({f"{getattr(o, '__name__', o)}.{n}" for o, n, _ in rec.get("patched", [])} - {"dequantize", "dequantize_tensor"})
    opts = ", ".join(f"{k}={v}" for k, v in rec.get("options", {}).items()) or "bit-exact unpack + F.linear"
    if rec.get("options", {}).get("fused_small_m") or rec.get("options", {}).get("fused_mfma"):
        opts += " (fused linears: same weights bit for bit, fp32 summation in the kernel's order; GGQ_EXACT=1 keeps unpack + F.linear everywhere)"
    return f"; patched {', '.join(names)}; {opts}"


def scratch_bytes(ref_dequant):
    """Device memory the opt-ins of this installation hold RIGHT NOW that the reference's VRAM estimate knows nothing about: the resident
    dense weights (dense_cache_gb) and the side-stream scratch and staging slots (overlap), in bytes, by option.  What ComfyUI is TOLD is
    the worst case, ``scratch_reservation``, through the wrapped ``ggml_save_to_state_dict`` (``_account_scratch``)."""
    rec = _installed.get(id(ref_dequant)) or {}
    out = {"dense_cache": rec["cache"].scratch_bytes() if rec.get("cache") is not None else 0,
           "overlap": rec["prefetcher"].scratch_bytes() if rec.get("prefetcher") is not None else 0}
    out["total"] = sum(out.values())
    return out


OVERLAP_DENSE_SLOTS, OVERLAP_PACKED_SLOTS = 2, 3      # overlap.LayerPrefetcher: dense scratch slots and packed staging slots per device


def scratch_reservation(ref_dequant_or_rec, largest_dense_bytes, largest_packed_bytes):
    """WORST-CASE bytes the memory-holding options of an installation can come to hold, by option: ``dense_cache_gb`` its whole budget;
    ``overlap`` 2 dense scratch slots + 3 packed staging slots, each grown to the largest layer.  The fused kernels, ``gather_embedding``
    and ``cpu_route_mb`` hold nothing beyond their call; a default install reserves 0."""
    rec = ref_dequant_or_rec if isinstance(ref_dequant_or_rec, dict) else (_installed.get(id(ref_dequant_or_rec)) or {})
    opts = rec.get("options", {})
    out = {"dense_cache": int(opts["dense_cache_gb"] * 1e9) if opts.get("dense_cache_gb") else 0,
           "overlap": OVERLAP_DENSE_SLOTS * int(largest_dense_bytes) + OVERLAP_PACKED_SLOTS * int(largest_packed_bytes) if opts.get("overlap") else 0}
    out["total"] = sum(out.values())
    return out


def _account_scratch(layer_cls, rec):
    """Wrap ``layer_cls.ggml_save_to_state_dict`` (reference ops.py:145-160): the fake state dict ComfyUI's model management sizes a GGUF
    model by.  The reference reports, for the layer marked ``largest_layer`` (loader.py:134-137), one ``temp.weight`` of the dense size --
    "space required for dequantizing the largest tensor" (ops.py:153-158); that is exactly what the default path needs.  Options that hold
    more add ONE more meta tensor in that same branch, ``temp.ggq_scratch``, uint8 of ``scratch_reservation(...)["total"]`` bytes, so that
    ComfyUI keeps that much VRAM free when it decides what to load ("a replacement that needs scratch beyond
    numel x 2 must account for it here").  Returns the (owner, name, original) record uninstall() restores."""
    reference_save = layer_cls.ggml_save_to_state_dict

    def ggml_save_to_state_dict(self, destination, prefix, keep_vars):
        out = reference_save(self, destination, prefix, keep_vars)
        if getattr(self, "largest_layer", False):
            weight = self.weight
            shape = getattr(weight, "tensor_shape", weight.shape)
            temp = destination.get(prefix + "temp.weight")
            elem = temp.element_size() if temp is not None else 2        # the dtype rule of ops.py:156
            dense = elem
            for d in shape:
                dense *= int(d)
            with torch._C.DisableTorchFunctionSubclass():
                packed = weight.numel() * weight.element_size()           # the bytes as stored (GGMLTensor.shape is the LOGICAL shape)
            extra = scratch_reservation(rec, dense, packed)["total"]
            if extra:
                destination[prefix + "temp.ggq_scratch"] = torch.empty(extra, device=torch.device("meta"), dtype=torch.uint8)
        return out

    ggml_save_to_state_dict.__wrapped__ = reference_save
    layer_cls.ggml_save_to_state_dict = ggml_save_to_state_dict
    return (layer_cls, "ggml_save_to_state_dict", reference_save)


def _gather_embedding(embedding_cls, unsupported):
    """Wrap ``embedding_cls.forward_ggml_cast_weights``; returns the (owner, name, original) record uninstall() restores."""
    from .dequant import _check_indices_default, dequantize_rows, dequantize_rows_traced
    reference_forward = embedding_cls.forward_ggml_cast_weights
    check = _check_indices_default(True)           # the reference's F.embedding fails on an id outside the table: so does the drop-in (GGQ_CHECK_INDICES=0: clamp)
    is_compiling = _hip._is_compiling

    def forward_ggml_cast_weights(self, input, out_dtype=None):
        weight = self.weight
        if ((input.is_cuda or (is_compiling() and _hip._TRACE_ANY_DEVICE)) and weight is not None and getattr(self, "max_norm", None) is None
                and not getattr(weight, "patches", None)):
            # the table's dtype the reference's way: out_dtype, else what cast_bias_weight(self, ...) falls back to (ops.py:196-197)
            table_dtype = out_dtype if out_dtype is not None else getattr(self, "dtype", torch.float32)
            if is_compiling():                     # torch.compile: the same kernel as the custom op ggq::dequantize_rows (None: the reference's method)
                rows = dequantize_rows_traced(weight.to(input.device), input, table_dtype, self.dequant_dtype, check)
                return rows.to(dtype=out_dtype) if rows is not None else reference_forward(self, input, out_dtype)
            try:
                return dequantize_rows(weight.to(input.device), input, table_dtype, self.dequant_dtype, check_indices=check).to(dtype=out_dtype)
            except unsupported:
                pass
        return reference_forward(self, input, out_dtype)

    forward_ggml_cast_weights.__wrapped__ = reference_forward
    embedding_cls.forward_ggml_cast_weights = forward_ggml_cast_weights
    return (embedding_cls, "forward_ggml_cast_weights", reference_forward)


def _fuse_linear(linear_cls, unsupported, small_m, mfma_max_m):
    """Wrap ``linear_cls.forward_ggml_cast_weights``: inputs of 1..4 rows -> fused.linear_small (when ``small_m``), inputs of up
    to ``mfma_max_m`` rows -> fused.linear_mfma; everything else, and everything either kernel declines, -> the reference's method.
    Returns the (owner, name, original) record uninstall() restores."""
    from .fused import linear_auto, linear_traced
    reference_forward = linear_cls.forward_ggml_cast_weights
    is_compiling = _hip._is_compiling

    def forward_ggml_cast_weights(self, input):
        weight = self.weight
        if is_compiling():
            # torch.compile (the reference allows full compile on torch >= 2.8, ops.py:11-42): a ctypes call is nothing Dynamo can put in a graph, so the
            # SAME kernels go in as the custom ops ggq::linear_small / ggq::linear_mfma (fused.linear_traced); None = this call is the reference's
            y = linear_traced(self, input, small_m, mfma_max_m)
            return y if y is not None else reference_forward(self, input)
        if weight is not None and input.is_cuda:
            try:
                # a CPU-resident weight (low-VRAM mode) is copied to the GPU only once the kernel is known to take the request
                # (weight_to): a declined request costs no copy, the reference's method below makes its own (ops.py:209)
                return linear_auto(input, weight, self.bias, self.dequant_dtype, input.device, small_m, mfma_max_m)
            except unsupported:
                pass
        return reference_forward(self, input)

    forward_ggml_cast_weights.__wrapped__ = reference_forward
    linear_cls.forward_ggml_cast_weights = forward_ggml_cast_weights
    return (linear_cls, "forward_ggml_cast_weights", reference_forward)


def dense_cache(ref_dequant):
    """The DenseCache of an installation (None when the option is off): ``.stats()``, ``.clear()``."""
    rec = _installed.get(id(ref_dequant))
    return rec["cache"] if rec else None


def prefetcher(ref_dequant):
    """The LayerPrefetcher of an installation (None when ``overlap`` is off): ``.stats()``."""
    rec = _installed.get(id(ref_dequant))
    return rec.get("prefetcher") if rec else None


def uninstall(ref_dequant):
    rec = _installed.pop(id(ref_dequant), None)
    if rec:
        for mod, name, fn in rec["patched"]:
            setattr(mod, name, fn)
        if rec.get("cache") is not None:
            rec["cache"].clear()
        if rec.get("prefetcher") is not None:
            rec["prefetcher"].close()
