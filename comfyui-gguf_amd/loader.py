"""The input side of the dequant path: GGUF file -> state dict of ``GGMLTensor`` without the third-party
``gguf`` package, plus what the reference cannot do: stream the packed weights straight into HBM in one
pass (``device=...``) so the dequant kernels find them resident.

Interface mirrored from the reference (same names, argument meaning, results, error types and messages):
    get_orig_shape(reader, tensor_name)                     loader.py:16-24
    get_field(reader, field_name, field_type)               loader.py:26-37
    get_list_field(reader, field_name, field_type)          loader.py:39-49
    gguf_sd_loader(path, handle_prefix, return_arch, is_text_model)     loader.py:51-141
``reader`` is a :class:`gguf_file.GGUFFile` (native parser) instead of a ``gguf.GGUFReader``; a field is
``(types, value)`` with the value already decoded, where gguf-py hands out ``parts`` / ``data`` indices.

Not mirrored (the reference's control plane, SURVEY.md section 2 -- left to the reference's own code,
which keeps working on top of this module's state dicts): key remapping for text encoders, tokenizer
reconstruction, mmproj discovery (loader.py:143-406), and architecture *detection* for files that carry
no ``general.architecture`` (tools/convert.py ``detect_arch``) -- pass it in as ``detect_arch=`` if needed.
"""
import logging

import torch

from .dequant import dequantize_tensor, is_quantized
from .gguf_file import ARRAY, INT32, STRING, GGUFFile
from .ops import GGMLTensor
from .qtypes import GGML_QUANT_SIZES as _BYTES_KNOWN       # the types sharding.tensor_cost can price
from .qtypes import GGMLQuantizationType as Q

# architectures the loaders accept (loader.py:12-14)
IMG_ARCH_LIST = {"flux", "sd1", "sdxl", "sd3", "aura", "hidream", "cosmos", "ltxv", "hyvid", "wan", "lumina2", "qwen_image"}
TXT_ARCH_LIST = {"t5", "t5encoder", "llama", "qwen2vl", "qwen3", "qwen3vl"}
VIS_TYPE_LIST = {"clip-vision", "mmproj"}

_SCALAR_TYPES = (int, float, bool)
_PLAIN_DTYPES = {Q.F32: torch.float32, Q.F16: torch.float16}      # stored as-is: viewed, never dequantized


# ---- metadata accessors ------------------------------------------------------------------------------

def get_orig_shape(reader, tensor_name):
    """The converter records the torch shape of reshaped tensors under ``comfy.gguf.orig_shape.<name>``
    (tools/convert.py:293-295); None when the file has no such entry."""
    key = "comfy.gguf.orig_shape." + tensor_name
    field = reader.get_field(key)
    if field is None:
        return None
    if field.types != [ARRAY, INT32]:
        raise TypeError(f"Bad original shape metadata for {key}: Expected ARRAY of INT32, got {field.types}")
    return torch.Size(int(dim) for dim in field.value)


def _values(field):
    return field.value if isinstance(field.value, tuple) else (field.value,)


def get_field(reader, field_name, field_type):
    """One metadata value as ``field_type`` (str / int / float / bool); None if the key is absent.  Strings
    are type-checked (this accessor reads the architecture string); of an array the LAST element counts."""
    if field_type is not str and field_type not in _SCALAR_TYPES:
        if reader.get_field(field_name) is None:
            return None
        raise TypeError(f"Unknown field type {field_type}")
    field = reader.get_field(field_name)
    if field is None:
        return None
    if field_type is str:
        if field.types != [STRING]:
            raise TypeError(f"Bad type for GGUF {field_name} key: expected string, got {field.types!r}")
        return field.value
    return field_type(_values(field)[-1])


def get_list_field(reader, field_name, field_type):
    """Every element of an array-valued key as a tuple of ``field_type``; None if the key is absent."""
    field = reader.get_field(field_name)
    if field is None:
        return None
    if field_type is not str and field_type not in _SCALAR_TYPES:
        raise TypeError(f"Unknown field type {field_type}")
    return tuple(field_type(v) for v in _values(field))


# ---- the state-dict loader -----------------------------------------------------------------------------

def _select(reader, handle_prefix):
    """(state-dict key, tensor) pairs.  If ANY tensor name carries ``handle_prefix``, only those are kept and
    the prefix is stripped; otherwise every tensor is kept under its own name (loader.py:57-71)."""
    named = [(t.name, t) for t in reader.tensors]
    if handle_prefix is None or not any(name.startswith(handle_prefix) for name, _ in named):
        return named
    cut = len(handle_prefix)
    return [(name[cut:], t) for name, t in named if name.startswith(handle_prefix)]


def _architecture(reader, path, keys, is_text_model, detect_arch):
    """-> (arch, compat).  Files without architecture metadata (stable-diffusion.cpp exports; the "pig" / "cow"
    placeholders) are image models loaded in compatibility mode, which needs the reference's key-based detector;
    a known architecture must match the kind of model being loaded (loader.py:73-92)."""
    arch = get_field(reader, "general.architecture", str)
    kind = get_field(reader, "general.type", str)
    if arch is None or arch in ("pig", "cow"):
        if is_text_model:
            raise ValueError(f"This gguf file is incompatible with llama.cpp!\nConsider using safetensors or a compatible gguf file\n({path})")
        compat = arch or "sd.cpp"
        try:
            if detect_arch is None:
                raise NotImplementedError("no architecture metadata and no detect_arch callable supplied")
            arch = detect_arch(set(keys)).arch
        except Exception as e:
            raise ValueError(f"This model is not currently supported - ({e})")
        return arch, compat
    if is_text_model:
        if arch not in TXT_ARCH_LIST and kind not in VIS_TYPE_LIST:
            raise ValueError(f"Unexpected text model architecture type in GGUF file: {arch!r}")
    elif arch not in IMG_ARCH_LIST:
        raise ValueError(f"Unexpected architecture type in GGUF file: {arch!r}")
    return arch, None


def _logical_shape(reader, tensor, arch, compat):
    """torch shape of a tensor: the recorded original shape if any, else the ggml dims reversed (ggml lists the
    fastest dimension first).  stable-diffusion.cpp SDXL exports keep 1x1 conv projections 4-D: trailing unit
    dims beyond two are dropped there (loader.py:108-116)."""
    shape = get_orig_shape(reader, tensor.name)
    if shape is not None:
        return shape
    dims = [int(d) for d in reversed(tensor.shape)]
    if compat == "sd.cpp" and arch == "sdxl" and tensor.name.endswith((".proj_in.weight", ".proj_out.weight")):
        while len(dims) > 2 and dims[-1] == 1:
            dims.pop()
    return torch.Size(dims)


def gguf_sd_loader(path, handle_prefix="model.diffusion_model.", return_arch=False, is_text_model=False, *,
                   device=None, detect_arch=None, upload_threads=0, shard=None, devices=None):
    """Read a GGUF file as a state dict of ``GGMLTensor`` (loader.py:51-141).

    ``device=None`` (the reference's behaviour): every tensor is a read-only mmap view on the CPU.
    ``device="cuda:N"``: the file's tensor-data section is streamed into ONE HBM buffer and every tensor is a
    view into it -- packed weights resident, 16-byte aligned, ready for the HIP kernels (``GGUFFile.upload``;
    the views keep the arena alive).
    ``shard=(rank, world_size)`` (with ``device``): one process per GPU, no collectives -- only the tensors
    ``sharding.partition`` assigns to ``rank`` (every rank computes the same assignment from the file's tensor table alone)
    are uploaded and returned.
    ``devices=["cuda:0", "cuda:1", ...]``: ONE process, several GPUs (ComfyUI is single-process) -- the same partition, shard r
    uploaded to ``devices[r]``, ONE state dict whose tensors live on different devices (``state_dict_plan`` then returns a
    ``grouped.ShardedPlan``).  No peer access, no collective.
    """
    if devices is not None:
        if device is not None or shard is not None:
            raise ValueError("devices=[...] places the shards itself: do not pass device= or shard=")
        devices = [torch.device(d) for d in devices]
        if not devices:
            raise ValueError("devices=[...] is empty")
    with GGUFFile(path) as reader:                      # ONE parse, whatever the placement (round 4 re-opened the file once per device)
        selected = _select(reader, handle_prefix)
        arch, compat = _architecture(reader, path, [key for key, _ in selected], is_text_model, detect_arch)
        if compat:
            logging.warning(f"Warning: This gguf model file is loaded in compatibility mode '{compat}' [arch:{arch}]")

        # where every selected tensor's bytes live: {tensor name: (arena, byte offset)}; None = the reference's CPU mmap views
        placed = None
        if shard is not None or devices is not None:
            if shard is not None and device is None:
                raise ValueError("shard=(rank, world_size) selects what is uploaded: it needs device=")
            from .sharding import partition
            rank, world = shard if shard is not None else (None, len(devices))
            manifest = [(key, t.tensor_type, tuple(int(d) for d in reversed(t.shape)) or (1,)) for key, t in selected]
            costed = [m if m[1] in _BYTES_KNOWN else (m[0], Q.F16, m[2]) for m in manifest]       # unknown types: cost as 2 B/element
            parts = partition(costed, world)                                                       # computed ONCE; every rank / device agrees
            placed = {}
            for r, dev in ([(rank, device)] if shard is not None else list(enumerate(devices))):
                mine = [selected[i][1] for i in parts[r]]
                if mine:
                    arena, where = reader.upload_tensors(dev, mine, threads=upload_threads)
                    placed.update({name: (arena, off) for name, off in where.items()})
            if shard is not None:
                selected = [selected[i] for i in parts[rank]]                                      # a rank returns its share only, in file order
        elif device is not None:
            if reader.alignment % 16:
                # general.alignment 8 / 24 / ...: offsets in the file are not all 16-byte aligned -- re-pack so that every tensor is
                arena, where = reader.upload_tensors(device, [t for _, t in selected], threads=upload_threads)
                placed = {name: (arena, off) for name, off in where.items()}
            else:
                arena = reader.upload(device, threads=upload_threads)
                placed = {t.name: (arena, t.offset) for _, t in selected}
        state_dict, counts = {}, {}
        for key, tensor in selected:                   # the file's order (the reference loader's), also when the tensors span devices
            if placed is not None:
                arena, off = placed[tensor.name]
                raw = arena[off: off + tensor.nbytes]
            else:
                raw = tensor.data
            shape = _logical_shape(reader, tensor, arch, compat)
            plain = _PLAIN_DTYPES.get(tensor.tensor_type)
            if plain is not None:
                raw = raw.view(plain).view(*shape)
            entry = GGMLTensor(raw, tensor_type=tensor.tensor_type, tensor_shape=shape)
            if tensor.tensor_type == Q.BF16 and len(shape) <= 1:
                entry = dequantize_tensor(entry, dtype=torch.float32)      # norms / biases stored as BF16: plain fp32 from here on
            state_dict[key] = entry
            label = getattr(tensor.tensor_type, "name", repr(tensor.tensor_type))
            counts[label] = counts.get(label, 0) + 1
        logging.info("gguf qtypes: " + ", ".join(f"{name} ({n})" for name, n in counts.items()))

    # the VRAM estimate reserves room for dequantizing the largest quantized tensor (ops.py:134-136,151-156)
    quantized = [key for key, value in state_dict.items() if is_quantized(value)]
    if quantized:
        state_dict[max(quantized, key=lambda key: state_dict[key].numel())].is_largest_weight = True

    return (state_dict, arch) if return_arch else state_dict


def _plan_items(state_dict):
    from .dequant import hip_supported
    keys = [k for k, v in state_dict.items() if is_quantized(v) and hip_supported(v.tensor_type) and v.is_cuda]
    if not keys:
        raise ValueError("no GPU-resident quantized tensors with a HIP unpacker in this state dict")
    return keys, [(state_dict[k].as_subclass(torch.Tensor), state_dict[k].tensor_type, tuple(state_dict[k].tensor_shape)) for k in keys]


def state_dict_plan(state_dict, dtype=torch.float16, dequant_dtype=None):
    """One ``DequantPlan`` over every GPU-resident quantized tensor of a state dict that has a HIP
    unpacker: the whole weight set dequantized by one launch per (format, mode).  Returns
    (plan, keys) with ``plan.outputs[i]`` the dense tensor of ``keys[i]``.  A state dict loaded with ``devices=[...]`` spans several
    GPUs: that is ``state_dict_sharded_plan`` (another return shape -- asking for it here raises instead of silently changing this one's)."""
    from .grouped import DequantPlan
    keys, items = _plan_items(state_dict)
    if len({it[0].device for it in items}) > 1:
        raise ValueError("this state dict spans several devices (gguf_sd_loader(devices=[...])): use state_dict_sharded_plan()")
    return DequantPlan(items, out_dtype=dtype, dequant_dtype=dequant_dtype), keys


def state_dict_sharded_plan(state_dict, dtype=torch.float16, dequant_dtype=None):
    """The same for a state dict whose tensors live on SEVERAL GPUs (``gguf_sd_loader(devices=[...])``): one ``grouped.ShardedPlan`` -- a
    ``DequantPlan`` per device, every launch enqueued by the calling thread.  Returns (plan, keys): ``keys`` in the state dict's order and
    ``plan.outputs_in_order()[i]`` the dense tensor of ``keys[i]`` (on that tensor's own device)."""
    from .grouped import ShardedPlan
    keys, items = _plan_items(state_dict)
    on = {}
    for i, it in enumerate(items):
        on.setdefault(it[0].device, []).append(i)
    shards = [(d, [items[i] for i in ix]) for d, ix in on.items()]
    return ShardedPlan(shards, out_dtype=dtype, dequant_dtype=dequant_dtype, indices=list(on.values())), keys
